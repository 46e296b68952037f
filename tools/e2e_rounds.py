#!/usr/bin/env python
"""Development aid (GPU box): bench.end_to_end at several round counts -- is a loss per batch or per run?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import lamehip  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
enc = lamehip.Encoder(44100, 128)
for rounds, nb, secs in ((6, 2, 30.0), (12, 2, 30.0), (6, 3, 30.0), (6, 2, 10.0)):
    r = bench.end_to_end(torch, lamehip, enc, 1024, 44100, dev, seconds=secs, rounds=rounds, nbatch=nb)
    print(rounds, nb, secs, "host", r["value"], "dev", r["device_packed"]["value"], "resident", r["hbm_resident_same_sample"],
          r["device_packed"]["one_batch_alone_s"], flush=True)
