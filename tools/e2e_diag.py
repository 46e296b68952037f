#!/usr/bin/env python
"""Development aid (GPU box): the device-packed pipeline of bench.end_to_end round by round (wall clock and the kernel's
own duration per batch), with the steps that differ from tools/e2e_probe.py switched one at a time."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import lamehip  # noqa: E402
import bench  # noqa: E402

os.environ.setdefault("LAMEHIP_PINNED_MAX_MB", "16384")
B, sr, seconds, rounds, nb = 1024, 44100, 30.0, 8, 2
n = int(sr * seconds)
dev = torch.device("cuda", 0)
enc = lamehip.Encoder(sr, 128)
host = bench.synth_on_device(torch, B, n, sr, 777, dev).cpu().numpy()
objs = []
for k in range(nb):
    b = lamehip.Batch(enc, B, n)
    b.pcm_host()[:, :, :n] = host
    for s in range(B):
        b.set_length(s, n)
    objs.append(b)
which = [0, 1, B // 2, B - 1]


def mark(b):
    for s in range(B):
        b.mark_pcm(s)


for variant in (os.environ.get("E2E_VARIANTS", "resident_first+bytes_view,bytes_view,sync").split(",")):
    if variant.startswith("resident_first"):
        b = objs[0]
        b.set_device_packing(False)
        mark(b)
        b.encode(sync=True)
        for _ in range(3):
            b.encode(sync=True)
    for b in objs:
        b.set_device_packing()
        mark(b)
        b.encode(sync=("seq" in variant))       # "seq": the objects' first launches one after the other
        b.fetch()
        if "seq" in variant:
            b.bytes_view(0)
    for b in objs:
        b.bytes_view(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stamps = []
    for r in range(rounds):
        b = objs[r % nb]
        if r >= nb:
            if "bytes_view" in variant:
                sum(len(b.bytes_view(s)) for s in which)
            else:
                b.sync()
            stamps.append((r - nb, round(time.perf_counter() - t0, 3), round(b.kernel_ms(), 1)))
        mark(b)
        b.upload()
        b.encode(sync=False)
        b.fetch()
    for b in objs:
        b.bytes_view(0)
    dt = time.perf_counter() - t0
    print(variant, "%.1f ms per batch, %.0f x" % (dt / rounds * 1e3, rounds * B * seconds / dt), stamps, flush=True)
