#!/usr/bin/env python
"""Development aid (GPU box): the device-packed pipeline of bench.end_to_end (nb batch objects, upload + encode + fetch per
round) for a kernel / memory-copy trace:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python tools/e2e_trace.py [objects] [seconds]
tools/e2e_timeline.py <dir> prints the timeline."""
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

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
B, sr, rounds = 1024, 44100, 6
os.environ.setdefault("LAMEHIP_PINNED_MAX_MB", "16384")
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
        b.mark_pcm(s)
    b.set_device_packing()
    b.encode(sync=False)
    b.fetch()
    b.bytes_view(0)
    objs.append(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(rounds):
    b = objs[r % nb]
    if r >= nb:
        b.bytes_view(0)
    for s in range(B):
        b.mark_pcm(s)
    b.upload()
    b.encode(sync=False)
    b.fetch()
for b in objs:
    b.bytes_view(0)
dt = time.perf_counter() - t0
print("%d objects: %.1f ms per batch, %.0f x" % (nb, dt / rounds * 1e3, rounds * B * seconds / dt))
