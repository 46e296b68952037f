#!/usr/bin/env python
"""Development aid (GPU box, LH_PROF build): two batch objects launched back to back on their own streams -- the kernels'
durations and the per-stream cycle totals (s_memtime at 100 MHz) of each: did the second launch's streams run slowly, or
were they resident in two rounds?"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lamehip  # noqa: E402
import bench  # noqa: E402

B, sr, seconds = 1024, 44100, float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
n = int(sr * seconds)
dev = torch.device("cuda", 0)
enc = lamehip.Encoder(sr, 128)
pcm = bench.synth_on_device(torch, B, n, sr, 0, dev)
objs = []
for k in range(2):
    b = lamehip.Batch(enc, B, n)
    for s in range(B):
        b.set_pcm_device(s, pcm[s, 0].data_ptr(), pcm[s, 1].data_ptr(), n)
    b.encode(sync=True)
    objs.append(b)
ssz = enc.lib.lamehip_abi_sizeof(4)
NP = 44


def totals(b):
    t = np.zeros((B, 2))
    for s in range(B):
        buf = C.create_string_buffer(ssz)
        assert enc.lib.lamehip_batch_get_state(b.b, s, buf, ssz) == ssz
        t[s] = np.frombuffer(buf.raw[-2 * NP * 8:], dtype=np.uint64).reshape(2, NP)[:, 0]
    return t.max(axis=1)


for trial in range(2):
    for b in objs:
        b.reset()
        for s in range(B):
            b.set_pcm_device(s, pcm[s, 0].data_ptr(), pcm[s, 1].data_ptr(), n)
    torch.cuda.synchronize()
    for b in objs:
        b.encode(sync=False)
    for b in objs:
        b.sync()
    for k, b in enumerate(objs):
        t = totals(b)
        print("trial %d object %d: kernel %.1f ms; per-stream cycles min %.4g mean %.4g max %.4g" % (trial, k, b.kernel_ms(), t.min(), t.mean(), t.max()), flush=True)
