#!/usr/bin/env python
"""Would placing streams on CUs by their cost pay?  Needs an LH_PROF build WITHOUT the issue-priority code (-DLH_NO_PRIO), so
that a stream's cycles show its own cost:  LAMEHIP_LIB=.../liblamehip_profnp.so python tools/placement_sim.py [seconds] [prefix_seconds]
Prints how well the search-call counters of a prefix predict a stream's cost, and the spread of CU-group means for the
dispatcher's grouping (workgroup id mod 256) against a snake placement by predicted cost."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lamehip  # noqa: E402
import bench  # noqa: E402


def main():
    B = 1024
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    pre = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    sr = 44100
    n = int(sr * secs)
    dev = torch.device("cuda", 0)
    enc = lamehip.Encoder(sr, 128)
    b = lamehip.Batch(enc, B, n)
    pcm = bench.synth_on_device(torch, B, n, sr, 0, dev)
    torch.cuda.synchronize()
    ssz = enc.lib.lamehip_abi_sizeof(4)
    NP = 44

    def run(length):
        b.reset()
        for s in range(B):
            b.set_pcm_device(s, pcm[s, 0].data_ptr(), pcm[s, 1].data_ptr(), length)
        b.encode()
        out = np.zeros((B, 2, NP))
        for s in range(B):
            buf = C.create_string_buffer(ssz)
            assert enc.lib.lamehip_batch_get_state(b.b, s, buf, ssz) == ssz
            out[s] = np.frombuffer(buf.raw[-2 * NP * 8:], dtype=np.uint64).reshape(2, NP)
        return out, b.kernel_ms()

    full, ms = run(n)
    part, ms_p = run(int(sr * pre))
    t = full[:, :, 0].max(axis=1)
    cls = np.zeros(B, dtype=int)
    cls[256:768] = 1
    cls[768:] = 2
    fac = np.array([t[cls == k].mean() for k in range(3)]) / t.mean()
    tn = t / fac[cls]
    print("kernel %.1f ms; class factors %s; content spread of a stream's cycles: sd %.2f %%, max/mean %.3f"
          % (ms, np.round(fac, 3), 100 * tn.std() / tn.mean(), tn.max() / tn.mean()))

    def model(x):
        return (3.9 * x[:, :, 10] + 2.4 * x[:, :, 13]).max(axis=1)
    for name, m in (("counters of the whole stream", model(full)), ("counters of the first %.1f s" % pre, model(part))):
        print("  %s: correlation with the cycles %.3f" % (name, np.corrcoef(m, tn)[0, 1]))

    def spread(assign):      # assign[s] = group of stream s
        g = np.array([tn[assign == k].mean() for k in range(256)])
        return g.max() / tn.mean()
    ids = np.arange(B)
    print("  CU groups as dispatched (id mod 256): slowest group / mean = %.4f" % spread(ids % 256))
    for name, m in (("whole-stream counters", model(full)), ("prefix counters", model(part)), ("the cycles themselves", tn)):
        order = np.argsort(-m)
        assign = np.zeros(B, dtype=int)
        for r, s in enumerate(order):
            j, pos = divmod(r, 256)
            assign[s] = pos if j % 2 == 0 else 255 - pos
        print("  snake placement by %s: slowest group / mean = %.4f" % (name, spread(assign)))


if __name__ == "__main__":
    main()
