#!/usr/bin/env python
"""How evenly do the streams of a bench batch finish?  A launch ends with its slowest stream; the streams of bench.py's
batch differ in content, so they differ in search iterations.  With the LH_PROF build
    LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stream_balance.py [streams] [seconds]
prints the distribution of per-stream cycle totals (both waves), then the kernel time of batches made of copies of the
fastest / median / slowest stream."""
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    sr = 44100
    n = int(sr * secs)
    dev = torch.device("cuda", 0)
    enc = lamehip.Encoder(sr, 128)
    b = lamehip.Batch(enc, B, n)
    pcm = bench.synth_on_device(torch, B, n, sr, 0, dev)
    torch.cuda.synchronize()

    def run(sel):
        b.reset()
        for s in range(B):
            k = sel(s)
            b.set_pcm_device(s, pcm[k, 0].data_ptr(), pcm[k, 1].data_ptr(), n)
        b.encode()
        b.reset()
        for s in range(B):
            k = sel(s)
            b.set_pcm_device(s, pcm[k, 0].data_ptr(), pcm[k, 1].data_ptr(), n)
        b.encode()
        return b.kernel_ms()

    ms = run(lambda s: s)
    ssz = enc.lib.lamehip_abi_sizeof(4)
    NP = 44
    tot = np.zeros((B, 2))
    have_prof = "prof" in os.environ.get("LAMEHIP_LIB", "")
    if have_prof:
        for s in range(B):
            buf = C.create_string_buffer(ssz)
            assert enc.lib.lamehip_batch_get_state(b.b, s, buf, ssz) == ssz
            prof = np.frombuffer(buf.raw[-2 * NP * 8:], dtype=np.uint64).reshape(2, NP)
            tot[s] = prof[:, 0]
        t = tot.max(axis=1)
        print("distinct streams: kernel %.2f ms; per-stream cycles (max of the two waves): min %.4g mean %.4g max %.4g, mean/max %.3f"
              % (ms, t.min(), t.mean(), t.max(), t.mean() / t.max()))
        print("  percentiles 1/10/50/90/99: " + " ".join("%.4g" % np.percentile(t, p) for p in (1, 10, 50, 90, 99)))
        print("  implied clock if the slowest stream spans the kernel: %.3f GHz" % (t.max() / (ms * 1e-3) / 1e9))
        order = np.argsort(t)
        picks = [("fastest", int(order[0])), ("median", int(order[B // 2])), ("slowest", int(order[-1]))]
        # position of the slow streams in the grid (XCD = workgroup id mod 8)
        slow = order[-32:]
        print("  32 slowest streams: ids %s" % sorted(int(x) for x in slow))
        print("  mean cycles by workgroup id mod 8: " + " ".join("%.4g" % t[i::8].mean() for i in range(8)))
    else:
        print("distinct streams: kernel %.2f ms (no LH_PROF build: copies of streams 0, 1, 2)" % ms)
        picks = [("stream 0", 0), ("stream 1", 1), ("stream 2", 2)]
    for name, k in picks:
        m = run(lambda s, k=k: k)
        extra = ""
        if have_prof:
            for s in range(B):
                buf = C.create_string_buffer(ssz)
                enc.lib.lamehip_batch_get_state(b.b, s, buf, ssz)
                tot[s] = np.frombuffer(buf.raw[-2 * NP * 8:], dtype=np.uint64).reshape(2, NP)[:, 0]
            t2 = tot.max(axis=1)
            extra = "; cycles min %.4g mean %.4g max %.4g" % (t2.min(), t2.mean(), t2.max())
            extra += "\n   mean by id // 64: " + " ".join("%.3g" % t2[i * 64:(i + 1) * 64].mean() for i in range(B // 64))
            extra += "\n   wave 0 / wave 1 mean: %.4g %.4g" % (tot[:, 0].mean(), tot[:, 1].mean())
        print("%d copies of the %s stream (%d): kernel %.2f ms = %.3f of the distinct batch%s" % (B, name, k, m, m / ms, extra))


if __name__ == "__main__":
    main()
