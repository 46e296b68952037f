#!/usr/bin/env python
"""Development aid (GPU box): where does the pipelined end-to-end path lose time?  Times `rounds' batches through
three batch objects with the upload / fetch steps switched on and off."""
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


def main():
    B, sr, rounds, nb = 1024, 44100, 9, int(sys.argv[1]) if len(sys.argv) > 1 else 3
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
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
    for up, fe, pack in ((0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 0, 0)):
        for b in objs:
            b.set_device_packing(bool(pack))
            b.encode(sync=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(rounds):
            b = objs[r % nb]
            if r >= nb:
                b.sync()
            if up:
                for s in range(B):
                    b.mark_pcm(s)
                b.upload()
            b.encode(sync=False)
            if fe and pack:
                b.fetch()
        for b in objs:
            b.sync()
        dt = time.perf_counter() - t0
        print("upload %d fetch %d devpack %d: %.1f ms per batch, %.0f x" % (up, fe, pack, dt / rounds * 1e3, rounds * B * seconds / dt))


if __name__ == "__main__":
    main()
