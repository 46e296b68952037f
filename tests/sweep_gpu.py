#!/usr/bin/env python
"""Settings sweep on the GPU box (test tool, not collected by pytest): every MPEG-1 bit rate x
sample rate x stereo mode x a set of quality levels that lame_init_params accepts, a few
awkward signals each, HIP payload against the CPU oracle frame by frame.
Usage: python tests/sweep_gpu.py [streams_per_setting] [seconds] [cbr|vbr|abr|all] [channels]
"vbr": vbr_mtrh -V0..-V9 x sample rate x stereo mode x quality 0 / 5 / 7 instead of the CBR grid;
"abr": ABR means (incl. values between the table rates) x sample rate x mode x quality 0 / 3 / 5 / 7."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import lamehip  # noqa: E402
from lamehip.types import struct_diff  # noqa: E402
import test_gpu_parity as tg  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
    what = sys.argv[3] if len(sys.argv) > 3 else "cbr"
    nch = int(sys.argv[4]) if len(sys.argv) > 4 else 2      # 1: mono input (the mode axis is skipped)
    orc = helpers.Oracle()
    bad = tot = nset = unsup = 0
    t0 = time.time()
    cbr_grid = [(br, q) for br in (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320)
                for q in (0, 2, 3, 5, 7, 9)] if what in ("cbr", "all") else []
    vbr_grid = [(-vq, q) for vq in range(10) for q in (0, 5, 7)] if what in ("vbr", "all") else []
    abr_grid = [(1000 + kb, q) for kb in (96, 100, 112, 128, 150, 160, 192, 215, 256, 320)
                for q in (0, 3, 5, 7)] if what in ("abr", "all") else []
    for sr in (32000, 44100, 48000):
        for br, q in cbr_grid + vbr_grid + abr_grid:    # br <= 0: vbr_mtrh at quality -br; >= 1000: ABR
            for mode in ((0, 1) if nch == 2 else (None,)):
                for _once in (0,):
                    try:
                        if br >= 1000:
                            enc = lamehip.Encoder(sr, mode=mode, quality=q, abr=br - 1000, channels=nch)
                        elif br > 0:
                            enc = lamehip.Encoder(sr, br, mode, q, channels=nch)
                        else:
                            enc = lamehip.Encoder(sr, mode=mode, quality=q, vbr_q=-br,
                                                  out_samplerate=sr if -br >= 7 else 0, channels=nch)
                    except RuntimeError:
                        unsup += 1
                        continue
                    cfg, tab = enc.config(), enc.tables()
                    if cfg.samplerate != sr:
                        # the reference would resample (a lower output rate for this bit rate's lowpass): the handle API's
                        # job (lh_resample.c, tests/test_resample.py), not the batch path's -- not a setting of this sweep
                        unsup += 1
                        enc.close()
                        continue
                    nset += 1
                    n = int(sr * secs)
                    pcms = [tg._stress_signal(abs(br) + q + 8 * i + i, n - 29 * i, sr) for i in range(B)]
                    b = lamehip.Batch(enc, B, n)
                    b.set_device_packing()      # the device bit packer is checked against the host packer too
                    if nch == 1:
                        pcms = [np.stack([x[0], x[0]]) for x in pcms]
                    for s, x in enumerate(pcms):
                        b.set_pcm(s, x[0], x[1])
                    b.encode()
                    for s, x in enumerate(pcms):
                        got = b.get_frames(s)
                        want = orc.encode_frames(cfg, tab, x)
                        tot += 1
                        ok = len(got) == len(want)
                        for f in range(len(got) if ok else 0):
                            d = struct_diff(want[f], got[f])
                            if d:
                                ok = False
                                print("MISMATCH", (sr, br, mode, q), "stream", s, "frame", f, d[:3], flush=True)
                                break
                        if ok and b.get_bytes(s) != b.pack(s):
                            ok = False
                            print("BYTES MISMATCH (device packer)", (sr, br, mode, q), "stream", s, flush=True)
                        bad += (not ok)
                    b.close()
                    enc.close()
        print("rate", sr, "settings", nset, "streams", tot, "bad", bad, "%.0fs" % (time.time() - t0), flush=True)
    print("TOTAL settings", nset, "unsupported", unsup, "streams", tot, "BAD", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
