#!/usr/bin/env python
"""Randomised parity hunt on the GPU box (test tool, not collected by pytest): many short streams
of awkward signals per setting, HIP payload against the CPU oracle, frame by frame.
Usage: python tests/fuzz_gpu.py [streams] [seconds] [seed0] [cbr|vbr|abr|old]   (old: the old VBR loop, vbr_rh)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import lamehip  # noqa: E402
from lamehip.types import struct_diff  # noqa: E402
import test_gpu_parity as tg  # noqa: E402

SETTINGS = [(44100, 128, None, None), (44100, 128, None, 0), (48000, 320, 1, None), (32000, 96, None, None),
            (44100, 192, 0, None), (44100, 256, None, 2), (48000, 128, None, 7), (32000, 320, 0, 5),
            (44100, 112, None, 5), (32000, 128, 0, 9), (44100, 224, 1, 4), (44100, 160, None, 3),
            (48000, 192, None, 1), (44100, 320, None, 6), (32000, 160, 1, 8)]


# bit rate <= 0: vbr_mtrh at quality -rate
VBR_SETTINGS = [(44100, -2, None, None), (44100, 0, None, None), (48000, -4, 1, None), (32000, -5, None, None),
                (44100, -6, 0, None), (44100, -1, None, 5), (48000, -3, None, 5), (32000, -2, 0, None),
                (44100, -9, None, None), (48000, -7, None, None), (44100, -8, 0, 5), (48000, 0, 0, None)]


# the same code for the old VBR loop (vbr_rh); -V 7.. would pick an MPEG-2 output rate
OLD_SETTINGS = [(44100, -2, None, None), (44100, 0, None, None), (48000, -4, 1, None), (32000, -5, None, None),
                (44100, -6, 0, None), (44100, -1, None, 5), (48000, -3, None, 6), (32000, -2, 0, None),
                (48000, -1, None, 0), (44100, -3, 3, None), (48000, 0, 0, 2), (44100, -4, None, 1)]


# bit rate >= 1000: ABR at a mean of rate - 1000 kb/s
ABR_SETTINGS = [(44100, 1128, None, None), (48000, 1190, 1, None), (32000, 1096, None, 5), (44100, 1256, 0, 2),
                (44100, 1080, None, 7), (48000, 1313, 0, None)]


# MPEG-2 / 2.5 output rates (one granule per frame), all four rate controls in one list: the output rate is pinned to the
# input rate, so that the checker is fed the same PCM; "lsf" on the command line
LSF_SETTINGS = [(22050, 64, None, None), (24000, 80, None, 2), (16000, 32, None, None), (22050, 160, 0, None),
                (24000, 64, None, 5), (16000, 64, 1, 0), (12000, 32, None, None), (11025, 40, None, None),
                (8000, 16, None, None), (8000, 24, 0, 7), (22050, 96, None, 9), (24000, 112, 1, 3),
                (22050, -4, None, None), (16000, -6, None, None), (24000, -2, 0, 5), (12000, -8, None, None), (8000, -9, None, None),
                (22050, 1056, None, None), (16000, 1024, None, 5), (11025, 1032, 0, None), (24000, 1120, 1, 2)]
LSF_OLD_SETTINGS = [(22050, -2, None, None), (24000, -4, None, None), (16000, -5, 0, None), (12000, -6, None, 5), (22050, 0, None, 2)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    orc = helpers.Oracle()
    bad = tot = 0
    t0 = time.time()
    which = sys.argv[4] if len(sys.argv) > 4 else "cbr"
    lsf = which.startswith("lsf")
    old_loop = which in ("old", "lsfold")
    for sr, br, mode, q in {"vbr": VBR_SETTINGS, "abr": ABR_SETTINGS, "old": OLD_SETTINGS, "lsf": LSF_SETTINGS,
                            "lsfold": LSF_OLD_SETTINGS}.get(which, SETTINGS):
        if br >= 1000:
            enc = lamehip.Encoder(sr, 0, mode, q, abr=br - 1000, out_samplerate=sr)    # (no rate change: the checker is fed the same PCM)
        elif br > 0:
            enc = lamehip.Encoder(sr, br, mode, q, out_samplerate=sr if lsf else 0)
        else:
            enc = lamehip.Encoder(sr, mode=mode, quality=q, vbr_q=-br, out_samplerate=sr if (-br >= 7 or lsf) else 0,
                                  vbr_mode=2 if old_loop else 4)
        cfg, tab = enc.config(), enc.tables()
        n = int(sr * secs)
        pcms = [tg._stress_signal(seed0 + i, n - 13 * (i % 31), sr) for i in range(B)]
        b = lamehip.Batch(enc, B, n)
        b.set_device_packing()
        for s, x in enumerate(pcms):
            b.set_pcm(s, x[0], x[1])
        b.encode()
        for s, x in enumerate(pcms):
            got = b.get_frames(s)
            want = orc.encode_frames(cfg, tab, x)
            tot += 1
            if len(got) != len(want):
                bad += 1
                print("LEN MISMATCH", (sr, br, mode, q), "seed", seed0 + s)
                continue
            for f in range(len(got)):
                d = struct_diff(want[f], got[f])
                if d:
                    bad += 1
                    print("MISMATCH", (sr, br, mode, q), "seed", seed0 + s, "frame", f, d[:3], flush=True)
                    break
            else:
                if b.get_bytes(s) != b.pack(s):
                    bad += 1
                    print("BYTES MISMATCH (device packer)", (sr, br, mode, q), "seed", seed0 + s, flush=True)
        b.close()
        enc.close()
        print("setting", (sr, br, mode, q), "streams", tot, "bad", bad, "%.0fs" % (time.time() - t0), flush=True)
    print("TOTAL streams", tot, "BAD", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
