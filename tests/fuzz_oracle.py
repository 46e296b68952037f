#!/usr/bin/env python
"""Randomised hunt on the CPU (test tool, not collected by pytest; needs oracle/_ref): the oracle restatement against
the compiled reference, frame by frame and byte by byte, over the stress signals of tests/test_gpu_parity.py and the
settings of tests/fuzz_gpu.py.
Usage: python tests/fuzz_oracle.py [streams per setting] [seconds] [seed0] [cbr|vbr|abr|old]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import lamehip  # noqa: E402
from lamehip.types import struct_diff  # noqa: E402
import fuzz_gpu as fz  # noqa: E402
import test_gpu_parity as tg  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    which = sys.argv[4] if len(sys.argv) > 4 else "cbr"
    ref, orc = helpers.Reference(), helpers.Oracle()
    bad = tot = 0
    t0 = time.time()
    lsf = which.startswith("lsf")
    old_loop = which in ("old", "lsfold")
    for sr, br, mode, q in {"vbr": fz.VBR_SETTINGS, "abr": fz.ABR_SETTINGS, "old": fz.OLD_SETTINGS, "lsf": fz.LSF_SETTINGS,
                            "lsfold": fz.LSF_OLD_SETTINGS}.get(which, fz.SETTINGS):
        kw = dict(mode=mode, quality=q)
        rkw = dict(mode=-1 if mode is None else mode, quality=-1 if q is None else q)
        if lsf:
            kw.update(out_samplerate=sr)
            rkw.update(out_samplerate=sr)
        if br >= 1000:
            kw.update(abr=br - 1000, out_samplerate=sr)
            rkw.update(abr=br - 1000, out_samplerate=sr)
        elif br <= 0:
            kw.update(vbr_q=-br, out_samplerate=sr if (-br >= 7 or lsf) else 0, vbr_mode=2 if old_loop else 4)
            rkw.update(vbr_q=-br, out_samplerate=sr if (-br >= 7 or lsf) else 0, vbr_mode=2 if old_loop else 4)
        enc = lamehip.Encoder(sr, max(br, 0) if br < 1000 else 0, require_device=False, **kw)
        cfg, tab = enc.config(), enc.tables()
        n = int(sr * secs)
        for i in range(B):
            x = tg._stress_signal(seed0 + i, n - 13 * (i % 31), sr)
            mp3, nf, frames, rcfg, rtab = ref.encode(x, sr, max(br, 0) if br < 1000 else 0, max_frames=4096, **rkw)
            want = orc.encode_frames(cfg, tab, x)
            helpers.normalize_tables(want)
            tot += 1
            d = None
            if len(want) != nf:
                d = ("frames", nf, len(want))
            else:
                for f in range(nf):
                    dd = struct_diff(frames[f], want[f])
                    if dd:
                        d = (f, dd[:3])
                        break
                if d is None and helpers.pack_frames(enc.lib, cfg, tab, want) != mp3:
                    d = ("bytes",)
            if d is not None:
                bad += 1
                print("MISMATCH", (sr, br, mode, q), "seed", seed0 + i, d, flush=True)
        enc.close()
        print("setting", (sr, br, mode, q), "streams", tot, "bad", bad, "%.0fs" % (time.time() - t0), flush=True)
    print("TOTAL streams", tot, "BAD", bad)


if __name__ == "__main__":
    main()
