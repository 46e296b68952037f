#!/usr/bin/env python
"""Settings sweep on the CPU (test tool, not collected by pytest; needs oracle/_ref, i.e. the
reference sources): rate control x sample rate x channel mode x quality level, awkward signals,
bytes of (oracle frames -> host packer) against the compiled reference.
Usage: python tests/sweep_oracle_vs_ref.py [cbr|vbr|abr|all] [signals_per_setting]
  cbr: every MPEG-1 bit rate;  vbr: -V0..-V9 (vbr_mtrh);  abr: means incl. values between the table rates;
  channel modes: stereo, joint stereo, dual channel, mono input, stereo mixed down to mono."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers  # noqa: E402
import lamehip  # noqa: E402
import test_gpu_parity as tg  # noqa: E402

# (label, MPEG mode or None, input channels)
MODES = [("st", 0, 2), ("js", 1, 2), ("dual", 2, 2), ("mono", None, 1), ("mix", 3, 2)]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "cbr"
    nsig = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ref, orc = helpers.Reference(), helpers.Oracle()
    grid = []
    if what in ("cbr", "all"):
        grid += [(dict(brate=br), q) for br in (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320)
                 for q in (0, 2, 3, 5, 7, 9)]
    if what in ("vbr", "all"):
        grid += [(dict(vbr_q=v), q) for v in range(10) for q in (0, 5, 7)]
    if what in ("abr", "all"):
        grid += [(dict(abr=kb), q) for kb in (64, 96, 100, 128, 150, 192, 215, 256, 320) for q in (0, 3, 5, 7)]
    bad = tot = unsup = 0
    for sr in (32000, 44100, 48000):
        for kw, q in grid:
            for label, mode, nch in MODES:
                kw2 = dict(kw)
                if kw.get("vbr_q", 0) >= 7:
                    kw2["out_samplerate"] = sr          # -V7.. would resample unless the rate is pinned
                try:
                    enc = lamehip.Encoder(sr, mode=mode, quality=q, require_device=False, channels=nch, **kw2)
                except RuntimeError:
                    unsup += 1
                    continue
                cfg, tab = enc.config(), enc.tables()
                if cfg.samplerate != sr:
                    # the reference converts the input rate first: tests/sweep_resample.py covers those
                    unsup += 1
                    enc.close()
                    continue
                n = int(sr * 1.2)
                for k in range(nsig):
                    x = tg._stress_signal(sum(kw.values()) + 7 * k + q + nch, n - 41 * k, sr)
                    if nch == 1:
                        x = np.stack([x[0], x[0]])
                    mp3o = helpers.pack_frames(enc.lib, cfg, tab, orc.encode_frames(cfg, tab, x))
                    rkw = dict(kw2)
                    br = rkw.pop("brate", 0)
                    mp3r = ref.encode(x, sr, br, -1 if mode is None else mode, q, channels=nch, **rkw)[0]
                    tot += 1
                    if mp3o != mp3r:
                        bad += 1
                        print("MISMATCH", sr, kw, label, q, k, len(mp3o), len(mp3r), flush=True)
                enc.close()
        print("rate", sr, "checked", tot, "bad", bad, flush=True)
    print("checked", tot, "bad", bad, "unsupported settings", unsup)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
