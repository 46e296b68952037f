#!/usr/bin/env python
"""Development aid (GPU box): encode a short synthetic stream with the library named by LAMEHIP_LIB and
print every field of the first frames that differ from the CPU oracle.
usage: tools/dbg_cmp.py [seconds] [quality ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import lamehip  # noqa: E402
from lamehip.types import struct_diff  # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    quals = [int(a) for a in sys.argv[2:]] or [None]
    orc = helpers.Oracle()
    pcm = helpers.synth_stream(1, int(44100 * secs))
    for q in quals:
        for kw in (dict(brate=128), dict(vbr_q=2)):
            enc = lamehip.Encoder(44100, quality=q, **kw)
            cfg, tab = enc.config(), enc.tables()
            b = lamehip.Batch(enc, 1, pcm.shape[1] + 16)
            b.set_pcm(0, pcm[0], pcm[1])
            b.encode()
            got = b.get_frames(0)
            want = orc.encode_frames(cfg, tab, pcm)
            bad = 0
            for f in range(min(len(got), len(want))):
                d = struct_diff(want[f], got[f])
                if d:
                    bad += 1
                    if bad <= 2:
                        print("q=%s %s frame %d: %r" % (q, kw, f, d))
            print("q=%s %s: %d of %d frames differ (use_best_huffman=%d)" % (q, kw, bad, len(got), cfg.use_best_huffman))
            b.close()
            enc.close()


if __name__ == "__main__":
    main()
