#!/usr/bin/env python
"""Development aid (GPU box): first differing fields device vs oracle for the old VBR loop."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import lamehip  # noqa: E402
from lamehip.types import struct_diff  # noqa: E402

orc = helpers.Oracle()
for name in sys.argv[1:] or ["vbrold2_js_44k"]:
    g, pcm = helpers.load_golden(name)
    enc = lamehip.Encoder(**helpers.golden_encoder_kwargs(g))
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
            if bad <= 3:
                print(name, "frame", f, [(x[0], x[1], x[2]) if not isinstance(x[1], list) and len(x) > 2 else (x[0], x[1]) for x in d[:10]])
    print(name, "%d of %d frames differ" % (bad, len(got)))
    b.close()
    enc.close()
