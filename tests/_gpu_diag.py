import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'deprecated-lame-mirror_amd'))
import helpers, lamehip
from lamehip.types import struct_diff
orc = helpers.Oracle()
for name in sys.argv[1:]:
    g, pcm = helpers.load_golden(name)
    enc = lamehip.Encoder(*helpers.golden_settings(g))
    b = lamehip.Batch(enc, 1, pcm.shape[1] + 16)
    b.set_pcm(0, pcm[0], pcm[1]); b.encode()
    got = b.get_frames(0); want = orc.encode_frames(enc.config(), enc.tables(), pcm)
    nbad = 0
    for f in range(len(got)):
        d = struct_diff(want[f], got[f])
        if d:
            nbad += 1
            if nbad <= 4: print(name, 'frame', f, d[:10])
    print(name, 'frames', len(got), 'bad', nbad)
    import numpy as np, ctypes as C
    from lamehip.types import LhFrameOut
    helpers.normalize_tables(got); helpers.normalize_tables(want)
    for f in range(min(3,len(got))):
        a = np.frombuffer(bytes(got[f]), np.uint8); w = np.frombuffer(bytes(want[f]), np.uint8)
        idx = np.nonzero(a != w)[0]
        print(' raw diff offsets frame', f, idx[:20], [(int(a[i]), int(w[i])) for i in idx[:10]], 'golden sha equal:', helpers.frame_sha(got[f]) == str(g['frame_sha256'][f]), helpers.frame_sha(want[f]) == str(g['frame_sha256'][f]))
