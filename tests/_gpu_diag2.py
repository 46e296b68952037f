import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'deprecated-lame-mirror_amd'))
import helpers, lamehip
from lamehip.types import struct_diff
orc = helpers.Oracle()
for rep in range(2):
  for name in sys.argv[1:]:
    g, pcm = helpers.load_golden(name)
    enc = lamehip.Encoder(*helpers.golden_settings(g))
    for B in (1, 2):
        b = lamehip.Batch(enc, B, pcm.shape[1] + 16)
        for s in range(B): b.set_pcm(s, pcm[0], pcm[1])
        b.encode()
        want = orc.encode_frames(enc.config(), enc.tables(), pcm)
        for s in range(B):
            got = b.get_frames(s)
            nbad = 0
            for f in range(len(got)):
                d = struct_diff(want[f], got[f])
                if d:
                    nbad += 1
                    if nbad <= 2: print(name, 'B', B, 's', s, 'frame', f, d[:8])
            print(rep, name, 'B', B, 'stream', s, 'frames', len(got), 'bad', nbad)
        b.close()
    enc.close()
