"""Per-stage parity (SURVEY.md 8(c) "G2"): what every stage of a frame hands on, device against the compiled
reference, bit for bit -- MDCT spectra, the psycho-acoustic band energies / thresholds (tolerance: 0 ulp), the allowed
noise, the smoothed perceptual entropies and the CBR bit budgets.  The reference side is committed as hashes
(tests/golden/stages_*.npz, made by tests/golden/make_stage_golden.py from oracle/_ref); the device side comes from
the LH_DEBUG_DUMP build of the library (make -C deprecated-lame-mirror_amd/csrc dump -> liblamehip_dump.so, a test
tool like the profiling build: its LhStreamState carries a tail the kernel fills per frame).  A red payload test
says THAT a frame differs; this one says in which stage."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

DUMP_LIB = os.path.join(helpers.ROOT, "deprecated-lame-mirror_amd", "lamehip", "liblamehip_dump.so")
NAMES = ("cbr128_js_44k", "vbr2_js_44k")
TAIL_FLOATS = 2 * 2 * 576 + 2 * 2 * 40 + 2 * (2 * 2 * 64) + 4 + 4 + 1 + 3

# runs in a child process: the binding loads ONE library per process (LAMEHIP_LIB)
CHILD = r'''
import ctypes as C, hashlib, json, sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(pkg)r)
import helpers, lamehip
def sha(a): return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
name = sys.argv[1]
g, pcm = helpers.load_golden(name)
enc = lamehip.Encoder(require_device=True, **helpers.golden_encoder_kwargs(g))
size = enc.lib.lamehip_abi_sizeof(4)
buf = C.create_string_buffer(size)
tail = %(tail)d
left, right = pcm[0], pcm[1]
n = len(left)
zeros = np.zeros(1152, np.int16)
rows, last = [], 0
for i in range((n + 1151) // 1152 + 3):
    a, b = left[1152 * i:1152 * i + 1152], right[1152 * i:1152 * i + 1152]
    if len(a) < 1152:
        a = np.concatenate([a, zeros[:1152 - len(a)]]); b = np.concatenate([b, zeros[:1152 - len(b)]])
    enc.encode(a, b)
    assert enc.lib.lamehip_get_state(enc.h, buf, size) == size
    fn = enc.lib.lame_get_frameNum(enc.h)
    if fn == last:
        continue
    assert fn == last + 1
    last = fn
    t = np.frombuffer(buf.raw[size - 4 * tail:], dtype=np.float32)
    o = 0
    xr = t[o:o + 2304].reshape(2, 2, 576); o += 2304
    xmin = t[o:o + 160].reshape(2, 2, 40)[:, :, :39]; o += 160
    en = t[o:o + 256].reshape(2, 2, 64)[:, :, :61]; o += 256
    thm = t[o:o + 256].reshape(2, 2, 64)[:, :, :61]; o += 256
    pe = t[o:o + 4]; o += 4
    targ = t[o:o + 5].view(np.int32); o += 5
    rows.append({"xr": sha(xr), "en": sha(en), "thm": sha(thm), "xmin": sha(xmin), "pe": sha(pe), "targ": sha(targ)})
print("STAGES " + json.dumps(rows))
'''


def test_stage_fixtures_are_committed_and_the_dump_build_exists():
    """CPU: both fixtures load, cover every group, and the dump library (built by __graft_entry__.build) carries the
    larger LhStreamState."""
    import ctypes as C
    for name in NAMES:
        z = np.load(os.path.join(helpers.ROOT, "tests", "golden", "stages_%s.npz" % name))
        assert int(z["nframes"]) >= 40
        for k in ("xr", "en", "thm", "xmin", "pe", "targ"):
            assert len(z[k]) == int(z["nframes"])
    assert os.path.exists(DUMP_LIB), "make -C deprecated-lame-mirror_amd/csrc dump"
    dump = C.CDLL(DUMP_LIB)
    prod = C.CDLL(os.path.join(helpers.ROOT, "deprecated-lame-mirror_amd", "lamehip", "liblamehip.so"))
    assert dump.lamehip_abi_sizeof(4) == prod.lamehip_abi_sizeof(4) + 4 * TAIL_FLOATS


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_every_stage_of_every_frame_matches_the_reference(name):
    z = np.load(os.path.join(helpers.ROOT, "tests", "golden", "stages_%s.npz" % name))
    code = CHILD % {"tests": os.path.join(helpers.ROOT, "tests"),
                    "pkg": os.path.join(helpers.ROOT, "deprecated-lame-mirror_amd"), "tail": TAIL_FLOATS}
    out = subprocess.run([sys.executable, "-c", code, name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900, env=dict(os.environ, LAMEHIP_LIB=DUMP_LIB))
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("STAGES ")]
    assert line, out.stdout[-2000:]
    rows = json.loads(line[0][7:])
    assert len(rows) == int(z["nframes"])
    groups = ("xr", "en", "thm", "xmin", "pe") + (("targ",) if name.startswith("cbr") else ())
    for k in groups:
        bad = [f for f in range(len(rows)) if rows[f][k] != str(z[k][f])]
        assert not bad, "stage output `%s' differs from the reference in frames %s" % (k, bad[:8])


# the same through the split pipeline (analysis kernels + sub-band kernel + the encode kernel that starts from their output):
# a batch of one stream fed 1152 samples at a time, one launch per round -- the dump tail holds the launch's last frame
CHILD_SPLIT = r"""
import ctypes as C, hashlib, json, sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(pkg)r)
import helpers, lamehip
def sha(a): return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
name = sys.argv[1]
g, pcm = helpers.load_golden(name)
enc = lamehip.Encoder(require_device=True, **helpers.golden_encoder_kwargs(g))
size = enc.lib.lamehip_abi_sizeof(4)
buf = C.create_string_buffer(size)
tail = %(tail)d
left, right = pcm[0], pcm[1]
n = len(left)
b = lamehip.Batch(enc, 1, n + 4608)
rows, done = {}, 0
def look(k):
    global done
    if k <= 0:
        return
    done += k
    split, _ = b.kernel_parts_ms()
    assert split, "the launch did not go through the split pipeline"
    assert enc.lib.lamehip_batch_get_state(b.b, 0, buf, size) == size
    t = np.frombuffer(buf.raw[size - 4 * tail:], dtype=np.float32)
    o = 0
    xr = t[o:o + 2304].reshape(2, 2, 576); o += 2304
    xmin = t[o:o + 160].reshape(2, 2, 40)[:, :, :39]; o += 160
    en = t[o:o + 256].reshape(2, 2, 64)[:, :, :61]; o += 256
    thm = t[o:o + 256].reshape(2, 2, 64)[:, :, :61]; o += 256
    pe = t[o:o + 4]; o += 4
    targ = t[o:o + 5].view(np.int32); o += 5
    rows[done - 1] = {"xr": sha(xr), "en": sha(en), "thm": sha(thm), "xmin": sha(xmin), "pe": sha(pe), "targ": sha(targ)}
for i in range(0, n, 1152):
    b.append(0, left[i:i + 1152], right[i:i + 1152])
    k = b.encode_available()
    b.sync()
    look(k)
k = b.finish()
b.sync()
look(k)
print("STAGES " + json.dumps({"frames": done, "rows": {str(f): r for f, r in rows.items()}}))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_every_stage_matches_the_reference_through_the_split_pipeline(name):
    """The analysis kernels' output at 0 ulp: the frames observed (the last frame of every launch; the flush encodes the
    stream's last few frames in one launch) carry the reference's spectra, band energies / thresholds, allowed noise,
    perceptual entropies and bit budgets."""
    z = np.load(os.path.join(helpers.ROOT, "tests", "golden", "stages_%s.npz" % name))
    code = CHILD_SPLIT % {"tests": os.path.join(helpers.ROOT, "tests"),
                          "pkg": os.path.join(helpers.ROOT, "deprecated-lame-mirror_amd"), "tail": TAIL_FLOATS}
    env = dict(os.environ, LAMEHIP_LIB=DUMP_LIB)
    env.pop("LAMEHIP_FUSED", None)
    out = subprocess.run([sys.executable, "-c", code, name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("STAGES ")]
    assert line, out.stdout[-2000:]
    res = json.loads(line[0][7:])
    # (the fixture follows the handle path for a few calls of zeros past the stream's end: one frame more than the flush makes)
    assert res["frames"] in (int(z["nframes"]), int(z["nframes"]) - 1)
    rows = {int(f): r for f, r in res["rows"].items()}
    assert len(rows) >= int(z["nframes"]) - 4
    groups = ("xr", "en", "thm", "xmin", "pe") + (("targ",) if name.startswith("cbr") else ())
    for k in groups:
        bad = [f for f in sorted(rows) if rows[f][k] != str(z[k][f])]
        assert not bad, "stage output `%s' differs from the reference in frames %s" % (k, bad[:8])
