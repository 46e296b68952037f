#!/usr/bin/env python
"""Per-stage golden vectors (SURVEY.md 8(c) "G2") from the COMPILED REFERENCE (oracle/_ref/libref_harness.so; build
container only -- /root/reference does not exist on the GPU box).  For two of the committed goldens' settings and PCM
the reference encodes call by call (1152 samples per lame_encode_buffer); after every call that produced a frame this
script hashes what the reference's stages handed on for that frame:

  xr     l3_side.tt[gr][ch].xr as the frame left it (MDCT spectra after the mid/side rotation and the short-block
         reordering, newmdct.c:944-1039, quantize.c:2006, :226-346)               [2][2][576] float32
  en/thm the band energies / masking thresholds calc_xmin was given (psymodel.c:1397 outputs, one granule late as
         the reference hands them on)                                              [2][2][61] float32 each
  xmin   what calc_xmin returned (quantize_pvt.c:589)                              [2][2][39] float32
  pe     the smoothed perceptual entropies on_pe was given (encoder.c:489-518)     [2][2] float32
  targ   the bit budgets after on_pe / reduce_side (quantize_pvt.c:428, :493)      [2][2] int32 (CBR) + mean_bits

calc_xmin / on_pe / reduce_side are observed through link-time wrappers in oracle/ref_harness.c (oracle/Makefile:
-Wl,--wrap); nothing of the reference is changed.  Output: tests/golden/stages_<name>.npz, SHA-256 (first 16 hex
digits) per frame and group.  tests/test_stage_fixtures.py compares the device (LH_DEBUG_DUMP build) with them bit
for bit: the tolerance on the psycho-acoustic energies is 0 ulp."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "deprecated-lame-mirror_amd"))
import helpers  # noqa: E402

NAMES = ("cbr128_js_44k", "vbr2_js_44k")
GROUPS = ("xr", "en", "thm", "xmin", "pe", "targ")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def split_stage(v):
    """refh_stage_get's flat vector -> dict of arrays (see oracle/ref_harness.c)"""
    o = 0
    out = {}
    for key, n in (("xmin", 4 * 39), ("en", 4 * 61), ("thm", 4 * 61), ("pe", 4), ("targ", 4), ("mean_bits", 1),
                   ("n_xmin", 1), ("n_on_pe", 1)):
        out[key] = v[o:o + n]
        o += n
    return out


def reference_stages(name):
    g, pcm = helpers.load_golden(name)
    ref = helpers.Reference()
    lib = ref.lib
    lib.refh_stage_get.argtypes = [C.c_void_p, C.c_int]
    kw = helpers.golden_encoder_kwargs(g)
    sr, br, mode, q = helpers.golden_settings(g)
    mode, q = (-1 if mode is None else mode), (-1 if q is None else q)
    vq = helpers.golden_vbr_q(g)
    if vq is None:
        h = C.c_void_p(lib.refh_open(sr, br, mode, q))
    else:
        h = C.c_void_p(lib.refh_open_vbr(sr, vq, mode, q, 0, 0))
    lib.refh_stage_watch(1)
    left, right = np.ascontiguousarray(pcm[0]), np.ascontiguousarray(pcm[1])
    n = len(left)
    out = C.create_string_buffer(16384)
    buf = (C.c_float * 1024)()
    xr = np.zeros((2, 2, 576), np.float32)
    rows, last = [], 0
    zeros = np.zeros(1152, np.int16)
    calls = (n + 1151) // 1152 + 3          # the last calls feed zeros: the frames the flush would bring
    for i in range(calls):
        a, b = left[1152 * i:1152 * i + 1152], right[1152 * i:1152 * i + 1152]
        if len(a) < 1152:
            a = np.concatenate([a, zeros[:1152 - len(a)]])
            b = np.concatenate([b, zeros[:1152 - len(b)]])
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        k = lib.refh_encode(h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), 1152, out, len(out))
        assert k >= 0
        m = lib.refh_stage_get(buf, 1024)
        assert m > 0
        fn = lib.refh_frame_number(h)
        if fn == last:
            continue
        assert fn == last + 1
        last = fn
        st = split_stage(np.frombuffer(buf, dtype=np.float32)[:m].copy())
        # (calc_xmin is not called for a granule without energy, quantize.c:2027-2039: that granule's entries keep what
        # the last frame that had energy there left -- on both sides, which start from zeros)
        lib.refh_get_xr(h, xr.ctypes.data_as(C.c_void_p))
        # refh_get_xr's order is [gr][ch] (oracle/ref_harness.c)
        rows.append({"xr": sha(xr), "en": sha(st["en"]), "thm": sha(st["thm"]), "xmin": sha(st["xmin"]),
                     "pe": sha(st["pe"]),
                     "targ": sha(np.concatenate([st["targ"], st["mean_bits"]]).astype(np.int32)) if vq is None else ""})
    lib.refh_close(h)
    return g, rows


def main():
    for name in NAMES:
        g, rows = reference_stages(name)
        path = os.path.join(HERE, "stages_%s.npz" % name)
        np.savez_compressed(path, name=name, nframes=len(rows), calls_of=1152,
                            **{k: np.array([r[k] for r in rows]) for k in GROUPS})
        print("%s: %d frames -> %s" % (name, len(rows), os.path.relpath(path)))


if __name__ == "__main__":
    main()
