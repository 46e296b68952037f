"""lh_replaygain.c (the radio gain of the LAME tag, measured on the host beside the encode) against the reference's
gain_analysis.c, called block by block the way lame_encode_buffer calls it: same blocks in, same tenths of a dB out,
for every supported rate, mono and stereo, blocks from one sample to a frame, titles in a row (--nogap)."""
import ctypes as C

import numpy as np
import pytest

import helpers
import lamehip


class LhReplayGain(C.Structure):
    _fields_ = [("rate_index", C.c_int), ("window", C.c_long), ("filled", C.c_long), ("lsum", C.c_double), ("rsum", C.c_double),
                ("hist", (C.c_float * 10) * 6), ("bins", C.c_uint32 * 12000), ("work", (C.c_float * (10 + 2404)) * 3)]


@pytest.mark.skipif(not helpers.have_reference(), reason="needs oracle/_ref (reference sources)")
@pytest.mark.parametrize("rate,channels,seed", [(44100, 2, 1), (48000, 2, 2), (32000, 1, 3), (44100, 1, 4), (24000, 2, 5)])
def test_radio_gain_matches_reference_block_by_block(rate, channels, seed, reference):
    lib = lamehip.load_library()
    ref = reference.lib
    rng = np.random.default_rng(seed)
    mine = LhReplayGain()
    assert lib.lh_rg_start(C.byref(mine), rate) == 0
    theirs = C.create_string_buffer(400000)      # replaygain_t (about 290 KB)
    ref.InitGainAnalysis.argtypes = [C.c_void_p, C.c_long]
    ref.AnalyzeSamples.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    ref.GetTitleGain.restype = C.c_float
    ref.GetTitleGain.argtypes = [C.c_void_p]
    lib.lh_rg_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    assert ref.InitGainAnalysis(theirs, rate) == 1
    for title in range(3):
        n_total = int(rate * (0.3 + 1.2 * title))
        x = helpers.synth_stream(900 + seed + title, n_total, rate).astype(np.float32) * np.float32(0.25 * (title + 1))
        at = 0
        while at < n_total:
            n = int(rng.choice([1, 3, 9, 10, 11, 576, 1152, rng.integers(1, 1153)]))
            n = min(n, n_total - at)
            l = np.ascontiguousarray(x[0, at:at + n])
            r = np.ascontiguousarray(x[1, at:at + n])
            assert lib.lh_rg_block(C.byref(mine), l.ctypes.data, r.ctypes.data, n, channels) == 0
            assert ref.AnalyzeSamples(theirs, l.ctypes.data, r.ctypes.data, n, channels) == 1
            at += n
        want = ref.GetTitleGain(theirs)
        got = lib.lh_rg_finish(C.byref(mine))
        assert got == (0 if want == -24601 else int(np.floor(np.float32(want) * 10.0 + 0.5))), (title, want, got)
    assert got != 0
