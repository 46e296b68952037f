"""What a frontend's progress display and file bookkeeping ask the handle for: frame counts, padding, samples still
buffered, and the bitrate / stereo mode / block type histograms (reference encoder.c:156-184, lame.c:2461-2610,
set_get.c:1989-2152) -- against the compiled reference, call by call."""
import ctypes as C

import numpy as np
import pytest

import helpers
import lamehip


def snapshot(lib, h):
    out = []
    for name, shape in (("lame_bitrate_kbps", 14), ("lame_bitrate_hist", 14), ("lame_stereo_mode_hist", 4),
                        ("lame_bitrate_stereo_mode_hist", 14 * 4), ("lame_block_type_hist", 6),
                        ("lame_bitrate_block_type_hist", 14 * 6)):
        a = (C.c_int * shape)()
        f = getattr(lib, name)
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_void_p]
        f(h, a)
        out.append(list(a))
    for name in ("lame_get_frameNum", "lame_get_totalframes", "lame_get_encoder_padding", "lame_get_mf_samples_to_encode",
                 "lame_get_encoder_delay", "lame_get_framesize"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
        out.append(f(h))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in,kw,out", [(44100, dict(vbr_q=4), 0), (48000, dict(brate=128), 0), (44100, dict(abr=140), 0),
                                            (44100, dict(brate=96), 0), (32000, dict(vbr_q=1, channels=1), 0)])
def test_progress_getters_match_reference_call_by_call(rate_in, kw, out, reference):
    import test_resample as T
    pcm = helpers.synth_stream(8100 + rate_in // 100, int(rate_in * 1.3), rate_in, 1.0 / 10)
    n = pcm.shape[1]
    rlib = reference.lib
    rlib.refh_gfp.restype = C.c_void_p
    rlib.lame_set_num_samples.argtypes = [C.c_void_p, C.c_ulong]
    rh = T.open_reference(reference, rate_in, kw, out)
    rg = C.c_void_p(rlib.refh_gfp(rh))
    enc = T.open_product(rate_in, kw, out, require_device=True)
    plib = enc.lib
    plib.lame_set_num_samples.argtypes = [C.c_void_p, C.c_ulong]
    assert snapshot(plib, enc.h)[7] == snapshot(rlib, rg)[7]          # no sample count announced yet
    rlib.lame_set_num_samples(rg, n)
    plib.lame_set_num_samples(enc.h, n)
    buf = C.create_string_buffer(100000)
    for pos in range(0, n, 4000):
        l = np.ascontiguousarray(pcm[0][pos:pos + 4000])
        r = np.ascontiguousarray(pcm[1][pos:pos + 4000])
        k = rlib.refh_encode(rh, l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), len(l), buf, len(buf))
        assert enc.encode(l, r) == buf.raw[:k]
        assert snapshot(plib, enc.h) == snapshot(rlib, rg), "after the call at %d" % pos
    k = rlib.refh_flush(rh, buf, len(buf))
    assert enc.flush() == buf.raw[:k]
    got, want = snapshot(plib, enc.h), snapshot(rlib, rg)
    assert got == want
    assert sum(got[1]) == got[6] and got[6] == got[7]                 # every frame counted; the announced count was right
    rlib.refh_close(rh)
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(brate=128), dict(vbr_q=3), dict(abr=120)], ids=["cbr128", "v3", "abr120"])
def test_nogap_file_boundary_matches_reference(kw, reference):
    """--nogap: lame_encode_flush_nogap + lame_init_bitstream between two files of one signal, tag frames included"""
    sr = 44100
    pcm = helpers.synth_stream(8300, int(sr * 1.6), sr, 1.0 / 10)
    cut = 30000
    rlib = reference.lib
    rlib.refh_gfp.restype = C.c_void_p
    rlib.refh_open_tag.restype = C.c_void_p
    if "abr" in kw:
        rh = C.c_void_p(rlib.refh_open_abr(sr, kw["abr"], -1, -1, 0, 1))
    elif "vbr_q" in kw:
        rh = C.c_void_p(rlib.refh_open_vbr(sr, kw["vbr_q"], -1, -1, 0, 1))
    else:
        rh = C.c_void_p(rlib.refh_open_tag(sr, kw["brate"], -1, -1))
    rg = C.c_void_p(rlib.refh_gfp(rh))
    enc = lamehip.Encoder(sr, kw.get("brate", 128), write_tag=True, vbr_q=kw.get("vbr_q"), abr=kw.get("abr"))
    plib = enc.lib
    buf, tag = C.create_string_buffer(200000), C.create_string_buffer(2880)
    plib.lame_encode_flush_nogap.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    plib.lame_init_bitstream.argtypes = [C.c_void_p]
    rlib.lame_init_bitstream.argtypes = [C.c_void_p]

    def feed(a, b):
        for pos in range(a, b, 1152):
            l = np.ascontiguousarray(pcm[0][pos:min(pos + 1152, b)])
            r = np.ascontiguousarray(pcm[1][pos:min(pos + 1152, b)])
            k = rlib.refh_encode(rh, l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), len(l), buf, len(buf))
            assert enc.encode(l, r) == buf.raw[:k], "call at %d" % pos

    feed(0, cut)
    k = rlib.refh_flush_nogap(rh, buf, len(buf))
    want = buf.raw[:k]
    k = plib.lame_encode_flush_nogap(enc.h, buf, len(buf))
    assert buf.raw[:k] == want
    k = rlib.refh_lametag(rh, tag, len(tag))
    assert enc.lametag_frame() == tag.raw[:k]                           # the first file's tag
    assert snapshot(plib, enc.h) == snapshot(rlib, rg)
    assert rlib.lame_init_bitstream(rg) == 0 and plib.lame_init_bitstream(enc.h) == 0
    assert snapshot(plib, enc.h) == snapshot(rlib, rg)
    feed(cut, pcm.shape[1])
    k = rlib.refh_flush(rh, buf, len(buf))
    assert enc.flush() == buf.raw[:k]
    k = rlib.refh_lametag(rh, tag, len(tag))
    assert enc.lametag_frame() == tag.raw[:k]
    assert snapshot(plib, enc.h) == snapshot(rlib, rg)
    rlib.refh_close(rh)
    enc.close()
