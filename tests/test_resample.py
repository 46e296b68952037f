"""Input rate != output rate: the reference converts inside lame_encode_buffer (util.c:520-697), either because
the caller asked for an output rate or because the bitrate's lowpass makes it pick a lower one
(lame.c:273-345).  Resolved constants, the oracle's restatement of converter + call pattern, the product's
host converter, and -- on the GPU -- the handle API call by call against the compiled reference."""
import ctypes as C

import numpy as np
import pytest

import helpers
import lamehip
from lamehip.types import LhConfig, LhFrameOut, struct_diff

# (input rate, settings, explicit output rate or 0, expected output rate)
CASES = [
    (44100, dict(brate=96), 0, 32000),          # lowpass 15.1 kHz -> 32 kHz
    (48000, dict(brate=112), 0, 44100),         # lowpass 15.6 kHz -> 44.1 kHz
    (48000, dict(brate=128), 44100, 44100),
    (44100, dict(brate=192), 48000, 48000),     # up
    (22050, dict(brate=128), 44100, 44100),     # whole-number ratio: 32 taps
    (96000, dict(brate=160), 0, 48000),
    (37800, dict(brate=128), 0, 32000),         # CD-ROM XA rate: the MPEG rate below the input
    (44100, dict(vbr_q=7), 0, 32000),           # -V7 maps to 32 kHz and a fractional quality
    (48000, dict(abr=100), 0, 32000),
    (48000, dict(brate=64, channels=1), 32000, 32000),
]
IDS = ["%d-%s-%d" % (i, "_".join("%s%s" % kv for kv in kw.items()), o) for i, kw, o, _ in CASES]


def open_product(rate_in, kw, out, require_device):
    enc = lamehip.Encoder.__new__(lamehip.Encoder)
    lib = enc.lib = lamehip.load_library()
    enc.h = C.c_void_p(lib.lame_init())
    lib.lame_set_in_samplerate(enc.h, rate_in)
    lib.lame_set_num_channels(enc.h, kw.get("channels", 2))
    lib.lame_set_bWriteVbrTag(enc.h, 0)
    if out:
        lib.lame_set_out_samplerate(enc.h, out)
    if "brate" in kw:
        lib.lame_set_brate(enc.h, kw["brate"])
    if "vbr_q" in kw:
        lib.lame_set_VBR(enc.h, 4)
        lib.lame_set_VBR_q(enc.h, kw["vbr_q"])
    if "abr" in kw:
        lib.lame_set_VBR(enc.h, 3)
        lib.lame_set_VBR_mean_bitrate_kbps(enc.h, kw["abr"])
    enc.rc = lib.lame_init_params(enc.h)
    assert enc.rc == 0 or (enc.rc == lamehip.ERR_NODEVICE and not require_device), lamehip.last_error()
    return enc


def open_reference(reference, rate_in, kw, out):
    lib = reference.lib
    lib.refh_option.argtypes = [C.c_char_p, C.c_float]
    lib.refh_option(None, 0)
    if out:
        lib.refh_option(b"out_samplerate", float(out))
    lib.refh_set_channels(kw.get("channels", 2))
    try:
        if "abr" in kw:
            h = lib.refh_open_abr(rate_in, kw["abr"], -1, -1, out, 0)
        elif "vbr_q" in kw:
            h = lib.refh_open_vbr(rate_in, kw["vbr_q"], -1, -1, out, 0)
        else:
            h = lib.refh_open(rate_in, kw["brate"], -1, -1)
    finally:
        lib.refh_option(None, 0)
        lib.refh_set_channels(2)
    assert h, "reference refused the settings"
    return C.c_void_p(h)


def reference_calls(reference, h, pcm, pattern):
    """bytes the reference returns call by call (the last entry of `pattern' repeats), then for the flush"""
    lib = reference.lib
    buf = C.create_string_buffer(400000)
    n, pos, c, out = pcm.shape[1], 0, 0, []
    while pos < n:
        m = min(pattern[min(c, len(pattern) - 1)], n - pos)
        c += 1
        l = np.ascontiguousarray(pcm[0][pos:pos + m])
        r = np.ascontiguousarray(pcm[1][pos:pos + m])
        k = lib.refh_encode(h, l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), m, buf, len(buf))
        assert k >= 0
        out.append((pos, m, buf.raw[:k]))
        pos += m
    k = lib.refh_flush(h, buf, len(buf))
    assert k >= 0
    return out, buf.raw[:k]


PATTERNS = [[1152], [40000, 1, 333, 5000], [577]]


@pytest.mark.skipif(not helpers.have_reference(), reason="needs oracle/_ref (reference sources)")
@pytest.mark.parametrize("rate_in,kw,out,rate_out", CASES, ids=IDS)
def test_resampled_oracle_matches_reference(rate_in, kw, out, rate_out, oracle, reference):
    """resolved constants and the bytes of a whole stream: converter restatement + frame oracle + packer"""
    pcm = helpers.synth_stream(7000 + rate_in // 100, int(rate_in * 0.8), rate_in, 1.0 / 9)
    enc = open_product(rate_in, kw, out, require_device=False)
    cfg, tab = enc.config(), enc.tables()
    assert cfg.samplerate == rate_out
    for pattern in PATTERNS[:2 if "brate" in kw else 1]:
        h = open_reference(reference, rate_in, kw, out)
        rcfg = LhConfig()
        reference.lib.refh_get_config(h, C.byref(rcfg))
        assert not struct_diff(rcfg, cfg, skip=("bitrate_index",))
        calls, tail = reference_calls(reference, h, pcm, pattern)
        reference.lib.refh_close(h)
        want = b"".join(c[2] for c in calls) + tail
        cap = int(pcm.shape[1] * rate_out / rate_in) + 8192
        fl, fr = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        nf, pad = C.c_int(0), C.c_int(0)
        pat = (C.c_int * len(pattern))(*pattern)
        lib = oracle.lib
        lib.orc_resample_stream.restype = C.c_long
        n = lib.orc_resample_stream(C.byref(cfg), rate_in, pcm[0].ctypes.data_as(C.c_void_p),
                                    np.ascontiguousarray(pcm[1]).ctypes.data_as(C.c_void_p), C.c_long(pcm.shape[1]),
                                    pat, len(pattern), fl.ctypes.data_as(C.c_void_p), fr.ctypes.data_as(C.c_void_p),
                                    C.c_long(cap), C.byref(nf), C.byref(pad))
        assert 0 < n <= cap
        frames = (LhFrameOut * nf.value)()
        lib.orc_encode_stream_f(C.byref(cfg), C.byref(tab), fl.ctypes.data_as(C.c_void_p), fr.ctypes.data_as(C.c_void_p),
                                C.c_long(n), nf.value, frames, nf.value)
        assert helpers.pack_frames(enc.lib, cfg, tab, list(frames)) == want
    enc.close()


@pytest.mark.parametrize("rate_in,rate_out", [(44100, 32000), (48000, 44100), (22050, 44100), (44100, 48000),
                                              (96000, 48000), (37800, 44100), (8000, 32000)])
def test_host_converter_equals_oracle_block_by_block(rate_in, rate_out, oracle):
    """lh_rs_block (the product's converter, host C) against the oracle's restatement: same blocks, same floats"""
    lib = lamehip.load_library()
    rng = np.random.default_rng(rate_in + rate_out)
    x = (rng.standard_normal(30000) * 9000).astype(np.float32)
    rs = C.create_string_buffer(200000)
    lib.lh_rs_init(rs, rate_in, rate_out)
    assert lib.lh_rs_needed(rate_in, rate_out) == 1 and lib.lh_rs_needed(44100, 44110) == 0
    olib = oracle.lib
    ors = C.create_string_buffer(200000)
    olib.orc_rs_setup(ors, rate_in, rate_out)
    pos, blk_a, blk_b = 0, np.zeros(1152, np.float32), np.zeros(1152, np.float32)
    lens = [5000, 1, 40, 1152, 9000, 17]
    c = 0
    while pos < len(x):
        m = min(lens[c % len(lens)], len(x) - pos)
        c += 1
        at = 0
        while m > 0:
            ua, ub = C.c_int(0), C.c_int(0)
            src = x[pos + at:]
            ka = lib.lh_rs_block(rs, 0, blk_a.ctypes.data_as(C.c_void_p), 1152, src.ctypes.data_as(C.c_void_p), m,
                                 C.byref(ua))
            kb = olib.orc_rs_fill(ors, blk_b.ctypes.data_as(C.c_void_p), 1152, src.ctypes.data_as(C.c_void_p), m,
                                  C.byref(ub), 0)
            assert ka == kb and ua.value == ub.value and ua.value > 0
            assert blk_a[:ka].tobytes() == blk_b[:kb].tobytes()
            at += ua.value
            m -= ua.value
        pos += at


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in,kw,out,rate_out", CASES, ids=IDS)
def test_resampled_batch_matches_reference_frontend_pattern(rate_in, kw, out, rate_out, reference):
    """A batch converts each stream the way the reference does when its frontend feeds it 1152 input samples per
    call: host packer and device packer both give the reference's bytes, streams of different lengths."""
    lens = [int(rate_in * 1.2), 5000, 1, int(rate_in * 0.5) + 13]
    pcms = [helpers.synth_stream(7200 + i, n, rate_in, 1.0 / 9) for i, n in enumerate(lens)]
    enc = open_product(rate_in, kw, out, require_device=True)
    b = lamehip.Batch(enc, len(pcms), max(lens))
    b.set_device_packing()
    for s, x in enumerate(pcms):
        b.set_pcm(s, x[0], x[1])
    b.encode()
    for s, x in enumerate(pcms):
        h = open_reference(reference, rate_in, kw, out)
        calls, tail = reference_calls(reference, h, x, [1152])
        reference.lib.refh_close(h)
        want = b"".join(c[2] for c in calls) + tail
        assert b.pack(s) == want, "stream %d" % s
        assert b.get_bytes(s) == want, "stream %d (device packer)" % s
    b.close()
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in,kw,out,rate_out", CASES, ids=IDS)
def test_resampled_handle_matches_reference_call_by_call(rate_in, kw, out, rate_out, reference):
    pcm = helpers.synth_stream(7100 + rate_in // 100, int(rate_in * 1.1), rate_in, 1.0 / 9)
    for pattern in PATTERNS:
        h = open_reference(reference, rate_in, kw, out)
        calls, tail = reference_calls(reference, h, pcm, pattern)
        reference.lib.refh_close(h)
        enc = open_product(rate_in, kw, out, require_device=True)
        assert enc.config().samplerate == rate_out
        for pos, m, want in calls:
            got = enc.encode(np.ascontiguousarray(pcm[0][pos:pos + m]), np.ascontiguousarray(pcm[1][pos:pos + m]))
            assert got == want, "call with %d samples at %d (pattern %r)" % (m, pos, pattern)
        assert enc.flush() == tail
        enc.close()


@pytest.mark.gpu
def test_resampled_tag_frame_matches_reference(reference):
    """the LAME tag's source-rate bits and padding field follow the input rate / the converter's delay"""
    rate_in, br = 48000, 128
    pcm = helpers.synth_stream(7300, int(rate_in * 0.7), rate_in, 1.0 / 9)
    lib = reference.lib
    lib.refh_option.argtypes = [C.c_char_p, C.c_float]
    lib.refh_option(None, 0)
    lib.refh_option(b"out_samplerate", 44100.0)
    try:
        stream, tag = helpers.reference_tagged(pcm, rate_in, br)
    finally:
        lib.refh_option(None, 0)
    enc = lamehip.Encoder(rate_in, br, write_tag=True, out_samplerate=44100)
    got = b""
    for i in range(0, pcm.shape[1], 1152):
        got += enc.encode(np.ascontiguousarray(pcm[0][i:i + 1152]), np.ascontiguousarray(pcm[1][i:i + 1152]))
    got += enc.flush()
    assert got == stream
    assert enc.lametag_frame() == tag
    b = lamehip.Batch(enc, 1, pcm.shape[1])
    b.set_device_packing()
    b.set_pcm(0, pcm[0], pcm[1])
    b.encode()
    assert b.pack_tagged(0) == tag + stream[len(tag):]
    assert b.get_bytes_tagged(0) == tag + stream[len(tag):]
    b.close()
    enc.close()
