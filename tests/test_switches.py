"""Frontend-level switches of lame.h that reach the hot path (-m f, --nores, -p, -c/-o/-e, --strictly-enforce-ISO,
--lowpass, --scale*, --noshort / --shortblocks): resolved constants, oracle bytes and -- on the GPU -- the
device payload against the compiled reference / the oracle."""
import ctypes as C

import numpy as np
import pytest

import helpers
import lamehip
from lamehip.types import struct_diff

CASES = [
    (dict(brate=128), {"force_ms": 1}), (dict(vbr_q=3), {"force_ms": 1}),
    (dict(brate=128), {"disable_reservoir": 1}), (dict(vbr_q=2), {"disable_reservoir": 1}),
    (dict(abr=150), {"disable_reservoir": 1}),
    (dict(brate=160), {"error_protection": 1}), (dict(vbr_q=4), {"error_protection": 1}),
    (dict(brate=128), {"copyright": 1, "original": 0, "emphasis": 1, "extension": 1}),
    (dict(brate=192), {"strict_ISO": 0}), (dict(vbr_q=0), {"strict_ISO": 1}), (dict(brate=320), {"strict_ISO": 1}),
    (dict(brate=128), {"lowpassfreq": 16500, "lowpasswidth": 1500}), (dict(vbr_q=2), {"lowpassfreq": 16000}),
    (dict(brate=256), {"lowpassfreq": -1}),
    (dict(brate=128), {"scale": 0.7}), (dict(vbr_q=2), {"scale": 1.4}),
    (dict(abr=128), {"scale_left": 0.5, "scale_right": 1.2}),
    (dict(brate=128), {"no_short_blocks": 1}), (dict(vbr_q=2), {"force_short_blocks": 1}),
    (dict(brate=160), {"allow_diff_short": 1}),
    (dict(vbr_q=2), {"VBR_min_bitrate_kbps": 96}), (dict(vbr_q=4), {"VBR_max_bitrate_kbps": 160}),
    (dict(vbr_q=5), {"VBR_min_bitrate_kbps": 128, "VBR_hard_min": 1}),
    (dict(abr=128), {"VBR_min_bitrate_kbps": 64, "VBR_max_bitrate_kbps": 192}),
    (dict(abr=250), {"VBR_max_bitrate_kbps": 224}),
    (dict(vbr_q=2), {"VBR_quality": 2.5}), (dict(vbr_q=0), {"VBR_quality": 5.31}),
    # (brate 0 = left alone, as a frontend that only says --preset does)
    (dict(brate=0), {"preset": 1001}), (dict(brate=0), {"preset": 1003}), (dict(brate=0), {"preset": 1006}),
    (dict(brate=0), {"preset": 150}), (dict(vbr_q=4), {"preset": 450}), (dict(brate=0), {"preset": 1002}),
    # the frontend's tuning switches (--nsmsfix, --athtype, --athcurve, --athlower, --athaa-type, --athaa-sensitivity,
    # --noath, --athshort, --interch, --temporal-masking, --highpass / --highpass-width, --ns-bass / -alto / -treble /
    # -sfb21 / --nssafejoint, -Y, --comp)
    (dict(brate=128), {"msfix": 2.0}), (dict(vbr_q=2), {"msfix": 0.5}), (dict(abr=160), {"msfix": 0.0}),
    (dict(brate=128), {"ATHtype": 2}), (dict(brate=192), {"ATHtype": 0}), (dict(vbr_q=3), {"ATHtype": 1}),
    (dict(brate=128), {"ATHcurve": 7.5}), (dict(vbr_q=2), {"ATHcurve": 1.0}),
    (dict(brate=160), {"ATHlower": 6.0}), (dict(vbr_q=4), {"ATHlower": -3.5}),
    (dict(brate=128), {"athaa_type": 0}), (dict(vbr_q=2), {"athaa_type": 1}), (dict(brate=128), {"athaa_sensitivity": 4.0}),
    (dict(vbr_q=2), {"athaa_sensitivity": -2.5}),
    (dict(brate=128), {"noATH": 1}), (dict(vbr_q=3), {"noATH": 1}), (dict(brate=128), {"ATHshort": 1}),
    (dict(brate=128), {"interChRatio": 0.3}), (dict(brate=128), {"useTemporal": 0}), (dict(vbr_q=2), {"useTemporal": 1}),
    (dict(brate=128), {"highpassfreq": 400}), (dict(vbr_q=2), {"highpassfreq": 1500, "highpasswidth": 700}),
    (dict(brate=128), {"highpassfreq": 10}), (dict(brate=192), {"highpassfreq": -1, "lowpassfreq": -1}),
    (dict(brate=128), {"exp_nspsytune": 1 | (8 << 2) | (60 << 8) | (4 << 14)}), (dict(vbr_q=2), {"exp_nspsytune": 1 | 2 | (12 << 20)}),
    (dict(vbr_q=0), {"exp_nspsytune": 1 | (56 << 20)}), (dict(abr=128), {"exp_nspsytune": 1 | 2}),
    (dict(vbr_q=0), {"experimentalY": 1}), (dict(brate=128), {"experimentalY": 1}),
    (dict(brate=0), {"compression_ratio": 8.0}), (dict(brate=0), {"compression_ratio": 5.0}), (dict(brate=128), {"compression_ratio": 14.0}),
    (dict(brate=0), {}),
    (dict(brate=128), {"VBR_quality": 4.7}), (dict(abr=140), {"VBR_quality": 6.3}),   # the fraction reaches the CBR / ABR tables too
    (dict(brate=176), {}), (dict(brate=144), {}),      # lowpass from the bitrate as asked, frame size from the rounded one
    # the old VBR loop (lame_set_VBR(vbr_rh), the frontend's --vbr-old); a low -B makes it raise the allowed noise
    # and search again (bitpressure_strategy)
    (dict(vbr_q=2, vbr_mode=2), {}), (dict(vbr_q=4, vbr_mode=2), {"VBR_max_bitrate_kbps": 96}),
    (dict(vbr_q=0, vbr_mode=2), {"VBR_max_bitrate_kbps": 64}), (dict(vbr_q=5, vbr_mode=2), {"VBR_min_bitrate_kbps": 128, "VBR_hard_min": 1}),
    (dict(vbr_q=3, vbr_mode=2), {"force_ms": 1}), (dict(vbr_q=6, vbr_mode=2), {"disable_reservoir": 1}),
    (dict(vbr_q=1, vbr_mode=2), {"ATHtype": 2}), (dict(vbr_q=0, vbr_mode=2), {"experimentalY": 1}),
    (dict(vbr_q=2, vbr_mode=2), {"VBR_quality": 2.5}), (dict(vbr_q=4, vbr_mode=2), {"force_short_blocks": 1}),
]
FLOAT_OPTS = ("scale", "scale_left", "scale_right", "VBR_quality", "ATHcurve", "ATHlower", "athaa_sensitivity", "interChRatio",
              "compression_ratio")
IDS = ["%s-%s" % ("_".join("%s%s" % kv for kv in kw.items()).replace("_vbr_mode2", "old"), "_".join(o)) for kw, o in CASES]


def open_with(kw, opts, require_device):
    """lame_init .. lame_init_params through the C ABI with the switches applied."""
    enc = lamehip.Encoder.__new__(lamehip.Encoder)
    lib = enc.lib = lamehip.load_library()
    enc.h = C.c_void_p(lib.lame_init())
    lib.lame_set_in_samplerate(enc.h, 44100)
    lib.lame_set_num_channels(enc.h, 2)
    lib.lame_set_bWriteVbrTag(enc.h, 0)
    if "brate" in kw:
        lib.lame_set_brate(enc.h, kw["brate"])
    if "vbr_q" in kw:
        lib.lame_set_VBR(enc.h, kw.get("vbr_mode", 4))
        lib.lame_set_VBR_q(enc.h, kw["vbr_q"])
    if "abr" in kw:
        lib.lame_set_VBR(enc.h, 3)
        lib.lame_set_VBR_mean_bitrate_kbps(enc.h, kw["abr"])
    for k, v in opts.items():
        f = getattr(lib, "lame_set_" + k)
        if k == "msfix":                # void lame_set_msfix(lame_t, double)
            f.argtypes, f.restype = [C.c_void_p, C.c_double], None
            f(enc.h, float(v))
            continue
        f.argtypes = [C.c_void_p, C.c_float if k in FLOAT_OPTS else C.c_int]
        rc = f(enc.h, float(v) if k in FLOAT_OPTS else int(v))
        assert rc == 0 or k == "preset"
    enc.rc = lib.lame_init_params(enc.h)
    assert enc.rc == 0 or (enc.rc == lamehip.ERR_NODEVICE and not require_device), lamehip.last_error()
    return enc


@pytest.mark.skipif(not helpers.have_reference(), reason="needs oracle/_ref (reference sources)")
@pytest.mark.parametrize("kw,opts", CASES, ids=IDS)
def test_switch_oracle_matches_reference(kw, opts, oracle, reference):
    sr = 44100
    pcm = helpers.synth_stream(500 + len(opts), int(sr * 0.9), sr, 1.0 / 9)
    lib = reference.lib
    lib.refh_option.argtypes = [C.c_char_p, C.c_float]
    lib.refh_option(None, 0)
    for k, v in opts.items():
        lib.refh_option(k.encode(), float(v))
    rkw = dict(kw)
    br = rkw.pop("brate", 0)
    try:
        mp3, nf, rframes, rcfg, rtab = reference.encode(pcm, sr, br, -1, -1, max_frames=2048, **rkw)
    finally:
        lib.refh_option(None, 0)
    enc = open_with(kw, opts, require_device=False)
    cfg, tab = enc.config(), enc.tables()
    assert not struct_diff(rcfg, cfg, skip=("bitrate_index",))
    assert not struct_diff(rtab, tab, skip=("fft_window", "fft_window_s", "fht_tw", "ma_max_i1", "ma_max_i2",
                                            "psy_l_to_s", "hgrid", "qthr", "vqthr", "vq3", "line_pad0", "line_pad1", "line_pad2", "mask_mid", "bvpack"))
    frames = oracle.encode_frames(cfg, tab, pcm)
    assert len(frames) == nf
    assert helpers.pack_frames(enc.lib, cfg, tab, frames) == mp3
    enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw,opts", CASES, ids=IDS)
def test_switch_device_matches_oracle(kw, opts, oracle):
    sr = 44100
    pcm = helpers.synth_stream(600 + len(opts), int(sr * 0.9), sr, 1.0 / 9)
    enc = open_with(kw, opts, require_device=True)
    cfg, tab = enc.config(), enc.tables()
    b = lamehip.Batch(enc, 1, pcm.shape[1])
    b.set_pcm(0, pcm[0], pcm[1])
    b.encode()
    want = oracle.encode_frames(cfg, tab, pcm)
    got = b.get_frames(0)
    bad = [f for f in range(len(want)) if struct_diff(want[f], got[f])]
    assert len(got) == len(want) and not bad, bad[:5]
    assert b.pack(0) == helpers.pack_frames(enc.lib, cfg, tab, want)
    b.close()
    enc.close()


REGRESSIONS = [
    # found by tests/fuzz_switches.py: the new-VBR loop's second pass (a low -B) ends a granule without big values / with an
    # empty region, whose region counts / table the bit count then leaves as the FIRST pass's finishing steps set them
    (dict(vbr_q=0), {"strict_ISO": 2, "VBR_max_bitrate_kbps": 96}, 778490755),
    (dict(vbr_q=3), {"ATHtype": 0, "lowpassfreq": -1, "VBR_max_bitrate_kbps": 96, "highpassfreq": 2500}, 84886511),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kw,opts,signal", REGRESSIONS, ids=["v0_B96", "v3_B96_highpass"])
def test_fuzz_regressions_device_matches_oracle(kw, opts, signal, oracle):
    import test_gpu_parity as tg
    sr = 44100
    pcm = tg._stress_signal(signal, int(sr * 1.2), sr)
    enc = open_with(kw, opts, require_device=True)
    cfg, tab = enc.config(), enc.tables()
    b = lamehip.Batch(enc, 1, pcm.shape[1])
    b.set_pcm(0, pcm[0], pcm[1])
    b.encode()
    want = oracle.encode_frames(cfg, tab, pcm)
    got = b.get_frames(0)
    bad = [(f, struct_diff(want[f], got[f])[:3]) for f in range(len(want)) if struct_diff(want[f], got[f])]
    assert len(got) == len(want) and not bad, bad[:3]
    assert b.pack(0) == helpers.pack_frames(enc.lib, cfg, tab, want)
    b.close()
    enc.close()


def test_lowest_bitrate_above_highest_is_refused():
    """-b 128 -B 96: the reference takes it and lets the reservoir run negative (ABR); here lame_init_params says no."""
    with pytest.raises(AssertionError):
        open_with(dict(abr=158), {"VBR_max_bitrate_kbps": 96, "VBR_min_bitrate_kbps": 128}, require_device=False)
