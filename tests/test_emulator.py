"""Runs the HIP kernel SOURCE (csrc/lh_kernels.hip) on the CPU through the fiber
emulator in tests/hipemu and checks its payload bit for bit against the oracle.
This is a development aid for machines without a GPU; the GPU parity tests proper
are in test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers
import lamehip
from lamehip.types import LhFrameOut, struct_diff

EMU_DIR = os.path.join(helpers.ROOT, "tests", "hipemu")


class LhStreamDesc(C.Structure):
    _fields_ = [("pcm_l", C.c_longlong), ("pcm_r", C.c_longlong), ("pcm_base", C.c_longlong),
                ("nsamples", C.c_longlong), ("out_index", C.c_longlong), ("frame_begin", C.c_int),
                ("frame_end", C.c_int), ("bytes_base", C.c_longlong), ("bytes_cap", C.c_longlong),
                ("flush", C.c_int), ("mid_rel", C.c_int)]


@pytest.fixture(scope="module")
def emu():
    helpers.locked_make([], EMU_DIR)
    return C.CDLL(os.path.join(EMU_DIR, "libhipemu_lame.so"))


@pytest.mark.parametrize("name,nframes", [("cbr128_js_44k", 8), ("cbr320_js_48k_bursts", 8),
                                          ("cbr256_js_44k_q2", 5), ("cbr96_js_32k", 6),
                                          ("vbr2_js_44k", 8), ("vbr4_js_44k_white", 8), ("vbr0_js_48k_bursts", 8),
                                          ("vbr5_st_32k", 6), ("abr128_js_44k", 6), ("abr150_js_32k_white_q5", 6),
                                          ("mono_cbr160_48k_bursts_q5", 6), ("mono_vbr2_44k", 6), ("mono_abr100_44k", 5),
                                          ("vbrold2_js_44k", 8), ("vbrold4_js_44k_white", 6), ("vbrold0_js_48k_bursts", 8),
                                          ("vbrold5_st_32k_q5", 6), ("vbrold1_js_44k_q0", 2), ("vbrold3_js_44k_silence", 4),
                                          ("mono_vbrold4_44k", 6),
                                          # MPEG-2 / 2.5: one-granule frames, partitioned scalefactors
                                          ("cbr64_js_22k_lsf", 8), ("cbr32_js_16k_bursts_lsf", 10), ("cbr16_js_8k_lsf", 8),
                                          ("vbr4_js_22k_lsf", 8), ("abr56_js_22k_lsf", 6), ("mono_cbr48_22k_lsf", 6),
                                          ("vbrold2_js_24k_lsf", 6)])
def test_kernel_source_matches_oracle(name, nframes, emu, oracle):
    g, pcm = helpers.load_golden(name)
    sr, br, mode, q = helpers.golden_settings(g)
    enc = lamehip.Encoder(require_device=False, **helpers.golden_encoder_kwargs(g))
    cfg, tab = enc.config(), enc.tables()
    want = oracle.encode_frames(cfg, tab, pcm, max_frames=nframes)
    n = pcm.shape[1]
    pool = np.concatenate([pcm[0], pcm[1]]).astype(np.int16)
    desc = LhStreamDesc(0, n, 0, n, 0, 0, nframes)
    state = C.create_string_buffer(enc.lib.lamehip_abi_sizeof(4))
    enc.lib.lh_state_init(state, C.byref(cfg))
    got = (LhFrameOut * nframes)()
    emu.lh_emu_encode(C.byref(cfg), C.byref(tab), pool.ctypes.data_as(C.c_void_p), C.byref(desc), state, got, 1)
    for f in range(nframes):
        d = struct_diff(want[f], got[f])
        assert not d, (f, d[:4])
    enc.close()


class LhMidPools(C.Structure):
    _fields_ = [("frames", C.c_void_p)]


# size of the analysis kernels' per-frame record (csrc/lh_device.h: LhMidFrame = LhMidSmall, LhMidLong, LhMidShort, LhMidXr)
MID_FRAME = 2 * 288 + 2 * 4 * 3 * 64 * 4 + 2 * 3 * 4 * 3 * 64 * 4 + 2 * 2 * 576 * 4


def split_encode(emu, cfg, tab, pool, descs, states, got, nstreams, max_frames, total_frames, poison=0x5a):
    """The split pipeline under the emulator: attack / scan / analysis kernels, the sub-band kernel, then the encode
    kernel that starts from their output.  The pool is filled with a poison pattern first: whatever the encode
    kernel reads must have been written by the analysis kernels of THIS launch."""
    lsf = cfg.mode_gr == 1
    buf = np.full(total_frames * MID_FRAME + 64, poison, dtype=np.uint8)
    pools = LhMidPools(buf.ctypes.data)
    sfx = "_lsf" if lsf else ""
    pcm = pool.ctypes.data_as(C.c_void_p)
    getattr(emu, "lh_emu_analysis" + sfx)(C.byref(cfg), C.byref(tab), pcm, None, descs, states, C.byref(pools), nstreams, max_frames)
    getattr(emu, "lh_emu_subband" + sfx)(C.byref(cfg), C.byref(tab), pcm, None, descs, states, C.byref(pools), nstreams, max_frames)
    getattr(emu, "lh_emu_encode_q" + sfx)(C.byref(cfg), C.byref(tab), pcm, None, descs, states, got, None, nstreams, C.byref(pools))


SPLIT_CASES = [("cbr128_js_44k", 8), ("cbr320_js_48k_bursts", 8), ("cbr96_js_32k", 6), ("vbr2_js_44k", 8),
               ("vbr0_js_48k_bursts", 8), ("vbr5_st_32k", 6), ("abr150_js_32k_white_q5", 6),
               ("mono_cbr160_48k_bursts_q5", 6), ("mono_vbr2_44k", 6), ("vbrold0_js_48k_bursts", 8),
               ("vbrold5_st_32k_q5", 6), ("cbr32_js_16k_bursts_lsf", 10), ("vbr4_js_22k_lsf", 8),
               ("mono_cbr48_22k_lsf", 6), ("vbrold2_js_24k_lsf", 6)]


@pytest.mark.parametrize("name,nframes", SPLIT_CASES)
def test_split_pipeline_source_matches_oracle(name, nframes, emu, oracle):
    """Analysis kernels + sub-band kernel + the -DLH_SPLIT encode kernel == oracle, frame for frame; in two launches
    (the second picks the stream up from LhStreamState alone), the second one ending on an odd frame count."""
    g, pcm = helpers.load_golden(name)
    enc = lamehip.Encoder(require_device=False, **helpers.golden_encoder_kwargs(g))
    cfg, tab = enc.config(), enc.tables()
    want = oracle.encode_frames(cfg, tab, pcm, max_frames=nframes)
    n = pcm.shape[1]
    pool = np.concatenate([pcm[0], pcm[1]]).astype(np.int16)
    state = C.create_string_buffer(enc.lib.lamehip_abi_sizeof(4))
    enc.lib.lh_state_init(state, C.byref(cfg))
    got = (LhFrameOut * nframes)()
    cut = 3
    for a, b in ((0, cut), (cut, nframes)):
        desc = LhStreamDesc(0, n, 0, n, a, a, b)
        split_encode(emu, cfg, tab, pool, C.byref(desc), state, got, 1, b - a, nframes)
    for f in range(nframes):
        d = struct_diff(want[f], got[f])
        assert not d, (f, d[:4])
    enc.close()


def test_split_and_fused_kernels_take_turns_on_one_stream(emu, oracle):
    """Either kernel leaves LhStreamState as the other expects it: fused, split, fused over consecutive frame ranges."""
    g, pcm = helpers.load_golden("cbr320_js_48k_bursts")
    enc = lamehip.Encoder(require_device=False, **helpers.golden_encoder_kwargs(g))
    cfg, tab = enc.config(), enc.tables()
    nframes = 9
    want = oracle.encode_frames(cfg, tab, pcm, max_frames=nframes)
    n = pcm.shape[1]
    pool = np.concatenate([pcm[0], pcm[1]]).astype(np.int16)
    state = C.create_string_buffer(enc.lib.lamehip_abi_sizeof(4))
    enc.lib.lh_state_init(state, C.byref(cfg))
    got = (LhFrameOut * nframes)()
    for k, (a, b) in enumerate(((0, 3), (3, 6), (6, 9))):
        desc = LhStreamDesc(0, n, 0, n, a, a, b)
        if k == 1:
            split_encode(emu, cfg, tab, pool, C.byref(desc), state, got, 1, b - a, nframes)
        else:
            emu.lh_emu_encode(C.byref(cfg), C.byref(tab), pool.ctypes.data_as(C.c_void_p), C.byref(desc), state, got, 1)
    for f in range(nframes):
        d = struct_diff(want[f], got[f])
        assert not d, (f, d[:4])
    enc.close()


@pytest.mark.parametrize("name,nframes", [("testcase_wav_cbr128", 6), ("cbr320_js_48k_bursts", 6)])
def test_kernel_source_frame_per_launch_with_poisoned_lds(name, nframes, emu, oracle):
    """One launch per frame, as lame_encode_buffer drives the device, with the LDS image
    overwritten before every launch: nothing may be carried from launch to launch except
    LhStreamState."""
    g, pcm = helpers.load_golden(name)
    sr, br, mode, q = helpers.golden_settings(g)
    enc = lamehip.Encoder(sr, br, mode, q, require_device=False)
    cfg, tab = enc.config(), enc.tables()
    want = oracle.encode_frames(cfg, tab, pcm, max_frames=nframes)
    n = pcm.shape[1]
    pool = np.concatenate([pcm[0], pcm[1]]).astype(np.int16)
    state = C.create_string_buffer(enc.lib.lamehip_abi_sizeof(4))
    enc.lib.lh_state_init(state, C.byref(cfg))
    got = (LhFrameOut * nframes)()
    C.c_int.in_dll(emu, "lh_emu_poison_lds").value = 1
    try:
        for f in range(nframes):
            desc = LhStreamDesc(0, n, 0, n, f, f, f + 1)
            emu.lh_emu_encode(C.byref(cfg), C.byref(tab), pool.ctypes.data_as(C.c_void_p), C.byref(desc), state,
                              got, 1)
    finally:
        C.c_int.in_dll(emu, "lh_emu_poison_lds").value = 0
    for f in range(nframes):
        d = struct_diff(want[f], got[f])
        assert not d, (f, d[:4])
    enc.close()


@pytest.mark.parametrize("name", ["cbr128_js_44k_silence", "cbr320_js_48k_bursts", "vbr4_js_44k_white", "abr150_js_32k_white_q5",
                                  "mono_vbr2_44k", "testcase_wav_cbr128", "vbrold0_js_48k_bursts"])
def test_device_bit_packer_source_matches_host_packer(name, emu, oracle):
    """lh_dev_emit.h under the emulator: the bytes the kernel assembles (headers, side information, Huffman
    data around the headers, stuffing, final padding) equal the host packer's, i.e. the reference's."""
    g, pcm = helpers.load_golden(name)
    enc = lamehip.Encoder(require_device=False, **helpers.golden_encoder_kwargs(g))
    cfg, tab = enc.config(), enc.tables()
    nframes = int(g["nframes"])
    n = pcm.shape[1]
    pool = np.concatenate([pcm[0], pcm[1]]).astype(np.int16)
    cap = nframes * 1500
    desc = LhStreamDesc(0, n, 0, n, 0, 0, nframes, 0, cap, 1, 0)
    state = C.create_string_buffer(enc.lib.lamehip_abi_sizeof(4))
    enc.lib.lh_state_init(state, C.byref(cfg))
    got = (LhFrameOut * nframes)()
    out = C.create_string_buffer(cap)
    emu.lh_emu_encode_bytes(C.byref(cfg), C.byref(tab), pool.ctypes.data_as(C.c_void_p), C.byref(desc), state, got, out, 1)
    want = g["mp3"].tobytes()
    assert out.raw[:len(want)] == want
    assert out.raw[len(want):len(want) + 64] == bytes(64)
    enc.close()


def test_device_bit_packer_flags_a_slice_that_is_too_small(emu):
    """A stream whose slice of the byte pool cannot hold its frames: nothing is written past the slice and
    the state's status says so (bit 8), which makes lamehip_batch_get_bytes refuse the stream."""
    g, pcm = helpers.load_golden("cbr128_js_44k_silence")
    enc = lamehip.Encoder(require_device=False, **helpers.golden_encoder_kwargs(g))
    cfg, tab = enc.config(), enc.tables()
    nframes = int(g["nframes"])
    n = pcm.shape[1]
    pool = np.concatenate([pcm[0], pcm[1]]).astype(np.int16)
    want = g["mp3"].tobytes()
    cap = len(want) // 2
    desc = LhStreamDesc(0, n, 0, n, 0, 0, nframes, 0, cap, 1, 0)
    ssz = enc.lib.lamehip_abi_sizeof(4)
    state = C.create_string_buffer(ssz)
    enc.lib.lh_state_init(state, C.byref(cfg))
    got = (LhFrameOut * nframes)()
    out = C.create_string_buffer(cap + 4096)
    emu.lh_emu_encode_bytes(C.byref(cfg), C.byref(tab), pool.ctypes.data_as(C.c_void_p), C.byref(desc), state, got, out, 1)
    assert out.raw[cap:] == bytes(4096)
    # LhStreamState (lh_device.h): ... pefirbuf[19], slot_lag, ResvSize, ResvMax, main_data_begin, OldValue[2],
    # CurrentStep[2], masking_lower, substep_shaping, frame_number, primed, status, pad[3], em_*
    off_pefir = 4 * (4 * 4 * 64 + 2 + 4 + 36 + 4 + 2 + 2 + 2 * 576)
    words = np.frombuffer(state.raw[off_pefir + 19 * 4:off_pefir + 19 * 4 + 13 * 4], dtype=np.int32)
    frame_number, primed, status = int(words[10]), int(words[11]), int(words[12])
    assert frame_number == nframes and primed == 1      # (the offsets are right)
    assert status & 8
    enc.close()


@pytest.mark.parametrize("name,nframes,forced", [("vbrold2_js_44k", 3, 1), ("vbrold0_js_48k_bursts", 5, 2), ("mono_vbrold4_44k", 3, 1)])
def test_old_vbr_loop_second_pass_source_matches_oracle(name, nframes, forced, emu, oracle, monkeypatch):
    """The old VBR loop's second pass over a frame (bitpressure_strategy: more noise allowed, smaller budgets, every
    granule searched again from the scalefactors the last pass left).  The budgets of the first pass add up to what
    the largest frame holds, so real input never gets there; LH_TEST_FORCE_PRESSURE makes the first `forced'
    evaluations of every frame fail in the oracle and in the emulator build of the kernel source alike."""
    monkeypatch.setenv("LH_TEST_FORCE_PRESSURE", str(forced))
    g, pcm = helpers.load_golden(name)
    enc = lamehip.Encoder(require_device=False, **helpers.golden_encoder_kwargs(g))
    cfg, tab = enc.config(), enc.tables()
    want = oracle.encode_frames(cfg, tab, pcm, max_frames=nframes)
    plain = None
    monkeypatch.delenv("LH_TEST_FORCE_PRESSURE")
    plain = oracle.encode_frames(cfg, tab, pcm, max_frames=nframes)
    assert any(struct_diff(want[f], plain[f]) for f in range(nframes)), "the forced pass changes the result"
    monkeypatch.setenv("LH_TEST_FORCE_PRESSURE", str(forced))
    n = pcm.shape[1]
    pool = np.concatenate([pcm[0], pcm[1]]).astype(np.int16)
    desc = LhStreamDesc(0, n, 0, n, 0, 0, nframes)
    state = C.create_string_buffer(enc.lib.lamehip_abi_sizeof(4))
    enc.lib.lh_state_init(state, C.byref(cfg))
    got = (LhFrameOut * nframes)()
    emu.lh_emu_encode(C.byref(cfg), C.byref(tab), pool.ctypes.data_as(C.c_void_p), C.byref(desc), state, got, 1)
    for f in range(nframes):
        d = struct_diff(want[f], got[f])
        assert not d, (f, d[:4])
    enc.close()


def test_new_vbr_second_pass_keeps_the_first_pass_side_info(emu, oracle):
    """A regression case of tests/fuzz_switches.py under the emulator (-V0 -B 96, first frame): the second pass over a frame
    ends a granule without big values, whose region counts stay what the first pass's finishing steps left."""
    import test_gpu_parity as tg
    from test_switches import open_with
    sr, nframes = 44100, 2
    pcm = tg._stress_signal(778490755, int(sr * 1.2), sr)
    enc = open_with(dict(vbr_q=0), {"strict_ISO": 2, "VBR_max_bitrate_kbps": 96}, require_device=False)
    cfg, tab = enc.config(), enc.tables()
    want = oracle.encode_frames(cfg, tab, pcm, max_frames=nframes)
    n = pcm.shape[1]
    pool = np.concatenate([pcm[0], pcm[1]]).astype(np.int16)
    desc = LhStreamDesc(0, n, 0, n, 0, 0, nframes)
    state = C.create_string_buffer(enc.lib.lamehip_abi_sizeof(4))
    enc.lib.lh_state_init(state, C.byref(cfg))
    got = (LhFrameOut * nframes)()
    emu.lh_emu_encode(C.byref(cfg), C.byref(tab), pool.ctypes.data_as(C.c_void_p), C.byref(desc), state, got, 1)
    for f in range(nframes):
        d = struct_diff(want[f], got[f])
        assert not d, (f, d[:4])
    enc.close()
