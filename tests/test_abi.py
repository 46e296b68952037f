"""The C-ABI shared library loads on a machine without a GPU and exports every
symbol include/lamehip.h declares; the POD layouts agree with the bindings."""
import ctypes as C
import os
import re

import helpers
import lamehip
from lamehip import types as T


def declared_symbols():
    txt = open(os.path.join(helpers.ROOT, "include", "lamehip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lame_[a-zA-Z_0-9]+|lamehip_[a-zA-Z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = lamehip.load_library()
    syms = declared_symbols()
    assert len(syms) > 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_pod_sizes_match_bindings():
    lib = lamehip.load_library()
    assert lib.lamehip_abi_sizeof(0) == C.sizeof(T.LhConfig)
    assert lib.lamehip_abi_sizeof(1) == C.sizeof(T.LhTables)
    assert lib.lamehip_abi_sizeof(2) == C.sizeof(T.LhFrameOut)
    assert lib.lamehip_abi_sizeof(3) == C.sizeof(T.LhGranule)


def test_unsupported_settings_are_refused_loudly():
    lib = lamehip.load_library()
    for setup in (lambda h: lib.lame_set_num_channels(h, 3),          # neither mono nor stereo
                  lambda h: lib.lame_set_free_format(h, 1),           # free format frames
                  lambda h: lib.lame_set_out_samplerate(h, 20000)):   # not an MPEG rate
        h = C.c_void_p(lib.lame_init())
        lib.lame_set_bWriteVbrTag(h, 0)
        setup(h)
        assert lib.lame_init_params(h) == -1
        assert len(lamehip.last_error()) > 0
        lib.lame_close(h)


def test_contradictory_presets_are_refused_loudly():
    """A bitrate preset followed by a VBR mode or by a different bitrate preset: the reference then encodes with what the first
    preset's row left in its tuning options; not rebuilt here (found by tests/fuzz_frontend.py), so lame_init_params says no."""
    lib = lamehip.load_library()
    for setup in (lambda h: (lib.lame_set_preset(h, 1003), lib.lame_set_VBR(h, 2)),         # --preset insane --vbr-old
                  lambda h: (lib.lame_set_preset(h, 192), lib.lame_set_preset(h, 500)),      # --preset 192 --preset extreme (V0)
                  lambda h: (lib.lame_set_preset(h, 1003), lib.lame_set_preset(h, 128))):    # --preset insane --preset 128
        h = C.c_void_p(lib.lame_init())
        lib.lame_set_bWriteVbrTag(h, 0)
        setup(h)
        assert lib.lame_init_params(h) == -1
        assert b"preset" in lib.lamehip_last_error()
        lib.lame_close(h)


def test_encode_before_init_params_is_minus_3():
    lib = lamehip.load_library()
    h = C.c_void_p(lib.lame_init())
    buf = C.create_string_buffer(8192)
    z = (C.c_short * 1152)()
    assert lib.lame_encode_buffer(h, z, z, 1152, buf, 8192) == -3   # reference lame.h:687-692
    lib.lame_close(h)


def test_error_callback_receives_failures():
    """lame_set_errorf (reference lame.h:346): a failed call on the handle reports through the
    callback; NULL silences; the handle keeps reporting the first lame_init_params result."""
    import ctypes as C
    import lamehip
    lib = lamehip.load_library()
    seen = []
    CB = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
    cb = CB(lambda fmt, ap: seen.append(fmt))
    lib.lame_set_errorf.argtypes = [C.c_void_p, CB]
    h = C.c_void_p(lib.lame_init())
    assert lib.lame_set_errorf(h, cb) == 0
    lib.lame_set_out_samplerate(h, 20000)       # not an MPEG rate
    lib.lame_set_brate(h, 64)
    rc = lib.lame_init_params(h)
    assert rc < 0 and len(seen) == 1 and seen[0] == b"lamehip: %s\n"
    assert b"unsupported" in lib.lamehip_last_error()
    lib.lame_close(h)
    h = C.c_void_p(lib.lame_init())
    lib.lame_set_errorf.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.lame_set_errorf(h, None) == 0
    lib.lame_set_out_samplerate(h, 20000)
    assert lib.lame_init_params(h) < 0 and len(seen) == 1
    lib.lame_close(h)
