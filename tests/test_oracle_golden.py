"""CPU checks against the committed golden vectors (generated from the real
reference by tests/golden/make_golden.py): host init (config + tables), the CPU
oracle's per-frame payload, and the product's host bit packer."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import helpers
import lamehip
from lamehip.types import LhConfig, struct_diff


@pytest.mark.parametrize("name", helpers.golden_names() + helpers.golden_names(vbr=True) + helpers.golden_names(kind="abr")
                         + helpers.golden_names(kind="mono"))
def test_golden(name, oracle):
    g, pcm = helpers.load_golden(name)
    sr, br, mode, q = helpers.golden_settings(g)
    enc = lamehip.Encoder(require_device=False, **helpers.golden_encoder_kwargs(g))
    cfg, tab = enc.config(), enc.tables()
    # resolved constants == the reference's SessionConfig_t subset
    # (goldens made before LhConfig grew its trailing highpassfreq / ath_flags words hold the shorter image; both are
    # 0 for every golden's settings -- no high-pass, no ATH switches)
    ref_cfg = LhConfig.from_buffer_copy(g["config"].tobytes().ljust(C.sizeof(LhConfig), b"\0"))
    # (captured after the run: in VBR mode bitrate_index is the last frame's, run-time state)
    assert not struct_diff(ref_cfg, cfg, skip=("bitrate_index",) if cfg.vbr else ())
    # generated tables == the reference's
    for tn, th in zip(g["table_names"], g["table_sha256"]):
        v = getattr(tab, str(tn))
        h = hashlib.sha256(bytes(v) if not isinstance(v, (int, float)) else repr(v).encode()).hexdigest()
        assert h == str(th), "table %s differs from the reference" % tn
    # oracle payload, frame by frame
    frames = oracle.encode_frames(cfg, tab, pcm)
    assert len(frames) == int(g["nframes"])
    assert enc.lib.lh_total_frames_fs(C.c_long(pcm.shape[1]), 576 * cfg.mode_gr) == int(g["nframes"])
    mp3 = helpers.pack_frames(enc.lib, cfg, tab, frames)      # packer keeps table 14 internally
    helpers.normalize_tables(frames)
    got = [helpers.frame_sha(fr) for fr in frames]
    want = [str(x) for x in g["frame_sha256"]]
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not bad, "frames %s differ from the reference" % bad[:8]
    # byte stream through the product's packer
    assert hashlib.sha256(mp3).hexdigest() == str(g["mp3_sha256"])
    assert mp3 == g["mp3"].tobytes()
    enc.close()
