"""Live pinning of the CPU restatement against the compiled reference (only in
trees where oracle/_ref was built): fresh seeds, several settings, every frame's
payload and the final bytes."""
import numpy as np
import pytest

import helpers
import lamehip
from lamehip.types import struct_diff

CASES = [(44100, 128, None, None, 5, 4.0), (48000, 320, 1, None, 6, 2.0), (44100, 192, 0, None, 7, 2.0),
         (32000, 96, None, None, 8, 2.0), (44100, 224, None, 1, 9, 1.5), (44100, 128, None, 4, 10, 2.0),
         (48000, 128, None, 7, 11, 1.5),
         # MPEG-2 / 2.5 (one granule per frame)
         (22050, 64, None, None, 12, 2.0), (24000, 96, 0, None, 13, 1.5), (16000, 32, None, 2, 14, 2.0),
         (12000, 32, None, None, 15, 2.0), (8000, 16, None, None, 16, 2.5), (11025, 40, None, 5, 17, 2.0)]


@pytest.mark.parametrize("sr,br,mode,q,seed,secs", CASES)
def test_oracle_matches_reference(sr, br, mode, q, seed, secs, oracle, reference):
    pcm = helpers.synth_stream(seed, int(sr * secs), sr, 1.0 / 7)
    mp3, nf, rframes, rcfg, rtab = reference.encode(pcm, sr, br, -1 if mode is None else mode,
                                                    -1 if q is None else q, max_frames=2048)
    enc = lamehip.Encoder(sr, br, mode, q, require_device=False)
    cfg, tab = enc.config(), enc.tables()
    assert not struct_diff(rcfg, cfg)
    assert not struct_diff(rtab, tab, skip=("fft_window", "fft_window_s", "fht_tw", "ma_max_i1", "ma_max_i2",
                                            "psy_l_to_s", "hgrid", "qthr", "vqthr", "vq3", "line_pad0", "line_pad1", "line_pad2", "mask_mid", "bvpack"))
    frames = oracle.encode_frames(cfg, tab, pcm)
    assert len(frames) == nf
    mine = helpers.pack_frames(enc.lib, cfg, tab, frames)
    helpers.normalize_tables(frames)
    for f in range(nf):
        d = struct_diff(rframes[f], frames[f], skip=("frame_bits",))
        assert not d, (f, d[:4])
    assert mine == mp3
    enc.close()


VBR_CASES = [(44100, 2, None, None, 21, 2.0, False), (44100, 4, None, None, 22, 1.5, True),
             (48000, 0, None, None, 23, 1.5, False), (32000, 6, 0, None, 24, 1.5, False),
             (44100, 5, None, 7, 25, 1.5, False), (44100, 9, None, 5, 26, 1.5, False), (48000, 8, None, None, 27, 1.0, True),
             (22050, 4, None, None, 28, 2.0, False), (16000, 6, None, None, 29, 1.5, True), (24000, 2, 0, 5, 30, 1.5, False)]


@pytest.mark.parametrize("sr,vq,mode,q,seed,secs,white", VBR_CASES)
def test_vbr_oracle_matches_reference(sr, vq, mode, q, seed, secs, white, oracle, reference):
    """vbr_mtrh (-V n): config, tables, every frame's payload incl. the chosen bitrate, final bytes."""
    pcm = helpers.synth_stream(seed, int(sr * secs), sr, 1.0 / 7, white)
    out = sr if vq >= 7 else 0          # -V7.. would resample unless the output rate is pinned
    mp3, nf, rframes, rcfg, rtab = reference.encode(pcm, sr, 0, -1 if mode is None else mode,
                                                    -1 if q is None else q, max_frames=2048, vbr_q=vq,
                                                    out_samplerate=out)
    enc = lamehip.Encoder(sr, mode=mode, quality=q, require_device=False, vbr_q=vq, out_samplerate=out)
    cfg, tab = enc.config(), enc.tables()
    assert not struct_diff(rcfg, cfg, skip=("bitrate_index",))      # run-time state in VBR mode
    assert not struct_diff(rtab, tab, skip=("fft_window", "fft_window_s", "fht_tw", "ma_max_i1", "ma_max_i2",
                                            "psy_l_to_s", "hgrid", "qthr", "vqthr", "vq3", "line_pad0", "line_pad1", "line_pad2", "mask_mid", "bvpack"))
    frames = oracle.encode_frames(cfg, tab, pcm)
    assert len(frames) == nf
    mine = helpers.pack_frames(enc.lib, cfg, tab, frames)
    helpers.normalize_tables(frames)
    for f in range(nf):
        d = struct_diff(rframes[f], frames[f], skip=("frame_bits",))
        assert not d, (f, d[:4])
    assert mine == mp3
    assert len({fr.bitrate_index for fr in frames}) > 1 or white
    enc.close()


ABR_CASES = [(44100, 128, None, None, 51, 1.5, False), (48000, 200, 0, None, 52, 1.2, False),
             (32000, 96, None, 5, 53, 1.2, True), (44100, 313, None, 0, 54, 1.0, False), (44100, 150, 1, 7, 55, 1.2, False),
             (22050, 56, None, None, 56, 1.5, False), (16000, 40, None, None, 57, 1.5, True)]


@pytest.mark.parametrize("sr,kb,mode,q,seed,secs,white", ABR_CASES)
def test_abr_oracle_matches_reference(sr, kb, mode, q, seed, secs, white, oracle, reference):
    """ABR (--abr n): config, tables, every frame's payload incl. the chosen bitrate, final bytes."""
    pcm = helpers.synth_stream(seed, int(sr * secs), sr, 1.0 / 7, white)
    mp3, nf, rframes, rcfg, rtab = reference.encode(pcm, sr, 0, -1 if mode is None else mode,
                                                    -1 if q is None else q, max_frames=2048, abr=kb)
    enc = lamehip.Encoder(sr, mode=mode, quality=q, require_device=False, abr=kb)
    cfg, tab = enc.config(), enc.tables()
    assert not struct_diff(rcfg, cfg, skip=("bitrate_index",))      # run-time state outside CBR
    assert not struct_diff(rtab, tab, skip=("fft_window", "fft_window_s", "fht_tw", "ma_max_i1", "ma_max_i2",
                                            "psy_l_to_s", "hgrid", "qthr", "vqthr", "vq3", "line_pad0", "line_pad1", "line_pad2", "mask_mid", "bvpack"))
    frames = oracle.encode_frames(cfg, tab, pcm)
    assert len(frames) == nf
    mine = helpers.pack_frames(enc.lib, cfg, tab, frames)
    helpers.normalize_tables(frames)
    for f in range(nf):
        d = struct_diff(rframes[f], frames[f], skip=("frame_bits",))
        assert not d, (f, d[:4])
    assert mine == mp3
    enc.close()


@pytest.mark.parametrize("sr,kw,q,nch,mode", [(44100, dict(brate=128), None, 1, None), (48000, dict(brate=64), 5, 1, None),
                                               (44100, dict(vbr_q=3), None, 1, None), (32000, dict(abr=72), None, 1, None),
                                               (44100, dict(brate=112), None, 2, 3), (48000, dict(vbr_q=1), None, 2, 3),
                                               (44100, dict(abr=150), 7, 2, 3), (22050, dict(brate=48), None, 1, None), (16000, dict(vbr_q=5), None, 1, None)])
def test_mono_oracle_matches_reference(sr, kw, q, nch, mode, oracle, reference):
    """One channel out: from one input channel (nch = 1) or from two mixed down (mode 3 = MONO)."""
    pcm = helpers.synth_stream(61 + sr // 1000, int(sr * 1.3), sr, 1.0 / 7)
    if nch == 1:
        pcm = np.stack([pcm[0], pcm[0]])
    rkw = dict(kw)
    br = rkw.pop("brate", 0)
    mp3, nf, rframes, rcfg, rtab = reference.encode(pcm, sr, br, -1 if mode is None else mode, -1 if q is None else q,
                                                    max_frames=2048, channels=nch, **rkw)
    enc = lamehip.Encoder(sr, mode=mode, quality=q, require_device=False, channels=nch, **kw)
    cfg, tab = enc.config(), enc.tables()
    assert cfg.channels == 1 and (cfg.pcm_mix != 0) == (nch == 2)
    assert not struct_diff(rcfg, cfg, skip=("bitrate_index",))
    assert not struct_diff(rtab, tab, skip=("fft_window", "fft_window_s", "fht_tw", "ma_max_i1", "ma_max_i2",
                                            "psy_l_to_s", "hgrid", "qthr", "vqthr", "vq3", "line_pad0", "line_pad1", "line_pad2", "mask_mid", "bvpack"))
    frames = oracle.encode_frames(cfg, tab, pcm)
    assert len(frames) == nf
    mine = helpers.pack_frames(enc.lib, cfg, tab, frames)
    helpers.normalize_tables(frames)
    for f in range(nf):
        d = struct_diff(rframes[f], frames[f], skip=("frame_bits",))
        assert not d, (f, d[:4])
    assert mine == mp3
    enc.close()


@pytest.mark.parametrize("sr,kw", [(44100, dict(brate=160)), (48000, dict(vbr_q=3)), (32000, dict(abr=128))])
def test_dual_channel_oracle_matches_reference(sr, kw, oracle, reference):
    """MPEG mode 2 (dual channel): no M/S, block types not coupled -- attacks in one channel only."""
    n = int(sr * 1.0)
    x = helpers.synth_stream(99, n, sr, 1.0 / 11)
    pcm = np.stack([x[0], (8000 * np.sin(2 * np.pi * 440 * np.arange(n) / sr)).astype(np.int16)])
    rkw = dict(kw)
    br = rkw.pop("brate", 0)
    mp3, nf, rframes, rcfg, rtab = reference.encode(pcm, sr, br, 2, -1, max_frames=2048, **rkw)
    enc = lamehip.Encoder(sr, mode=2, require_device=False, **kw)
    cfg, tab = enc.config(), enc.tables()
    assert not struct_diff(rcfg, cfg, skip=("bitrate_index",))
    frames = oracle.encode_frames(cfg, tab, pcm)
    assert sum(fr.gr[g][0].block_type != fr.gr[g][1].block_type for fr in frames for g in range(2)) > 5
    assert helpers.pack_frames(enc.lib, cfg, tab, frames) == mp3
    enc.close()


def test_odd_lengths_and_flush_framing(oracle, reference):
    for n in (1, 500, 1151, 1152, 1153, 1152 * 3, 1152 * 3 + 17, 5000):
        pcm = helpers.synth_stream(n, n)
        mp3, nf, _, _, _ = reference.encode(pcm, 44100, 128)
        enc = lamehip.Encoder(44100, 128, require_device=False)
        frames = oracle.encode_frames(enc.config(), enc.tables(), pcm)
        assert len(frames) == nf
        assert helpers.pack_frames(enc.lib, enc.config(), enc.tables(), frames) == mp3
        enc.close()
