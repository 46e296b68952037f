#!/usr/bin/env python
"""Generates the committed golden vectors from the REAL reference (oracle/_ref,
built from /root/reference by oracle/Makefile).  Runs only in the build
container.  Each fixture holds the recipe of its input, the reference's MP3 bytes
and a SHA-256 per frame of the reference's side-info payload (LhFrameOut image
captured after each lame_encode_buffer call).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import shutil
import sys
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

CASES = [
    # name, samplerate, brate, mode, quality, seed, seconds, burst_interval, white
    ("cbr128_js_44k", 44100, 128, -1, -1, 101, 1.5, None, False),
    ("cbr128_js_44k_white", 44100, 128, -1, -1, 102, 1.0, None, True),
    ("cbr320_js_48k_bursts", 48000, 320, 1, -1, 103, 1.0, 1.0 / 40, False),
    ("cbr192_st_44k", 44100, 192, 0, -1, 104, 1.0, None, False),
    ("cbr96_js_32k", 32000, 96, -1, -1, 105, 1.0, None, False),
    ("cbr128_js_44k_q0", 44100, 128, -1, 0, 106, 0.8, None, False),
    ("cbr256_js_44k_q2", 44100, 256, -1, 2, 107, 0.8, None, False),
    ("cbr160_js_44k_q5", 44100, 160, -1, 5, 108, 0.8, None, False),
    ("cbr128_js_44k_silence", 44100, 128, -1, -1, -1, 0.5, None, False),
    # MPEG-2 / 2.5 (LSF: one granule per frame, partitioned scalefactors)
    ("cbr64_js_22k_lsf", 22050, 64, -1, -1, 601, 1.2, None, False),
    ("cbr56_js_24k_lsf", 24000, 56, -1, -1, 602, 1.0, None, False),
    ("cbr32_js_16k_bursts_lsf", 16000, 32, -1, -1, 603, 1.2, 1.0 / 12, False),
    ("cbr32_js_12k_lsf", 12000, 32, -1, -1, 604, 1.2, None, False),             # MPEG-2.5
    ("cbr16_js_8k_lsf", 8000, 16, -1, -1, 605, 1.5, None, False),               # MPEG-2.5, 17 / 9 coded bands
    ("cbr160_st_22k_q2_lsf", 22050, 160, 0, 2, 606, 0.8, None, True),
]

VBR_CASES = [
    # name, samplerate, vbr_q (vbr_mtrh -V n), mode, quality, seed, seconds, burst_interval, white
    ("vbr2_js_44k", 44100, 2, -1, -1, 201, 1.5, None, False),          # BASELINE.json config 3
    ("vbr4_js_44k_white", 44100, 4, -1, -1, 202, 1.0, None, True),     # out-of-bits strategies
    ("vbr0_js_48k_bursts", 48000, 0, -1, -1, 203, 1.0, 1.0 / 40, False),   # sfb21 bands, short blocks
    ("vbr5_st_32k", 32000, 5, 0, -1, 204, 1.0, None, False),           # rescaled fractional quality
    ("vbr6_js_44k_q7", 44100, 6, -1, 7, 205, 0.8, None, False),        # guessed scalefactors
    ("vbr3_js_44k_q5", 44100, 3, -1, 5, 206, 0.8, None, False),        # no best-huffman pass
    ("vbr2_js_44k_silence", 44100, 2, -1, -1, -1, 0.5, None, False),
    ("vbr4_js_22k_lsf", 22050, 4, -1, -1, 607, 1.2, None, False),     # LSF scalefactor ranges with preflag
    ("vbr6_js_16k_white_lsf", 16000, 6, -1, -1, 608, 1.0, None, True),
]


OLD_CASES = [
    # the old VBR loop, lame_set_VBR(vbr_rh) / the frontend's --vbr-old: name, samplerate, -V n, mode, quality, seed,
    # seconds, burst_interval, white
    ("vbrold2_js_44k", 44100, 2, -1, -1, 501, 1.2, None, False),
    ("vbrold4_js_44k_white", 44100, 4, -1, -1, 502, 0.8, None, True),
    ("vbrold0_js_48k_bursts", 48000, 0, -1, -1, 503, 0.8, 1.0 / 40, False),    # sfb21 bands searched, short blocks
    ("vbrold5_st_32k_q5", 32000, 5, 0, 5, 504, 0.8, None, False),
    ("vbrold1_js_44k_q0", 44100, 1, -1, 0, 505, 0.6, None, True),              # one band per pass, full search
    ("vbrold3_js_44k_silence", 44100, 3, -1, -1, -1, 0.5, None, False),
    ("mono_vbrold4_44k", 44100, 4, 3, -1, 506, 0.8, None, False),
    ("vbrold2_js_24k_lsf", 24000, 2, -1, -1, 611, 1.0, None, False),
]


ABR_CASES = [
    # name, samplerate, mean kb/s (--abr n), mode, quality, seed, seconds, burst_interval, white
    ("abr128_js_44k", 44100, 128, -1, -1, 301, 1.2, None, False),
    ("abr200_st_48k_bursts", 48000, 200, 0, -1, 302, 1.0, 1.0 / 40, False),
    ("abr150_js_32k_white_q5", 32000, 150, -1, 5, 303, 0.8, None, True),
    ("abr320_js_44k_q0", 44100, 320, -1, 0, 304, 0.8, None, False),
    ("abr112_js_44k_silence", 44100, 112, -1, -1, -1, 0.5, None, False),
    ("abr56_js_22k_lsf", 22050, 56, -1, -1, 609, 1.0, None, False),
]


MONO_CASES = [
    # name, samplerate, kwargs of Reference.encode (brate / vbr_q / abr), quality, seed, seconds, burst, white
    ("mono_cbr96_44k", 44100, dict(brate=96), -1, 401, 1.0, None, False),
    ("mono_cbr160_48k_bursts_q5", 48000, dict(brate=160), 5, 402, 1.0, 1.0 / 30, False),
    ("mono_vbr2_44k", 44100, dict(vbr_q=2), -1, 403, 1.0, None, False),
    ("mono_vbr5_32k_white", 32000, dict(vbr_q=5), -1, 404, 0.8, None, True),
    ("mono_abr100_44k", 44100, dict(abr=100), -1, 405, 1.0, None, False),
    ("mono_cbr48_22k_lsf", 22050, dict(brate=48), -1, 610, 1.0, None, False),
]


def frame_hash(fr):
    return hashlib.sha256(bytes(fr)).hexdigest()


def table_hashes(tab):
    out = {}
    for name, _ in tab._fields_:
        if name in ("fft_window", "fft_window_s", "fht_tw", "ma_max_i1", "ma_max_i2", "psy_l_to_s", "hgrid", "qthr", "vqthr", "vq3",
                    "line_pad0", "line_pad1", "line_pad2", "mask_mid"):
            continue            # file-local in the reference (not visible through the harness), or derived for the device
        v = getattr(tab, name)
        out[name] = hashlib.sha256(bytes(v) if not isinstance(v, (int, float)) else repr(v).encode()).hexdigest()
    return out


def main():
    ref = helpers.Reference()
    # the reference's own test input (data file of its `make test`)
    src = "/root/reference/testcase.wav"
    shutil.copyfile(src, os.path.join(HERE, "testcase.wav"))
    w = wave.open(src)
    n = w.getnframes()
    pcm = np.frombuffer(w.readframes(n), dtype=np.int16).reshape(-1, 2).T
    cases = [("testcase_wav_cbr128", 44100, 128, -1, -1, None, pcm, -1), ("testcase_wav_vbr2", 44100, 0, -1, -1, None, pcm, 2)]
    for name, sr, br, mode, q, seed, secs, burst, white in CASES:
        nn = int(sr * secs)
        x = np.zeros((2, nn), np.int16) if seed < 0 else helpers.synth_stream(seed, nn, sr, burst, white)
        cases.append((name, sr, br, mode, q, (seed, secs, burst, white), x, -1))
    for name, sr, vq, mode, q, seed, secs, burst, white in VBR_CASES:
        nn = int(sr * secs)
        x = np.zeros((2, nn), np.int16) if seed < 0 else helpers.synth_stream(seed, nn, sr, burst, white)
        cases.append((name, sr, 0, mode, q, (seed, secs, burst, white), x, vq))
    cases = [c + (-1,) for c in cases]
    for name, sr, kb, mode, q, seed, secs, burst, white in ABR_CASES:
        nn = int(sr * secs)
        x = np.zeros((2, nn), np.int16) if seed < 0 else helpers.synth_stream(seed, nn, sr, burst, white)
        cases.append((name, sr, 0, mode, q, (seed, secs, burst, white), x, -1, kb))
    cases = [c + (2,) for c in cases]
    for name, sr, kw, q, seed, secs, burst, white in MONO_CASES:
        x = helpers.synth_stream(seed, int(sr * secs), sr, burst, white)
        x = np.stack([x[0], x[0]])      # mono: the one channel the encoder reads
        cases.append((name, sr, kw.get("brate", 0), -1, q, (seed, secs, burst, white), x, kw.get("vbr_q", -1),
                      kw.get("abr", -1), 1))
    cases = [c + (4,) for c in cases]
    for name, sr, vq, mode, q, seed, secs, burst, white in OLD_CASES:
        nn = int(sr * secs)
        x = np.zeros((2, nn), np.int16) if seed < 0 else helpers.synth_stream(seed, nn, sr, burst, white)
        nch = 1 if name.startswith("mono_") else 2
        if nch == 1:
            x = np.stack([x[0], x[0]])
        cases.append((name, sr, 0, -1 if nch == 1 else mode, q, (seed, secs, burst, white), x, vq, -1, nch, 2))
    only = sys.argv[1] if len(sys.argv) > 1 else ""     # make_golden.py [substring]: only the fixtures whose name holds it
    for name, sr, br, mode, q, recipe, x, vq, abr, nch, vmode in cases:
        if only not in name:
            continue
        mp3, nf, frames, cfg, tab = ref.encode(x, sr, br, mode, q, max_frames=4096, vbr_q=None if vq < 0 else vq,
                                               abr=None if abr < 0 else abr, channels=nch, vbr_mode=vmode)
        hashes = [frame_hash(frames[f]) for f in range(nf)]
        th = table_hashes(tab)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            samplerate=sr, brate=br, mode=mode, quality=q, vbr_q=vq, abr=abr, channels=nch, vbr_mode=vmode,
            recipe=np.array([-2 if recipe is None else recipe[0],
                             0 if recipe is None else recipe[1],
                             0 if (recipe is None or recipe[2] is None) else recipe[2],
                             0 if recipe is None else int(recipe[3])], dtype=np.float64),
            nsamples=x.shape[1], nframes=nf,
            mp3=np.frombuffer(mp3, dtype=np.uint8),
            mp3_sha256=hashlib.sha256(mp3).hexdigest(),
            frame_sha256=np.array(hashes),
            config=np.frombuffer(bytes(cfg), dtype=np.uint8),
            table_names=np.array(list(th.keys())), table_sha256=np.array(list(th.values())),
            first_frames=np.frombuffer(b"".join(bytes(frames[f]) for f in range(min(nf, 3))), dtype=np.uint8))
        print(name, "frames", nf, "bytes", len(mp3))


if __name__ == "__main__":
    main()
