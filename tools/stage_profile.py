#!/usr/bin/env python
"""Per-stage cycle breakdown of lh_encode_kernel from the LH_PROF build
(make -C deprecated-lame-mirror_amd/csrc prof).  Usage on the GPU box:
    LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py [streams] [seconds] [vbr_q]
With vbr_q (vbr_mtrh) slots 3..10 mean: quant total, init+xmin+xrpow, geometry+scalefactor search,
scalefac_store+huffman_divide, geometry, constrain+bitcount, quantise+count, second-pass granules.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
import numpy as np  # noqa: E402
import lamehip  # noqa: E402

NAMES = ["frame total", "psy (2 granules)", "polyphase+mdct", "quant total (2 gr)", "init+xrpow+xmin",
         "outer_loop", "scalefac_store+huffman_divide", "  bin_search", "  balance_noise", "  calc_noise",
         "count_bits calls", "count_bits total", "  quantise part", "calc_noise calls", "  mask: partition sums",
         "  mask: + tonality, prefix sums", "  mask: + spreading", "  mask: total (all calls)", "  bhd: band tables built",
         "  bhd: + region 0 (16 lanes)", "  bhd: + region 1 (128 pairs)", "  bhd: + best (r0, r1) per sum", "  bhd: + first pass done",
         "  bhd: + second pass done (when it ran to the end)", "t: window staged (+ barrier skew)", "t: psy + ATH adjust done",
         "t: mdct, qtabs, M/S, PE FIR done", "t: granule loop done", "-", "  psy: attack detection", "  psy: + long FFT",
         "  psy: + power spectra, table staging", "  psy: + energy/loudness sums", "  psy: + long masking (+MS)",
         "  psy: + partition->sfb", "  psy: + short blocks", "  psy: + pre-echo",
         "  bal: amp_scalefac_bands", "  bal: + loop_break", "  bal: + scale_bitcount", "  bss: zero bands",
         "  bss: + scale/preflag", "  bss: + scfsi", "  bss: + scale_bitcount"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    n = int(44100 * secs)
    vq = int(sys.argv[3]) if len(sys.argv) > 3 else None
    enc = lamehip.Encoder(44100, 128, vbr_q=vq)
    b = lamehip.Batch(enc, B, n)
    rng = np.random.Generator(np.random.PCG64(900))
    t = np.arange(n) / 44100.0
    base = []
    for i in range(8):          # music-like: partials + noise bursts (same flavour as the tests' signal)
        x = sum(0.5 / (k + 1) * np.sin(2 * np.pi * 220.0 * 2 ** (k / 2.0) * t + rng.uniform(0, 6.28, (2, 1)))
                for k in range(8))
        x = x + 0.01 * rng.standard_normal((2, n))
        for s0 in range(7000, n - 2000, 14700):
            x[:, s0:s0 + 2000] += 0.6 * np.exp(-np.arange(2000) / 300.0) * rng.standard_normal((2, 2000))
        base.append((x / np.abs(x).max() * 0.8 * 32767).astype(np.int16))
    for s in range(B):
        b.set_pcm(s, base[s % 8][0], base[s % 8][1])
    if os.environ.get("LAMEHIP_PROFILE_PACK"):
        b.set_device_packing(True)       # the device bit packer's share: compare with a run without it
        # (the split pipeline's encode kernel leaves the masking slots free: lh_emit_frame's marks are there)
        NAMES[14:18] = ["  emit_frame: entry barrier", "  emit_frame: + side info, state, header", "  emit_frame: + frame bit string",
                        "  emit_frame: + bytes scattered"]
    b.encode()
    ms = b.kernel_ms()
    ssz = enc.lib.lamehip_abi_sizeof(4)
    NP = 44
    tot = np.zeros((2, NP))
    for s in range(0, B, max(1, B // 64)):
        buf = C.create_string_buffer(ssz)
        assert enc.lib.lamehip_batch_get_state(b.b, s, buf, ssz) == ssz
        prof = np.frombuffer(buf.raw[-2 * NP * 8:], dtype=np.uint64).reshape(2, NP)
        tot += prof
    frames = b.frames(0)
    if not tot.any():
        sys.exit("stage_profile: every counter is zero -- this library's kernel for these settings was not built with -DLH_PROF "
                 "(liblamehip_prof.so profiles the MPEG-1 CBR / ABR objects, fused and split; the VBR and MPEG-2 / 2.5 objects are "
                 "the product's)")
    print("batch %d x %.1f s: kernel %.2f ms, %d frames/stream" % (B, secs, ms, frames))
    for w in range(2):
        print("wave %d (cycles per frame, share of frame):" % w)
        for i, nm in enumerate(NAMES[:NP]):
            v = tot[w][i] / (frames * len(range(0, B, max(1, B // 64))))
            print("   %-34s %12.0f  %5.1f%%" % (nm, v, 100.0 * tot[w][i] / max(tot[w][0], 1)))


if __name__ == "__main__":
    main()
