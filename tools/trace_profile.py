#!/usr/bin/env python
"""Where a CBR search iteration's cycles go (make -C deprecated-lame-mirror_amd/csrc trace: -DLH_TRACE, lh_dev_common.h).
On the GPU box:
    LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_trace.so python tools/trace_profile.py [streams] [seconds]
Every mark adds the cycles since the wave's previous mark to the segment that ends at it; a mark itself costs what segment 63
(two adjacent marks) shows, and that is taken off per visit."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import lamehip  # noqa: E402
import helpers  # noqa: E402

NAMES = {1: "count_bits: entry (loop control before it)", 2: "count_bits: step broadcast, products, threshold look-ups issued, band masks",
         3: "count_bits: quantise + selection", 4: "count_bits: top pairs (wave max x2), region split (LDS + scalar)",
         5: "count_bits: count1 quadruples, region maxima (wave max x3)", 6: "count_bits: class entry, grid look-ups, transposed sum",
         7: "count_bits: table choice, totals", 10: "calc_noise: entry (the gain loop's exit checks before it)",
         11: "calc_noise: band steps, xr + pow43 look-ups, squared errors to LDS", 12: "calc_noise: band sums (serial, lane = band)",
         13: "calc_noise: distortion, log, cache", 14: "calc_noise: over_count, SSD / max reduction",
         20: "balance_noise: entry", 21: "balance_noise: amp_scalefac_bands (wave max, trigger, ballot, scale)",
         22: "balance_noise: loop_break", 23: "balance_noise: scale_bitcount", 24: "balance_noise: rest (scalefac_scale / subblock gain, 2nd bitcount)",
         30: "loop: gain loop exit, recount rule (after the last count)", 31: "loop: noise commit + quant_compare",
         32: "loop: candidate kept as the best (image to LDS, gb = gw)", 33: "loop: age / exit tests, back to the top",
         34: "loop: gain loop left at its first count (exit test)", 35: "loop: ... after 1 rise of the gain",
         36: "loop: ... after 2 rises", 37: "loop: ... after 3", 38: "loop: ... after 4 or more",
         40: "stage: before bin_search", 41: "bin_search: its own control (between its counts) + exit", 42: "stage: first calc_noise kept, gw = gb",
         51: "stage: lq_load (granule to registers, grids to LDS)", 52: "stage: zero-band constants", 53: "stage: after the loop (table_select, scalefactors out)",
         63: "(a mark by itself)"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    n = int(44100 * secs)
    enc = lamehip.Encoder(44100, 128)
    lib = enc.lib
    buf = (C.c_ulonglong * (2 * 2 * 64))()
    b = lamehip.Batch(enc, B, n)
    base = [helpers.synth_stream(500 + i, n, 44100) for i in range(8)]
    for s in range(B):
        b.set_pcm(s, base[s % 8][0], base[s % 8][1])
    b.encode()              # warm-up
    assert lib.lh_trace_fetch(buf) == 0
    b.encode()
    assert lib.lh_trace_fetch(buf) == 0
    a = np.array(buf[:], dtype=np.float64).reshape(2, 2, 64)
    frames = b.frames(0)
    print("batch %d x %.1f s: kernels %.2f ms %s, %d frames/stream" % (B, secs, b.kernel_ms(), b.kernel_parts_ms()[1], frames))
    per = B * frames
    mark = a[:, 0, 63].sum() / max(a[:, 1, 63].sum(), 1)
    print("a mark costs %.1f cycles; figures below are per frame and wave (mean of the two waves), mark cost taken off" % mark)
    print("%-4s %-86s %9s %8s %9s" % ("seg", "what ends at this mark", "cycles", "visits", "cyc/visit"))
    tot = 0.0
    for k in sorted(NAMES):
        cyc = a[:, 0, k].sum() / 2 / per
        vis = a[:, 1, k].sum() / 2 / per
        if vis == 0:
            continue
        net = cyc - vis * mark
        if k != 63:
            tot += net
        print("%-4d %-86s %9.0f %8.2f %9.0f" % (k, NAMES[k], net, vis, net / vis))
    print("sum of the segments (the search stage, marks taken off): %.0f cycles per frame and wave" % tot)


if __name__ == "__main__":
    main()
