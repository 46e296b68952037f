#!/usr/bin/env python
"""Development aid: kernel time of one batch launch (no packing, no checks) -- for experiments whose
output is deliberately wrong.  usage: LAMEHIP_LIB=... python tools/time_kernel.py [streams] [seconds] [reps] [channels] [brate]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import lamehip  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    channels = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    brate = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    n = int(44100 * secs)
    enc = lamehip.Encoder(44100, brate, channels=channels)
    b = lamehip.Batch(enc, B, n)
    base = [helpers.synth_stream(100 + i, n) for i in range(8)]
    for s in range(B):
        b.set_pcm(s, base[s % 8][0], base[s % 8][1])
    ms = []
    for _ in range(reps + 1):
        b.encode(sync=True)
        ms.append(b.kernel_ms())
    print("kernel ms", " ".join("%.2f" % x for x in ms[1:]), "best %.2f" % min(ms[1:]), "-> %.0f x real-time (%d ch, %d kb/s)"
          % (B * secs / (min(ms[1:]) / 1e3), channels, brate))


main()
