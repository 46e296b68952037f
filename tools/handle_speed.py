#!/usr/bin/env python
"""One stream through the lame.h-shaped handle API (one launch per call).  On the GPU box: python tools/handle_speed.py [seconds]
Calls of 1152 samples, as the reference's frontend makes them, and larger ones (a call encodes every frame that became complete
in ONE launch): what a launch costs beyond its frames.  Round 6: 76 x real time at 1152 samples per call (round 5: 67 x), 80 ... 83 x from 4 frames per
call on -- holding frames back inside the library to launch less often would gain 7 %, and was not built."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import lamehip  # noqa: E402


def run(pcm, frames):
    enc = lamehip.Encoder(44100, 128)
    n = pcm.shape[1]
    out = b""
    t0 = time.perf_counter()
    for i in range(0, n, 1152 * frames):
        out += enc.encode(pcm[0][i:i + 1152 * frames], pcm[1][i:i + 1152 * frames])
    out += enc.flush()
    dt = time.perf_counter() - t0
    enc.close()
    return out, dt


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    pcm = helpers.synth_stream(77, int(44100 * secs), 44100)
    run(pcm[:, :44100], 1)          # warm-up (module load, first launch)
    ref = None
    for k in (1, 4, 16, 32, 64, 128):
        out, dt = run(pcm, k)
        ref = out if ref is None else ref
        print("frames per call %4d: %.3f s for %.0f s of audio = %.1f x real time, bytes %s" % (k, dt, secs, secs / dt, "identical" if out == ref else "DIFFER"))


if __name__ == "__main__":
    main()
