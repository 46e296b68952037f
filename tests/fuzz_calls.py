#!/usr/bin/env python
"""Randomised hunt over the handle API on the GPU box (test tool, not collected by pytest; needs oracle/_ref, which travels):
random input rate / rate control / channels, random chunk sizes -- the bytes every lame_encode_buffer call and the flush
return against the compiled reference's, call by call (with input-rate conversion when the rates differ).
Usage: python tests/fuzz_calls.py [cases] [seed]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import lamehip  # noqa: E402
import test_resample as tr  # noqa: E402
import test_gpu_parity as tg  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ref = helpers.Reference()
    bad = refused = done = 0
    for c in range(cases):
        rate_in = int(rng.choice([32000, 44100, 48000, 37800, 96000, 22050, 44100, 48000]))
        rc = int(rng.integers(0, 3))
        kw = [dict(brate=int(rng.choice([96, 112, 128, 160, 192, 256, 320]))), dict(vbr_q=int(rng.integers(0, 8))), dict(abr=int(rng.integers(100, 280)))][rc]
        if rng.integers(0, 4) == 0:
            kw["channels"] = 1
        out = int(rng.choice([0, 0, 32000, 44100, 48000]))
        pattern = [int(v) for v in rng.choice([1, 7, 576, 577, 1151, 1152, 1153, 2000, 4608, 9999, 30000], size=int(rng.integers(1, 6)))]
        x = tg._stress_signal(int(rng.integers(0, 1 << 30)), int(rate_in * float(rng.uniform(0.3, 1.2))), rate_in)
        try:
            enc = tr.open_product(rate_in, kw, out, require_device=True)
        except AssertionError:
            refused += 1
            continue
        try:
            h = tr.open_reference(ref, rate_in, kw, out)
        except AssertionError:
            enc.close()
            bad += 1
            print("MISMATCH case", c, rate_in, kw, out, "the reference refuses what the library accepts", flush=True)
            continue
        calls, tail = tr.reference_calls(ref, h, x, pattern)
        ref.lib.refh_close(h)
        what = None
        for pos, m, want in calls:
            got = enc.encode(x[0][pos:pos + m], x[1][pos:pos + m])
            if got != want:
                what = ("call at", pos, m, len(got), len(want))
                break
        if what is None:
            got = enc.flush()
            if got != tail:
                what = ("flush", len(got), len(tail))
        enc.close()
        done += 1
        if what is not None:
            bad += 1
            print("MISMATCH case", c, rate_in, kw, out, pattern, what, flush=True)
        if (c + 1) % 25 == 0:
            print("cases", c + 1, "compared", done, "refused", refused, "bad", bad, flush=True)
    print("TOTAL compared", done, "refused", refused, "BAD", bad)


if __name__ == "__main__":
    main()
