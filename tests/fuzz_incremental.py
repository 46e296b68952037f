#!/usr/bin/env python
"""Randomised hunt over the incremental batch API on the GPU box (test tool, not collected by pytest; needs oracle/_ref, which
travels): random rate control, several streams fed in random pieces; what every stream drains after a round against what the
compiled reference's lame_encode_buffer returns for that stream's call of the round, and the finish against its flush.
Usage: python tests/fuzz_incremental.py [cases] [seed]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import lamehip  # noqa: E402
import test_gpu_parity as tg  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ref = helpers.Reference()
    lib = ref.lib
    lib.refh_open_tag.restype = C.c_void_p
    buf = C.create_string_buffer(800000)
    bad = 0
    for c in range(cases):
        sr = int(rng.choice([32000, 44100, 48000]))
        rc = int(rng.integers(0, 4))
        kw = [dict(brate=int(rng.choice([96, 128, 192, 320]))), dict(vbr_q=int(rng.integers(0, 7))), dict(abr=int(rng.integers(100, 280))),
              dict(vbr_q=int(rng.integers(0, 7)), vbr_mode=2)][rc]
        B = int(rng.integers(1, 7))
        lens = [int(sr * float(rng.uniform(0.05, 1.0))) for _ in range(B)]
        pcms = [tg._stress_signal(int(rng.integers(0, 1 << 30)), lens[s], sr) for s in range(B)]
        hs = []
        for s in range(B):
            lib.refh_set_vbr_mode(kw.get("vbr_mode", 4))
            if "abr" in kw:
                h = lib.refh_open_abr(sr, kw["abr"], -1, -1, sr, 0)
            elif "vbr_q" in kw:
                h = lib.refh_open_vbr(sr, kw["vbr_q"], -1, -1, sr, 0)
            else:
                h = lib.refh_open(sr, kw["brate"], -1, -1)
            lib.refh_set_vbr_mode(4)
            hs.append(C.c_void_p(h))
        try:
            enc = lamehip.Encoder(sr, out_samplerate=sr if "brate" not in kw else 0, **kw)
        except RuntimeError:
            for h in hs:
                lib.refh_close(h)
            continue
        if enc.config().samplerate != sr or not all(hs):
            enc.close()
            for h in hs:
                if h:
                    lib.refh_close(h)
            continue
        b = lamehip.Batch(enc, B, max(lens) + 16)
        pos = [0] * B
        what = None
        rounds = 0
        while what is None and any(pos[s] < lens[s] for s in range(B)):
            want = []
            for s in range(B):
                n = int(rng.choice([0, 1, 13, 575, 576, 1152, 1153, 2304, 4000, 9000, 20000]))
                n = min(n, lens[s] - pos[s])
                l = np.ascontiguousarray(pcms[s][0][pos[s]:pos[s] + n])
                r = np.ascontiguousarray(pcms[s][1][pos[s]:pos[s] + n])
                k = lib.refh_encode(hs[s], l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), n, buf, len(buf))
                assert k >= 0
                want.append(buf.raw[:k])
                if n:
                    b.append(s, l, r)
                pos[s] += n
            b.encode_available()
            for s in range(B):
                if b.drain(s) != want[s]:
                    what = ("round", rounds, "stream", s)
                    break
            rounds += 1
        if what is None:
            b.finish()
            for s in range(B):
                k = lib.refh_flush(hs[s], buf, len(buf))
                if b.drain(s) != buf.raw[:k]:
                    what = ("flush of stream", s)
                    break
        for h in hs:
            lib.refh_close(h)
        b.close()
        enc.close()
        if what is not None:
            bad += 1
            print("MISMATCH case", c, sr, kw, B, lens, what, flush=True)
    print("TOTAL cases", cases, "BAD", bad)


if __name__ == "__main__":
    main()
