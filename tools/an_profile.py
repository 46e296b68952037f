#!/usr/bin/env python
"""Phase profile of the analysis kernels (make -C deprecated-lame-mirror_amd/csrc aprof).  On the GPU box:
    LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_aprof.so python tools/an_profile.py [streams] [seconds]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import lamehip  # noqa: E402
import helpers  # noqa: E402

NAMES = {0: "an: start-up (descriptor, addresses)", 1: "an: samples staged", 2: "an: long FHT", 3: "an: power spectra",
         4: "an: spreading matrix staged", 5: "an: energy / loudness sums", 6: "an: long masking", 7: "an: short blocks",
         8: "sb: priming (window + polyphase)", 9: "sb: window staged", 10: "sb: polyphase", 11: "sb: MDCT + alias", 12: "sb: spectra stored",
         34: "an: masking: partition sums", 35: "an: masking: + tonality, prefix sums", 36: "an: masking: + spreading walk"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    n = int(44100 * secs)
    enc = lamehip.Encoder(44100, 128)
    b = lamehip.Batch(enc, B, n)
    base = [helpers.synth_stream(500 + i, n, 44100) for i in range(8)]
    for s in range(B):
        b.set_pcm(s, base[s % 8][0], base[s % 8][1])
    b.encode()
    print("batch %d x %.1f s: %.2f ms, parts %s" % (B, secs, b.kernel_ms(), b.kernel_parts_ms()))
    ssz = enc.lib.lamehip_abi_sizeof(4)
    NP = 44
    tot = np.zeros((2, NP))
    ns = 0
    for s in range(0, B, max(1, B // 64)):
        buf = C.create_string_buffer(ssz)
        assert enc.lib.lamehip_batch_get_state(b.b, s, buf, ssz) == ssz
        tot += np.frombuffer(buf.raw[-2 * NP * 8:], dtype=np.uint64).reshape(2, NP)
        ns += 1
    frames = b.frames(0)
    for w in range(2):
        print("wave %d (cycles per frame)" % w)
        for k in sorted(NAMES):
            print("   %-40s %10.0f" % (NAMES[k], tot[w][k] / ns / frames))


if __name__ == "__main__":
    main()
