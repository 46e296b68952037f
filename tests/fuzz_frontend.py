#!/usr/bin/env python
"""Randomised hunt at the link-level drop-in (test tool for the GPU box, not collected by pytest): random command lines of
the reference's public switches run through the reference's frontend twice -- once linked with the reference's own library
(oracle/_ref/lame_reference) and once with liblamehip.so (oracle/_ref/lame_frontend) -- and the files compared byte for byte.
Both binaries are built in the build container (`make -C oracle frontend frontend-ref') and travel in oracle/_ref.
Usage: python tests/fuzz_frontend.py [cases] [seed]"""
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_gpu_parity as tg  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "lame_reference")
OURS = os.path.join(ROOT, "oracle", "_ref", "lame_frontend")


def wav(path, pcm, sr, nch):
    data = (pcm.T if nch == 2 else pcm[0]).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, nch, sr, sr * 2 * nch, 2 * nch, 16)
                + b"data" + struct.pack("<I", len(data)) + data)


def pick(rng):
    a = []
    rc = int(rng.integers(0, 6))
    if rc == 0:
        a += ["-b", str(int(rng.choice([96, 112, 128, 160, 192, 224, 256, 320])))]
        if rng.integers(0, 3) == 0:
            a += ["--cbr"]
    elif rc == 1:
        a += ["-V", str(rng.choice(["0", "1", "2", "3", "4", "5", "6", "2.5", "4.7"]))]
        if rng.integers(0, 3) == 0:
            a += ["--vbr-new"]
    elif rc == 2:
        a += ["--vbr-old", "-V", str(int(rng.integers(0, 7)))]
    elif rc == 3:
        a += ["--abr", str(int(rng.integers(100, 300)))]
    elif rc == 4:
        a += ["--preset", str(rng.choice(["standard", "extreme", "insane", "medium", "fast standard", "192", "cbr 160"]))]
        a = a[:1] + a[1].split()
    if rng.integers(0, 4) == 0:
        # a second word on the rate control, after the first (presets apply their values when they are read)
        a += [["-V", str(int(rng.integers(0, 8)))], ["-b", str(int(rng.choice([112, 160, 256])))], ["--abr", str(int(rng.integers(100, 250)))],
              ["--preset", str(rng.choice(["standard", "extreme", "medium", "insane", "128"]))], ["--vbr-old"], ["--vbr-new"], ["--cbr"]][int(rng.integers(0, 7))]
    for _ in range(int(rng.integers(0, 4))):
        k = int(rng.integers(0, 28))
        a += [["-m", str(rng.choice(["s", "j", "f", "m", "d"]))], ["-q", str(int(rng.integers(0, 10)))], ["-k"], ["-p"], ["--nores"],
              ["--lowpass", str(rng.choice(["14", "16.5", "19"]))], ["--highpass", str(rng.choice(["0.2", "1.2"]))],
              ["--scale", str(rng.choice(["0.6", "1.2"]))], ["-Y"], ["-t"], ["--noreplaygain"], ["--strictly-enforce-ISO"],
              ["-B", str(int(rng.choice([160, 224, 320])))], ["-b", str(int(rng.choice([64, 96])))], ["-F"], ["--comp", str(rng.choice(["7", "11"]))],
              ["--resample", str(rng.choice(["32", "44.1", "48"]))], ["-a"], ["--scale-l", "0.7"], ["--scale-r", "1.3"], ["--lowpass-width", "1"],
              ["--nogap"], ["-c"], ["-o"], ["-e", str(rng.choice(["n", "5", "c"]))], ["--clipdetect"], ["--replaygain-fast"], ["-S"]][k]
    return a


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = refused = both_refuse = done = 0
    with tempfile.TemporaryDirectory() as d:
        for c in range(cases):
            sr = int(rng.choice([44100, 48000, 32000, 44100]))
            nch = 1 if rng.integers(0, 5) == 0 else 2
            x = tg._stress_signal(int(rng.integers(0, 1 << 30)), int(sr * float(rng.uniform(0.2, 1.5))), sr)
            src = os.path.join(d, "in.wav")
            wav(src, x, sr, nch)
            args = pick(rng)
            outs = []
            for exe in (REF, OURS):
                dst = os.path.join(d, "o_%s.mp3" % os.path.basename(exe))
                if os.path.exists(dst):
                    os.unlink(dst)
                r = subprocess.run([exe, "--quiet"] + args + [src, dst], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                outs.append(open(dst, "rb").read() if r.returncode == 0 and os.path.exists(dst) else None)
            if outs[0] is None and outs[1] is None:
                both_refuse += 1
            elif outs[1] is None:
                refused += 1
            elif outs[0] != outs[1]:
                bad += 1
                print("MISMATCH case", c, sr, nch, " ".join(args), None if outs[0] is None else len(outs[0]), len(outs[1]), flush=True)
            else:
                done += 1
            if (c + 1) % 50 == 0:
                print("cases", c + 1, "identical", done, "refused by the library", refused, "refused by both", both_refuse, "bad", bad, flush=True)
    print("TOTAL identical", done, "refused by the library", refused, "refused by both", both_refuse, "BAD", bad)


if __name__ == "__main__":
    main()
