#!/usr/bin/env python
"""Generates tests/golden/op_matrix_md5.json: the reference's own option matrices -- every line of
/root/reference/test/{CBRABR,VBR,nores,misc}.op (what test/lametest.py:34-66 feeds a lame binary and compares byte by byte
with a reference binary's output) -- run through the REFERENCE's frontend linked with the reference's own library
(oracle/_ref/lame_reference, `make -C oracle frontend-ref') on the reference's testcase.wav: option line, exit code, size and
MD5 of the file written.  Run in the build container (needs /root/reference); tests/test_op_matrix.py runs the product over
the same lines on the GPU box."""
import hashlib
import json
import os
import shlex
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OPS = ("CBRABR", "VBR", "nores", "misc")


def lines_of(name):
    seen, out = set(), []
    for line in open(os.path.join("/root/reference/test", name + ".op")):
        line = line.strip()
        if line in seen:
            continue            # (nores.op ends in empty lines: the default settings, once)
        seen.add(line)
        out.append(line)
    return out


def run(exe, opts, wav, dst):
    if os.path.exists(dst):
        os.unlink(dst)
    r = subprocess.run([exe, "--quiet"] + shlex.split(opts) + [wav, dst], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0 or not os.path.exists(dst):
        return {"opts": opts, "rc": r.returncode if r.returncode else 1}
    data = open(dst, "rb").read()
    return {"opts": opts, "rc": 0, "size": len(data), "md5": hashlib.md5(data).hexdigest()}


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "frontend-ref"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "oracle", "_ref", "lame_reference")
    wav = os.path.join(HERE, "testcase.wav")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name in OPS:
            out[name] = [run(exe, opts, wav, os.path.join(d, "o.mp3")) for opts in lines_of(name)]
            print(name, len(out[name]), "lines,", sum(1 for r in out[name] if r["rc"]), "refused by the reference's frontend")
    json.dump(out, open(os.path.join(HERE, "op_matrix_md5.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
