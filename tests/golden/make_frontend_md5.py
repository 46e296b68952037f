#!/usr/bin/env python
"""Generates tests/golden/frontend_md5.json: size and MD5 of the files the REFERENCE's frontend, linked with the
reference's own library (oracle/_ref/lame_reference, `make -C oracle frontend-ref`), writes for the command lines
of tests/test_frontend_dropin.py.  Run in the build container (needs /root/reference)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_frontend_dropin as T  # noqa: E402


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "frontend-ref"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "oracle", "_ref", "lame_reference")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        wavs = T.input_files(d)
        for name, wav, args in T.COMMANDS:
            dst = os.path.join(d, "o.mp3")
            subprocess.check_call([exe, "--quiet"] + args + [wavs[wav], dst])
            data = open(dst, "rb").read()
            out[name] = {"size": len(data), "md5": hashlib.md5(data).hexdigest()}
            print(name, out[name])
    json.dump(out, open(os.path.join(HERE, "frontend_md5.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
