"""The reference's own option matrices: every line of /root/reference/test/{CBRABR,VBR,nores,misc}.op
-- what the reference's test/lametest.py:34-66 feeds a lame binary to compare its output byte by byte with a reference
binary's -- through the reference's frontend linked with liblamehip.so (oracle/_ref/lame_frontend) on the reference's
testcase.wav, against size and MD5 of what the reference's frontend writes with the reference's library
(tests/golden/op_matrix_md5.json, made by tests/golden/make_op_matrix_md5.py in the build container).  Every line is
either identical or refused by lame_init_params for a stated reason -- never silently different."""
import hashlib
import json
import os
import shlex
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

import helpers

ROOT = helpers.ROOT
EXE = os.path.join(ROOT, "oracle", "_ref", "lame_frontend")
GOLD = os.path.join(ROOT, "tests", "golden", "op_matrix_md5.json")
WAV = os.path.join(ROOT, "tests", "golden", "testcase.wav")

# option lines the library refuses (lame_init_params returns -1, the frontend exits with an error), and why
REFUSED = {
    "--freeformat -b 33": "free format is out of scope (lame_set_free_format(1) is refused: INTEGRATION.md)",
    "--freeformat -b 330": "free format is out of scope (lame_set_free_format(1) is refused: INTEGRATION.md)",
}


def test_op_matrix_fixture_covers_the_references_files():
    gold = json.load(open(GOLD))
    assert sorted(gold) == ["CBRABR", "VBR", "misc", "nores"]
    assert [len(gold[k]) for k in sorted(gold)] == [118, 195, 16, 3]
    assert all(r["rc"] == 0 and r["size"] > 0 for k in gold for r in gold[k])
    assert all(o in [r["opts"] for r in gold["misc"]] for o in REFUSED)


def _run(job):
    k, i, opts, dst = job
    if os.path.exists(dst):
        os.unlink(dst)
    r = subprocess.run([EXE, "--quiet"] + shlex.split(opts) + [WAV, dst], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0 or not os.path.exists(dst):
        return k, i, None, r.stderr[-300:]
    data = open(dst, "rb").read()
    return k, i, (len(data), hashlib.md5(data).hexdigest()), ""


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/lame_frontend was not built (needs the reference's sources)")
def test_every_line_of_the_references_option_matrices(tmp_path):
    gold = json.load(open(GOLD))
    jobs = []
    for k in sorted(gold):
        for i, r in enumerate(gold[k]):
            jobs.append((k, i, r["opts"], str(tmp_path / ("%s_%d.mp3" % (k, i)))))
    with ThreadPoolExecutor(4) as pool:
        res = list(pool.map(_run, jobs))
    differs, refused = [], {}
    for k, i, got, err in res:
        want = gold[k][i]
        if got is None:
            refused[want["opts"]] = err
        elif got != (want["size"], want["md5"]):
            differs.append((k, want["opts"], got[0], want["size"]))
    assert not differs, differs[:10]
    assert sorted(refused) == sorted(REFUSED), {o: refused[o] for o in refused if o not in REFUSED}
