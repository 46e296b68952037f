"""The two double-precision expressions of the old VBR loop (reference quantize.c:1420-1428) as the device evaluates
them (csrc/lh_dev_math.h: lh_vbrold_adjust, lh_vbrold_masking_lower) against this host's libm.  Both take a float and
give a float, so tools/sweep_vbrold_math.c can compare them on EVERY input of their domains (7.1 10^9 evaluations at stride
1, negative pe included: 0 differ on glibc 2.35); the test suite runs every 257th."""
import os
import subprocess

import helpers


def test_device_exp_and_pow_match_libm_on_a_sweep(tmp_path):
    exe = str(tmp_path / "sweep_vbrold_math")
    subprocess.check_call(["gcc", "-O2", "-fno-fast-math", "-ffp-contract=off", "-fopenmp", "-DLH_EMU",
                           "-I", os.path.join(helpers.ROOT, "deprecated-lame-mirror_amd", "csrc"),
                           "-I", os.path.join(helpers.ROOT, "include"),
                           os.path.join(helpers.ROOT, "tools", "sweep_vbrold_math.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe, "257"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert " 0 differ" in out.stdout
