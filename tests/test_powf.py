"""The device powf restatement (csrc/lh_dev_math.h) against the host libm powf
whose results are baked into the reference's output (athAdjust, NS_INTERP)."""
import os
import subprocess
import tempfile

import helpers

SRC = r'''
#define LH_EMU
#include "lh_dev_math.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
hipemu_state *hipemu_g = 0;
int main(){
  unsigned long long bad=0, n=0;
  for (uint32_t s=0; s<2; s++) for (uint32_t b=0x35000000u; b<0x42700000u; b+=97) {
    float y = lh_u32_as_f32(b | (s<<31)), a = powf(10.0f, y), c = lh_powf(10.0f, y); n++;
    if (memcmp(&a,&c,4)) bad++;
  }
  float rs[2] = {(float)(0.6*0.6f), (float)(0.3*0.6f)};
  for (int k=0;k<2;k++) for (uint32_t b=1; b<0x7f800000u; b+=1013) {
    float x = lh_u32_as_f32(b), a = powf(x, rs[k]), c = lh_powf(x, rs[k]); n++;
    if (memcmp(&a,&c,4)) bad++;
  }
  float sp[] = {0.0f, INFINITY, 1.0f, 1e-45f, 3e38f};
  for (float x: sp) for (float y: {0.36f, 0.18f, -0.5f, 0.0f, 100.0f, -100.0f}) { float a=powf(x,y), c=lh_powf(x,y); n++; if (memcmp(&a,&c,4)) bad++; }
  /* log10f / logf of the VBR quality-7 scalefactor guess */
  for (uint32_t b=1; b<0x7f800000u; b+=211) {
    float x = lh_u32_as_f32(b), a = log10f(x), c = lh_log10f(x), e = logf(x), f = lh_logf(x); n += 2;
    if (memcmp(&a,&c,4)) bad++;
    if (memcmp(&e,&f,4)) bad++;
  }
  printf("%llu %llu\n", n, bad);
  return bad != 0;
}
'''


def test_device_powf_is_bit_identical_to_host_powf():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma",
                               "-I" + os.path.join(helpers.PKG, "csrc"),
                               "-I" + os.path.join(helpers.ROOT, "tests", "hipemu"), "-o", exe, src, "-lm"])
        out = subprocess.check_output([exe]).decode().split()
        assert int(out[0]) > 3_000_000 and int(out[1]) == 0
