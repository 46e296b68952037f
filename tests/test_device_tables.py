"""Huffman table metadata that the kernels carry as immediates (csrc/lh_dev_quant.h)
must equal the generated standard tables (csrc/lh_static_tables.h)."""
import os
import subprocess
import tempfile

import helpers

SRC = r'''
#define LH_EMU
#include "hipemu.h"
#define LH_CONST static const
#include "lh_static_tables.h"
#include "lh_dev_common.h"
#include "lh_dev_quant.h"
#include <stdio.h>

int main(){
  int bad = 0;
  for (int t = 0; t < 34; t++) {
    if (lh_ht_offset[t] >= 0 && lh_ht_off(t) != lh_ht_offset[t]) { printf("off %d\n", t); bad++; }
    if (t < 32 && lh_ht_xlen_c(t) != lh_ht_xlen[t]) { printf("xlen %d %u %u\n", t, lh_ht_xlen_c(t), lh_ht_xlen[t]); bad++; }
    if (t >= 16 && t < 32 && lh_ht_linmax_c(t) != lh_ht_linmax[t]) { printf("linmax %d\n", t); bad++; }
  }
  static const int noesc[15] = { 1, 2, 5, 7, 7, 10, 10, 13, 13, 13, 13, 13, 13, 13, 13 };   /* takehiro.c:505-507 */
  for (unsigned mx = 1; mx <= 15; mx++) if (lh_huf_noESC(mx) != noesc[mx-1]) { printf("noesc %u\n", mx); bad++; }
  if (sizeof(LhLds) > 40960) { printf("LDS image too large: %zu\n", sizeof(LhLds)); bad++; }
  printf("sizeof(LhLds)=%zu bad=%d\n", sizeof(LhLds), bad);
  return bad != 0;
}
'''


def test_huffman_immediates_and_lds_budget():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-mfma", "-Wno-unknown-pragmas",
                               "-I" + os.path.join(helpers.PKG, "csrc"), "-I" + os.path.join(helpers.ROOT, "include"),
                               "-I" + os.path.join(helpers.ROOT, "tests", "hipemu"), "-o", exe, src,
                               os.path.join(helpers.ROOT, "tests", "hipemu", "hipemu.cpp"), "-lm"])
        out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
        assert out.returncode == 0, out.stdout
