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


GRID_SRC = r'''
#define LH_CONST static const
#include "lh_static_tables.h"
#include <stdio.h>
#include <string.h>

/* the grids filled table by table (the host fills them cell by cell): field f of the cell of (x, y) */
static unsigned grid[704];
static void put(int base, int cols, int t, int field, int x0, int y0)
{
    int n = (t == 14) ? 16 : lh_ht_xlen[t];     /* the reference's ht[14] has no alphabet size of its own: it is read as 16 x 16 next to 13 and 15 (takehiro.c:598-616) */
    for (int x = 0; x < n; x++)
        for (int y = 0; y < n; y++) {
            unsigned len = lh_ht_len[lh_ht_offset[t] + x * n + y];
            grid[base + (x0 + x) * cols + (y0 + y)] |= len << (10 * field);
        }
}
int main(void)
{
    memset(grid, 0, sizeof grid);
    for (int i = 0; i < 256; i++) {
        unsigned e = lh_largetbl[i];
        grid[i] = (e >> 16) | ((e & 0xffffu) << 10) | ((unsigned) (((i >> 4) == 15) + ((i & 15) == 15)) << 20);
    }
    put(256, 16, 13, 0, 0, 0); put(256, 16, 14, 1, 0, 0); put(256, 16, 15, 2, 0, 0);
    put(512, 16, 10, 0, 0, 0); put(512, 16, 11, 1, 0, 0); put(512, 16, 12, 2, 0, 0);
    put(512, 16, 7, 0, 0, 8); put(512, 16, 8, 1, 0, 8); put(512, 16, 9, 2, 0, 8);
    put(512, 16, 5, 0, 8, 0); put(512, 16, 6, 1, 8, 0); put(512, 16, 5, 2, 8, 0);
    put(512, 16, 2, 0, 8, 4); put(512, 16, 3, 1, 8, 4); put(512, 16, 2, 2, 8, 4);
    put(512, 16, 1, 0, 8, 8); put(512, 16, 1, 1, 8, 8); put(512, 16, 1, 2, 8, 8);
    for (int i = 0; i < 704; i++) printf("%u\n", grid[i]);
    return 0;
}
'''


def test_huffman_length_grids_of_the_search():
    """LhTables.hgrid (built by lh_host_init.c, copied into LDS by lq_load) against the standard tables, filled
    here table by table: ESC grid from the packed ESC lengths, tables 13-15, and the small alphabets at their
    origins (a table with fewer than three candidates repeats its first)."""
    import numpy as np
    import lamehip
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "g.c")
        open(src, "w").write(GRID_SRC)
        exe = os.path.join(d, "g")
        subprocess.check_call(["gcc", "-O1", "-I" + os.path.join(helpers.PKG, "csrc"), "-o", exe, src])
        want = np.array([int(x) for x in subprocess.check_output([exe], text=True).split()], dtype=np.uint32)
    enc = lamehip.Encoder(44100, 128, require_device=False)
    got = np.ctypeslib.as_array(enc.tables().hgrid).astype(np.uint32)
    assert np.array_equal(got, want)
    enc.close()
