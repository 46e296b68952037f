/*
 * lh_lds_analysis.h -- the LDS image of lh_analysis.hip's workgroups (included by lh_dev_common.h in place of the encode
 * kernel's LhLds).  What limits this kernel is how many of its waves a SIMD holds (its chains are serial: section 4 of
 * DESIGN.md, "Round 5"), and what limits those is LDS: one work area is the two channels' sample spans, then -- in place --
 * their FHT buffers, then -- in place again -- the four power spectra; 13.1 KB in all, twelve workgroups per CU (six waves
 * per SIMD) where the fused kernel's psy scratch (19.2 KB) allowed eight.
 */
#define LH_AN_ROW (LH_HBLKSIZE + 3)     /* 516: a power spectrum's row (16-byte aligned rows: the serial sums read four values at a time) */
struct LhLds {
    LhSmallState ss;            /* (named by lh_compute_masking's recurrence half, which is not instantiated there) */
    LhCtxShared ctx;
    LhRgSlot rg[2];
    int     uselong[2];
    int     pad[2];
    float   work[4 * LH_AN_ROW] __attribute__((aligned(16)));  /* samples [2][1024] -> FHT [2][1024] -> spectra [4][516] */
    float   eshort[4][LH_HBLKSIZE_S + 3] __attribute__((aligned(16)));     /* power spectra of one short sub-block */
    float   eb[4 * 64];
    float   thr[4 * 64];        /* lh_compute_masking parks a channel's tonality factors here during the spreading */
};
static_assert(4 * LH_AN_ROW >= 2 * LH_BLKSIZE, "both channels' spans fit the spectra's place");
__shared__ LhLds lh_lds __attribute__((aligned(16)));
