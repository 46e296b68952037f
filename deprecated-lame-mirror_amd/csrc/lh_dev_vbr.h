/*
 * lh_dev_vbr.h -- the "new VBR" (vbr_mt / vbr_mtrh) quantisation loop on the device.
 *
 * What it computes is the reference's VBR_new_iteration_loop + VBR_encode_frame
 * (reference libmp3lame/quantize.c:1582-1751, libmp3lame/vbrquantize.c); how it is
 * laid out is not: the reference walks the scalefactor bands one after the other
 * and bisects each band's step size with up to 24 serial noise evaluations.  Here a
 * wave owns one channel and
 *   - all bands of the granule bisect together (lane = band holds the search state),
 *   - a noise evaluation is two phases: lanes = groups of four lines quantise and
 *     form the group's double-precision error sum, then lane = band adds its groups
 *     in the reference's order (float accumulator, double addend), which is the only
 *     serial chain left (<= 48 steps),
 *   - the constraint solvers (long_block_constrain / short_block_constrain /
 *     set_subblock_gain / set_scalefacs) are wave reductions + one lane per band.
 * Bit-exactness rule as everywhere: every float/double expression keeps the
 * reference's operand types and order (TAKEHIRO_IEEE754_HACK build).
 *
 * LDS (per channel image Q, fields that only the CBR loop uses are re-used):
 *   Q.xrpow        |xr|^(3/4)                       Q.l3_xmin   allowed distortion
 *   Q.save_xrpow   double group sums [<= 183]       Q.sfb_mode  energy_above_cutoff (from calc_xmin)
 *   Q.pn_noise..   group -> band bytes [<= 183]     Q.pn_step   lines of the band that take part (n)
 *   Q.pseudohalf   first group of the band          Q.sfb_f / Q.distort   ipow20 / pow20 of the band's trial step
 */
#ifndef LH_DEV_VBR_H
#define LH_DEV_VBR_H

#include "lh_dev_quant.h"

LH_DEVFN int
lh_imax(int a, int b)
{
    return a > b ? a : b;
}

LH_DEVFN int
lh_imin(int a, int b)
{
    return a < b ? a : b;
}

/* maximum over the wave of values that start from 0 (the reference's "if (m < v) m = v" with m = 0) */
LH_DEVFN int
lh_wave_max0(int v)
{
    return (int) lh_wave_max_u32((unsigned) (v > 0 ? v : 0));
}

LH_DEVFN int
lh_wave_min_i32(int v)
{
    return (int) (lh_wave_min_u32((unsigned) v ^ 0x80000000u) ^ 0x80000000u);
}

/* scalefactor ranges of the side information (reference vbrquantize.c:527-539, MPEG-1) */
LH_DEVFN int
lh_vbr_range_long(int s)
{
    return s < 11 ? 15 : (s < 21 ? 7 : 0);
}

LH_DEVFN int
lh_vbr_range_short(int s)
{
    return s < 18 ? 15 : (s < 36 ? 7 : 0);
}

/* pow43[] with its LDS head */
LH_DEVFN float
lh_pow43(const LhTables * T, const LhQTabs * qt, int k)
{
    float   v = qt->pow43h[k & 255];
    if (k >= 256)
        v = T->pow43[k];
    return v;
}

/* ---- per-granule geometry of the search --------------------------------------------- */
struct LhVbrGeo {
    int     nb;                 /* bands the block type has (22 / 39) */
    int     visited;            /* this lane's band starts at or below max_nonzero_coeff */
    int     n;                  /* its lines that take part: min(width, room) */
    int     start;
    int     ngroups;            /* groups of four lines over all searched bands */
    int     ng, gstart;         /* this band's groups */
    int     maxng;              /* longest band in groups (wave-uniform) */
    int     nfused;             /* trial steps whose group sums fit side by side (432 doubles of scratch) */
};

LH_DEVFN LhVbrGeo
lh_vbr_geometry(const LhCtx & c, LhChanLds & Q, const LhQR & R)
{
    LhVbrGeo G;
    int const s = c.lane;
    int const sc = s < LH_SFBMAX ? s : LH_SFBMAX;
    int const width = Q.width[sc];
    uint8_t *gband = (uint8_t *) Q.pn_noise;
    G.nb = (R.block_type == LH_SHORT_TYPE) ? 39 : 22;
    G.start = Q.start[sc];
    G.visited = (s < G.nb) && (G.start <= R.mnc);
    G.n = G.visited ? lh_imin(width, R.mnc - G.start + 1) : 0;
    /* only bands whose step is searched get groups: below psymax and with energy above the
     * masking threshold (block_sf, reference vbrquantize.c:437-438) */
    G.ng = (G.visited && s < R.psymax && Q.sfb_mode[sc]) ? (G.n + 3) >> 2 : 0;
    G.gstart = 0;
    for (int u = 0; u < 39; u++) {
        int const t = (int) lh_bcast_u32((unsigned) G.ng, u);
        G.gstart += (u < s) ? t : 0;
    }
    G.ngroups = (int) lh_wave_sum_u32((unsigned) G.ng);
    G.maxng = (int) lh_wave_max_u32((unsigned) G.ng);
    G.nfused = (3 * G.ngroups <= 432) ? 3 : 2;
    LH_WAVE_SYNC();
    if (s <= LH_SFBMAX) {
        Q.pn_step[s] = G.n;
        Q.pseudohalf[s] = G.gstart;
    }
    for (int i = 0; i < G.ng; i++)
        gband[G.gstart + i] = (uint8_t) s;
    LH_WAVE_SYNC();
    return G;
}

/* "Too noisy?" for every band at up to three trial steps at once: bad[v] = xmin < noise(sf[v])
 * for the lanes (bands) with want[v] != 0, where noise is calc_sfb_noise_x34 (reference
 * vbrquantize.c:210-262).  NV = how many variants run side by side (G.nfused).
 *
 * The reference's noise is a float accumulator that takes one double group sum after the other;
 * that serial chain (<= 48 links per band) is only walked when it can matter: all terms are
 * non-negative, so the chain's result lies within n * 2^-24 (relative) of the true sum, and the
 * groups also add their sums to a per-band float with LDS atomics (any order, same bound).  When
 * xmin is farther than 6e-5 (relative) from that approximate sum -- almost always -- the comparison
 * is decided; otherwise the exact chain runs. */
#ifndef LH_VBR_MARGIN
#define LH_VBR_MARGIN 6e-5      /* tests widen it to drive every comparison through the exact chain */
#endif
/* -1 (all ones) when x < 0, else 0, as plain vector arithmetic: "a < b" as lh_sign_mask(a - b) is exact (a
 * float difference is negative exactly when a < b; denormals are kept) and, unlike a comparison, does not
 * pass through a scalar register pair, where a run of compare / use pairs would queue up */
LH_DEVFN uint32_t
lh_sign_mask(float x)
{
    return (uint32_t) ((int32_t) lh_f32_as_u32(x) >> 31);
}

/* Inclusive sums of f[] over runs of equal keys inside the rows of 16 lanes (the lanes of a run are neighbours, keys < 2^32 - 1):
 * afterwards the last lane a run has in a row holds the sums over the run's lanes of that row.  Four row_shr steps on the
 * cross-lane network, no LDS.  The additions form a tree; for sums whose use tolerates any order. */
template < int N > LH_DEVFN void
lh_row_seg_scan_addf(float (&f)[N], uint32_t key)
{
    uint32_t const k = key + 1u;        /* a lane without a source reads 0, which is no key */
#define LH_SEG_STEP(D) { \
        uint32_t const kd_ = lh_row_shr_u32 < D > (k); \
        _Pragma("unroll") for (int i_ = 0; i_ < N; i_++) { \
            float const fd_ = lh_u32_as_f32(lh_row_shr_u32 < D > (lh_f32_as_u32(f[i_]))); \
            f[i_] += (kd_ == k) ? fd_ : 0.0f; } }
    LH_SEG_STEP(1) LH_SEG_STEP(2) LH_SEG_STEP(4) LH_SEG_STEP(8)
#undef LH_SEG_STEP
}

/* table[i] of a table in HBM with the byte offset formed in 32 bits: a wave-uniform base plus a 32-bit lane
 * offset is one addressing mode of the global load (no 64-bit address arithmetic per lane, and no shared
 * zero register that would serialise a batch of look-ups) */
LH_DEVFN lh_f32x4
lh_gather_f32x4(const float *table, uint32_t i)
{
    return *(const lh_f32x4 *) ((const char *) table + (i << 4));
}

template < int NV > LH_DEVFN void
lh_vbr_noisy_n(const LhCtx & c, LhChanLds & Q, const LhVbrGeo & G, const float *xr, float xmin, const int sf[3],
               const int want[3], int bad[3])
{
    const LhTables *T = c.T;
    const uint8_t *gband = (const uint8_t *) Q.pn_noise;
    double *gsum = (double *) Q.save_xrpow;     /* NV x G.ngroups doubles: save_xrpow and ix[0] behind it */
    int    *step = (int *) Q.sfb_f;             /* the band's trial steps, 9 bits each (511 = none) */
    float  *approx[3] = { Q.distort, Q.l3_xmin, (float *) Q.sf[1] };
    int const s = c.lane;
    int     on[3], maxng = 0, ambiguous = 0;
    LH_PT(t_n);
    LH_PC(13);
    LH_WAVE_SYNC();
    {
        unsigned pack = 0;
#pragma unroll
        for (int v = 0; v < 3; v++) {
            int const sfc = sf[v < NV ? v : 0] < 0 ? 0 : (sf[v < NV ? v : 0] > 255 ? 255 : sf[v < NV ? v : 0]);
            on[v] = (v < NV) && want[v] && G.visited;
            /* bands that are not asked for quantise with step 0: their trial step may lie below the
             * band's floor, where the quantiser's rounding trick no longer yields a table index */
            pack |= (unsigned) (on[v] ? sfc : 511) << (9 * v);
            if (s <= LH_SFBMAX)
                approx[v][s] = 0.0f;
        }
        if (s <= LH_SFBMAX)
            step[s] = (int) pack;
    }
    LH_WAVE_SYNC();
    /* phase A: lane = group of four lines of one band */
#pragma unroll
    for (int r = 0; r < 3; r++) {
        if (64 * r < G.ngroups) {
            int const gi = s + 64 * r;
            int const gic = gi < G.ngroups ? gi : 0;
            int const b = gband[gic];
            int const k0 = 4 * (gic - Q.pseudohalf[b]);
            int const line = Q.start[b] + k0;
            int const cnt = Q.pn_step[b] - k0;  /* >= 1 */
            unsigned const pack = (unsigned) step[b];
            float   x34[4], ax[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int const lc = (k < cnt) ? line + k : line;
                x34[k] = Q.xrpow[lc];
                ax[k] = lh_fabsf(xr[lc]);
            }
            /* straight-line code.  The first rounding is a float addition, the second a comparison with the class's
             * threshold (LhTables.vq3, tests/test_quantizer_identity.py), which sits beside the two values of pow43 the class
             * can take: ONE 16-byte look-up per line and variant, all of them in flight together.  They go to HBM (L1/L2
             * resident) unconditionally; trial steps below the final one quantise to large values, which LDS heads would
             * not cover */
            {
                float   sfpow[NV];
                float   a[NV][4];
                lh_f32x4 cls[NV][4];
                float   p43[NV][4], part[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    int const sv = (int) ((pack >> (9 * v)) & 511u);
                    int const svc = sv > 255 ? 0 : sv;
                    float const sfpow34 = sv > 255 ? 0.0f : LH_VBR_IPOW20[svc];
                    sfpow[v] = LH_VBR_POW20[svc];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        a[v][k] = sfpow34 * x34[k];
                        uint32_t const k1 = lh_f32_as_u32(a[v][k] + (float) LH_MAGIC_FLOAT) - (uint32_t) LH_MAGIC_INT;
                        cls[v][k] = lh_gather_f32x4(&T->vq3[0][0], k1);
                    }
                }
                LH_SCHED_FENCE();
#pragma unroll
                for (int v = 0; v < NV; v++)
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        {
                        uint32_t const below = lh_sign_mask(a[v][k] - cls[v][k].x);
                        p43[v][k] = lh_u32_as_f32((lh_f32_as_u32(cls[v][k].y) & below) | (lh_f32_as_u32(cls[v][k].z) & ~below));
                    }
                LH_SCHED_FENCE();
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    double  e[4], gs;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float const d = ax[k] - sfpow[v] * p43[v][k];
                        e[k] = (k < cnt) ? (double) d : 0.0;
                    }
                    gs = (e[0] * e[0] + e[1] * e[1]) + (e[2] * e[2] + e[3] * e[3]);
                    if (gi < G.ngroups)
                        gsum[v * G.ngroups + gi] = gs;
                    part[v] = (gi < G.ngroups) ? (float) gs : 0.0f;
                }
                /* the approximate band sums: a band's groups are neighbouring lanes -- summed over the cross-lane network inside
                 * the rows of 16 lanes, and the last lane a band has in a row adds the row's share (64 lanes adding to a handful
                 * of LDS words one after the other was a fifth of the VBR frame: profiles/r05_vbr_stage_profile.txt) */
                lh_row_seg_scan_addf < NV > (part, (gi < G.ngroups) ? (uint32_t) b : 0xfffffff0u);
                if (gi < G.ngroups && ((s & 15) == 15 || cnt <= 4)) {
#pragma unroll
                    for (int v = 0; v < NV; v++)
                        lh_lds_addf(&approx[v][b], part[v]);
                }
            }
        }
    }
    LH_WAVE_SYNC();
    LH_PA(12, t_n);
#pragma unroll
    for (int v = 0; v < 3; v++) {
        /* |chain - sum| <= n 2^-24 sum, |approx - sum| <= (n + 1) 2^-24 sum, n <= 48 */
        double const a = (double) approx[v][s <= LH_SFBMAX ? s : LH_SFBMAX];
        double const x = (double) xmin;
        bad[v] = on[v] && (x < a * (1.0 - LH_VBR_MARGIN));
        if (on[v] && !bad[v] && !(x > a * (1.0 + LH_VBR_MARGIN))) {
            ambiguous = 1;
            maxng = G.ng > maxng ? G.ng : maxng;
        }
    }
    if (LH_RARE(lh_ballot(ambiguous))) {
        /* exact: lane = band, float accumulator += double group sum, in order */
        float   acc[3] = { 0.0f, 0.0f, 0.0f };
        LH_PC(14);
        maxng = (int) lh_wave_max_u32((unsigned) maxng);
        for (int i = 0; i < maxng; i++) {
            int const in = i < G.ng;
#pragma unroll
            for (int v = 0; v < NV; v++) {
                int const idx = (on[v] && in) ? v * G.ngroups + G.gstart + i : 0;
                double const gv = gsum[idx];
                float const nx = (float) ((double) acc[v] + gv);
                acc[v] = (on[v] && in) ? nx : acc[v];
            }
        }
        if (ambiguous) {
#pragma unroll
            for (int v = 0; v < NV; v++)
                bad[v] = on[v] && (xmin < acc[v]);
        }
    }
    LH_PA(11, t_n);
}

/* block_sf (reference vbrquantize.c:397-489): per-band step indices sfw (lane = band) and the
 * floor sfm below which the band would overflow the quantiser; returns vbrmax */
LH_DEVFN int
lh_vbr_band_steps(const LhCtx & c, LhChanLds & Q, const LhQR & R, const LhVbrGeo & G, const float *xr,
                  int &sfw, int &sfm, int &mingain_l, int mingain_s[3])
{
    int const s = c.lane;
    int const sc = s < LH_SFBMAX ? s : LH_SFBMAX;
    int     m1, m2 = 0;
    /* band maxima of |xr|^(3/4): lines above mnc are zero in Q.xrpow */
    LH_WAVE_SYNC();
    if (s <= LH_SFBMAX)
        Q.sfb_f[s] = 0.0f;
    LH_WAVE_SYNC();
    for (int i = s; i < 576; i += 64)
        lh_lds_max((int *) &Q.sfb_f[Q.sfb_of_line[i]], (int) lh_f32_as_u32(Q.xrpow[i]));
    LH_WAVE_SYNC();
    {
        /* find_lowest_scalefac (reference vbrquantize.c:142-159) */
        float const xmax = Q.sfb_f[sc];
        int     sf = 128, del = 64;
        m1 = 255;
        for (int k = 0; k < 8; k++) {
            float const v = LH_VBR_IPOW20[sf] * xmax;
            if (v <= (float) LH_IXMAX) {
                m1 = sf;
                sf -= del;
            }
            else
                sf += del;
            del >>= 1;
        }
    }
    sfm = G.visited ? m1 : 0;
    mingain_l = lh_wave_max0(sfm);
    mingain_s[0] = lh_wave_max0((s % 3 == 0) ? sfm : 0);
    mingain_s[1] = lh_wave_max0((s % 3 == 1) ? sfm : 0);
    mingain_s[2] = lh_wave_max0((s % 3 == 2) ? sfm : 0);
    {
        int const regular = G.visited && s < R.psymax;      /* widths are >= 4 in MPEG-1 */
        int const active = regular && Q.sfb_mode[sc];
        float const xmin = Q.l3_xmin[sc];
        /* find_scalefac_x34 (reference vbrquantize.c:347-382), all bands at once; the memo of the
         * reference only saves work, the noise of a step is a pure function */
        int     sf = 128, ok = 255, del = 128, seen = 0;
        if (c.full_outer_loop < 0) {
            /* quality 7..9: closed-form estimate instead of the search (calc_scalefac +
             * guess_scalefac_x34, reference vbrquantize.c:315-333) */
            float const cc = 5.799142446f;
            int const gs = 210 + (int) (cc * lh_log10f(xmin / (float) (G.n > 0 ? G.n : 1)) - .5f);
            sf = gs < m1 ? m1 : (gs >= 255 ? 255 : gs);
        }
        else
        for (int k = 0; k < 8; k++) {
            int const skip = (sf <= m1);
            int const need = active && !skip;
            int     bad;
            del >>= 1;
            {
                /* tri_calc_sfb_noise_x34 (reference vbrquantize.c:275-308): too noisy at sf, sf + 1
                 * or sf - 1.  The three are evaluated side by side when the scratch holds them. */
                int const sfv[3] = { sf, sf + 1, sf - 1 };
                int     wv[3] = { need, need && sf < 255, need && sf > 0 };
                int     bv[3];
                if (G.nfused >= 3)
                    lh_vbr_noisy_n < 3 > (c, Q, G, xr, xmin, sfv, wv, bv);
                else {
                    lh_vbr_noisy_n < 2 > (c, Q, G, xr, xmin, sfv, wv, bv);
                    wv[0] = wv[2] && !bv[0] && !bv[1];
                    bv[2] = 0;
                    if (lh_ballot(wv[0])) {
                        int const sf1[3] = { sf - 1, 0, 0 };
                        int     b1[3];
                        wv[1] = wv[2] = 0;
                        lh_vbr_noisy_n < 1 > (c, Q, G, xr, xmin, sf1, wv, b1);
                        bv[2] = b1[0];
                    }
                }
                bad = bv[0] || bv[1] || bv[2];
            }
            if (skip)
                sf += del;
            else if (bad)
                sf -= del;
            else {
                ok = sf;
                sf += del;
                seen = 1;
            }
        }
        if (c.full_outer_loop >= 0) {
            if (seen)
                sf = ok;
            if (sf <= m1)
                sf = m1;
        }
        m2 = sf;
        {
            int     maxsf = lh_wave_max0(regular ? (active ? m2 : 255) : 0);
            int const below = lh_wave_max0((active && m2 < 255) ? m2 + 1 : 0) - 1;  /* m_o */
            /* bands past psymax (sfb21 / sfb12 when they carry no scalefactor): running maximum */
            int const t0 = (int) lh_bcast_u32((unsigned) sfm, lh_imin(R.psymax, 63));
            int const t1 = (int) lh_bcast_u32((unsigned) sfm, lh_imin(R.psymax + 1, 63));
            int const t2 = (int) lh_bcast_u32((unsigned) sfm, lh_imin(R.psymax + 2, 63));
            int const r0 = lh_imax(maxsf, t0), r1 = lh_imax(r0, t1), r2 = lh_imax(r1, t2);
            int const k = s - R.psymax;
            int const tail = (k == 0) ? r0 : (k == 1) ? r1 : r2;
            /* sfm is 0 for bands that are not visited, so r2 is the final running maximum */
            maxsf = r2;
            if (regular)
                sfw = active ? m2 : 255;
            else if (G.visited)
                sfw = tail;
            else
                sfw = maxsf;
            if (below > -1) {
                maxsf = below;
                if (sfw == 255)
                    sfw = below;
            }
            return maxsf;
        }
    }
}

/* set_scalefacs (reference vbrquantize.c:653-700); sft = step - vbrmax of lane's band */
LH_DEVFN void
lh_vbr_scalefacs(const LhCtx & c, LhChanLds & Q, const LhQR & R, const LhGrR & g, int sfm, int sft, int range)
{
    const LhQTabs *qt = LH_QT;
    int const s = c.lane;
    int const sc = s < LH_SFBMAX ? s : LH_SFBMAX;
    int const ifqstep = (g.scalefac_scale == 0) ? 2 : 4;
    int const shift = (g.scalefac_scale == 0) ? 1 : 2;
    int const pre = (g.preflag && s < 22) ? (int) qt->pretab[s < 22 ? s : 0] : 0;
    int     sc_out = 0;
    if (g.preflag && s >= 11)
        sft += pre * ifqstep;
    if (s < R.sfbmax && sft < 0) {
        int const gain = g.global_gain - lh_sbg(g, Q.window[sc]) * 8 - pre * ifqstep;
        int const m = gain - sfm;
        sc_out = (ifqstep - 1 - sft) >> shift;
        if (sc_out > range)
            sc_out = range;
        if (sc_out > 0 && (sc_out << shift) > m)
            sc_out = m >> shift;
    }
    LH_WAVE_SYNC();
    if (s < LH_SFBMAX)
        Q.sf[0][s] = sc_out;
    LH_WAVE_SYNC();
}

/* long_block_constrain (reference vbrquantize.c:826-978) */
LH_DEVFN void
lh_vbr_constrain_long(const LhCtx & c, LhChanLds & Q, const LhQR & R, LhGrR & g, int sfw, int sfm, int vbrmax,
                      int mingain_l)
{
    const LhQTabs *qt = LH_QT;
    int const s = c.lane;
    int const in = s < R.psymax;
    int const r = lh_vbr_range_long(s);
    /* the range with preflag: MPEG-1's own, or the LSF partitions' 7 7 7 7 7 7 3 3 3 3 3 0 ... (max_range_long_lsf_pretab,
     * reference vbrquantize.c:577-579, 861) */
    int const rp = LH_IS_LSF ? (s < 6 ? 7 : (s < 11 ? 3 : 0)) : r;
    int const pt = (s < 22) ? (int) qt->pretab[s < 22 ? s : 0] : 0;
    int const v = vbrmax - sfw;
    int     delta = lh_wave_max0(in ? v : 0);
    int     over0 = lh_wave_max0(in ? v - 2 * r : 0);
    int     over1 = lh_wave_max0(in ? v - 4 * r : 0);
    int     over0p = lh_wave_max0(in ? v - 2 * (rp + pt) : 0);
    int     over1p = lh_wave_max0(in ? v - 4 * (rp + pt) : 0);
    int     pre0, pre1 = 0, mover;
    {
        int const gain = lh_imax(vbrmax - over0p, mingain_l);
        pre0 = !lh_ballot(in && (gain - sfm) - 2 * pt <= 0);
    }
    if (pre0) {
        int const gain = lh_imax(vbrmax - over1p, mingain_l);
        pre1 = !lh_ballot(in && (gain - sfm) - 4 * pt <= 0);
    }
    if (!pre0)
        over0p = over0;
    if (!pre1)
        over1p = over1;
    if (c.ns != 2) {
        over1 = over0;
        over1p = over0p;
    }
    mover = lh_imin(lh_imin(over0, over0p), lh_imin(over1, over1p));
    if (delta > mover)
        delta = mover;
    vbrmax -= delta;
    if (vbrmax < mingain_l)
        vbrmax = mingain_l;
    over0 -= mover;
    over0p -= mover;
    over1 -= mover;
    over1p -= mover;
    if (over0 == 0) {
        g.scalefac_scale = 0;
        g.preflag = 0;
    }
    else if (over0p == 0) {
        g.scalefac_scale = 0;
        g.preflag = 1;
    }
    else if (over1 == 0) {
        g.scalefac_scale = 1;
        g.preflag = 0;
    }
    else if (over1p == 0) {
        g.scalefac_scale = 1;
        g.preflag = 1;
    }
    g.global_gain = vbrmax < 0 ? 0 : (vbrmax > 255 ? 255 : vbrmax);
    lh_vbr_scalefacs(c, Q, R, g, sfm, sfw - vbrmax, g.preflag ? rp : r);
}

/* short_block_constrain + set_subblock_gain (reference vbrquantize.c:748-815, 553-642) */
LH_DEVFN void
lh_vbr_constrain_short(const LhCtx & c, LhChanLds & Q, const LhQR & R, LhGrR & g, int sfw, int sfm, int vbrmax,
                       int mingain_l, const int mingain_s[3])
{
    int const s = c.lane;
    int const in = s < R.psymax;
    int const r = lh_vbr_range_short(s);
    int const v = vbrmax - sfw;
    int     delta = lh_wave_max0(in ? v : 0);
    int     over0 = lh_wave_max0(in ? v - (4 * 14 + 2 * r) : 0);
    int     over1 = lh_wave_max0(in ? v - (4 * 14 + 4 * r) : 0);
    int     mover, sft;
    if (c.ns == 2)
        mover = lh_imin(over0, over1);
    else
        mover = over0;
    if (delta > mover)
        delta = mover;
    vbrmax -= delta;
    over0 -= mover;
    over1 -= mover;
    if (over0 == 0)
        g.scalefac_scale = 0;
    else if (over1 == 0)
        g.scalefac_scale = 1;
    if (vbrmax < mingain_l)
        vbrmax = mingain_l;
    g.global_gain = vbrmax < 0 ? 0 : (vbrmax > 255 ? 255 : vbrmax);
    sft = sfw - vbrmax;
    {
        /* set_subblock_gain: lane = band, window = band % 3 */
        int const shift = (g.scalefac_scale == 0) ? 1 : 2;
        int const psydiv = lh_imin(18, R.psymax);
        int const nv = -sft;
        int const win = s % 3;
        int const inall = s < LH_SFBMAX;
        int     sbg[3], min_sbg = 7;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int const mine = inall && win == i;
            int     need1 = lh_wave_max0((mine && s < psydiv) ? nv : 0);
            int const need2 = lh_wave_max0((mine && s >= psydiv) ? nv : 0);
            int const least = lh_imin(1000, lh_wave_min_i32(mine ? nv : 1000));
            int const a = need1 - (15 << shift), b = need2 - (7 << shift);
            int     x;
            need1 = lh_imax(a, b);
            x = (least > 0) ? (least >> 3) : 0;
            if (need1 > 0)
                x = lh_imax(x, (need1 + 7) >> 3);
            if (x > 0 && mingain_s[i] > (g.global_gain - x * 8))
                x = (g.global_gain - mingain_s[i]) >> 3;
            if (x > 7)
                x = 7;
            if (min_sbg > x)
                min_sbg = x;
            sbg[i] = x;
        }
        sft += (win == 0 ? sbg[0] : win == 1 ? sbg[1] : sbg[2]) * 8;
        if (min_sbg > 0) {
            sbg[0] -= min_sbg;
            sbg[1] -= min_sbg;
            sbg[2] -= min_sbg;
            g.global_gain -= min_sbg * 8;
        }
        g.subblock_gain[0] = sbg[0];
        g.subblock_gain[1] = sbg[1];
        g.subblock_gain[2] = sbg[2];
    }
    lh_vbr_scalefacs(c, Q, R, g, sfm, sft, r);
}

/* that->alloc() + bitcount() */
LH_DEVFN void
lh_vbr_constrain(const LhCtx & c, LhChanLds & Q, const LhQR & R, LhGrR & g, int sfw, int sfm, int vbrmax,
                 int mingain_l, const int mingain_s[3])
{
    if (R.block_type == LH_SHORT_TYPE)
        lh_vbr_constrain_short(c, Q, R, g, sfw, sfm, vbrmax, mingain_l, mingain_s);
    else
        lh_vbr_constrain_long(c, Q, R, g, sfw, sfm, vbrmax, mingain_l);
    (void) lh_scale_bitcount(c, Q, R, g, 0);
}

/* quantizeAndCountBits (reference vbrquantize.c:996-1002 with quantize_x34 :505-570): lane = pair */
LH_DEVFN int
lh_vbr_quantize_count(const LhCtx & c, LhChanLds & Q, LhQR & R, LhGrR & g)
{
    const LhTables *T = c.T;
    const LhQTabs *qt = LH_QT;
    uint32_t *ix2 = (uint32_t *) Q.ix[0];
    int const s = c.lane;
    int const pm = R.mnc >> 1;
    uint32_t pk[5];
    LH_WAVE_SYNC();
    if (s <= LH_SFBMAX) {
        int const sc = s < LH_SFBMAX ? s : LH_SFBMAX - 1;
        int const ifqstep = (g.scalefac_scale == 0) ? 2 : 4;
        int const pre = (g.preflag && s < 22) ? (int) qt->pretab[s < 22 ? s : 0] : 0;
        int const st = (Q.sf[0][sc] + pre) * ifqstep + lh_sbg(g, Q.window[sc]) * 8;
        Q.sfb_f[s] = LH_VBR_IPOW20[(unsigned) (g.global_gain - st) & 255u];
    }
    LH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 5; k++) {
        int const p = s + 64 * k;
        int const pc = (k < 4 || p < 288) ? p : 287;
        lh_f32x2 const x2 = ((const lh_f32x2 *) Q.xrpow)[pc];
        float const i0 = Q.sfb_f[Q.sfb_of_line[2 * pc]], i1 = Q.sfb_f[Q.sfb_of_line[2 * pc + 1]];
        int const q0 = lh_quant_line(T, qt, i0, x2.x);
        int const q1 = lh_quant_line(T, qt, i1, x2.y);
        uint32_t v = (uint32_t) (q0 & 0xffff) | ((uint32_t) q1 << 16);
        if (p > pm)
            v = 0u;             /* lines above max_nonzero_coeff stay zero */
        pk[k] = v;
        if (k < 4 || p < 288)
            ix2[p] = v;
    }
    LH_WAVE_SYNC();
    g.part2_3_length = lh_noquant_count_bits(c, Q, R, g, 0, 0, pk);
    return g.part2_3_length;
}

/* tryThatOne (reference vbrquantize.c:1139-1150) */
LH_DEVFN int
lh_vbr_try(const LhCtx & c, LhChanLds & Q, LhQR & R, LhGrR & g, int steps, int sfm, int vbrmax, int mingain_l,
           const int mingain_s[3])
{
    lh_vbr_constrain(c, Q, R, g, steps, sfm, vbrmax, mingain_l, mingain_s);
    return lh_vbr_quantize_count(c, Q, R, g) + g.part2_length;
}

/* flattenDistribution (reference vbrquantize.c:1101-1136), lane = band; returns the new maximum */
LH_DEVFN int
lh_vbr_flatten(const LhCtx & c, int in, int &out, int dm, int k, int p)
{
    int     x = in;
    if (dm > 0) {
        x = in + (k * (p - in)) / dm;
        x = x < 0 ? 0 : (x > 255 ? 255 : x);
    }
    out = x;
    return lh_wave_max0(c.lane < LH_SFBMAX ? x : 0);
}

/* outOfBitsStrategy (reference vbrquantize.c:1153-1228); steps = this lane's (cut) sfwork */
LH_DEVFN void
lh_vbr_fit(const LhCtx & c, LhChanLds & Q, LhQR & R, LhGrR & g, int steps, int sfm, int target, int mingain_l,
           const int mingain_s[3])
{
    int const inb = c.lane < LH_SFBMAX;
    int const dm = lh_wave_max0(inb ? 255 - steps : 0);
    int const p = g.global_gain;
    int     wrk = steps;
    for (int stage = 0; stage < 2; stage++) {
        int     mid = stage ? (255 + p) / 2 : dm / 2;
        int     lo = stage ? p : 0;
        int     hi = stage ? 255 : dm;
        int     best = -1;
        for (;;) {
            int const top = stage ? lh_vbr_flatten(c, steps, wrk, dm, dm, mid) : lh_vbr_flatten(c, steps, wrk, dm, mid, p);
            int const nbits = lh_vbr_try(c, Q, R, g, wrk, sfm, top, mingain_l, mingain_s);
            if (nbits <= target) {
                best = mid;
                hi = mid - 1;
            }
            else
                lo = mid + 1;
            if (lo <= hi)
                mid = (lo + hi) / 2;
            else
                break;
        }
        if (best >= 0) {
            if (mid != best) {
                int const top = stage ? lh_vbr_flatten(c, steps, wrk, dm, dm, best) : lh_vbr_flatten(c, steps, wrk, dm, best, p);
                (void) lh_vbr_try(c, Q, R, g, wrk, sfm, top, mingain_l, mingain_s);
            }
            return;
        }
    }
    {
        /* searchGlobalStepsizeMax (reference vbrquantize.c:1037-1069) on the last flattened set */
        int const gain = g.global_gain;
        int     curr = gain, good = 1024, lo = gain, hi = 512;
        while (lo <= hi) {
            int     nbits, sh, top;
            curr = (lo + hi) >> 1;
            sh = wrk + (curr - gain);
            sh = sh < sfm ? sfm : sh;
            sh = sh > 255 ? 255 : sh;
            top = lh_wave_max0(inb ? sh : 0);
            lh_vbr_constrain(c, Q, R, g, sh, sfm, top, mingain_l, mingain_s);
            nbits = lh_vbr_quantize_count(c, Q, R, g);
            if (nbits == 0 || (nbits + g.part2_length) < target) {
                hi = curr - 1;
                good = curr;
            }
            else {
                lo = curr + 1;
                if (good == 1024)
                    good = curr;
            }
        }
        if (good != curr) {
            int     sh = wrk + (good - gain), top;
            sh = sh < sfm ? sfm : sh;
            sh = sh > 255 ? 255 : sh;
            top = lh_wave_max0(inb ? sh : 0);
            lh_vbr_constrain(c, Q, R, g, sh, sfm, top, mingain_l, mingain_s);
            (void) lh_vbr_quantize_count(c, Q, R, g);
        }
    }
}

/* ---- one granule of one channel (wave) --------------------------------------------------
 * pass 0: VBR_new_prepare's per-granule part + "searches scalefactors" + "encode as is";
 * pass 1: "alter our encoded data, until it fits" with the budget `target'.
 * gate = max_bits of the granule from on_pe (after the frame-level scaling). */
/* USUAL: the first pass over a normal long block of an MPEG-1 stream (block type and pass are constants: the short-block
 * geometry, reordering and constraint solver and the second pass are not compiled into that variant) */
template < int USUAL > LH_DEVFN void
lh_vbr_granule_body(int qch, int gr, int rch, int pass, int gate, int target, int substep, LhGranule * o,
                    const int8_t * g0sf)
{
    LhCtx   c = lh_ctx_load();
    LhLds & L = lh_lds;
    if (USUAL) {
        c.rate8k = 0;
        pass = 0;
    }
    LhChanLds & Q = L.u.quant.ch[qch];
    LhVbrSave & sv = L.vbr[gr][qch];
    float  *xr = L.xr[qch][gr];
    int const s = c.lane;
    int const sc = s < LH_SFBMAX ? s : LH_SFBMAX;
    LhQR    R;
    LhGrR   g;
    int     live, nonzero, sfw = 0, sfm = 0, mingain_l = 0, mingain_s[3] = { 0, 0, 0 };
    gr = lh_uni_i(gr);
    pass = lh_uni_i(pass);
    gate = lh_uni_i(gate);
    target = lh_uni_i(target);

    LH_PT(t_q);
    LH_PT(t_a);
    lh_init_outer_loop_body(c, Q, R, g, xr, USUAL ? LH_NORM_TYPE : lh_uni_i(L.block_type[gr][qch]), lh_uni_i(substep), pass == 0);
    LH_PA(15, t_a);
    if (pass == 0) {
        {
            int const slot = (lh_uni_i(L.psy_slot) + gr) % 3;
            lh_calc_xmin_body(c, Q, R, xr, L.psy_en[slot][rch], L.psy_thm[slot][rch]);
            LH_WAVE_SYNC();
            LH_DBG_XMIN(c, gr, qch, rch, Q, R.psymax);
        }
    }
    else {
        R.mnc = lh_uni_i(sv.mnc);
        g.table_select[0] = lh_uni_i(sv.table_select[0]);
        g.table_select[1] = lh_uni_i(sv.table_select[1]);
        g.table_select[2] = lh_uni_i(sv.table_select[2]);
        g.region0_count = lh_uni_i(sv.region0_count);
        g.region1_count = lh_uni_i(sv.region1_count);
    }
    LH_PA(16, t_a);
    nonzero = lh_init_xrpow(c, Q, R, g, xr);    /* silent granules: all of ix[0] cleared */
    if (pass == 0 && s == 0) {
        sv.mnc = R.mnc;
        sv.ath_over = R.ath_over;
        sv.nonzero = nonzero;
    }
    live = nonzero && gate > 0;
    LH_PA(4, t_a);
    if (live) {
        LhVbrGeo G;
        if (pass == 0) {
            int     vbrmax;
            LH_PT(t_s);
            G = lh_vbr_geometry(c, Q, R);
            LH_PA(7, t_s);
            vbrmax = lh_vbr_band_steps(c, Q, R, G, xr, sfw, sfm, mingain_l, mingain_s);
            LH_PA(5, t_s);
            LH_PT(t_c);
            lh_vbr_constrain(c, Q, R, g, sfw, sfm, vbrmax, mingain_l, mingain_s);
            LH_PA(8, t_c);
            if (s <= LH_SFBMAX) {
                sv.sfwork[s] = (uint8_t) sfw;
                sv.sfmin[s] = (uint8_t) sfm;
            }
            if (s == 0) {
                sv.mingain_l = mingain_l;
                sv.mingain_s[0] = mingain_s[0];
                sv.mingain_s[1] = mingain_s[1];
                sv.mingain_s[2] = mingain_s[2];
                sv.global_gain = g.global_gain;
            }
            {
                LH_PT(t_qc);
                (void) lh_vbr_quantize_count(c, Q, R, g);
                LH_PA(9, t_qc);
            }
        }
        else {
            int     cut;
            LH_PC(10);
            LH_WAVE_SYNC();
            sfw = sv.sfwork[sc];
            sfm = sv.sfmin[sc];
            mingain_l = lh_uni_i(sv.mingain_l);
            mingain_s[0] = lh_uni_i(sv.mingain_s[0]);
            mingain_s[1] = lh_uni_i(sv.mingain_s[1]);
            mingain_s[2] = lh_uni_i(sv.mingain_s[2]);
            cut = lh_uni_i(sv.global_gain);
            g.global_gain = cut;
            sfw = sfw < cut ? sfw : cut;        /* cutDistribution, reference vbrquantize.c:1091-1098 */
            lh_vbr_fit(c, Q, R, g, sfw, sfm, target, mingain_l, mingain_s);
        }
    }
    else if (nonzero) {
        /* a granule with energy but no bits (cannot happen with the reference's on_pe): all zero */
        for (int i = s; i < 576; i += 64)
            Q.ix[0][i] = 0;
        LH_WAVE_SYNC();
    }
    /* reduce_bit_usage (reference vbrquantize.c:1231-1247) */
    LH_PT(t_f);
    lh_best_scalefac_store_body(c, Q, R, g, gr, LH_AS_GLOBAL(const int8_t, g0sf), lh_uni_i(L.block_type[0][qch]),
                                L.scfsi[qch]);
    LH_PA(17, t_f);
    if (c.cfg->use_best_huffman == 1)
        lh_best_huffman_divide_body(c, Q, R, g);
    LH_PA(6, t_f);
    if (pass == 0 && s == 0) {
        /* (the reference's first pass ends with these finishing steps too: the second finds the granule as THEY left it) */
        sv.table_select[0] = g.table_select[0];
        sv.table_select[1] = g.table_select[1];
        sv.table_select[2] = g.table_select[2];
        sv.region0_count = g.region0_count;
        sv.region1_count = g.region1_count;
    }
    lh_store_granule(c, Q, R, g, xr, LH_AS_GLOBAL(LhGranule, o));
    LH_DBG_XR(c, gr, qch, xr);
    if (lh_uni_i(lh_lds.ctx.bytes != nullptr)) {
        lh_rg_put(c, R, g);
        lh_emit_part_stage(qch, gr);
    }
    LH_PA(3, t_q);
    if (s == 0)
        sv.use_bits = g.part2_3_length + g.part2_length;
    LH_WAVE_SYNC();
}

LH_STAGEFN void
lh_vbr_granule(int qch, int gr, int rch, int pass, int gate, int target, int substep, LhGranule * o,
               const int8_t * g0sf)
{
    lh_vbr_granule_body < 0 > (qch, gr, rch, pass, gate, target, substep, o, g0sf);
}

LH_STAGEFN void
lh_vbr_granule_n(int qch, int gr, int rch, int gate, int substep, LhGranule * o, const int8_t * g0sf)
{
    lh_vbr_granule_body < 1 > (qch, gr, rch, 0, gate, 0, substep, o, g0sf);
}


/* full-frame bits a bitrate index offers (ResvFrameBegin's return value, reference reservoir.c:82-146) */
LH_DEVFN int
lh_vbr_full_bits(const LhConfig * cfg, int index, int ResvSize, int *mean_bits, int *resv_max)
{
    int const frameLength = lh_frame_bits(cfg, index, 0);
    int const meanBits = (frameLength - cfg->sideinfo_len * 8) / LH_NGR;
    int const resvLimit = (8 * 256) * LH_NGR - 8;
    int     ResvMax = cfg->buffer_constraint - frameLength, full;
    if (ResvMax > resvLimit)
        ResvMax = resvLimit;
    if (ResvMax < 0 || cfg->disable_reservoir)
        ResvMax = 0;
    full = meanBits * LH_NGR + (ResvSize < ResvMax ? ResvSize : ResvMax);
    if (full > cfg->buffer_constraint)
        full = cfg->buffer_constraint;
    *mean_bits = meanBits;
    *resv_max = ResvMax;
    return full;
}

/* the bit budgets VBR_encode_frame fixes when the first pass used too much
 * (reference vbrquantize.c:1366-1527); wave-uniform scalar code */
LH_DEVFN void
lh_vbr_share(int share[2], const int use[2], int slack)
{
    if (share[0] > use[0] + slack) {
        share[1] += share[0];
        share[1] -= use[0] + slack;
        share[0] = use[0] + slack;
    }
    if (share[1] > use[1] + slack) {
        share[0] += share[1];
        share[0] -= use[1] + slack;
        share[1] = use[1] + slack;
    }
}

LH_DEVFN void
lh_vbr_budgets(int ngr, int nch, const int max_bits[2][2], const int use_ch[2][2], const int use_gr[2], int max_fr,
               int max_ch[2][2])
{
    int     max_gr[2], ok = 1, sum_fr = 0;
    max_ch[0][1] = max_ch[1][1] = 0;
    for (int gr = 0; gr < ngr; ++gr) {
        max_gr[gr] = 0;
        for (int ch = 0; ch < nch; ++ch) {
            max_ch[gr][ch] = (use_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL) ? LH_MAX_BITS_PER_CHANNEL : use_ch[gr][ch];
            max_gr[gr] += max_ch[gr][ch];
        }
        if (max_gr[gr] > LH_MAX_BITS_PER_GRANULE) {
            float   f[2] = { 0.0f, 0.0f }, sm = 0.0f;
            for (int ch = 0; ch < nch; ++ch) {
                if (max_ch[gr][ch] > 0) {
                    f[ch] = (float) sqrt(sqrt((double) max_ch[gr][ch]));
                    sm += f[ch];
                }
            }
            for (int ch = 0; ch < nch; ++ch)
                max_ch[gr][ch] = (sm > 0) ? (int) (LH_MAX_BITS_PER_GRANULE * f[ch] / sm) : 0;
            if (nch > 1) {
                lh_vbr_share(max_ch[gr], use_ch[gr], 32);
                for (int ch = 0; ch < nch; ++ch)
                    if (max_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                        max_ch[gr][ch] = LH_MAX_BITS_PER_CHANNEL;
            }
            max_gr[gr] = 0;
            for (int ch = 0; ch < nch; ++ch)
                max_gr[gr] += max_ch[gr][ch];
        }
        sum_fr += max_gr[gr];
    }
    if (sum_fr > max_fr) {
        {
            float   f[2] = { 0.0f, 0.0f }, sm = 0.0f;
            for (int gr = 0; gr < ngr; ++gr) {
                if (max_gr[gr] > 0) {
                    f[gr] = (float) sqrt((double) max_gr[gr]);
                    sm += f[gr];
                }
            }
            for (int gr = 0; gr < ngr; ++gr)
                max_gr[gr] = (sm > 0) ? (int) (max_fr * f[gr] / sm) : 0;
        }
        if (ngr > 1) {          /* (reference vbrquantize.c:1452-1468: only two granules have a share to pass on) */
            lh_vbr_share(max_gr, use_gr, 125);
            for (int gr = 0; gr < ngr; ++gr)
                if (max_gr[gr] > LH_MAX_BITS_PER_GRANULE)
                    max_gr[gr] = LH_MAX_BITS_PER_GRANULE;
        }
        for (int gr = 0; gr < ngr; ++gr) {
            float   f[2] = { 0.0f, 0.0f }, sm = 0.0f;
            for (int ch = 0; ch < nch; ++ch) {
                if (max_ch[gr][ch] > 0) {
                    f[ch] = (float) sqrt((double) max_ch[gr][ch]);
                    sm += f[ch];
                }
            }
            for (int ch = 0; ch < nch; ++ch)
                max_ch[gr][ch] = (sm > 0) ? (int) (max_gr[gr] * f[ch] / sm) : 0;
            if (nch > 1) {
                lh_vbr_share(max_ch[gr], use_ch[gr], 32);
                for (int ch = 0; ch < nch; ++ch)
                    if (max_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                        max_ch[gr][ch] = LH_MAX_BITS_PER_CHANNEL;
            }
        }
    }
    sum_fr = 0;
    for (int gr = 0; gr < ngr; ++gr) {
        int     sum_gr = 0;
        for (int ch = 0; ch < nch; ++ch) {
            sum_gr += max_ch[gr][ch];
            if (max_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                ok = 0;
        }
        sum_fr += sum_gr;
        if (sum_gr > LH_MAX_BITS_PER_GRANULE)
            ok = 0;
    }
    if (sum_fr > max_fr)
        ok = 0;
    if (!ok)
        for (int gr = 0; gr < ngr; ++gr)
            for (int ch = 0; ch < nch; ++ch)
                max_ch[gr][ch] = max_bits[gr][ch];
}

/* VBR_new_iteration_loop (reference quantize.c:1650-1751) for the whole workgroup: budgets,
 * the four granules (wave = channel), the second pass when the frame does not fit, and the
 * choice of the frame's bitrate.  Leaves ResvSize untouched apart from the bits used (the
 * caller finishes the reservoir bookkeeping with the chosen bitrate_index). */
/* out of line: the CBR / ABR frame code keeps its registers (the VBR frame logic inlined there cost
 * the CBR loop spills).  In: pe_use through L.pe_use; out through L.frame_bits (bitrate index),
 * L.max_bits (bits used), L.mean_bits (ResvSize after the frame's bits), L.targ_bits[0] (substep). */
LH_STAGEFN void
lh_vbr_frame(LhFrameOut * fo_in, int mode_ext, int msoff)
{
    LhCtx const c = lh_ctx_load();
    LhFrameOut *fo = LH_AS_GLOBAL(LhFrameOut, fo_in);
    LhLds & L = lh_lds;
    const LhConfig *cfg = c.cfg;
    int const w = c.wave, tid = c.tid;
    float   pe_use[2][2] = { {lh_uni_f(L.pe_use[0][0]), lh_uni_f(L.pe_use[0][1])},
    {lh_uni_f(L.pe_use[1][0]), lh_uni_f(L.pe_use[1][1])}
    };
    int     ResvSize = lh_uni_i(lh_lds.ss.ResvSize), substep = lh_uni_i(lh_lds.ss.substep_shaping);
    int     bitrate_index, total_bits;
    int const maxi = cfg->vbr_max_bitrate_index;
    mode_ext = lh_uni_i(mode_ext);
    msoff = lh_uni_i(msoff);
    int const nch = cfg->channels;
    constexpr int ngr = LH_NGR;
    int     avg, resv_top, top_bits, dummy;
    int     max_bits[2][2], use_ch[2][2], use_gr[2], use_fr, max_fr = 0, bits = 0;
    int     analog_silence, pad, used, ok;

    top_bits = lh_vbr_full_bits(cfg, maxi, ResvSize, &avg, &resv_top);
    pad = resv_top;
    for (int gr = 0; gr < ngr; gr++) {
        (void) lh_on_pe(cfg, ResvSize, resv_top, &substep, pe_use[gr], max_bits[gr], avg, 0);
        bits += max_bits[gr][0] + max_bits[gr][1];
    }
    for (int gr = 0; gr < ngr; gr++)
        for (int ch = 0; ch < 2; ch++)
            if (bits > top_bits && bits > 0) {
                max_bits[gr][ch] *= top_bits;
                max_bits[gr][ch] /= bits;
            }
    LH_SYNC_WG();
    if (mode_ext == LH_MPG_MD_MS_LR) {
        float const k = (float) (LH_SQRT2 * 0.5);
        for (int i = tid; i < 2 * 576; i += LH_NT) {
            int const gr = i >= 576, j = i - 576 * gr;
            float const l = L.xr[0][gr][j];
            float const r = L.xr[1][gr][j];
            L.xr[0][gr][j] = (l + r) * k;
            L.xr[1][gr][j] = (l - r) * k;
        }
    }
    LH_SYNC_WG();
    for (int gr = 0; gr < ngr; gr++) {
        if (w < nch)
            if (!LH_IS_LSF && lh_uni_i(L.block_type[gr][w]) == LH_NORM_TYPE)
                lh_vbr_granule_n(w, gr, msoff + w, max_bits[gr][w], substep, &fo->gr[gr][w], fo->gr[0][w].scalefac);
            else
                lh_vbr_granule(w, gr, msoff + w, 0, max_bits[gr][w], 0, substep, &fo->gr[gr][w], fo->gr[0][w].scalefac);
        else {
            /* mono: no second channel, its payload slot is all zero */
            uint32_t *z = (uint32_t *) &fo->gr[gr][w];
            for (int i = c.lane; i < (int) (sizeof(LhGranule) / 4); i += 64)
                z[i] = 0u;
        }
    }
    LH_SYNC_WG();
    analog_silence = 1;
    use_fr = 0;
    for (int gr = 0; gr < ngr; gr++) {
        use_gr[gr] = 0;
        use_ch[gr][1] = 0;
        for (int ch = 0; ch < nch; ch++) {
            LhVbrSave const &sv = L.vbr[gr][ch];
            if (lh_uni_i(sv.ath_over))
                analog_silence = 0;
            if (!lh_uni_i(sv.nonzero))
                max_bits[gr][ch] = 0;   /* silent granule needs no bits */
            use_ch[gr][ch] = lh_uni_i(sv.use_bits);
            use_gr[gr] += use_ch[gr][ch];
            max_fr += max_bits[gr][ch];
        }
        use_fr += use_gr[gr];
    }
    if (analog_silence)
        pad = 0;
    ok = (use_fr <= max_fr);
    for (int gr = 0; gr < ngr; gr++) {
        if (use_gr[gr] > LH_MAX_BITS_PER_GRANULE)
            ok = 0;
        for (int ch = 0; ch < nch; ch++)
            if (use_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                ok = 0;
    }
    used = use_fr;
    if (!ok) {
        int     max_ch[2][2];
        lh_vbr_budgets(ngr, nch, max_bits, use_ch, use_gr, max_fr, max_ch);
        LH_SYNC_WG();
        for (int gr = 0; gr < ngr; gr++)
            if (w < nch)
                lh_vbr_granule(w, gr, msoff + w, 1, max_bits[gr][w], max_ch[gr][w], substep, &fo->gr[gr][w],
                               fo->gr[0][w].scalefac);
        LH_SYNC_WG();
        used = 0;
        for (int gr = 0; gr < ngr; gr++)
            for (int ch = 0; ch < nch; ch++)
                used += lh_uni_i(L.vbr[gr][ch].use_bits);
    }
    /* smallest frame that holds the bits; a larger one while the reservoir could not take the rest */
    {
        int     i = (analog_silence && !cfg->enforce_min_bitrate) ? 1 : cfg->vbr_min_bitrate_index, j;
        for (; i < maxi; i++)
            if (used <= lh_vbr_full_bits(cfg, i, ResvSize, &dummy, &dummy))
                break;
        if (i > maxi)
            i = maxi;
        if (pad > 0) {
            for (j = maxi; j > i; --j)
                if (lh_vbr_full_bits(cfg, j, ResvSize, &dummy, &dummy) - used <= pad)
                    break;
            i = j;
        }
        bitrate_index = i;
    }
    ResvSize -= used;
    total_bits = used;
    LH_SYNC_WG();
    if (tid == 0) {
        L.frame_bits = bitrate_index;
        L.max_bits = total_bits;
        L.mean_bits = ResvSize;
        L.targ_bits[0] = substep;
    }
    LH_SYNC_WG();
}

/* ---- ABR (reference quantize.c:1768-1884, calc_target_bits): the bit budget of every
 * granule/channel from the mean bitrate and the perceptual entropy; wave-uniform scalar code.
 * The granule work itself is the CBR loop's (lh_encode_frame). */
LH_DEVFN void
lh_abr_target_bits(const LhConfig * cfg, int ResvSize, int substep, float pe[2][2], const float ms_ener_ratio[2],
                   const int block_type[2][2], int mode_ext, int targ_bits[2][2], int *analog_silence_bits)
{
    int const nch = cfg->channels, parts = LH_NGR * nch;
    int const side_bits = cfg->sideinfo_len * 8;
    int     unused_mean, unused_max, frame_cap, per_part, granted = 0;
    float   base;
    targ_bits[0][1] = targ_bits[1][1] = 0;
    /* the most a frame may hold (top bitrate + reservoir), and what silence gets (lowest bitrate) */
    frame_cap = lh_vbr_full_bits(cfg, cfg->vbr_max_bitrate_index, ResvSize, &unused_mean, &unused_max);
    *analog_silence_bits = (lh_frame_bits(cfg, 1, 0) - side_bits) / parts;
    {
        /* a granule-channel's share of the mean bitrate (9 % more while substep bit 0 is set) */
        int     frame = cfg->vbr_avg_bitrate_kbps * (576 * LH_NGR) * 1000;
        if (substep & 1)
            frame *= 1.09;
        frame /= cfg->samplerate;
        per_part = (frame - side_bits) / parts;
    }
    {
        /* 93 .. 100 % of it as the base, by compression ratio (the rest feeds the reservoir) */
        float   f = .93 + .07 * (11.0 - cfg->compression_ratio) / (11.0 - 5.5);
        f = f < .90 ? .90 : f;
        f = f > 1.00 ? 1.00 : f;
        base = f;
    }
    for (int gr = 0; gr < LH_NGR; gr++) {
        int     granule = 0;
        for (int ch = 0; ch < nch; ch++) {
            int     t = base * per_part;
            if (pe[gr][ch] > 700) {
                /* demanding granules get (pe - 700) / 1.4 more: at least half a share for short
                 * blocks, at most one and a half */
                int     more = (pe[gr][ch] - 700) / 1.4;
                if (block_type[gr][ch] == LH_SHORT_TYPE && more < per_part / 2)
                    more = per_part / 2;
                if (more > per_part * 3 / 2)
                    more = per_part * 3 / 2;
                else if (more < 0)
                    more = 0;
                t += more;
            }
            t = t > LH_MAX_BITS_PER_CHANNEL ? LH_MAX_BITS_PER_CHANNEL : t;
            targ_bits[gr][ch] = t;
            granule += t;
        }
        if (granule > LH_MAX_BITS_PER_GRANULE)
            for (int ch = 0; ch < nch; ++ch)
                targ_bits[gr][ch] = targ_bits[gr][ch] * LH_MAX_BITS_PER_GRANULE / granule;
    }
    if (mode_ext == LH_MPG_MD_MS_LR)
        for (int gr = 0; gr < LH_NGR; gr++)
            lh_reduce_side(targ_bits[gr], ms_ener_ratio[gr], per_part * nch, LH_MAX_BITS_PER_GRANULE);
    for (int gr = 0; gr < LH_NGR; gr++)
        for (int ch = 0; ch < nch; ch++) {
            if (targ_bits[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                targ_bits[gr][ch] = LH_MAX_BITS_PER_CHANNEL;
            granted += targ_bits[gr][ch];
        }
    if (granted > frame_cap && granted > 0)
        for (int gr = 0; gr < LH_NGR; gr++)
            for (int ch = 0; ch < nch; ch++)
                targ_bits[gr][ch] = targ_bits[gr][ch] * frame_cap / granted;
}

#endif
