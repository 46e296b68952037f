/*
 * lh_dev_psy_core.h -- psycho-acoustic model, wave-parallel: the pieces that do not touch the workgroup's LDS image
 * by name (transforms, spectra, partition tables), shared by the fused encode kernel (lh_dev_psy.h) and the analysis
 * kernels (lh_analysis.hip).  Header comment of the model as a whole:
 *
 * psycho-acoustic model, wave-parallel (one workgroup = one
 * stream, wave w = channel w; the mid/side pseudo-channels 2/3 are handled by
 * wave 0/1 after their own channel).
 *
 * What the reference computes with serial loops per granule
 * (psymodel.c:655-1597, fft.c:63-289) is mapped as follows:
 *   - high-pass FIR + sub-block peaks: lanes over the 576 samples, peaks by
 *     wave max-reduction (exact: max is order independent)
 *   - windowed FHT: the 128 (long) / 96 (3 short) independent butterfly units of
 *     every radix-4 pass spread over the lanes, spectra in LDS
 *   - partition energies, tonality index, spreading convolution, thresholds:
 *     one lane per partition band (<= 64), inner sums stay serial in the
 *     reference's order because float addition is not associative
 *   - partition -> scalefactor-band conversion, PE: one lane per independent
 *     serial chain
 * All float expressions keep the reference's evaluation order and types; this
 * translation unit is compiled with -ffp-contract=off.
 */
#ifndef LH_DEV_PSY_CORE_H
#define LH_DEV_PSY_CORE_H

#include "lh_dev_common.h"

#define LH_NSFIRLEN 21
#define LH_RPELEV  2
#define LH_RPELEV2 16
#define LH_PREECHO_ATT0 0.8
#define LH_PREECHO_ATT1 0.6
#define LH_PREECHO_ATT2 0.3
#define LH_VO_SCALE (1./( 14752*14752 )/(LH_BLKSIZE/2))

LH_DEVCONST float lh_psy_tab[9] = {
    1.0f, 0.79433f, 0.63096f, 0.63096f, 0.63096f, 0.63096f, 0.63096f, 0.25119f, 0.11749f
};
LH_DEVCONST int lh_mask_add_delta[9] = { 2, 2, 2, 1, 1, 1, 0, 0, -1 };
/* lh_psy_tab[i] / lh_mask_add_delta[i] for a per-lane index 0..8 without a memory access (the tables'
 * values as immediates: both sit on dependency chains of the masking spread) */
LH_DEVFN float
lh_psy_tab_at(int i)
{
    return (i == 0) ? 1.0f : (i == 1) ? 0.79433f : (i < 7) ? 0.63096f : (i == 7) ? 0.25119f : 0.11749f;
}

LH_DEVFN int
lh_mask_add_delta_at(int i)
{
    return (i < 3) ? 2 : (i < 6) ? 1 : (i < 8) ? 0 : -1;
}

LH_DEVCONST float lh_hp_fir[10] = {
    (float) (-8.65163e-18 * 2), (float) (-0.00851586 * 2), (float) (-6.74764e-18 * 2),
    (float) (0.0209036 * 2), (float) (-3.36639e-17 * 2), (float) (-0.0438162 * 2),
    (float) (-1.54175e-17 * 2), (float) (0.0931738 * 2), (float) (-5.52212e-17 * 2),
    (float) (-0.313819 * 2)
};
LH_DEVCONST float lh_regcoef_s[12] = {
    11.8f, 13.6f, 17.2f, 32.f, 46.5f, 51.3f, 57.5f, 67.1f, 71.5f, 84.6f, 97.6f, 130.f
};
LH_DEVCONST float lh_regcoef_l[21] = {
    6.8f, 5.8f, 5.8f, 6.4f, 6.5f, 9.9f, 12.1f, 14.4f, 15.f, 18.9f, 21.6f, 26.9f, 34.2f, 40.2f,
    46.8f, 56.5f, 60.7f, 73.9f, 85.7f, 93.4f, 126.1f
};

/* lh_mask_add() near the diagonal (|kk - b| <= delta; reference psymodel.c:294-341 up to the end of the
 * `b <= delta' block): (larger + smaller) x table2[i], i the cell of the ratio larger / smaller, or the plain
 * sum from ma_max_i1 on.  Neither the quotient nor its logarithm is formed: cell i applies exactly when
 * larger lies above i of the nine exact products mid[j] x smaller (LhTables.mask_mid; the ninth is the
 * boundary of ma_max_i1 and stands for a factor of 1).  A masker of 0 -- the reference's early exits --
 * puts the other above all nine: the sum, which is the other masker.  Both maskers are sums and products
 * of non-negative, finite terms, so the reference's clamps of negative inputs have nothing to do. */
#define LH_T2(a_) ((float) ((a_) * (a_)))
LH_DEVFN float
lh_mask_add_near(const double (&mid)[10], float m1, float m2)
{
    float const hi = __builtin_fmaxf(m1, m2), lo = __builtin_fminf(m1, m2);
    double const h = (double) hi, l = (double) lo;
    float   f = LH_T2(1.33352);
    f = (h > mid[0] * l) ? LH_T2(1.35879) : f;
    f = (h > mid[1] * l) ? LH_T2(1.38454) : f;
    f = (h > mid[2] * l) ? LH_T2(1.39497) : f;
    f = (h > mid[3] * l) ? LH_T2(1.40548) : f;
    f = (h > mid[4] * l) ? LH_T2(1.3537) : f;
    f = (h > mid[5] * l) ? LH_T2(1.30382) : f;
    f = (h > mid[6] * l) ? LH_T2(1.22321) : f;
    f = (h > mid[7] * l) ? LH_T2(1.14758) : f;
    f = (h > mid[8] * l) ? 1.0f : f;
    return (m1 + m2) * f;
}

/* reference psymodel.c:443-454 */
LH_DEVFN float
lh_ns_interp(float x, float y, float r)
{
    if (r >= 1.0f)
        return x;
    if (r <= 0.0f)
        return y;
    if (y > 0.0f)
        return lh_powf(x / y, r) * y;
    return 0.0f;
}

/* one radix-4 FHT pass over `n' points in LDS: the n/8 butterfly units of the
 * pass (reference fft.c:70-146, one unit = one trip of an inner do-while) are
 * dealt to the lanes.  `unit0'/`nunits' let the three short transforms share
 * one call. */
/* plane rotation of (x, y) by the angle whose cosine / sine are c / s: the component along the new
 * axis and the one across it */
struct LhRot {
    float   along, across;
};

LH_DEVFN LhRot
lh_rot(float c, float s, float x, float y)
{
    LhRot   r;
    r.along = c * x + s * y;
    r.across = s * x - c * y;
    return r;
}

/* One radix-4 butterfly unit.  A unit on its block's axes (i == 0) and one off them read and write the same eight places --
 * lo + {0, k1, k2, k3} and hi + {0, k1, k2, k3} with hi = lo + kx on the axes and the mirror position k1 - i off them -- and
 * differ in the arithmetic in between; a wave has both kinds in every pass.  Both are computed from ONE set of loads and the
 * lane keeps its own (a select per result): as a branch the two kinds ran one after the other, each with its own loads, its
 * own wait for them and its own stores. */
LH_DEVFN void
lh_fht_unit(lh_f32x4 tw, float *fz, int k1, int u)
{
    int const kx = k1 >> 1;
    int const k2 = k1 << 1, k3 = k2 + k1, k4 = k2 << 1;
    int const blk = u / kx, i = u - blk * kx;
    int const axis = (i == 0);
    float  *lo = fz + blk * k4 + i;
    float  *hi = fz + blk * k4 + (axis ? kx : k1 - i);
    float const p0 = lo[0], p1 = lo[k1], p2 = lo[k2], p3 = lo[k3];
    float const q0 = hi[0], q1 = hi[k1], q2 = hi[k2], q3 = hi[k3];
    /* on the axes: no rotation, the mirrored quarter only scales by sqrt 2 */
    float const s01 = p0 + p1, d01 = p0 - p1, s23 = p2 + p3, d23 = p2 - p3;
    float const r2 = (float) (LH_SQRT2 * q2), r3 = (float) (LH_SQRT2 * q3);
    float const t01 = q0 + q1, u01 = q0 - q1;
    float const a_l0 = s01 + s23, a_l1 = d01 + d23, a_l2 = s01 - s23, a_l3 = d01 - d23;
    float const a_h0 = t01 + r2, a_h1 = u01 + r3, a_h2 = t01 - r2, a_h3 = u01 - r3;
    /* off the axes: the second and fourth quarters turn by the double angle (tw.z, tw.w), then the two half-sums turn by
     * the single angle (tw.x, tw.y) */
    LhRot const rq1 = lh_rot(tw.z, tw.w, p1, q1);
    LhRot const rq3 = lh_rot(tw.z, tw.w, p3, q3);
    float const le = p0 + rq1.along, lm = p0 - rq1.along;
    float const he = q0 + rq1.across, hm = q0 - rq1.across;
    float const l2e = p2 + rq3.along, l2m = p2 - rq3.along;
    float const h2e = q2 + rq3.across, h2m = q2 - rq3.across;
    LhRot const ra = lh_rot(tw.x, tw.y, l2e, h2m);
    LhRot const rb = lh_rot(tw.y, tw.x, h2e, l2m);
    lo[0] = axis ? a_l0 : le + ra.along;
    lo[k1] = axis ? a_l1 : lm + rb.across;
    lo[k2] = axis ? a_l2 : le - ra.along;
    lo[k3] = axis ? a_l3 : lm - rb.across;
    hi[0] = axis ? a_h0 : he + rb.along;
    hi[k1] = axis ? a_h1 : hm + ra.across;
    hi[k2] = axis ? a_h2 : he - rb.along;
    hi[k3] = axis ? a_h3 : hm - ra.across;
}

LH_DEVFN unsigned
lh_rev8(unsigned v)
{
    v = ((v & 0xf0u) >> 4) | ((v & 0x0fu) << 4);
    v = ((v & 0xccu) >> 2) | ((v & 0x33u) << 2);
    v = ((v & 0xaau) >> 1) | ((v & 0x55u) << 1);
    return v;
}

/* windowed 1024-point FHT of channel ch starting at frame-buffer index `base'
 * (reference fft.c:245-289); result in x[1024] (LDS), one wave */
LH_DEVFN void
lh_fft_long(const LhCtx & c, int ch, int base, float *x)
{
    const float *w = c.T->fft_window;
    int     lane = c.lane;
    for (int jj = lane; jj < LH_BLKSIZE / 8; jj += 64) {
        float   f0, f1, f2, f3, ww;
        float  *o = x + 4 * jj;
        int     i = (int) lh_rev8((unsigned) jj);
        f0 = w[i] * lh_smp(c, ch, base + i);
        ww = w[i + 0x200] * lh_smp(c, ch, base + i + 0x200);
        f1 = f0 - ww;
        f0 = f0 + ww;
        f2 = w[i + 0x100] * lh_smp(c, ch, base + i + 0x100);
        ww = w[i + 0x300] * lh_smp(c, ch, base + i + 0x300);
        f3 = f2 - ww;
        f2 = f2 + ww;
        o[0] = f0 + f2;
        o[2] = f0 - f2;
        o[1] = f1 + f3;
        o[3] = f1 - f3;
        f0 = w[i + 0x001] * lh_smp(c, ch, base + i + 0x001);
        ww = w[i + 0x201] * lh_smp(c, ch, base + i + 0x201);
        f1 = f0 - ww;
        f0 = f0 + ww;
        f2 = w[i + 0x101] * lh_smp(c, ch, base + i + 0x101);
        ww = w[i + 0x301] * lh_smp(c, ch, base + i + 0x301);
        f3 = f2 - ww;
        f2 = f2 + ww;
        o[LH_BLKSIZE / 2 + 0] = f0 + f2;
        o[LH_BLKSIZE / 2 + 2] = f0 - f2;
        o[LH_BLKSIZE / 2 + 1] = f1 + f3;
        o[LH_BLKSIZE / 2 + 3] = f1 - f3;
    }
    {
        /* the twiddle factors of all four passes (HBM, 16 bytes per butterfly unit) are requested
         * before the first pass: they arrive while the passes before theirs run */
        lh_f32x4 tw[4][2];
#pragma unroll
        for (int stage = 0; stage < 4; stage++) {
            int const kx = 2 << (2 * stage);
#pragma unroll
            for (int q = 0; q < 2; q++)
                tw[stage][q] = *(const lh_f32x4 *) c.T->fht_tw[stage][(lane + 64 * q) % kx];
        }
        LH_WAVE_SYNC_MEM();
#pragma unroll
        for (int stage = 0; stage < 4; stage++) {
            int const k1 = 4 << (2 * stage);
#pragma unroll
            for (int q = 0; q < 2; q++)
                lh_fht_unit(tw[stage][q], x, k1, lane + 64 * q);
            LH_WAVE_SYNC_MEM();
        }
    }
}

/* three windowed 256-point FHTs (reference fft.c:193-243); x[3][256] in LDS, one wave */
LH_DEVFN void
lh_fft_short(const LhCtx & c, int ch, int base, float *x)
{
    const float *ws = c.T->fft_window_s;
    int     lane = c.lane;
    for (int t = lane; t < 3 * (LH_BLKSIZE_S / 8); t += 64) {
        int const b = t >> 5, j = t & 31;
        int const k = (576 / 3) * (b + 1) + base;
        float   f0, f1, f2, f3, w;
        float  *o = x + b * LH_BLKSIZE_S + 4 * j;
        int     i = (int) lh_rev8((unsigned) (j << 2));
        f0 = ws[i] * lh_smp(c, ch, i + k);
        w = ws[0x7f - i] * lh_smp(c, ch, i + k + 0x80);
        f1 = f0 - w;
        f0 = f0 + w;
        f2 = ws[i + 0x40] * lh_smp(c, ch, i + k + 0x40);
        w = ws[0x3f - i] * lh_smp(c, ch, i + k + 0xc0);
        f3 = f2 - w;
        f2 = f2 + w;
        o[0] = f0 + f2;
        o[2] = f0 - f2;
        o[1] = f1 + f3;
        o[3] = f1 - f3;
        f0 = ws[i + 0x01] * lh_smp(c, ch, i + k + 0x01);
        w = ws[0x7e - i] * lh_smp(c, ch, i + k + 0x81);
        f1 = f0 - w;
        f0 = f0 + w;
        f2 = ws[i + 0x41] * lh_smp(c, ch, i + k + 0x41);
        w = ws[0x3e - i] * lh_smp(c, ch, i + k + 0xc1);
        f3 = f2 - w;
        f2 = f2 + w;
        o[LH_BLKSIZE_S / 2 + 0] = f0 + f2;
        o[LH_BLKSIZE_S / 2 + 2] = f0 - f2;
        o[LH_BLKSIZE_S / 2 + 1] = f1 + f3;
        o[LH_BLKSIZE_S / 2 + 3] = f1 - f3;
    }
    LH_WAVE_SYNC_MEM();
    for (int stage = 0, k1 = 4; stage < 3; stage++, k1 <<= 2) {
        for (int t = lane; t < 3 * (LH_BLKSIZE_S / 8); t += 64) {
            int const b = t >> 5, u = t & 31;
            lh_f32x4 const tw = *(const lh_f32x4 *) c.T->fht_tw[stage][u % (k1 >> 1)];
            lh_fht_unit(tw, x + b * LH_BLKSIZE_S, k1, u);
        }
        LH_WAVE_SYNC_MEM();
    }
}

/* power spectrum of chn from the two channel spectra (reference psymodel.c:664-688,
 * 713-736); n = transform length, out has n/2+1 entries; one wave */
LH_DEVFN void
lh_fft_energy(const LhCtx & c, int chn, const float *wl, const float *wr, int n, float *out)
{
    float const sqrt2_half = (float) (LH_SQRT2 * 0.5f);
    int const h = n >> 1;
    for (int m = c.lane; m <= h; m += 64) {
        int const ire = m, iim = (m == 0) ? 0 : (n - m);
        float   re, im;
        if (chn == 0) {
            re = wl[ire];
            im = wl[iim];
        }
        else if (chn == 1) {
            re = wr[ire];
            im = wr[iim];
        }
        else if (chn == 2) {
            re = (wl[ire] + wr[ire]) * sqrt2_half;
            im = (wl[iim] + wr[iim]) * sqrt2_half;
        }
        else {
            re = (wl[ire] - wr[ire]) * sqrt2_half;
            im = (wl[iim] - wr[iim]) * sqrt2_half;
        }
        if (m == 0)
            out[0] = re * re;
        else
            out[m] = (re * re + im * im) * 0.5f;
    }
}

/* the same for the two pseudo-channels wave w owns under joint stereo -- its own channel (L or R) and
 * mid (w = 0) or side (w = 1) -- from one reading of the two spectra */
LH_DEVFN void
lh_fft_energy_pair(const LhCtx & c, int w, const float *wl, const float *wr, int n, float *out_own, float *out_ms)
{
    float const sqrt2_half = (float) (LH_SQRT2 * 0.5f);
    int const h = n >> 1;
    for (int m = c.lane; m <= h; m += 64) {
        int const ire = m, iim = (m == 0) ? 0 : (n - m);
        float const lre = wl[ire], lim = wl[iim], rre = wr[ire], rim = wr[iim];
        float const ore = w ? rre : lre, oim = w ? rim : lim;
        float const mre = (w ? lre - rre : lre + rre) * sqrt2_half, mim = (w ? lim - rim : lim + rim) * sqrt2_half;
        out_own[m] = (m == 0) ? ore * ore : (ore * ore + oim * oim) * 0.5f;
        out_ms[m] = (m == 0) ? mre * mre : (mre * mre + mim * mim) * 0.5f;
    }
}

/* serial partition -> scalefactor band accumulation (reference psymodel.c:350-393);
 * executed by ONE lane per (channel, table) chain */
/* partitions -> scalefactor bands (reference psymodel.c:350-409), one lane per band.
 * The reference walks the bands in order and carries (b, enn, thmm) from band to band.  Where
 * the walk stands when it reaches band sb depends on the tables only: with
 * bl_k = min(bo[k], npart) the recurrence b_{k+1} = max(b_k, bl_k) + 1, b_0 = 0 has the closed
 * form b_sb = sb + max(0, max_{k<sb}(bl_k - k)); the band is zero-filled when an earlier band
 * already ran into npart (b_sb - 1 >= npart), and otherwise its sums start from the
 * w_next-weighted value of partition b_sb - 1.  The float additions inside the band keep the
 * reference's order.  All lanes of the wave must call. */
LH_DEVFN void
lh_partition2sfb_wave(LhPsyBand const *gd, float const *eb, float const *thr, float *enn_out,
                      float *thm_out, int out_stride, float thm_scale, int replicate3, int lane,
                      int active)
{
    int const npart = gd->npart;
    int const n_sb = gd->n_sb;
    int const sb = lane;
    float   enn = 0.0f, thmm = 0.0f, tv;
    int     b, live, m;
    {
        /* max_{k < sb} (bl_k - k), at least 0: a running maximum over the lanes, shifted by one */
        int const bo_k = gd->bo[lane < n_sb ? lane : 0];
        int const d = (bo_k < npart ? bo_k : npart) - lane;
        uint32_t const run = lh_wave_scan_max_u32((lane < n_sb && d > 0) ? (uint32_t) d : 0u);
        uint32_t const prev = lh_shfl_u32(run, (lane - 1) & 63);
        m = lane > 0 ? (int) prev : 0;
    }
    LH_WAVE_SYNC_MEM();
    b = sb + m;
    live = (sb == 0) || (b - 1 < npart);
    if (!active || sb >= n_sb)
        return;
    if (live) {
        int const bo_sb = gd->bo[sb];
        int const b_lim = bo_sb < npart ? bo_sb : npart;
        if (sb > 0) {
            float const carry_w = 1.0f - gd->bo_weight[sb - 1];
            enn = carry_w * eb[b - 1];
            thmm = carry_w * thr[b - 1];
        }
        while (b < b_lim) {
            enn += eb[b];
            thmm += thr[b];
            b++;
        }
        if (b < npart) {
            float const w_curr = gd->bo_weight[sb];
            enn += w_curr * eb[b];
            thmm += w_curr * thr[b];
        }
        tv = thm_scale < 0 ? thmm : thmm * thm_scale;
    }
    else {
        enn = 0;
        tv = thm_scale < 0 ? 0.0f : 0.0f * thm_scale;
    }
    enn_out[sb * out_stride] = enn;
    thm_out[sb * out_stride] = tv;
    if (replicate3) {
        enn_out[sb * out_stride + 1] = enn_out[sb * out_stride + 2] = enn;
        thm_out[sb * out_stride + 1] = thm_out[sb * out_stride + 2] = tv;
    }
}

/* The same for a long-block granule, all at once: lanes 0..21 take the 22 long bands (table psy_l), lanes
 * 32..44 the 13 short-band estimates made from the long partitions (psy_l_to_s: threshold / 64, written to
 * all three windows), and the wave's NC pseudo-channels go side by side -- one pass instead of four.
 * en / thm: the channel's 61-entry band image (22 long values, then 13 x 3 short ones). */
struct LhSfbChan {
    const float *eb, *thr;      /* partition energies / thresholds (LDS, 64 each) */
    float  *en, *thm;
};

template < int NC > LH_DEVFN void
lh_partition2sfb_long(const LhTables * T, const LhSfbChan (&ch)[NC], int lane)
{
    int const est = lane >= 32;                 /* the lane works on the long->short estimate */
    int const sb = est ? lane - 32 : lane;
    LhPsyBand const *gd = est ? &T->psy_l_to_s : &T->psy_l;
    int const npart = gd->npart;
    int const n_sb = est ? T->psy_l_to_s.n_sb : T->psy_l.n_sb;
    int const mine = sb < n_sb;
    int const sbc = mine ? sb : 0;
    int const bo_sb = gd->bo[sbc];
    float const w_prev = gd->bo_weight[sbc > 0 ? sbc - 1 : 0], w_curr = gd->bo_weight[sbc];
    int const b_lim = bo_sb < npart ? bo_sb : npart;
    int     m;
    {
        /* where the reference's band walk stands on reaching band sb (see lh_partition2sfb_wave): a running
         * maximum over the lanes of each table's group, shifted by one lane */
        int const d = b_lim - sb;
        uint32_t const v = (mine && d > 0) ? (uint32_t) d : 0u;
        uint32_t const run_l = lh_wave_scan_max_u32(est ? 0u : v);
        uint32_t const run_e = lh_wave_scan_max_u32((est && lane >= 32) ? v : 0u);  /* lanes below 32 add nothing */
        uint32_t const run = est ? run_e : run_l;
        uint32_t const prev = lh_shfl_u32(run, (lane - 1) & 63);
        m = (sb > 0) ? (int) prev : 0;
    }
    LH_WAVE_SYNC_MEM();
    {
        int const b0 = sb + m;
        int const live = (sb == 0) || (b0 - 1 < npart);
        int const steps = (mine && live && b_lim > b0) ? b_lim - b0 : 0;
        int const nmax = lh_uni_i((int) lh_wave_max_u32((uint32_t) steps));
        int const bend = b0 + steps;            /* where the lane's walk ends */
        float   enn[NC], thmm[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            int const carry = mine && live && sb > 0;
            float const cw = 1.0f - w_prev;
            float const e = ch[q].eb[carry ? b0 - 1 : 0], t = ch[q].thr[carry ? b0 - 1 : 0];
            enn[q] = carry ? cw * e : 0.0f;
            thmm[q] = carry ? cw * t : 0.0f;
        }
        for (int i = 0; i < nmax; i++) {
            int const on = i < steps;
            int const b = on ? b0 + i : 0;
#pragma unroll
            for (int q = 0; q < NC; q++) {
                float const e = ch[q].eb[b], t = ch[q].thr[b];
                float const se = enn[q] + e, st = thmm[q] + t;
                enn[q] = on ? se : enn[q];
                thmm[q] = on ? st : thmm[q];
            }
        }
        {
            int const tail = mine && live && bend < npart;
#pragma unroll
            for (int q = 0; q < NC; q++) {
                float const e = ch[q].eb[tail ? bend : 0], t = ch[q].thr[tail ? bend : 0];
                float const se = enn[q] + w_curr * e, st = thmm[q] + w_curr * t;
                float   ev, tv;
                enn[q] = tail ? se : enn[q];
                thmm[q] = tail ? st : thmm[q];
                ev = live ? enn[q] : 0.0f;
                tv = live ? thmm[q] : 0.0f;
                if (est)
                    tv = tv * (float) (1. / 64.f);
                if (mine) {
                    if (est) {
                        ch[q].en[22 + 3 * sb] = ch[q].en[22 + 3 * sb + 1] = ch[q].en[22 + 3 * sb + 2] = ev;
                        ch[q].thm[22 + 3 * sb] = ch[q].thm[22 + 3 * sb + 1] = ch[q].thm[22 + 3 * sb + 2] = tv;
                    }
                    else {
                        ch[q].en[sb] = ev;
                        ch[q].thm[sb] = tv;
                    }
                }
            }
        }
    }
}

/* tonality index of partition b (reference psymodel.c:583-652 / 958-1028) from the maxima / averages of
 * the partition itself (m1, a1) and of its neighbours below (m0, a0) and above (m2, a2) */
LH_DEVFN int
lh_mask_index(LhPsyBand const *gd, int b, float m0, float m1, float m2, float a0, float a1, float a2)
{
    float   m, a;
    int     k, nl;
    int const np = gd->npart;
    if (b == 0) {
        a = a1 + a2;
        m = m1;
        if (m < m2)
            m = m2;
        nl = gd->numlines[0] + gd->numlines[1] - 1;
        if (!(a > 0.0f))
            return 0;
        a = 20.0f * (m * 2.0f - a) / (a * nl);
    }
    else if (b == np - 1) {
        a = a0 + a1;
        m = m0;
        if (m < m1)
            m = m1;
        nl = gd->numlines[b - 1] + gd->numlines[b] - 1;
        if (!(a > 0.0f))
            return 0;
        a = 20.0f * (m * 2.0f - a) / (a * nl);
    }
    else {
        a = a0 + a1 + a2;
        m = m0;
        if (m < m1)
            m = m1;
        if (m < m2)
            m = m2;
        nl = gd->numlines[b - 1] + gd->numlines[b] + gd->numlines[b + 1] - 1;
        if (!(a > 0.0f))
            return 0;
        a = 20.0f * (m * 3.0f - a) / (a * nl);
    }
    k = (int) a;
    if (k > 8)
        k = 8;
    return k;
}

/* lh_mask_add() for a partition outside the band around the diagonal (|kk - b| > delta), where the
 * reference only asks whether the ratio of the two maskers is below c = ma_max_i2: larger + smaller, or
 * the larger alone (psymodel.c:294-341, the tail after the `b <= delta' block).  The ratio itself
 * is not needed for that.  The correctly rounded float quotient hi / lo is below c exactly when the true
 * quotient is below the midpoint of c and its predecessor (the quotient cannot sit on that midpoint: it has
 * 25 significant bits, so midpoint x lo has at least 25 and is no float), i.e. when hi < midpoint x lo --
 * and that product, 25 x 24 bits, is exact in double.  lo = 0 (the reference's early exits) falls out as
 * "the larger alone".  Both maskers are sums and products of non-negative terms here, so the reference's
 * clamps of negative inputs have nothing to do.  No division, no table and no branch on the chain, which
 * is what the wave waits for.  (bound = LhTables.mask_mid[9].) */
LH_DEVFN float
lh_mask_add_far(float m1, float m2, double bound)
{
    float const hi = __builtin_fmaxf(m1, m2), lo = __builtin_fminf(m1, m2);
    return ((double) hi < bound * (double) lo) ? m1 + m2 : hi;
}

/* Partition energies + tonality + spreading, one lane per partition, for the NC (1 or 2) pseudo-channels
 * a wave owns -- side by side: the two channels' chains are independent, so one's table look-ups and
 * dependent additions fill the other's waits, and the loop control and the partition tables are shared.
 * is_long selects the long-block variant with the pre-echo clamp against the two previous granules
 * (reference psymodel.c:1134-1262) or the short-block variant (:1031-1131). */
struct LhMaskChan {
    int     chn;
    const float *energy;        /* power spectrum (LDS) */
    float  *eb, *thr;           /* partition energies / thresholds out (LDS, 64 each) */
    float  *nb1, *nb2;          /* long blocks: this partition's spread energy of the last / last but one long call
                                 * (reference nb_l1 / nb_l2), a register of the lane for the whole launch */
    LhMidMask *raw;             /* analysis kernels (FRONT): where the values before the recurrences go (HBM), see LhMidMask */
};


/* FRONT = 1 (analysis kernels, lh_analysis.hip): everything up to the recurrences -- the partition's energy, its spread
 * energy x weight and its cap go to ch[].raw (HBM); the encode kernel's lh_masking_tail (lh_dev_psy.h) takes it from there.
 * The translation unit's LDS image must have the member `ss' (LhSmallState) either way. */
template < int NC, int FRONT = 0 > LH_DEVFN void
lh_compute_masking(const LhCtx & c, int is_long, const LhMaskChan (&ch)[NC], const float *s3)
{
    /* s3: the spreading matrix, either in HBM (LhTables) or staged in LDS by the caller */
    double  mid[10];            /* wave-uniform: scalar register pairs */
#pragma unroll
    for (int j = 0; j < 10; j++)
        mid[j] = lh_uni_f64(c.T->mask_mid[j]);
    double const far_bound = mid[9];
    LhPsyBand const *gd = is_long ? &c.T->psy_l : &c.T->psy_s;
    int const b = c.lane;
    int const np = gd->npart;
    int const on = b < np;
    /* the lane's table entries (HBM, L2-resident): all requested up front, one wait */
    int const bc = on ? b : 0;
    int const t_numlines = gd->numlines[bc], t_first = gd->s3ind[bc][0], t_last = gd->s3ind[bc][1], t_row = gd->s3_row[bc];
    float const t_rnum = gd->rnumlines[bc], t_mlow = gd->masking_lower[bc], t_minval = gd->minval[bc];
    float   ebb[NC], m[NC], avg[NC], th[NC], ecb[NC];
    int     tone[NC], delta[NC], dd[NC];
    LH_PT(t_mk);
#pragma unroll
    for (int q = 0; q < NC; q++)
        ebb[q] = m[q] = th[q] = ecb[q] = 0;
    {
        /* A partition's energy is the sum of its lines in order (up to 83 of them for the widest
         * one); the loads do not depend on the sum, so eight go out together and the additions follow. */
        int const n = on ? t_numlines : 0;
        int const j0 = (int) lh_wave_scan_u32((uint32_t) n) - n;      /* the partition's first line */
        float const rn = on ? t_rnum : 0.0f;
        int const nmax = lh_uni_i((int) lh_wave_max_u32((uint32_t) n));
#if !defined(LH_EMU)
        /* On the device the lanes are switched off as their partitions end (v_cmpx narrows EXEC before every term, a
         * lane that is off keeps sum and maximum): three instructions per term and channel, nothing selected per
         * load.  The loads run on past a partition's end (the channel's spectrum, then whatever follows it in the
         * workgroup's image); EXEC is restored after each block of eight terms. */
        float   nx[NC][8];
#pragma unroll
        for (int q = 0; q < NC; q++)
#pragma unroll
            for (int u = 0; u < 8; u++)
                nx[q][u] = ch[q].energy[j0 + u];
        for (int i = 0; i < nmax; i += 8) {
            int const rem = n - i;
            /* the next block's terms are read before this block's additions (the last trip reads a block nobody adds) */
            float   cu[NC][8];
#pragma unroll
            for (int q = 0; q < NC; q++)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    cu[q][u] = nx[q][u];
                    nx[q][u] = ch[q].energy[j0 + i + 8 + u];
                }
#pragma unroll
            for (int q = 0; q < NC; q++) {
                float const a0 = cu[q][0], a1 = cu[q][1], a2 = cu[q][2], a3 = cu[q][3], a4 = cu[q][4], a5 = cu[q][5], a6 = cu[q][6], a7 = cu[q][7];
                unsigned long long sv, tm;
#define LH_PS_TERM(K, A) "v_cmpx_gt_i32_e64 %[tm], %[rem], " #K "\n\tv_add_f32 %[eb], %[eb], %[" #A "]\n\tv_max_f32 %[mx], %[mx], %[" #A "]\n\t"
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             LH_PS_TERM(0, a0) LH_PS_TERM(1, a1) LH_PS_TERM(2, a2) LH_PS_TERM(3, a3)
                             LH_PS_TERM(4, a4) LH_PS_TERM(5, a5) LH_PS_TERM(6, a6) LH_PS_TERM(7, a7)
                             "s_mov_b64 exec, %[sv]"
                             : [eb] "+v"(ebb[q]), [mx] "+v"(m[q]), [sv] "=&s"(sv), [tm] "=&s"(tm)
                             : [rem] "v"(rem), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3),
                               [a4] "v"(a4), [a5] "v"(a5), [a6] "v"(a6), [a7] "v"(a7));
#undef LH_PS_TERM
            }
        }
#else
        for (int i = 0; i < nmax; i += 8) {
            float   el[NC][8];
#pragma unroll
            for (int q = 0; q < NC; q++)
#pragma unroll
                for (int u = 0; u < 8; u++)
                    el[q][u] = ch[q].energy[j0 + ((i + u < n) ? i + u : 0)];
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int q = 0; q < NC; q++) {
                    /* a term beyond the partition's end is replaced by +0.0f: sum and maximum (both
                     * non-negative) stay as they are */
                    float const e = (i + u < n) ? el[q][u] : 0.0f;
                    ebb[q] += e;
                    m[q] = __builtin_fmaxf(m[q], e);
                }
        }
#endif
#pragma unroll
        for (int q = 0; q < NC; q++) {
            avg[q] = ebb[q] * rn;
            ch[q].eb[b] = ebb[q];       /* 0 above the last partition */
        }
    }
    LH_PA(14, t_mk);
    {
        /* The tonality index needs the neighbours' maximum and average: lane exchanges.  What the
         * spreading reads of a partition kk is its energy and the factor psy_tab[index[kk]]: the
         * factors of the whole channel sit in thr[] until the thresholds, which each lane writes
         * over its own factor when every lane is through with the spreading, replace them. */
        int const lo = (b > 0) ? b - 1 : 0, hi = (b < 63) ? b + 1 : 63;
#pragma unroll
        for (int q = 0; q < NC; q++) {
            float const m0 = lh_shfl_f32(m[q], lo), m2 = lh_shfl_f32(m[q], hi);
            float const a0 = lh_shfl_f32(avg[q], lo), a2 = lh_shfl_f32(avg[q], hi);
            tone[q] = on ? lh_mask_index(gd, b, m0, m[q], m2, a0, avg[q], a2) : 0;
            ch[q].thr[b] = lh_psy_tab_at(tone[q]);
            delta[q] = lh_mask_add_delta_at(tone[q]);
        }
    }
    /* The reference walks kk = s3ind[b][0] .. s3ind[b][1], adding partition kk's spread energy to the
     * running sum with mask_add(), whose expensive branch (a division, fast_log2, table2) only applies
     * within delta = 2, 1, 0 or -1 partitions of b.  All lanes walk together and the diagonal comes at
     * a different step for each, so the walk is laid out relative to the diagonal: n1 steps ending at
     * kk = b - 3 (cheap rule), the five steps kk = b - 2 .. b + 2 (expensive rule where |kk - b| <=
     * delta), n3 steps from kk = b + 3 (cheap rule); a lane sits out the steps outside its own range.
     * The order of a lane's additions is the reference's. */
    int const first = on ? t_first : 1, last = on ? t_last : 0;
    int const krel = (on ? t_row : 0) - first;  /* s3 index of partition kk = krel + kk */
    int     n1, n3;
    {
        /* sum of the indices over the lane's range, from the wave's prefix sums; trip counts */
        uint32_t t[2];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            uint32_t const P = lh_wave_scan_u32((uint32_t) tone[q]);
            uint32_t const pl = lh_shfl_u32(P, last & 63), pf = lh_shfl_u32(P, (first - 1) & 63);
            dd[q] = (int) (pl - ((first > 0) ? pf : 0u));
        }
        t[0] = (uint32_t) ((on && b - 3 - first > 0) ? b - 3 - first : 0);
        t[1] = (uint32_t) ((on && last - b - 2 > 0) ? last - b - 2 : 0);
        lh_wave_max_n < 2 > (t);
        n1 = lh_uni_i((int) t[0]);
        n3 = lh_uni_i((int) t[1]);
    }
    LH_WAVE_SYNC_MEM();
    LH_PA(15, t_mk);
#define LH_SPREAD_X(q_, kk_) (s3[krel + (kk_)] * ch[q_].eb[(kk_)] * ch[q_].thr[(kk_)])
#pragma unroll
    for (int q = 0; q < NC; q++)
        ecb[q] = on ? LH_SPREAD_X(q, first) : 0.0f;
    {
        int     kk = b - 2 - n1;
        int     act = kk > first && kk <= last;
        float   x[NC];
#pragma unroll
        for (int q = 0; q < NC; q++)
            x[q] = LH_SPREAD_X(q, act ? kk : first);
        for (int j = 0; j < n1; j++) {
            int const act_n = (kk + 1 > first) && (kk + 1 <= last) && (j + 1 < n1);
            int const kn = act_n ? kk + 1 : first;
#pragma unroll
            for (int q = 0; q < NC; q++) {
                float const xn = LH_SPREAD_X(q, kn);
                float const r = lh_mask_add_far(ecb[q], x[q], far_bound);
                ecb[q] = act ? r : ecb[q];
                x[q] = xn;
            }
            act = act_n;
            ++kk;
        }
    }
    for (int j = 0; j < 5; j++) {
        int const kk = b - 2 + j;
        int const act = kk > first && kk <= last;
        int const d = (j < 2) ? 2 - j : j - 2;          /* |kk - b| */
        int const kc = act ? kk : first;
#pragma unroll
        for (int q = 0; q < NC; q++) {
            int const near = act && d <= delta[q];
            float const x = LH_SPREAD_X(q, kc);
            float   r = lh_mask_add_far(ecb[q], x, far_bound);
            if (lh_ballot(near)) {
                float const rn = lh_mask_add_near(mid, ecb[q], x);
                r = near ? rn : r;
            }
            ecb[q] = act ? r : ecb[q];
        }
    }
    {
        int     kk = b + 3;
        int     act = kk > first && kk <= last;
        float   x[NC];
#pragma unroll
        for (int q = 0; q < NC; q++)
            x[q] = LH_SPREAD_X(q, act ? kk : first);
        for (int j = 0; j < n3; j++) {
            int const act_n = (kk + 1 > first) && (kk + 1 <= last);
            int const kn = act_n ? kk + 1 : first;
#pragma unroll
            for (int q = 0; q < NC; q++) {
                float const xn = LH_SPREAD_X(q, kn);
                float const r = lh_mask_add_far(ecb[q], x[q], far_bound);
                ecb[q] = act ? r : ecb[q];
                x[q] = xn;
            }
            act = act_n;
            ++kk;
        }
    }
#undef LH_SPREAD_X
    LH_PA(16, t_mk);
    if (FRONT) {
        int const dd_n = last - first + 1;
#pragma unroll
        for (int q = 0; q < NC; q++) {
            int const di = on ? (1 + 2 * dd[q]) / (2 * dd_n) : 0;
            float const avg_mask = lh_psy_tab_at(di) * 0.5f;
            float   e = ecb[q], x = m[q];
            e *= avg_mask;
            x *= t_minval;
            x *= avg_mask;
            ch[q].raw->v[0][b] = ebb[q];        /* 0 above the last partition */
            ch[q].raw->v[1][b] = on ? e : 0.0f;
            ch[q].raw->v[2][b] = on ? x : 0.0f;
        }
        LH_WAVE_SYNC_MEM();     /* every lane has read the factors it needs: the caller may reuse thr[] */
        return;
    }
    if (on) {
        float const masking_lower = t_mlow * lh_lds.ss.masking_lower;
        int const dd_n = last - first + 1;
#pragma unroll
        for (int q = 0; q < NC; q++) {
            float   x, avg_mask, e = ecb[q], t;
            int const di = (1 + 2 * dd[q]) / (2 * dd_n);
            avg_mask = lh_psy_tab_at(di) * 0.5f;
            e *= avg_mask;
            if (is_long) {
                int const bt_old = lh_lds.ss.blocktype_old[ch[q].chn & 1];
                float const n1v = *ch[q].nb1, n2v = *ch[q].nb2;
                if (bt_old == LH_SHORT_TYPE) {
                    float const ecb_limit = LH_RPELEV * n1v;
                    if (ecb_limit > 0)
                        t = __builtin_fminf(e, ecb_limit);
                    else {
                        float const alt = (float) (ebb[q] * LH_PREECHO_ATT2);
                        t = __builtin_fminf(e, alt);
                    }
                }
                else {
                    float   lim2 = LH_RPELEV2 * n2v;
                    float   lim1 = LH_RPELEV * n1v;
                    float   lim;
                    if (lim2 <= 0)
                        lim2 = e;
                    if (lim1 <= 0)
                        lim1 = e;
                    if (bt_old == LH_NORM_TYPE)
                        lim = __builtin_fminf(lim1, lim2);
                    else
                        lim = lim1;
                    t = __builtin_fminf(e, lim);
                }
                *ch[q].nb2 = n1v;
                *ch[q].nb1 = e;
            }
            else
                t = e;
            x = m[q];
            x *= t_minval;
            x *= avg_mask;
            if (t > x)
                t = x;
            if (masking_lower > 1)
                t *= masking_lower;
            if (t > ebb[q])
                t = ebb[q];
            if (masking_lower < 1)
                t *= masking_lower;
            th[q] = t;
        }
    }
    LH_WAVE_SYNC_MEM();         /* every lane has read the factors it needs */
#pragma unroll
    for (int q = 0; q < NC; q++)
        ch[q].thr[b] = th[q];   /* 0 above the last partition */
    LH_WAVE_SYNC_MEM();
    LH_PA(17, t_mk);
}

#endif
