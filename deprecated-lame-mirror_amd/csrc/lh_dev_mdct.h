/*
 * lh_dev_mdct.h -- polyphase analysis + MDCT, wave-parallel
 * (reference newmdct.c:430-1039).
 *
 * Mapping for one channel (= one wave) and one frame:
 *   - the 36 polyphase time slots are independent: lane s computes slot s (512
 *     taps folded to 32 sub-band samples by the fixed butterfly network)
 *   - the 2 x 32 (granule, sub-band) MDCTs are independent: lane = gr*32 + band
 *   - the alias-reduction butterflies touch disjoint line pairs: lanes over pairs
 * Spectra must be bit-exact (they feed every integer decision downstream), so
 * each lane evaluates exactly the reference's operation sequence.
 */
#ifndef LH_DEV_MDCT_H
#define LH_DEV_MDCT_H

#include "lh_dev_common.h"

#define LH_ENW lh_enwindow
#define LH_WIN(t,i) lh_mdct_win[(t)*36+(i)]
/* the same where the index differs from lane to lane (the window of a lane's block type, the alias butterflies' constants):
 * vector loads -- lh_subband.hip keeps a copy of the table in LDS for them (LH_MDCT_WIN); constant indices stay immediates */
#ifndef LH_MDCT_WIN
#define LH_MDCT_WIN lh_mdct_win
#endif
#define LH_WINV(t,i) LH_MDCT_WIN[(t)*36+(i)]
#define LH_TANTAB(i) LH_WIN(LH_SHORT_TYPE, 3 + (i))
#define LH_CX(i) LH_WIN(LH_SHORT_TYPE, 12 + (i))
#define LH_CA(i) LH_WIN(LH_SHORT_TYPE, 20 + (i))
#define LH_CS(i) LH_WIN(LH_SHORT_TYPE, 28 + (i))
#define LH_WK(n) LH_ENW[287 - 18 * (n)]

#define LH_BX(m,s,c)  do { float xr_ = a[m] - a[s]; a[s] += a[m]; a[m] = xr_ * (c); } while (0)
#define LH_BY(s,m,c)  do { float xr_ = a[s] - a[m]; a[s] += a[m]; a[m] = xr_ * (c); } while (0)
#define LH_BYD(s,m)   do { float xr_ = (float) (LH_SQRT2 * (a[s] - a[m])); a[s] += a[m]; a[m] = xr_; } while (0)
#define LH_SW(p,q)    do { float xr_ = a[p]; a[p] = a[q] - xr_; a[q] = a[q] + xr_; } while (0)
#define LH_FS(p,q)    do { float xr_ = a[p]; a[p] += a[q]; a[q] -= xr_; } while (0)
#define LH_CH0(p,q)   do { xr = a[p] - a[q]; a[p] = xr; } while (0)
#define LH_CH(p)      do { xr = a[p] - xr; a[p] = xr; } while (0)
/* the remaining step kinds of the 32-point network (reference newmdct.c:552-705), so that the whole
 * network reads as one list of steps:
 *   SUB(p,q)   a[p] -= a[q]
 *   XS(m,s)    difference scaled by sqrt 2 in double, sum in place
 *   XS2(m,s)   the same, minus the new sum
 *   NYD(s,m)   sqrt 2 * (a[m] - a[s]) minus the new sum   (the difference is formed as -a[s] + a[m])
 *   MYD(s,m)   -sqrt 2 * (a[s] - a[m]) minus the new sum
 *   MYP(s,m)   -sqrt 2 * (a[s] + a[m]) minus the new difference */
#define LH_SUB(p,q)   do { a[p] -= a[q]; } while (0)
#define LH_XS(m,s)    do { float d_ = a[m] - a[s]; a[s] += a[m]; a[m] = (float) (d_ * LH_SQRT2); } while (0)
#define LH_XS2(m,s)   do { float d_ = a[m] - a[s]; a[s] += a[m]; a[m] = (float) (d_ * LH_SQRT2 - a[s]); } while (0)
#define LH_NYD(s,m)   do { float d_ = (float) (LH_SQRT2 * (-a[s] + a[m])); a[s] += a[m]; a[m] = d_ - a[s]; } while (0)
#define LH_MYD(s,m)   do { float d_ = (float) (-LH_SQRT2 * (a[s] - a[m])); a[s] += a[m]; a[m] = d_ - a[s]; } while (0)
#define LH_MYP(s,m)   do { float d_ = (float) (-LH_SQRT2 * (a[s] + a[m])); a[s] -= a[m]; a[m] = d_ - a[s]; } while (0)

/* Polyphase time slot, split in two for the GPU (reference newmdct.c:430-814):
 * stage 1 -- the 16 independent tap sums of a slot (15 folded window rows + the
 * centre row), one (slot, row) unit per lane-iteration, written to pre[32];
 * stage 2 -- the fixed 32-point butterfly network, one slot per lane, in place. */
/* LH_MF_SWZ(i): where sample i of the staged window lies in mf[ch][].  The rows of a slot read 32 samples at a stride of 64,
 * neighbouring rows neighbouring samples, neighbouring slots samples 32 apart -- the four slots of a wave's trip land on the
 * same 15 LDS banks.  lh_subband.hip stores the window with bit 4 of the index flipped in every odd block of 32
 * (i ^ ((i >> 1) & 16)): odd slots then use the other half of the banks, and an offset of a multiple of 64 commutes with the
 * swizzle, so the strided reads keep their immediate offsets.  The fused kernel (whose psy model reads the same window) keeps
 * the plain layout. */
#ifndef LH_MF_SWZ
#define LH_MF_SWZ(i) (i)
#endif
/* LH_ENW_TAPS: where the tap sums read the window's coefficients (a row per lane: vector loads).  The encode kernel reads
 * the constant table; lh_subband.hip -- whose 36 x 16 tap sums per channel and frame made it bound by the vector-memory
 * path, 32 coefficient loads per sum -- keeps a copy in LDS (rows 18 words apart: no two of the fifteen share a bank). */
#ifndef LH_ENW_TAPS
#define LH_ENW_TAPS LH_ENW
#endif
/* one of the fifteen folded rows (n < 15) of a slot; w[] = the row's eighteen coefficients (LH_ENW_TAPS + 18 n), which a lane
 * -- always at the same row -- holds in registers for all slots of a frame */
LH_DEVFN void
lh_subband_row(int ch, int x, int n, const float (&w)[18], float *pre)
{
    const float *mf = lh_lds.mf[ch];
    int const x1 = x - n;
    int const x2 = x - 62 + n;
    /* the four strided runs (bases; + / - 64 k from there) */
    const float *a2 = mf + LH_MF_SWZ(x2 - 224), *a1 = mf + LH_MF_SWZ(x1 + 224 - 448);
    const float *b1 = mf + LH_MF_SWZ(x1 - 256), *b2 = mf + LH_MF_SWZ(x2 + 256 - 448);
    float   s, t, d;
    s = a2[0] * w[0];
    t = a1[448] * w[0];
#pragma unroll
    for (int k = 1; k < 8; k++) {
        s += a2[64 * k] * w[k];
        t += a1[448 - 64 * k] * w[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        s += b1[64 * k] * w[8 + k];
        t -= b2[448 - 64 * k] * w[8 + k];
    }
    s *= w[16];
    d = t - s;
    pre[2 * n] = t + s;
    pre[2 * n + 1] = w[17] * d;
}

/* the centre row of a slot */
LH_DEVFN void
lh_subband_centre(int ch, int x, float *pre)
{
    const float *mf = lh_lds.mf[ch];
    int const x1 = x - 15;
    const float *wp = LH_ENW_TAPS + 280;
    float   s, t;
#define LH_MFS(i) mf[LH_MF_SWZ(i)]
    t = LH_MFS(x1 - 16) * wp[-10];
    s = LH_MFS(x1 - 32) * wp[-2];
    t += (LH_MFS(x1 - 48) - LH_MFS(x1 + 16)) * wp[-9];
    s += LH_MFS(x1 - 96) * wp[-1];
    t += (LH_MFS(x1 - 80) + LH_MFS(x1 + 48)) * wp[-8];
    s += LH_MFS(x1 - 160) * wp[0];
    t += (LH_MFS(x1 - 112) - LH_MFS(x1 + 80)) * wp[-7];
    s += LH_MFS(x1 - 224) * wp[1];
    t += (LH_MFS(x1 - 144) + LH_MFS(x1 + 112)) * wp[-6];
    s -= LH_MFS(x1 + 32) * wp[2];
    t += (LH_MFS(x1 - 176) - LH_MFS(x1 + 144)) * wp[-5];
    s -= LH_MFS(x1 + 96) * wp[3];
    t += (LH_MFS(x1 - 208) + LH_MFS(x1 + 176)) * wp[-4];
    s -= LH_MFS(x1 + 160) * wp[4];
    t += (LH_MFS(x1 - 240) - LH_MFS(x1 + 208)) * wp[-3];
    s -= LH_MFS(x1 + 224);
#undef LH_MFS
    /* u = s - t and v = s + t, combined with rows 14/15 at the start of stage 2 */
    pre[30] = s - t;
    pre[31] = s + t;
}

LH_STAGEFN void
lh_subband_network(int ch, int g, int slot)
{
    float  *io = &lh_lds.u.mdct.sb[ch][g][slot * LH_SB_STRIDE];
    float   a[32];
    float   xr;
#pragma unroll
    for (int i = 0; i < 32; i++)
        a[i] = io[i];
    {
        float const u = a[30], v = a[31];
        float const t = a[14];
        float const s = a[15] - t;
        a[31] = v + t;
        a[30] = u + s;
        a[15] = u - s;
        a[14] = v - t;
    }
    LH_BX(28, 0, LH_WK(2));
    LH_BX(29, 1, LH_WK(2));
    LH_BX(26, 2, LH_WK(4));
    LH_BX(27, 3, LH_WK(4));
    LH_BX(24, 4, LH_WK(6));
    LH_BX(25, 5, LH_WK(6));
    LH_XS(22, 6);
    LH_XS2(23, 7);
    LH_SUB(7, 6);
    LH_SUB(22, 7);
    LH_SUB(23, 22);
    LH_SW(6, 31);
    LH_SW(7, 30);
    LH_SW(22, 15);
    LH_SW(23, 14);
    LH_BX(20, 8, LH_WK(10));
    LH_BX(21, 9, LH_WK(10));
    LH_BX(18, 10, LH_WK(12));
    LH_BX(19, 11, LH_WK(12));
    LH_BX(16, 12, LH_WK(14));
    LH_BX(17, 13, LH_WK(14));
    LH_BX(24, 20, LH_WK(12));
    LH_BX(25, 21, LH_WK(12));
    LH_BY(4, 8, LH_WK(12));
    LH_BY(5, 9, LH_WK(12));
    LH_BY(0, 12, LH_WK(4));
    LH_BY(1, 13, LH_WK(4));
    LH_BY(16, 28, LH_WK(4));
    LH_BX(29, 17, LH_WK(4));
    LH_BYD(2, 10);
    LH_BYD(3, 11);
    LH_NYD(18, 26);
    LH_NYD(19, 27);
    /* a chain of eight exchanges (value p against r: a[p] = a[r] - a[p], a[r] += old a[p]), each
     * preceded by the subtractions that still need the old values */
    LH_SUB(19, 3);
    LH_SUB(3, 2);
    LH_SW(2, 31);
    LH_SUB(11, 19);
    LH_SUB(18, 3);
    LH_SW(3, 30);
    LH_SUB(27, 11);
    LH_SUB(19, 18);
    LH_SW(18, 15);
    LH_SUB(10, 19);
    LH_SW(19, 14);
    LH_SUB(11, 10);
    LH_SW(10, 23);
    LH_SUB(26, 11);
    LH_SW(11, 22);
    LH_SUB(27, 26);
    LH_SW(26, 7);
    LH_SW(27, 6);

    LH_BYD(0, 4);
    LH_BYD(1, 5);
    LH_BYD(16, 20);
    LH_BYD(17, 21);
    LH_MYD(8, 12);
    LH_MYD(9, 13);
    LH_MYD(25, 29);
    LH_MYP(24, 28);

    LH_CH0(24, 16);
    LH_CH(20);
    LH_CH(28);
    LH_CH0(25, 17);
    LH_CH(21);
    LH_CH(29);
    LH_CH0(17, 1);
    LH_CH(9);
    LH_CH(25);
    LH_CH(5);
    LH_CH(21);
    LH_CH(13);
    LH_CH(29);
    LH_CH0(1, 0);
    LH_CH(16);
    LH_CH(17);
    LH_CH(8);
    LH_CH(9);
    LH_CH(24);
    LH_CH(25);
    LH_CH(4);
    LH_CH(5);
    LH_CH(20);
    LH_CH(21);
    LH_CH(12);
    LH_CH(13);
    LH_CH(28);
    LH_CH(29);

    LH_FS(0, 31);
    LH_FS(1, 30);
    LH_FS(16, 15);
    LH_FS(17, 14);
    LH_FS(8, 23);
    LH_FS(9, 22);
    LH_FS(24, 7);
    LH_FS(25, 6);
    LH_FS(4, 27);
    LH_FS(5, 26);
    LH_FS(20, 11);
    LH_FS(21, 10);
    LH_FS(12, 19);
    LH_FS(13, 18);
    LH_FS(28, 3);
    LH_FS(29, 2);
#pragma unroll
    for (int i = 0; i < 32; i++)
        io[i] = a[i];
}

/* Three 6-point transforms of a short block, in place on 18 interleaved values (value i of window l at
 * [3 i + l]; reference newmdct.c:832-867).  Per window: the six inputs fold into two sums and two
 * differences and a scaled middle pair, which then rotate by 30 / 60 degrees.  The constants and the
 * order of the double-precision products are the reference's (the results are compared bit for bit). */
LH_DEVFN void
lh_mdct_short(float *inout)
{
    float const w0 = LH_WIN(LH_SHORT_TYPE, 0), w1 = LH_WIN(LH_SHORT_TYPE, 1), w2 = LH_WIN(LH_SHORT_TYPE, 2);
    double const k_mid = 2.069978111953089e-11, k_a = 1.907525191737280e-11, k_b = 1.907525191737281e-11;
    double const cos30 = 0.86602540378443870761;
    for (int l = 0; l < 3; l++) {
        float  *v = inout + l;
        float const x0 = v[0], x1 = v[3], x2 = v[6], x3 = v[9], x4 = v[12], x5 = v[15];
        float const lo_a = x2 * w0 - x5, lo_b = x0 * w2 - x3;
        float const hi_a = x5 * w0 + x2, hi_b = x3 * w2 + x0;
        float const lo_sum = lo_a + lo_b, lo_dif = lo_a - lo_b;
        float const hi_sum = hi_a + hi_b, hi_dif = -hi_a + hi_b;
        float const mid_lo = (float) ((x1 * w1 - x4) * k_mid);
        float const mid_hi = (float) ((x4 * w1 + x1) * k_mid);
        float const r1 = (float) (lo_dif * cos30 * k_b);
        float const q1 = (float) (hi_sum * 0.5 * k_b + mid_hi);
        float const r2 = (float) (lo_sum * 0.5 * k_b - mid_lo);
        float const q2 = (float) (hi_dif * cos30 * k_b);
        v[0] = (float) (lo_sum * k_a + mid_lo);
        v[15] = (float) (-hi_sum * k_a + mid_hi);
        v[3] = r1 - q1;
        v[6] = r1 + q1;
        v[9] = r2 + q2;
        v[12] = r2 - q2;
    }
}

/* a x[ia] (+/-) y (+/-) b x[ib] (+/-) c x[ic], summed left to right: one row of the 9-point rotations of
 * lh_mdct_long.  neg_* say which terms enter negated (x - y and x + (-y) are the same float). */
LH_DEVFN float
lh_mdct_row(float a, int ia, int neg_a, float y, int neg_y, float b, int ib, int neg_b, float c, int ic, int neg_c)
{
    float const ta = a * LH_CX(ia), tb = b * LH_CX(ib), tc = c * LH_CX(ic);
    float   acc = neg_a ? -ta : ta;
    acc = neg_y ? acc - y : acc + y;
    acc = neg_b ? acc - tb : acc + tb;
    acc = neg_c ? acc - tc : acc + tc;
    return acc;
}

/* 18-point MDCT of a long block (reference newmdct.c:869-941): the windowed input folds into four
 * differences and four sums per half; each half yields one direct pair, then three pairs from the rows of
 * a 3 x 3 rotation whose coefficients are permutations of LH_CX(0..2) / LH_CX(3..5). */
LH_DEVFN void
lh_mdct_long(float *out, float const *in)
{
    float const cx6 = LH_CX(6), cx7 = LH_CX(7);
    {
        /* differences of the upper nine, sums of the lower nine */
        float const d1 = in[17] - in[9], d3 = in[15] - in[11], d4 = in[14] - in[12];
        float const s5 = in[0] + in[8], s6 = in[1] + in[7], s7 = in[2] + in[6], s8 = in[3] + in[5];
        float const even = s5 + s7 - s8, odd = s6 - in[4];
        float const cd = (d1 - d3 - d4) * cx6, sd = even * cx7 + odd;
        float const d2 = (in[16] - in[10]) * cx6, s6m = s6 * cx7 + in[4];
        float   cr, sr;
        out[17] = even - odd;
        out[5] = cd + sd;
        out[6] = cd - sd;
        cr = lh_mdct_row(d1, 0, 0, d2, 0, d3, 1, 0, d4, 2, 0);
        sr = lh_mdct_row(s5, 4, 1, s6m, 0, s7, 5, 1, s8, 3, 0);
        out[1] = cr + sr;
        out[2] = cr - sr;
        cr = lh_mdct_row(d1, 1, 0, d2, 1, d3, 2, 1, d4, 0, 0);
        sr = lh_mdct_row(s5, 5, 1, s6m, 0, s7, 3, 1, s8, 4, 0);
        out[9] = cr + sr;
        out[10] = cr - sr;
        cr = lh_mdct_row(d1, 2, 0, d2, 1, d3, 0, 0, d4, 1, 1);
        sr = lh_mdct_row(s5, 3, 0, s6m, 1, s7, 4, 0, s8, 5, 1);
        out[13] = cr + sr;
        out[14] = cr - sr;
    }
    {
        /* differences of the lower nine, sums of the upper nine */
        float const e1 = in[8] - in[0], e3 = in[6] - in[2], e4 = in[5] - in[3];
        float const t5 = in[17] + in[9], t6 = in[16] + in[10], t7 = in[15] + in[11], t8 = in[14] + in[12];
        float const even = t5 + t7 + t8, odd = t6 + in[13];
        float const cd = even * cx7 - odd, sd = (e1 - e3 + e4) * cx6;
        float const e2 = (in[7] - in[1]) * cx6, t6m = in[13] - t6 * cx7;
        float   cr, sr;
        out[0] = even + odd;
        out[11] = cd + sd;
        out[12] = cd - sd;
        cr = lh_mdct_row(t5, 3, 0, t6m, 1, t7, 4, 0, t8, 5, 0);
        sr = lh_mdct_row(e1, 2, 0, e2, 0, e3, 0, 0, e4, 1, 0);
        out[3] = cr + sr;
        out[4] = cr - sr;
        cr = lh_mdct_row(t5, 5, 1, t6m, 0, t7, 3, 1, t8, 4, 1);
        sr = lh_mdct_row(e1, 1, 0, e2, 0, e3, 2, 1, e4, 0, 1);
        out[7] = cr + sr;
        out[8] = cr - sr;
        cr = lh_mdct_row(t5, 4, 1, t6m, 0, t7, 5, 1, t8, 3, 1);
        sr = lh_mdct_row(e1, 0, 0, e2, 1, e3, 1, 0, e4, 2, 1);
        out[15] = cr + sr;
        out[16] = cr - sr;
    }
}

/* polyphase filtering of the 36 slots of the current frame window of channel
 * `ch' into sb[1..2]; one wave (reference newmdct.c:958-973, 984-991) */
LH_STAGEFN void
lh_polyphase(int ch)
{
    LhCtx const c = lh_ctx_load();
    float   (*sb)[LH_SB_GRANULE] = lh_lds.u.mdct.sb[ch];
    const float *amp = c.T->amp_filter;
    /* stage 1: 36 slots x 16 tap rows.  A lane's units (lane + 64 k) are all at row lane & 15: the fifteen folded rows take
     * the loop with their coefficients in registers (the loads cannot be moved out of it by the compiler: the loop stores),
     * the lanes of row 15 sit it out, and the 36 centre rows follow as one trip of 36 lanes -- not nine trips in which every
     * wave walked through both kinds of row */
    {
        int const n = c.lane & 15;
        float   w[18];
#pragma unroll
        for (int k = 0; k < 18; k++)
            w[k] = LH_ENW_TAPS[18 * (n < 15 ? n : 0) + k];
        if (n < 15) {
            for (int s = c.lane >> 4; s < 36; s += 4) {
                int const gr = s / 18, slot = s - gr * 18;
                lh_subband_row(ch, 286 + 32 * s, n, w, &sb[1 + gr][slot * LH_SB_STRIDE]);
            }
        }
        if (c.lane < 36) {
            int const s = c.lane;
            int const gr = s / 18, slot = s - gr * 18;
            lh_subband_centre(ch, 286 + 32 * s, &sb[1 + gr][slot * LH_SB_STRIDE]);
        }
    }
    LH_WAVE_SYNC();
    /* stage 2: butterfly network per slot */
    if (c.lane < 36) {
        int const s = c.lane;
        int const gr = s / 18, slot = s - gr * 18;
        float  *out = &sb[1 + gr][slot * LH_SB_STRIDE];
        lh_subband_network(ch, 1 + gr, slot);
        if (slot & 1) {
            /* compensate for the inversion in the analysis filter */
            for (int band = 1; band < 32; band += 2)
                out[band] *= -1;
        }
    }
    LH_WAVE_SYNC();
    /* lowpass: scale each new sub-band sample once (it is reused as band0 next granule).  A lane's storage column is the same
     * on every trip (64 apart), and the column order -- bits 1..4 of the band reversed -- is its own inverse: the band and
     * its factor are found once, and bands the filter leaves alone (factor 1: all below the lowpass) take no trip at all. */
    {
        int const col = c.lane & 31;
        float const a = amp[lh_sb_order[col]];
        if (!(a < 1e-12) && a < 1.0) {
            for (int i = c.lane; i < 2 * 576; i += 64) {
                int const g = i / 576, r0 = i - g * 576;
                sb[1 + g][(r0 >> 5) * LH_SB_STRIDE + col] *= a;
            }
        }
    }
    LH_WAVE_SYNC();
}

/* MDCT + alias reduction for both granules of channel ch; one wave
 * (reference newmdct.c:978-1033) */
LH_STAGEFN void
lh_mdct_granules(int ch)
{
    LhCtx const c = lh_ctx_load();
    LhLds & L = lh_lds;
    float   (*sb)[LH_SB_GRANULE] = lh_lds.u.mdct.sb[ch];
    const float *amp = c.T->amp_filter;
    int const gr = c.lane >> 5, band = c.lane & 31;
    int const type = L.block_type[gr][ch];
    float  *mdct_enc = &L.xr[ch][gr][band * 18];
    float const *band0 = &sb[gr][lh_sb_order[band]];
    float const *band1 = &sb[1 + gr][lh_sb_order[band]];
    if (amp[band] < 1e-12) {
        for (int k = 0; k < 18; k++)
            mdct_enc[k] = 0.0f;
    }
    else if (type == LH_SHORT_TYPE) {
        float   tmp[18];
#pragma unroll
        for (int k = -3; k < 0; k++) {
            float const w = LH_WIN(LH_SHORT_TYPE, k + 3);
            tmp[k * 3 + 9] = band0[(9 + k) * LH_SB_STRIDE] * w - band0[(8 - k) * LH_SB_STRIDE];
            tmp[k * 3 + 18] = band0[(14 - k) * LH_SB_STRIDE] * w + band0[(15 + k) * LH_SB_STRIDE];
            tmp[k * 3 + 10] = band0[(15 + k) * LH_SB_STRIDE] * w - band0[(14 - k) * LH_SB_STRIDE];
            tmp[k * 3 + 19] = band1[(2 - k) * LH_SB_STRIDE] * w + band1[(3 + k) * LH_SB_STRIDE];
            tmp[k * 3 + 11] = band1[(3 + k) * LH_SB_STRIDE] * w - band1[(2 - k) * LH_SB_STRIDE];
            tmp[k * 3 + 20] = band1[(8 - k) * LH_SB_STRIDE] * w + band1[(9 + k) * LH_SB_STRIDE];
        }
        lh_mdct_short(tmp);
#pragma unroll
        for (int k = 0; k < 18; k++)
            mdct_enc[k] = tmp[k];
    }
    else {
        float   work[18], outv[18];
#pragma unroll
        for (int k = -9; k < 0; k++) {
            float   a, b;
            a = LH_WINV(type, k + 27) * band1[(k + 9) * LH_SB_STRIDE]
                + LH_WINV(type, k + 36) * band1[(8 - k) * LH_SB_STRIDE];
            b = LH_WINV(type, k + 9) * band0[(k + 9) * LH_SB_STRIDE]
                - LH_WINV(type, k + 18) * band0[(8 - k) * LH_SB_STRIDE];
            work[k + 9] = a - b * LH_TANTAB(k + 9);
            work[k + 18] = a * LH_TANTAB(k + 9) + b;
        }
        lh_mdct_long(outv, work);
#pragma unroll
        for (int k = 0; k < 18; k++)
            mdct_enc[k] = outv[k];
    }
    LH_WAVE_SYNC();
    /* aliasing reduction butterflies: pairs (band*18 + k, band*18 - 1 - k), band 1..31, k 0..7 */
    for (int t = c.lane; t < 2 * 31 * 8; t += 64) {
        int const g = t / (31 * 8), r = t - g * (31 * 8);
        int const bnd = 1 + r / 8, k = r & 7;
        if (L.block_type[g][ch] != LH_SHORT_TYPE) {
            float  *p = &L.xr[ch][g][bnd * 18];
            float   bu, bd;
            float const ca = LH_WINV(LH_SHORT_TYPE, 20 + k), cs = LH_WINV(LH_SHORT_TYPE, 28 + k);
            bu = p[k] * ca + p[-1 - k] * cs;
            bd = p[k] * cs - p[-1 - k] * ca;
            p[-1 - k] = bu;
            p[k] = bd;
        }
    }
    LH_WAVE_SYNC();
}

#endif
