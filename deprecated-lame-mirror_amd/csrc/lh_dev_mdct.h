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

/* Polyphase time slot, split in two for the GPU (reference newmdct.c:430-814):
 * stage 1 -- the 16 independent tap sums of a slot (15 folded window rows + the
 * centre row), one (slot, row) unit per lane-iteration, written to pre[32];
 * stage 2 -- the fixed 32-point butterfly network, one slot per lane, in place. */
LH_DEVFN void
lh_subband_taps(const LhCtx & c, int ch, int x, int n, float *pre)
{
    if (n < 15) {
        int const x1 = x - n;
        int const x2 = x - 62 + n;
        const float *wp = LH_ENW + 10 + 18 * n;
        float   w, s, t;
        w = wp[-10];
        s = lh_smp(c, ch, x2 - 224) * w;
        t = lh_smp(c, ch, x1 + 224) * w;
#pragma unroll
        for (int k = 1; k < 8; k++) {
            w = wp[-10 + k];
            s += lh_smp(c, ch, x2 - 224 + 64 * k) * w;
            t += lh_smp(c, ch, x1 + 224 - 64 * k) * w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            w = wp[-2 + k];
            s += lh_smp(c, ch, x1 - 256 + 64 * k) * w;
            t -= lh_smp(c, ch, x2 + 256 - 64 * k) * w;
        }
        s *= wp[6];
        w = t - s;
        pre[2 * n] = t + s;
        pre[2 * n + 1] = wp[7] * w;
    }
    else {
        int const x1 = x - 15;
        const float *wp = LH_ENW + 280;
        float   s, t;
        t = lh_smp(c, ch, x1 - 16) * wp[-10];
        s = lh_smp(c, ch, x1 - 32) * wp[-2];
        t += (lh_smp(c, ch, x1 - 48) - lh_smp(c, ch, x1 + 16)) * wp[-9];
        s += lh_smp(c, ch, x1 - 96) * wp[-1];
        t += (lh_smp(c, ch, x1 - 80) + lh_smp(c, ch, x1 + 48)) * wp[-8];
        s += lh_smp(c, ch, x1 - 160) * wp[0];
        t += (lh_smp(c, ch, x1 - 112) - lh_smp(c, ch, x1 + 80)) * wp[-7];
        s += lh_smp(c, ch, x1 - 224) * wp[1];
        t += (lh_smp(c, ch, x1 - 144) + lh_smp(c, ch, x1 + 112)) * wp[-6];
        s -= lh_smp(c, ch, x1 + 32) * wp[2];
        t += (lh_smp(c, ch, x1 - 176) - lh_smp(c, ch, x1 + 144)) * wp[-5];
        s -= lh_smp(c, ch, x1 + 96) * wp[3];
        t += (lh_smp(c, ch, x1 - 208) + lh_smp(c, ch, x1 + 176)) * wp[-4];
        s -= lh_smp(c, ch, x1 + 160) * wp[4];
        t += (lh_smp(c, ch, x1 - 240) - lh_smp(c, ch, x1 + 208)) * wp[-3];
        s -= lh_smp(c, ch, x1 + 224);
        /* u = s - t and v = s + t, combined with rows 14/15 at the start of stage 2 */
        pre[30] = s - t;
        pre[31] = s + t;
    }
}

LH_STAGEFN void
lh_subband_network(int ch, int g, int slot)
{
    float  *io = &lh_lds.u.mdct.sb[ch][g][slot * 32];
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
    xr = a[22] - a[6];
    a[6] += a[22];
    a[22] = (float) (xr * LH_SQRT2);
    xr = a[23] - a[7];
    a[7] += a[23];
    a[23] = (float) (xr * LH_SQRT2 - a[7]);
    a[7] -= a[6];
    a[22] -= a[7];
    a[23] -= a[22];
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
    xr = (float) (LH_SQRT2 * (-a[18] + a[26]));
    a[18] += a[26];
    a[26] = xr - a[18];
    xr = (float) (LH_SQRT2 * (-a[19] + a[27]));
    a[19] += a[27];
    a[27] = xr - a[19];

    xr = a[2];
    a[19] -= a[3];
    a[3] -= xr;
    a[2] = a[31] - xr;
    a[31] += xr;
    xr = a[3];
    a[11] -= a[19];
    a[18] -= xr;
    a[3] = a[30] - xr;
    a[30] += xr;
    xr = a[18];
    a[27] -= a[11];
    a[19] -= xr;
    a[18] = a[15] - xr;
    a[15] += xr;
    xr = a[19];
    a[10] -= xr;
    a[19] = a[14] - xr;
    a[14] += xr;
    xr = a[10];
    a[11] -= xr;
    a[10] = a[23] - xr;
    a[23] += xr;
    xr = a[11];
    a[26] -= xr;
    a[11] = a[22] - xr;
    a[22] += xr;
    xr = a[26];
    a[27] -= xr;
    a[26] = a[7] - xr;
    a[7] += xr;
    xr = a[27];
    a[27] = a[6] - xr;
    a[6] += xr;

    LH_BYD(0, 4);
    LH_BYD(1, 5);
    LH_BYD(16, 20);
    LH_BYD(17, 21);
    xr = (float) (-LH_SQRT2 * (a[8] - a[12]));
    a[8] += a[12];
    a[12] = xr - a[8];
    xr = (float) (-LH_SQRT2 * (a[9] - a[13]));
    a[9] += a[13];
    a[13] = xr - a[9];
    xr = (float) (-LH_SQRT2 * (a[25] - a[29]));
    a[25] += a[29];
    a[29] = xr - a[25];
    xr = (float) (-LH_SQRT2 * (a[24] + a[28]));
    a[24] -= a[28];
    a[28] = xr - a[24];

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

/* reference newmdct.c:832-867, in place on 18 values */
LH_DEVFN void
lh_mdct_short(float *inout)
{
    for (int l = 0; l < 3; l++) {
        float   tc0, tc1, tc2, ts0, ts1, ts2;
        float  *p = inout + l;
        ts0 = p[2 * 3] * LH_WIN(LH_SHORT_TYPE, 0) - p[5 * 3];
        tc0 = p[0 * 3] * LH_WIN(LH_SHORT_TYPE, 2) - p[3 * 3];
        tc1 = ts0 + tc0;
        tc2 = ts0 - tc0;
        ts0 = p[5 * 3] * LH_WIN(LH_SHORT_TYPE, 0) + p[2 * 3];
        tc0 = p[3 * 3] * LH_WIN(LH_SHORT_TYPE, 2) + p[0 * 3];
        ts1 = ts0 + tc0;
        ts2 = -ts0 + tc0;
        tc0 = (float) ((p[1 * 3] * LH_WIN(LH_SHORT_TYPE, 1) - p[4 * 3]) * 2.069978111953089e-11);
        ts0 = (float) ((p[4 * 3] * LH_WIN(LH_SHORT_TYPE, 1) + p[1 * 3]) * 2.069978111953089e-11);
        p[3 * 0] = (float) (tc1 * 1.907525191737280e-11 + tc0);
        p[3 * 5] = (float) (-ts1 * 1.907525191737280e-11 + ts0);
        tc2 = (float) (tc2 * 0.86602540378443870761 * 1.907525191737281e-11);
        ts1 = (float) (ts1 * 0.5 * 1.907525191737281e-11 + ts0);
        p[3 * 1] = tc2 - ts1;
        p[3 * 2] = tc2 + ts1;
        tc1 = (float) (tc1 * 0.5 * 1.907525191737281e-11 - tc0);
        ts2 = (float) (ts2 * 0.86602540378443870761 * 1.907525191737281e-11);
        p[3 * 3] = tc1 + ts2;
        p[3 * 4] = tc1 - ts2;
    }
}

/* reference newmdct.c:869-941 */
LH_DEVFN void
lh_mdct_long(float *out, float const *in)
{
    float   ct, st;
    {
        float   tc1, tc2, tc3, tc4, ts5, ts6, ts7, ts8;
        tc1 = in[17] - in[9];
        tc3 = in[15] - in[11];
        tc4 = in[14] - in[12];
        ts5 = in[0] + in[8];
        ts6 = in[1] + in[7];
        ts7 = in[2] + in[6];
        ts8 = in[3] + in[5];
        out[17] = (ts5 + ts7 - ts8) - (ts6 - in[4]);
        st = (ts5 + ts7 - ts8) * LH_CX(7) + (ts6 - in[4]);
        ct = (tc1 - tc3 - tc4) * LH_CX(6);
        out[5] = ct + st;
        out[6] = ct - st;
        tc2 = (in[16] - in[10]) * LH_CX(6);
        ts6 = ts6 * LH_CX(7) + in[4];
        ct = tc1 * LH_CX(0) + tc2 + tc3 * LH_CX(1) + tc4 * LH_CX(2);
        st = -ts5 * LH_CX(4) + ts6 - ts7 * LH_CX(5) + ts8 * LH_CX(3);
        out[1] = ct + st;
        out[2] = ct - st;
        ct = tc1 * LH_CX(1) - tc2 - tc3 * LH_CX(2) + tc4 * LH_CX(0);
        st = -ts5 * LH_CX(5) + ts6 - ts7 * LH_CX(3) + ts8 * LH_CX(4);
        out[9] = ct + st;
        out[10] = ct - st;
        ct = tc1 * LH_CX(2) - tc2 + tc3 * LH_CX(0) - tc4 * LH_CX(1);
        st = ts5 * LH_CX(3) - ts6 + ts7 * LH_CX(4) - ts8 * LH_CX(5);
        out[13] = ct + st;
        out[14] = ct - st;
    }
    {
        float   ts1, ts2, ts3, ts4, tc5, tc6, tc7, tc8;
        ts1 = in[8] - in[0];
        ts3 = in[6] - in[2];
        ts4 = in[5] - in[3];
        tc5 = in[17] + in[9];
        tc6 = in[16] + in[10];
        tc7 = in[15] + in[11];
        tc8 = in[14] + in[12];
        out[0] = (tc5 + tc7 + tc8) + (tc6 + in[13]);
        ct = (tc5 + tc7 + tc8) * LH_CX(7) - (tc6 + in[13]);
        st = (ts1 - ts3 + ts4) * LH_CX(6);
        out[11] = ct + st;
        out[12] = ct - st;
        ts2 = (in[7] - in[1]) * LH_CX(6);
        tc6 = in[13] - tc6 * LH_CX(7);
        ct = tc5 * LH_CX(3) - tc6 + tc7 * LH_CX(4) + tc8 * LH_CX(5);
        st = ts1 * LH_CX(2) + ts2 + ts3 * LH_CX(0) + ts4 * LH_CX(1);
        out[3] = ct + st;
        out[4] = ct - st;
        ct = -tc5 * LH_CX(5) + tc6 - tc7 * LH_CX(3) - tc8 * LH_CX(4);
        st = ts1 * LH_CX(1) + ts2 - ts3 * LH_CX(2) - ts4 * LH_CX(0);
        out[7] = ct + st;
        out[8] = ct - st;
        ct = -tc5 * LH_CX(4) + tc6 - tc7 * LH_CX(5) - tc8 * LH_CX(3);
        st = ts1 * LH_CX(0) - ts2 + ts3 * LH_CX(1) - ts4 * LH_CX(2);
        out[15] = ct + st;
        out[16] = ct - st;
    }
}

/* polyphase filtering of the 36 slots of the current frame window of channel
 * `ch' into sb[1..2]; one wave (reference newmdct.c:958-973, 984-991) */
LH_STAGEFN void
lh_polyphase(int ch)
{
    LhCtx const c = lh_ctx_load();
    float   (*sb)[576] = lh_lds.u.mdct.sb[ch];
    const float *amp = c.T->amp_filter;
    /* stage 1: 36 slots x 16 tap rows */
    for (int u = c.lane; u < 36 * 16; u += 64) {
        int const s = u >> 4, n = u & 15;
        int const gr = s / 18, slot = s - gr * 18;
        lh_subband_taps(c, ch, 286 + 32 * s, n, &sb[1 + gr][slot * 32]);
    }
    LH_WAVE_SYNC();
    /* stage 2: butterfly network per slot */
    if (c.lane < 36) {
        int const s = c.lane;
        int const gr = s / 18, slot = s - gr * 18;
        float  *out = &sb[1 + gr][slot * 32];
        lh_subband_network(ch, 1 + gr, slot);
        if (slot & 1) {
            /* compensate for the inversion in the analysis filter */
            for (int band = 1; band < 32; band += 2)
                out[band] *= -1;
        }
    }
    LH_WAVE_SYNC();
    /* lowpass: scale each new sub-band sample once (it is reused as band0 next granule) */
    for (int i = c.lane; i < 2 * 576; i += 64) {
        int const g = i / 576, r = i - g * 576;
        int const col = r & 31;
        /* band index of this storage column: sb[...][k*32 + order[band]] */
        int     band = 0;
        for (int b = 0; b < 32; b++)
            if (lh_sb_order[b] == col)
                band = b;
        if (!(amp[band] < 1e-12) && amp[band] < 1.0)
            sb[1 + g][r] *= amp[band];
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
    float   (*sb)[576] = lh_lds.u.mdct.sb[ch];
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
            tmp[k * 3 + 9] = band0[(9 + k) * 32] * w - band0[(8 - k) * 32];
            tmp[k * 3 + 18] = band0[(14 - k) * 32] * w + band0[(15 + k) * 32];
            tmp[k * 3 + 10] = band0[(15 + k) * 32] * w - band0[(14 - k) * 32];
            tmp[k * 3 + 19] = band1[(2 - k) * 32] * w + band1[(3 + k) * 32];
            tmp[k * 3 + 11] = band1[(3 + k) * 32] * w - band1[(2 - k) * 32];
            tmp[k * 3 + 20] = band1[(8 - k) * 32] * w + band1[(9 + k) * 32];
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
            a = LH_WIN(type, k + 27) * band1[(k + 9) * 32]
                + LH_WIN(type, k + 36) * band1[(8 - k) * 32];
            b = LH_WIN(type, k + 9) * band0[(k + 9) * 32]
                - LH_WIN(type, k + 18) * band0[(8 - k) * 32];
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
            bu = p[k] * LH_CA(k) + p[-1 - k] * LH_CS(k);
            bd = p[k] * LH_CS(k) - p[-1 - k] * LH_CA(k);
            p[-1 - k] = bu;
            p[k] = bd;
        }
    }
    LH_WAVE_SYNC();
}

#endif
