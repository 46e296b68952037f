/*
 * lh_replaygain.c -- the "radio gain" of the stream that lame_encode_buffer feeds, for the LAME tag
 * (reference gain_analysis.c:199-476 + lame.c:1715-1720, 1565-1580; host C, handle API only: the frontend switches
 * it on by default, the library's own default is off).
 *
 * The measure (ReplayGain proposal): both channels pass an equal-loudness filter -- a 10th order IIR fitted to
 * the inverse loudness curve, then a 2nd order high-pass --, the mean square of every 50 ms window goes into a
 * histogram with 0.01 dB bins, and the title's level is the bin that 95 % of the windows stay below; the gain is
 * what brings that to the 89 dB reference (pink noise calibration 64.82 dB).
 *
 * Bit-exactness with the reference needs three things kept: the products of a filter tap are formed in float and
 * added in double in the reference's grouping; a window's squares are added piece by piece, where a piece is what
 * one call contributes to the window -- its first ten samples apart (the reference reads those from a prefix
 * buffer) --, inside a piece first n mod 4 single squares, then sums of four; the histogram bin is the truncated
 * double.  What is this file's own: a streaming formulation (ten samples of input / intermediate / output history
 * per channel instead of the reference's window-sized buffers that are shifted at every window end).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "lh_host.h"
#include "lh_replaygain_tables.h"

#define RG_ORDER 10

int
lh_rg_start(LhReplayGain * g, int samplerate)
{
    static const int rates[9] = { 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000 };
    int     i;
    memset(g, 0, sizeof(*g));
    g->rate_index = -1;
    for (i = 0; i < 9; i++)
        if (rates[i] == samplerate)
            g->rate_index = i;
    if (g->rate_index < 0)
        return -1;
    g->window = (samplerate * 1L + 20L - 1) / 20L;      /* 50 ms, rounded up */
    return 0;
}

/* one piece of one channel: n samples at x (the ten before them in hist_in), through both filters; the outputs'
 * squares are added up in the reference's grouping */
static double
piece_energy(LhReplayGain * g, int ch, const float *x, int n)
{
    const float *ky = lh_rg_yule[g->rate_index], *kb = lh_rg_butter[g->rate_index];
    float  *in = g->work[0], *mid = g->work[1], *out = g->work[2];
    const float *o;
    double  sum = 0;
    int     i;
    memcpy(in, g->hist_in[ch], RG_ORDER * sizeof(float));
    memcpy(in + RG_ORDER, x, (size_t) n * sizeof(float));
    memcpy(mid, g->hist_mid[ch], RG_ORDER * sizeof(float));
    memcpy(out, g->hist_out[ch], RG_ORDER * sizeof(float));
    for (i = 0; i < n; i++) {
        const float *p = in + RG_ORDER + i;
        float  *q = mid + RG_ORDER + i;
        double const y0 = p[-10] * ky[0], y2 = p[-9] * ky[1], y4 = p[-8] * ky[2], y6 = p[-7] * ky[3];
        double const s00 = y0 + y2 + y4 + y6;
        double const y8 = p[-6] * ky[4], yA = p[-5] * ky[5], yC = p[-4] * ky[6], yE = p[-3] * ky[7];
        double const s01 = y8 + yA + yC + yE;
        double const yG = p[-2] * ky[8] + p[-1] * ky[9];
        double const yK = p[0] * ky[10];
        double const s1 = s00 + s01 + yG + yK;
        double const x1 = q[-10] * ky[11] + q[-9] * ky[12];
        double const x5 = q[-8] * ky[13] + q[-7] * ky[14];
        double const x9 = q[-6] * ky[15] + q[-5] * ky[16];
        double const xD = q[-4] * ky[17] + q[-3] * ky[18];
        double const xH = q[-2] * ky[19] + q[-1] * ky[20];
        double const s2 = x1 + x5 + x9 + xD + xH;
        q[0] = (float) (s1 - s2);
    }
    for (i = 0; i < n; i++) {
        const float *p = mid + RG_ORDER + i;
        float  *q = out + RG_ORDER + i;
        double const s1 = p[-2] * kb[0] + p[-1] * kb[2] + p[0] * kb[4];
        double const s2 = q[-2] * kb[1] + q[-1] * kb[3];
        q[0] = (float) (s1 - s2);
    }
    o = out + RG_ORDER;
    for (i = n & 3; i > 0; i--) {
        double const v = *o++;
        sum += v * v;
    }
    for (i = n / 4; i > 0; i--) {
        double const a = o[0] * o[0], b = o[1] * o[1], c = o[2] * o[2], d = o[3] * o[3];
        double const four = a + b + c + d;
        sum += four;
        o += 4;
    }
    /* the last ten of each become the history (a piece shorter than ten keeps part of the old one) */
    memcpy(g->hist_in[ch], in + n, RG_ORDER * sizeof(float));
    memcpy(g->hist_mid[ch], mid + n, RG_ORDER * sizeof(float));
    memcpy(g->hist_out[ch], out + n, RG_ORDER * sizeof(float));
    return sum;
}

/* one block as lame_encode_buffer hands it to the analysis (at most one frame's worth of samples) */
int
lh_rg_block(LhReplayGain * g, const float *l, const float *r, int n, int channels)
{
    int     at = 0;
    if (g->rate_index < 0 || n <= 0)
        return 0;
    if (n > LH_RG_MAX_BLOCK)
        return -1;
    if (channels == 1)
        r = l;
    while (at < n) {
        long    take = n - at;
        if (take > g->window - g->filled)
            take = g->window - g->filled;
        if (at < RG_ORDER && take > RG_ORDER - at)
            take = RG_ORDER - at;       /* the call's first ten samples are a piece of their own */
        g->lsum += piece_energy(g, 0, l + at, (int) take);
        g->rsum += piece_energy(g, 1, r + at, (int) take);
        at += (int) take;
        g->filled += take;
        if (g->filled == g->window) {
            double const level = 100 * 10. * log10((g->lsum + g->rsum) / g->filled * 0.5 + 1.e-37);
            size_t  bin = (level <= 0) ? 0 : (size_t) level;
            if (bin >= LH_RG_BINS)
                bin = LH_RG_BINS - 1;
            g->bins[bin]++;
            g->lsum = g->rsum = 0.;
            g->filled = 0;
        }
    }
    return 0;
}

/* the title's gain in tenths of a dB as the tag stores it (reference lame.c:1571-1578), 0 when no window was
 * completed; the state is ready for the next title afterwards (--nogap) */
int
lh_rg_finish(LhReplayGain * g)
{
    unsigned long windows = 0, upper, seen = 0;
    size_t  i;
    int     tenths = 0;
    if (g->rate_index < 0)
        return 0;
    for (i = 0; i < LH_RG_BINS; i++)
        windows += g->bins[i];
    if (windows != 0) {
        float   gain;
        upper = (unsigned long) (uint32_t) ceil((uint32_t) windows * (1. - 0.95));
        for (i = LH_RG_BINS; i-- > 0;) {
            seen += g->bins[i];
            if (seen >= upper)
                break;
        }
        gain = (float) ((float) 64.82 - (float) i / (float) 100);
        tenths = (int) floor(gain * 10.0 + 0.5);
    }
    memset(g->bins, 0, sizeof(g->bins));
    memset(g->hist_in, 0, sizeof(g->hist_in));
    memset(g->hist_mid, 0, sizeof(g->hist_mid));
    memset(g->hist_out, 0, sizeof(g->hist_out));
    g->filled = 0;
    g->lsum = g->rsum = 0.;
    return tenths;
}
