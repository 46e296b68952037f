/*
 * orc_vbr.c -- CPU restatement of the reference's "new VBR" (vbr_mt / vbr_mtrh)
 * iteration loop.  TEST INFRASTRUCTURE ONLY (see orc_common.h).
 *
 * Follows reference libmp3lame/quantize.c:1582-1751 (VBR_new_prepare,
 * VBR_new_iteration_loop) and libmp3lame/vbrquantize.c (whole file: per-band
 * scalefactor search on x^(3/4), the long/short block constraint solvers, the
 * out-of-bits strategies and VBR_encode_frame).  Part of the single translation
 * unit lame_oracle.c: uses the static helpers of orc_quant.c.
 *
 * The reference is built with TAKEHIRO_IEEE754_HACK, so the quantiser works on
 * doubles with the 2^23 magic-number rounding; the float/double widths below
 * are the reference's.
 */

#define ORC_MAGIC_FLOAT (65536 * 128)
#define ORC_MAGIC_INT   0x4b000000

typedef struct OrcVbrGr {       /* algo_t, reference vbrquantize.c:47-55 */
    OrcStream *S;
    OrcGr  *gi;
    const float *xr34;
    int     is_short;
    int     guess;              /* full_outer_loop < 0: closed-form scalefactor guess instead of the search */
    int     mingain_l;
    int     mingain_s[3];
    int     sfwork[LH_SFBMAX];
    int     sfmin[LH_SFBMAX];
} OrcVbrGr;

/* scalefactor ranges the side information can carry, reference vbrquantize.c:527-539 (MPEG-1) */
static const unsigned char orc_range_short[LH_SBMAX_S * 3] = {
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7,
    0, 0, 0
};
static const unsigned char orc_range_long[LH_SBMAX_L] = {
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 0
};
/* MPEG-2 / 2.5 with preflag: the scalefactor partitions of table 2 carry 3, 2, 0, 0 bits (reference vbrquantize.c:577-579) */
static const unsigned char orc_range_long_lsf_pretab[LH_SBMAX_L] = {
    7, 7, 7, 7, 7, 7, 3, 3, 3, 3, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0
};

/* one value through the x^(3/4) quantiser (k_34_4 per element, reference vbrquantize.c:162-199) */
static inline int
vbr_q34(const LhTables * T, double x)
{
    union { float f; int i; } u;
    x += ORC_MAGIC_FLOAT;
    u.f = x;
    u.f = x + T->adj43asm[u.i - ORC_MAGIC_INT];
    return u.i - ORC_MAGIC_INT;
}

/* largest |xr|^(3/4) of a band (vec_max_c, reference vbrquantize.c:110-139) */
static float
vbr_band_max(const float *x, unsigned n)
{
    float   m = 0;
    unsigned k;
    for (k = 0; k < n; k++)
        if (m < x[k])
            m = x[k];
    return m;
}

/* smallest step index that keeps the band inside the quantiser's table
 * (find_lowest_scalefac, reference vbrquantize.c:142-159) */
static int
vbr_lowest_sf(const LhTables * T, float xmax34)
{
    int     ok = 255, sf = 128, del = 64, k;
    float const limit = LH_IXMAX;
    for (k = 0; k < 8; k++) {
        float const v = T->ipow20[sf] * xmax34;
        if (v <= limit) {
            ok = sf;
            sf -= del;
        }
        else
            sf += del;
        del >>= 1;
    }
    return ok;
}

/* quantisation noise of one band at step index sf (calc_sfb_noise_x34,
 * reference vbrquantize.c:210-262): groups of four, pairwise sums in double */
static float
vbr_band_noise(const LhTables * T, const float *xr, const float *xr34, unsigned bw, int sf)
{
    float const sfpow = T->pow20[sf + LH_QMAX2];
    float const sfpow34 = T->ipow20[sf];
    float   acc = 0;
    unsigned done = 0;
    while (done < bw) {
        double  e[4] = { 0, 0, 0, 0 };
        unsigned n = bw - done, k;
        if (n > 4)
            n = 4;
        for (k = 0; k < n; k++) {
            double const q = sfpow34 * xr34[done + k];
            int const l3 = vbr_q34(T, q);
            e[k] = fabsf(xr[done + k]) - sfpow * T->pow43[l3];
        }
        acc += (e[0] * e[0] + e[1] * e[1]) + (e[2] * e[2] + e[3] * e[3]);
        done += n;
    }
    return acc;
}

typedef struct OrcNoiseMemo {
    unsigned char valid[256];
    float   value[256];
} OrcNoiseMemo;

static float
vbr_memo_noise(const LhTables * T, OrcNoiseMemo * memo, const float *xr, const float *xr34, unsigned bw, int sf)
{
    if (!memo->valid[sf]) {
        memo->valid[sf] = 1;
        memo->value[sf] = vbr_band_noise(T, xr, xr34, bw, sf);
    }
    return memo->value[sf];
}

/* "too noisy" at sf or at either neighbour (tri_calc_sfb_noise_x34, reference vbrquantize.c:275-308) */
static int
vbr_too_noisy(const LhTables * T, OrcNoiseMemo * memo, const float *xr, const float *xr34, float xmin,
              unsigned bw, int sf)
{
    if (xmin < vbr_memo_noise(T, memo, xr, xr34, bw, sf))
        return 1;
    if (sf < 255 && xmin < vbr_memo_noise(T, memo, xr, xr34, bw, sf + 1))
        return 1;
    if (sf > 0 && xmin < vbr_memo_noise(T, memo, xr, xr34, bw, sf - 1))
        return 1;
    return 0;
}

/* largest step index whose noise stays under the allowed distortion
 * (find_scalefac_x34, reference vbrquantize.c:347-382) */
static int
vbr_search_sf(const LhTables * T, const float *xr, const float *xr34, float xmin, unsigned bw, int sf_min)
{
    OrcNoiseMemo memo;
    int     sf = 128, ok = 255, del = 128, seen = 0, k;
    memset(memo.valid, 0, sizeof(memo.valid));
    for (k = 0; k < 8; k++) {
        del >>= 1;
        if (sf <= sf_min)
            sf += del;
        else if (vbr_too_noisy(T, &memo, xr, xr34, xmin, bw, sf))
            sf -= del;
        else {
            ok = sf;
            sf += del;
            seen = 1;
        }
    }
    if (seen)
        sf = ok;
    if (sf <= sf_min)
        sf = sf_min;
    return sf;
}

/* closed-form estimate used at quality 7 (calc_scalefac + guess_scalefac_x34,
 * reference vbrquantize.c:315-333) */
static int
vbr_guess_sf(float xmin, unsigned bw, int sf_min)
{
    float const c = 5.799142446;
    int const g = 210 + (int) (c * log10f(xmin / (int) bw) - .5f);
    if (g < sf_min)
        return sf_min;
    if (g >= 255)
        return 255;
    return g;
}

/* per-band step indices for a granule (block_sf, reference vbrquantize.c:397-489) */
static int
vbr_band_steps(OrcVbrGr * V, const float *xmin)
{
    const LhTables *T = V->S->tab;
    OrcGr const *gi = V->gi;
    unsigned const top = (unsigned) gi->max_nonzero_coeff;
    unsigned j = 0, win = 0;
    int     sfb = 0, maxsf = 0, below255 = -1;

    V->mingain_l = 0;
    V->mingain_s[0] = V->mingain_s[1] = V->mingain_s[2] = 0;
    while (j <= top) {
        unsigned const w = (unsigned) gi->width[sfb];
        unsigned const room = top - j + 1;
        unsigned const n = (w > room) ? room : w;
        int const floor_sf = vbr_lowest_sf(T, vbr_band_max(V->xr34 + j, n));
        int     sf;
        V->sfmin[sfb] = floor_sf;
        if (V->mingain_l < floor_sf)
            V->mingain_l = floor_sf;
        if (V->mingain_s[win] < floor_sf)
            V->mingain_s[win] = floor_sf;
        if (++win > 2)
            win = 0;
        if (sfb < gi->psymax && w > 2) {
            if (gi->energy_above_cutoff[sfb]) {
                sf = V->guess ? vbr_guess_sf(xmin[sfb], n, floor_sf)
                    : vbr_search_sf(T, gi->xr + j, V->xr34 + j, xmin[sfb], n, floor_sf);
                if (maxsf < sf)
                    maxsf = sf;
                if (below255 < sf && sf < 255)
                    below255 = sf;
            }
            else {
                sf = 255;
                maxsf = 255;
            }
        }
        else {
            if (maxsf < floor_sf)
                maxsf = floor_sf;
            sf = maxsf;
        }
        V->sfwork[sfb] = sf;
        ++sfb;
        j += w;
    }
    for (; sfb < LH_SFBMAX; ++sfb) {
        V->sfwork[sfb] = maxsf;
        V->sfmin[sfb] = 0;
    }
    if (below255 > -1) {
        maxsf = below255;
        for (sfb = 0; sfb < LH_SFBMAX; ++sfb)
            if (V->sfwork[sfb] == 255)
                V->sfwork[sfb] = below255;
    }
    return maxsf;
}

/* quantise the granule with its current gains (quantize_x34, reference vbrquantize.c:505-570);
 * lines above max_nonzero_coeff keep what l3_enc held */
static void
vbr_quantize(const OrcVbrGr * V)
{
    const LhTables *T = V->S->tab;
    OrcGr  *gi = V->gi;
    int const ifqstep = (gi->scalefac_scale == 0) ? 2 : 4;
    unsigned const top = (unsigned) gi->max_nonzero_coeff;
    unsigned j = 0, pos = 0;
    int     sfb = 0;
    while (j <= top) {
        int const s = (gi->scalefac[sfb] + (gi->preflag ? lh_pretab[sfb] : 0)) * ifqstep
            + gi->subblock_gain[gi->window[sfb]] * 8;
        unsigned const sfac = (unsigned) (gi->global_gain - s) & 255u;
        float const sfpow34 = T->ipow20[sfac];
        unsigned const w = (unsigned) gi->width[sfb];
        unsigned const room = top - j + 1;
        unsigned const n = (w <= room) ? w : room;
        unsigned k;
        for (k = 0; k < n; k++) {
            double const q = sfpow34 * V->xr34[pos + k];
            gi->l3_enc[pos + k] = vbr_q34(T, q);
        }
        pos += n;
        j += w;
        ++sfb;
    }
}

/* subblock gains that bring the short-block scalefactors into range
 * (set_subblock_gain, reference vbrquantize.c:553-642); sf[] = step - global, <= 0 wanted */
static void
vbr_subblock_gain(OrcGr * gi, const int mingain_s[3], int sf[LH_SFBMAX])
{
    int const shift = (gi->scalefac_scale == 0) ? 1 : 2;
    unsigned psydiv = 18, sfb, i;
    int     min_sbg = 7;
    int    *sbg = gi->subblock_gain;
    if (psydiv > (unsigned) gi->psymax)
        psydiv = (unsigned) gi->psymax;
    for (i = 0; i < 3; ++i) {
        int     need1 = 0, need2 = 0, least = 1000, a, b;
        for (sfb = i; sfb < psydiv; sfb += 3) {
            int const v = -sf[sfb];
            if (need1 < v)
                need1 = v;
            if (least > v)
                least = v;
        }
        for (; sfb < LH_SFBMAX; sfb += 3) {
            int const v = -sf[sfb];
            if (need2 < v)
                need2 = v;
            if (least > v)
                least = v;
        }
        a = need1 - (15 << shift);
        b = need2 - (7 << shift);
        need1 = (a > b) ? a : b;
        sbg[i] = (least > 0) ? (least >> 3) : 0;
        if (need1 > 0) {
            int const up = (need1 + 7) >> 3;
            if (sbg[i] < up)
                sbg[i] = up;
        }
        if (sbg[i] > 0 && mingain_s[i] > (gi->global_gain - sbg[i] * 8))
            sbg[i] = (gi->global_gain - mingain_s[i]) >> 3;
        if (sbg[i] > 7)
            sbg[i] = 7;
        if (min_sbg > sbg[i])
            min_sbg = sbg[i];
    }
    for (sfb = 0; sfb < LH_SFBMAX; sfb += 3) {
        sf[sfb + 0] += sbg[0] * 8;
        sf[sfb + 1] += sbg[1] * 8;
        sf[sfb + 2] += sbg[2] * 8;
    }
    if (min_sbg > 0) {
        for (i = 0; i < 3; ++i)
            sbg[i] -= min_sbg;
        gi->global_gain -= min_sbg * 8;
    }
}

/* scalefactors from the (negative) step offsets (set_scalefacs, reference vbrquantize.c:653-700) */
static void
vbr_scalefacs(OrcGr * gi, const int *sfmin, int sf[LH_SFBMAX], const unsigned char *range)
{
    int const ifqstep = (gi->scalefac_scale == 0) ? 2 : 4;
    int const shift = (gi->scalefac_scale == 0) ? 1 : 2;
    int     sfb;
    if (gi->preflag)
        for (sfb = 11; sfb < gi->sfbmax; ++sfb)
            sf[sfb] += lh_pretab[sfb] * ifqstep;
    for (sfb = 0; sfb < gi->sfbmax; ++sfb) {
        int const gain = gi->global_gain - gi->subblock_gain[gi->window[sfb]] * 8
            - (gi->preflag ? lh_pretab[sfb] : 0) * ifqstep;
        int     sc = 0;
        if (sf[sfb] < 0) {
            int const m = gain - sfmin[sfb];
            sc = (ifqstep - 1 - sf[sfb]) >> shift;
            if (sc > range[sfb])
                sc = range[sfb];
            if (sc > 0 && (sc << shift) > m)
                sc = m >> shift;
        }
        gi->scalefac[sfb] = sc;
    }
    for (; sfb < LH_SFBMAX; ++sfb)
        gi->scalefac[sfb] = 0;
}

static int
vbr_clamp_gain(int g)
{
    return g < 0 ? 0 : (g > 255 ? 255 : g);
}

/* short blocks (short_block_constrain, reference vbrquantize.c:748-815) */
static void
vbr_constrain_short(const OrcVbrGr * V, const int steps[LH_SFBMAX], int vbrmax)
{
    OrcGr  *gi = V->gi;
    int     over0 = 0, over1 = 0, delta = 0, mover, sfb;
    int     tmp[LH_SFBMAX];
    for (sfb = 0; sfb < gi->psymax; ++sfb) {
        int const v = vbrmax - steps[sfb];
        int const v0 = v - (4 * 14 + 2 * orc_range_short[sfb]);
        int const v1 = v - (4 * 14 + 4 * orc_range_short[sfb]);
        if (delta < v)
            delta = v;
        if (over0 < v0)
            over0 = v0;
        if (over1 < v1)
            over1 = v1;
    }
    if (V->S->cfg->noise_shaping == 2)
        mover = (over0 < over1) ? over0 : over1;
    else
        mover = over0;
    if (delta > mover)
        delta = mover;
    vbrmax -= delta;
    over0 -= mover;
    over1 -= mover;
    if (over0 == 0)
        gi->scalefac_scale = 0;
    else if (over1 == 0)
        gi->scalefac_scale = 1;
    if (vbrmax < V->mingain_l)
        vbrmax = V->mingain_l;
    gi->global_gain = vbr_clamp_gain(vbrmax);
    for (sfb = 0; sfb < LH_SFBMAX; ++sfb)
        tmp[sfb] = steps[sfb] - vbrmax;
    vbr_subblock_gain(gi, V->mingain_s, tmp);
    vbr_scalefacs(gi, V->sfmin, tmp, orc_range_short);
}

/* long blocks: choose scalefac_scale / preflag (long_block_constrain, reference
 * vbrquantize.c:826-978) */
static void
vbr_constrain_long(const OrcVbrGr * V, const int steps[LH_SFBMAX], int vbrmax)
{
    OrcGr  *gi = V->gi;
    int const floor_gain = V->mingain_l;
    int     over0 = 0, over1 = 0, over0p = 0, over1p = 0, delta = 0, mover, sfb;
    int     pre0 = 1, pre1 = 1;
    int     tmp[LH_SFBMAX];
    /* the ranges with preflag: MPEG-1's own, or the LSF table's (reference vbrquantize.c:861) */
    const unsigned char *rangep = (V->S->cfg->mode_gr == 2) ? orc_range_long : orc_range_long_lsf_pretab;
    for (sfb = 0; sfb < gi->psymax; ++sfb) {
        int const v = vbrmax - steps[sfb];
        int const r = orc_range_long[sfb], rp = rangep[sfb] + lh_pretab[sfb];
        if (delta < v)
            delta = v;
        if (over0 < v - 2 * r)
            over0 = v - 2 * r;
        if (over1 < v - 4 * r)
            over1 = v - 4 * r;
        if (over0p < v - 2 * rp)
            over0p = v - 2 * rp;
        if (over1p < v - 4 * rp)
            over1p = v - 4 * rp;
    }
    {
        int     gain = vbrmax - over0p;
        if (gain < floor_gain)
            gain = floor_gain;
        for (sfb = 0; sfb < gi->psymax; ++sfb)
            if ((gain - V->sfmin[sfb]) - 2 * lh_pretab[sfb] <= 0) {
                pre0 = 0;
                pre1 = 0;
                break;
            }
    }
    if (pre1) {
        int     gain = vbrmax - over1p;
        if (gain < floor_gain)
            gain = floor_gain;
        for (sfb = 0; sfb < gi->psymax; ++sfb)
            if ((gain - V->sfmin[sfb]) - 4 * lh_pretab[sfb] <= 0) {
                pre1 = 0;
                break;
            }
    }
    if (!pre0)
        over0p = over0;
    if (!pre1)
        over1p = over1;
    if (V->S->cfg->noise_shaping != 2) {
        over1 = over0;
        over1p = over0p;
    }
    mover = (over0 < over0p) ? over0 : over0p;
    mover = (mover < over1) ? mover : over1;
    mover = (mover < over1p) ? mover : over1p;
    if (delta > mover)
        delta = mover;
    vbrmax -= delta;
    if (vbrmax < floor_gain)
        vbrmax = floor_gain;
    over0 -= mover;
    over0p -= mover;
    over1 -= mover;
    over1p -= mover;
    if (over0 == 0) {
        gi->scalefac_scale = 0;
        gi->preflag = 0;
        rangep = orc_range_long;
    }
    else if (over0p == 0) {
        gi->scalefac_scale = 0;
        gi->preflag = 1;
    }
    else if (over1 == 0) {
        gi->scalefac_scale = 1;
        gi->preflag = 0;
        rangep = orc_range_long;
    }
    else if (over1p == 0) {
        gi->scalefac_scale = 1;
        gi->preflag = 1;
    }
    gi->global_gain = vbr_clamp_gain(vbrmax);
    for (sfb = 0; sfb < LH_SFBMAX; ++sfb)
        tmp[sfb] = steps[sfb] - vbrmax;
    vbr_scalefacs(gi, V->sfmin, tmp, rangep);
}

static void
vbr_constrain(const OrcVbrGr * V, const int steps[LH_SFBMAX], int vbrmax)
{
    if (V->is_short)
        vbr_constrain_short(V, steps, vbrmax);
    else
        vbr_constrain_long(V, steps, vbrmax);
    /* bitcount(): the scalefactors always fit by construction (reference vbrquantize.c:982-993) */
    (void) scale_bitcount(V->S, V->gi);
}

static int
vbr_quantize_count(const OrcVbrGr * V)
{
    vbr_quantize(V);
    V->gi->part2_3_length = noquant_count_bits(V->S, V->gi, 0);
    return V->gi->part2_3_length;
}

/* bits (incl. scalefactors) with these step indices (tryThatOne, reference vbrquantize.c:1139-1150) */
static int
vbr_try(const OrcVbrGr * V, const int steps[LH_SFBMAX], int vbrmax)
{
    float const keep = V->gi->xrpow_max;
    int     nbits;
    vbr_constrain(V, steps, vbrmax);
    nbits = vbr_quantize_count(V);
    nbits += V->gi->part2_length;
    V->gi->xrpow_max = keep;
    return nbits;
}

/* all steps shifted by delta (tryGlobalStepsize, reference vbrquantize.c:1009-1033); Huffman bits only */
static int
vbr_try_shift(const OrcVbrGr * V, const int steps[LH_SFBMAX], int delta)
{
    float const keep = V->gi->xrpow_max;
    int     shifted[LH_SFBMAX], i, nbits, vbrmax = 0;
    for (i = 0; i < LH_SFBMAX; ++i) {
        int     g = steps[i] + delta;
        if (g < V->sfmin[i])
            g = V->sfmin[i];
        if (g > 255)
            g = 255;
        if (vbrmax < g)
            vbrmax = g;
        shifted[i] = g;
    }
    vbr_constrain(V, shifted, vbrmax);
    nbits = vbr_quantize_count(V);
    V->gi->xrpow_max = keep;
    return nbits;
}

/* last resort: bisect the global gain (searchGlobalStepsizeMax, reference vbrquantize.c:1037-1069) */
static void
vbr_bisect_gain(const OrcVbrGr * V, const int steps[LH_SFBMAX], int target)
{
    OrcGr const *gi = V->gi;
    int const gain = gi->global_gain;
    int     curr = gain, good = 1024, lo = gain, hi = 512;
    while (lo <= hi) {
        int     nbits;
        curr = (lo + hi) >> 1;
        nbits = vbr_try_shift(V, steps, curr - gain);
        if (nbits == 0 || (nbits + gi->part2_length) < target) {
            hi = curr - 1;
            good = curr;
        }
        else {
            lo = curr + 1;
            if (good == 1024)
                good = curr;
        }
    }
    if (good != curr)
        (void) vbr_try_shift(V, steps, good - gain);
}

/* pull every band k/dm of the way towards p (flattenDistribution, reference vbrquantize.c:1101-1136) */
static int
vbr_flatten(const int in[LH_SFBMAX], int out[LH_SFBMAX], int dm, int k, int p)
{
    int     i, top = 0;
    for (i = 0; i < LH_SFBMAX; ++i) {
        int     x = in[i];
        if (dm > 0) {
            x = in[i] + (k * (p - in[i])) / dm;
            if (x < 0)
                x = 0;
            else if (x > 255)
                x = 255;
        }
        out[i] = x;
        if (top < x)
            top = x;
    }
    return top;
}

/* make the granule fit `target' bits (outOfBitsStrategy, reference vbrquantize.c:1153-1228):
 * first flatten the noise shaping, then raise the common step, then bisect the gain */
static void
vbr_fit(const OrcVbrGr * V, const int steps[LH_SFBMAX], int target)
{
    int     wrk[LH_SFBMAX];
    int     dm = 0, i, stage;
    int const p = V->gi->global_gain;
    for (i = 0; i < LH_SFBMAX; ++i)     /* sfDepth, reference vbrquantize.c:1073-1088 */
        if (dm < 255 - steps[i])
            dm = 255 - steps[i];
    for (stage = 0; stage < 2; stage++) {
        int     mid = stage ? (255 + p) / 2 : dm / 2;
        int     lo = stage ? p : 0;
        int     hi = stage ? 255 : dm;
        int     best = -1;
        for (;;) {
            int const top = stage ? vbr_flatten(steps, wrk, dm, dm, mid) : vbr_flatten(steps, wrk, dm, mid, p);
            int const nbits = vbr_try(V, wrk, top);
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
                int const top = stage ? vbr_flatten(steps, wrk, dm, dm, best) : vbr_flatten(steps, wrk, dm, best, p);
                (void) vbr_try(V, wrk, top);
            }
            return;
        }
    }
    vbr_bisect_gain(V, wrk, target);
}

/* reference vbrquantize.c:1231-1247 */
static int
vbr_reduce_bits(OrcStream * S, int gr, int ch)
{
    OrcGr  *gi = &S->tt[gr][ch];
    best_scalefac_store(S, gr, ch);
    if (S->cfg->use_best_huffman == 1)
        best_huffman_divide(S, gi);
    return gi->part2_3_length + gi->part2_length;
}

/* split a budget between two parts in proportion to f(need) and hand unused bits of one
 * part to the other (the repeated pattern of reference vbrquantize.c:1381-1500) */
static void
vbr_share(int share[2], const int use[2], int slack)
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

/* quantise all four granules with as few bits as the masking allows, then squeeze them
 * into max_bits if necessary (VBR_encode_frame, reference vbrquantize.c:1250-1580) */
static int
vbr_encode_frame(OrcStream * S, float xr34[2][2][576], float xmin[2][2][LH_SFBMAX], int max_bits[2][2])
{
    OrcVbrGr V[2][2];
    int     max_ch[2][2], max_gr[2] = { 0, 0 }, max_fr = 0;
    int     use_ch[2][2], use_gr[2], use_fr;
    int     gr, ch, ok, sum_fr;
    int const nch = S->cfg->channels;

    for (gr = 0; gr < S->cfg->mode_gr; ++gr)
        for (ch = 0; ch < nch; ++ch) {
            OrcVbrGr *v = &V[gr][ch];
            max_ch[gr][ch] = max_bits[gr][ch];
            max_gr[gr] += max_bits[gr][ch];
            max_fr += max_bits[gr][ch];
            v->S = S;
            v->gi = &S->tt[gr][ch];
            v->xr34 = xr34[gr][ch];
            v->is_short = (v->gi->block_type == LH_SHORT_TYPE);
            v->guess = (S->cfg->full_outer_loop < 0);
        }
    /* scalefactor search */
    for (gr = 0; gr < S->cfg->mode_gr; ++gr)
        for (ch = 0; ch < nch; ++ch)
            if (max_bits[gr][ch] > 0) {
                OrcVbrGr *v = &V[gr][ch];
                int const vbrmax = vbr_band_steps(v, xmin[gr][ch]);
                vbr_constrain(v, v->sfwork, vbrmax);
            }
    /* encode as it is */
    use_fr = 0;
    for (gr = 0; gr < S->cfg->mode_gr; ++gr) {
        use_gr[gr] = 0;
        for (ch = 0; ch < nch; ++ch) {
            if (max_bits[gr][ch] > 0) {
                memset(S->tt[gr][ch].l3_enc, 0, sizeof(S->tt[gr][ch].l3_enc));
                (void) vbr_quantize_count(&V[gr][ch]);
            }
            use_ch[gr][ch] = vbr_reduce_bits(S, gr, ch);
            use_gr[gr] += use_ch[gr][ch];
        }
        use_fr += use_gr[gr];
    }
    if (use_fr <= max_fr) {
        ok = 1;
        for (gr = 0; gr < S->cfg->mode_gr; ++gr) {
            if (use_gr[gr] > LH_MAX_BITS_PER_GRANULE)
                ok = 0;
            for (ch = 0; ch < nch; ++ch)
                if (use_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                    ok = 0;
        }
        if (ok)
            return use_fr;
    }
    /* too many bits: fix a budget per granule and channel */
    ok = 1;
    sum_fr = 0;
    for (gr = 0; gr < S->cfg->mode_gr; ++gr) {
        max_gr[gr] = 0;
        for (ch = 0; ch < nch; ++ch) {
            max_ch[gr][ch] = (use_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL) ? LH_MAX_BITS_PER_CHANNEL : use_ch[gr][ch];
            max_gr[gr] += max_ch[gr][ch];
        }
        if (max_gr[gr] > LH_MAX_BITS_PER_GRANULE) {
            float   f[2] = { 0.0f, 0.0f }, s = 0.0f;
            for (ch = 0; ch < nch; ++ch) {
                if (max_ch[gr][ch] > 0) {
                    f[ch] = sqrt(sqrt(max_ch[gr][ch]));
                    s += f[ch];
                }
                else
                    f[ch] = 0;
            }
            for (ch = 0; ch < nch; ++ch)
                max_ch[gr][ch] = (s > 0) ? (int) (LH_MAX_BITS_PER_GRANULE * f[ch] / s) : 0;
            if (nch > 1) {
                vbr_share(max_ch[gr], use_ch[gr], 32);
                for (ch = 0; ch < nch; ++ch)
                    if (max_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                        max_ch[gr][ch] = LH_MAX_BITS_PER_CHANNEL;
            }
            max_gr[gr] = 0;
            for (ch = 0; ch < nch; ++ch)
                max_gr[gr] += max_ch[gr][ch];
        }
        sum_fr += max_gr[gr];
    }
    if (sum_fr > max_fr) {
        {
            float   f[2] = { 0.0f, 0.0f }, s = 0.0f;
            for (gr = 0; gr < S->cfg->mode_gr; ++gr) {
                if (max_gr[gr] > 0) {
                    f[gr] = sqrt(max_gr[gr]);
                    s += f[gr];
                }
                else
                    f[gr] = 0;
            }
            for (gr = 0; gr < S->cfg->mode_gr; ++gr)
                max_gr[gr] = (s > 0) ? (int) (max_fr * f[gr] / s) : 0;
        }
        if (S->cfg->mode_gr > 1) {      /* reference vbrquantize.c:1452-1468 */
            vbr_share(max_gr, use_gr, 125);
            for (gr = 0; gr < S->cfg->mode_gr; ++gr)
                if (max_gr[gr] > LH_MAX_BITS_PER_GRANULE)
                    max_gr[gr] = LH_MAX_BITS_PER_GRANULE;
        }
        for (gr = 0; gr < S->cfg->mode_gr; ++gr) {
            float   f[2] = { 0.0f, 0.0f }, s = 0.0f;
            for (ch = 0; ch < nch; ++ch) {
                if (max_ch[gr][ch] > 0) {
                    f[ch] = sqrt(max_ch[gr][ch]);
                    s += f[ch];
                }
                else
                    f[ch] = 0;
            }
            for (ch = 0; ch < nch; ++ch)
                max_ch[gr][ch] = (s > 0) ? (int) (max_gr[gr] * f[ch] / s) : 0;
            if (nch > 1) {
                vbr_share(max_ch[gr], use_ch[gr], 32);
                for (ch = 0; ch < nch; ++ch)
                    if (max_ch[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                        max_ch[gr][ch] = LH_MAX_BITS_PER_CHANNEL;
            }
        }
    }
    sum_fr = 0;
    for (gr = 0; gr < S->cfg->mode_gr; ++gr) {
        int     sum_gr = 0;
        for (ch = 0; ch < nch; ++ch) {
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
    if (!ok)                    /* fall back to the on_pe split */
        for (gr = 0; gr < S->cfg->mode_gr; ++gr)
            for (ch = 0; ch < nch; ++ch)
                max_ch[gr][ch] = max_bits[gr][ch];
    /* best_scalefac_store ran already: undo its bookkeeping before the second pass */
    for (ch = 0; ch < nch; ++ch)
        S->scfsi[ch][0] = S->scfsi[ch][1] = S->scfsi[ch][2] = S->scfsi[ch][3] = 0;
    for (gr = 0; gr < S->cfg->mode_gr; ++gr)
        for (ch = 0; ch < nch; ++ch)
            S->tt[gr][ch].scalefac_compress = 0;
    use_fr = 0;
    for (gr = 0; gr < S->cfg->mode_gr; ++gr) {
        use_gr[gr] = 0;
        for (ch = 0; ch < nch; ++ch) {
            OrcVbrGr *v = &V[gr][ch];
            if (max_bits[gr][ch] > 0) {
                int     i;
                int const cut = v->gi->global_gain;     /* cutDistribution, reference vbrquantize.c:1091-1098 */
                for (i = 0; i < LH_SFBMAX; ++i)
                    if (v->sfwork[i] > cut)
                        v->sfwork[i] = cut;
                vbr_fit(v, v->sfwork, max_ch[gr][ch]);
            }
            use_ch[gr][ch] = vbr_reduce_bits(S, gr, ch);
            use_gr[gr] += use_ch[gr][ch];
        }
        use_fr += use_gr[gr];
    }
    return use_fr;              /* the reference aborts if this exceeds max_fr */
}

/* reference quantize.c:1582-1646 (VBR_new_prepare) + 1650-1751 (VBR_new_iteration_loop) */
void
orc_vbr_new_iteration_loop(OrcStream * S, float pe[2][2], const OrcRatio ratio[2][2])
{
    const LhConfig *cfg = S->cfg;
    static float xr34[2][2][576];
    float   xmin[2][2][LH_SFBMAX];
    int     frame_bits[16];
    int     max_bits[2][2];
    int     gr, ch, i, j, analog_silence = 1, avg, bits = 0, pad, used, top_bits, mean_bits;

    memset(xr34, 0, sizeof(xr34));
    /* prepare: budget per granule/channel at the highest allowed bitrate */
    S->bitrate_index = cfg->vbr_max_bitrate_index;
    (void) ResvFrameBegin(S, &avg);
    pad = S->ResvMax;
    for (i = 1; i <= cfg->vbr_max_bitrate_index; i++) {  /* get_framebits, reference quantize.c:1340-1362 */
        S->bitrate_index = i;   /* ends on the maximum again, and so does ResvMax */
        frame_bits[i] = ResvFrameBegin(S, &mean_bits);
    }
    top_bits = frame_bits[cfg->vbr_max_bitrate_index];
    for (gr = 0; gr < S->cfg->mode_gr; gr++) {
        (void) on_pe(S, pe, max_bits[gr], avg, gr, 0);
        if (S->mode_ext == LH_MPG_MD_MS_LR) {
            for (i = 0; i < 576; ++i) {         /* ms_convert, reference quantize.c:48-59 */
                float   l = S->tt[gr][0].xr[i];
                float   r = S->tt[gr][1].xr[i];
                S->tt[gr][0].xr[i] = (l + r) * (float) (ORC_SQRT2 * 0.5);
                S->tt[gr][1].xr[i] = (l - r) * (float) (ORC_SQRT2 * 0.5);
            }
        }
        for (ch = 0; ch < cfg->channels; ++ch) {
            OrcGr  *gi = &S->tt[gr][ch];
            S->masking_lower = cfg->masking_lower_long; /* pow(10, mask_adjust * 0.1) for every block type */
            init_outer_loop(S, gi);
            if (0 != calc_xmin(S, &ratio[gr][ch], gi, xmin[gr][ch]))
                analog_silence = 0;
            bits += max_bits[gr][ch];
        }
    }
    for (gr = 0; gr < S->cfg->mode_gr; gr++)
        for (ch = 0; ch < cfg->channels; ch++)
            if (bits > top_bits && bits > 0) {
                max_bits[gr][ch] *= top_bits;
                max_bits[gr][ch] /= bits;
            }
    if (analog_silence)
        pad = 0;

    for (gr = 0; gr < S->cfg->mode_gr; gr++)
        for (ch = 0; ch < cfg->channels; ch++)
            if (0 == init_xrpow(S, &S->tt[gr][ch], xr34[gr][ch]))
                max_bits[gr][ch] = 0;   /* silent granule needs no bits */

    used = vbr_encode_frame(S, xr34, xmin, max_bits);

    /* smallest frame that holds the bits, stretched while the reservoir cannot take the rest */
    i = (analog_silence && !cfg->enforce_min_bitrate) ? 1 : cfg->vbr_min_bitrate_index;
    for (; i < cfg->vbr_max_bitrate_index; i++)
        if (used <= frame_bits[i])
            break;
    if (i > cfg->vbr_max_bitrate_index)
        i = cfg->vbr_max_bitrate_index;
    if (pad > 0) {
        for (j = cfg->vbr_max_bitrate_index; j > i; --j)
            if (frame_bits[j] - used <= pad)
                break;
        S->bitrate_index = j;
    }
    else
        S->bitrate_index = i;
    (void) ResvFrameBegin(S, &mean_bits);
    for (gr = 0; gr < S->cfg->mode_gr; gr++)
        for (ch = 0; ch < cfg->channels; ch++)
            S->ResvSize -= S->tt[gr][ch].part2_3_length + S->tt[gr][ch].part2_length;  /* ResvAdjust */
    ResvFrameEnd(S, mean_bits);
}
