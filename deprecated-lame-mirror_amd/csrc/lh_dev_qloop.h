/*
 * lh_dev_qloop.h -- the CBR / ABR noise-shaping search (reference quantize.c:367-429,
 * 540-1197; takehiro.c:113-801; quantize_pvt.c:750-913) with the granule held in
 * registers.
 *
 * Why this shape: one wavefront issues at most one instruction every ~5 cycles on
 * gfx950 whatever the instruction is (tools/ubench/lat2.hip), and a stream offers
 * only its two channel-waves, so the length of the serial search is
 * (instructions per iteration) x 5 cycles + stalls.  The search therefore keeps
 * everything it touches every iteration in registers --
 *   lines:  lane l owns the pairs l, l+64, ... (5 slots; |xr|, xrpow, the working
 *           and the best quantised image as packed 16-bit pairs, the band of each
 *           pair),
 *   bands:  lane s owns scalefactor band s (working / best scalefactor, 1/xmin,
 *           distortion, the calc_noise cache),
 * and goes to LDS only where lanes have to exchange data: the 0/1 quadruples of
 * the count1 region, the per-band step of calc_noise, and the squared errors that
 * the band lanes add up in the reference's order.  Per-band decisions travel as
 * ballot masks.  The quantised image and the scalefactors are written to the LDS
 * image (LhChanLds.ix[0], sf[0]) once, when the search is over, where the
 * out-of-line finishing stages expect them.
 */
#ifndef LH_DEV_QLOOP_H
#define LH_DEV_QLOOP_H

#include "lh_dev_quant.h"


struct LhQS {
    /* lane = pair slots */
    float   xp[10];             /* xrpow */
    uint32_t pw[5];             /* working image: low half = even line */
    int     bnd[5];             /* band of the pair (63: the slot holds no pair) */
    uint32_t vm[5];             /* all ones when the pair lies at or below max_nonzero_coeff, else 0 */
    int     bigq;               /* a quantiser call of this search saw a product above 255: the image may hold values >= 256 (sticky; 0
                                 * since lq_load: calc_noise then skips its own test of the image's high bytes) */
    int     bq[5];              /* bnd, or 31 where vm is 0: no band mask of a long-block granule (22 bands) has that bit, so a
                                 * one-bit field extract at bq selects nothing for a pair that is not there (lq_quantize) */
    float   lmax;               /* the lane's largest xrpow (xrpow_max = the maximum over the lanes) */
    /* lane = band */
    int     sfw, sfbest;        /* scalefactor: working, best */
    float   rxmin;              /* 1 / l3_xmin */
    int     wid, sta, win, pre; /* geometry of the band, pretab */
    int     pnstep;             /* calc_noise_data */
    float   pnnoise, pnlog;
    float   dist;               /* distort[] of the working image */
    int     ph;                 /* pseudohalf */
    /* lane = region (0..2): table_select of the working / best image */
    int     tselw, tselb;
    /* lanes 0..15: ipow20[210 + lane] and IXMAX / ipow20[210 + lane].  ipow20[g] = 2^(-3 (g - 210) / 16) as
     * the host rounded it, and ipow20[210 + 16 a + b] = ldexp(ipow20[210 + b], -3 a) holds for the
     * whole table (checked by power_tables_scale_exactly() in lh_host_init.c: a power of two scales exactly), so
     * these 16 values give 1 / step and the xrpow bound of count_bits for every global_gain */
    float   istepv, thrv;
    int     sbg8;               /* lane = band: 8 * subblock_gain[window] of the working image */
    int     sfbl;               /* lane = long band: its first line (576 from lane 23 on): count_bits' pn_sfb_count1 */
    float   m0, m1, m2, m3;     /* POW20(210 .. 213): the four mantissas of the step table (wave-uniform) */
    /* lane = band: what calc_noise finds for the band while every line of it is quantised to zero -- the sum of the
     * squares of its lines, a constant of the granule (lq_zero_band_noise) */
    float   zk;
    int     nzend;              /* wave-uniform: the lines from here on are zero in the working image (set by every count) */
    /* The usual-case stages (pad = 1, a constant there) keep calc_noise's squared errors band by band, every band on a
     * 32-byte boundary and padded with zeros to a multiple of eight terms: the band lanes then add whole blocks of eight
     * (adding +0.0f changes nothing) and are switched off once per block, not once per pair.  sqi: where the slot's pair
     * goes (byte offset into the scratch; pairs of the bands that are not summed go to a dump), sqb: lane = band, the
     * band's first term. */
    int     pad;
    uint32_t sqi[5];
    int     sqb;
#if defined(LH_TRACE) && !defined(LH_EMU)
    mutable LhTr tr;            /* development aid: cycles per segment of the search (lh_dev_common.h) */
#endif
};
#define LQ_SQ_DUMP 568          /* (floats; the padded bands end at 488 for the widest table: checked in lq_load) */

LH_DEVFN int
lq_bit(uint64_t m, int b)
{
    return (int) ((m >> b) & 1ull);
}

/* step of band `lane' (reference takehiro.c:330-336, quantize_pvt.c:833-836) */
LH_DEVFN int
lq_band_step(const LhQS & S, const LhGrR & g)
{
    int const pre = g.preflag ? S.pre : 0;
    return g.global_gain - ((S.sfw + pre) << (g.scalefac_scale + 1)) - S.sbg8;
}

/* ---- Huffman length grids (layout: LhChanLds in lh_dev_common.h; contents: LhTables.hgrid, built on the
 * host by build_huffman_grids, lh_host_init.c) ---- */

/* per class of a region maximum (0..15: the maximum itself; 16 + bit length of max - 15 for the ESC
 * tables): A = byte offset of the candidate group's (0,0) cell inside LhChanLds,
 * B = first table | second table << 8 | linbits of the first << 16 | of the second << 24 | esc << 31
 * (reference takehiro.c:618-647, huf_tbl_noESC; the two linear searches over linbits) */
LH_DEVFN void
lq_class_tabs(int cls, uint32_t *A, uint32_t *B)
{
    uint32_t const big = (uint32_t) __builtin_offsetof(LhChanLds, hl3_big);
    uint32_t const small = (uint32_t) __builtin_offsetof(LhChanLds, hl3_small);
    if (cls <= 15) {
        int const t1 = (cls == 0) ? 0 : lh_huf_noESC((unsigned) cls);
        uint32_t org;
        switch (t1) {
        case 0: org = small + 4 * LQ_ORG_ZERO; break;
        case 1: org = small + 4 * LQ_ORG_T1; break;
        case 2: org = small + 4 * LQ_ORG_T2; break;
        case 5: org = small + 4 * LQ_ORG_T5; break;
        case 7: org = small + 4 * LQ_ORG_T7; break;
        case 10: org = small + 4 * LQ_ORG_T10; break;
        default: org = big + 1024; break;
        }
        *A = org;
        *B = (uint32_t) t1 | ((uint32_t) (t1 + 1) << 8);
    }
    else {
        int const blen = cls - 16;      /* 1..13 */
        int const t24 = (int) ((0x7777665432100000ull >> (4 * (blen & 15))) & 15u);
        int const t16 = (int) ((0x7777766554432100ull >> (4 * (blen & 15))) & 15u);
        int const choice2 = 24 + t24, choice = 16 + (t16 > t24 ? t16 : t24);
        *A = big;
        *B = (uint32_t) choice | ((uint32_t) choice2 << 8) | (lh_ht_xlen_c(choice) << 16) | (lh_ht_xlen_c(choice2) << 24) | 0x80000000u;
    }
}

/* ldexp by a wave-uniform exponent (exact: the results stay normal) */
LH_DEVFN float
lq_ldexp(float v, int e)
{
#ifdef LH_EMU
    return ldexpf(v, e);
#else
    return __builtin_amdgcn_ldexpf(v, e);
#endif
}

/* one line through the quantiser with the rounding table read from HBM only (the rare path for
 * values >= 256; no LDS / HBM pointer mix, which would make the loads FLAT) */
LH_DEVFN int
lq_quant_line_hbm(const LhTables * T, float istep, float xp)
{
    double  x0 = (double) (istep * xp);
    float   f;
    int     k;
    x0 += LH_MAGIC_FLOAT;
    f = (float) x0;
    k = (int) lh_f32_as_u32(f) - LH_MAGIC_INT;
    f = (float) (x0 + T->adj43asm[k]);
    return (int) lh_f32_as_u32(f) - LH_MAGIC_INT;
}

/* load the granule into registers; Q.xrpow, Q.l3_xmin and the geometry arrays were written by
 * lh_init_outer_loop / lh_init_xrpow / lh_calc_xmin.  Then the arrays the search does not need in
 * LDS make room for the Huffman length grids. */
LH_DEVFN void
lq_load(const LhCtx & c, LhQS & S, LhChanLds & Q, const LhQR & R, const LhGrR & g, int qch)
{
    const LhQTabs *qt = LH_QT;
    int const pm = R.mnc >> 1;
    float   mx = 0.0f;
    LH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 5; k++) {
        int const p = c.lane + 64 * k;
        int const ok = (k < 4 || p < 288);
        int const pc = ok ? p : 287;
        lh_f32x2 const x = ((const lh_f32x2 *) Q.xrpow)[pc];
        int const b = Q.sfb_of_line[2 * pc];
        S.xp[2 * k] = ok ? x.x : 0.0f;
        S.xp[2 * k + 1] = ok ? x.y : 0.0f;
        S.bnd[k] = ok ? b : 63;
        S.vm[k] = (ok && p <= pm) ? 0xffffffffu : 0u;
        S.bq[k] = (ok && p <= pm) ? b : 31;
        S.pw[k] = 0u;
        mx = S.xp[2 * k] > mx ? S.xp[2 * k] : mx;
        mx = S.xp[2 * k + 1] > mx ? S.xp[2 * k + 1] : mx;
    }
    S.lmax = mx;
    {
        int const s = c.lane <= LH_SFBMAX ? c.lane : LH_SFBMAX;
        S.sfw = 0;
        S.sfbest = 0;
        S.rxmin = 1.f / Q.l3_xmin[s];
        S.wid = Q.width[s];
        S.sta = Q.start[s];
        S.win = Q.window[s];
        S.pre = (c.lane < LH_SBMAX_L) ? (int) qt->pretab[c.lane < 22 ? c.lane : 0] : 0;
        S.pnstep = 0;
        S.pnnoise = 0;
        S.pnlog = 0;
        S.dist = 0;
        S.ph = Q.pseudohalf[s];
    }
    S.zk = 0;
    S.bigq = 0;
    S.nzend = 0;                /* (the working image starts all zero) */
    S.tselw = 0;                /* table_select is all zero after lh_init_outer_loop */
    S.tselb = S.tselw;
    S.m0 = lh_uni_f(c.T->pow20[210 + LH_QMAX2]);
    S.m1 = lh_uni_f(c.T->pow20[211 + LH_QMAX2]);
    S.m2 = lh_uni_f(c.T->pow20[212 + LH_QMAX2]);
    S.m3 = lh_uni_f(c.T->pow20[213 + LH_QMAX2]);
    S.istepv = c.T->ipow20[210 + (c.lane & 15)];
    S.thrv = (LH_IXMAX) / S.istepv;
    if (c.lane >= 32)
        S.istepv = (1.0f - 0.4054f) / S.istepv;     /* lanes 32..47: the 0/1 comparator's threshold per mantissa */
    S.sbg8 = 0;                 /* so is subblock_gain */
    S.sfbl = (c.lane < LH_SBMAX_L + 1) ? (int) qt->sfb_l[c.lane < LH_SBMAX_L + 1 ? c.lane : 0] : 576;
    LH_WAVE_SYNC();
    {
        /* the three grids are constants of the launch: 704 words from HBM, issued together */
        const uint32_t *hg = c.T->hgrid;
        uint32_t w[11];
#pragma unroll
        for (int j = 0; j < 11; j++)
            w[j] = hg[c.lane + 64 * j];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            Q.hl3_big[0][c.lane + 64 * j] = w[j];
            Q.hl3_big[1][c.lane + 64 * j] = w[4 + j];
        }
#pragma unroll
        for (int j = 0; j < 3; j++)
            Q.hl3_small[c.lane + 64 * j] = w[8 + j];
    }
    if (S.pad) {
        /* (the scratch is the channel's xrpow copy, read above: all zero first, the squares overwrite their places) */
        uint32_t const w8 = (c.lane < LH_SBPSY_L) ? (uint32_t) ((S.wid + 7) & ~7) : 0u;
        uint32_t const upto = lh_wave_scan_u32(w8);
        S.sqb = (int) (upto - w8);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int const b = S.bnd[k] & 63;
            int const at = (int) lh_shfl_u32((uint32_t) S.sqb, b) + 2 * (c.lane + 64 * k) - (int) lh_shfl_u32((uint32_t) S.sta, b);
            /* (a line above max_nonzero_coeff is no term of any sum: its place keeps the zero it has from here, its square
             * goes to the dump word -- no masking of the squares on the way) */
            S.sqi[k] = 4u * (uint32_t) ((S.bnd[k] < LH_SBPSY_L && S.vm[k]) ? at : LQ_SQ_DUMP);
        }
        LH_WAVE_SYNC();
        for (int i = c.lane; i < 576 / 4; i += 64) {
            lh_f32x4 z;
            z.x = z.y = z.z = z.w = 0.0f;
            ((lh_f32x4 *) Q.xrpow)[i] = z;
        }
    }
    LH_WAVE_SYNC();
}

/* The quantiser half (reference takehiro.c:281-414, quantize_xrpow with its calc_noise_data short cuts): the
 * working image S.pw at g.global_gain and the working scalefactors.  Returns 0 -- nothing changed -- when
 * a line is too large for the step (count_bits answers LARGE_BITS then). */
template < int USE_PREV, int NS > LH_DEVFN int
lq_quantize(const LhCtx & c, LhQS & S, const LhQR & R, const LhGrR & g)
{
    LQ_MARK("cb_begin");
    LQ_T(S, 1);
    const LhTables *T = c.T;
    const LhQTabs *qt = LH_QT;
    int const lane = c.lane;
    float   istep, cmpv = 0.0f;
    int     big_prod;
    {
        int const d = g.global_gain - 210, ga = d >> 4, gb_ = d & 15;
        istep = lq_ldexp(lh_u32_as_f32(lh_bcast_u32(lh_f32_as_u32(S.istepv), gb_)), -3 * ga);
        cmpv = lq_ldexp(lh_u32_as_f32(lh_bcast_u32(lh_f32_as_u32(S.istepv), 32 + gb_)), 3 * ga);
        /* (the lane's largest xrpow bounds its products -- a float product is monotone in either factor -- and a product of
         * at most 255 rounds to at most 255 and lies far below IXMAX_VAL: one multiplication and one comparison decide both
         * that no line is too large for the step and that the rounding offsets in LDS suffice) */
        big_prod = lh_ballot(istep * S.lmax > 255.0f) != 0;
        if (LH_RARE(big_prod)) {
            float const thr = lq_ldexp(lh_u32_as_f32(lh_bcast_u32(lh_f32_as_u32(S.thrv), gb_)), 3 * ga);
            if (lh_ballot(S.lmax > thr))
                return 0;
            S.bigq = 1;
        }
    }
    /* the products, their first rounding and the look-ups of the second one go out BEFORE the band masks are formed
     * (scalar work with branches of its own): the masks then fill the look-ups' LDS round trip instead of preceding it */
    float   qa[10], qt_[10];
    uint32_t qb[10];
#pragma unroll
    for (int k = 0; k < NS; k++) {
        qa[2 * k] = istep * S.xp[2 * k];
        qa[2 * k + 1] = istep * S.xp[2 * k + 1];
        qb[2 * k] = lh_f32_as_u32(qa[2 * k] + (float) LH_MAGIC_FLOAT);
        qb[2 * k + 1] = lh_f32_as_u32(qa[2 * k + 1] + (float) LH_MAGIC_FLOAT);
        qt_[2 * k] = qt->qthr[qb[2 * k] & 255u];
        qt_[2 * k + 1] = qt->qthr[qb[2 * k + 1] & 255u];
    }
    /* ---- which bands are quantised, and how (lane = band) ---- */
    /* (without the previous call's data: every band anew, none by the 0 / 1 comparator; bit 31 stays clear, see LhQS.bq) */
    uint64_t ncmask = 0x7fffffffull, m01mask = 0;
    int     zero_mnc = 0, plain = 1;
    int const pm = R.mnc >> 1;
    if (USE_PREV && (g.global_gain == R.pn_global_gain || R.pn_sfb_count1 > 0)) {
        int const sfbmax = (R.block_type == LH_SHORT_TYPE) ? 38 : 21;
        int const prev_data_use = (g.global_gain == R.pn_global_gain);
        uint64_t const all = (2ull << sfbmax) - 1ull;
        int const s = lane;
        int const step = lq_band_step(S, g);
        int const cached = prev_data_use && (S.pnstep == step);
        /* the band that holds line max_nonzero_coeff never takes the 0/1 comparator (see
         * lh_count_bits in lh_dev_quant.h) */
        int const m01 = R.pn_sfb_count1 > 0 && s >= R.pn_sfb_count1 && S.pnstep > 0 && step >= S.pnstep
            && s != R.s_mnc;
        ncmask = lh_ballot(s <= sfbmax && !cached);
        m01mask = lh_ballot(s <= sfbmax && m01);
        {
            int const cached_m = !lq_bit(ncmask, R.s_mnc);
            int const later = (R.s_mnc < 63) ? ((ncmask >> (R.s_mnc + 1)) != 0) : 0;
            zero_mnc = cached_m && later;
        }
        plain = (ncmask == all) && (m01mask == 0);
    }
    LQ_MARK("cb_quant");
    LQ_T(S, 2);
    /* ---- quantise: all pairs, straight line; the selection follows.  The first rounding is a
     * float addition: (float) ((double) x + 2^23) and x + 2^23f agree for every float x >= 0; the
     * second is a comparison with the first float that the reference's expression rounds up to k
     * (LhTables.qthr; both in tests/test_quantizer_identity.py) ---- */
    {
        uint32_t nq[5];
#if !defined(LH_EMU) && !defined(LH_NO_JOIN)
        /* all of the threshold look-ups behind ONE s_waitcnt (the compiler's own would be one per comparison: eight issue
         * slots of a wave that has none to spare; the look-ups went out before the band masks were formed) */
        if (NS == 4)
            asm volatile("" : "+v"(qt_[0]), "+v"(qt_[1]), "+v"(qt_[2]), "+v"(qt_[3]), "+v"(qt_[4]), "+v"(qt_[5]), "+v"(qt_[6]), "+v"(qt_[7]));
        else
            asm volatile("" : "+v"(qt_[0]), "+v"(qt_[1]), "+v"(qt_[2]), "+v"(qt_[3]), "+v"(qt_[4]), "+v"(qt_[5]), "+v"(qt_[6]), "+v"(qt_[7]),
                         "+v"(qt_[8]), "+v"(qt_[9]));
#endif
#pragma unroll
        for (int k = 0; k < NS; k++) {
            float const a0 = qa[2 * k], a1 = qa[2 * k + 1];
            uint32_t const b0 = qb[2 * k], b1 = qb[2 * k + 1];
            float const t0 = qt_[2 * k], t1 = qt_[2 * k + 1];
            uint32_t const r0 = b0 - (a0 < t0 ? 1u : 0u), r1 = b1 - (a1 < t1 ? 1u : 0u);
#if !defined(LH_EMU) && !defined(LH_NO_BFE_SELECT)
            /* (the two low halves into one word: one v_perm_b32 instead of v_and + v_lshl_or; masked where it is used) */
            nq[k] = __builtin_amdgcn_perm(r1, r0, 0x05040100u);
#else
            nq[k] = ((r0 & 0xffffu) | (r1 << 16)) & S.vm[k];
#endif
        }
        if (LH_RARE(big_prod)) {
            /* rare: a quantised value >= 256, its rounding offset lives in HBM */
#pragma unroll
            for (int k = 0; k < NS; k++) {
                int const q0 = lq_quant_line_hbm(T, istep, S.xp[2 * k]);
                int const q1 = lq_quant_line_hbm(T, istep, S.xp[2 * k + 1]);
                nq[k] = ((uint32_t) (q0 & 0xffff) | ((uint32_t) q1 << 16)) & S.vm[k];
            }
        }
#if !defined(LH_EMU) && !defined(LH_NO_BFE_SELECT)
        /* (with the previous call's data in play, a long-block granule takes the selecting path below every time: "all bands
         * anew, none by the 0 / 1 comparator" is one more pair of masks to it, not a branch of its own) */
        int const merged = USE_PREV && R.block_type != LH_SHORT_TYPE;
#else
        int const merged = 0;
#endif
        if (plain && !merged) {
#pragma unroll
            for (int k = 0; k < NS; k++)
                S.pw[k] = nq[k] & S.vm[k];
        }
        else {
            /* (1 - 0.4054) / istep: istep is one of 16 mantissas times a power of two, and a correctly rounded quotient
             * scales exactly -- the 16 quotients sit in lanes 32..47 of S.istepv */
            float const compareval0 = cmpv;
#if !defined(LH_EMU) && !defined(LH_NO_BFE_SELECT)
            /* long-block granules have 22 bands: the two band masks fit a word each, a signed one-bit field extract at the
             * lane's band (v_bfe_i32) is 0 or ~0 for the whole word, and two v_bfi_b32 take the place of the two selects --
             * four instructions per slot instead of eight (two 64-bit bit tests of three instructions each + the selects) */
            if (R.block_type != LH_SHORT_TYPE) {
                int const ncw = (int) (uint32_t) ncmask, z1w = (int) (uint32_t) m01mask;
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    /* (at bq: a pair above max_nonzero_coeff extracts bit 31 = 0 twice and keeps its zero, so neither
                     * candidate needs the mask) */
                    uint32_t const NC = (uint32_t) __builtin_amdgcn_sbfe(ncw, (unsigned) S.bq[k], 1u);
                    uint32_t const Z1 = (uint32_t) __builtin_amdgcn_sbfe(z1w, (unsigned) S.bq[k], 1u);
                    uint32_t const v01 = lh_vec_u32(((compareval0 > S.xp[2 * k + 1]) ? 0u : 0x10000u)
                                                    | ((compareval0 > S.xp[2 * k]) ? 0u : 1u));
                    uint32_t const t = (Z1 & v01) | (~Z1 & nq[k]);
                    S.pw[k] = (NC & t) | (~NC & S.pw[k]);
                }
            }
            else
#endif
#pragma unroll
            for (int k = 0; k < NS; k++) {
                int const nc = lq_bit(ncmask, S.bnd[k]), z1 = lq_bit(m01mask, S.bnd[k]);
                /* (through lh_vec_u32: formed for every lane, not under an EXEC mask per slot with its branch) */
                uint32_t const v01 = lh_vec_u32((((compareval0 > S.xp[2 * k]) ? 0u : 1u)
                                                 | (((compareval0 > S.xp[2 * k + 1]) ? 0u : 1u) << 16)) & S.vm[k]);
                S.pw[k] = nc ? (z1 ? v01 : (nq[k] & S.vm[k])) : S.pw[k];
            }
            if (LH_RARE(zero_mnc)) {
                /* (rare and wave-uniform: a branch that falls through instead of four selects on every call) */
#pragma unroll
                for (int k = 0; k < NS; k++)
                    if (lane + 64 * k == pm)
                        S.pw[k] &= 0xffffu;
            }
        }
    }
    if (LH_RARE(R.substep_shaping & 2)) {
        int const gain = g.global_gain + g.scalefac_scale;
        float const roundfac = (float) (0.634521682242439 / T->ipow20[gain]);
        uint64_t const phm = lh_ballot(lane < R.sfbmax && S.ph);
#pragma unroll
        for (int k = 0; k < NS; k++) {
            if (lq_bit(phm, S.bnd[k])) {
                uint32_t v = S.pw[k];
                if (!(S.xp[2 * k] >= roundfac))
                    v &= 0xffff0000u;
                if (!(S.xp[2 * k + 1] >= roundfac))
                    v &= 0x0000ffffu;
                S.pw[k] = v;
            }
        }
    }
    return 1;
}

/* The counting half (reference takehiro.c:654-801, noquant_count_bits + count_bits' bookkeeping) on the working
 * image.  Returns the bit count; g's count fields follow except table_select, which lives in S.tselw
 * (lane = region). */
template < int USE_PREV, int NS > LH_DEVFN int
lq_count(const LhCtx & c, LhQS & S, LhQR & R, LhGrR & g, LhChanLds & Q)
{
    const LhQTabs *qt = LH_QT;
    int const lane = c.lane;
    LQ_MARK("cb_count");
    LQ_T(S, 3);
    /* ---- count ---- */
    {
        int     top_nz, top_big, i, bv, nquad, bits;
        int     e0, e1, e2, a1, a2;
        if (USE_PREV)
            R.pn_sfb_count1 = 0;
#if !defined(LH_EMU)
        /* What the count1 region will look up depends on the image alone, not on where the region starts: the clamped words,
         * the neighbour's word and the look-ups themselves go out before the reduction below, whose DPP chain has idle issue
         * slots (two chains leave one wait state per step) and whose result the look-ups' round trip then does not follow. */
        uint32_t cl[5], qlen[5];
        {
            uint32_t const qtab = lh_lds_off(&Q) + (uint32_t) ((const char *) qt->t3233p - (const char *) &Q);
#pragma unroll
            for (int k = 0; k < NS; k++)
                cl[k] = lh_pk_min_u16(S.pw[k], 0x000f000fu);
#pragma unroll
            for (int k = 0; k < NS; k++) {
                uint32_t const nxt = (k + 1 < NS) ? cl[k + 1 < NS ? k + 1 : k] : 0u;
                uint32_t const u1 = lh_lane_above_u32(cl[k], nxt);
                uint32_t const t = cl[k] | (u1 << 2);
                qlen[k] = lh_lds_read_u32(lh_dot2_u16(t, 4u | (8u << 16), qtab));
            }
        }
#endif
        {
            /* highest non-zero pair + 1 and highest pair holding a value > 1 + 1: slots ascend, the
             * last hit of a lane is its highest; then the maximum over the lanes */
            uint32_t t[2] = { 0u, 0u };
#pragma unroll
            for (int k = 0; k < NS; k++) {
                uint32_t const p1 = (uint32_t) (lane + 64 * k + 1);
                t[0] = (S.pw[k] != 0u) ? p1 : t[0];
                t[1] = ((S.pw[k] & 0xfffefffeu) != 0u) ? p1 : t[1];
            }
            lh_wave_max_n < 2 > (t);
            top_nz = (int) t[0];
            top_big = (int) t[1];
        }
        i = 2 * top_nz;
        g.count1 = i;
        S.nzend = i;
        nquad = (i - 2 * top_big) / 4;
        bv = i - 4 * nquad;
        g.big_values = bv;
        if (R.block_type == LH_SHORT_TYPE) {
            a1 = 3 * lh_uni_i((int) qt->sfb_s3);
            a2 = bv;
        }
        else if (R.block_type == LH_NORM_TYPE) {
            if (bv > 0) {
                uint32_t const pack = (uint32_t) lh_uni_i((int) qt->bvpack[(bv >> 1) - 1]);
                g.region0_count = (int) (pack & 15u);
                g.region1_count = (int) ((pack >> 4) & 15u);
                a1 = (int) ((pack >> 8) & 1023u);
                a2 = (int) ((pack >> 18) & 1023u);
            }
            else
                a1 = a2 = 0;
        }
        else {
            if (bv > 0) {
                g.region0_count = 7;
                g.region1_count = LH_SBMAX_L - 1 - 7 - 1;
            }
            a1 = lh_uni_i((int) qt->sfb_l[7 + 1]);
            a2 = bv;
        }
        a1 = (a1 < bv) ? a1 : bv;
        a2 = (a2 < bv) ? a2 : bv;
        e0 = lh_uni_i(a1 >> 1);
        e1 = lh_uni_i(a2 >> 1);
        e2 = (R.block_type == LH_NORM_TYPE) ? (bv >> 1) : e1;
        /* region 1 may reach beyond the second slot: any block type but the normal one (it runs up to big_values there), or
         * band edges other than those of 44.1 / 48 kHz */
        int const wide1 = (R.block_type != LH_NORM_TYPE) || (c.l21 > 418);
        LQ_MARK("cb_quads");
        LQ_T(S, 4);
        LH_WAVE_ORDER();
        uint32_t red[1];        /* the count1 region's lengths with both of its tables, t32 << 16 | t33 */
        uint32_t pre_l, pre_h, qtot;
        /* The quadruples of the count1 region: lines bv + 4 q .. + 3 = the pairs pb + 2 q and pb + 2 q + 1 (pb = bv / 2),
         * which lie in neighbouring lanes of the same slot (or in lane 63 and the next slot's lane 0).  The lane of a
         * quadruple's first pair fetches the second with one wave shift -- no trip through LDS --, forms the index
         * v | w << 1 | x << 2 | y << 3 from the two packed words (all four values are 0 or 1 there) and looks up both
         * tables' lengths; lanes whose pair starts no quadruple of the region look up something and drop it. */
        uint32_t qsum = 0;
        {
            int const pb = bv >> 1;
            uint32_t const first = (uint32_t) (lane - pb);      /* pair - pb of slot 0 */
            /* (a pair at an odd distance from pb starts no quadruple: out of every range) */
            uint32_t const firstq = (first & 1u) ? 0x7ffff000u : first;
            uint32_t const n2 = (uint32_t) (2 * nquad);
#ifndef LH_EMU
            /* On the device the index goes straight into the table's LDS address with one v_dot2_u32_u16: the halves of
             * t are v + 4 x and w + 4 y, and 4 (v + 2 w + 4 x + 8 y) = 4 lo + 8 hi.  The values are clamped to 15
             * first (the clamped words serve the grid look-ups below as well), so that the lanes whose pairs hold
             * larger values -- their result is dropped -- read inside the workgroup's image. */
#pragma unroll
            for (int k = 0; k < NS; k++)
                qsum += ((firstq + 64u * (uint32_t) k) < n2) ? qlen[k] : 0u;
#else
#pragma unroll
            for (int k = 0; k < NS; k++) {
                uint32_t const nxt = (k + 1 < NS) ? S.pw[k + 1 < NS ? k + 1 : k] : 0u;
                uint32_t const u1 = lh_lane_above_u32(S.pw[k], nxt);
                uint32_t const t = S.pw[k] | (u1 << 2);
                uint32_t const j = (t | (t >> 15)) & 15u;
                uint32_t const len = qt->t3233p[j];
                qsum += ((firstq + 64u * (uint32_t) k) < n2) ? len : 0u;
            }
#endif
        }
        LQ_MARK("cb_max");
        uint32_t m[3] = { 0u, 0u, 0u };
        {
            /* (the slots' candidates first, then three-way maxima: v_max3_u32 halves the chain of two-way ones) */
            uint32_t cm[3][5];
#pragma unroll
            for (int k = 0; k < NS; k++) {
                int const p = lane + 64 * k;
                uint32_t const lo = S.pw[k] & 0xffffu, hi = S.pw[k] >> 16;
                uint32_t const mx = lo > hi ? lo : hi;
                /* (region 0 ends within the first slot's 64 pairs whatever the block type, region 1 of a normal block with
                 * the band edges of 44.1 / 48 kHz within the first two: build_region_split, lh_host_init.c, checks the tables
                 * -- the other slots' comparisons, selects and terms are not in the code) */
                int const in0 = (k == 0) && p < e0, in1 = (k <= 1 || wide1) && p < e1, in2 = p < e2;
                cm[0][k] = in0 ? mx : 0u;
                cm[1][k] = (in1 && !in0) ? mx : 0u;
                cm[2][k] = (in2 && !in1) ? mx : 0u;
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
#if !defined(LH_EMU)
                uint32_t a, r;
                if (NS > 4)
                    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(a) : "v"(cm[j][2]), "v"(cm[j][3]), "v"(cm[j][4]));
                else if (NS > 3)
                    asm("v_max_u32_e32 %0, %1, %2" : "=v"(a) : "v"(cm[j][2]), "v"(cm[j][3]));
                else
                    a = (NS > 2) ? cm[j][2] : 0u;
                asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(cm[j][0]), "v"(cm[j][1]), "v"(a));
                m[j] = r;
#else
                uint32_t r = 0;
                for (int k = 0; k < NS; k++)
                    r = cm[j][k] > r ? cm[j][k] : r;
                m[j] = r;
#endif
            }
            lh_wave_max_n < 3 > (m);
        }
        red[0] = qsum;
        LQ_MARK("cb_lookup");
        LQ_T(S, 5);
        /* lane r < 3 works out region r's candidate tables; the grid origins go back to all lanes */
        uint32_t PB, esc;
        uint32_t G0, G1, G2;
        {
            uint32_t const mr = (lane == 0) ? m[0] : (lane == 1) ? m[1] : m[2];
            uint32_t const d = mr - 15u;
            uint32_t const cls = (mr > 15u) ? (uint32_t) (16 + 32 - lh_clz32(d)) : mr;
            uint32_t PA = qt->ctabA[cls & 31u];
            PB = qt->ctabB[cls & 31u];
            esc = PB >> 31;
#ifndef LH_EMU
            PA += lh_lds_off(&Q);       /* the origins as LDS addresses (see the look-ups below) */
#endif
            G0 = lh_bcast_u32(PA, 0);
            G1 = lh_bcast_u32(PA, 1);
            G2 = lh_bcast_u32(PA, 2);
        }
        {
            uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
            const char *qb = (const char *) &Q;
            /* (on the device G0..G2 are LDS addresses: a cell's address is then one v_dot2_u32_u16 --
             * x * 64 + y * 4 + origin -- on the clamped pair) */
#pragma unroll
            for (int k = 0; k < NS; k++) {
                int const p = lane + 64 * k;
                int const in0 = (k == 0) && p < e0, in1 = (k <= 1 || wide1) && p < e1, in2 = p < e2;
#ifndef LH_EMU
                uint32_t const go = in0 ? G0 : (in1 ? G1 : G2);
                uint32_t const v = lh_lds_read_u32(lh_dot2_u16(cl[k], 64u | (4u << 16), go));
#else
                uint32_t const cl = lh_pk_min_u16(S.pw[k], 0x000f000fu);
                uint32_t const off = ((cl << 6) & 0x3c0u) | (cl >> 14);      /* (x * 16 + y) * 4 */
                uint32_t const go = in0 ? G0 : (in1 ? G1 : G2);
                uint32_t const v = *(const uint32_t *) (qb + go + off);
#endif
                acc0 += in0 ? v : 0u;
                acc1 += in1 ? v : 0u;
                acc2 += in2 ? v : 0u;
            }
            /* no code is longer than 21 bits (sign bits included), so the 10-bit fields hold the sums
             * over 8 lanes x 5 pairs; lh_wave_sum_regions() spreads them out on the way */
            pre_l = pre_h = 0;
            qtot = lh_wave_sum_regions(acc0, acc1, acc2, red[0], &pre_l, &pre_h);
        }
        LQ_MARK("cb_decide");
        LQ_T(S, 6);
        {
            int const c1a = (int) (qtot >> 16), c1b = (int) (qtot & 0xffffu);
            bits = c1a;
            g.count1table_select = 0;
            if (c1a > c1b) {
                bits = c1b;
                g.count1table_select = 1;
            }
            g.count1bits = bits;
        }
        {
            /* lane r: choose_table of region r from its sums (reference takehiro.c:618-647).  The sums came
             * as prefixes (p < e0, p < e1, p < e2): a region's own is its prefix minus the lane below's. */
            uint32_t const Lr = pre_l - lh_lane_below_u32(pre_l), Hr = pre_h - lh_lane_below_u32(pre_h);
            uint32_t const x0 = lh_vec_u32((uint32_t) (0 < e0)), x1 = lh_vec_u32((uint32_t) (e0 < e1)), x2 = lh_vec_u32((uint32_t) (e1 < e2));
            int const ex = (int) ((lane == 0) ? x0 : (lane == 1) ? x1 : x2);
            uint32_t const a = Lr & 0xffffu, b = Lr >> 16;
            uint32_t const tA = PB & 31u, tB = (PB >> 8) & 31u, lin1 = (PB >> 16) & 15u, lin2 = (PB >> 24) & 15u;
            uint32_t const a2_ = a + Hr * lin1, b2_ = b + Hr * lin2, c2_ = esc ? 0x7fffffffu : Hr;
            uint32_t best = a2_, t = tA;
            if (a2_ > b2_) {
                best = b2_;
                t = tB;
            }
            if (best > c2_) {
                best = c2_;
                t = tA + 2u;
            }
            if (lane < 3 && ex)
                S.tselw = (int) t;
            best = (lane < 3 && ex) ? best : 0u;
            bits += (int) (lh_bcast_u32(best, 0) + lh_bcast_u32(best, 1) + lh_bcast_u32(best, 2));
        }
        if (USE_PREV && R.block_type == LH_NORM_TYPE && bv != 0)
            R.pn_sfb_count1 = lh_popc64(lh_ballot(S.sfbl < bv));    /* (big_values <= 576: lanes from 23 on never count) */
        LQ_MARK("cb_end");
        LQ_T(S, 7);
        return bits;
    }
}

template < int NS > LH_DEVFN void lq_noise_squares(const LhCtx & c, const LhQS & S, const LhGrR & g, LhChanLds & Q, const float *xr);

/* reference takehiro.c:768-801, count_bits */
template < int USE_PREV, int NS > LH_DEVFN int
lq_count_bits(const LhCtx & c, LhQS & S, LhQR & R, LhGrR & g, LhChanLds & Q, const float *xr = nullptr)
{
    int     bits = LH_LARGE_BITS;
    LH_PC(10);
    LH_PT(t_cb);
    if (lq_quantize < USE_PREV, NS > (c, S, R, g)) {
#if defined(LH_NOISE_EARLY) && !defined(LH_EMU)
        /* (experiment of round 6, DESIGN.md section 4: calc_noise's first half beside the count's reductions and look-ups,
         * in one instruction stream; wasted when the candidate does not fit and the gain rises) */
        if (USE_PREV)
            lq_noise_squares < NS > (c, S, g, Q, xr);
#endif
        bits = lq_count < USE_PREV, NS > (c, S, R, g, Q);
    }
    LH_PA(11, t_cb);
    return bits;
}

/* Lane = band: the sum of the band's n squares (sq + 2 jj on), added in the reference's order; maxw = the largest n of
 * the wave (wave-uniform).  Shared by calc_noise and by the constants of all-zero bands (lq_zero_band_noise). */
LH_DEVFN float
lq_band_sums(const float *sq, int n, int jj, int maxw, int fresh)
{
    float   noise = 0;
    {
        /* Lane = band adds its squares in the reference's order: a serial float sum, so the wave runs as many
         * steps as the widest band that changed has lines.  Every lane reads on past its own band's end (the next
         * band's squares, then whatever follows in the channel's LDS image: at most 158 lines beyond a band's start)
         * and keeps the value its sum had after its own last line -- a compare and a select per pair instead of an
         * address select per load.  Eight terms per trip from two 16-byte reads (band starts are even, the array is
         * 16-byte aligned: a band that starts on an odd pair reads its first pair alone). */
        const lh_f32x2 *sq2 = (const lh_f32x2 *) sq;
#if !defined(LH_EMU)
        /* On the device the lanes are switched off as their bands end: v_cmpx narrows EXEC after every pair of
         * terms and a lane that is off keeps its sum -- three instructions per pair of terms instead of five, no copy
         * to keep.  Sixteen terms per block from four 16-byte reads (ds_read2_b64: a band may start on an odd pair,
         * the reads need 8-byte alignment only); the reads of the next block are issued before the adds of this one.
         * EXEC is restored at the end of each block; what is read beyond a band's end is not added. */
        float   kept;
        {
            struct alignas(8) Q4 { float x, y, z, w; };
            const Q4 *src = (const Q4 *) (sq2 + jj);
            int     rem = n;            /* terms the lane still has to add (<= 0: none) */
            Q4      a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
#define LH_CN_PAIR(K, A, B) "v_cmpx_gt_i32_e64 %[tm], %[rem], " #K "\n\tv_add_f32 %[ns], %[ns], %[" #A "]\n\tv_add_f32 %[ns], %[ns], %[" #B "]\n\t"
#define LH_CN_BLOCK(U, V, W, X) do { unsigned long long sv_, tm_; \
                asm volatile("s_mov_b64 %[sv], exec\n\t" \
                             LH_CN_PAIR(0, a0, a1) LH_CN_PAIR(2, a2, a3) LH_CN_PAIR(4, b0, b1) LH_CN_PAIR(6, b2, b3) \
                             LH_CN_PAIR(8, c0, c1) LH_CN_PAIR(10, c2, c3) LH_CN_PAIR(12, d0, d1) LH_CN_PAIR(14, d2, d3) \
                             "s_mov_b64 exec, %[sv]" \
                             : [ns] "+v"(noise), [sv] "=&s"(sv_), [tm] "=&s"(tm_) \
                             : [rem] "v"(rem), [a0] "v"(U.x), [a1] "v"(U.y), [a2] "v"(U.z), [a3] "v"(U.w), \
                               [b0] "v"(V.x), [b1] "v"(V.y), [b2] "v"(V.z), [b3] "v"(V.w), \
                               [c0] "v"(W.x), [c1] "v"(W.y), [c2] "v"(W.z), [c3] "v"(W.w), \
                               [d0] "v"(X.x), [d1] "v"(X.y), [d2] "v"(X.z), [d3] "v"(X.w)); } while (0)
            for (int k0 = 0; k0 < maxw; k0 += 32) {
                Q4 const b0 = src[4], b1 = src[5], b2 = src[6], b3 = src[7];
                LH_CN_BLOCK(a0, a1, a2, a3);
                rem -= 16;
                if (k0 + 16 >= maxw)
                    break;
                a0 = src[8];
                a1 = src[9];
                a2 = src[10];
                a3 = src[11];
                LH_CN_BLOCK(b0, b1, b2, b3);
                rem -= 16;
                src += 8;
            }
#undef LH_CN_BLOCK
#undef LH_CN_PAIR
            kept = noise;
        }
#else
        float   kept = 0.0f;
        int     at = jj, done = 0;
        if (lh_ballot(fresh && (jj & 1))) {
            /* odd starting pair: one pair first, so that the rest is aligned */
            lh_f32x2 const t = sq2[at];
            int const odd = jj & 1;
            float const a = noise + t.x, b = a + t.y;
            noise = odd ? b : noise;
            at += odd;
            done += 2 * odd;
            kept = (done == n) ? noise : kept;
        }
        for (int k0 = 0; k0 < maxw; k0 += 8) {
            lh_f32x4 const u = *(const lh_f32x4 *) &sq2[at], v = *(const lh_f32x4 *) &sq2[at + 2];
            noise += u.x;
            noise += u.y;
            kept = (done + 2 == n) ? noise : kept;
            noise += u.z;
            noise += u.w;
            kept = (done + 4 == n) ? noise : kept;
            noise += v.x;
            noise += v.y;
            kept = (done + 6 == n) ? noise : kept;
            noise += v.z;
            noise += v.w;
            kept = (done + 8 == n) ? noise : kept;
            at += 4;
            done += 8;
        }
#endif
        noise = kept;
    }
    return noise;
}

/* The same over the padded layout (LhQS.pad): the band's terms start at sq[base] (a 32-byte boundary), zeros follow its last
 * term up to a multiple of eight.  Sixteen terms per block from four 16-byte reads; a lane is switched off for a whole
 * group of eight once its band has ended; the next block's reads are issued before this block's additions. */
LH_DEVFN float
lq_band_sums_pad(const float *sq, int base, int n, int maxw)
{
    float   noise = 0;
#if !defined(LH_EMU) && !defined(LH_SUMS_C)
    /* One instruction stream for the whole walk: blocks of sixteen terms (four 16-byte reads) in two register sets, the next
     * block's reads in flight under this block's additions; EXEC only ever narrows -- a lane leaves when its band's terms
     * are through (checked per octet: the zero padding runs up to a multiple of eight) and the loop ends with the last lane,
     * so neither the wave's largest width nor a scalar loop count is needed; EXEC comes back at the end, behind a wait for
     * the reads still in flight (the temporaries are plain clobbers, all of them registers a callee need not preserve: v[192:199], v[208:215], v[224:231], v[240:247] -- the others of that range would be saved to scratch at the stage's entry, 4 KB per call that the L2 writes back to HBM). */
    (void) maxw;
    int     rem = n;            /* terms the lane still has to add (<= 0: none) */
    uint32_t ad = lh_lds_off(sq) + 4u * (uint32_t) base;
    unsigned long long sv_, tm_;
#define LQ_OCT(R0, R1, R2, R3, R4, R5, R6, R7) \
        "v_add_f32 %[ns], %[ns], v" #R0 "\n\tv_add_f32 %[ns], %[ns], v" #R1 "\n\tv_add_f32 %[ns], %[ns], v" #R2 "\n\t" \
        "v_add_f32 %[ns], %[ns], v" #R3 "\n\tv_add_f32 %[ns], %[ns], v" #R4 "\n\tv_add_f32 %[ns], %[ns], v" #R5 "\n\t" \
        "v_add_f32 %[ns], %[ns], v" #R6 "\n\tv_add_f32 %[ns], %[ns], v" #R7 "\n\t"
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_gt_i32_e64 %[tm], %[rem], 0\n\t"
                 "s_cbranch_execz .Llq_sum_done_%=\n\t"
                 "ds_read_b128 v[192:195], %[ad]\n\t"
                 "ds_read_b128 v[196:199], %[ad] offset:16\n\t"
                 "ds_read_b128 v[208:211], %[ad] offset:32\n\t"
                 "ds_read_b128 v[212:215], %[ad] offset:48\n\t"
                 ".Llq_sum_loop_%=:\n\t"
                 "ds_read_b128 v[224:227], %[ad] offset:64\n\t"
                 "ds_read_b128 v[228:231], %[ad] offset:80\n\t"
                 "ds_read_b128 v[240:243], %[ad] offset:96\n\t"
                 "ds_read_b128 v[244:247], %[ad] offset:112\n\t"
                 "s_waitcnt lgkmcnt(4)\n\t"
                 LQ_OCT(192, 193, 194, 195, 196, 197, 198, 199)
                 "v_cmpx_gt_i32_e64 %[tm], %[rem], 8\n\t"
                 LQ_OCT(208, 209, 210, 211, 212, 213, 214, 215)
                 "v_cmpx_gt_i32_e64 %[tm], %[rem], 16\n\t"
                 "s_cbranch_execz .Llq_sum_done_%=\n\t"
                 "ds_read_b128 v[192:195], %[ad] offset:128\n\t"
                 "ds_read_b128 v[196:199], %[ad] offset:144\n\t"
                 "ds_read_b128 v[208:211], %[ad] offset:160\n\t"
                 "ds_read_b128 v[212:215], %[ad] offset:176\n\t"
                 "s_waitcnt lgkmcnt(4)\n\t"
                 LQ_OCT(224, 225, 226, 227, 228, 229, 230, 231)
                 "v_cmpx_gt_i32_e64 %[tm], %[rem], 24\n\t"
                 LQ_OCT(240, 241, 242, 243, 244, 245, 246, 247)
                 "v_add_u32_e32 %[rem], -32, %[rem]\n\t"
                 "v_add_u32_e32 %[ad], 0x80, %[ad]\n\t"
                 "v_cmpx_gt_i32_e64 %[tm], %[rem], 0\n\t"
                 "s_cbranch_execnz .Llq_sum_loop_%=\n\t"
                 ".Llq_sum_done_%=:\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [ns] "+v"(noise), [rem] "+v"(rem), [ad] "+v"(ad), [sv] "=&s"(sv_), [tm] "=&s"(tm_)
                 :
                 : "memory", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v208", "v209", "v210", "v211",
                   "v212", "v213", "v214", "v215", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231",
                   "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
#undef LQ_OCT
#elif !defined(LH_EMU)
    struct alignas(16) Q4 { float x, y, z, w; };
    const Q4 *src = (const Q4 *) (sq + base);
    int     rem = n;            /* terms the lane still has to add (<= 0: none) */
    Q4      a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
#define LH_CP_OCT(K, A, B) "v_cmpx_gt_i32_e64 %[tm], %[rem], " #K "\n\t" \
                           "v_add_f32 %[ns], %[ns], %[" #A "0]\n\tv_add_f32 %[ns], %[ns], %[" #A "1]\n\t" \
                           "v_add_f32 %[ns], %[ns], %[" #A "2]\n\tv_add_f32 %[ns], %[ns], %[" #A "3]\n\t" \
                           "v_add_f32 %[ns], %[ns], %[" #B "0]\n\tv_add_f32 %[ns], %[ns], %[" #B "1]\n\t" \
                           "v_add_f32 %[ns], %[ns], %[" #B "2]\n\tv_add_f32 %[ns], %[ns], %[" #B "3]\n\t"
#define LH_CP_BLOCK(U, V, W, X) do { unsigned long long sv_, tm_; \
                asm volatile("s_mov_b64 %[sv], exec\n\t" LH_CP_OCT(0, a, b) LH_CP_OCT(8, c, d) "s_mov_b64 exec, %[sv]" \
                             : [ns] "+v"(noise), [sv] "=&s"(sv_), [tm] "=&s"(tm_) \
                             : [rem] "v"(rem), [a0] "v"(U.x), [a1] "v"(U.y), [a2] "v"(U.z), [a3] "v"(U.w), \
                               [b0] "v"(V.x), [b1] "v"(V.y), [b2] "v"(V.z), [b3] "v"(V.w), \
                               [c0] "v"(W.x), [c1] "v"(W.y), [c2] "v"(W.z), [c3] "v"(W.w), \
                               [d0] "v"(X.x), [d1] "v"(X.y), [d2] "v"(X.z), [d3] "v"(X.w)); } while (0)
    for (int k0 = 0; k0 < maxw; k0 += 32) {
        Q4 const b0 = src[4], b1 = src[5], b2 = src[6], b3 = src[7];
        LH_CP_BLOCK(a0, a1, a2, a3);
        rem -= 16;
        if (k0 + 16 >= maxw)
            break;
        a0 = src[8];
        a1 = src[9];
        a2 = src[10];
        a3 = src[11];
        LH_CP_BLOCK(b0, b1, b2, b3);
        rem -= 16;
        src += 8;
    }
#undef LH_CP_BLOCK
#undef LH_CP_OCT
#else
    (void) maxw;
    for (int i = 0; i < n; i++)
        noise += sq[base + i];
#endif
    return noise;
}

/* The noise of every band for the case that all of its lines are quantised to zero (reference quantize_pvt.c:750-790 with
 * ix = 0: temp = |xr| - pow43[0] * step = |xr|, whatever the step): the squares of the band's lines added in order, over
 * as many lines as calc_noise looks at (max_nonzero_coeff cuts the last band short).  Once per search, after lq_load. */
template < int NS > LH_DEVFN void
lq_zero_band_noise(const LhCtx & c, LhQS & S, const LhQR & R, LhChanLds & Q, const float *xr)
{
    float  *sq = Q.xrpow;
    int const s = c.lane;
    int const mine = (s < R.psymax);
    int     l = 0, j = 0, maxw;
    if (mine) {
        l = S.wid >> 1;
        j = S.sta;
        if ((j + S.wid) > R.mnc) {
            int const usefullsize = R.mnc - j + 1;
            l = (usefullsize > 0) ? (usefullsize >> 1) : 0;
        }
    }
    LH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < NS; k++) {
        int const p = c.lane + 64 * k;
        lh_f32x2 const ax = ((const lh_f32x2 *) xr)[(k < 4 || c.lane < 32) ? c.lane + 64 * k : 287];
        float const t0 = lh_fabsf(ax.x), t1 = lh_fabsf(ax.y);
        lh_f32x2 v;
        v.x = t0 * t0;
        v.y = t1 * t1;
        if (S.pad)
            *(lh_f32x2 *) ((char *) sq + S.sqi[k]) = v;
        else if (k < 4 || p < 288)
            ((lh_f32x2 *) sq)[p] = v;
    }
    LH_WAVE_ORDER();
    maxw = (int) lh_wave_max_u32(mine ? (unsigned) (2 * l) : 0u);
    S.zk = S.pad ? lq_band_sums_pad(sq, S.sqb, mine ? 2 * l : 0, maxw) : lq_band_sums(sq, 2 * l, (j < 576) ? (j >> 1) : 0, maxw, mine);
    LH_WAVE_SYNC();
}

/* What calc_noise finds for the working image, before it is taken over: the band's entry of calc_noise_data
 * and its distortion (lane = band), and the totals.  The search forms it for every candidate the bit count
 * looks at and takes over the one the reference would have computed. */
struct LhNoiseTmp {
    int     pnstep;
    float   pnnoise, pnlog, dist;
    LhNoiseRes res;
};

/* POW20(st) = 2^((st - 210) / 4) for the band step of lane = band: one of four mantissas (the table's own entries for
 * 210..213, in scalar registers) times a power of two -- pow20[i + 4] = 2 pow20[i] holds for the whole table
 * (power_tables_scale_exactly(), lh_host_init.c), so no table look-up is needed */
LH_DEVFN float
lq_pow20_of_step(const LhQS & S, int st)
{
    int const d = st - 210, q = d >> 2, r = d & 3;
    /* (through readfirstlane: a select between plain loads of S's fields would become one load
     * from a selected address and pin all of S in scratch memory, see lh_sbg()) */
    float const m0 = lh_uni_f(S.m0), m1 = lh_uni_f(S.m1), m2 = lh_uni_f(S.m2), m3 = lh_uni_f(S.m3);
    float const m = (r & 2) ? ((r & 1) ? m3 : m2) : ((r & 1) ? m1 : m0);
    return lq_ldexp(m, q);
}

/* calc_noise's first half: the squared error of every line of the working image at the working steps, to LDS (sq = the
 * channel's dead xrpow copy), where the band lanes add them up.  A function of the image, the scalefactors and the gain
 * only -- with LH_NOISE_EARLY the bit count that produced the image issues it beside its own reductions and look-ups. */
template < int NS > LH_DEVFN void
lq_noise_squares(const LhCtx & c, const LhQS & S, const LhGrR & g, LhChanLds & Q, const float *xr)
{
    const LhTables *T = c.T;
    const LhQTabs *qt = LH_QT;
    float  *sq = Q.xrpow;       /* the LDS copy of xrpow is dead while the search runs */
    float const step = lq_pow20_of_step(S, lq_band_step(S, g));
    int     big = 0;
    /* the band's step goes to its lines with one cross-lane read per slot; the squared errors
     * of all lines go to LDS, where the band lanes add them up in the reference's order */
    float   stp[5], p43[10];
    lh_f32x2 ax[5];
#pragma unroll
    for (int k = 0; k < NS; k++) {
        unsigned const q0 = S.pw[k] & 0xffffu, q1 = S.pw[k] >> 16;
        ax[k] = ((const lh_f32x2 *) xr)[(k < 4 || c.lane < 32) ? c.lane + 64 * k : 287];
        stp[k] = lh_shfl_f32(step, S.bnd[k]);
        p43[2 * k] = qt->pow43h[q0 & 255u];
        p43[2 * k + 1] = qt->pow43h[q1 & 255u];
        big |= (int) ((q0 | q1) >> 8);
    }
    /* (values of 256 and more exist only when a quantiser call of this search said so: LhQS.bigq) */
    if (LH_RARE(S.bigq) && lh_ballot(big != 0)) {
#pragma unroll
        for (int k = 0; k < NS; k++) {
            unsigned const q0 = S.pw[k] & 0xffffu, q1 = S.pw[k] >> 16;
            if (q0 >= 256u)
                p43[2 * k] = T->pow43[q0];
            if (q1 >= 256u)
                p43[2 * k + 1] = T->pow43[q1];
        }
    }
#pragma unroll
    for (int k = 0; k < NS; k++) {
        int const p = c.lane + 64 * k;
        float const t0 = lh_fabsf(ax[k].x) - p43[2 * k] * stp[k];
        float const t1 = lh_fabsf(ax[k].y) - p43[2 * k + 1] * stp[k];
        lh_f32x2 v;
        v.x = t0 * t0;
        v.y = t1 * t1;
        if (S.pad)
            *(lh_f32x2 *) ((char *) sq + S.sqi[k]) = v;
        else if (k < 4 || p < 288)
            ((lh_f32x2 *) sq)[p] = v;
    }
}

/* reference quantize_pvt.c:750-913 on the working image: into `t', S and R stay as they are.  HAVE_SQ: the squared
 * errors are in LDS already (lq_noise_squares ran with the bit count of this very image). */
template < int NS, int HAVE_SQ = 0 > LH_DEVFN void
lq_calc_noise(const LhCtx & c, const LhQS & S, const LhQR & R, const LhGrR & g, LhChanLds & Q, const float *xr,
              LhNoiseTmp & t)
{
    LhNoiseRes & res = t.res;
    int const s = c.lane;
    float  *sq = Q.xrpow;       /* the LDS copy of xrpow is dead while the search runs */
    float   noise = 0, noise_s = 0;
    int     l = 0, j = 0, maxw;
    LH_PC(13);
    LQ_MARK("cn_begin");
    LQ_T(S, 10);
    int const st = lq_band_step(S, g);
    int const fresh = (s < R.psymax) && !(S.pnstep == st);
    int const zb = fresh && (S.sta >= S.nzend);
    if (fresh) {
        l = S.wid >> 1;
        j = S.sta;
        if ((j + S.wid) > R.mnc) {
            int const usefullsize = R.mnc - j + 1;
            l = (usefullsize > 0) ? (usefullsize >> 1) : 0;
        }
    }
    if (!HAVE_SQ)
        lq_noise_squares < NS > (c, S, g, Q, xr);
    /* a band that starts at or above the end of the non-zero lines adds up the squares of its own lines, whatever
     * the step: the constant is there since lq_zero_band_noise, and the band sits out the serial sum below (the
     * widest bands are at the top of the spectrum, where little is quantised to anything else) */
    l = zb ? 0 : l;
    maxw = (int) lh_wave_max_u32(fresh ? (unsigned) (2 * l) : 0u);
    LH_WAVE_ORDER();
    LQ_MARK("cn_sum");
    LQ_T(S, 11);
    {
        int const n = 2 * l;
        int const jj = (j < 576) ? (j >> 1) : 0;
        noise = S.pad ? lq_band_sums_pad(sq, S.sqb, n, maxw) : lq_band_sums(sq, n, jj, maxw, fresh);
        noise = zb ? S.zk : noise;
    }
    LQ_MARK("cn_log");
    LQ_T(S, 12);
    t.pnstep = S.pnstep;
    t.pnnoise = S.pnnoise;
    t.pnlog = S.pnlog;
    t.dist = S.dist;
    if (s < R.psymax) {
        float   distort_;
        if (!fresh) {
            distort_ = S.rxmin * S.pnnoise;
            noise = S.pnlog;
        }
        else {
            t.pnstep = st;
            t.pnnoise = noise;
            distort_ = S.rxmin * noise;
            float   l2;
            LH_FAST_LOG2_VIA(LH_LOGT_LDS, (distort_ > 1E-20f) ? distort_ : 1E-20f, l2);
            noise = (float) (l2 * LH_LOG2_OVER_LOG10);
            t.pnlog = noise;
        }
        t.dist = distort_;
        noise_s = noise;
    }
    LQ_MARK("cn_agg");
    LQ_T(S, 13);
    {
        int const mine = (s < R.psymax);
        int     tmp = 0;
        if (mine && noise_s > 0.0f) {
            tmp = (int) (noise_s * 10 + .5);
            if (tmp < 1)
                tmp = 1;
        }
        res.over_count = lh_popc64(lh_ballot(mine && noise_s > 0.0f));
        lh_wave_sum_maxf((unsigned) (tmp * tmp), mine ? noise_s : -20.0f, &res.over_SSD, &res.max_noise);
        res.tot_noise = 0;      /* not read by the comparator of this path (quant_comp 9) */
        res.over_noise = 0;
    }
    LQ_MARK("cn_end");
    LQ_T(S, 14);
}

/* calc_noise's results become the working image's (the point where the reference calls it) */
LH_DEVFN void
lq_noise_commit(LhQS & S, LhQR & R, const LhGrR & g, const LhNoiseTmp & t)
{
    S.pnstep = t.pnstep;
    S.pnnoise = t.pnnoise;
    S.pnlog = t.pnlog;
    S.dist = t.dist;
    R.pn_global_gain = g.global_gain;
}

/* multiply the lines of the bands in `bands' by factor (xrpow only grows, so the lane's maximum
 * follows) */
template < int NS > LH_DEVFN void
lq_scale_mask(LhQS & S, uint64_t bands, float factor, int wide = 1)
{
#pragma unroll
    for (int k = 0; k < NS; k++) {
        /* (x * 1.0f == x for every float: one select on the factor instead of one per line) */
        float   f;
#if !defined(LH_EMU) && !defined(LH_NO_BFE_SELECT)
        /* (long blocks, 22 bands: the band's bit spread over a word by v_bfe_i32 picks the factor's or 1.0f's bits -- two
         * instructions instead of the 64-bit test's three and a select, see lq_quantize) */
        if (!wide) {
            uint32_t const on = (uint32_t) __builtin_amdgcn_sbfe((int) (uint32_t) bands, (unsigned) S.bnd[k], 1u);
            f = lh_u32_as_f32((on & lh_f32_as_u32(factor)) | (~on & 0x3f800000u));
        }
        else
#endif
            f = lq_bit(bands, S.bnd[k]) ? factor : 1.0f;
        float const v0 = S.xp[2 * k] * f, v1 = S.xp[2 * k + 1] * f;
        S.xp[2 * k] = v0;
        S.xp[2 * k + 1] = v1;
        S.lmax = __builtin_fmaxf(__builtin_fmaxf(v0, v1), S.lmax);      /* (no NaNs here: one v_max3_f32) */
    }
}

/* the same with a factor per band: flag / fac are lane = band values, distributed through LDS */
template < int NS > LH_DEVFN void
lq_scale_bands(const LhCtx & c, LhQS & S, LhChanLds & Q, int flag, float fac)
{
    LH_WAVE_SYNC();
    if (c.lane <= LH_SFBMAX)
        Q.sfb_f[c.lane] = flag ? fac : 1.0f;
    LH_WAVE_SYNC();
    {
        uint64_t const bands = lh_ballot(c.lane <= LH_SFBMAX && flag);
#pragma unroll
        for (int k = 0; k < NS; k++) {
            int const on = lq_bit(bands, S.bnd[k]);
            float const f = Q.sfb_f[S.bnd[k] < LH_SFBMAX ? S.bnd[k] : LH_SFBMAX];
            float const v0 = on ? S.xp[2 * k] * f : S.xp[2 * k];
            float const v1 = on ? S.xp[2 * k + 1] * f : S.xp[2 * k + 1];
            S.xp[2 * k] = v0;
            S.xp[2 * k + 1] = v1;
            S.lmax = __builtin_fmaxf(__builtin_fmaxf(v0, v1), S.lmax);
        }
    }
    LH_WAVE_SYNC();
}

/* Amplify the bands whose distortion reaches the trigger (reference quantize.c:720-796,
 * amp_scalefac_bands): the trigger is the largest distortion (noise_shaping_amp 2), its root or
 * 95 % of it (1), or 1 resp. 95 % (0); with amp 2 only the first such band is taken, and every second time
 * round a band flagged for substep shaping is passed over.  Written with selects: the cases differ
 * in a value, not in the work. */
template < int NS > LH_DEVFN void
lq_amp_scalefac_bands(const LhCtx & c, LhQS & S, const LhQR & R, LhGrR & g)
{
    int const s = c.lane;
    int const mine = s < R.sfbmax;
    float const dist = mine ? S.dist : 0.0f;
    float const widen = g.scalefac_scale ? (float) 1.68179283050742922612 : (float) 1.29683955465100964055;
    float const peak = lh_u32_as_f32(lh_wave_max_u32(lh_f32_as_u32(dist > 0.0f ? dist : 0.0f)));
    int const loud = peak > 1.0;
    float const damped = (float) (peak * .95);
    float const rooted = loud ? sqrtf(peak) : damped;   /* = (float) sqrt((double) peak): tests/test_quantizer_identity.py */
    float const capped = loud ? 1.0f : damped;
    float const trigger = (c.ns_amp == 2) ? peak : (c.ns_amp == 1) ? rooted : capped;
    uint64_t const cand = lh_ballot(mine && !(dist < trigger));
    int const only_first = (c.ns_amp == 2) && cand != 0;
    int const first = only_first ? lh_ffs64(cand) : R.sfbmax - 1;      /* the last band that is looked at */
    int const shaping = (R.substep_shaping & 2) != 0;
    uint64_t const flagged = lh_ballot(mine && S.ph);
    /* amp 2 with substep shaping: a flagged first band only drops its flag this time */
    int const pass_first = only_first && shaping && lq_bit(flagged, first);
    int const hit = mine && s <= first && lq_bit(cand, s);
    int const amplify = hit && !(pass_first && s == first);
    S.ph = (hit && shaping) ? !S.ph : S.ph;
    S.sfw += amplify;
    lq_scale_mask < NS > (S, lh_ballot(amplify), widen, R.block_type == LH_SHORT_TYPE);
}

/* reference takehiro.c:1135-1188 (MPEG-1) on the working scalefactors.  The two maxima the
 * reference takes (slen1 / slen2 part) are only compared with powers of two, so each lane
 * contributes the thermometer code of its scalefactor's bit length, the parts side by side,
 * and one OR over the wave replaces two maximum reductions. */
LH_DEVFN int
lq_scale_bitcount(const LhCtx & c, LhQS & S, const LhQR & R, LhGrR & g)
{
    int     k;
    int     v = (c.lane < R.sfbmax) ? S.sfw : 0;
    uint32_t th;
    if (LH_IS_LSF)
        return lh_scale_bitcount_lsf(c, R, g, v);       /* MPEG-2 / 2.5: four partitions, no preflag search */
    if (R.block_type != LH_SHORT_TYPE) {
        if (!g.preflag) {
            int const inr = (c.lane >= 11 && c.lane < LH_SBPSY_L);
            uint64_t const below = lh_ballot(inr && v < S.pre);
            if (below == 0) {
                g.preflag = 1;
                if (inr) {
                    v -= S.pre;
                    S.sfw = v;
                }
            }
        }
    }
    {
        int     bl = 32 - lh_clz32((uint32_t) (v > 0 ? v : 0));      /* 0 for v <= 0 */
        uint32_t t;
        bl = bl > 8 ? 8 : bl;
        t = (1u << bl) - 1u;
        th = lh_wave_or_u32((c.lane < R.sfbdivide) ? t : ((c.lane < R.sfbmax) ? (t << 8) : 0u));
    }
    g.part2_length = LH_LARGE_BITS;
    k = c.lane & 15;
    {
        unsigned key = 0xffffffffu, best;
        int const s1 = (int) ((0x4433322211130000ull >> (4 * k)) & 15u);
        int const s2 = (int) ((0x3232132132103210ull >> (4 * k)) & 15u);
        int const sz = (R.block_type == LH_SHORT_TYPE) ? 18 * (s1 + s2) : 11 * s1 + 10 * s2;
        /* max < 2^s  <=>  bit s of the thermometer code is clear */
        if (c.lane < 16 && !((th >> s1) & 1u) && !((th >> (8 + s2)) & 1u))
            key = ((unsigned) sz << 8) | (unsigned) k;
        best = lh_row0_min_u32(key);
        if (best != 0xffffffffu) {
            g.part2_length = (int) (best >> 8);
            g.scalefac_compress = (int) (best & 255u);
        }
    }
    return g.part2_length == LH_LARGE_BITS;
}

/* reference quantize.c:808-833 */
template < int NS > LH_DEVFN void
lq_inc_scalefac_scale(const LhCtx & c, LhQS & S, const LhQR & R, LhGrR & g)
{
    float const ifqstep34 = (float) 1.29683955465100964055;
    int     amp = 0;
    if (c.lane < R.sfbmax) {
        int     s = S.sfw;
        if (g.preflag)
            s += S.pre;
        if (s & 1) {
            s++;
            amp = 1;
        }
        S.sfw = s >> 1;
    }
    g.preflag = 0;
    g.scalefac_scale = 1;
    lq_scale_mask < NS > (S, lh_ballot(amp), ifqstep34, R.block_type == LH_SHORT_TYPE);
}

/* reference quantize.c:847-921 (short blocks) */
template < int NS > LH_DEVFN int
lq_inc_subblock_gain(const LhCtx & c, LhQS & S, LhChanLds & Q, const LhQR & R, LhGrR & g)
{
    const LhTables *T = c.T;
    int const k = c.lane;
    if (lh_ballot(k < R.sfb_lmax && S.sfw >= 16))
        return 1;
    for (int window = 0; window < 3; window++) {
        int const mine = (k >= R.sfb_lmax + window && k < R.sfbmax && ((k - R.sfb_lmax - window) % 3) == 0);
        int const s1 = (int) lh_wave_max_u32((mine && k < R.sfbdivide && S.sfw > 0) ? (unsigned) S.sfw : 0u);
        int const s2 = (int) lh_wave_max_u32((mine && k >= R.sfbdivide && S.sfw > 0) ? (unsigned) S.sfw : 0u);
        /* the band above the last scalefactor band of this window (the reference's loop variable after
         * its second loop): the first index >= sfbdivide of the form sfb_lmax + window + 3 n that is
         * >= sfbmax */
        int     top = R.sfb_lmax + window;
        while (top < R.sfbmax)
            top += 3;
        if (s1 < 16 && s2 < 8)
            continue;
        if (lh_sbg(g, window) >= 7)
            return 1;
        g.subblock_gain[0] += (window == 0);
        g.subblock_gain[1] += (window == 1);
        g.subblock_gain[2] += (window == 2);
        S.sbg8 = lh_sbg(g, S.win) * 8;
        {
            int     mode = 0;
            float   f = 1.0f;
            if (mine) {
                int     s = S.sfw;
                s = s - (4 >> g.scalefac_scale);
                if (s >= 0)
                    S.sfw = s;
                else {
                    int const gain = 210 + (s << (g.scalefac_scale + 1));
                    S.sfw = 0;
                    mode = 1;
                    f = T->ipow20[gain];
                }
            }
            else if (k == top && k <= LH_SFBMAX) {
                mode = 1;
                f = T->ipow20[202];
            }
            lq_scale_bands < NS > (c, S, Q, mode, f);
        }
    }
    return 0;
}

/* reference quantize.c:540-551 */
LH_DEVFN int
lq_loop_break(const LhCtx & c, const LhQS & S, const LhQR & R, const LhGrR & g)
{
    int const s = c.lane;
    int const z = (s < R.sfbmax) && (S.sfw + (S.sbg8 >> 3) == 0);        /* sbg8 = 8 subblock_gain[window of the band] */
    return lh_ballot(z) == 0;
}

/* reference quantize.c:940-988 */
template < int NS > LH_DEVFN int
lq_balance_noise(const LhCtx & c, LhQS & S, LhChanLds & Q, const LhQR & R, LhGrR & g)
{
    int     status;
    LQ_MARK("bn_amp");
    LQ_T(S, 20);
    lq_amp_scalefac_bands < NS > (c, S, R, g);
    LQ_MARK("bn_break");
    LQ_T(S, 21);
    status = lq_loop_break(c, S, R, g);
    if (status)
        return 0;
    LQ_MARK("bn_sbc");
    LQ_T(S, 22);
    status = lq_scale_bitcount(c, S, R, g);
    LQ_MARK("bn_rest");
    LQ_T(S, 23);
    if (!status)
        return 1;
    if (c.ns > 1) {
        S.ph = 0;
        if (!g.scalefac_scale) {
            lq_inc_scalefac_scale < NS > (c, S, R, g);
            status = 0;
        }
        else {
            if (R.block_type == LH_SHORT_TYPE && c.subblock_gain > 0)
                status = lq_inc_subblock_gain < NS > (c, S, Q, R, g) || lq_loop_break(c, S, R, g);
        }
    }
    if (!status)
        status = lq_scale_bitcount(c, S, R, g);
    return !status;
}

/* where the reference calls calc_noise: the noise of the candidate counted last becomes the working image's */
template < int NS, int HAVE_SQ = 0 > LH_DEVFN void
lq_noise_point(const LhCtx & c, LhQS & S, LhQR & R, const LhGrR & g, LhChanLds & Q, const float *xr, LhNoiseTmp & t,
               LhNoiseRes & res)
{
    lq_calc_noise < NS, HAVE_SQ > (c, S, R, g, Q, xr, t);
    lq_noise_commit(S, R, g, t);
    res = t.res;
}

/* The global gain at which the granule just fits (reference quantize.c:367-429, bin_search): start from
 * the gain the channel's previous granule ended at and walk towards the target in steps of 4 or 2; once
 * the walk has crossed the target (or hit an end of the range) every further step halves, and a step
 * of 1 ends the search.  Should the last trial be over the budget, the gain rises one by one until it fits.  The
 * channel remembers where it ended and whether it had to move far (lh_lds.ss.OldValue / CurrentStep). */
template < int NS > LH_DEVFN int
lq_bin_search(const LhCtx & c, LhQS & S, LhQR & R, LhGrR & g, LhChanLds & Q, int desired_rate, int ch)
{
    int const from = lh_uni_i(lh_lds.ss.OldValue[ch]);
    int const want = desired_rate - g.part2_length;
    int     stride = lh_uni_i(lh_lds.ss.CurrentStep[ch]);
    int     last_move = 0;      /* +1: the gain went up (too many bits), -1: down, 0: no trial yet */
    int     halving = 0;
    int     bits;
    g.global_gain = from;
    for (;;) {
        int     move;
        bits = lq_count_bits < 0, NS > (c, S, R, g, Q);
        if (stride == 1 || bits == want)
            break;
        move = (bits > want) ? 1 : -1;
        if (last_move == -move)
            halving = 1;
        if (halving)
            stride /= 2;
        last_move = move;
        g.global_gain += move * stride;
        if (g.global_gain < 0 || g.global_gain > 255) {
            g.global_gain = g.global_gain < 0 ? 0 : 255;
            halving = 1;
        }
    }
    while (bits > want && g.global_gain < 255) {
        g.global_gain++;
        bits = lq_count_bits < 0, NS > (c, S, R, g, Q);
    }
    if (c.lane == 0) {
        lh_lds.ss.CurrentStep[ch] = (from - g.global_gain >= 4) ? 4 : 2;
        lh_lds.ss.OldValue[ch] = g.global_gain;
    }
    g.part2_3_length = bits;
    return bits;
}

/* what the old VBR loop (OLD = 1) keeps besides: xrpow of the best image, to continue from in the next search of the
 * same granule (save_xrpow, reference quantize.c:1031, 1155, 1188-1190) */
struct LhQOld {
    float   xpb[10];
    float   lmaxb;
};

/* the working image becomes the best one: it goes to its final place in LDS (Q.ix[0]) */
template < int NS, int OLD = 0 > LH_DEVFN void
lq_keep_best(const LhCtx & c, LhQS & S, LhChanLds & Q, LhQOld * keep = nullptr)
{
#pragma unroll
    for (int k = 0; k < NS; k++) {
        int const p = c.lane + 64 * k;
        if (k < 4 || p < 288)
            ((uint32_t *) Q.ix[0])[p] = S.pw[k];
    }
    S.sfbest = S.sfw;
    S.tselb = S.tselw;
    if (OLD) {
#pragma unroll
        for (int k = 0; k < 10; k++)
            keep->xpb[k] = S.xp[k];
        keep->lmaxb = S.lmax;
    }
}

/* reference quantize.c:1010-1197; gb = cod_info.  On return the best image and its scalefactors
 * are in Q.ix[0] / Q.sf[0]. */
template < int NS, int OLD = 0 > LH_DEVFN int
lq_outer_loop(const LhCtx & c, LhQS & S, LhChanLds & Q, LhQR & R, LhGrR & gb, const float *xr, int ch, int targ_bits,
              int sfb21_extra = 0, LhQOld * keep = nullptr)
{
    LhGrR   gw;
    LhNoiseRes best_noise_info;
    LhNoiseTmp nt;
    int     huff_bits, better, age;
    int     best_part2_3_length = 9999999;

    {
        LH_PT(t_bs);
        LQ_T(S, 40);
        (void) lq_bin_search < NS > (c, S, R, gb, Q, targ_bits, ch);
        LQ_T(S, 41);
        LH_PA(7, t_bs);
    }
    best_noise_info.over_count = 100;
    if (c.ns) {
        R.pn_global_gain = 0;
        R.pn_sfb_count1 = 0;
        if (OLD) {
            /* a fresh calc_noise_data for every search (memset, reference quantize.c:1038) */
            S.pnstep = 0;
            S.pnnoise = 0;
            S.pnlog = 0;
        }
        lq_noise_point < NS > (c, S, R, gb, Q, xr, nt, best_noise_info);
        best_noise_info.bits = gb.part2_3_length;
        lq_keep_best < NS, OLD > (c, S, Q, keep);
        gw = gb;
        age = 0;
        LQ_T(S, 42);
        do {
            LhNoiseRes noise_info;
            int const search_limit = (R.substep_shaping & 2) ? 20 : 3;
            int     maxggain = 255;
            if (OLD && sfb21_extra) {
                /* the bands above the last one with a scalefactor cannot be amplified: once they are distorted
                 * the search is over (reference quantize.c:1077-1084) */
                int const s = c.lane;
                int const hit = (s == R.sfbmax || (R.block_type == LH_SHORT_TYPE && (s == R.sfbmax + 1 || s == R.sfbmax + 2)))
                    && S.dist > 1.0f;
                if (lh_ballot(hit) != 0)
                    break;
            }
            {
                LH_PT(t_bn);
                LQ_T(S, 33);            /* loop top: what lies between the comparison's outcome and the next balance_noise */
                int const bn = lq_balance_noise < NS > (c, S, Q, R, gw);
                LQ_T(S, 24);
                LH_PA(8, t_bn);
                if (bn == 0)
                    break;
            }
            if (gw.scalefac_scale)
                maxggain = 254;
            huff_bits = targ_bits - gw.part2_length;
            if (huff_bits <= 0)
                break;
            int     pn_before;  /* pn_sfb_count1 as the last count found it */
#if defined(LH_TRACE) && !defined(LH_EMU)
            int     tr_retries = 0;
#endif
            for (;;) {
                pn_before = R.pn_sfb_count1;
                gw.part2_3_length = lq_count_bits < 1, NS > (c, S, R, gw, Q, xr);
                if (!(gw.part2_3_length > huff_bits && gw.global_gain <= maxggain))
                    break;
                gw.global_gain++;
#if defined(LH_TRACE) && !defined(LH_EMU)
                tr_retries++;
#endif
            }
#if defined(LH_TRACE) && !defined(LH_EMU)
            /* (segments 34 .. 38: how many times the gain had to rise before the candidate fitted: 0, 1, 2, 3, more) */
            LQ_T(S, 34 + (tr_retries < 4 ? tr_retries : 4));
#endif
            if (gw.global_gain > maxggain)
                break;
            if (best_noise_info.over_count == 0) {
                /* The reference counts the same candidate once more here.  Everything that count reads is what the last
                 * one read or left -- the image of the cached bands, the tables of empty regions, the fields of gw --
                 * except pn_sfb_count1, which the last count rewrote and which picks the bands for the 0 / 1 comparator:
                 * when it came out as it went in, the second count is the first one again, and it is not run. */
                int const same = (R.pn_sfb_count1 == pn_before);
                if (!same || (gw.part2_3_length > best_part2_3_length && gw.global_gain <= maxggain)) {
                    gw.global_gain += same;     /* (the count that was not run said: too many bits) */
                    while ((gw.part2_3_length = lq_count_bits < 1, NS > (c, S, R, gw, Q, xr)) > best_part2_3_length
                           && gw.global_gain <= maxggain)
                        gw.global_gain++;
                }
                if (gw.global_gain > maxggain)
                    break;
            }
            {
                LH_PT(t_cn);
                LQ_T(S, 30);            /* the gain loop's exit checks, the recount rule */
#if defined(LH_NOISE_EARLY) && !defined(LH_EMU)
                lq_noise_point < NS, 1 > (c, S, R, gw, Q, xr, nt, noise_info);
#else
                lq_noise_point < NS > (c, S, R, gw, Q, xr, nt, noise_info);
#endif
                LH_PA(9, t_cn);
            }
            noise_info.bits = gw.part2_3_length;
            better = lh_quant_compare(best_noise_info, noise_info);
            LQ_T(S, 31);                /* commit + comparison */
            if (better) {
                best_part2_3_length = gb.part2_3_length;
                best_noise_info = noise_info;
                lq_keep_best < NS, OLD > (c, S, Q, keep);
                gb = gw;
                age = 0;
                LQ_T(S, 32);            /* the candidate kept as the best */
            }
            else {
                if (c.full_outer_loop == 0) {
                    if (++age > search_limit && best_noise_info.over_count == 0)
                        break;
                }
            }
        }
        while ((gw.global_gain + gw.scalefac_scale) < 255);
    }
    else
        lq_keep_best < NS, OLD > (c, S, Q, keep);
    if (OLD) {
        /* the next search of this granule continues from the best image: its xrpow, scalefactors, subblock gains */
#pragma unroll
        for (int k = 0; k < 10; k++)
            S.xp[k] = keep->xpb[k];
        S.lmax = keep->lmaxb;
        S.sfw = S.sfbest;
        S.tselw = S.tselb;      /* (tables of regions the next count leaves empty stay as they are) */
        S.sbg8 = 8 * lh_sbg(gb, S.win);
    }
    gb.table_select[0] = (int) lh_bcast_u32((uint32_t) S.tselb, 0);
    gb.table_select[1] = (int) lh_bcast_u32((uint32_t) S.tselb, 1);
    gb.table_select[2] = (int) lh_bcast_u32((uint32_t) S.tselb, 2);
    /* hand the result to the finishing stages through the LDS image (the quantised lines are
     * there already; with four slots the fifth is zero since lq_load) */
    LH_WAVE_SYNC();
    if (c.lane <= LH_SFBMAX)
        Q.sf[0][c.lane] = S.sfbest;
    LH_WAVE_SYNC();
    return best_noise_info.over_count;
}

/* out-of-line entries (own register allocation): R / g travel through the channel's LDS slot.  One per
 * slot count; the fifth slot (lines 512..575) drops out of the search when nothing is quantised
 * there and its xrpow is all zero (it could otherwise still raise xrpow_max): the usual case below
 * 20 kHz. */
/* SPEC: the stage compiled for the usual case (lh_granule_is_usual, lh_dev_common.h) -- what that checked is a constant in
 * its code, and what the other cases need (the region split and band ranges of the other block types, the amplification
 * rules of the other presets, substep shaping, the band above the last scalefactor band) is not in it; SPEC = the class's
 * noise shaping (lh_cfg_class) */
template < int NS, int SPEC = 0, int SHORT = 0 > LH_DEVFN void
lq_stage_body(int qch, int gr, int targ_bits)
{
    LhCtx   c = lh_ctx_load();
    LhQR    R = lh_uniform(lh_lds.rg[qch].R);
    LhGrR   g = lh_uniform(lh_lds.rg[qch].g);
    if (SPEC) {
        lh_pin_usual(c, SPEC);
        if (SHORT) {
            c.subblock_gain = 1;
            lh_pin_usual_short(R);
        }
        else
            lh_pin_usual(R);
    }
    LhChanLds & Q = lh_lds.u.quant.ch[qch];
    LhQS    S;
#if defined(LH_NOISE_EARLY) || defined(LH_NO_PAD)
    S.pad = 0;
#else
    S.pad = (SPEC != 0 && !SHORT);
#endif
    R.s_mnc = lh_uni_i((int) Q.sfb_of_line[R.mnc]);
#if defined(LH_TRACE) && !defined(LH_EMU)
    lh_tr_begin(S.tr);
#endif
    lq_load(c, S, Q, R, g, qch);
    LQ_T(S, 51);
    lq_zero_band_noise < NS > (c, S, R, Q, lh_lds.xr[qch][lh_uni_i(gr)]);
    LQ_T(S, 52);
    (void) lq_outer_loop < NS > (c, S, Q, R, g, lh_lds.xr[qch][lh_uni_i(gr)], qch, lh_uni_i(targ_bits));
    LQ_T(S, 53);
    LQ_T(S, 63);                /* (nothing between two marks: what a mark costs) */
    lh_rg_put(c, R, g);
#if defined(LH_TRACE) && !defined(LH_EMU)
    lh_tr_flush(S.tr, c.wave);
#endif
}

#ifdef LH_VBR_OLD
/* The old VBR loop's search for one granule (reference quantize.c:1245-1331, VBR_encode_granule): a bisection over
 * the bit budget between min_bits and max_bits; every trial is the search above at that budget, continuing from the
 * best quantisation found so far (in registers all the time, and so is the one that fitted with the fewest bits,
 * which is set aside: image, scalefactors, xrpow).  `cont': the granule comes with the scalefactors
 * of an earlier pass over the frame (R / g in the channel's slot, Q.sf[0]); xrpow, the allowed noise and the
 * geometry are fresh in Q as for any search. */
/* SPEC (1: without, 2: with the sfb21 band): the same for the usual case of this loop -- a normal long block of an MPEG-1 stream
 * under the vbr_rh presets (noise shaping 1, amplification rule 1, no full outer loop, no substep shaping): lh_vbrold_class() */
template < int NS, int SPEC = 0 > LH_DEVFN void
lq_vbrold_body(int qch, int gr, int min_bits, int max_bits, int cont)
{
    LhCtx   c = lh_ctx_load();
    LhQR    R = lh_uniform(lh_lds.rg[qch].R);
    LhGrR   g = lh_uniform(lh_lds.rg[qch].g);
    if (SPEC) {
        c.ns = 1;
        c.ns_amp = 1;
        c.full_outer_loop = 0;
        c.sfb21_extra = (SPEC == 2);
        c.rate8k = 0;
        R.block_type = LH_NORM_TYPE;
        R.sfb_lmax = LH_SBPSY_L;
        R.sfb_smin = LH_SBPSY_S;
        R.psy_lmax = (SPEC == 2) ? LH_SBMAX_L : LH_SBPSY_L;
        R.psymax = R.psy_lmax;
        R.sfbmax = LH_SBPSY_L;
        R.sfbdivide = 11;
        if ((R.substep_shaping & 2) != 0)
            __builtin_unreachable();
    }
    LhChanLds & Q = lh_lds.u.quant.ch[qch];
    const float *xr = lh_lds.xr[qch][lh_uni_i(gr)];
    LhQS    S;
    LhQOld  K;
    LhGrR   gbst = g;
    float   xpbst[10], lmaxbst = 0.0f;
    uint32_t pwbst[5] = { 0u, 0u, 0u, 0u, 0u };
    int     sfbst = 0, tselbst = 0;
    int const band = c.lane <= LH_SFBMAX ? c.lane : LH_SFBMAX;
    int     found = 0, dbits, this_bits;
    min_bits = lh_uni_i(min_bits);
    max_bits = lh_uni_i(max_bits);
    int const top = max_bits;
    R.s_mnc = lh_uni_i((int) Q.sfb_of_line[R.mnc]);
    S.pad = 0;
    lq_load(c, S, Q, R, g, qch);
    lq_zero_band_noise < NS > (c, S, R, Q, xr);
    if (lh_uni_i(cont)) {
        S.sfw = Q.sf[0][band];
        S.sfbest = S.sfw;
        S.sbg8 = 8 * lh_sbg(g, S.win);
        S.tselw = (c.lane == 0) ? g.table_select[0] : (c.lane == 1) ? g.table_select[1] : g.table_select[2];
        S.tselb = S.tselw;
    }
#pragma unroll
    for (int k = 0; k < 10; k++)
        xpbst[k] = 0.0f;
    this_bits = (max_bits + min_bits) / 2;
    do {
        int const sfb21 = c.sfb21_extra && !(this_bits > top - 42);
        int const over = lq_outer_loop < NS, 1 > (c, S, Q, R, g, xr, qch, this_bits, sfb21, &K);
        if (over <= 0) {
            /* it can be done with these bits: set it aside, try fewer */
            found = 1;
            gbst = g;
            sfbst = S.sfbest;
            tselbst = S.tselb;
            lmaxbst = S.lmax;
#pragma unroll
            for (int k = 0; k < 10; k++)
                xpbst[k] = S.xp[k];
            LH_WAVE_SYNC();
#pragma unroll
            for (int k = 0; k < 5; k++) {
                int const p = c.lane + 64 * k;
                pwbst[k] = ((const uint32_t *) Q.ix[0])[p < 288 ? p : 287];
            }
            max_bits = g.part2_3_length - 32;
            dbits = max_bits - min_bits;
            this_bits = (max_bits + min_bits) / 2;
        }
        else {
            /* try more, from the best one so far */
            min_bits = this_bits + 32;
            dbits = max_bits - min_bits;
            this_bits = (max_bits + min_bits) / 2;
            if (found) {
                found = 2;
                g = gbst;
                S.sfbest = sfbst;
                S.sfw = sfbst;
                S.tselb = tselbst;
                S.tselw = tselbst;
                S.sbg8 = 8 * lh_sbg(g, S.win);
                S.lmax = lmaxbst;
#pragma unroll
                for (int k = 0; k < 10; k++)
                    S.xp[k] = xpbst[k];
                LH_WAVE_SYNC();
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    int const p = c.lane + 64 * k;
                    if (p < 288)
                        ((uint32_t *) Q.ix[0])[p] = pwbst[k];
                }
                LH_WAVE_SYNC();
            }
        }
    }
    while (dbits > 12);
    LH_WAVE_SYNC();
    if (c.lane <= LH_SFBMAX)
        Q.sf[0][c.lane] = S.sfbest;
    LH_WAVE_SYNC();
    lh_rg_put(c, R, g);
}

LH_STAGEFN void
lq_vbrold_stage5(int qch, int gr, int min_bits, int max_bits, int cont)
{
    lq_vbrold_body < 5 > (qch, gr, min_bits, max_bits, cont);
}

LH_STAGEFN void
lq_vbrold_stage4(int qch, int gr, int min_bits, int max_bits, int cont)
{
    lq_vbrold_body < 4 > (qch, gr, min_bits, max_bits, cont);
}

/* 0: none; 1 / 2: the usual case without / with the sfb21 band */
LH_DEVFN int
lh_vbrold_class(const LhCtx & c, int block_type, int substep)
{
    int const ok = !LH_IS_LSF && block_type == LH_NORM_TYPE && c.ns == 1 && c.ns_amp == 1 && c.full_outer_loop == 0
        && (substep & 2) == 0;
    return ok ? 1 + (c.sfb21_extra != 0) : 0;
}

LH_STAGEFN void
lq_vbrold_stage5n(int qch, int gr, int min_bits, int max_bits, int cont)
{
    lq_vbrold_body < 5, 2 > (qch, gr, min_bits, max_bits, cont);
}

LH_STAGEFN void
lq_vbrold_stage4n(int qch, int gr, int min_bits, int max_bits, int cont)
{
    lq_vbrold_body < 4, 2 > (qch, gr, min_bits, max_bits, cont);
}

LH_STAGEFN void
lq_vbrold_stage5m(int qch, int gr, int min_bits, int max_bits, int cont)
{
    lq_vbrold_body < 5, 1 > (qch, gr, min_bits, max_bits, cont);
}

LH_STAGEFN void
lq_vbrold_stage4m(int qch, int gr, int min_bits, int max_bits, int cont)
{
    lq_vbrold_body < 4, 1 > (qch, gr, min_bits, max_bits, cont);
}
#endif


LH_STAGEFN void
lq_outer_loop_stage5(int qch, int gr, int targ_bits)
{
    lq_stage_body < 5 > (qch, gr, targ_bits);
}

LH_STAGEFN void
lq_outer_loop_stage4(int qch, int gr, int targ_bits)
{
    lq_stage_body < 4 > (qch, gr, targ_bits);
}

/* The usual case as stages of their own (round 6: +6.3 % on the headline): a long block of the normal
 * type under the settings lq_stage_is_usual() names -- all of them constants in this code, so the other block types' region
 * split and band ranges, the other presets' amplification rules, substep shaping and the sfb21 band are not compiled into
 * it: count_bits 570 -> ~545 instructions, the rest of an iteration ~350 -> ~250, 194 VGPRs instead of 226. */
LH_STAGEFN void
lq_outer_loop_stage4n(int qch, int gr, int targ_bits)
{
    lq_stage_body < 4, 2 > (qch, gr, targ_bits);
}

LH_STAGEFN void
lq_outer_loop_stage5n(int qch, int gr, int targ_bits)
{
    lq_stage_body < 5, 2 > (qch, gr, targ_bits);
}

/* (short blocks of the same classes: s / t) */
LH_STAGEFN void
lq_outer_loop_stage4s(int qch, int gr, int targ_bits)
{
    lq_stage_body < 4, 2, 1 > (qch, gr, targ_bits);
}

LH_STAGEFN void
lq_outer_loop_stage5s(int qch, int gr, int targ_bits)
{
    lq_stage_body < 5, 2, 1 > (qch, gr, targ_bits);
}

LH_STAGEFN void
lq_outer_loop_stage4t(int qch, int gr, int targ_bits)
{
    lq_stage_body < 4, 1, 1 > (qch, gr, targ_bits);
}

LH_STAGEFN void
lq_outer_loop_stage5t(int qch, int gr, int targ_bits)
{
    lq_stage_body < 5, 1, 1 > (qch, gr, targ_bits);
}

/* (noise shaping 1: the presets above 128 kb/s) */
LH_STAGEFN void
lq_outer_loop_stage4m(int qch, int gr, int targ_bits)
{
    lq_stage_body < 4, 1 > (qch, gr, targ_bits);
}

LH_STAGEFN void
lq_outer_loop_stage5m(int qch, int gr, int targ_bits)
{
    lq_stage_body < 5, 1 > (qch, gr, targ_bits);
}

/* which of the two applies: wave-uniform; Q.xrpow and R.mnc are final (after lh_calc_xmin) */
LH_DEVFN int
lq_needs_tail(const LhCtx & c, const LhChanLds & Q, const LhQR & R)
{
    int const i = 512 + 2 * (c.lane & 31);
    return (R.mnc >= 512) || (lh_ballot(Q.xrpow[i] != 0.0f || Q.xrpow[i + 1] != 0.0f) != 0);
}

#endif
