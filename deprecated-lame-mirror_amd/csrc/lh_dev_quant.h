/*
 * lh_dev_quant.h -- CBR quantisation / noise-shaping / Huffman-selection loop
 * on one wavefront per stream-channel
 * (reference quantize.c:48-1232,1988-2050, quantize_pvt.c:428-913,
 * takehiro.c:113-1327, reservoir.c:82-293).
 *
 * Data layout: the 576 spectral lines of the granule live in LDS (xr, xrpow,
 * two int16 quantised images: best-so-far and working).  Lane l owns the line
 * pairs l, l+64, ... (288 pairs), so a pair never straddles lanes and LDS
 * accesses are stride-1.  The noise-shaping search is a chain of wave-uniform
 * decisions; inside each step
 *   - quantisation is per line (lanes over pairs),
 *   - ix_max / Huffman bit sums / "last non-zero pair" are wave reductions,
 *   - per-scalefactor-band sums (xmin, noise) keep the reference's serial
 *     order inside the band and run one band per lane.
 * Wave-uniform scalars of gr_info live in registers (LhGrR / LhQR: every lane
 * carries an identical copy), per-band arrays in LDS are written by lane == band
 * only, per-line arrays by the lane owning the line; LH_WAVE_SYNC() separates a
 * write phase from the reads of other lanes.  The code is therefore correct for
 * any interleaving of the lanes between two syncs (the CPU fiber emulator of
 * tests/hipemu relies on that), not just for lockstep execution.
 */
#ifndef LH_DEV_QUANT_H
#define LH_DEV_QUANT_H

#include "lh_dev_common.h"

#define LH_MAGIC_FLOAT (65536*(128))
#define LH_MAGIC_INT 0x4b000000

LH_DEVCONST int lh_slen1_n[16] = { 1, 1, 1, 1, 8, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16 };
LH_DEVCONST int lh_slen2_n[16] = { 1, 2, 4, 8, 1, 2, 4, 8, 2, 4, 8, 2, 4, 8, 4, 8 };
LH_DEVCONST int lh_slen1_tab[16] = { 0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4 };
LH_DEVCONST int lh_slen2_tab[16] = { 0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3 };
LH_DEVCONST int lh_scfsi_band[5] = { 0, 6, 11, 16, 21 };
LH_DEVCONST int lh_scale_short[16] = { 0, 18, 36, 54, 54, 36, 54, 72, 54, 72, 90, 72, 90, 108, 108, 126 };
LH_DEVCONST int lh_scale_long[16] = { 0, 10, 20, 30, 33, 21, 31, 41, 32, 42, 52, 43, 53, 63, 64, 74 };
LH_DEVCONST int lh_huf_tbl_noESC[15] = { 1, 2, 5, 7, 7, 10, 10, 13, 13, 13, 13, 13, 13, 13, 13 };

/* Huffman table metadata as immediates (ISO 11172-3 table B.7; equality with the
 * generated lh_ht_xlen / lh_ht_linmax / lh_ht_offset arrays is asserted by
 * tests/test_abi.py::test_huffman_immediates).  Avoids chains of dependent loads. */
LH_DEVFN int
lh_ht_off(int t)
{
    switch (t) {
    case 1: return 0;
    case 2: return 4;
    case 3: return 13;
    case 5: return 22;
    case 6: return 38;
    case 7: return 54;
    case 8: return 90;
    case 9: return 126;
    case 10: return 162;
    case 11: return 226;
    case 12: return 290;
    case 13: return 354;
    case 14: return 610;
    case 15: return 866;
    default: return (t >= 32) ? (t == 32 ? 1634 : 1650) : (t >= 24 ? 1378 : 1122);
    }
}

LH_DEVFN unsigned
lh_ht_xlen_c(int t)
{
    /* tables 0..15: alphabet size per dimension; 16..31: linbits; one byte per table */
    unsigned long long const k = (t < 8) ? 0x0604040003030200ull
        : (t < 16) ? 0x1000100808080606ull : (t < 24) ? 0x0D0A080604030201ull : 0x0D0B090807060504ull;
    return (unsigned) ((k >> (8 * (t & 7))) & 0xffu);
}

LH_DEVFN unsigned
lh_ht_linmax_c(int t)
{
    return (1u << lh_ht_xlen_c(t)) - 1u;        /* tables 16..31 */
}

LH_DEVFN int
lh_huf_noESC(unsigned mx)       /* first candidate table for a region maximum 1..15 */
{
    return mx == 1 ? 1 : mx == 2 ? 2 : mx == 3 ? 5 : mx <= 5 ? 7 : mx <= 7 ? 10 : 13;
}

#define LH_HLEN(t)  (qt->ht_len + lh_ht_off(t))

/* ---------------------------------------------------------------------- */
/* one line through the x^(3/4) quantiser (reference takehiro.c:144-200)     */
LH_DEVFN int
lh_quant_line(const LhTables * T, const LhQTabs * qt, float istep, float xp)
{
    /* first roundings below 256 (nearly all): a float addition, then the second rounding as a comparison
     * (LhTables.qthr, in LDS; tests/test_quantizer_identity.py); the rest follows the reference's
     * expression with the offset from HBM */
    float const a = istep * xp;
    int const k = (int) lh_f32_as_u32(a + (float) LH_MAGIC_FLOAT) - LH_MAGIC_INT;
    int     q = k - (a < qt->qthr[k & 255] ? 1 : 0);
    if (k >= 256) {
        double const x0 = (double) a + LH_MAGIC_FLOAT;
        q = (int) lh_f32_as_u32((float) (x0 + T->adj43asm[k])) - LH_MAGIC_INT;
    }
    return q;
}

/* Huffman cost of the pairs [lo,hi) of ix with the best table; all lanes take
 * part (reference choose_table_nonMMX, takehiro.c:423-647).  v[][] are this
 * lane's 5 pairs, pair index p = lane + 64 k. */
LH_DEVFN int
lh_choose_table_wave(const LhCtx & c, const int v[5][2], int lo, int hi, int *bits)
{
    const LhQTabs *qt = LH_QT;
    unsigned mx = 0;
    int const plo = lo >> 1, phi = hi >> 1;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        int const p = c.lane + 64 * k;
        if (p >= plo && p < phi) {
            unsigned const m = (unsigned) (v[k][0] > v[k][1] ? v[k][0] : v[k][1]);
            mx = m > mx ? m : mx;
        }
    }
    mx = lh_wave_max_u32(mx);
    if (mx == 0)
        return 0;
    if (mx <= 15) {
        if (mx == 1) {
            unsigned s = 0;
            const uint8_t *h1 = LH_HLEN(1);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                int const p = c.lane + 64 * k;
                if (p >= plo && p < phi)
                    s += h1[v[k][0] + v[k][0] + v[k][1]];
            }
            s = lh_wave_sum_u32(s);
            *bits += (int) s;
            return 1;
        }
        if (mx <= 3) {
            int     t1 = lh_huf_noESC(mx);
            unsigned const xlen = lh_ht_xlen_c(t1);
            const uint32_t *table = (t1 == 2) ? qt->table23 : qt->table56;
            unsigned s = 0, s2;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                int const p = c.lane + 64 * k;
                if (p >= plo && p < phi)
                    s += table[(unsigned) v[k][0] * xlen + (unsigned) v[k][1]];
            }
            s = lh_wave_sum_u32(s);
            s2 = s & 0xffffu;
            s >>= 16u;
            if (s > s2) {
                s = s2;
                t1++;
            }
            *bits += (int) s;
            return t1;
        }
        {
            int const t1 = lh_huf_noESC(mx);
            unsigned const xlen = lh_ht_xlen_c(t1);
            const uint8_t *h1 = LH_HLEN(t1), *h2 = LH_HLEN(t1 + 1), *h3 = LH_HLEN(t1 + 2);
            unsigned w0 = 0, w1 = 0;
            unsigned s1, s2, s3;
            int     t;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                int const p = c.lane + 64 * k;
                if (p >= plo && p < phi) {
                    unsigned const x = (unsigned) v[k][0] * xlen + (unsigned) v[k][1];
                    w0 += (unsigned) h1[x] | ((unsigned) h2[x] << 16);
                    w1 += (unsigned) h3[x];
                }
            }
            w0 = lh_wave_sum_u32(w0);
            w1 = lh_wave_sum_u32(w1);
            s1 = w0 & 0xffffu;
            s2 = w0 >> 16;
            s3 = w1;
            t = t1;
            if (s1 > s2) {
                s1 = s2;
                t++;
            }
            if (s1 > s3) {
                s1 = s3;
                t = t1 + 2;
            }
            *bits += (int) s1;
            return t;
        }
    }
    if (mx > LH_IXMAX) {
        *bits = LH_LARGE_BITS;
        return -1;
    }
    {
        int     choice, choice2;
        unsigned const m15 = mx - 15u;
        unsigned w0 = 0, w1 = 0;
        unsigned sa, sb, n15;
        for (choice2 = 24; choice2 < 32; choice2++)
            if (lh_ht_linmax_c(choice2) >= m15)
                break;
        for (choice = choice2 - 8; choice < 24; choice++)
            if (lh_ht_linmax_c(choice) >= m15)
                break;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int const p = c.lane + 64 * k;
            if (p >= plo && p < phi) {
                unsigned x = (unsigned) v[k][0], y = (unsigned) v[k][1];
                unsigned e;
                if (x >= 15u) {
                    x = 15u;
                    w1++;
                }
                if (y >= 15u) {
                    y = 15u;
                    w1++;
                }
                e = lh_largetbl[(x << 4) + y];     /* the constant in HBM: not a hot path */
                w0 += e;        /* high half: table 16.. lengths, low half: table 24.. lengths */
            }
        }
        w0 = lh_wave_sum_u32(w0);
        n15 = lh_wave_sum_u32(w1);
        sa = (w0 >> 16) + n15 * lh_ht_xlen_c(choice);
        sb = (w0 & 0xffffu) + n15 * lh_ht_xlen_c(choice2);
        if (sa > sb) {
            sa = sb;
            choice = choice2;
        }
        *bits += (int) sa;
        return choice;
    }
}

/* the same selection done serially by ONE lane over ix[lo,hi) in LDS; used where
 * many independent regions are evaluated at once, one per lane (best_huffman_divide) */
LH_DEVFN int
lh_choose_table_lane(const LhQTabs * qt, const int16_t * ix, int lo, int hi, int *bits)
{
    unsigned mx = 0;
    for (int i = lo; i < hi; i++) {
        unsigned const x = (unsigned) ix[i];
        mx = x > mx ? x : mx;
    }
    if (mx == 0)
        return 0;
    if (mx <= 15) {
        if (mx == 1) {
            unsigned s = 0;
            const uint8_t *h1 = LH_HLEN(1);
            for (int i = lo; i < hi; i += 2)
                s += h1[ix[i] + ix[i] + ix[i + 1]];
            *bits += (int) s;
            return 1;
        }
        if (mx <= 3) {
            int     t1 = lh_huf_noESC(mx);
            unsigned const xlen = lh_ht_xlen_c(t1);
            const uint32_t *table = (t1 == 2) ? qt->table23 : qt->table56;
            unsigned s = 0, s2;
            for (int i = lo; i < hi; i += 2)
                s += table[(unsigned) ix[i] * xlen + (unsigned) ix[i + 1]];
            s2 = s & 0xffffu;
            s >>= 16u;
            if (s > s2) {
                s = s2;
                t1++;
            }
            *bits += (int) s;
            return t1;
        }
        {
            int const t1 = lh_huf_noESC(mx);
            unsigned const xlen = lh_ht_xlen_c(t1);
            const uint8_t *h1 = LH_HLEN(t1), *h2 = LH_HLEN(t1 + 1), *h3 = LH_HLEN(t1 + 2);
            unsigned s1 = 0, s2 = 0, s3 = 0;
            int     t;
            for (int i = lo; i < hi; i += 2) {
                unsigned const x = (unsigned) ix[i] * xlen + (unsigned) ix[i + 1];
                s1 += h1[x];
                s2 += h2[x];
                s3 += h3[x];
            }
            t = t1;
            if (s1 > s2) {
                s1 = s2;
                t++;
            }
            if (s1 > s3) {
                s1 = s3;
                t = t1 + 2;
            }
            *bits += (int) s1;
            return t;
        }
    }
    if (mx > LH_IXMAX) {
        *bits = LH_LARGE_BITS;
        return -1;
    }
    {
        int     choice, choice2;
        unsigned const m15 = mx - 15u;
        unsigned sa = 0, sb = 0, n15 = 0;
        for (choice2 = 24; choice2 < 32; choice2++)
            if (lh_ht_linmax_c(choice2) >= m15)
                break;
        for (choice = choice2 - 8; choice < 24; choice++)
            if (lh_ht_linmax_c(choice) >= m15)
                break;
        for (int i = lo; i < hi; i += 2) {
            unsigned x = (unsigned) ix[i], y = (unsigned) ix[i + 1], e;
            if (x >= 15u) {
                x = 15u;
                n15++;
            }
            if (y >= 15u) {
                y = 15u;
                n15++;
            }
            e = lh_largetbl[(x << 4) + y];     /* the constant in HBM: not a hot path */
            sa += e >> 16;
            sb += e & 0xffffu;
        }
        sa += n15 * lh_ht_xlen_c(choice);
        sb += n15 * lh_ht_xlen_c(choice2);
        if (sa > sb) {
            sa = sb;
            choice = choice2;
        }
        *bits += (int) sa;
        return choice;
    }
}

/* Look-up parameters of one region, packed for a per-lane select: the candidate tables are
 * read from the byte pool ht_len at o1/o2/o3 + x * xlen + y, or (ESC classes) from largetbl at
 * x * 16 + y; unused candidates alias the first one (their sums are not read). */
struct LhRegionLut {
    uint32_t pa;                /* xlen | esc << 8 | o1 << 16 */
    uint32_t pb;                /* o2 | o3 << 16 */
};

LH_DEVFN LhRegionLut
lh_region_lut(unsigned mx)
{
    LhRegionLut r;
    if (mx == 0 || mx > LH_IXMAX) {
        r.pa = 2u;              /* nothing is read from the sums; keep the indices in bounds */
        r.pb = 0u;
    }
    else if (mx > 15) {
        unsigned const o = (unsigned) lh_ht_off(13);    /* any 16 x 16 table keeps the byte reads in bounds */
        r.pa = 16u | (1u << 8) | (o << 16);
        r.pb = o | (o << 16);
    }
    else {
        int const t1 = lh_huf_noESC(mx);
        unsigned const o1 = (unsigned) lh_ht_off(t1);
        unsigned const o2 = (t1 >= 2) ? (unsigned) lh_ht_off(t1 + 1) : o1;
        unsigned const o3 = (t1 >= 7) ? (unsigned) lh_ht_off(t1 + 2) : o1;
        r.pa = lh_ht_xlen_c(t1) | (o1 << 16);
        r.pb = o2 | (o3 << 16);
    }
    return r;
}

/* the scalar half of choose_table: table index and bits from the wave totals */
LH_DEVFN int
lh_region_decide(unsigned mx, unsigned w0, unsigned w1, int *bits)
{
    if (mx <= 15u) {
        /* straight-line selects (scalar unit): first candidate table and the number of
         * candidates as nibble / 2-bit tables indexed by the region maximum */
        int const t1 = (int) ((0xDDDDDDDDAA775210ull >> (4u * mx)) & 15u);
        int const nc = (int) ((0xFFFFFFA4u >> (2u * mx)) & 3u);
        unsigned const s1 = w0 & 0xffffu, s2 = w0 >> 16, s3 = w1;
        int const take2 = (nc >= 2) && (s1 > s2);
        unsigned const sa = take2 ? s2 : s1;
        int const take3 = (nc >= 3) && (sa > s3);
        unsigned const sb = take3 ? s3 : sa;
        *bits += (nc > 0) ? (int) sb : 0;
        return take3 ? t1 + 2 : (take2 ? t1 + 1 : t1);
    }
    if (mx > LH_IXMAX) {
        *bits = LH_LARGE_BITS;
        return -1;
    }
    {
        int     choice, choice2;
        unsigned const m15 = mx - 15u;
        unsigned sa, sb;
        /* smallest table of 24..31, and of choice2-8..23, whose linbits hold mx - 15: the
         * reference's two linear searches (takehiro.c:631-640) as a function of the bit
         * length of mx - 15 (linbits 16..23: 1,2,3,4,6,8,10,13; 24..31: 4..9,11,13) */
        {
            int const blen = 32 - lh_clz32(m15);                /* 1..13 */
            int const t24 = (int) ((0x7777665432100000ull >> (4 * blen)) & 15u);
            int const t16 = (int) ((0x7777766554432100ull >> (4 * blen)) & 15u);
            choice2 = 24 + t24;
            choice = 16 + (t16 > t24 ? t16 : t24);
        }
        sa = (w0 >> 16) + w1 * lh_ht_xlen_c(choice);
        sb = (w0 & 0xffffu) + w1 * lh_ht_xlen_c(choice2);
        if (sa > sb) {
            sa = sb;
            choice = choice2;
        }
        *bits += (int) sa;
        return choice;
    }
}

/* index of the highest set bit + 1 over five 64-bit ballot words (word k covers pairs 64k..) */
LH_DEVFN int
lh_top_of_masks(const uint64_t m[5])
{
    int     top = 0;
#pragma unroll
    for (int k = 0; k < 5; k++)
        if (m[k])
            top = 64 * k + 64 - lh_clz64(m[k]);
    return top;
}

/* Huffman bit count of a quantised image held as packed pairs in registers (pk[k] = pair
 * lane + 64 k: low half = even line); the image is also in LDS (Q.ix[which]) and the LDS
 * copy is what the count1 quadruples are read from.  Reference takehiro.c:654-765.
 * The cross-lane work is arranged in few dependent steps: ballots find count1 / big_values,
 * the three region maxima are reduced together, and so are all table sums. */
LH_DEVFN int
lh_noquant_count_bits(const LhCtx & c, LhChanLds & Q, LhQR & R, LhGrR & g, int which, int use_prev,
                      const uint32_t pk[5])
{
    const LhQTabs *qt = LH_QT;
    const uint32_t *ix2 = (const uint32_t *) Q.ix[which];
    int const lane = c.lane;
    int const i0p = (((R.mnc + 2) >> 1) > 288) ? 288 : ((R.mnc + 2) >> 1);
    uint64_t mk[5];
    int     top_nz, top_big, i, bv, nquad, bits;
    int     e0, e1, e2;             /* pair index where regions 0, 1, 2 end */
    int     a1, a2;
    unsigned quads = 0, sfbcnt_in = 0;
    LH_PT(t_nq);

    if (use_prev)
        R.pn_sfb_count1 = 0;
#pragma unroll
    for (int k = 0; k < 5; k++)
        mk[k] = lh_ballot((lane + 64 * k) < i0p && pk[k] != 0);
    top_nz = lh_top_of_masks(mk);
    i = 2 * top_nz;
    g.count1 = i;
#pragma unroll
    for (int k = 0; k < 5; k++)
        mk[k] = lh_ballot((lane + 64 * k) < top_nz && (pk[k] & 0xfffefffeu) != 0);
    top_big = lh_top_of_masks(mk);
    nquad = (i - 2 * top_big) / 4;
    bv = i - 4 * nquad;
    g.big_values = bv;
    /* region layout (scalar) */
    if (R.block_type == LH_SHORT_TYPE) {
        a1 = 3 * (int) qt->sfb_s3;
        a2 = bv;
    }
    else if (R.block_type == LH_NORM_TYPE) {
        if (bv > 0) {
            uint32_t const pack = qt->bvpack[(bv >> 1) - 1];
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
        a1 = qt->sfb_l[7 + 1];
        a2 = bv;
    }
    a1 = (a1 < bv) ? a1 : bv;
    a2 = (a2 < bv) ? a2 : bv;
    e0 = a1 >> 1;
    e1 = a2 >> 1;
    e2 = (R.block_type == LH_NORM_TYPE) ? (bv >> 1) : e1;
    if (use_prev && R.block_type == LH_NORM_TYPE)
        sfbcnt_in = (lane < LH_SBMAX_L + 1) ? qt->sfb_l[lane] : 576u;
    LH_PA(19, t_nq);
    /* count1 region: quadruples of 0/1 values, from the LDS image (at most 144 of them: three
     * per lane, loaded unconditionally at clamped positions) */
    LH_WAVE_SYNC();
    {
        unsigned idx[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            int const qd = lane + 64 * k;
            int const b2 = (bv >> 1) + 2 * qd;
            int const b2c = b2 < 286 ? b2 : 286;
            uint32_t const u0 = ix2[b2c], u1 = ix2[b2c + 1];
            idx[k] = ((((u0 & 1u) * 2 + ((u0 >> 16) & 1u)) * 2 + (u1 & 1u)) * 2 + ((u1 >> 16) & 1u));
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
            quads += ((lane + 64 * k) < nquad) ? qt->t3233[idx[k]] : 0u;
    }
    LH_PA(20, t_nq);
    {
        /* region maxima: three independent reductions the scheduler can interleave */
        unsigned m0 = 0, m1 = 0, m2 = 0;
        unsigned w00, w01, w10, w11, w20, w21;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (k < 2 || 64 * k < e2) {         /* wave-uniform: blocks above big_values hold nothing */
                int const p = lane + 64 * k;
                unsigned const lo = pk[k] & 0xffffu, hi = pk[k] >> 16;
                unsigned const m = lo > hi ? lo : hi;
                if (p < e0)
                    m0 = m > m0 ? m : m0;
                else if (p < e1)
                    m1 = m > m1 ? m : m1;
                else if (p < e2)
                    m2 = m > m2 ? m : m2;
            }
        }
        m0 = lh_wave_max_u32(m0);
        m1 = lh_wave_max_u32(m1);
        m2 = lh_wave_max_u32(m2);
        LH_PA(21, t_nq);
        {
            /* One look-up sequence per pair, not per (region, pair): each lane selects its
             * region's parameters, all loads of the (up to) five pairs are in flight together
             * and the results are added to the accumulators of the pair's region.  Blocks of
             * 64 pairs above big_values are skipped (wave-uniform). */
            LhRegionLut l0, l1, l2;     /* = lh_region_lut(m), tabulated in LDS */
            l0 = lh_region_lut(m0);
            l1 = lh_region_lut(m1);
            l2 = lh_region_lut(m2);
            unsigned v0[5], v1[5];
            w00 = w01 = w10 = w11 = w20 = w21 = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                v0[k] = v1[k] = 0;
                if (k < 2 || 64 * k < e2) {
                    int const p = lane + 64 * k;
                    uint32_t const pa = (p < e0) ? l0.pa : (p < e1) ? l1.pa : (p < e2) ? l2.pa : 2u;
                    uint32_t const pb = (p < e0) ? l0.pb : (p < e1) ? l1.pb : (p < e2) ? l2.pb : 0u;
                    unsigned const x = pk[k] & 0xffffu, y = pk[k] >> 16;
                    unsigned const xc = x < 15u ? x : 15u, yc = y < 15u ? y : 15u;
                    unsigned const idx = xc * (pa & 0xffu) + yc;
                    unsigned const b1 = qt->ht_len[(pa >> 16) + idx];
                    unsigned const b2 = qt->ht_len[(pb & 0xffffu) + idx];
                    unsigned const b3 = qt->ht_len[(pb >> 16) + idx];
                    unsigned const e = lh_largetbl[idx & 255u];
                    int const esc = (pa >> 8) & 1u;
                    v0[k] = esc ? e : (b1 | (b2 << 16));
                    v1[k] = esc ? (unsigned) (x >= 15u) + (unsigned) (y >= 15u) : b3;
                }
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                int const p = lane + 64 * k;
                int const r0 = (p < e0), r1 = (p >= e0 && p < e1), r2 = (p >= e1 && p < e2);
                if (!(k < 2 || 64 * k < e2))
                    continue;
                w00 += r0 ? v0[k] : 0u;
                w01 += r0 ? v1[k] : 0u;
                w10 += r1 ? v0[k] : 0u;
                w11 += r1 ? v1[k] : 0u;
                w20 += r2 ? v0[k] : 0u;
                w21 += r2 ? v1[k] : 0u;
            }
        }
        LH_PA(22, t_nq);
        /* independent reductions, back to back so that their steps interleave */
        quads = lh_wave_sum_u32(quads);
        w00 = lh_wave_sum_u32(w00);
        w10 = lh_wave_sum_u32(w10);
        w20 = lh_wave_sum_u32(w20);
        {
            unsigned const wa = lh_wave_sum_u32(w01 | (w11 << 16));
            w21 = lh_wave_sum_u32(w21);
            w01 = wa & 0xffffu;
            w11 = wa >> 16;
        }
        LH_PA(23, t_nq);
        {
            int const c1a = (int) (quads >> 16), c1b = (int) (quads & 0xffffu);
            bits = c1a;
            g.count1table_select = 0;
            if (c1a > c1b) {
                bits = c1b;
                g.count1table_select = 1;
            }
            g.count1bits = bits;
        }
        if (bv == 0)
            return bits;
        /* same order as the reference: region 2 (long blocks), then 0, then 1 */
        if (e1 < e2)
            g.table_select[2] = lh_region_decide(m2, w20, w21, &bits);
        if (0 < e0)
            g.table_select[0] = lh_region_decide(m0, w00, w01, &bits);
        if (e0 < e1)
            g.table_select[1] = lh_region_decide(m1, w10, w11, &bits);
    }
    /* use_best_huffman == 2 (best_huffman_divide inside the loop) is not selected by any quality level */
    if (use_prev && R.block_type == LH_NORM_TYPE) {
        /* first band whose start is >= big_values (band starts ascend) */
        R.pn_sfb_count1 = lh_popc64(lh_ballot(lane < LH_SBMAX_L + 1 && (int) sfbcnt_in < bv));
    }
    return bits;
}

/* ---------------------------------------------------------------------- */
/* MPEG-2 / 2.5 (reference takehiro.c:1195-1317, mpeg2_scale_bitcount): the scalefactors travel in four partitions of
 * fixed sizes (ISO 13818-3 2.4.3.2), each with its own field width = the bit length of the partition's largest value;
 * table 0 without preflag (6 5 5 5 long bands / 9 9 9 9 short values; ranges 15 15 7 7), table 2 with it (11 10 / 18 18;
 * ranges 7 3).  Lane s holds scalefactor s (`v').  Each lane contributes the thermometer code of its value's bit length
 * in its partition's byte, one OR over the wave gives all four maxima's bit lengths; the rest is scalar.  Returns the
 * number of partitions over their range (then g stays as it is, as in the reference). */
LH_DEVFN int
lh_scale_bitcount_lsf(const LhCtx & c, const LhQR & R, LhGrR & g, int v)
{
    int const sh = (R.block_type == LH_SHORT_TYPE);
    /* partition ends (in scalefactor values) and range bit lengths, one byte each */
    uint32_t const ends = g.preflag ? (sh ? 0x24242412u : 0x1515150bu) : (sh ? 0x241b1209u : 0x15100b06u);
    uint32_t const rbits = g.preflag ? 0x00000203u : 0x03030404u;
    int const s = c.lane;
    int const part = (s >= (int) (ends & 255u)) + (s >= (int) ((ends >> 8) & 255u)) + (s >= (int) ((ends >> 16) & 255u));
    int     bl = 32 - lh_clz32((uint32_t) (v > 0 ? v : 0));
    uint32_t th;
    int     over = 0, slen[4], n[4], p, prev = 0;
    bl = bl > 8 ? 8 : bl;
    th = lh_wave_or_u32((s < (int) (ends >> 24)) ? (((1u << bl) - 1u) << (8 * part)) : 0u);
#pragma unroll
    for (p = 0; p < 4; p++) {
        int const e = (int) ((ends >> (8 * p)) & 255u);
        slen[p] = lh_popc64((uint64_t) ((th >> (8 * p)) & 255u));
        n[p] = e - prev;
        prev = e;
        over += slen[p] > (int) ((rbits >> (8 * p)) & 255u);
    }
    if (!over) {
        g.scalefac_compress = g.preflag ? 500 + slen[0] * 3 + slen[1]
            : (((slen[0] * 5) + slen[1]) << 4) + (slen[2] << 2) + slen[3];
        g.part2_length = slen[0] * n[0] + slen[1] * n[1] + slen[2] * n[2] + slen[3] * n[3];
    }
    return over;
}

/* reference takehiro.c:1135-1188 (MPEG-1); band s on lane s, maxima by wave reduction */
LH_DEVFN int
lh_scale_bitcount(const LhCtx & c, LhChanLds & Q, const LhQR & R, LhGrR & g, int which)
{
    const LhQTabs *qt = LH_QT;
    int    *sf = Q.sf[which];
    int     k, max_slen1, max_slen2;
    int     v;
    LH_WAVE_SYNC();
    v = (c.lane < R.sfbmax) ? sf[c.lane] : 0;
    if (LH_IS_LSF)
        return lh_scale_bitcount_lsf(c, R, g, v);
    if (R.block_type != LH_SHORT_TYPE) {
        if (!g.preflag) {
            int const inr = (c.lane >= 11 && c.lane < LH_SBPSY_L);
            uint64_t const below = lh_ballot(inr && v < (int) qt->pretab[inr ? c.lane : 0]);
            if (below == 0) {
                g.preflag = 1;
                if (inr) {
                    v -= qt->pretab[c.lane];
                    sf[c.lane] = v;
                }
                LH_WAVE_SYNC();
            }
        }
    }
    max_slen1 = (int) lh_wave_max_u32((c.lane < R.sfbdivide && v > 0) ? (unsigned) v : 0u);
    max_slen2 = (int) lh_wave_max_u32((c.lane >= R.sfbdivide && c.lane < R.sfbmax && v > 0) ? (unsigned) v : 0u);
    /* the reference scans the 16 (slen1, slen2) pairs in order and keeps the first strictly
     * smaller size (takehiro.c:1184-1196): the minimum of (size, index), one lane per pair */
    g.part2_length = LH_LARGE_BITS;
    k = c.lane & 15;
    {
        unsigned key = 0xffffffffu, best;
        /* slen1 / slen2 of pair k as nibble tables (reference takehiro.c:1102-1126); the
         * table sizes are 11 slen1 + 10 slen2 (long) and 18 (slen1 + slen2) (short) */
        int const s1 = (int) ((0x4433322211130000ull >> (4 * k)) & 15u);
        int const s2 = (int) ((0x3232132132103210ull >> (4 * k)) & 15u);
        int const sz = (R.block_type == LH_SHORT_TYPE) ? 18 * (s1 + s2) : 11 * s1 + 10 * s2;
        if (c.lane < 16 && max_slen1 < (1 << s1) && max_slen2 < (1 << s2))
            key = ((unsigned) sz << 8) | (unsigned) k;
        best = lh_wave_min_u32(key);
        if (best != 0xffffffffu) {
            g.part2_length = (int) (best >> 8);
            g.scalefac_compress = (int) (best & 255u);
        }
    }
    return g.part2_length == LH_LARGE_BITS;
}

/* reference quantize_pvt.c:554-573 */
/* logt_lds (wave-uniform): calc_noise's copy of the logarithm's table is in LDS (every loop but the new VBR one, whose step
 * tables lie there): the two look-ups are LDS round trips instead of trips to the vector cache */
LH_DEVFN float
lh_ath_adjust(const LhTables * T, float a, float x, float athFloor, float ATHfixpoint, int logt_lds = 0)
{
    float const o = 90.30873362f;
    float const p = (ATHfixpoint < 1.f) ? 94.82444863f : ATHfixpoint;
    float const v = a * a;
    float   lx, lv = 0.0f;
    if (logt_lds) {
        LH_FAST_LOG2_VIA(LH_LOGT_LDS, x, lx);
        if (v > 1E-20f)
            LH_FAST_LOG2_VIA(LH_LOGT_LDS, v, lv);
    }
    else {
        lx = lh_fast_log2(T->log_table, x);
        if (v > 1E-20f)
            lv = lh_fast_log2(T->log_table, v);
    }
    float   u = (float) (lx * (LH_LOG2_OVER_LOG10 * (10.0f)));
    float   w = 0.0f;
    u -= athFloor;
    if (v > 1E-20f)
        w = (float) (1.f + lv * (LH_LOG2_OVER_LOG10 * (10.0f / o)));
    if (w < 0)
        w = 0.f;
    u *= w;
    u += athFloor + o - p;
    return lh_powf(10.f, 0.1f * u);
}

/* reference quantize_pvt.c:589-747: one lane per scalefactor band (window) */
LH_DEVFN void
lh_calc_xmin_body(const LhCtx & c, LhChanLds & Q, LhQR & R, const float *xr, const float *ren, const float *rthm)
{
    const LhConfig *cfg = c.cfg;
    const LhTables *T = c.T;
    const LhQTabs *qt = LH_QT;
    int const s = c.lane;
    float const adj = lh_lds.ss.ath_adjust_factor;
    int const logt_lds = lh_uni_i(!(cfg->vbr == 1 || cfg->vbr == 4));
    int     over = 0;
    if (s < R.psymax) {
        int const is_long = (s < R.psy_lmax);
        int const sfb = is_long ? s : (R.sfb_smin + (s - R.psy_lmax) / 3);
        int const b = is_long ? 0 : (s - R.psy_lmax) % 3;
        float   en0 = 0.0f, xmin, ath, rh1, rh2, rh3, fact, e, t;
        int const width = Q.width[s];
        int     j = Q.start[s];
        if (is_long) {
            ath = lh_ath_adjust(T, adj, T->ath_l[sfb], T->ath_floor, cfg->ATHfixpoint, logt_lds);
            fact = T->longfact[sfb];
            e = ren[sfb];
            t = rthm[sfb];
        }
        else {
            ath = lh_ath_adjust(T, adj, T->ath_s[sfb], T->ath_floor, cfg->ATHfixpoint, logt_lds);
            fact = T->shortfact[sfb];
            e = ren[22 + sfb * 3 + b];
            t = rthm[22 + sfb * 3 + b];
        }
        ath *= fact;
        rh1 = ath / width;
        rh2 = (float) 2.2204460492503131e-16;
        {
            /* the two sums are serial (the reference's order); band starts and widths are even in
             * every MPEG-1 table, so lines come in aligned pairs, four pairs per trip with their
             * loads issued ahead of the dependent adds, nothing conditional on the add chains */
            const lh_f32x2 *xp = (const lh_f32x2 *) (xr + j);
            int const np = width >> 1;
            int     p = 0;
            /* (the next trip's four pairs are read before this trip's additions; the last trip reads four pairs
             * beyond the band that nobody adds) */
            lh_f32x2 n0 = xp[0], n1 = xp[1], n2 = xp[2], n3 = xp[3];
            for (; p + 4 <= np; p += 4) {
                lh_f32x2 const a0 = n0, a1 = n1, a2 = n2, a3 = n3;
                n0 = xp[p + 4];
                n1 = xp[p + 5];
                n2 = xp[p + 6];
                n3 = xp[p + 7];
#ifndef LH_EMU
                {
                    /* one statement, so that the reads above stay ahead of it; min(x2, rh1) is (x2 < rh1) ? x2 : rh1 for
                     * every pair of floats (equal values have equal bits here: both are positive) */
                    float   t_;
#define LH_XM_TERM(X) "v_mul_f32 %[t], %[" #X "], %[" #X "]\n\tv_add_f32 %[en], %[en], %[t]\n\tv_min_f32 %[t], %[t], %[rh1]\n\tv_add_f32 %[rh2], %[rh2], %[t]\n\t"
                    asm volatile(LH_XM_TERM(a0) LH_XM_TERM(a1) LH_XM_TERM(a2) LH_XM_TERM(a3)
                                 LH_XM_TERM(a4) LH_XM_TERM(a5) LH_XM_TERM(a6) LH_XM_TERM(a7)
                                 : [en] "+v"(en0), [rh2] "+v"(rh2), [t] "=&v"(t_)
                                 : [rh1] "v"(rh1), [a0] "v"(a0.x), [a1] "v"(a0.y), [a2] "v"(a1.x), [a3] "v"(a1.y),
                                   [a4] "v"(a2.x), [a5] "v"(a2.y), [a6] "v"(a3.x), [a7] "v"(a3.y));
#undef LH_XM_TERM
                }
#else
                float const v[8] = { a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y };
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    float const x2 = v[u] * v[u];
                    en0 += x2;
                    rh2 += (x2 < rh1) ? x2 : rh1;
                }
#endif
            }
            for (; p < np; p++) {
                lh_f32x2 const a0 = xp[p];
                float   x2 = a0.x * a0.x;
                en0 += x2;
                rh2 += (x2 < rh1) ? x2 : rh1;
                x2 = a0.y * a0.y;
                en0 += x2;
                rh2 += (x2 < rh1) ? x2 : rh1;
            }
        }
        if (en0 < ath)
            rh3 = en0;
        else if (rh2 < ath)
            rh3 = ath;
        else
            rh3 = rh2;
        xmin = rh3;
        if (e > 1e-12f) {
            float   x = en0 * t / e;
            x *= fact;
            if (xmin < x)
                xmin = x;
        }
        xmin = (float) ((xmin > 2.2204460492503131e-16) ? xmin : 2.2204460492503131e-16);
        Q.l3_xmin[s] = xmin;
        Q.sfb_mode[s] = (en0 > xmin + 1e-14f) ? 1 : 0;  /* energy_above_cutoff (read by the VBR loop only) */
        over = (en0 > ath);
    }
    R.ath_over = (lh_ballot(over) != 0);
    {
        /* highest non-zero coefficient */
        unsigned top = 0;
        int     max_nonzero;
        for (int k = c.lane; k < 576; k += 64)
            if (k > 0 && lh_fabsf(xr[k]) > 1e-12f)
                top = (unsigned) k;
        top = lh_wave_max_u32(top);
        max_nonzero = (int) top;
        if (R.block_type != LH_SHORT_TYPE)
            max_nonzero |= 1;
        else {
            max_nonzero /= 6;
            max_nonzero *= 6;
            max_nonzero += 5;
        }
        if (c.sfb21_extra == 0 && cfg->samplerate < 44000) {
            int     limit;
            if (R.block_type != LH_SHORT_TYPE)
                limit = qt->sfb_l[LH_RATE8K(c) ? 17 : 21] - 1;
            else
                limit = 3 * T->sfb_s[LH_RATE8K(c) ? 9 : 12] - 1;
            if (max_nonzero > limit)
                max_nonzero = limit;
        }
        R.mnc = max_nonzero;
    }
    LH_WAVE_SYNC();
    if (cfg->use_temporal_masking && s < R.psymax && s >= R.psy_lmax && ((s - R.psy_lmax) % 3) == 0) {
        float   x0 = Q.l3_xmin[s], x1 = Q.l3_xmin[s + 1], x2 = Q.l3_xmin[s + 2];
        if (x0 > x1)
            x1 += (x0 - x1) * T->decay;
        if (x1 > x2)
            x2 += (x1 - x2) * T->decay;
        Q.l3_xmin[s + 1] = x1;
        Q.l3_xmin[s + 2] = x2;
    }
    LH_WAVE_SYNC();
}

/* out-of-line entry: R / g travel through the wave's LDS slot (lh_rg_put), the body works on
 * scalar copies */
LH_STAGEFN void
lh_calc_xmin(int qch, int gr, int rch)
{
    LhCtx const c = lh_ctx_load();
    LhQR    R = lh_uniform(lh_lds.rg[qch].R);
    {
        int const slot = (lh_uni_i(lh_lds.psy_slot) + gr) % 3;   /* the granule's ratios: see LhLds.psy_en */
        lh_calc_xmin_body(c, lh_lds.u.quant.ch[qch], R, lh_lds.xr[qch][gr], lh_lds.psy_en[slot][rch],
                          lh_lds.psy_thm[slot][rch]);
    }
    if (c.lane == 0)
        lh_lds.rg[qch].R = R;
    LH_WAVE_SYNC();
}

/* the same for the usual case (lh_granule_is_usual): every band a long one, no short-block tail */
LH_STAGEFN void
lh_calc_xmin_n(int qch, int gr, int rch)
{
    LhCtx   c = lh_ctx_load();
    LhQR    R = lh_uniform(lh_lds.rg[qch].R);
    lh_pin_usual(c, 0);
    lh_pin_usual(R);
    {
        int const slot = (lh_uni_i(lh_lds.psy_slot) + gr) % 3;
        lh_calc_xmin_body(c, lh_lds.u.quant.ch[qch], R, lh_lds.xr[qch][gr], lh_lds.psy_en[slot][rch],
                          lh_lds.psy_thm[slot][rch]);
    }
    if (c.lane == 0)
        lh_lds.rg[qch].R = R;
    LH_WAVE_SYNC();
}

/* ---------------------------------------------------------------------- */
/* geometry of the granule + spectrum re-ordering for short blocks
 * (reference quantize.c:226-346) */
LH_DEVFN void
lh_init_outer_loop_body(const LhCtx & c, LhChanLds & Q, LhQR & R, LhGrR & g, float *xr, int block_type, int substep,
                        int reorder = 1)
{
    const LhTables *T = c.T;
    const LhQTabs *qt = LH_QT;
    int const sfb21 = c.sfb21_extra;
    g.part2_3_length = 0;
    g.big_values = 0;
    g.count1 = 0;
    g.global_gain = 210;
    g.scalefac_compress = 0;
    g.table_select[0] = g.table_select[1] = g.table_select[2] = 0;
    g.subblock_gain[0] = g.subblock_gain[1] = g.subblock_gain[2] = g.subblock_gain[3] = 0;
    g.region0_count = g.region1_count = 0;
    g.preflag = 0;
    g.scalefac_scale = 0;
    g.count1table_select = 0;
    g.part2_length = 0;
    g.count1bits = 0;
    g.xrpow_max = 0;
    R.block_type = block_type;
    /* (an 8 kHz stream codes 17 long / 9 short bands: reference quantize.c:252-256, 284-294) */
    R.sfb_lmax = LH_RATE8K(c) ? 17 : LH_SBPSY_L;
    R.sfb_smin = LH_RATE8K(c) ? 9 : LH_SBPSY_S;
    R.psy_lmax = LH_RATE8K(c) ? 17 : (sfb21 ? LH_SBMAX_L : LH_SBPSY_L);
    R.psymax = R.psy_lmax;
    R.sfbmax = R.sfb_lmax;
    R.sfbdivide = 11;
    if (block_type == LH_SHORT_TYPE) {
        R.sfb_smin = 0;
        R.sfb_lmax = 0;
        R.psymax = LH_RATE8K(c) ? 3 * 9 : 3 * ((sfb21 ? LH_SBMAX_S : LH_SBPSY_S));
        R.sfbmax = LH_RATE8K(c) ? 3 * 9 : 3 * LH_SBPSY_S;
        R.sfbdivide = R.sfbmax - 18;
        R.psy_lmax = 0;
    }
    R.mnc = 575;
    R.pn_global_gain = 0;
    R.pn_sfb_count1 = 0;
    R.substep_shaping = substep;
    R.ath_over = 0;
    R.s_mnc = 0;
    LH_WAVE_SYNC();
    if (c.lane <= LH_SFBMAX) {
        int const s = c.lane;
        Q.sf[0][s] = 0;
        Q.sf[1][s] = 0;
        if (s < 2)
            Q.zero2[s] = 0.0f;
        if (s == LH_SFBMAX) {
            Q.width[s] = 0;
            Q.window[s] = 3;
            Q.start[s] = 576;
        }
        else if (block_type == LH_SHORT_TYPE) {
            int const sfb = s / 3, win = s - 3 * sfb;
            int const wd = T->sfb_s[sfb + 1] - T->sfb_s[sfb];
            Q.width[s] = wd;
            Q.window[s] = win;
            Q.start[s] = 3 * T->sfb_s[sfb] + win * wd;
        }
        else if (s < LH_SBMAX_L) {
            Q.width[s] = qt->sfb_l[s + 1] - qt->sfb_l[s];
            Q.window[s] = 3;
            Q.start[s] = qt->sfb_l[s];
        }
        else {
            Q.width[s] = 0;
            Q.window[s] = 3;
            Q.start[s] = 576;
        }
    }
    LH_WAVE_SYNC();
    {
        /* band of every line: constant per block type, built by the host (LhTables.sfb_line_*) */
        const uint32_t *src = (const uint32_t *) ((block_type == LH_SHORT_TYPE) ? T->sfb_line_s : T->sfb_line_l);
        uint32_t *dst = (uint32_t *) Q.sfb_of_line;
        uint32_t const a = src[c.lane], b = src[c.lane + 64], d = src[(c.lane < 16) ? c.lane + 128 : 0];
        dst[c.lane] = a;
        dst[c.lane + 64] = b;
        if (c.lane < 16)
            dst[c.lane + 128] = d;
    }
    if (block_type == LH_SHORT_TYPE && reorder) {
        /* window-major re-ordering inside each short band */
        float  *tmp = Q.save_xrpow;
        for (int i = c.lane; i < 576; i += 64)
            tmp[i] = xr[i];
        LH_WAVE_SYNC();
        for (int d = c.lane; d < 576; d += 64) {
            int     sfb = 0;
            for (int k = 1; k < LH_SBMAX_S; k++)
                if (3 * T->sfb_s[k] <= d)
                    sfb = k;
            {
                int const wd = T->sfb_s[sfb + 1] - T->sfb_s[sfb];
                int const r = d - 3 * T->sfb_s[sfb];
                int const win = r / wd, l = T->sfb_s[sfb] + (r - win * wd);
                xr[d] = tmp[3 * l + win];
            }
        }
    }
    LH_WAVE_SYNC();
}

LH_STAGEFN void
lh_init_outer_loop(int qch, int gr, int block_type, int substep, int reorder = 1)
{
    LhCtx const c = lh_ctx_load();
    LhQR    R;
    LhGrR   g;
    lh_init_outer_loop_body(c, lh_lds.u.quant.ch[qch], R, g, lh_lds.xr[qch][gr], lh_uni_i(block_type),
                            lh_uni_i(substep), lh_uni_i(reorder));
    lh_rg_put(c, R, g);
}

LH_STAGEFN void
lh_init_outer_loop_n(int qch, int gr, int substep)
{
    LhCtx   c = lh_ctx_load();
    LhQR    R;
    LhGrR   g;
    lh_pin_usual(c, 0);
    lh_init_outer_loop_body(c, lh_lds.u.quant.ch[qch], R, g, lh_lds.xr[qch][gr], LH_NORM_TYPE, lh_uni_i(substep), 0);
    lh_rg_put(c, R, g);
}

/* Lines above max_nonzero_coeff are zero in every quantised image (the reference clears
 * them inside each quantize_xrpow call, takehiro.c:296-301, 386-391); here they are
 * cleared once per granule, after lh_calc_xmin fixed R.mnc, and never written again. */
LH_DEVFN void
lh_zero_tail(const LhCtx & c, LhChanLds & Q, const LhQR & R)
{
    for (int i = c.lane; i < 576; i += 64)
        if (i > R.mnc)
            Q.ix[0][i] = 0;
    LH_WAVE_SYNC();
}

/* reference quantize.c:72-144; returns 1 when there is energy to quantise */
LH_DEVFN int
lh_init_xrpow(const LhCtx & c, LhChanLds & Q, const LhQR & R, LhGrR & g, const float *xr)
{
    unsigned mxabs = 0, mxpow = 0;
    int     nonzero;
    /* lines above max_nonzero_coeff count as zero (reference quantize.c:78-83: upper); the CBR
     * loop calls this before calc_xmin, with mnc still 575 */
    for (int i = c.lane; i < 576; i += 64) {
        float const tmp = (i <= R.mnc) ? lh_fabsf(xr[i]) : 0.0f;
        float const xp = (float) sqrt((double) tmp * sqrt((double) tmp));
        unsigned const ub = lh_f32_as_u32(tmp), up = lh_f32_as_u32(xp);
        Q.xrpow[i] = xp;
        mxabs = ub > mxabs ? ub : mxabs;
        mxpow = up > mxpow ? up : mxpow;
    }
    mxabs = lh_wave_max_u32(mxabs);
    mxpow = lh_wave_max_u32(mxpow);
    g.xrpow_max = lh_u32_as_f32(mxpow);
    {
        /* the reference sums |xr| serially and tests sum > 1e-20.  All terms are
         * non-negative, so max <= sum <= 576*max*(1+2^-23)^576: only inside that
         * band does the serial order matter, and there it is evaluated serially. */
        float const mx = lh_u32_as_f32(mxabs);
        if (mx > 1E-20f)
            nonzero = 1;
        else if (mx * 577.0f <= 1E-20f)
            nonzero = 0;
        else {
            float   sum = 0;
            LH_WAVE_SYNC();
            for (int i = 0; i <= R.mnc; i++)
                sum += lh_fabsf(xr[i]);
            nonzero = sum > (float) 1E-20;
        }
    }
    LH_WAVE_SYNC();
    if (nonzero) {
        int const j = (R.substep_shaping & 2) ? 1 : 0;
        if (c.lane < R.psymax)
            Q.pseudohalf[c.lane] = j;
        LH_WAVE_SYNC();
        return 1;
    }
    for (int i = c.lane; i < 576; i += 64)
        Q.ix[0][i] = 0;
    LH_WAVE_SYNC();
    return 0;
}

/* reference quantize.c:585-686 (comparator 9, the only one the presets of this path select) */
LH_DEVFN int
lh_quant_compare(const LhNoiseRes & best, const LhNoiseRes & calc)
{
#if !defined(LH_EMU)
    /* Scalar integers throughout: every operand is wave-uniform, and truth values the compiler keeps as lane masks are
     * merged with vector selects (a dozen instructions of this rule, once per iteration of the search).  The one float
     * comparison runs on the vector unit -- there is no scalar one -- and comes back through v_readfirstlane. */
    unsigned const fewer = (calc.bits < best.bits) ? 1u : 0u;
    if (best.over_count > 0) {
        unsigned const le = (calc.over_SSD <= best.over_SSD) ? 1u : 0u;
        return (int) ((calc.over_SSD == best.over_SSD) ? fewer : le);
    }
    return (int) ((unsigned) lh_uni_i((calc.max_noise < 0) &&
                                      ((calc.max_noise * 10 + calc.bits) <= (best.max_noise * 10 + best.bits))) & fewer);
#else
    int     better;
    if (best.over_count > 0) {
        better = calc.over_SSD <= best.over_SSD;
        if (calc.over_SSD == best.over_SSD)
            better = calc.bits < best.bits;
    }
    else {
        better = ((calc.max_noise < 0) &&
                  ((calc.max_noise * 10 + calc.bits) <= (best.max_noise * 10 + best.bits)));
    }
    if (best.over_count == 0)
        better = better && calc.bits < best.bits;
    return better;
#endif
}

/* ---------------------------------------------------------------------- */
/* reference takehiro.c:964-1094; gr0's final scalefactors come from the output slot
 * g0sf (int8, -1 = shared).  scfsi_out[4] is written by lane 0. */
LH_DEVFN void
lh_best_scalefac_store_body(const LhCtx & c, LhChanLds & Q, const LhQR & R, LhGrR & g, int gr,
                            const int8_t * g0sf, int g0_block_type, int *scfsi_out)
{
    /* reference takehiro.c:1021-1094 (+ scfsi_calc :964-1019).  Lane = scalefactor band; every
     * "for all bands" test of the reference is a ballot, its maxima / minima are wave
     * reductions; the band's scalefactor lives in a register (sfv) until the end. */
    const LhQTabs *qt = LH_QT;
    int    *sf = Q.sf[0];
    const uint32_t *ix2 = (const uint32_t *) Q.ix[0];
    int const s = c.lane;
    int const inband = (s < R.sfbmax);
    int     sfv;
    int     recalc = 0;
    int     scfsi[4] = { 0, 0, 0, 0 };
    LH_PT(t_bs0);
    LH_WAVE_SYNC();
    sfv = inband ? sf[s] : 0;
    /* bands whose lines are all zero get the wildcard -2: every non-zero pair flags its band */
    if (s <= LH_SFBMAX)
        Q.sfb_mode[s] = 0;
    LH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 5; k++) {
        int const p = s + 64 * k;
        int const pc = (k < 4 || p < 288) ? p : 287;
        uint32_t const v = ix2[pc];
        int const band = Q.sfb_of_line[2 * pc];
        if ((k < 4 || p < 288) && v != 0u)
            Q.sfb_mode[band] = 1;
    }
    LH_WAVE_SYNC();
    LH_PA(40, t_bs0);
    if (inband && !Q.sfb_mode[s])
        sfv = -2;
    if (lh_ballot(inband && sfv == -2))
        recalc = -2;
    if (!g.scalefac_scale && !g.preflag) {
        uint64_t const pos = lh_ballot(inband && sfv > 0);
        uint64_t const odd = lh_ballot(inband && sfv > 0 && (sfv & 1));
        if (pos && !odd) {
            if (inband && sfv > 0)
                sfv >>= 1;
            g.scalefac_scale = recalc = 1;
        }
    }
    if (!g.preflag && R.block_type != LH_SHORT_TYPE && !LH_IS_LSF) {
        int const hi = (s >= 11 && s < LH_SBPSY_L);
        int const pre = qt->pretab[s < 22 ? s : 0];
        if (!lh_ballot(hi && sfv < pre && sfv != -2)) {
            if (hi && sfv > 0)
                sfv -= pre;
            g.preflag = recalc = 1;
        }
    }
    LH_PA(41, t_bs0);
    if (gr == 1 && !LH_IS_LSF && g0_block_type != LH_SHORT_TYPE && R.block_type != LH_SHORT_TYPE) {
        /* scfsi_calc: share a group of scalefactors with granule 0 when all of them agree */
        int const in21 = (s < LH_SBPSY_L);
        int const g0 = in21 ? (int) g0sf[s] : 0;
        uint64_t const mism = lh_ballot(in21 && g0 != sfv && sfv >= 0);
        int     grp = 0, c1, c2;
        unsigned s1, s2, key, best;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint64_t const gm = ((1ull << lh_scfsi_band[i + 1]) - 1ull) & ~((1ull << lh_scfsi_band[i]) - 1ull);
            scfsi[i] = !(mism & gm);
            if (i > 0 && s >= lh_scfsi_band[i])
                grp = i;
        }
        if (in21 && (grp == 0 ? scfsi[0] : grp == 1 ? scfsi[1] : grp == 2 ? scfsi[2] : scfsi[3]))
            sfv = -1;
        c1 = lh_popc64(lh_ballot(s < 11 && sfv != -1));
        c2 = lh_popc64(lh_ballot(s >= 11 && in21 && sfv != -1));
        s1 = lh_wave_max_u32((s < 11 && sfv > 0) ? (unsigned) sfv : 0u);
        s2 = lh_wave_max_u32((s >= 11 && in21 && sfv > 0) ? (unsigned) sfv : 0u);
        /* the reference takes the first strictly smaller candidate in ascending order: the
         * minimum of (bits, index) */
        key = 0xffffffffu;
        {
            int const k16 = s & 15;
            int const l1 = (int) ((0x4433322211130000ull >> (4 * k16)) & 15u);
            int const l2 = (int) ((0x3232132132103210ull >> (4 * k16)) & 15u);
            if (s < 16 && (int) s1 < (1 << l1) && (int) s2 < (1 << l2))
                key = ((unsigned) (l1 * c1 + l2 * c2) << 8) | (unsigned) s;
        }
        best = lh_wave_min_u32(key);
        if (best != 0xffffffffu && g.part2_length > (int) (best >> 8)) {
            g.part2_length = (int) (best >> 8);
            g.scalefac_compress = (int) (best & 255u);
        }
        recalc = 0;
    }
    if (inband) {
        if (sfv == -2)
            sfv = 0;
        sf[s] = sfv;
    }
    LH_WAVE_SYNC();
    LH_PA(42, t_bs0);
    if (recalc)
        (void) lh_scale_bitcount(c, Q, R, g, 0);
    LH_PA(43, t_bs0);
    if (c.lane == 0)
        for (int i = 0; i < 4; i++)
            scfsi_out[i] = scfsi[i];
}

LH_STAGEFN void
lh_best_scalefac_store(int qch, int gr, const int8_t * g0sf, int g0_block_type)
{
    LhCtx const c = lh_ctx_load();
    LhQR const R = lh_uniform(lh_lds.rg[qch].R);
    LhGrR   g = lh_uniform(lh_lds.rg[qch].g);
    lh_best_scalefac_store_body(c, lh_lds.u.quant.ch[qch], R, g, lh_uni_i(gr), LH_AS_GLOBAL(const int8_t, g0sf),
                                lh_uni_i(g0_block_type), lh_lds.scfsi[qch]);
    lh_rg_put(c, R, g);
}

LH_STAGEFN void
lh_best_scalefac_store_n(int qch, int gr, const int8_t * g0sf, int g0_block_type)
{
    LhCtx   c = lh_ctx_load();
    LhQR    R = lh_uniform(lh_lds.rg[qch].R);
    LhGrR   g = lh_uniform(lh_lds.rg[qch].g);
    lh_pin_usual(c, 0);
    lh_pin_usual(R);
    lh_best_scalefac_store_body(c, lh_lds.u.quant.ch[qch], R, g, lh_uni_i(gr), LH_AS_GLOBAL(const int8_t, g0sf),
                                lh_uni_i(g0_block_type), lh_lds.scfsi[qch]);
    lh_rg_put(c, R, g);
}

/* reference takehiro.c:809-957.  The reference evaluates up to 16 + 128 region
 * splits with serial choose_table calls; here every candidate split is costed
 * by its own lane (serial scan of its region in LDS), then the reference's
 * first-minimum selection is replayed wave-uniformly. */
/* ---- per-band Huffman length sums for best_huffman_divide -------------------------------
 * Every region the search looks at is a run of whole scalefactor bands (the last one cut at
 * big_values), and a region's bit count for a table is an integer sum over its pairs.  So the
 * pairs are visited once: each adds its code length for every table that can hold it to the
 * accumulator of its band (LDS atomics), a prefix sum over the bands follows, and any region's
 * choose_table (takehiro.c:546-650) becomes a difference of prefix sums per candidate table.
 * Ten words per band: seven hold two tables each (16 bits per half: a band has < 2^16 bits),
 * then the ESC pair (largetbl layout), the count of values >= 15 and the count of non-zero pairs. */
#define LH_BHD_NW 10
#define LH_BHD_STRIDE 24

LH_DEVFN void
lh_bhd_pair_words(const LhQTabs * qt, unsigned x, unsigned y, unsigned w[LH_BHD_NW])
{
    unsigned const m = x > y ? x : y;
    unsigned const xc = x < 15u ? x : 15u, yc = y < 15u ? y : 15u;
    unsigned const i2 = xc * 2u + yc, i3 = xc * 3u + yc, i4 = xc * 4u + yc, i6 = xc * 6u + yc, i8 = xc * 8u + yc,
        i16 = xc * 16u + yc;
    const uint8_t *h = qt->ht_len;
    unsigned const t1 = h[lh_ht_off(1) + i2], t2 = h[lh_ht_off(2) + i3], t3 = h[lh_ht_off(3) + i3];
    unsigned const t5 = h[lh_ht_off(5) + i4], t6 = h[lh_ht_off(6) + i4];
    unsigned const t7 = h[lh_ht_off(7) + i6], t8 = h[lh_ht_off(8) + i6], t9 = h[lh_ht_off(9) + i6];
    unsigned const t10 = h[lh_ht_off(10) + i8], t11 = h[lh_ht_off(11) + i8], t12 = h[lh_ht_off(12) + i8];
    unsigned const t13 = h[lh_ht_off(13) + i16], t14 = h[lh_ht_off(14) + i16], t15 = h[lh_ht_off(15) + i16];
    unsigned const v2 = m < 2u, v3 = m < 3u, v4 = m < 4u, v6 = m < 6u, v8 = m < 8u, v16 = m < 16u;
    w[0] = (v2 ? t1 : 0u) | ((v3 ? t2 : 0u) << 16);
    w[1] = (v3 ? t3 : 0u) | ((v4 ? t5 : 0u) << 16);
    w[2] = (v4 ? t6 : 0u) | ((v6 ? t7 : 0u) << 16);
    w[3] = (v6 ? t8 : 0u) | ((v6 ? t9 : 0u) << 16);
    w[4] = (v8 ? t10 : 0u) | ((v8 ? t11 : 0u) << 16);
    w[5] = (v8 ? t12 : 0u) | ((v16 ? t13 : 0u) << 16);
    w[6] = (v16 ? t14 : 0u) | ((v16 ? t15 : 0u) << 16);
    w[7] = lh_largetbl[i16];
    w[8] = (unsigned) (x >= 15u) + (unsigned) (y >= 15u);
    w[9] = (m != 0u);
}

/* build the tables for the image Q.ix[0], pairs below bigv; tab = [LH_BHD_NW][STRIDE] exclusive
 * prefix sums over the bands (entry 22 = total), bmx[22] = band maxima; whole wave */
LH_DEVFN void
lh_bhd_build(const LhCtx & c, LhChanLds & Q, int bigv, int *tab, int *bmx)
{
    const LhQTabs *qt = LH_QT;
    const uint32_t *ix2 = (const uint32_t *) Q.ix[0];
    int const npair = bigv >> 1;
    for (int i = c.lane; i < (LH_BHD_NW + 1) * LH_BHD_STRIDE; i += 64)
        tab[i] = 0;             /* bmx follows tab */
    LH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 5; k++) {
        int const p = c.lane + 64 * k;
        int const pc = (k < 4 || p < 288) ? p : 287;
        uint32_t const v = ix2[pc];
        int const band = Q.sfb_of_line[2 * pc];
        unsigned w[LH_BHD_NW];
        lh_bhd_pair_words(qt, v & 0xffffu, v >> 16, w);
        if (p < npair) {
            unsigned const m = (v & 0xffffu) > (v >> 16) ? (v & 0xffffu) : (v >> 16);
#pragma unroll
            for (int j = 0; j < LH_BHD_NW; j++)
                lh_lds_add(&tab[j * LH_BHD_STRIDE + band], (int) w[j]);
            lh_lds_max(&bmx[band], (int) m);
        }
    }
    LH_WAVE_SYNC();
    if (c.lane < LH_BHD_NW) {
        int    *row = &tab[c.lane * LH_BHD_STRIDE];
        int     acc = 0;
        for (int b = 0; b < LH_SBMAX_L; b++) {
            int const v = row[b];
            row[b] = acc;
            acc += v;
        }
        row[LH_SBMAX_L] = acc;
    }
    LH_WAVE_SYNC();
}

/* choose_table for the bands [blo, bhi), whose largest value is mx, from the prefix tables; the pair whose
 * words are qw is taken out again */
LH_DEVCONST unsigned lh_bhd_none[LH_BHD_NW] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };

LH_DEVFN int
lh_bhd_region(const int *tab, unsigned mx, int blo, int bhi, const unsigned (&qw)[LH_BHD_NW], int *bits)
{
    unsigned w0, w1, d[LH_BHD_NW];
    if (mx == 0)
        return 0;
    /* (qw is all zero when no pair is taken out: an array by reference stays in registers, a pointer that may
     * be null put it in scratch memory) */
#pragma unroll
    for (int j = 0; j < LH_BHD_NW - 1; j++)
        d[j] = (unsigned) (tab[j * LH_BHD_STRIDE + bhi] - tab[j * LH_BHD_STRIDE + blo]) - qw[j];
    {
        /* (the words of the region's candidate tables picked by selects: the lanes of a wave hold all kinds of maxima, and
         * as a chain of branches every kind present ran on its own) */
        unsigned const c2 = (d[0] >> 16) | ((d[1] & 0xffffu) << 16), c3 = (d[1] >> 16) | ((d[2] & 0xffffu) << 16);
        unsigned const c45 = (d[2] >> 16) | ((d[3] & 0xffffu) << 16), c8 = (d[5] >> 16) | ((d[6] & 0xffffu) << 16);
        w0 = (mx > 15u) ? d[7] : (mx >= 8u) ? c8 : (mx >= 6u) ? d[4] : (mx >= 4u) ? c45
            : (mx == 3u) ? c3 : (mx == 2u) ? c2 : (d[0] & 0xffffu);
        w1 = (mx > 15u) ? d[8] : (mx >= 8u) ? (d[6] >> 16) : (mx >= 6u) ? (d[5] & 0xffffu) : (mx >= 4u) ? (d[3] >> 16) : 0u;
        return lh_region_decide(mx, w0, w1, bits);
    }
}

/* lane l holds v[l]: *below = max(v[0 .. l)), *from = max(v[l .. 63]) (log-step scans through lane
 * exchanges; the two chains run side by side) */
LH_DEVFN void
lh_bhd_scan_max(int lane, unsigned v, unsigned *below, unsigned *from)
{
#if !defined(LH_EMU)
    /* On the device, for the caller there is (v = 0 from lane 32 on: 22 bands): both scans on the DPP network instead of
     * thirteen round trips through the LDS crossbar.  The running maximum is lh_wave_scan_max_u32 and a shift by one
     * lane; the maximum from a lane on is a scan towards lower lanes inside the rows of sixteen (row_shl 1, 2, 4, 8), and
     * row 0 takes row 1's total (its lane 16) on top. */
    {
        unsigned const p_ = lh_wave_scan_max_u32(v);
        unsigned q_ = v, t_;
        *below = lh_dpp < 0x138, 0u > (p_);     /* wave_shr:1 -- lane 0 has nobody below it */
        t_ = lh_dpp < 0x101, 0u > (q_);
        q_ = t_ > q_ ? t_ : q_;
        t_ = lh_dpp < 0x102, 0u > (q_);
        q_ = t_ > q_ ? t_ : q_;
        t_ = lh_dpp < 0x104, 0u > (q_);
        q_ = t_ > q_ ? t_ : q_;
        t_ = lh_dpp < 0x108, 0u > (q_);
        q_ = t_ > q_ ? t_ : q_;
        {
            unsigned const row1 = (unsigned) __builtin_amdgcn_readlane((int) q_, 16);
            *from = (lane < 16 && row1 > q_) ? row1 : q_;
        }
        return;
    }
#endif
    unsigned p = v, q = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned const tp = lh_shfl_u32(p, (lane - d) & 63), tq = lh_shfl_u32(q, (lane + d) & 63);
        p = (lane >= d && tp > p) ? tp : p;
        q = (lane + d < 64 && tq > q) ? tq : q;
    }
    {
        unsigned const e = lh_shfl_u32(p, (lane - 1) & 63);
        *below = lane > 0 ? e : 0u;
    }
    *from = q;
}

/* exclusive running minimum over the lanes: min(v[0 .. l)) (0x7fffffff for lane 0) */
LH_DEVFN unsigned
lh_bhd_scan_min_excl(int lane, unsigned v)
{
#if !defined(LH_EMU)
    {
        /* the same on the DPP network: row_shr 1, 2, 4, 8, row_bcast 15 / 31, then one lane up (seven round trips
         * through the LDS crossbar otherwise) */
        unsigned p_ = v, t_;
        t_ = lh_dpp < 0x111, 0xffffffffu > (p_);
        p_ = t_ < p_ ? t_ : p_;
        t_ = lh_dpp < 0x112, 0xffffffffu > (p_);
        p_ = t_ < p_ ? t_ : p_;
        t_ = lh_dpp < 0x114, 0xffffffffu > (p_);
        p_ = t_ < p_ ? t_ : p_;
        t_ = lh_dpp < 0x118, 0xffffffffu > (p_);
        p_ = t_ < p_ ? t_ : p_;
        t_ = lh_dpp_rows < 0x142, 0xa, 0xffffffffu > (p_);
        p_ = t_ < p_ ? t_ : p_;
        t_ = lh_dpp_rows < 0x143, 0xc, 0xffffffffu > (p_);
        p_ = t_ < p_ ? t_ : p_;
        return lh_dpp < 0x138, 0x7fffffffu > (p_);
    }
#endif
    unsigned p = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned const tp = lh_shfl_u32(p, (lane - d) & 63);
        p = (lane >= d && tp < p) ? tp : p;
    }
    {
        unsigned const e = lh_shfl_u32(p, (lane - 1) & 63);
        return lane > 0 ? e : 0x7fffffffu;
    }
}

LH_DEVFN void
lh_best_huffman_divide_body(const LhCtx & c, LhChanLds & Q, const LhQR & R, LhGrR & g)
{
    const LhQTabs *qt = LH_QT;
    if (LH_IS_LSF && R.block_type == LH_SHORT_TYPE)
        return;                 /* an LSF short block is left alone (reference takehiro.c:898-900) */
    const int16_t *ix = Q.ix[0];
    int const lane = c.lane;
    int const bigv0 = g.big_values;
    int const count1bits0 = g.count1bits;
    int    *tab = (int *) Q.xrpow;      /* xrpow is dead after the outer loop: per-band length sums */
    int    *bmx = tab + LH_BHD_NW * LH_BHD_STRIDE;
    /* what lane s keeps for the region split with r0 + r1 = s (recalc_divide_init's r01_bits / r01_div /
     * r0_tbl / r1_tbl, reference takehiro.c:809-870) */
    int     s_bits = LH_LARGE_BITS, s_div = 0, s_t0 = 0, s_t1 = 0;
    unsigned band_max = 0, max_from = 0;        /* lane = band: its maximum; the maximum of the bands from it on */

    LH_PT(t_bhd);
    LH_WAVE_SYNC();
    if (R.block_type == LH_NORM_TYPE) {
        /* recalc_divide_init: lanes 0..15 cost region 0 for r0 = lane; then each of the
         * 128 (r0, r1) pairs costs region 1; results go through LDS (save_xrpow is dead) */
        int    *r0bits_a = (int *) Q.save_xrpow;      /* [16] */
        int    *r0t_a = r0bits_a + 16;                /* [16] */
        int    *comb_bits = r0bits_a + 32;            /* [128] */
        int    *comb_tbl = comb_bits + 128;           /* [128] */
        unsigned max_below;
        int     nr0;
        lh_bhd_build(c, Q, bigv0, tab, bmx);
        LH_PA(18, t_bhd);
        band_max = (lane < LH_SBMAX_L) ? (unsigned) bmx[lane] : 0u;
        lh_bhd_scan_max(lane, band_max, &max_below, &max_from);
        /* the reference's loops over r0 stop at the first band boundary >= big_values: boundaries ascend */
        nr0 = lh_popc64(lh_ballot(lane < 16 && (int) qt->sfb_l[lane < 16 ? lane + 1 : 0] < bigv0));
        {
            /* region 0 = bands [0, r0]: its maximum is the next lane's `below' (exchanged by all lanes:
             * a lane that sits out cannot be read from) */
#if !defined(LH_EMU)
            unsigned const mx0 = lh_dpp < 0x130, 0u > (max_below);      /* wave_shl:1 (lanes 0..15 are looked at) */
#else
            unsigned const mx0 = lh_shfl_u32(max_below, (lane + 1) & 63);
#endif
            if (lane < 16) {
                int const r0 = lane;
                int     b = 0, t = 0;
                if (r0 < nr0)
                    t = lh_bhd_region(tab, mx0, 0, r0 + 1, lh_bhd_none, &b);
                r0bits_a[r0] = b;
                r0t_a[r0] = t;
            }
        }
        LH_WAVE_SYNC();
        LH_PA(19, t_bhd);
        for (int cmb = lane; cmb < 128; cmb += 64) {
            int const r0 = cmb >> 3, r1 = cmb & 7;
            int const hi = r0 + r1 + 2;
            int const e2 = qt->sfb_l[hi < 23 ? hi : 23];
            int     b = LH_LARGE_BITS, t = 0;
            if (r0 < nr0 && e2 < bigv0) {
                unsigned mx = 0;
                for (int bd = r0 + 1; bd < hi; bd++) {
                    unsigned const m = (unsigned) bmx[bd];
                    mx = m > mx ? m : mx;
                }
                b = r0bits_a[r0];
                t = lh_bhd_region(tab, mx, r0 + 1, hi, lh_bhd_none, &b);
            }
            comb_bits[cmb] = b;
            comb_tbl[cmb] = t;
        }
        LH_WAVE_SYNC();
        LH_PA(20, t_bhd);
        if (lane < 23) {
            /* first minimum in (r0 ascending) order over the splits with r0 + r1 = lane: r0 runs over
             * at most eight values, and whether region 1 ends below big_values depends on the sum only */
            int const sidx = lane;
            int const ok2 = (int) qt->sfb_l[sidx + 2 < 23 ? sidx + 2 : 23] < bigv0;
            int const lo = sidx > 7 ? sidx - 7 : 0, hi = (sidx < nr0 - 1) ? sidx : nr0 - 1;
            int     v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int const r0 = (lo + u <= hi) ? lo + u : lo;
                v[u] = comb_bits[(r0 * 8 + sidx - r0) & 127];
            }
            if (ok2) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (lo + u <= hi && s_bits > v[u]) {
                        s_bits = v[u];
                        s_div = lo + u;
                    }
                }
            }
            if (s_bits < LH_LARGE_BITS) {
                s_t0 = r0t_a[s_div];
                s_t1 = comb_tbl[s_div * 8 + sidx - s_div];
            }
            else
                s_div = 0;
        }
        LH_WAVE_SYNC();
        LH_PA(21, t_bhd);
    }
    /* recalc_divide_sub against (bigv, count1bits): first with the original counts,
     * then (maybe) with one more quadruple moved into the count1 region */
    for (int pass = 0; pass < 2; pass++) {
        int     bigv, c1bits, c1, c1sel;
        if (pass == 0) {
            if (R.block_type != LH_NORM_TYPE)
                continue;
            bigv = bigv0;
            c1bits = count1bits0;
            c1 = g.count1;
            c1sel = g.count1table_select;
        }
        else {
            int     i = bigv0, a1, a2, nq;
            if (i == 0 || (unsigned) (ix[i - 2] | ix[i - 1]) > 1)
                return;
            i = g.count1 + 2;
            if (i > 576)
                return;
            c1 = i;
            /* the quadruples from c1 down to big_values, counted with both count1 tables: one lane
             * per quadruple (up to 144 of them) */
            nq = (c1 - g.big_values + 3) >> 2;
            {
                const uint32_t *ix2 = (const uint32_t *) ix;
                unsigned acc = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    int const q = lane + 64 * k;
                    int const top = c1 - 4 * q;                 /* the quadruple is ix[top - 4 .. top - 1] */
                    int const tc = (q < nq) ? top : 4;
                    uint32_t const u0 = ix2[(tc - 4) >> 1], u1 = ix2[(tc - 2) >> 1];
                    unsigned const idx = ((((u0 & 1u) * 2 + ((u0 >> 16) & 1u)) * 2 + (u1 & 1u)) * 2 + ((u1 >> 16) & 1u));
                    acc += (q < nq) ? qt->t3233[idx] : 0u;
                }
                acc = lh_wave_sum_u32(acc);
                a1 = (int) (acc >> 16);
                a2 = (int) (acc & 0xffffu);
            }
            i = c1 - 4 * nq;
            bigv = i;
            c1sel = 0;
            if (a1 > a2) {
                a1 = a2;
                c1sel = 1;
            }
            c1bits = a1;
            if (R.block_type != LH_NORM_TYPE) {
                /* start / stop / short blocks: two fixed regions */
                int     v[5][2];
                int     p23 = a1, t0 = g.table_select[0], t1 = g.table_select[1];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    int const p = lane + 64 * k;
                    v[k][0] = (p < 288) ? ix[2 * p] : 0;
                    v[k][1] = (p < 288) ? ix[2 * p + 1] : 0;
                }
                a1 = qt->sfb_l[7 + 1];
                if (a1 > i)
                    a1 = i;
                if (a1 > 0)
                    t0 = lh_choose_table_wave(c, v, 0, a1, &p23);
                if (i > a1)
                    t1 = lh_choose_table_wave(c, v, a1, i, &p23);
                if (g.part2_3_length > p23) {
                    g.part2_3_length = p23;
                    g.count1 = c1;
                    g.big_values = bigv;
                    g.count1table_select = c1sel;
                    g.count1bits = c1bits;
                    g.table_select[0] = t0;
                    g.table_select[1] = t1;
                }
                return;
            }
        }
        {
            /* lane r2 costs region 2 = [sfb_l[r2], bigv).  Second pass: the pair [bigv, bigv0) has moved
             * to the count1 region; it is taken out of the band sums (its values are 0/1, so its
             * band's maximum can only drop from 1 to 0, and only when it was the band's last
             * non-zero pair) */
            unsigned qw[LH_BHD_NW] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };     /* first pass: nothing is taken out */
            int const minus_q = (pass == 1 && bigv == bigv0 - 2);
            unsigned mfrom = max_from;
            int     r2b = 0, r2t = 0, lower, live;
            if (pass == 1) {
                /* (always the case for long blocks: big_values and count1 differ by whole quadruples) */
                uint32_t const v = ((const uint32_t *) ix)[(bigv0 >> 1) - 1];
                int const bq = Q.sfb_of_line[bigv0 - 2];
                unsigned below;
                int     mq;
                lh_bhd_pair_words(qt, v & 0xffffu, v >> 16, qw);
                mq = bmx[bq];
                if (mq == 1 && v != 0u
                    && tab[9 * LH_BHD_STRIDE + bq + 1] - tab[9 * LH_BHD_STRIDE + bq] == 1)
                    mq = 0;
                lh_bhd_scan_max(lane, (lane == bq) ? (unsigned) mq : band_max, &below, &mfrom);
                if (!minus_q)
                    return;     /* not reachable for long blocks; the reference's general case is not built */
            }
            /* the sum r0 + r1 = r2 - 2 is lane r2 - 2's */
#if !defined(LH_EMU)
            /* (two lanes up on the DPP network; lanes 0 and 1 are not looked at) */
            lower = (int) lh_dpp < 0x138, 0u > (lh_dpp < 0x138, 0u > ((uint32_t) s_bits));
#else
            lower = (int) lh_shfl_u32((uint32_t) s_bits, (lane - 2) & 63);
#endif
            live = lane >= 2 && lane < LH_SBMAX_L + 1 && (int) qt->sfb_l[(lane >= 2 && lane < 23) ? lane : 0] < bigv;
            if (live) {
                r2b = lower + c1bits;
                if (lower < LH_LARGE_BITS)
                    r2t = lh_bhd_region(tab, mfrom, lane, LH_SBMAX_L, qw, &r2b);
            }
            lower += c1bits;
            {
                /* The reference walks r2 = 2, 3, ... and stops at the first boundary >= bigv, or when
                 * the best length so far is <= r01_bits + count1bits; in between it takes every strictly
                 * better total.  So: the running minimum before each lane, the first lane that stops
                 * the walk, the first lane below it that reaches the minimum. */
                unsigned const key = live ? (unsigned) r2b : 0x7fffffffu;
                unsigned const before0 = lh_bhd_scan_min_excl(lane, (lane >= 2) ? key : 0x7fffffffu);
                unsigned const p0 = (unsigned) g.part2_3_length;
                unsigned const before = before0 < p0 ? before0 : p0;
                uint64_t const stop = lh_ballot(lane >= 2 && (!live || before <= (unsigned) lower));
                int const brk = lh_ffs64(stop);         /* lane 23 is never live: there is always a stop */
                unsigned const mine = (lane >= 2 && lane < brk) ? key : 0x7fffffffu;
                unsigned const best = lh_wave_min_u32(mine);
                if (best < p0) {
                    int const sel = lh_ffs64(lh_ballot(mine == best));
                    g.part2_3_length = (int) best;
                    g.big_values = bigv;
                    g.count1 = c1;
                    g.count1bits = c1bits;
                    g.count1table_select = c1sel;
                    g.region0_count = (int) lh_bcast_u32((uint32_t) s_div, sel - 2);
                    g.region1_count = sel - 2 - g.region0_count;
                    g.table_select[0] = (int) lh_bcast_u32((uint32_t) s_t0, sel - 2);
                    g.table_select[1] = (int) lh_bcast_u32((uint32_t) s_t1, sel - 2);
                    g.table_select[2] = (int) lh_bcast_u32((uint32_t) r2t, sel);
                }
            }
            if (pass == 0)
                LH_PA(22, t_bhd);
            else
                LH_PA(23, t_bhd);
        }
    }
}

LH_STAGEFN void
lh_best_huffman_divide(int qch)
{
    LhCtx const c = lh_ctx_load();
    LhQR const R = lh_uniform(lh_lds.rg[qch].R);
    LhGrR   g = lh_uniform(lh_lds.rg[qch].g);
    lh_best_huffman_divide_body(c, lh_lds.u.quant.ch[qch], R, g);
    lh_rg_put(c, R, g);
}

LH_STAGEFN void
lh_best_huffman_divide_n(int qch)
{
    LhCtx   c = lh_ctx_load();
    LhQR    R = lh_uniform(lh_lds.rg[qch].R);
    LhGrR   g = lh_uniform(lh_lds.rg[qch].g);
    lh_pin_usual(c, 0);
    lh_pin_usual(R);
    lh_best_huffman_divide_body(c, lh_lds.u.quant.ch[qch], R, g);
    lh_rg_put(c, R, g);
}

/* ---------------------------------------------------------------------- */
/* reservoir + bit allocation, wave-uniform integer arithmetic               */
/* (reference reservoir.c:82-293, quantize_pvt.c:428-545, bitstream.c:60-88) */
LH_DEVFN int
lh_frame_bits(const LhConfig * cfg, int bitrate_index, int padding)
{
    int const bit_rate = cfg->version ? lh_bitrate_mpeg1[bitrate_index] : lh_bitrate_mpeg2[bitrate_index];
    return 8 * ((cfg->version + 1) * 72000 * bit_rate / cfg->samplerate + padding);
}

/* What the reservoir lets a granule spend (reference reservoir.c:171-218, ResvMaxBits): a target -- the
 * mean, plus whatever the reservoir holds beyond 90 % of its size, or minus a tenth while it is being
 * filled -- and a reserve that may be handed out on top (up to 60 % of the reservoir's size).  `flags'
 * is substep_shaping: bit 0 shrinks the usable reservoir, bit 7 records "draining". */
struct LhGranuleBudget {
    int     target, reserve;
};

LH_DEVFN LhGranuleBudget
lh_granule_budget(const LhConfig * cfg, int held, int size, int *flags, int mean, int count_mean)
{
    LhGranuleBudget b;
    int const usable = (*flags & 1) ? (int) (size * 0.9) : size;
    int const level = count_mean ? held + mean : held;
    int const cushion = (size * 6) / 10;
    int     overflow = 0;
    b.target = mean;
    if (level * 10 > usable * 9) {
        overflow = level - (usable * 9) / 10;
        b.target += overflow;
        *flags |= 0x80;
    }
    else {
        *flags &= 0x7f;
        if (!cfg->disable_reservoir && !(*flags & 1))
            b.target = (int) (b.target - .1 * mean);
    }
    b.reserve = (level < cushion ? level : cushion) - overflow;
    if (b.reserve < 0)
        b.reserve = 0;
    return b;
}

/* Split a granule's budget over its channels by perceptual entropy (reference quantize_pvt.c:428-486,
 * on_pe): every channel starts from an equal share, asks for pe / 700 times that (at most three quarters
 * of the mean more), the requests are scaled down to the reserve, and the total is capped.  Returns the
 * granule's ceiling (target + reserve). */
LH_DEVFN int
lh_on_pe(const LhConfig * cfg, int ResvSize, int ResvMax, int *substep, const float pe[2],
         int targ_bits[2], int mean_bits, int cbr)
{
    int const nch = cfg->channels;
    LhGranuleBudget const budget = lh_granule_budget(cfg, ResvSize, ResvMax, substep, mean_bits, cbr);
    int const ceiling = (budget.target + budget.reserve > LH_MAX_BITS_PER_GRANULE) ? LH_MAX_BITS_PER_GRANULE
        : budget.target + budget.reserve;
    int const share = (LH_MAX_BITS_PER_CHANNEL < budget.target / nch) ? LH_MAX_BITS_PER_CHANNEL : budget.target / nch;
    int const most = mean_bits * 3 / 4;
    int     want[2] = { 0, 0 }, wanted = 0, left = budget.reserve, total = 0;
    targ_bits[1] = 0;           /* mono: the second channel has no budget */
    for (int ch = 0; ch < nch; ++ch) {
        int     w = (int) (share * pe[ch] / 700.0 - share);
        w = w > most ? most : w;
        w = w < 0 ? 0 : w;
        if (w + share > LH_MAX_BITS_PER_CHANNEL)
            w = (LH_MAX_BITS_PER_CHANNEL - share > 0) ? LH_MAX_BITS_PER_CHANNEL - share : 0;
        want[ch] = w;
        wanted += w;
    }
    for (int ch = 0; ch < nch; ++ch) {
        int const granted = (wanted > budget.reserve && wanted > 0) ? budget.reserve * want[ch] / wanted : want[ch];
        targ_bits[ch] = share + granted;
        left -= granted;
        total += targ_bits[ch];
    }
    (void) left;
    if (total > LH_MAX_BITS_PER_GRANULE)
        for (int ch = 0; ch < nch; ++ch)
            targ_bits[ch] = targ_bits[ch] * LH_MAX_BITS_PER_GRANULE / total;
    return ceiling;
}

/* Mid/side frames: bits move from the side channel to the mid channel according to how little of the
 * energy the side holds (reference quantize_pvt.c:489-541, reduce_side): up to a third of the side's
 * share at ms_ener_ratio 0, nothing from 0.5 on; the side keeps at least 125 bits, and the pair stays
 * within the granule's ceiling. */
LH_DEVFN void
lh_reduce_side(int targ_bits[2], float ms_ener_ratio, int mean_bits, int max_bits)
{
    float   tilt = (float) (.33 * (.5 - ms_ener_ratio) / .5);
    int     shift, both;
    tilt = tilt < 0 ? 0 : tilt;
    tilt = tilt > .5 ? .5 : tilt;
    shift = (int) (tilt * .5 * (targ_bits[0] + targ_bits[1]));
    if (shift > LH_MAX_BITS_PER_CHANNEL - targ_bits[0])
        shift = LH_MAX_BITS_PER_CHANNEL - targ_bits[0];
    shift = shift < 0 ? 0 : shift;
    if (targ_bits[1] >= 125) {
        int const side_after = targ_bits[1] - shift;
        if (side_after > 125) {
            targ_bits[0] += (targ_bits[0] < mean_bits) ? shift : 0;
            targ_bits[1] = side_after;
        }
        else {
            targ_bits[0] += targ_bits[1] - 125;
            targ_bits[1] = 125;
        }
    }
    both = targ_bits[0] + targ_bits[1];
    if (both > max_bits) {
        targ_bits[0] = (max_bits * targ_bits[0]) / both;
        targ_bits[1] = (max_bits * targ_bits[1]) / both;
    }
}

#endif
