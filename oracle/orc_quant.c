/*
 * orc_quant.c -- CPU restatement of the CBR quantisation / noise-shaping loop
 * (reference libmp3lame/quantize.c:48-1232,1988-2050, quantize_pvt.c:428-913,
 * takehiro.c:113-1327, reservoir.c:82-293).  TEST INFRASTRUCTURE ONLY.
 */
#include "orc_common.h"
#include "../deprecated-lame-mirror_amd/csrc/lh_static_tables.h"

#define MAGIC_FLOAT (65536*(128))
#define MAGIC_INT 0x4b000000
#define IPOW20(T,x)  ((T)->ipow20[x])
#define POW20(T,x)   ((T)->pow20[(x)+LH_QMAX2])

static const int slen1_n[16] = { 1, 1, 1, 1, 8, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16 };
static const int slen2_n[16] = { 1, 2, 4, 8, 1, 2, 4, 8, 2, 4, 8, 2, 4, 8, 4, 8 };
static const int slen1_tab[16] = { 0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4 };
static const int slen2_tab[16] = { 0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3 };
static const int scfsi_band[5] = { 0, 6, 11, 16, 21 };
static const int scale_short[16] = { 0, 18, 36, 54, 54, 36, 54, 72, 54, 72, 90, 72, 90, 108, 108, 126 };
static const int scale_mixed[16] = { 0, 18, 36, 54, 51, 35, 53, 71, 52, 70, 88, 69, 87, 105, 104, 122 };
static const int scale_long[16] = { 0, 10, 20, 30, 33, 21, 31, 41, 32, 42, 52, 43, 53, 63, 64, 74 };

#define HLEN(t)  (lh_ht_len + lh_ht_offset[t])

/* ---------------------------------------------------------------------- */
/* quantiser (reference takehiro.c:113-414)                                 */
typedef union {
    float   f;
    int     i;
} fi_union;

static void
quantize_lines_xrpow_01(unsigned int l, float istep, const float *xr, int *ix)
{
    const float compareval0 = (1.0f - 0.4054f) / istep;
    unsigned int i;
    for (i = 0; i < l; i += 2) {
        float const xr_0 = xr[i + 0];
        float const xr_1 = xr[i + 1];
        ix[i + 0] = (compareval0 > xr_0) ? 0 : 1;
        ix[i + 1] = (compareval0 > xr_1) ? 0 : 1;
    }
}

static void
quantize_lines_xrpow(const LhTables * T, unsigned int l, float istep, const float *xp, int *pi)
{
    fi_union *fi = (fi_union *) pi;
    unsigned int n = (l >> 1) * 2, i;   /* pairs only, as the reference drops an odd tail */
    for (i = 0; i < n; i++) {
        double  x0 = istep * xp[i];
        x0 += MAGIC_FLOAT;
        fi[i].f = x0;
        fi[i].f = x0 + T->adj43asm[fi[i].i - MAGIC_INT];
        fi[i].i -= MAGIC_INT;
    }
}

static void
quantize_xrpow(const LhTables * T, const float *xp, int *pi, float istep, OrcGr const *cod_info,
               OrcNoiseData const *prev_noise)
{
    int     sfb, sfbmax, j = 0, prev_data_use;
    int    *iData = pi;
    int     accumulate = 0, accumulate01 = 0;
    int    *acc_iData = iData;
    const float *acc_xp = xp;

    prev_data_use = (prev_noise && (cod_info->global_gain == prev_noise->global_gain));
    sfbmax = (cod_info->block_type == LH_SHORT_TYPE) ? 38 : 21;
    for (sfb = 0; sfb <= sfbmax; sfb++) {
        int     step = -1;
        if (prev_data_use || cod_info->block_type == LH_NORM_TYPE) {
            step = cod_info->global_gain
                - ((cod_info->scalefac[sfb] + (cod_info->preflag ? lh_pretab[sfb] : 0))
                   << (cod_info->scalefac_scale + 1))
                - cod_info->subblock_gain[cod_info->window[sfb]] * 8;
        }
        if (prev_data_use && (prev_noise->step[sfb] == step)) {
            if (accumulate) {
                quantize_lines_xrpow(T, accumulate, istep, acc_xp, acc_iData);
                accumulate = 0;
            }
            if (accumulate01) {
                quantize_lines_xrpow_01(accumulate01, istep, acc_xp, acc_iData);
                accumulate01 = 0;
            }
        }
        else {
            int     l = cod_info->width[sfb];
            if ((j + cod_info->width[sfb]) > cod_info->max_nonzero_coeff) {
                int     usefullsize = cod_info->max_nonzero_coeff - j + 1;
                memset(&pi[cod_info->max_nonzero_coeff], 0,
                       sizeof(int) * (576 - cod_info->max_nonzero_coeff));
                l = usefullsize;
                if (l < 0)
                    l = 0;
                sfb = sfbmax + 1;
            }
            if (!accumulate && !accumulate01) {
                acc_iData = iData;
                acc_xp = xp;
            }
            if (prev_noise && prev_noise->sfb_count1 > 0 && sfb >= prev_noise->sfb_count1 &&
                prev_noise->step[sfb] > 0 && step >= prev_noise->step[sfb]) {
                if (accumulate) {
                    quantize_lines_xrpow(T, accumulate, istep, acc_xp, acc_iData);
                    accumulate = 0;
                    acc_iData = iData;
                    acc_xp = xp;
                }
                accumulate01 += l;
            }
            else {
                if (accumulate01) {
                    quantize_lines_xrpow_01(accumulate01, istep, acc_xp, acc_iData);
                    accumulate01 = 0;
                    acc_iData = iData;
                    acc_xp = xp;
                }
                accumulate += l;
            }
            if (l <= 0) {
                if (accumulate01) {
                    quantize_lines_xrpow_01(accumulate01, istep, acc_xp, acc_iData);
                    accumulate01 = 0;
                }
                if (accumulate) {
                    quantize_lines_xrpow(T, accumulate, istep, acc_xp, acc_iData);
                    accumulate = 0;
                }
                break;
            }
        }
        if (sfb <= sfbmax) {
            iData += cod_info->width[sfb];
            xp += cod_info->width[sfb];
            j += cod_info->width[sfb];
        }
    }
    if (accumulate)
        quantize_lines_xrpow(T, accumulate, istep, acc_xp, acc_iData);
    if (accumulate01)
        quantize_lines_xrpow_01(accumulate01, istep, acc_xp, acc_iData);
}

/* ---------------------------------------------------------------------- */
/* Huffman bit counting (reference takehiro.c:423-647)                      */
static int
ix_max(const int *ix, const int *end)
{
    int     max1 = 0, max2 = 0;
    do {
        int const x1 = *ix++;
        int const x2 = *ix++;
        if (max1 < x1)
            max1 = x1;
        if (max2 < x2)
            max2 = x2;
    } while (ix < end);
    if (max1 < max2)
        max1 = max2;
    return max1;
}

static int
count_bit_ESC(const int *ix, const int *const end, int t1, const int t2, unsigned int *const s)
{
    unsigned int const linbits = lh_ht_xlen[t1] * 65536u + lh_ht_xlen[t2];
    unsigned int sum = 0, sum2;
    do {
        unsigned int x = *ix++;
        unsigned int y = *ix++;
        if (x >= 15u) {
            x = 15u;
            sum += linbits;
        }
        if (y >= 15u) {
            y = 15u;
            sum += linbits;
        }
        x <<= 4u;
        x += y;
        sum += lh_largetbl[x];
    } while (ix < end);
    sum2 = sum & 0xffffu;
    sum >>= 16u;
    if (sum > sum2) {
        sum = sum2;
        t1 = t2;
    }
    *s += sum;
    return t1;
}

static const int huf_tbl_noESC[] = { 1, 2, 5, 7, 7, 10, 10, 13, 13, 13, 13, 13, 13, 13, 13 };

static int
choose_table(const int *ix, const int *const end, int *const _s)
{
    unsigned int *s = (unsigned int *) _s;
    unsigned int max;
    int     choice, choice2;
    max = ix_max(ix, end);
    if (max <= 15) {
        if (max == 0)
            return 0;
        if (max == 1) {
            unsigned int sum1 = 0;
            const uint8_t *const hlen1 = HLEN(1);
            do {
                unsigned int const x0 = *ix++;
                unsigned int const x1 = *ix++;
                sum1 += hlen1[x0 + x0 + x1];
            } while (ix < end);
            *s += sum1;
            return 1;
        }
        if (max <= 3) {
            int     t1 = huf_tbl_noESC[max - 1];
            const unsigned int xlen = lh_ht_xlen[t1];
            uint32_t const *table = (t1 == 2) ? &lh_table23[0] : &lh_table56[0];
            unsigned int sum = 0, sum2;
            do {
                unsigned int const x0 = *ix++;
                unsigned int const x1 = *ix++;
                sum += table[x0 * xlen + x1];
            } while (ix < end);
            sum2 = sum & 0xffffu;
            sum >>= 16u;
            if (sum > sum2) {
                sum = sum2;
                t1++;
            }
            *s += sum;
            return t1;
        }
        {
            int     t1 = huf_tbl_noESC[max - 1];
            unsigned int sum1 = 0, sum2 = 0, sum3 = 0;
            const unsigned int xlen = lh_ht_xlen[t1];
            const uint8_t *const hlen1 = HLEN(t1);
            const uint8_t *const hlen2 = HLEN(t1 + 1);
            const uint8_t *const hlen3 = HLEN(t1 + 2);
            int     t;
            do {
                unsigned int x0 = *ix++;
                unsigned int x1 = *ix++;
                unsigned int x = x0 * xlen + x1;
                sum1 += hlen1[x];
                sum2 += hlen2[x];
                sum3 += hlen3[x];
            } while (ix < end);
            t = t1;
            if (sum1 > sum2) {
                sum1 = sum2;
                t++;
            }
            if (sum1 > sum3) {
                sum1 = sum3;
                t = t1 + 2;
            }
            *s += sum1;
            return t;
        }
    }
    if (max > LH_IXMAX) {
        *s = LH_LARGE_BITS;
        return -1;
    }
    max -= 15u;
    for (choice2 = 24; choice2 < 32; choice2++)
        if (lh_ht_linmax[choice2] >= max)
            break;
    for (choice = choice2 - 8; choice < 24; choice++)
        if (lh_ht_linmax[choice] >= max)
            break;
    return count_bit_ESC(ix, end, choice, choice2, s);
}

static void best_huffman_divide(OrcStream * S, OrcGr * gi);

/* reference takehiro.c:654-765 */
static int
noquant_count_bits(OrcStream * S, OrcGr * const gi, OrcNoiseData * prev_noise)
{
    const LhTables *T = S->tab;
    int     bits = 0;
    int     i, a1, a2;
    int const *const ix = gi->l3_enc;

    i = ((gi->max_nonzero_coeff + 2) >> 1) << 1;
    if (i > 576)
        i = 576;
    if (prev_noise)
        prev_noise->sfb_count1 = 0;
    for (; i > 1; i -= 2)
        if (ix[i - 1] | ix[i - 2])
            break;
    gi->count1 = i;
    a1 = a2 = 0;
    for (; i > 3; i -= 4) {
        int     x4 = ix[i - 4];
        int     x3 = ix[i - 3];
        int     x2 = ix[i - 2];
        int     x1 = ix[i - 1];
        int     p;
        if ((unsigned int) (x4 | x3 | x2 | x1) > 1)
            break;
        p = ((x4 * 2 + x3) * 2 + x2) * 2 + x1;
        a1 += lh_t32l[p];
        a2 += lh_t33l[p];
    }
    bits = a1;
    gi->count1table_select = 0;
    if (a1 > a2) {
        bits = a2;
        gi->count1table_select = 1;
    }
    gi->count1bits = bits;
    gi->big_values = i;
    if (i == 0)
        return bits;

    if (gi->block_type == LH_SHORT_TYPE) {
        a1 = 3 * T->sfb_s[3];
        if (a1 > gi->big_values)
            a1 = gi->big_values;
        a2 = gi->big_values;
    }
    else if (gi->block_type == LH_NORM_TYPE) {
        a1 = gi->region0_count = T->bv_scf[i - 2];
        a2 = gi->region1_count = T->bv_scf[i - 1];
        a2 = T->sfb_l[a1 + a2 + 2];
        a1 = T->sfb_l[a1 + 1];
        if (a2 < i)
            gi->table_select[2] = choose_table(ix + a2, ix + i, &bits);
    }
    else {
        gi->region0_count = 7;
        gi->region1_count = LH_SBMAX_L - 1 - 7 - 1;
        a1 = T->sfb_l[7 + 1];
        a2 = i;
        if (a1 > a2)
            a1 = a2;
    }
    a1 = (a1 < i) ? a1 : i;
    a2 = (a2 < i) ? a2 : i;
    if (0 < a1)
        gi->table_select[0] = choose_table(ix, ix + a1, &bits);
    if (a1 < a2)
        gi->table_select[1] = choose_table(ix + a1, ix + a2, &bits);
    if (S->cfg->use_best_huffman == 2) {
        gi->part2_3_length = bits;
        best_huffman_divide(S, gi);
        bits = gi->part2_3_length;
    }
    if (prev_noise) {
        if (gi->block_type == LH_NORM_TYPE) {
            int     sfb = 0;
            while (T->sfb_l[sfb] < gi->big_values)
                sfb++;
            prev_noise->sfb_count1 = sfb;
        }
    }
    return bits;
}

/* reference takehiro.c:767-801 */
static int
count_bits(OrcStream * S, const float *const xr, OrcGr * const gi, OrcNoiseData * prev_noise)
{
    const LhTables *T = S->tab;
    int    *const ix = gi->l3_enc;
    float const w = (LH_IXMAX) / IPOW20(T, gi->global_gain);
    if (gi->xrpow_max > w) {
        return LH_LARGE_BITS;
    }
    quantize_xrpow(T, xr, ix, IPOW20(T, gi->global_gain), gi, prev_noise);
    if (S->substep_shaping & 2) {
        int     sfb, j = 0;
        int const gain = gi->global_gain + gi->scalefac_scale;
        const float roundfac = 0.634521682242439 / IPOW20(T, gain);
        for (sfb = 0; sfb < gi->sfbmax; sfb++) {
            int const width = gi->width[sfb];
            if (!S->pseudohalf[sfb])
                j += width;
            else {
                int     k;
                for (k = j, j += width; k < j; ++k)
                    ix[k] = (xr[k] >= roundfac) ? ix[k] : 0;
            }
        }
    }
    {
        int r_ = noquant_count_bits(S, gi, prev_noise);
        return r_;
    }
}

/* reference takehiro.c:809-957 */
static void
recalc_divide_init(OrcStream * S, OrcGr const *cod_info, int const *const ix, int r01_bits[],
                   int r01_div[], int r0_tbl[], int r1_tbl[])
{
    const LhTables *T = S->tab;
    int     r0, r1, bigv, r0t, r1t, bits;
    bigv = cod_info->big_values;
    for (r0 = 0; r0 <= 7 + 15; r0++)
        r01_bits[r0] = LH_LARGE_BITS;
    for (r0 = 0; r0 < 16; r0++) {
        int const a1 = T->sfb_l[r0 + 1];
        int     r0bits;
        if (a1 >= bigv)
            break;
        r0bits = 0;
        r0t = choose_table(ix, ix + a1, &r0bits);
        for (r1 = 0; r1 < 8; r1++) {
            int const a2 = T->sfb_l[r0 + r1 + 2];
            if (a2 >= bigv)
                break;
            bits = r0bits;
            r1t = choose_table(ix + a1, ix + a2, &bits);
            if (r01_bits[r0 + r1] > bits) {
                r01_bits[r0 + r1] = bits;
                r01_div[r0 + r1] = r0;
                r0_tbl[r0 + r1] = r0t;
                r1_tbl[r0 + r1] = r1t;
            }
        }
    }
}

static void
recalc_divide_sub(OrcStream * S, const OrcGr * cod_info2, OrcGr * const gi, const int *const ix,
                  const int r01_bits[], const int r01_div[], const int r0_tbl[],
                  const int r1_tbl[])
{
    const LhTables *T = S->tab;
    int     bits, r2, a2, bigv, r2t;
    bigv = cod_info2->big_values;
    for (r2 = 2; r2 < LH_SBMAX_L + 1; r2++) {
        a2 = T->sfb_l[r2];
        if (a2 >= bigv)
            break;
        bits = r01_bits[r2 - 2] + cod_info2->count1bits;
        if (gi->part2_3_length <= bits)
            break;
        r2t = choose_table(ix + a2, ix + bigv, &bits);
        if (gi->part2_3_length <= bits)
            continue;
        memcpy(gi, cod_info2, sizeof(OrcGr));
        gi->part2_3_length = bits;
        gi->region0_count = r01_div[r2 - 2];
        gi->region1_count = r2 - 2 - r01_div[r2 - 2];
        gi->table_select[0] = r0_tbl[r2 - 2];
        gi->table_select[1] = r1_tbl[r2 - 2];
        gi->table_select[2] = r2t;
    }
}

static void
best_huffman_divide(OrcStream * S, OrcGr * const gi)
{
    const LhTables *T = S->tab;
    int     i, a1, a2;
    OrcGr   cod_info2;
    int const *const ix = gi->l3_enc;
    int     r01_bits[7 + 15 + 1];
    int     r01_div[7 + 15 + 1];
    int     r0_tbl[7 + 15 + 1];
    int     r1_tbl[7 + 15 + 1];

    /* (an LSF short block is left alone: reference takehiro.c:898-900) */
    if (gi->block_type == LH_SHORT_TYPE && S->cfg->mode_gr == 1)
        return;
    memcpy(&cod_info2, gi, sizeof(OrcGr));
    if (gi->block_type == LH_NORM_TYPE) {
        recalc_divide_init(S, gi, ix, r01_bits, r01_div, r0_tbl, r1_tbl);
        recalc_divide_sub(S, &cod_info2, gi, ix, r01_bits, r01_div, r0_tbl, r1_tbl);
    }
    i = cod_info2.big_values;
    if (i == 0 || (unsigned int) (ix[i - 2] | ix[i - 1]) > 1)
        return;
    i = gi->count1 + 2;
    if (i > 576)
        return;
    memcpy(&cod_info2, gi, sizeof(OrcGr));
    cod_info2.count1 = i;
    a1 = a2 = 0;
    for (; i > cod_info2.big_values; i -= 4) {
        int const p = ((ix[i - 4] * 2 + ix[i - 3]) * 2 + ix[i - 2]) * 2 + ix[i - 1];
        a1 += lh_t32l[p];
        a2 += lh_t33l[p];
    }
    cod_info2.big_values = i;
    cod_info2.count1table_select = 0;
    if (a1 > a2) {
        a1 = a2;
        cod_info2.count1table_select = 1;
    }
    cod_info2.count1bits = a1;
    if (cod_info2.block_type == LH_NORM_TYPE)
        recalc_divide_sub(S, &cod_info2, gi, ix, r01_bits, r01_div, r0_tbl, r1_tbl);
    else {
        cod_info2.part2_3_length = a1;
        a1 = T->sfb_l[7 + 1];
        if (a1 > i)
            a1 = i;
        if (a1 > 0)
            cod_info2.table_select[0] = choose_table(ix, ix + a1, (int *) &cod_info2.part2_3_length);
        if (i > a1)
            cod_info2.table_select[1] =
                choose_table(ix + a1, ix + i, (int *) &cod_info2.part2_3_length);
        if (gi->part2_3_length > cod_info2.part2_3_length)
            memcpy(gi, &cod_info2, sizeof(OrcGr));
    }
}

/* reference takehiro.c:1135-1188 (MPEG-1) */
static int
mpeg1_scale_bitcount(OrcGr * const cod_info)
{
    int     k, sfb, max_slen1 = 0, max_slen2 = 0;
    const int *tabp;
    int    *const scalefac = cod_info->scalefac;

    if (cod_info->block_type == LH_SHORT_TYPE) {
        tabp = scale_short;
        if (cod_info->mixed_block_flag)
            tabp = scale_mixed;
    }
    else {
        tabp = scale_long;
        if (!cod_info->preflag) {
            for (sfb = 11; sfb < LH_SBPSY_L; sfb++)
                if (scalefac[sfb] < lh_pretab[sfb])
                    break;
            if (sfb == LH_SBPSY_L) {
                cod_info->preflag = 1;
                for (sfb = 11; sfb < LH_SBPSY_L; sfb++)
                    scalefac[sfb] -= lh_pretab[sfb];
            }
        }
    }
    for (sfb = 0; sfb < cod_info->sfbdivide; sfb++)
        if (max_slen1 < scalefac[sfb])
            max_slen1 = scalefac[sfb];
    for (; sfb < cod_info->sfbmax; sfb++)
        if (max_slen2 < scalefac[sfb])
            max_slen2 = scalefac[sfb];
    cod_info->part2_length = LH_LARGE_BITS;
    for (k = 0; k < 16; k++) {
        if (max_slen1 < slen1_n[k] && max_slen2 < slen2_n[k] && cod_info->part2_length > tabp[k]) {
            cod_info->part2_length = tabp[k];
            cod_info->scalefac_compress = k;
        }
    }
    return cod_info->part2_length == LH_LARGE_BITS;
}

/* MPEG-2 / 2.5 (reference takehiro.c:1195-1317): the scalefactors travel in four partitions of fixed band counts
 * (ISO 13818-3 2.4.3.2; lh_nr_of_sfb_block[table][long / short / mixed][partition]), each with its own field width;
 * scalefac_compress encodes the four widths.  Table 0 without preflag, table 2 with it (table 1 is never chosen, as
 * in the reference).  Over the range: part2_length stays what it was. */
static int
mpeg2_scale_bitcount(OrcGr * const cod_info)
{
    static const int max_range_sfac_tab[6][4] = {
        {15, 15, 7, 7}, {15, 15, 7, 0}, {7, 3, 0, 0}, {15, 31, 31, 0}, {7, 7, 7, 0}, {3, 3, 0, 0}
    };
    static const int log2tab[] = { 0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4 };
    int const table_number = cod_info->preflag ? 2 : 0;
    int const row_in_table = (cod_info->block_type == LH_SHORT_TYPE) ? 1 : 0;
    const uint8_t *partition_table = &lh_nr_of_sfb_block[(table_number * 3 + row_in_table) * 4];
    int const *const scalefac = cod_info->scalefac;
    int     partition, nr_sfb, window, over, i, sfb, max_sfac[4] = { 0, 0, 0, 0 };

    if (row_in_table == 1) {
        for (sfb = 0, partition = 0; partition < 4; partition++) {
            nr_sfb = partition_table[partition] / 3;
            for (i = 0; i < nr_sfb; i++, sfb++)
                for (window = 0; window < 3; window++)
                    if (scalefac[sfb * 3 + window] > max_sfac[partition])
                        max_sfac[partition] = scalefac[sfb * 3 + window];
        }
    }
    else {
        for (sfb = 0, partition = 0; partition < 4; partition++) {
            nr_sfb = partition_table[partition];
            for (i = 0; i < nr_sfb; i++, sfb++)
                if (scalefac[sfb] > max_sfac[partition])
                    max_sfac[partition] = scalefac[sfb];
        }
    }
    for (over = 0, partition = 0; partition < 4; partition++)
        if (max_sfac[partition] > max_range_sfac_tab[table_number][partition])
            over++;
    if (!over) {
        int     slen[4];
        for (partition = 0; partition < 4; partition++)
            slen[partition] = log2tab[max_sfac[partition]];
        if (table_number == 0)
            cod_info->scalefac_compress = (((slen[0] * 5) + slen[1]) << 4) + (slen[2] << 2) + slen[3];
        else
            cod_info->scalefac_compress = 500 + (slen[0] * 3) + slen[1];
        cod_info->part2_length = 0;
        for (partition = 0; partition < 4; partition++)
            cod_info->part2_length += slen[partition] * partition_table[partition];
    }
    return over;
}

/* reference takehiro.c:1319-1329 */
static int
scale_bitcount(const OrcStream * S, OrcGr * const cod_info)
{
    return (S->cfg->mode_gr == 2) ? mpeg1_scale_bitcount(cod_info) : mpeg2_scale_bitcount(cod_info);
}

/* reference takehiro.c:964-1014 */
static void
scfsi_calc(OrcStream * S, int ch)
{
    unsigned int i;
    int     s1, s2, c1, c2, sfb;
    OrcGr  *const gi = &S->tt[1][ch];
    OrcGr const *const g0 = &S->tt[0][ch];

    for (i = 0; i < 4; i++) {
        for (sfb = scfsi_band[i]; sfb < scfsi_band[i + 1]; sfb++)
            if (g0->scalefac[sfb] != gi->scalefac[sfb] && gi->scalefac[sfb] >= 0)
                break;
        if (sfb == scfsi_band[i + 1]) {
            for (sfb = scfsi_band[i]; sfb < scfsi_band[i + 1]; sfb++)
                gi->scalefac[sfb] = -1;
            S->scfsi[ch][i] = 1;
        }
    }
    s1 = c1 = 0;
    for (sfb = 0; sfb < 11; sfb++) {
        if (gi->scalefac[sfb] == -1)
            continue;
        c1++;
        if (s1 < gi->scalefac[sfb])
            s1 = gi->scalefac[sfb];
    }
    s2 = c2 = 0;
    for (; sfb < LH_SBPSY_L; sfb++) {
        if (gi->scalefac[sfb] == -1)
            continue;
        c2++;
        if (s2 < gi->scalefac[sfb])
            s2 = gi->scalefac[sfb];
    }
    for (i = 0; i < 16; i++) {
        if (s1 < slen1_n[i] && s2 < slen2_n[i]) {
            int const c = slen1_tab[i] * c1 + slen2_tab[i] * c2;
            if (gi->part2_length > c) {
                gi->part2_length = c;
                gi->scalefac_compress = (int) i;
            }
        }
    }
}

/* reference takehiro.c:1021-1094 */
static void
best_scalefac_store(OrcStream * S, const int gr, const int ch)
{
    OrcGr  *const gi = &S->tt[gr][ch];
    int     sfb, i, j, l;
    int     recalc = 0;

    j = 0;
    for (sfb = 0; sfb < gi->sfbmax; sfb++) {
        int const width = gi->width[sfb];
        for (l = j, j += width; l < j; ++l)
            if (gi->l3_enc[l] != 0)
                break;
        if (l == j)
            gi->scalefac[sfb] = recalc = -2;
    }
    if (!gi->scalefac_scale && !gi->preflag) {
        int     s = 0;
        for (sfb = 0; sfb < gi->sfbmax; sfb++)
            if (gi->scalefac[sfb] > 0)
                s |= gi->scalefac[sfb];
        if (!(s & 1) && s != 0) {
            for (sfb = 0; sfb < gi->sfbmax; sfb++)
                if (gi->scalefac[sfb] > 0)
                    gi->scalefac[sfb] >>= 1;
            gi->scalefac_scale = recalc = 1;
        }
    }
    if (!gi->preflag && gi->block_type != LH_SHORT_TYPE && S->cfg->mode_gr == 2) {
        for (sfb = 11; sfb < LH_SBPSY_L; sfb++)
            if (gi->scalefac[sfb] < lh_pretab[sfb] && gi->scalefac[sfb] != -2)
                break;
        if (sfb == LH_SBPSY_L) {
            for (sfb = 11; sfb < LH_SBPSY_L; sfb++)
                if (gi->scalefac[sfb] > 0)
                    gi->scalefac[sfb] -= lh_pretab[sfb];
            gi->preflag = recalc = 1;
        }
    }
    for (i = 0; i < 4; i++)
        S->scfsi[ch][i] = 0;
    if (S->cfg->mode_gr == 2 && gr == 1 && S->tt[0][ch].block_type != LH_SHORT_TYPE
        && S->tt[1][ch].block_type != LH_SHORT_TYPE) {
        scfsi_calc(S, ch);
        recalc = 0;
    }
    for (sfb = 0; sfb < gi->sfbmax; sfb++)
        if (gi->scalefac[sfb] == -2)
            gi->scalefac[sfb] = 0;
    if (recalc)
        (void) scale_bitcount(S, gi);
}

/* ---------------------------------------------------------------------- */
/* reference quantize_pvt.c:554-573 */
float
orc_ath_adjust(const LhTables * T, float a, float x, float athFloor, float ATHfixpoint)
{
    float const o = 90.30873362f;
    float const p = (ATHfixpoint < 1.f) ? 94.82444863f : ATHfixpoint;
    float   u = orc_fast_log2(T, x) * (ORC_LOG2 / ORC_LOG10 * (10.0f));
    float const v = a * a;
    float   w = 0.0f;
    u -= athFloor;
    if (v > 1E-20f)
        w = 1.f + orc_fast_log2(T, v) * (ORC_LOG2 / ORC_LOG10 * (10.0f / o));
    if (w < 0)
        w = 0.f;
    u *= w;
    u += athFloor + o - p;
    return powf(10.f, 0.1f * u);
}

/* reference quantize_pvt.c:589-747 */
static int
calc_xmin(OrcStream * S, OrcRatio const *const ratio, OrcGr * const cod_info, float *pxmin)
{
    const LhConfig *cfg = S->cfg;
    const LhTables *T = S->tab;
    int     sfb, gsfb, j = 0, ath_over = 0, k;
    const float *const xr = cod_info->xr;
    int     max_nonzero;

    for (gsfb = 0; gsfb < cod_info->psy_lmax; gsfb++) {
        float   en0, xmin;
        float   rh1, rh2, rh3;
        int     width, l;
        xmin = orc_ath_adjust(T, S->ath_adjust_factor, T->ath_l[gsfb], T->ath_floor, cfg->ATHfixpoint);
        xmin *= T->longfact[gsfb];
        width = cod_info->width[gsfb];
        rh1 = xmin / width;
        rh2 = 2.2204460492503131e-16;   /* DBL_EPSILON */
        en0 = 0.0;
        for (l = 0; l < width; ++l) {
            float const xa = xr[j++];
            float const x2 = xa * xa;
            en0 += x2;
            rh2 += (x2 < rh1) ? x2 : rh1;
        }
        if (en0 > xmin)
            ath_over++;
        if (en0 < xmin)
            rh3 = en0;
        else if (rh2 < xmin)
            rh3 = xmin;
        else
            rh3 = rh2;
        xmin = rh3;
        {
            float const e = ratio->en.l[gsfb];
            if (e > 1e-12f) {
                float   x;
                x = en0 * ratio->thm.l[gsfb] / e;
                x *= T->longfact[gsfb];
                if (xmin < x)
                    xmin = x;
            }
        }
        xmin = (xmin > 2.2204460492503131e-16) ? xmin : 2.2204460492503131e-16;
        cod_info->energy_above_cutoff[gsfb] = (en0 > xmin + 1e-14f) ? 1 : 0;
        *pxmin++ = xmin;
    }
    max_nonzero = 0;
    for (k = 575; k > 0; --k) {
        if (fabs(xr[k]) > 1e-12f) {
            max_nonzero = k;
            break;
        }
    }
    if (cod_info->block_type != LH_SHORT_TYPE)
        max_nonzero |= 1;
    else {
        max_nonzero /= 6;
        max_nonzero *= 6;
        max_nonzero += 5;
    }
    if (cfg->sfb21_extra == 0 && cfg->samplerate < 44000) {
        int const sfb_l = (cfg->samplerate <= 8000) ? 17 : 21;
        int const sfb_s = (cfg->samplerate <= 8000) ? 9 : 12;
        int     limit = 575;
        if (cod_info->block_type != LH_SHORT_TYPE)
            limit = T->sfb_l[sfb_l] - 1;
        else
            limit = 3 * T->sfb_s[sfb_s] - 1;
        if (max_nonzero > limit)
            max_nonzero = limit;
    }
    cod_info->max_nonzero_coeff = max_nonzero;

    for (sfb = cod_info->sfb_smin; gsfb < cod_info->psymax; sfb++, gsfb += 3) {
        int     width, b, l;
        float   tmpATH;
        tmpATH = orc_ath_adjust(T, S->ath_adjust_factor, T->ath_s[sfb], T->ath_floor, cfg->ATHfixpoint);
        tmpATH *= T->shortfact[sfb];
        width = cod_info->width[gsfb];
        for (b = 0; b < 3; b++) {
            float   en0 = 0.0, xmin = tmpATH;
            float   rh1, rh2, rh3;
            rh1 = tmpATH / width;
            rh2 = 2.2204460492503131e-16;
            for (l = 0; l < width; ++l) {
                float const xa = xr[j++];
                float const x2 = xa * xa;
                en0 += x2;
                rh2 += (x2 < rh1) ? x2 : rh1;
            }
            if (en0 > tmpATH)
                ath_over++;
            if (en0 < tmpATH)
                rh3 = en0;
            else if (rh2 < tmpATH)
                rh3 = tmpATH;
            else
                rh3 = rh2;
            xmin = rh3;
            {
                float const e = ratio->en.s[sfb][b];
                if (e > 1e-12f) {
                    float   x;
                    x = en0 * ratio->thm.s[sfb][b] / e;
                    x *= T->shortfact[sfb];
                    if (xmin < x)
                        xmin = x;
                }
            }
            xmin = (xmin > 2.2204460492503131e-16) ? xmin : 2.2204460492503131e-16;
            cod_info->energy_above_cutoff[gsfb + b] = (en0 > xmin + 1e-14f) ? 1 : 0;
            *pxmin++ = xmin;
        }
        if (cfg->use_temporal_masking) {
            if (pxmin[-3] > pxmin[-3 + 1])
                pxmin[-3 + 1] += (pxmin[-3] - pxmin[-3 + 1]) * T->decay;
            if (pxmin[-3 + 1] > pxmin[-3 + 2])
                pxmin[-3 + 2] += (pxmin[-3 + 1] - pxmin[-3 + 2]) * T->decay;
        }
    }
    return ath_over;
}

/* reference quantize_pvt.c:750-796 */
static float
calc_noise_core(const LhTables * T, const OrcGr * const cod_info, int *startline, int l, float step)
{
    float   noise = 0;
    int     j = *startline;
    const int *const ix = cod_info->l3_enc;
    if (j > cod_info->count1) {
        while (l--) {
            float   temp;
            temp = cod_info->xr[j];
            j++;
            noise += temp * temp;
            temp = cod_info->xr[j];
            j++;
            noise += temp * temp;
        }
    }
    else if (j > cod_info->big_values) {
        float   ix01[2];
        ix01[0] = 0;
        ix01[1] = step;
        while (l--) {
            float   temp;
            temp = fabs(cod_info->xr[j]) - ix01[ix[j]];
            j++;
            noise += temp * temp;
            temp = fabs(cod_info->xr[j]) - ix01[ix[j]];
            j++;
            noise += temp * temp;
        }
    }
    else {
        while (l--) {
            float   temp;
            temp = fabs(cod_info->xr[j]) - T->pow43[ix[j]] * step;
            j++;
            noise += temp * temp;
            temp = fabs(cod_info->xr[j]) - T->pow43[ix[j]] * step;
            j++;
            noise += temp * temp;
        }
    }
    *startline = j;
    return noise;
}

/* reference quantize_pvt.c:815-913 */
static int
calc_noise(const LhTables * T, OrcGr const *const cod_info, float const *l3_xmin, float *distort,
           OrcNoiseResult * const res, OrcNoiseData * prev_noise)
{
    int     sfb, l, over = 0;
    float   over_noise_db = 0;
    float   tot_noise_db = 0;
    float   max_noise = -20.0;
    int     j = 0;
    const int *scalefac = cod_info->scalefac;

    res->over_SSD = 0;
    for (sfb = 0; sfb < cod_info->psymax; sfb++) {
        int const s = cod_info->global_gain
            - (((*scalefac++) + (cod_info->preflag ? lh_pretab[sfb] : 0))
               << (cod_info->scalefac_scale + 1))
            - cod_info->subblock_gain[cod_info->window[sfb]] * 8;
        float const r_l3_xmin = 1.f / *l3_xmin++;
        float   distort_ = 0.0f;
        float   noise = 0.0f;

        if (prev_noise && (prev_noise->step[sfb] == s)) {
            j += cod_info->width[sfb];
            distort_ = r_l3_xmin * prev_noise->noise[sfb];
            noise = prev_noise->noise_log[sfb];
        }
        else {
            float const step = POW20(T, s);
            l = cod_info->width[sfb] >> 1;
            if ((j + cod_info->width[sfb]) > cod_info->max_nonzero_coeff) {
                int     usefullsize = cod_info->max_nonzero_coeff - j + 1;
                if (usefullsize > 0)
                    l = usefullsize >> 1;
                else
                    l = 0;
            }
            noise = calc_noise_core(T, cod_info, &j, l, step);
            if (prev_noise) {
                prev_noise->step[sfb] = s;
                prev_noise->noise[sfb] = noise;
            }
            distort_ = r_l3_xmin * noise;
            noise = orc_fast_log2(T, (distort_ > 1E-20f) ? distort_ : 1E-20f) * (ORC_LOG2 / ORC_LOG10);
            if (prev_noise)
                prev_noise->noise_log[sfb] = noise;
        }
        *distort++ = distort_;
        if (prev_noise)
            prev_noise->global_gain = cod_info->global_gain;
        tot_noise_db += noise;
        if (noise > 0.0) {
            int     tmp;
            tmp = (int) (noise * 10 + .5);
            if (tmp < 1)
                tmp = 1;
            res->over_SSD += tmp * tmp;
            over++;
            over_noise_db += noise;
        }
        max_noise = (max_noise > noise) ? max_noise : noise;
    }
    res->over_count = over;
    res->tot_noise = tot_noise_db;
    res->over_noise = over_noise_db;
    res->max_noise = max_noise;
    return over;
}

/* ---------------------------------------------------------------------- */
/* reservoir (reference reservoir.c:82-293, bitstream.c:60-88)              */
static int
getframebits(OrcStream * S)
{
    int     bit_rate = (S->cfg->version ? lh_bitrate_mpeg1 : lh_bitrate_mpeg2)[S->bitrate_index];
    return 8 * ((S->cfg->version + 1) * 72000 * bit_rate / S->cfg->samplerate + S->padding);
}

static int
ResvFrameBegin(OrcStream * S, int *mean_bits)
{
    const LhConfig *cfg = S->cfg;
    int     fullFrameBits, resvLimit, maxmp3buf, frameLength, meanBits;
    frameLength = getframebits(S);
    meanBits = (frameLength - cfg->sideinfo_len * 8) / cfg->mode_gr;
    resvLimit = (8 * 256) * cfg->mode_gr - 8;
    maxmp3buf = cfg->buffer_constraint;
    S->ResvMax = maxmp3buf - frameLength;
    if (S->ResvMax > resvLimit)
        S->ResvMax = resvLimit;
    if (S->ResvMax < 0 || cfg->disable_reservoir)
        S->ResvMax = 0;
    fullFrameBits = meanBits * cfg->mode_gr + ((S->ResvSize < S->ResvMax) ? S->ResvSize : S->ResvMax);
    if (fullFrameBits > maxmp3buf)
        fullFrameBits = maxmp3buf;
    S->resvDrain_pre = 0;
    *mean_bits = meanBits;
    return fullFrameBits;
}

static void
ResvMaxBits(OrcStream * S, int mean_bits, int *targ_bits, int *extra_bits, int cbr)
{
    int     add_bits, targBits, extraBits;
    int     ResvSize = S->ResvSize, ResvMax = S->ResvMax;
    if (cbr)
        ResvSize += mean_bits;
    if (S->substep_shaping & 1)
        ResvMax *= 0.9;
    targBits = mean_bits;
    if (ResvSize * 10 > ResvMax * 9) {
        add_bits = ResvSize - (ResvMax * 9) / 10;
        targBits += add_bits;
        S->substep_shaping |= 0x80;
    }
    else {
        add_bits = 0;
        S->substep_shaping &= 0x7f;
        if (!S->cfg->disable_reservoir && !(S->substep_shaping & 1))
            targBits -= .1 * mean_bits;
    }
    extraBits = (ResvSize < (S->ResvMax * 6) / 10 ? ResvSize : (S->ResvMax * 6) / 10);
    extraBits -= add_bits;
    if (extraBits < 0)
        extraBits = 0;
    *targ_bits = targBits;
    *extra_bits = extraBits;
}

static void
ResvFrameEnd(OrcStream * S, int mean_bits)
{
    int     stuffingBits, over_bits;
    S->ResvSize += mean_bits * S->cfg->mode_gr;
    stuffingBits = 0;
    S->resvDrain_post = 0;
    S->resvDrain_pre = 0;
    if ((over_bits = S->ResvSize % 8) != 0)
        stuffingBits += over_bits;
    over_bits = (S->ResvSize - stuffingBits) - S->ResvMax;
    if (over_bits > 0)
        stuffingBits += over_bits;
    {
        int     m = S->main_data_begin * 8;
        int     mdb_bytes = ((m < stuffingBits) ? m : stuffingBits) / 8;
        S->resvDrain_pre += 8 * mdb_bytes;
        stuffingBits -= 8 * mdb_bytes;
        S->ResvSize -= 8 * mdb_bytes;
        S->main_data_begin -= mdb_bytes;
    }
    S->resvDrain_post += stuffingBits;
    S->ResvSize -= stuffingBits;
}

/* reference quantize_pvt.c:428-487 */
static int
on_pe(OrcStream * S, float pe[][2], int targ_bits[2], int mean_bits, int gr, int cbr)
{
    int     extra_bits = 0, tbits, bits;
    int     add_bits[2] = { 0, 0 };
    int     max_bits, ch;
    int const nch = S->cfg->channels;

    ResvMaxBits(S, mean_bits, &tbits, &extra_bits, cbr);
    max_bits = tbits + extra_bits;
    if (max_bits > LH_MAX_BITS_PER_GRANULE)
        max_bits = LH_MAX_BITS_PER_GRANULE;
    for (bits = 0, ch = 0; ch < nch; ++ch) {
        targ_bits[ch] = (LH_MAX_BITS_PER_CHANNEL < tbits / nch) ? LH_MAX_BITS_PER_CHANNEL : tbits / nch;
        add_bits[ch] = targ_bits[ch] * pe[gr][ch] / 700.0 - targ_bits[ch];
        if (add_bits[ch] > mean_bits * 3 / 4)
            add_bits[ch] = mean_bits * 3 / 4;
        if (add_bits[ch] < 0)
            add_bits[ch] = 0;
        if (add_bits[ch] + targ_bits[ch] > LH_MAX_BITS_PER_CHANNEL) {
            int     v = LH_MAX_BITS_PER_CHANNEL - targ_bits[ch];
            add_bits[ch] = (0 > v) ? 0 : v;
        }
        bits += add_bits[ch];
    }
    if (bits > extra_bits && bits > 0) {
        for (ch = 0; ch < nch; ++ch)
            add_bits[ch] = extra_bits * add_bits[ch] / bits;
    }
    for (ch = 0; ch < nch; ++ch) {
        targ_bits[ch] += add_bits[ch];
        extra_bits -= add_bits[ch];
    }
    for (bits = 0, ch = 0; ch < nch; ++ch)
        bits += targ_bits[ch];
    if (bits > LH_MAX_BITS_PER_GRANULE) {
        for (ch = 0; ch < nch; ++ch) {
            targ_bits[ch] *= LH_MAX_BITS_PER_GRANULE;
            targ_bits[ch] /= bits;
        }
    }
    return max_bits;
}

/* reference quantize_pvt.c:492-545 */
static void
reduce_side(int targ_bits[2], float ms_ener_ratio, int mean_bits, int max_bits)
{
    int     move_bits;
    float   fac;
    fac = .33 * (.5 - ms_ener_ratio) / .5;
    if (fac < 0)
        fac = 0;
    if (fac > .5)
        fac = .5;
    move_bits = fac * .5 * (targ_bits[0] + targ_bits[1]);
    if (move_bits > LH_MAX_BITS_PER_CHANNEL - targ_bits[0])
        move_bits = LH_MAX_BITS_PER_CHANNEL - targ_bits[0];
    if (move_bits < 0)
        move_bits = 0;
    if (targ_bits[1] >= 125) {
        if (targ_bits[1] - move_bits > 125) {
            if (targ_bits[0] < mean_bits)
                targ_bits[0] += move_bits;
            targ_bits[1] -= move_bits;
        }
        else {
            targ_bits[0] += targ_bits[1] - 125;
            targ_bits[1] = 125;
        }
    }
    move_bits = targ_bits[0] + targ_bits[1];
    if (move_bits > max_bits) {
        targ_bits[0] = (max_bits * targ_bits[0]) / move_bits;
        targ_bits[1] = (max_bits * targ_bits[1]) / move_bits;
    }
}

/* reference quantize.c:159-223: the old VBR loop drops what lies below the threshold in quiet at the top of the
 * spectrum (the parts of scalefactor band 21 / 12), from the highest line downwards, before it looks at a granule */
static void
psfb21_analogsilence(OrcStream * S, OrcGr * const cod_info)
{
    const LhTables *T = S->tab;
    float  *const xr = cod_info->xr;
    if (cod_info->block_type != LH_SHORT_TYPE) {
        int     gsfb, stop = 0;
        for (gsfb = LH_PSFB21 - 1; gsfb >= 0 && !stop; gsfb--) {
            int const start = T->psfb21[gsfb];
            int const end = T->psfb21[gsfb + 1];
            int     j;
            float   ath21 = orc_ath_adjust(T, S->ath_adjust_factor, T->ath_psfb21[gsfb], T->ath_floor, 0);
            if (T->longfact[21] > 1e-12f)
                ath21 *= T->longfact[21];
            for (j = end - 1; j >= start; j--) {
                if (fabs(xr[j]) < ath21)
                    xr[j] = 0;
                else {
                    stop = 1;
                    break;
                }
            }
        }
    }
    else {
        int     block;
        for (block = 0; block < 3; block++) {
            int     gsfb, stop = 0;
            for (gsfb = LH_PSFB12 - 1; gsfb >= 0 && !stop; gsfb--) {
                int const start = T->sfb_s[12] * 3 + (T->sfb_s[13] - T->sfb_s[12]) * block
                    + (T->psfb12[gsfb] - T->psfb12[0]);
                int const end = start + (T->psfb12[gsfb + 1] - T->psfb12[gsfb]);
                int     j;
                float   ath12 = orc_ath_adjust(T, S->ath_adjust_factor, T->ath_psfb12[gsfb], T->ath_floor, 0);
                if (T->shortfact[12] > 1e-12f)
                    ath12 *= T->shortfact[12];
                for (j = end - 1; j >= start; j--) {
                    if (fabs(xr[j]) < ath12)
                        xr[j] = 0;
                    else {
                        stop = 1;
                        break;
                    }
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------- */
/* reference quantize.c:226-346 */
static void
init_outer_loop(OrcStream * S, OrcGr * const cod_info)
{
    const LhTables *T = S->tab;
    int     sfb, j;
    cod_info->part2_3_length = 0;
    cod_info->big_values = 0;
    cod_info->count1 = 0;
    cod_info->global_gain = 210;
    cod_info->scalefac_compress = 0;
    cod_info->table_select[0] = 0;
    cod_info->table_select[1] = 0;
    cod_info->table_select[2] = 0;
    cod_info->subblock_gain[0] = 0;
    cod_info->subblock_gain[1] = 0;
    cod_info->subblock_gain[2] = 0;
    cod_info->subblock_gain[3] = 0;
    cod_info->region0_count = 0;
    cod_info->region1_count = 0;
    cod_info->preflag = 0;
    cod_info->scalefac_scale = 0;
    cod_info->count1table_select = 0;
    cod_info->part2_length = 0;
    if (S->cfg->samplerate <= 8000) {
        /* an 8 kHz stream codes 17 long / 9 short bands (reference quantize.c:252-256) */
        cod_info->sfb_lmax = 17;
        cod_info->sfb_smin = 9;
        cod_info->psy_lmax = 17;
    }
    else {
        cod_info->sfb_lmax = LH_SBPSY_L;
        cod_info->sfb_smin = LH_SBPSY_S;
        cod_info->psy_lmax = S->cfg->sfb21_extra ? LH_SBMAX_L : LH_SBPSY_L;
    }
    cod_info->psymax = cod_info->psy_lmax;
    cod_info->sfbmax = cod_info->sfb_lmax;
    cod_info->sfbdivide = 11;
    for (sfb = 0; sfb < LH_SBMAX_L; sfb++) {
        cod_info->width[sfb] = T->sfb_l[sfb + 1] - T->sfb_l[sfb];
        cod_info->window[sfb] = 3;
    }
    if (cod_info->block_type == LH_SHORT_TYPE) {
        float   ixwork[576];
        float  *ix;
        cod_info->sfb_smin = 0;
        cod_info->sfb_lmax = 0;
        if (S->cfg->samplerate <= 8000) {
            cod_info->psymax = cod_info->sfb_lmax + 3 * (9 - cod_info->sfb_smin);
            cod_info->sfbmax = cod_info->sfb_lmax + 3 * (9 - cod_info->sfb_smin);
        }
        else {
            cod_info->psymax = cod_info->sfb_lmax
                + 3 * ((S->cfg->sfb21_extra ? LH_SBMAX_S : LH_SBPSY_S) - cod_info->sfb_smin);
            cod_info->sfbmax = cod_info->sfb_lmax + 3 * (LH_SBPSY_S - cod_info->sfb_smin);
        }
        cod_info->sfbdivide = cod_info->sfbmax - 18;
        cod_info->psy_lmax = cod_info->sfb_lmax;
        ix = &cod_info->xr[T->sfb_l[cod_info->sfb_lmax]];
        memcpy(ixwork, cod_info->xr, 576 * sizeof(float));
        for (sfb = cod_info->sfb_smin; sfb < LH_SBMAX_S; sfb++) {
            int const start = T->sfb_s[sfb];
            int const end = T->sfb_s[sfb + 1];
            int     window, l;
            for (window = 0; window < 3; window++)
                for (l = start; l < end; l++)
                    *ix++ = ixwork[3 * l + window];
        }
        j = cod_info->sfb_lmax;
        for (sfb = cod_info->sfb_smin; sfb < LH_SBMAX_S; sfb++) {
            cod_info->width[j] = cod_info->width[j + 1] = cod_info->width[j + 2]
                = T->sfb_s[sfb + 1] - T->sfb_s[sfb];
            cod_info->window[j] = 0;
            cod_info->window[j + 1] = 1;
            cod_info->window[j + 2] = 2;
            j += 3;
        }
    }
    cod_info->count1bits = 0;
    cod_info->max_nonzero_coeff = 575;
    memset(cod_info->scalefac, 0, sizeof(cod_info->scalefac));
    if (S->cfg->vbr == 2)       /* vbr_rh only, reference quantize.c:343-345 */
        psfb21_analogsilence(S, cod_info);
}

/* reference quantize.c:72-144 */
static int
init_xrpow(OrcStream * S, OrcGr * const cod_info, float xrpow[576])
{
    float   sum = 0;
    int     i;
    int const upper = cod_info->max_nonzero_coeff;
    cod_info->xrpow_max = 0;
    memset(&(xrpow[upper]), 0, (576 - upper) * sizeof(xrpow[0]));
    for (i = 0; i <= upper; ++i) {
        float   tmp = fabs(cod_info->xr[i]);
        sum += tmp;
        xrpow[i] = sqrt(tmp * sqrt(tmp));
        if (xrpow[i] > cod_info->xrpow_max)
            cod_info->xrpow_max = xrpow[i];
    }
    if (sum > (float) 1E-20) {
        int     j = 0;
        if (S->substep_shaping & 2)
            j = 1;
        for (i = 0; i < cod_info->psymax; i++)
            S->pseudohalf[i] = j;
        return 1;
    }
    memset(&cod_info->l3_enc[0], 0, sizeof(int) * 576);
    return 0;
}

/* reference quantize.c:367-429 */
static int
bin_search_StepSize(OrcStream * S, OrcGr * const cod_info, int desired_rate, const int ch,
                    const float xrpow[576])
{
    int     nBits;
    int     CurrentStep = S->CurrentStep[ch];
    int     flag_GoneOver = 0;
    int const start = S->OldValue[ch];
    int     Direction = 0;      /* 0 none, 1 up, 2 down */
    cod_info->global_gain = start;
    desired_rate -= cod_info->part2_length;
    for (;;) {
        int     step;
        nBits = count_bits(S, xrpow, cod_info, 0);
        if (CurrentStep == 1 || nBits == desired_rate)
            break;
        if (nBits > desired_rate) {
            if (Direction == 2)
                flag_GoneOver = 1;
            if (flag_GoneOver)
                CurrentStep /= 2;
            Direction = 1;
            step = CurrentStep;
        }
        else {
            if (Direction == 1)
                flag_GoneOver = 1;
            if (flag_GoneOver)
                CurrentStep /= 2;
            Direction = 2;
            step = -CurrentStep;
        }
        cod_info->global_gain += step;
        if (cod_info->global_gain < 0) {
            cod_info->global_gain = 0;
            flag_GoneOver = 1;
        }
        if (cod_info->global_gain > 255) {
            cod_info->global_gain = 255;
            flag_GoneOver = 1;
        }
    }
    while (nBits > desired_rate && cod_info->global_gain < 255) {
        cod_info->global_gain++;
        nBits = count_bits(S, xrpow, cod_info, 0);
    }
    S->CurrentStep[ch] = (start - cod_info->global_gain >= 4) ? 4 : 2;
    S->OldValue[ch] = cod_info->global_gain;
    cod_info->part2_3_length = nBits;
    return nBits;
}

/* reference quantize.c:540-551 */
static int
loop_break(const OrcGr * const cod_info)
{
    int     sfb;
    for (sfb = 0; sfb < cod_info->sfbmax; sfb++)
        if (cod_info->scalefac[sfb] + cod_info->subblock_gain[cod_info->window[sfb]] == 0)
            return 0;
    return 1;
}

/* reference quantize.c:585-686, only the comparator every preset on this path selects (9),
 * so any other value is treated like the reference's `default:' label, i.e. 9) */
static int
quant_compare(const int quant_comp, const OrcNoiseResult * const best, OrcNoiseResult * const calc)
{
    int     better;
    switch (quant_comp) {
    default:
    case 9:
        if (best->over_count > 0) {
            better = calc->over_SSD <= best->over_SSD;
            if (calc->over_SSD == best->over_SSD)
                better = calc->bits < best->bits;
        }
        else {
            better = ((calc->max_noise < 0) &&
                      ((calc->max_noise * 10 + calc->bits) <= (best->max_noise * 10 + best->bits)));
        }
        break;
    }
    if (best->over_count == 0)
        better = better && calc->bits < best->bits;
    return better;
}

/* reference quantize.c:720-796 */
static void
amp_scalefac_bands(OrcStream * S, OrcGr * const cod_info, float const *distort, float xrpow[576],
                   int bRefine)
{
    const LhConfig *cfg = S->cfg;
    int     j, sfb;
    float   ifqstep34, trigger;
    int     noise_shaping_amp;

    if (cod_info->scalefac_scale == 0)
        ifqstep34 = 1.29683955465100964055;
    else
        ifqstep34 = 1.68179283050742922612;
    trigger = 0;
    for (sfb = 0; sfb < cod_info->sfbmax; sfb++)
        if (trigger < distort[sfb])
            trigger = distort[sfb];
    noise_shaping_amp = cfg->noise_shaping_amp;
    if (noise_shaping_amp == 3) {
        if (bRefine == 1)
            noise_shaping_amp = 2;
        else
            noise_shaping_amp = 1;
    }
    switch (noise_shaping_amp) {
    case 2:
        break;
    case 1:
        if (trigger > 1.0)
            trigger = pow(trigger, .5);
        else
            trigger *= .95;
        break;
    case 0:
    default:
        if (trigger > 1.0)
            trigger = 1.0;
        else
            trigger *= .95;
        break;
    }
    j = 0;
    for (sfb = 0; sfb < cod_info->sfbmax; sfb++) {
        int const width = cod_info->width[sfb];
        int     l;
        j += width;
        if (distort[sfb] < trigger)
            continue;
        if (S->substep_shaping & 2) {
            S->pseudohalf[sfb] = !S->pseudohalf[sfb];
            if (!S->pseudohalf[sfb] && cfg->noise_shaping_amp == 2)
                return;
        }
        cod_info->scalefac[sfb]++;
        for (l = -width; l < 0; l++) {
            xrpow[j + l] *= ifqstep34;
            if (xrpow[j + l] > cod_info->xrpow_max)
                cod_info->xrpow_max = xrpow[j + l];
        }
        if (cfg->noise_shaping_amp == 2)
            return;
    }
}

/* reference quantize.c:808-833 */
static void
inc_scalefac_scale(OrcGr * const cod_info, float xrpow[576])
{
    int     l, j, sfb;
    const float ifqstep34 = 1.29683955465100964055;
    j = 0;
    for (sfb = 0; sfb < cod_info->sfbmax; sfb++) {
        int const width = cod_info->width[sfb];
        int     s = cod_info->scalefac[sfb];
        if (cod_info->preflag)
            s += lh_pretab[sfb];
        j += width;
        if (s & 1) {
            s++;
            for (l = -width; l < 0; l++) {
                xrpow[j + l] *= ifqstep34;
                if (xrpow[j + l] > cod_info->xrpow_max)
                    cod_info->xrpow_max = xrpow[j + l];
            }
        }
        cod_info->scalefac[sfb] = s >> 1;
    }
    cod_info->preflag = 0;
    cod_info->scalefac_scale = 1;
}

/* reference quantize.c:847-921 */
static int
inc_subblock_gain(OrcStream * S, OrcGr * const cod_info, float xrpow[576])
{
    const LhTables *T = S->tab;
    int     sfb, window;
    int    *const scalefac = cod_info->scalefac;

    for (sfb = 0; sfb < cod_info->sfb_lmax; sfb++)
        if (scalefac[sfb] >= 16)
            return 1;
    for (window = 0; window < 3; window++) {
        int     s1, s2, l, j;
        s1 = s2 = 0;
        for (sfb = cod_info->sfb_lmax + window; sfb < cod_info->sfbdivide; sfb += 3)
            if (s1 < scalefac[sfb])
                s1 = scalefac[sfb];
        for (; sfb < cod_info->sfbmax; sfb += 3)
            if (s2 < scalefac[sfb])
                s2 = scalefac[sfb];
        if (s1 < 16 && s2 < 8)
            continue;
        if (cod_info->subblock_gain[window] >= 7)
            return 1;
        cod_info->subblock_gain[window]++;
        j = T->sfb_l[cod_info->sfb_lmax];
        for (sfb = cod_info->sfb_lmax + window; sfb < cod_info->sfbmax; sfb += 3) {
            float   amp;
            int const width = cod_info->width[sfb];
            int     s = scalefac[sfb];
            s = s - (4 >> cod_info->scalefac_scale);
            if (s >= 0) {
                scalefac[sfb] = s;
                j += width * 3;
                continue;
            }
            scalefac[sfb] = 0;
            {
                int const gain = 210 + (s << (cod_info->scalefac_scale + 1));
                amp = IPOW20(T, gain);
            }
            j += width * (window + 1);
            for (l = -width; l < 0; l++) {
                xrpow[j + l] *= amp;
                if (xrpow[j + l] > cod_info->xrpow_max)
                    cod_info->xrpow_max = xrpow[j + l];
            }
            j += width * (3 - window - 1);
        }
        {
            float const amp = IPOW20(T, 202);
            j += cod_info->width[sfb] * (window + 1);
            for (l = -cod_info->width[sfb]; l < 0; l++) {
                xrpow[j + l] *= amp;
                if (xrpow[j + l] > cod_info->xrpow_max)
                    cod_info->xrpow_max = xrpow[j + l];
            }
        }
    }
    return 0;
}

/* reference quantize.c:940-988 */
static int
balance_noise(OrcStream * S, OrcGr * const cod_info, float const *distort, float xrpow[576],
              int bRefine)
{
    const LhConfig *cfg = S->cfg;
    int     status;
    amp_scalefac_bands(S, cod_info, distort, xrpow, bRefine);
    status = loop_break(cod_info);
    if (status)
        return 0;
    status = scale_bitcount(S, cod_info);
    if (!status)
        return 1;
    if (cfg->noise_shaping > 1) {
        memset(&S->pseudohalf[0], 0, sizeof(S->pseudohalf));
        if (!cod_info->scalefac_scale) {
            inc_scalefac_scale(cod_info, xrpow);
            status = 0;
        }
        else {
            if (cod_info->block_type == LH_SHORT_TYPE && cfg->subblock_gain > 0)
                status = inc_subblock_gain(S, cod_info, xrpow) || loop_break(cod_info);
        }
    }
    if (!status)
        status = scale_bitcount(S, cod_info);
    return !status;
}

/* reference quantize.c:1010-1197 */
static int
outer_loop(OrcStream * S, OrcGr * const cod_info, const float *const l3_xmin, float xrpow[576],
           const int ch, const int targ_bits)
{
    const LhConfig *cfg = S->cfg;
    const LhTables *T = S->tab;
    OrcGr   cod_info_w;
    float   save_xrpow[576];
    float   distort[LH_SFBMAX];
    OrcNoiseResult best_noise_info;
    int     huff_bits;
    int     better;
    int     age;
    OrcNoiseData prev_noise;
    int     best_part2_3_length = 9999999;
    int     bEndOfSearch = 0;
    int     bRefine = 0;
    int     best_ggain_pass1 = 0;

    (void) bin_search_StepSize(S, cod_info, targ_bits, ch, xrpow);
    if (!cfg->noise_shaping)
        return 100;
    memset(&prev_noise, 0, sizeof(prev_noise));
    memset(&best_noise_info, 0, sizeof(best_noise_info));
    (void) calc_noise(T, cod_info, l3_xmin, distort, &best_noise_info, &prev_noise);
    best_noise_info.bits = cod_info->part2_3_length;
    cod_info_w = *cod_info;
    age = 0;
    memcpy(save_xrpow, xrpow, sizeof(float) * 576);

    while (!bEndOfSearch) {
        do {
            OrcNoiseResult noise_info;
            int     search_limit;
            int     maxggain = 255;
            memset(&noise_info, 0, sizeof(noise_info));
            if (S->substep_shaping & 2)
                search_limit = 20;
            else
                search_limit = 3;
            if (cfg->sfb21_extra && !S->sfb21_off) {
                if (distort[cod_info_w.sfbmax] > 1.0)
                    break;
                if (cod_info_w.block_type == LH_SHORT_TYPE
                    && (distort[cod_info_w.sfbmax + 1] > 1.0
                        || distort[cod_info_w.sfbmax + 2] > 1.0))
                    break;
            }
            if (balance_noise(S, &cod_info_w, distort, xrpow, bRefine) == 0)
                break;
            if (cod_info_w.scalefac_scale)
                maxggain = 254;
            huff_bits = targ_bits - cod_info_w.part2_length;
            if (huff_bits <= 0)
                break;
            while ((cod_info_w.part2_3_length
                    = count_bits(S, xrpow, &cod_info_w, &prev_noise)) > huff_bits
                   && cod_info_w.global_gain <= maxggain)
                cod_info_w.global_gain++;
            if (cod_info_w.global_gain > maxggain)
                break;
            if (best_noise_info.over_count == 0) {
                while ((cod_info_w.part2_3_length
                        = count_bits(S, xrpow, &cod_info_w, &prev_noise)) > best_part2_3_length
                       && cod_info_w.global_gain <= maxggain)
                    cod_info_w.global_gain++;
                if (cod_info_w.global_gain > maxggain)
                    break;
            }
            (void) calc_noise(T, &cod_info_w, l3_xmin, distort, &noise_info, &prev_noise);
            noise_info.bits = cod_info_w.part2_3_length;
            if (cod_info->block_type != LH_SHORT_TYPE)
                better = cfg->quant_comp;
            else
                better = cfg->quant_comp_short;
            better = quant_compare(better, &best_noise_info, &noise_info);
            if (better) {
                best_part2_3_length = cod_info->part2_3_length;
                best_noise_info = noise_info;
                *cod_info = cod_info_w;
                age = 0;
                memcpy(save_xrpow, xrpow, sizeof(float) * 576);
            }
            else {
                if (cfg->full_outer_loop == 0) {
                    if (++age > search_limit && best_noise_info.over_count == 0)
                        break;
                    if ((cfg->noise_shaping_amp == 3) && bRefine && age > 30)
                        break;
                    if ((cfg->noise_shaping_amp == 3) && bRefine &&
                        (cod_info_w.global_gain - best_ggain_pass1) > 15)
                        break;
                }
            }
        }
        while ((cod_info_w.global_gain + cod_info_w.scalefac_scale) < 255);

        if (cfg->noise_shaping_amp == 3) {
            if (!bRefine) {
                cod_info_w = *cod_info;
                memcpy(xrpow, save_xrpow, sizeof(float) * 576);
                age = 0;
                best_ggain_pass1 = cod_info_w.global_gain;
                bRefine = 1;
            }
            else
                bEndOfSearch = 1;
        }
        else
            bEndOfSearch = 1;
    }
    /* substep_shaping & 1 (trancate_smallspectrums) is never set on the CBR path
     * (quality switch sets 0 or 2, reference lame.c:437-463) */
    if (cfg->vbr == 2)          /* restore for reuse on next try, reference quantize.c:1188-1190 */
        memcpy(xrpow, save_xrpow, sizeof(float) * 576);
    return best_noise_info.over_count;
}

/* reference quantize.c:1988-2050 */
void
orc_cbr_iteration_loop(OrcStream * S, float pe[2][2], const float ms_ener_ratio[2],
                       const OrcRatio ratio[2][2])
{
    const LhConfig *cfg = S->cfg;
    float   l3_xmin[LH_SFBMAX];
    float   xrpow[576];
    int     targ_bits[2];
    int     mean_bits, max_bits;
    int     gr, ch;

    (void) ResvFrameBegin(S, &mean_bits);
    for (gr = 0; gr < S->cfg->mode_gr; gr++) {
        max_bits = on_pe(S, pe, targ_bits, mean_bits, gr, gr);
        if (S->mode_ext == LH_MPG_MD_MS_LR) {
            /* ms_convert, reference quantize.c:48-59 */
            int     i;
            for (i = 0; i < 576; ++i) {
                float   l = S->tt[gr][0].xr[i];
                float   r = S->tt[gr][1].xr[i];
                S->tt[gr][0].xr[i] = (l + r) * (float) (ORC_SQRT2 * 0.5);
                S->tt[gr][1].xr[i] = (l - r) * (float) (ORC_SQRT2 * 0.5);
            }
            reduce_side(targ_bits, ms_ener_ratio[gr], mean_bits, max_bits);
        }
        for (ch = 0; ch < cfg->channels; ch++) {
            OrcGr  *cod_info = &S->tt[gr][ch];
            if (cod_info->block_type != LH_SHORT_TYPE)
                S->masking_lower = cfg->masking_lower_long;
            else
                S->masking_lower = cfg->masking_lower_short;
            init_outer_loop(S, cod_info);
            if (init_xrpow(S, cod_info, xrpow)) {
                (void) calc_xmin(S, &ratio[gr][ch], cod_info, l3_xmin);
                (void) outer_loop(S, cod_info, l3_xmin, xrpow, ch, targ_bits[ch]);
            }
            /* iteration_finish_one, reference quantize.c:1213-1232 */
            best_scalefac_store(S, gr, ch);
            if (cfg->use_best_huffman == 1)
                best_huffman_divide(S, cod_info);
            S->ResvSize -= cod_info->part2_3_length + cod_info->part2_length;
        }
    }
    ResvFrameEnd(S, mean_bits);
}
