/*
 * orc_vbr_old.c -- CPU oracle, the old VBR loop (vbr_rh): TEST INFRASTRUCTURE ONLY (see orc_common.h).
 *
 * reference quantize.c:1245-1331 (VBR_encode_granule), 1340-1362 (get_framebits), 1391-1454
 * (VBR_old_prepare), 1456-1480 (bitpressure_strategy), 1491-1578 (VBR_old_iteration_loop).
 *
 * The loop looks, granule by granule, for the smallest bit budget at which the CBR search (outer_loop) leaves
 * no band with audible noise: a bisection over the budget, each trial continuing from the best quantisation
 * found so far.  The frame then takes the smallest bitrate that holds what the four searches used; should even the
 * largest one be too small, the allowed noise is raised and everything runs once more.
 */
#include "orc_common.h"
#include <stdio.h>
#include <stdlib.h>

/* reference quantize.c:1245-1331 */
static void
vbr_old_encode_granule(OrcStream * S, OrcGr * const cod_info, const float *const l3_xmin, float xrpow[576],
                       const int ch, int min_bits, int max_bits)
{
    static OrcGr bst_cod_info;
    float   bst_xrpow[576];
    int const Max_bits = max_bits;
    int     real_bits = max_bits + 1;
    int     this_bits = (max_bits + min_bits) / 2;
    int     dbits, over, found = 0;

    memset(bst_cod_info.l3_enc, 0, sizeof(bst_cod_info.l3_enc));
    do {
        S->sfb21_off = (this_bits > Max_bits - 42);
        over = outer_loop(S, cod_info, l3_xmin, xrpow, ch, this_bits);
        if (over <= 0) {
            /* no band is distorted: it can be done with real_bits; keep it and try fewer */
            found = 1;
            real_bits = cod_info->part2_3_length;
            bst_cod_info = *cod_info;
            memcpy(bst_xrpow, xrpow, sizeof(float) * 576);
            max_bits = real_bits - 32;
            dbits = max_bits - min_bits;
            this_bits = (max_bits + min_bits) / 2;
        }
        else {
            /* try more, from the best quantisation so far */
            min_bits = this_bits + 32;
            dbits = max_bits - min_bits;
            this_bits = (max_bits + min_bits) / 2;
            if (found) {
                found = 2;
                *cod_info = bst_cod_info;
                memcpy(xrpow, bst_xrpow, sizeof(float) * 576);
            }
        }
    } while (dbits > 12);
    S->sfb21_off = 0;
    if (found == 2)
        memcpy(cod_info->l3_enc, bst_cod_info.l3_enc, sizeof(int) * 576);
    (void) real_bits;
}

/* reference quantize.c:1391-1454 */
static int
vbr_old_prepare(OrcStream * S, float pe[2][2], const float ms_ener_ratio[2], const OrcRatio ratio[2][2],
                float l3_xmin[2][2][LH_SFBMAX], int frameBits[16], int min_bits[2][2], int max_bits[2][2])
{
    const LhConfig *cfg = S->cfg;
    float   masking_lower_db, adjust = 0.0;
    int     gr, ch, i;
    int     analog_silence = 1;
    int     avg, mxb, bits = 0, dummy;

    S->bitrate_index = cfg->vbr_max_bitrate_index;
    avg = ResvFrameBegin(S, &avg) / cfg->mode_gr;
    for (i = 1; i <= cfg->vbr_max_bitrate_index; i++) {        /* get_framebits, reference quantize.c:1340-1362 */
        S->bitrate_index = i;
        frameBits[i] = ResvFrameBegin(S, &dummy);
    }
    for (gr = 0; gr < cfg->mode_gr; gr++) {
        mxb = on_pe(S, pe, max_bits[gr], avg, gr, 0);
        if (S->mode_ext == LH_MPG_MD_MS_LR) {
            for (i = 0; i < 576; ++i) {         /* ms_convert, reference quantize.c:48-59 */
                float   l = S->tt[gr][0].xr[i];
                float   r = S->tt[gr][1].xr[i];
                S->tt[gr][0].xr[i] = (l + r) * (float) (ORC_SQRT2 * 0.5);
                S->tt[gr][1].xr[i] = (l - r) * (float) (ORC_SQRT2 * 0.5);
            }
            reduce_side(max_bits[gr], ms_ener_ratio[gr], avg, mxb);
        }
        for (ch = 0; ch < cfg->channels; ++ch) {
            OrcGr  *const cod_info = &S->tt[gr][ch];
            if (cod_info->block_type != LH_SHORT_TYPE) {
                adjust = 1.28 / (1 + exp(3.5 - pe[gr][ch] / 300.)) - 0.05;
                masking_lower_db = cfg->mask_adjust - adjust;
            }
            else {
                adjust = 2.56 / (1 + exp(3.5 - pe[gr][ch] / 300.)) - 0.14;
                masking_lower_db = cfg->mask_adjust_short - adjust;
            }
            S->masking_lower = pow(10.0, masking_lower_db * 0.1);
            init_outer_loop(S, cod_info);
            if (calc_xmin(S, &ratio[gr][ch], cod_info, l3_xmin[gr][ch]))
                analog_silence = 0;
            min_bits[gr][ch] = 126;
            bits += max_bits[gr][ch];
        }
    }
    for (gr = 0; gr < cfg->mode_gr; gr++)
        for (ch = 0; ch < cfg->channels; ch++) {
            if (bits > frameBits[cfg->vbr_max_bitrate_index] && bits > 0) {
                max_bits[gr][ch] *= frameBits[cfg->vbr_max_bitrate_index];
                max_bits[gr][ch] /= bits;
            }
            if (min_bits[gr][ch] > max_bits[gr][ch])
                min_bits[gr][ch] = max_bits[gr][ch];
        }
    return analog_silence;
}

/* reference quantize.c:1456-1480 */
static void
vbr_old_bitpressure(OrcStream * S, float l3_xmin[2][2][LH_SFBMAX], int min_bits[2][2], int max_bits[2][2])
{
    const LhConfig *cfg = S->cfg;
    int     gr, ch, sfb;
    for (gr = 0; gr < cfg->mode_gr; gr++)
        for (ch = 0; ch < cfg->channels; ch++) {
            OrcGr const *const gi = &S->tt[gr][ch];
            float  *pxmin = l3_xmin[gr][ch];
            for (sfb = 0; sfb < gi->psy_lmax; sfb++)
                *pxmin++ *= 1. + .029 * sfb * sfb / LH_SBMAX_L / LH_SBMAX_L;
            if (gi->block_type == LH_SHORT_TYPE)
                for (sfb = gi->sfb_smin; sfb < LH_SBMAX_S; sfb++) {
                    *pxmin++ *= 1. + .029 * sfb * sfb / LH_SBMAX_S / LH_SBMAX_S;
                    *pxmin++ *= 1. + .029 * sfb * sfb / LH_SBMAX_S / LH_SBMAX_S;
                    *pxmin++ *= 1. + .029 * sfb * sfb / LH_SBMAX_S / LH_SBMAX_S;
                }
            max_bits[gr][ch] = (min_bits[gr][ch] > 0.9 * max_bits[gr][ch]) ? min_bits[gr][ch] : 0.9 * max_bits[gr][ch];
        }
}

/* reference quantize.c:1491-1578 */
void
orc_vbr_old_iteration_loop(OrcStream * S, float pe[2][2], const float ms_ener_ratio[2], const OrcRatio ratio[2][2])
{
    const LhConfig *cfg = S->cfg;
    float   l3_xmin[2][2][LH_SFBMAX];
    float   xrpow[576];
    int     frameBits[16];
    int     used_bits, bits, mean_bits;
    int     min_bits[2][2], max_bits[2][2];
    int     ch, gr, analog_silence;

    /* Test hook: the budgets of VBR_old_prepare add up to what the largest frame holds and every search stays inside
     * its budget, so real input does not reach the second pass (only a search that ends at global_gain 255 could);
     * LH_TEST_FORCE_PRESSURE=n makes the first n evaluations of a frame fail, here and in the emulator build of
     * the kernel alike, so that the pass is compared at all. */
    int     forced = getenv("LH_TEST_FORCE_PRESSURE") ? atoi(getenv("LH_TEST_FORCE_PRESSURE")) : 0;

    analog_silence = vbr_old_prepare(S, pe, ms_ener_ratio, ratio, l3_xmin, frameBits, min_bits, max_bits);
    for (;;) {
        used_bits = 0;
        for (gr = 0; gr < cfg->mode_gr; gr++)
            for (ch = 0; ch < cfg->channels; ch++) {
                OrcGr  *const cod_info = &S->tt[gr][ch];
                int     ret = init_xrpow(S, cod_info, xrpow);
                if (ret == 0 || max_bits[gr][ch] == 0)
                    continue;   /* nothing to quantise */
                vbr_old_encode_granule(S, cod_info, l3_xmin[gr][ch], xrpow, ch, min_bits[gr][ch], max_bits[gr][ch]);
                /* substep_shaping & 1 (trancate_smallspectrums) is never set by lame_init_params */
                used_bits += cod_info->part2_3_length + cod_info->part2_length;
            }
        if (analog_silence && !cfg->enforce_min_bitrate)
            S->bitrate_index = 1;
        else
            S->bitrate_index = cfg->vbr_min_bitrate_index;
        for (; S->bitrate_index < cfg->vbr_max_bitrate_index; S->bitrate_index++)
            if (used_bits <= frameBits[S->bitrate_index])
                break;
        bits = ResvFrameBegin(S, &mean_bits);
        if (used_bits <= bits && forced-- <= 0)
            break;
        vbr_old_bitpressure(S, l3_xmin, min_bits, max_bits);
    }
    for (gr = 0; gr < cfg->mode_gr; gr++)
        for (ch = 0; ch < cfg->channels; ch++) {
            OrcGr  *const cod_info = &S->tt[gr][ch];
            /* iteration_finish_one, reference quantize.c:1213-1232 */
            best_scalefac_store(S, gr, ch);
            if (cfg->use_best_huffman == 1)
                best_huffman_divide(S, cod_info);
            S->ResvSize -= cod_info->part2_3_length + cod_info->part2_length;
        }
    ResvFrameEnd(S, mean_bits);
}
