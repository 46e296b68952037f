/*
 * orc_abr.c -- CPU restatement of the reference's ABR iteration loop.  TEST
 * INFRASTRUCTURE ONLY (see orc_common.h).  Follows reference libmp3lame/quantize.c:
 * calc_target_bits (:1768-1884) and ABR_iteration_loop (:1900-1972); the granule work is
 * the CBR loop's (outer_loop etc. in orc_quant.c).  Part of lame_oracle.c.
 */

/* bit budget of every granule/channel from the mean bitrate and the perceptual entropy */
static void
abr_target_bits(OrcStream * S, float pe[2][2], const float ms_ener_ratio[2], int targ_bits[2][2],
                int *analog_silence_bits, int *max_frame_bits)
{
    const LhConfig *cfg = S->cfg;
    float   res_factor;
    int     gr, ch, totbits, mean_bits;
    int const framesize = 576 * cfg->mode_gr;

    S->bitrate_index = cfg->vbr_max_bitrate_index;
    *max_frame_bits = ResvFrameBegin(S, &mean_bits);
    S->bitrate_index = 1;
    mean_bits = getframebits(S) - cfg->sideinfo_len * 8;
    *analog_silence_bits = mean_bits / (cfg->mode_gr * cfg->channels);

    mean_bits = cfg->vbr_avg_bitrate_kbps * framesize * 1000;
    if (S->substep_shaping & 1)
        mean_bits *= 1.09;
    mean_bits /= cfg->samplerate;
    mean_bits -= cfg->sideinfo_len * 8;
    mean_bits /= (cfg->mode_gr * cfg->channels);

    res_factor = .93 + .07 * (11.0 - cfg->compression_ratio) / (11.0 - 5.5);
    if (res_factor < .90)
        res_factor = .90;
    if (res_factor > 1.00)
        res_factor = 1.00;
    for (gr = 0; gr < S->cfg->mode_gr; gr++) {
        int     sum = 0;
        for (ch = 0; ch < cfg->channels; ch++) {
            targ_bits[gr][ch] = res_factor * mean_bits;
            if (pe[gr][ch] > 700) {
                int     add_bits = (pe[gr][ch] - 700) / 1.4;
                targ_bits[gr][ch] = res_factor * mean_bits;
                if (S->tt[gr][ch].block_type == LH_SHORT_TYPE) {
                    if (add_bits < mean_bits / 2)
                        add_bits = mean_bits / 2;
                }
                if (add_bits > mean_bits * 3 / 2)
                    add_bits = mean_bits * 3 / 2;
                else if (add_bits < 0)
                    add_bits = 0;
                targ_bits[gr][ch] += add_bits;
            }
            if (targ_bits[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                targ_bits[gr][ch] = LH_MAX_BITS_PER_CHANNEL;
            sum += targ_bits[gr][ch];
        }
        if (sum > LH_MAX_BITS_PER_GRANULE)
            for (ch = 0; ch < cfg->channels; ++ch) {
                targ_bits[gr][ch] *= LH_MAX_BITS_PER_GRANULE;
                targ_bits[gr][ch] /= sum;
            }
    }
    if (S->mode_ext == LH_MPG_MD_MS_LR)
        for (gr = 0; gr < S->cfg->mode_gr; gr++)
            reduce_side(targ_bits[gr], ms_ener_ratio[gr], mean_bits * cfg->channels, LH_MAX_BITS_PER_GRANULE);
    totbits = 0;
    for (gr = 0; gr < S->cfg->mode_gr; gr++)
        for (ch = 0; ch < cfg->channels; ch++) {
            if (targ_bits[gr][ch] > LH_MAX_BITS_PER_CHANNEL)
                targ_bits[gr][ch] = LH_MAX_BITS_PER_CHANNEL;
            totbits += targ_bits[gr][ch];
        }
    if (totbits > *max_frame_bits && totbits > 0)
        for (gr = 0; gr < S->cfg->mode_gr; gr++)
            for (ch = 0; ch < cfg->channels; ch++) {
                targ_bits[gr][ch] *= *max_frame_bits;
                targ_bits[gr][ch] /= totbits;
            }
}

void
orc_abr_iteration_loop(OrcStream * S, float pe[2][2], const float ms_ener_ratio[2], const OrcRatio ratio[2][2])
{
    const LhConfig *cfg = S->cfg;
    float   l3_xmin[LH_SFBMAX];
    float   xrpow[576];
    int     targ_bits[2][2];
    int     mean_bits = 0, max_frame_bits, analog_silence_bits, gr, ch, i;

    abr_target_bits(S, pe, ms_ener_ratio, targ_bits, &analog_silence_bits, &max_frame_bits);
    for (gr = 0; gr < S->cfg->mode_gr; gr++) {
        if (S->mode_ext == LH_MPG_MD_MS_LR)
            for (i = 0; i < 576; ++i) {         /* ms_convert, reference quantize.c:48-59 */
                float   l = S->tt[gr][0].xr[i];
                float   r = S->tt[gr][1].xr[i];
                S->tt[gr][0].xr[i] = (l + r) * (float) (ORC_SQRT2 * 0.5);
                S->tt[gr][1].xr[i] = (l - r) * (float) (ORC_SQRT2 * 0.5);
            }
        for (ch = 0; ch < cfg->channels; ch++) {
            OrcGr  *cod_info = &S->tt[gr][ch];
            S->masking_lower = (cod_info->block_type != LH_SHORT_TYPE) ? cfg->masking_lower_long
                : cfg->masking_lower_short;
            init_outer_loop(S, cod_info);
            if (init_xrpow(S, cod_info, xrpow)) {
                int const ath_over = calc_xmin(S, &ratio[gr][ch], cod_info, l3_xmin);
                if (0 == ath_over)      /* analog silence */
                    targ_bits[gr][ch] = analog_silence_bits;
                (void) outer_loop(S, cod_info, l3_xmin, xrpow, ch, targ_bits[gr][ch]);
            }
            best_scalefac_store(S, gr, ch);     /* iteration_finish_one, reference quantize.c:1213-1232 */
            if (cfg->use_best_huffman == 1)
                best_huffman_divide(S, cod_info);
            S->ResvSize -= cod_info->part2_3_length + cod_info->part2_length;
        }
    }
    /* the smallest frame that brings the reservoir back to a non-negative size */
    for (S->bitrate_index = cfg->vbr_min_bitrate_index; S->bitrate_index <= cfg->vbr_max_bitrate_index;
         S->bitrate_index++)
        if (ResvFrameBegin(S, &mean_bits) >= 0)
            break;
    ResvFrameEnd(S, mean_bits);
}
