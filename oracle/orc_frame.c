/*
 * orc_frame.c -- CPU restatement of the per-frame driver and of the stream
 * framing (reference libmp3lame/encoder.c:56-137,189-574, lame.c:1671-1775,
 * 2041-2120).  TEST INFRASTRUCTURE ONLY.
 */
#include <stdlib.h>
#include "orc_common.h"

/* reference psymodel.c:1897-1922, 2075-2076; lame.c:962-963, 2285-2302 */
void
orc_stream_init(OrcStream * S, const LhConfig * cfg, const LhTables * tab)
{
    int     i, j, sb;
    memset(S, 0, sizeof(*S));
    S->cfg = cfg;
    S->tab = tab;
    for (i = 0; i < 4; ++i) {
        for (j = 0; j < LH_CBANDS; ++j) {
            S->nb_l1[i][j] = 1e20;
            S->nb_l2[i][j] = 1e20;
            S->nb_s1[i][j] = S->nb_s2[i][j] = 1.0;
        }
        for (sb = 0; sb < LH_SBMAX_L; sb++) {
            S->en[i].l[sb] = 1e20;
            S->thm[i].l[sb] = 1e20;
        }
        for (j = 0; j < 3; ++j) {
            for (sb = 0; sb < LH_SBMAX_S; sb++) {
                S->en[i].s[sb][j] = 1e20;
                S->thm[i].s[sb][j] = 1e20;
            }
            S->last_attacks[i] = 0;
        }
        for (j = 0; j < 9; j++)
            S->last_en_subshort[i][j] = 10.;
    }
    S->loudness_sq_save[0] = S->loudness_sq_save[1] = 0.0;
    S->blocktype_old[0] = S->blocktype_old[1] = LH_NORM_TYPE;
    S->ath_adjust_factor = 0.01;
    S->ath_adjust_limit = 1.0;
    for (i = 0; i < 19; i++)
        S->pefirbuf[i] = 700 * cfg->mode_gr * cfg->channels;
    S->slot_lag = cfg->frac_SpF;
    S->OldValue[0] = S->OldValue[1] = 180;
    S->CurrentStep[0] = S->CurrentStep[1] = 4;
    S->masking_lower = 1;
    S->substep_shaping = cfg->substep_shaping;
    S->bitrate_index = cfg->bitrate_index;
}

/* reference encoder.c:56-137 */
static void
adjust_ATH(OrcStream * S)
{
    const LhTables *T = S->tab;
    float   gr2_max, max_pow;
    if (T->ath_use_adjust == 0) {
        S->ath_adjust_factor = 1.0;
        return;
    }
    max_pow = S->loudness_sq[0][0];
    gr2_max = S->loudness_sq[1][0];
    if (S->cfg->channels == 2) {
        max_pow += S->loudness_sq[0][1];
        gr2_max += S->loudness_sq[1][1];
    }
    else {
        max_pow += max_pow;
        gr2_max += gr2_max;
    }
    if (S->cfg->mode_gr == 2)
        max_pow = (max_pow > gr2_max) ? max_pow : gr2_max;
    max_pow *= 0.5;
    max_pow *= T->aa_sensitivity_p;
    if (max_pow > 0.03125) {
        if (S->ath_adjust_factor >= 1.0)
            S->ath_adjust_factor = 1.0;
        else if (S->ath_adjust_factor < S->ath_adjust_limit)
            S->ath_adjust_factor = S->ath_adjust_limit;
        S->ath_adjust_limit = 1.0;
    }
    else {
        float const adj_lim_new = 31.98 * max_pow + 0.000625;
        if (S->ath_adjust_factor >= adj_lim_new) {
            S->ath_adjust_factor *= adj_lim_new * 0.075 + 0.925;
            if (S->ath_adjust_factor < adj_lim_new)
                S->ath_adjust_factor = adj_lim_new;
        }
        else {
            if (S->ath_adjust_limit >= adj_lim_new)
                S->ath_adjust_factor = adj_lim_new;
            else if (S->ath_adjust_factor < S->ath_adjust_limit)
                S->ath_adjust_factor = S->ath_adjust_limit;
        }
        S->ath_adjust_limit = adj_lim_new;
    }
}

static void
pack_frame(OrcStream * S, LhFrameOut * fo, int mdb_for_header)
{
    int     gr, ch, i;
    int const nch = S->cfg->channels;
    memset(fo, 0, sizeof(*fo));
    for (gr = 0; gr < S->cfg->mode_gr; gr++) {
        for (ch = 0; ch < nch; ch++) {
            OrcGr const *gi = &S->tt[gr][ch];
            LhGranule *g = &fo->gr[gr][ch];
            for (i = 0; i < 576; i++) {
                int     v = gi->l3_enc[i];
                if (v != 0 && gi->xr[i] < 0.0f)
                    v = -v;
                g->l3_enc[i] = (int16_t) v;
            }
            for (i = 0; i < LH_SFBMAX; i++)
                g->scalefac[i] = (int8_t) gi->scalefac[i];
            g->part2_3_length = (int16_t) gi->part2_3_length;
            g->part2_length = (int16_t) gi->part2_length;
            g->big_values = (int16_t) gi->big_values;
            g->count1 = (int16_t) gi->count1;
            g->global_gain = (int16_t) gi->global_gain;
            g->scalefac_compress = (int16_t) gi->scalefac_compress;
            g->block_type = (int8_t) gi->block_type;
            g->mixed_block_flag = (int8_t) gi->mixed_block_flag;
            for (i = 0; i < 3; i++) {
                g->table_select[i] = (int8_t) gi->table_select[i];
                g->subblock_gain[i] = (int8_t) gi->subblock_gain[i];
            }
            g->region0_count = (int8_t) gi->region0_count;
            g->region1_count = (int8_t) gi->region1_count;
            g->preflag = (int8_t) gi->preflag;
            g->scalefac_scale = (int8_t) gi->scalefac_scale;
            g->count1table_select = (int8_t) gi->count1table_select;
            g->sfbmax = (int8_t) gi->sfbmax;
            g->sfbdivide = (int8_t) gi->sfbdivide;
            g->count1bits = (int16_t) gi->count1bits;
        }
    }
    for (ch = 0; ch < nch; ch++)
        for (i = 0; i < 4; i++)
            fo->scfsi[ch][i] = (int8_t) S->scfsi[ch][i];
    (void) mdb_for_header;
    fo->main_data_begin = (int16_t) S->main_data_begin; /* value for the NEXT frame, like the harness */
    fo->resvDrain_pre = (int16_t) S->resvDrain_pre;
    fo->resvDrain_post = (int16_t) S->resvDrain_post;
    fo->bitrate_index = (int8_t) S->bitrate_index;
    fo->padding = (int8_t) S->padding;
    fo->mode_ext = (int8_t) S->mode_ext;
    fo->resv_size = S->ResvSize;
}

/* reference encoder.c:305-574 (without the bit packing, which stays on the host) */
int
orc_encode_frame(OrcStream * S, const float *inbuf_l, const float *inbuf_r, LhFrameOut * out)
{
    const LhConfig *cfg = S->cfg;
    OrcRatio masking_LR[2][2];
    OrcRatio masking_MS[2][2];
    const OrcRatio (*masking)[2];
    const float *inbuf[2];
    float   tot_ener[2][4];
    float   ms_ener_ratio[2] = { .5, .5 };
    float   pe[2][2] = { {0., 0.}, {0., 0.} }, pe_MS[2][2] = { {0., 0.}, {0., 0.} };
    float   (*pe_use)[2];
    int     ch, gr, mdb_header;
    int const nch = cfg->channels;

    inbuf[0] = inbuf_l;
    inbuf[1] = inbuf_r;
    if (S->frame_init_done == 0) {
        /* lame_encode_frame_init, reference encoder.c:189-236: prime the
         * polyphase/MDCT overlap with one pass over a zero-prefixed buffer */
        static float primebuff0[286 + 1152 + 576];
        static float primebuff1[286 + 1152 + 576];
        int     i, j;
        S->frame_init_done = 1;
        memset(primebuff0, 0, sizeof(primebuff0));
        memset(primebuff1, 0, sizeof(primebuff1));
        for (i = 0, j = 0; i < 286 + 576 * (1 + cfg->mode_gr); ++i) {
            if (i >= 576 * cfg->mode_gr) {
                primebuff0[i] = inbuf[0][j];
                if (nch == 2)
                    primebuff1[i] = inbuf[1][j];
                ++j;
            }
        }
        for (gr = 0; gr < cfg->mode_gr; gr++)
            for (ch = 0; ch < nch; ch++)
                S->tt[gr][ch].block_type = LH_SHORT_TYPE;
        orc_mdct_sub48(S, primebuff0, primebuff1);
    }
    S->padding = 0;
    if ((S->slot_lag -= cfg->frac_SpF) < 0) {
        S->slot_lag += cfg->samplerate;
        S->padding = 1;
    }
    {
        const float *bufp[2] = { 0, 0 };
        int     blocktype[2];
        for (gr = 0; gr < cfg->mode_gr; gr++) {
            for (ch = 0; ch < nch; ch++)
                bufp[ch] = &inbuf[ch][576 + gr * 576 - LH_FFTOFFSET];
            orc_psycho_anal(S, bufp, gr, masking_LR, masking_MS, pe[gr], pe_MS[gr], tot_ener[gr],
                            blocktype);
            if (cfg->mode == LH_MODE_JOINT_STEREO) {
                ms_ener_ratio[gr] = tot_ener[gr][2] + tot_ener[gr][3];
                if (ms_ener_ratio[gr] > 0)
                    ms_ener_ratio[gr] = tot_ener[gr][3] / ms_ener_ratio[gr];
            }
            for (ch = 0; ch < nch; ch++) {
                S->tt[gr][ch].block_type = blocktype[ch];
                S->tt[gr][ch].mixed_block_flag = 0;
            }
        }
    }
    adjust_ATH(S);
    orc_mdct_sub48(S, inbuf[0], inbuf[1]);

    S->mode_ext = LH_MPG_MD_LR_LR;
    if (cfg->force_ms)
        S->mode_ext = LH_MPG_MD_MS_LR;
    else if (cfg->mode == LH_MODE_JOINT_STEREO) {
        float   sum_pe_MS = 0;
        float   sum_pe_LR = 0;
        for (gr = 0; gr < cfg->mode_gr; gr++) {
            for (ch = 0; ch < nch; ch++) {
                sum_pe_MS += pe_MS[gr][ch];
                sum_pe_LR += pe[gr][ch];
            }
        }
        if (sum_pe_MS <= 1.00 * sum_pe_LR) {
            OrcGr const *const gi0 = &S->tt[0][0];
            OrcGr const *const gi1 = &S->tt[cfg->mode_gr - 1][0];
            if (gi0[0].block_type == gi0[1].block_type && gi1[0].block_type == gi1[1].block_type)
                S->mode_ext = LH_MPG_MD_MS_LR;
        }
    }
    if (S->mode_ext == LH_MPG_MD_MS_LR) {
        masking = (const OrcRatio (*)[2]) masking_MS;
        pe_use = pe_MS;
    }
    else {
        masking = (const OrcRatio (*)[2]) masking_LR;
        pe_use = pe;
    }
    {
        static float const fircoef[9] = {
            -0.0207887 * 5, -0.0378413 * 5, -0.0432472 * 5, -0.031183 * 5,
            7.79609e-18 * 5, 0.0467745 * 5, 0.10091 * 5, 0.151365 * 5,
            0.187098 * 5
        };
        int     i;
        float   f;
        for (i = 0; i < 18; i++)
            S->pefirbuf[i] = S->pefirbuf[i + 1];
        f = 0.0;
        for (gr = 0; gr < cfg->mode_gr; gr++)
            for (ch = 0; ch < nch; ch++)
                f += pe_use[gr][ch];
        S->pefirbuf[18] = f;
        f = S->pefirbuf[9];
        for (i = 0; i < 9; i++)
            f += (S->pefirbuf[i] + S->pefirbuf[18 - i]) * fircoef[i];
        f = (670 * 5 * cfg->mode_gr * nch) / f;
        for (gr = 0; gr < cfg->mode_gr; gr++)
            for (ch = 0; ch < nch; ch++)
                pe_use[gr][ch] *= f;
    }
    mdb_header = S->main_data_begin;
    if (cfg->vbr == 3)
        orc_abr_iteration_loop(S, pe_use, ms_ener_ratio, masking);
    else if (cfg->vbr == 2)
        orc_vbr_old_iteration_loop(S, pe_use, ms_ener_ratio, masking);
    else if (cfg->vbr)
        orc_vbr_new_iteration_loop(S, pe_use, masking);
    else
        orc_cbr_iteration_loop(S, pe_use, ms_ener_ratio, masking);
    mdb_header = S->main_data_begin;    /* after ResvFrameEnd: the value the header carries */
    {
        /* main_data_begin bookkeeping of format_bitstream, reference bitstream.c:917-935 */
        int     bits = 8 * cfg->sideinfo_len, frame_bits;
        int     bit_rate = (cfg->version ? lh_bitrate_mpeg1 : lh_bitrate_mpeg2)[S->bitrate_index];
        for (gr = 0; gr < cfg->mode_gr; gr++)
            for (ch = 0; ch < nch; ch++)
                bits += S->tt[gr][ch].part2_3_length + S->tt[gr][ch].part2_length;
        bits += S->resvDrain_post;
        frame_bits = 8 * ((cfg->version + 1) * 72000 * bit_rate / cfg->samplerate + S->padding);
        S->main_data_begin += (frame_bits - bits) / 8;
        if (out) {
            pack_frame(S, out, mdb_header);
            out->frame_bits = frame_bits;
        }
    }
    ++S->frame_number;
    return 0;
}

/* number of frames lame_encode_buffer + lame_encode_flush produce for n samples
 * (reference lame.c:1671-1775, 2041-2120) */
/* (fs = samples per frame: 1152, or 576 for MPEG-2 / 2.5; the window a frame needs is BLKSIZE + fs - FFTOFFSET,
 * reference lame.c:1627-1648) */
int
orc_total_frames_fs(long n, int fs)
{
    long    mf_size = LH_MF_START, to_encode = LH_ENCDELAY + LH_POSTDELAY;
    long    frames = 0, fed = 0;
    int     end_padding, frames_left;
    int const needed = LH_BLKSIZE + fs - LH_FFTOFFSET;
    to_encode += n;
    while (fed < n) {
        long    m = n - fed;
        if (m > fs)
            m = fs;
        /* fill_buffer copies min(framesize, remaining) samples per inner iteration */
        mf_size += m;
        fed += m;
        if (mf_size >= needed) {
            frames++;
            mf_size -= fs;
            to_encode -= fs;
        }
    }
    to_encode -= LH_POSTDELAY;
    end_padding = fs - (int) (to_encode % fs);
    if (end_padding < 576)
        end_padding += fs;
    frames_left = (int) ((to_encode + end_padding) / fs);
    return (int) (frames + frames_left);
}

int
orc_total_frames(long n)
{
    return orc_total_frames_fs(n, 1152);
}

/* Encode a whole planar s16 stream; window of frame f is pcm[1152 f - 528 ...],
 * zero outside [0, n).  Returns the number of frames. */
int
orc_encode_stream(const LhConfig * cfg, const LhTables * tab, const short *l, const short *r,
                  long n, LhFrameOut * frames, int max_frames, float *xr_out)
{
    OrcStream *S = (OrcStream *) malloc(sizeof(OrcStream));
    int const fs = 576 * cfg->mode_gr;
    int     nf = orc_total_frames_fs(n, fs), f, i;
    static float mf[2][LH_MF_NEEDED];
    orc_stream_init(S, cfg, tab);
    for (f = 0; f < nf; f++) {
        LhFrameOut tmp;
        long    base = (long) fs * f - LH_MF_START;
        for (i = 0; i < LH_MF_NEEDED; i++) {
            long    p = base + i;
            if (p >= 0 && p < n) {
                if (cfg->pcm_mix != 0) {        /* downmix: u = xl * m00 + xr * m01 (reference lame.c:1802-1834) */
                    mf[0][i] = (float) l[p] * cfg->pcm_scale + (float) r[p] * cfg->pcm_mix;
                    mf[1][i] = 0;
                }
                else {
                    mf[0][i] = (float) l[p] * cfg->pcm_scale;
                    mf[1][i] = (float) r[p] * cfg->pcm_scale_r;
                }
            }
            else {
                mf[0][i] = 0;
                mf[1][i] = 0;
            }
        }
        orc_encode_frame(S, mf[0], mf[1], &tmp);
        if (f < max_frames) {
            if (frames)
                frames[f] = tmp;
            if (xr_out) {
                int     gr, ch;
                for (gr = 0; gr < cfg->mode_gr; gr++)
                    for (ch = 0; ch < 2; ch++)
                        memcpy(xr_out + ((f * 2 + gr) * 2 + ch) * 576, S->tt[gr][ch].xr,
                               576 * sizeof(float));
            }
        }
    }
    free(S);
    return nf;
}

int
orc_sizeof_stream(void)
{
    return (int) sizeof(OrcStream);
}

/* same flat layout as refh_get_state() in ref_harness.c */
void
orc_get_state(OrcStream * S, float *nb_l1, float *nb_l2, float *en, float *thm, float *misc)
{
    int     i = 0;
    memcpy(nb_l1, S->nb_l1, sizeof(S->nb_l1));
    memcpy(nb_l2, S->nb_l2, sizeof(S->nb_l2));
    memcpy(en, S->en, sizeof(S->en));
    memcpy(thm, S->thm, sizeof(S->thm));
    misc[i++] = S->ath_adjust_factor;
    misc[i++] = S->ath_adjust_limit;
    misc[i++] = S->loudness_sq_save[0];
    misc[i++] = S->loudness_sq_save[1];
    misc[i++] = S->tot_ener[0];
    misc[i++] = S->tot_ener[1];
    misc[i++] = S->tot_ener[2];
    misc[i++] = S->tot_ener[3];
    misc[i++] = (float) S->last_attacks[0];
    misc[i++] = (float) S->last_attacks[1];
    misc[i++] = (float) S->last_attacks[2];
    misc[i++] = (float) S->last_attacks[3];
    misc[i++] = (float) S->blocktype_old[0];
    misc[i++] = (float) S->blocktype_old[1];
    misc[i++] = (float) S->ResvSize;
    misc[i++] = (float) S->slot_lag;
    misc[i++] = S->masking_lower;
    misc[i++] = (float) S->OldValue[0];
    misc[i++] = (float) S->OldValue[1];
    misc[i++] = (float) S->CurrentStep[0];
    misc[i++] = (float) S->CurrentStep[1];
    misc[i++] = S->pefirbuf[18];
}
