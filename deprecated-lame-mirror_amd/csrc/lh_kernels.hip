/*
 * lh_kernels.hip -- the batched MP3 encode inner loop as one persistent HIP
 * kernel for gfx950 (MI355X).
 *
 * Decomposition (DESIGN.md section 3): one workgroup of two wavefronts per
 * stream; wave w owns channel w (and the mid/side pseudo-channel w+2 in the
 * psycho-acoustic model).  A workgroup walks its stream's frames in order --
 * the bit reservoir, the block-type state machine and the pre-echo history
 * chain frames together -- and streams are independent, so a batch of B streams
 * is B workgroups with no inter-workgroup communication.  All per-frame working
 * data (spectra, quantised lines, FFT buffers) stays in LDS; HBM traffic is the
 * PCM read, the per-stream carried state and the side-info payload
 * (LhFrameOut) that the host bit packer consumes.
 *
 * Per frame (reference lame_encode_mp3_frame, encoder.c:305-574):
 *   psycho-acoustics x2 granules -> ATH adjust -> polyphase+MDCT -> M/S decision
 *   -> PE smoothing -> CBR iteration loop x2 granules -> payload.
 */
#include <stdint.h>
#include <math.h>

#ifdef LH_EMU
#include "hipemu.h"
#define LH_CONST static const
#else
#include <hip/hip_runtime.h>
#define LH_CONST __device__ static const
#endif

#ifdef LH_EMU
#include <string.h>
extern "C" { int lh_emu_poison_lds = 0; }
#endif
#include "lh_static_tables.h"
#include "lh_dev_common.h"
#include "lh_dev_psy.h"
#include "lh_dev_mdct.h"
#include "lh_dev_quant.h"
#include "lh_dev_qloop.h"

/* reference encoder.c:56-137, wave-uniform */
LH_DEVFN void
lh_adjust_ATH(const LhTables * T, const float loud[2][2], float *factor, float *limit)
{
    float   gr2_max, max_pow;
    float   f = *factor, lim = *limit;
    if (T->ath_use_adjust == 0) {
        *factor = 1.0;
        return;
    }
    max_pow = loud[0][0];
    gr2_max = loud[1][0];
    max_pow += loud[0][1];
    gr2_max += loud[1][1];
    max_pow = (max_pow > gr2_max) ? max_pow : gr2_max;
    max_pow = (float) (max_pow * 0.5);
    max_pow *= T->aa_sensitivity_p;
    if (max_pow > 0.03125) {
        if (f >= 1.0)
            f = 1.0;
        else if (f < lim)
            f = lim;
        lim = 1.0;
    }
    else {
        float const adj_lim_new = (float) (31.98 * max_pow + 0.000625);
        if (f >= adj_lim_new) {
            f = (float) (f * (adj_lim_new * 0.075 + 0.925));
            if (f < adj_lim_new)
                f = adj_lim_new;
        }
        else {
            if (lim >= adj_lim_new)
                f = adj_lim_new;
            else if (f < lim)
                f = lim;
        }
        lim = adj_lim_new;
    }
    *factor = f;
    *limit = lim;
}

LH_DEVCONST float lh_pe_fir[9] = {
    (float) (-0.0207887 * 5), (float) (-0.0378413 * 5), (float) (-0.0432472 * 5),
    (float) (-0.031183 * 5), (float) (7.79609e-18 * 5), (float) (0.0467745 * 5),
    (float) (0.10091 * 5), (float) (0.151365 * 5), (float) (0.187098 * 5)
};

/* stage the quantiser lookup tables into LDS (whole workgroup) */
LH_DEVFN void
lh_load_qtabs(const LhCtx & c, LhQTabs & q)
{
    for (int i = c.tid; i < 256; i += LH_NT) {
        q.largetbl[i] = lh_largetbl[i];
        q.pow43h[i] = c.T->pow43[i];
        q.adj43h[i] = c.T->adj43asm[i];
    }
    for (int i = c.tid; i < (int) sizeof(q.ht_len); i += LH_NT)
        q.ht_len[i] = lh_ht_len[i];
    for (int i = c.tid; i < 288; i += LH_NT) {
        /* big_values = 2 i + 2: the region split the reference looks up with bv_scf[bv - 2], [bv - 1] */
        int const bv = 2 * i + 2;
        int const r0 = c.T->bv_scf[bv - 2], r1 = c.T->bv_scf[bv - 1];
        int const a1 = c.T->sfb_l[r0 + 1], a2 = c.T->sfb_l[(r0 + r1 + 2 < LH_SBMAX_L) ? r0 + r1 + 2 : LH_SBMAX_L];
        q.bvpack[i] = (uint32_t) r0 | ((uint32_t) r1 << 4) | ((uint32_t) a1 << 8) | ((uint32_t) a2 << 18);
    }
    if (c.tid == 0) {
        q.sfb_s3 = (uint16_t) c.T->sfb_s[3];
        q.pad = 0;
    }
    if (c.tid < 17) {
        LhRegionLut const r = lh_region_lut((unsigned) c.tid);
        q.lut_pa[c.tid] = r.pa;
        q.lut_pb[c.tid] = r.pb;
    }
    if (c.tid < 32)
        lq_class_tabs(c.tid, &q.ctabA[c.tid], &q.ctabB[c.tid]);
    if (c.tid < 9)
        q.table23[c.tid] = lh_table23[c.tid];
    if (c.tid < 16) {
        q.table56[c.tid] = lh_table56[c.tid];
        q.t32l[c.tid] = lh_t32l[c.tid];
        q.t33l[c.tid] = lh_t33l[c.tid];
        q.t3233[c.tid] = ((uint32_t) lh_t32l[c.tid] << 16) | lh_t33l[c.tid];
    }
    if (c.tid < 24) {
        q.sfb_l[c.tid] = (uint16_t) ((c.tid < 23) ? c.T->sfb_l[c.tid] : 576);
        q.pretab[c.tid] = (c.tid < 22) ? lh_pretab[c.tid] : 0;
    }
}

/* write one granule of one channel to the payload; one wave */
LH_DEVFN void
lh_store_granule(const LhCtx & c, const LhChanLds & Q, const LhQR & R, const LhGrR & g,
                 const float *xr, LhGranule * o)
{
    const int16_t *ix = Q.ix[0];
    for (int i = c.lane; i < 576; i += 64) {
        int     v = ix[i];
        if (v != 0 && xr[i] < 0.0f)
            v = -v;
        o->l3_enc[i] = (int16_t) v;
    }
    if (c.lane < LH_SFBMAX)
        o->scalefac[c.lane] = (int8_t) Q.sf[0][c.lane];
    if (c.lane == 0) {
        o->pad0 = 0;
        o->part2_3_length = (int16_t) g.part2_3_length;
        o->part2_length = (int16_t) g.part2_length;
        o->big_values = (int16_t) g.big_values;
        o->count1 = (int16_t) g.count1;
        o->global_gain = (int16_t) g.global_gain;
        o->scalefac_compress = (int16_t) g.scalefac_compress;
        o->block_type = (int8_t) R.block_type;
        o->mixed_block_flag = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            o->table_select[i] = (int8_t) g.table_select[i];
            o->subblock_gain[i] = (int8_t) g.subblock_gain[i];
        }
        o->region0_count = (int8_t) g.region0_count;
        o->region1_count = (int8_t) g.region1_count;
        o->preflag = (int8_t) g.preflag;
        o->scalefac_scale = (int8_t) g.scalefac_scale;
        o->count1table_select = (int8_t) g.count1table_select;
        o->sfbmax = (int8_t) R.sfbmax;
        o->sfbdivide = (int8_t) R.sfbdivide;
        o->pad1 = 0;
        o->count1bits = (int16_t) g.count1bits;
        o->pad2 = 0;
    }
}

#include "lh_dev_emit.h"
#include "lh_dev_vbr.h"

/* one frame of one stream; executed by the whole workgroup */
LH_DEVFN void
lh_encode_frame(LhCtx & c, LhFrameOut * fo)
{
    LhLds & L = lh_lds;
    const LhConfig *cfg = c.cfg;
    const LhTables *T = c.T;
    LhStreamState *st = c.st;
    int const w = c.wave, lane = c.lane, tid = c.tid;
    /* mono: wave 1 has no channel; it runs the transforms on the duplicated PCM (never read) and
     * only keeps pace through the barriers of the quantisation stage */
    int const nch = cfg->channels;

    /* ---- polyphase priming on the first frame (reference encoder.c:189-236) ---- */
#if defined(LH_PROF) && !defined(LH_EMU)
    if (lane < LH_NPROF)
        L.prof[w][lane] = 0;
#endif
    LH_PT(t_frame);
    if (!st->primed) {
        lh_stage_window(c, L.mf, c.frame_base - 1152);
        LH_SYNC_WG();
        lh_polyphase(w);
        for (int i = lane; i < 576; i += 64)
            st->sb_prev[w][i] = L.u.mdct.sb[w][2][i];
        LH_SYNC_WG();
        if (tid == 0)
            st->primed = 1;
    }
    LH_SYNC_WG();

    lh_stage_window(c, L.mf, c.frame_base);
    LH_SYNC_WG();

    /* ---- padding (reference encoder.c:348-352) ---- */
    int     padding = 0;
    int     slot_lag = st->slot_lag - cfg->frac_SpF;
    if (slot_lag < 0) {
        slot_lag += cfg->samplerate;
        padding = 1;
    }


    LH_PA(24, t_frame);
    /* ---- stage 1: psycho-acoustic model, two granules ---- */
    LH_PT(t_psy);
    for (int gr = 0; gr < 2; gr++)
        lh_psy_granule(gr);
    LH_PA(1, t_psy);

    float   ms_ener_ratio[2] = { .5f, .5f };
    if (cfg->mode == LH_MODE_JOINT_STEREO) {
        for (int gr = 0; gr < 2; gr++) {
            float   r = L.tot_ener[gr][2] + L.tot_ener[gr][3];
            if (r > 0)
                r = L.tot_ener[gr][3] / r;
            ms_ener_ratio[gr] = r;
        }
    }

    /* ---- ATH auto adjustment (reference encoder.c:397) ---- */
    {
        float   factor = st->ath_adjust_factor, limit = st->ath_adjust_limit;
        float   loud[2][2];
        loud[0][0] = L.loudness_sq[0][0];
        loud[1][0] = L.loudness_sq[1][0];
        /* one channel counts twice (reference encoder.c:72-79) */
        loud[0][1] = (cfg->channels == 2) ? L.loudness_sq[0][1] : loud[0][0];
        loud[1][1] = (cfg->channels == 2) ? L.loudness_sq[1][1] : loud[1][0];
        lh_adjust_ATH(T, loud, &factor, &limit);
        LH_SYNC_WG();
        if (tid == 0) {
            st->ath_adjust_factor = factor;
            st->ath_adjust_limit = limit;
        }
    }
    LH_SYNC_WG();

    LH_PA(25, t_frame);
    /* ---- stage 2: polyphase + MDCT (reference encoder.c:405) ---- */
    LH_PT(t_mdct);
    for (int i = lane; i < 576; i += 64)
        L.u.mdct.sb[w][0][i] = st->sb_prev[w][i];
    lh_polyphase(w);
    LH_SYNC_WG();               /* last read of mf (both channels) before xr overwrites it */
    lh_mdct_granules(w);
    for (int i = lane; i < 576; i += 64)
        st->sb_prev[w][i] = L.u.mdct.sb[w][2][i];
    LH_SYNC_WG();
    LH_PA(2, t_mdct);
    lh_load_qtabs(c, L.qt);     /* mf is dead; xr stays */
    if (cfg->vbr == 1 || cfg->vbr == 4) {
        /* step tables of the VBR scalefactor search (over the unused second quantised image) */
        for (int i = tid; i < 256; i += LH_NT) {
            LH_VBR_IPOW20[i] = T->ipow20[i];
            LH_VBR_POW20[i] = T->pow20[i + LH_QMAX2];
        }
    }
    LH_SYNC_WG();

    /* ---- stage 3: M/S decision (reference encoder.c:413-461) ---- */
    int     mode_ext = LH_MPG_MD_LR_LR;
    if (cfg->force_ms)
        mode_ext = LH_MPG_MD_MS_LR;
    else if (cfg->mode == LH_MODE_JOINT_STEREO) {
        float   sum_pe_MS = 0, sum_pe_LR = 0;
        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < 2; ch++) {
                sum_pe_MS += L.pe[gr][2 + ch];
                sum_pe_LR += L.pe[gr][ch];
            }
        if (sum_pe_MS <= 1.00 * sum_pe_LR) {
            if (L.block_type[0][0] == L.block_type[0][1] && L.block_type[1][0] == L.block_type[1][1])
                mode_ext = LH_MPG_MD_MS_LR;
        }
    }
    int const msoff = (mode_ext == LH_MPG_MD_MS_LR) ? 2 : 0;

    /* ---- PE smoothing FIR (reference encoder.c:489-518) ---- */
    float   pe_use[2][2];
    {
        float   buf[19], f;
        for (int i = 0; i < 18; i++)
            buf[i] = st->pefirbuf[i + 1];
        f = 0.0;
        for (int gr = 0; gr < 2; gr++) {
            pe_use[gr][1] = 0.0f;
            for (int ch = 0; ch < nch; ch++) {
                pe_use[gr][ch] = L.pe[gr][msoff + ch];
                f += pe_use[gr][ch];
            }
        }
        buf[18] = f;
        f = buf[9];
        for (int i = 0; i < 9; i++)
            f += (buf[i] + buf[18 - i]) * lh_pe_fir[i];
        f = (670 * 5 * 2 * nch) / f;
        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < nch; ch++)
                pe_use[gr][ch] *= f;
        LH_SYNC_WG();
        if (tid < 19)
            st->pefirbuf[tid] = buf[tid];
    }

    LH_PA(26, t_frame);
    /* ---- stage 4: CBR iteration loop (reference quantize.c:1988-2050) ---- */
    int     ResvSize = st->ResvSize, ResvMax, mdb = st->main_data_begin;
    int     substep = st->substep_shaping;
    int     bitrate_index = cfg->bitrate_index;
    int     frame_bits = lh_frame_bits(cfg, bitrate_index, padding);
    int     mean_bits = (frame_bits - cfg->sideinfo_len * 8) / cfg->mode_gr;
    int     total_bits = 0;
    int const vbr_new = (cfg->vbr == 1 || cfg->vbr == 4), abr = (cfg->vbr == 3);
    int     abr_targ[2][2] = { {0, 0}, {0, 0} }, analog_silence_bits = 0;
    if (vbr_new) {
        LH_SYNC_WG();
        if (tid == 0)
            for (int gr = 0; gr < 2; gr++)
                for (int ch = 0; ch < 2; ch++)
                    L.pe_use[gr][ch] = pe_use[gr][ch];
        LH_SYNC_WG();
        lh_vbr_frame(fo, mode_ext, msoff);
        bitrate_index = lh_uni_i(L.frame_bits);
        total_bits = lh_uni_i(L.max_bits);
        ResvSize = lh_uni_i(L.mean_bits);
        substep = lh_uni_i(L.targ_bits[0]);
        frame_bits = lh_frame_bits(cfg, bitrate_index, 0);
        mean_bits = (frame_bits - cfg->sideinfo_len * 8) / cfg->mode_gr;
    }
    if (abr) {
        int const bt[2][2] = { {L.block_type[0][0], L.block_type[0][1]}, {L.block_type[1][0], L.block_type[1][1]} };
        lh_abr_target_bits(cfg, ResvSize, substep, pe_use, ms_ener_ratio, bt, mode_ext, abr_targ, &analog_silence_bits);
    }
    {
        /* ResvFrameBegin */
        int const resvLimit = (8 * 256) * cfg->mode_gr - 8;
        ResvMax = cfg->buffer_constraint - frame_bits;
        if (ResvMax > resvLimit)
            ResvMax = resvLimit;
        if (ResvMax < 0 || cfg->disable_reservoir)
            ResvMax = 0;
    }
    for (int gr = 0; gr < 2 && !vbr_new; gr++) {
        int     targ_bits[2] = { abr_targ[gr][0], abr_targ[gr][1] };
        int     max_bits = 0;
        if (!abr)
            max_bits = lh_on_pe(cfg, ResvSize, ResvMax, &substep, pe_use[gr], targ_bits, mean_bits, gr);
        LH_SYNC_WG();
        if (mode_ext == LH_MPG_MD_MS_LR) {
            float const k = (float) (LH_SQRT2 * 0.5);
            for (int i = tid; i < 576; i += LH_NT) {
                float const l = L.xr[0][gr][i];
                float const r = L.xr[1][gr][i];
                L.xr[0][gr][i] = (l + r) * k;
                L.xr[1][gr][i] = (l - r) * k;
            }
            if (!abr)
                lh_reduce_side(targ_bits, ms_ener_ratio[gr], mean_bits, max_bits);
        }
        LH_SYNC_WG();
        if (w >= nch) {
            /* no second channel: its payload slot is all zero */
            uint32_t *z = (uint32_t *) &fo->gr[gr][w];
            for (int i = lane; i < (int) (sizeof(LhGranule) / 4); i += 64)
                z[i] = 0u;
            if (lane == 0)
                L.bits_used[w] = 0;
        }
        else {
            int const ch = w;
            LhChanLds & Q = L.u.quant.ch[ch];
            LhQR    R;
            LhGrR   g;
            float  *xr = L.xr[ch][gr];
            LhGranule *o = &fo->gr[gr][ch];
            LH_PT(t_q);
            /* R and g never have their address taken (they stay in scalar registers through
             * the inlined outer loop); the out-of-line stages exchange them through the wave's
             * LDS slot */
            lh_init_outer_loop(ch, gr, L.block_type[gr][ch], substep);
            R = lh_uniform(L.rg[ch].R);
            g = lh_uniform(L.rg[ch].g);
            if (lh_init_xrpow(c, Q, R, g, xr)) {
                lh_rg_put(c, R, g);
                lh_calc_xmin(ch, gr, msoff + ch);
                R = lh_uniform(L.rg[ch].R);
                lh_zero_tail(c, Q, R);
                LH_PA(4, t_q);
                LH_PT(t_ol);
                if (abr && !R.ath_over)
                    targ_bits[ch] = analog_silence_bits;    /* reference quantize.c:1953-1954 */
                lh_rg_put(c, R, g);
                if (lq_needs_tail(c, Q, R))
                    lq_outer_loop_stage5(ch, gr, targ_bits[ch]);
                else
                    lq_outer_loop_stage4(ch, gr, targ_bits[ch]);
                R = lh_uniform(L.rg[ch].R);
                g = lh_uniform(L.rg[ch].g);
                LH_PA(5, t_ol);
            }
            LH_PT(t_fin);
            lh_rg_put(c, R, g);
            lh_best_scalefac_store(ch, gr, fo->gr[0][ch].scalefac, L.block_type[0][ch]);
            if (cfg->use_best_huffman == 1)
                lh_best_huffman_divide(ch);
            g = lh_uniform(L.rg[ch].g);
            LH_PA(6, t_fin);
            lh_store_granule(c, Q, R, g, xr, o);
            if (lh_uni_i(L.ctx.bytes != nullptr))
                lh_emit_part_stage(ch, gr);     /* R / g are in the wave's LDS slot since the last stage call */
            LH_PA(3, t_q);
            if (lane == 0)
                L.bits_used[ch] = g.part2_3_length + g.part2_length;
        }
        LH_SYNC_WG();
        ResvSize -= L.bits_used[0] + L.bits_used[1];
        total_bits += L.bits_used[0] + L.bits_used[1];
        LH_SYNC_WG();
    }
    if (abr) {
        /* the smallest frame that brings the reservoir back to a non-negative size
         * (reference quantize.c:1964-1969) */
        int     i, mb, rm;
        for (i = cfg->vbr_min_bitrate_index; i < cfg->vbr_max_bitrate_index; i++)
            if (lh_vbr_full_bits(cfg, i, ResvSize, &mb, &rm) >= 0)
                break;
        (void) lh_vbr_full_bits(cfg, i, ResvSize, &mb, &rm);
        bitrate_index = i;
        frame_bits = lh_frame_bits(cfg, bitrate_index, 0);
        mean_bits = mb;
        ResvMax = rm;
    }
    LH_PA(27, t_frame);
    /* ---- ResvFrameEnd (reference reservoir.c:238-293) ---- */
    int     drain_pre = 0, drain_post = 0;
    {
        int     stuffingBits = 0, over_bits;
        ResvSize += mean_bits * cfg->mode_gr;
        if ((over_bits = ResvSize % 8) != 0)
            stuffingBits += over_bits;
        over_bits = (ResvSize - stuffingBits) - ResvMax;
        if (over_bits > 0)
            stuffingBits += over_bits;
        {
            int const m = mdb * 8;
            int const mdb_bytes = ((m < stuffingBits) ? m : stuffingBits) / 8;
            drain_pre += 8 * mdb_bytes;
            stuffingBits -= 8 * mdb_bytes;
            ResvSize -= 8 * mdb_bytes;
            mdb -= mdb_bytes;
        }
        drain_post += stuffingBits;
        ResvSize -= stuffingBits;
    }
    int const mdb_header = mdb;         /* the back pointer this frame's header carries */
    /* main_data_begin bookkeeping of format_bitstream (reference bitstream.c:917-935) */
    {
        int const bits = 8 * cfg->sideinfo_len + total_bits + drain_post;
        mdb += (frame_bits - bits) / 8;
    }
    if (tid == 0) {
        for (int ch = 0; ch < 2; ch++)
            for (int i = 0; i < 4; i++)
                fo->scfsi[ch][i] = (ch < nch) ? (int8_t) L.scfsi[ch][i] : (int8_t) 0;
        fo->main_data_begin = (int16_t) mdb;
        fo->resvDrain_pre = (int16_t) drain_pre;
        fo->resvDrain_post = (int16_t) drain_post;
        fo->bitrate_index = (int8_t) bitrate_index;
        fo->padding = (int8_t) padding;
        fo->mode_ext = (int8_t) mode_ext;
        for (int i = 0; i < 7; i++)
            fo->pad[i] = 0;
        fo->resv_size = ResvSize;
        fo->frame_bits = frame_bits;
        st->slot_lag = slot_lag;
        st->ResvSize = ResvSize;
        st->ResvMax = ResvMax;
        st->main_data_begin = mdb;
        st->substep_shaping = substep;
        /* what the next frame's psy model finds in sv_qnt.masking_lower: the CBR loop leaves the value
         * of its last granule/channel (channel 0 for mono), the VBR loop always the long-block one (reference quantize.c:1622) */
        st->masking_lower = (vbr_new || L.block_type[1][nch - 1] != LH_SHORT_TYPE) ? cfg->masking_lower_long
            : cfg->masking_lower_short;
        st->frame_number = st->frame_number + 1;
        if (mdb * 8 != ResvSize)
            st->status |= 1;    /* reservoir inconsistency (reference bitstream.c:947) */
    }
    if (lh_uni_i(L.ctx.bytes != nullptr))
        lh_emit_frame(fo, drain_pre, drain_post, frame_bits / 8, mdb_header, bitrate_index, padding, mode_ext,
                      c.d.flush && (int) ((c.frame_base + LH_MF_START) / 1152) == c.d.frame_end - 1);
    LH_PA(0, t_frame);
    LH_SYNC_WG();
#if defined(LH_PROF) && !defined(LH_EMU)
    if (lane < LH_NPROF)
        st->prof[w][lane] += L.prof[w][lane];
    LH_SYNC_WG();
#endif
}

#ifndef LH_EMU
extern "C" __global__ void __launch_bounds__(LH_NT, 2)
#else
void
#endif
lh_encode_kernel(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf,
                 const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, uint8_t * bytes,
                 int nstreams)
{
    LhLds & L = lh_lds;
    int const sidx = (int) blockIdx.x;
    if (sidx >= nstreams)
        return;
#ifdef LH_EMU
    /* test aid: LDS does not survive from launch to launch on the device; make any
     * dependence on stale contents visible to the CPU emulation */
    if (lh_emu_poison_lds) {
        if (threadIdx.x == 0)
            memset((void *) &L, 0xA5, sizeof(L));
        LH_SYNC_WG();
    }
#endif
    LhCtx   c;
    c.cfg = cfg;
    c.T = T;
    c.st = &states[sidx];
    c.pcm = pcm;
    c.pcmf = pcmf;
    c.d = descs[sidx];
    c.tid = (int) threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = lh_uni_i(c.tid >> 6);      /* scalar: everything indexed by the wave id gets scalar addressing */
    lh_ctx_hot(c);
    if (c.tid == 0) {
        L.ctx.cfg = c.cfg;
        L.ctx.T = c.T;
        L.ctx.st = c.st;
        L.ctx.pcm = c.pcm;
        L.ctx.pcmf = c.pcmf;
        L.ctx.bytes = bytes;
        L.ctx.d = c.d;
    }
    /* partition start tables (prefix sums of numlines): constant for the launch, kept in LDS */
    if (c.tid < 64) {
        int     a = 0, b = 0;
        for (int k = 0; k < c.tid; k++) {
            a += T->psy_l.numlines[k];
            b += T->psy_s.numlines[k];
        }
        L.pstart_l[c.tid] = (uint16_t) a;
        L.pstart_s[c.tid] = (uint16_t) b;
    }
    for (int f = c.d.frame_begin; f < c.d.frame_end; f++) {
        c.frame_base = 1152LL * f - LH_MF_START;
        if (c.tid == 0)
            L.ctx.frame_base = c.frame_base;    /* read by the stages after the next workgroup barrier */
        lh_encode_frame(c, &out[c.d.out_index + (f - c.d.frame_begin)]);
    }
}

#ifndef LH_EMU
/* device self-test of the cross-lane primitives in lh_wave.h: each reduction is
 * compared with a serial evaluation through LDS; out[0] = number of mismatches */
extern "C" __global__ void __launch_bounds__(64)
lh_selftest_kernel(unsigned *out, unsigned seed)
{
    __shared__ unsigned vals[64];
    unsigned const lane = threadIdx.x;
    unsigned bad = 0;
    for (unsigned round = 0; round < 64; round++) {
        unsigned x = (lane * 2654435761u) ^ (seed + round * 40503u);
        x ^= x >> 13;
        x *= 1274126177u;
        x ^= x >> 16;
        if (round & 1)
            x &= 0xffffu;       /* small values: sums stay below 2^32 */
        vals[lane] = x;
        __syncthreads();
        unsigned rsum = 0, rmax = 0, rmin = 0xffffffffu, ror = 0;
        unsigned long long rbal = 0;
        for (int i = 0; i < 64; i++) {
            rsum += vals[i];
            rmax = vals[i] > rmax ? vals[i] : rmax;
            rmin = vals[i] < rmin ? vals[i] : rmin;
            ror |= vals[i];
            if (vals[i] & 4u)
                rbal |= 1ull << i;
        }
        bad += (lh_wave_sum_u32(x) != rsum);
        bad += (lh_wave_max_u32(x) != rmax);
        bad += (lh_wave_min_u32(x) != rmin);
        bad += (lh_wave_or_u32(x) != ror);
        bad += (lh_ballot((x & 4u) != 0) != rbal);
        bad += (lh_bcast_u32(x, (int) (round & 63)) != vals[round & 63]);
        bad += (lh_ffs64(rbal) != (rbal ? __builtin_ctzll(rbal) : -1));
        bad += (lh_clz64(rbal | 1ull) != __builtin_clzll(rbal | 1ull));
        {
            /* float maximum over values of both signs */
            float const f = (float) (int) (x & 0xffffu) - 32768.0f;
            float   rf = -1e30f;
            for (int i = 0; i < 64; i++) {
                float const fi = (float) (int) (vals[i] & 0xffffu) - 32768.0f;
                rf = fi > rf ? fi : rf;
            }
            bad += (lh_wave_max_f32(f) != rf);
        }
        __syncthreads();
    }
    bad = lh_wave_sum_u32(bad);
    if (lane == 0)
        out[0] = bad;
}

extern "C" int
lh_launch_selftest(unsigned *d_out, unsigned seed, void *stream)
{
    hipLaunchKernelGGL(lh_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, d_out, seed);
    return (int) hipGetLastError();
}

/* host-side launcher with a C ABI for lh_api.cpp */
extern "C" int
lh_launch_encode(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf,
                 const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, uint8_t * bytes,
                 int nstreams, void *stream)
{
    if (nstreams <= 0)
        return 0;
    hipLaunchKernelGGL(lh_encode_kernel, dim3((unsigned) nstreams), dim3(LH_NT), 0,
                       (hipStream_t) stream, cfg, T, pcm, pcmf, descs, states, out, bytes, nstreams);
    return (int) hipGetLastError();
}

/* ---- test aid: leave a known garbage pattern in every VGPR/AGPR, in 64 KiB of LDS and in
 * scratch memory of all CUs, so that a kernel launched afterwards that depends on
 * uninitialised registers, LDS or scratch fails deterministically (lamehip_debug_poison). */
extern "C" __global__ void __launch_bounds__(64)
lh_poison_kernel(unsigned pattern, unsigned *sink)
{
    __shared__ unsigned lds[16384];
    volatile unsigned priv[512];
    for (int i = (int) threadIdx.x; i < 16384; i += 64)
        lds[i] = (pattern & 0xffffff00u) | 0x33u;
    for (int i = 0; i < 512; i++)
        priv[i] = (pattern & 0xffffff00u) | 0x44u;
    asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n v_mov_b32 v56, %0\n v_mov_b32 v57, %0\n v_mov_b32 v58, %0\n v_mov_b32 v59, %0\n v_mov_b32 v60, %0\n v_mov_b32 v61, %0\n v_mov_b32 v62, %0\n v_mov_b32 v63, %0\n v_mov_b32 v64, %0\n v_mov_b32 v65, %0\n v_mov_b32 v66, %0\n v_mov_b32 v67, %0\n v_mov_b32 v68, %0\n v_mov_b32 v69, %0\n v_mov_b32 v70, %0\n v_mov_b32 v71, %0\n v_mov_b32 v72, %0\n v_mov_b32 v73, %0\n v_mov_b32 v74, %0\n v_mov_b32 v75, %0\n v_mov_b32 v76, %0\n v_mov_b32 v77, %0\n v_mov_b32 v78, %0\n v_mov_b32 v79, %0\n v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %0\n v_mov_b32 v83, %0\n v_mov_b32 v84, %0\n v_mov_b32 v85, %0\n v_mov_b32 v86, %0\n v_mov_b32 v87, %0\n v_mov_b32 v88, %0\n v_mov_b32 v89, %0\n v_mov_b32 v90, %0\n v_mov_b32 v91, %0\n v_mov_b32 v92, %0\n v_mov_b32 v93, %0\n v_mov_b32 v94, %0\n v_mov_b32 v95, %0\n v_mov_b32 v96, %0\n v_mov_b32 v97, %0\n v_mov_b32 v98, %0\n v_mov_b32 v99, %0\n v_mov_b32 v100, %0\n v_mov_b32 v101, %0\n v_mov_b32 v102, %0\n v_mov_b32 v103, %0\n v_mov_b32 v104, %0\n v_mov_b32 v105, %0\n v_mov_b32 v106, %0\n v_mov_b32 v107, %0\n v_mov_b32 v108, %0\n v_mov_b32 v109, %0\n v_mov_b32 v110, %0\n v_mov_b32 v111, %0\n v_mov_b32 v112, %0\n v_mov_b32 v113, %0\n v_mov_b32 v114, %0\n v_mov_b32 v115, %0\n v_mov_b32 v116, %0\n v_mov_b32 v117, %0\n v_mov_b32 v118, %0\n v_mov_b32 v119, %0\n v_mov_b32 v120, %0\n v_mov_b32 v121, %0\n v_mov_b32 v122, %0\n v_mov_b32 v123, %0\n v_mov_b32 v124, %0\n v_mov_b32 v125, %0\n v_mov_b32 v126, %0\n v_mov_b32 v127, %0\n v_mov_b32 v128, %0\n v_mov_b32 v129, %0\n v_mov_b32 v130, %0\n v_mov_b32 v131, %0\n v_mov_b32 v132, %0\n v_mov_b32 v133, %0\n v_mov_b32 v134, %0\n v_mov_b32 v135, %0\n v_mov_b32 v136, %0\n v_mov_b32 v137, %0\n v_mov_b32 v138, %0\n v_mov_b32 v139, %0\n v_mov_b32 v140, %0\n v_mov_b32 v141, %0\n v_mov_b32 v142, %0\n v_mov_b32 v143, %0\n v_mov_b32 v144, %0\n v_mov_b32 v145, %0\n v_mov_b32 v146, %0\n v_mov_b32 v147, %0\n v_mov_b32 v148, %0\n v_mov_b32 v149, %0\n v_mov_b32 v150, %0\n v_mov_b32 v151, %0\n v_mov_b32 v152, %0\n v_mov_b32 v153, %0\n v_mov_b32 v154, %0\n v_mov_b32 v155, %0\n v_mov_b32 v156, %0\n v_mov_b32 v157, %0\n v_mov_b32 v158, %0\n v_mov_b32 v159, %0\n v_mov_b32 v160, %0\n v_mov_b32 v161, %0\n v_mov_b32 v162, %0\n v_mov_b32 v163, %0\n v_mov_b32 v164, %0\n v_mov_b32 v165, %0\n v_mov_b32 v166, %0\n v_mov_b32 v167, %0\n v_mov_b32 v168, %0\n v_mov_b32 v169, %0\n v_mov_b32 v170, %0\n v_mov_b32 v171, %0\n v_mov_b32 v172, %0\n v_mov_b32 v173, %0\n v_mov_b32 v174, %0\n v_mov_b32 v175, %0\n v_mov_b32 v176, %0\n v_mov_b32 v177, %0\n v_mov_b32 v178, %0\n v_mov_b32 v179, %0\n v_mov_b32 v180, %0\n v_mov_b32 v181, %0\n v_mov_b32 v182, %0\n v_mov_b32 v183, %0\n v_mov_b32 v184, %0\n v_mov_b32 v185, %0\n v_mov_b32 v186, %0\n v_mov_b32 v187, %0\n v_mov_b32 v188, %0\n v_mov_b32 v189, %0\n v_mov_b32 v190, %0\n v_mov_b32 v191, %0\n v_mov_b32 v192, %0\n v_mov_b32 v193, %0\n v_mov_b32 v194, %0\n v_mov_b32 v195, %0\n v_mov_b32 v196, %0\n v_mov_b32 v197, %0\n v_mov_b32 v198, %0\n v_mov_b32 v199, %0\n v_mov_b32 v200, %0\n v_mov_b32 v201, %0\n v_mov_b32 v202, %0\n v_mov_b32 v203, %0\n v_mov_b32 v204, %0\n v_mov_b32 v205, %0\n v_mov_b32 v206, %0\n v_mov_b32 v207, %0\n v_mov_b32 v208, %0\n v_mov_b32 v209, %0\n v_mov_b32 v210, %0\n v_mov_b32 v211, %0\n v_mov_b32 v212, %0\n v_mov_b32 v213, %0\n v_mov_b32 v214, %0\n v_mov_b32 v215, %0\n v_mov_b32 v216, %0\n v_mov_b32 v217, %0\n v_mov_b32 v218, %0\n v_mov_b32 v219, %0\n v_mov_b32 v220, %0\n v_mov_b32 v221, %0\n v_mov_b32 v222, %0\n v_mov_b32 v223, %0\n v_mov_b32 v224, %0\n v_mov_b32 v225, %0\n v_mov_b32 v226, %0\n v_mov_b32 v227, %0\n v_mov_b32 v228, %0\n v_mov_b32 v229, %0\n v_mov_b32 v230, %0\n v_mov_b32 v231, %0\n v_mov_b32 v232, %0\n v_mov_b32 v233, %0\n v_mov_b32 v234, %0\n v_mov_b32 v235, %0\n v_mov_b32 v236, %0\n v_mov_b32 v237, %0\n v_mov_b32 v238, %0\n v_mov_b32 v239, %0\n v_mov_b32 v240, %0\n v_mov_b32 v241, %0\n v_mov_b32 v242, %0\n v_mov_b32 v243, %0\n v_mov_b32 v244, %0\n v_mov_b32 v245, %0\n v_mov_b32 v246, %0\n v_mov_b32 v247, %0\n v_mov_b32 v248, %0\n v_mov_b32 v249, %0\n v_mov_b32 v250, %0\n v_mov_b32 v251, %0\n v_mov_b32 v252, %0\n v_mov_b32 v253, %0\n v_mov_b32 v254, %0\n v_mov_b32 v255, %0\n" :: "v"((pattern & 0xffffff00u) | 0x11u) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0\n v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %0\n v_accvgpr_write_b32 a6, %0\n v_accvgpr_write_b32 a7, %0\n v_accvgpr_write_b32 a8, %0\n v_accvgpr_write_b32 a9, %0\n v_accvgpr_write_b32 a10, %0\n v_accvgpr_write_b32 a11, %0\n v_accvgpr_write_b32 a12, %0\n v_accvgpr_write_b32 a13, %0\n v_accvgpr_write_b32 a14, %0\n v_accvgpr_write_b32 a15, %0\n v_accvgpr_write_b32 a16, %0\n v_accvgpr_write_b32 a17, %0\n v_accvgpr_write_b32 a18, %0\n v_accvgpr_write_b32 a19, %0\n v_accvgpr_write_b32 a20, %0\n v_accvgpr_write_b32 a21, %0\n v_accvgpr_write_b32 a22, %0\n v_accvgpr_write_b32 a23, %0\n v_accvgpr_write_b32 a24, %0\n v_accvgpr_write_b32 a25, %0\n v_accvgpr_write_b32 a26, %0\n v_accvgpr_write_b32 a27, %0\n v_accvgpr_write_b32 a28, %0\n v_accvgpr_write_b32 a29, %0\n v_accvgpr_write_b32 a30, %0\n v_accvgpr_write_b32 a31, %0\n v_accvgpr_write_b32 a32, %0\n v_accvgpr_write_b32 a33, %0\n v_accvgpr_write_b32 a34, %0\n v_accvgpr_write_b32 a35, %0\n v_accvgpr_write_b32 a36, %0\n v_accvgpr_write_b32 a37, %0\n v_accvgpr_write_b32 a38, %0\n v_accvgpr_write_b32 a39, %0\n v_accvgpr_write_b32 a40, %0\n v_accvgpr_write_b32 a41, %0\n v_accvgpr_write_b32 a42, %0\n v_accvgpr_write_b32 a43, %0\n v_accvgpr_write_b32 a44, %0\n v_accvgpr_write_b32 a45, %0\n v_accvgpr_write_b32 a46, %0\n v_accvgpr_write_b32 a47, %0\n v_accvgpr_write_b32 a48, %0\n v_accvgpr_write_b32 a49, %0\n v_accvgpr_write_b32 a50, %0\n v_accvgpr_write_b32 a51, %0\n v_accvgpr_write_b32 a52, %0\n v_accvgpr_write_b32 a53, %0\n v_accvgpr_write_b32 a54, %0\n v_accvgpr_write_b32 a55, %0\n v_accvgpr_write_b32 a56, %0\n v_accvgpr_write_b32 a57, %0\n v_accvgpr_write_b32 a58, %0\n v_accvgpr_write_b32 a59, %0\n v_accvgpr_write_b32 a60, %0\n v_accvgpr_write_b32 a61, %0\n v_accvgpr_write_b32 a62, %0\n v_accvgpr_write_b32 a63, %0\n v_accvgpr_write_b32 a64, %0\n v_accvgpr_write_b32 a65, %0\n v_accvgpr_write_b32 a66, %0\n v_accvgpr_write_b32 a67, %0\n v_accvgpr_write_b32 a68, %0\n v_accvgpr_write_b32 a69, %0\n v_accvgpr_write_b32 a70, %0\n v_accvgpr_write_b32 a71, %0\n v_accvgpr_write_b32 a72, %0\n v_accvgpr_write_b32 a73, %0\n v_accvgpr_write_b32 a74, %0\n v_accvgpr_write_b32 a75, %0\n v_accvgpr_write_b32 a76, %0\n v_accvgpr_write_b32 a77, %0\n v_accvgpr_write_b32 a78, %0\n v_accvgpr_write_b32 a79, %0\n v_accvgpr_write_b32 a80, %0\n v_accvgpr_write_b32 a81, %0\n v_accvgpr_write_b32 a82, %0\n v_accvgpr_write_b32 a83, %0\n v_accvgpr_write_b32 a84, %0\n v_accvgpr_write_b32 a85, %0\n v_accvgpr_write_b32 a86, %0\n v_accvgpr_write_b32 a87, %0\n v_accvgpr_write_b32 a88, %0\n v_accvgpr_write_b32 a89, %0\n v_accvgpr_write_b32 a90, %0\n v_accvgpr_write_b32 a91, %0\n v_accvgpr_write_b32 a92, %0\n v_accvgpr_write_b32 a93, %0\n v_accvgpr_write_b32 a94, %0\n v_accvgpr_write_b32 a95, %0\n v_accvgpr_write_b32 a96, %0\n v_accvgpr_write_b32 a97, %0\n v_accvgpr_write_b32 a98, %0\n v_accvgpr_write_b32 a99, %0\n v_accvgpr_write_b32 a100, %0\n v_accvgpr_write_b32 a101, %0\n v_accvgpr_write_b32 a102, %0\n v_accvgpr_write_b32 a103, %0\n v_accvgpr_write_b32 a104, %0\n v_accvgpr_write_b32 a105, %0\n v_accvgpr_write_b32 a106, %0\n v_accvgpr_write_b32 a107, %0\n v_accvgpr_write_b32 a108, %0\n v_accvgpr_write_b32 a109, %0\n v_accvgpr_write_b32 a110, %0\n v_accvgpr_write_b32 a111, %0\n v_accvgpr_write_b32 a112, %0\n v_accvgpr_write_b32 a113, %0\n v_accvgpr_write_b32 a114, %0\n v_accvgpr_write_b32 a115, %0\n v_accvgpr_write_b32 a116, %0\n v_accvgpr_write_b32 a117, %0\n v_accvgpr_write_b32 a118, %0\n v_accvgpr_write_b32 a119, %0\n v_accvgpr_write_b32 a120, %0\n v_accvgpr_write_b32 a121, %0\n v_accvgpr_write_b32 a122, %0\n v_accvgpr_write_b32 a123, %0\n v_accvgpr_write_b32 a124, %0\n v_accvgpr_write_b32 a125, %0\n v_accvgpr_write_b32 a126, %0\n v_accvgpr_write_b32 a127, %0\n v_accvgpr_write_b32 a128, %0\n v_accvgpr_write_b32 a129, %0\n v_accvgpr_write_b32 a130, %0\n v_accvgpr_write_b32 a131, %0\n v_accvgpr_write_b32 a132, %0\n v_accvgpr_write_b32 a133, %0\n v_accvgpr_write_b32 a134, %0\n v_accvgpr_write_b32 a135, %0\n v_accvgpr_write_b32 a136, %0\n v_accvgpr_write_b32 a137, %0\n v_accvgpr_write_b32 a138, %0\n v_accvgpr_write_b32 a139, %0\n v_accvgpr_write_b32 a140, %0\n v_accvgpr_write_b32 a141, %0\n v_accvgpr_write_b32 a142, %0\n v_accvgpr_write_b32 a143, %0\n v_accvgpr_write_b32 a144, %0\n v_accvgpr_write_b32 a145, %0\n v_accvgpr_write_b32 a146, %0\n v_accvgpr_write_b32 a147, %0\n v_accvgpr_write_b32 a148, %0\n v_accvgpr_write_b32 a149, %0\n v_accvgpr_write_b32 a150, %0\n v_accvgpr_write_b32 a151, %0\n v_accvgpr_write_b32 a152, %0\n v_accvgpr_write_b32 a153, %0\n v_accvgpr_write_b32 a154, %0\n v_accvgpr_write_b32 a155, %0\n v_accvgpr_write_b32 a156, %0\n v_accvgpr_write_b32 a157, %0\n v_accvgpr_write_b32 a158, %0\n v_accvgpr_write_b32 a159, %0\n v_accvgpr_write_b32 a160, %0\n v_accvgpr_write_b32 a161, %0\n v_accvgpr_write_b32 a162, %0\n v_accvgpr_write_b32 a163, %0\n v_accvgpr_write_b32 a164, %0\n v_accvgpr_write_b32 a165, %0\n v_accvgpr_write_b32 a166, %0\n v_accvgpr_write_b32 a167, %0\n v_accvgpr_write_b32 a168, %0\n v_accvgpr_write_b32 a169, %0\n v_accvgpr_write_b32 a170, %0\n v_accvgpr_write_b32 a171, %0\n v_accvgpr_write_b32 a172, %0\n v_accvgpr_write_b32 a173, %0\n v_accvgpr_write_b32 a174, %0\n v_accvgpr_write_b32 a175, %0\n v_accvgpr_write_b32 a176, %0\n v_accvgpr_write_b32 a177, %0\n v_accvgpr_write_b32 a178, %0\n v_accvgpr_write_b32 a179, %0\n v_accvgpr_write_b32 a180, %0\n v_accvgpr_write_b32 a181, %0\n v_accvgpr_write_b32 a182, %0\n v_accvgpr_write_b32 a183, %0\n v_accvgpr_write_b32 a184, %0\n v_accvgpr_write_b32 a185, %0\n v_accvgpr_write_b32 a186, %0\n v_accvgpr_write_b32 a187, %0\n v_accvgpr_write_b32 a188, %0\n v_accvgpr_write_b32 a189, %0\n v_accvgpr_write_b32 a190, %0\n v_accvgpr_write_b32 a191, %0\n v_accvgpr_write_b32 a192, %0\n v_accvgpr_write_b32 a193, %0\n v_accvgpr_write_b32 a194, %0\n v_accvgpr_write_b32 a195, %0\n v_accvgpr_write_b32 a196, %0\n v_accvgpr_write_b32 a197, %0\n v_accvgpr_write_b32 a198, %0\n v_accvgpr_write_b32 a199, %0\n v_accvgpr_write_b32 a200, %0\n v_accvgpr_write_b32 a201, %0\n v_accvgpr_write_b32 a202, %0\n v_accvgpr_write_b32 a203, %0\n v_accvgpr_write_b32 a204, %0\n v_accvgpr_write_b32 a205, %0\n v_accvgpr_write_b32 a206, %0\n v_accvgpr_write_b32 a207, %0\n v_accvgpr_write_b32 a208, %0\n v_accvgpr_write_b32 a209, %0\n v_accvgpr_write_b32 a210, %0\n v_accvgpr_write_b32 a211, %0\n v_accvgpr_write_b32 a212, %0\n v_accvgpr_write_b32 a213, %0\n v_accvgpr_write_b32 a214, %0\n v_accvgpr_write_b32 a215, %0\n v_accvgpr_write_b32 a216, %0\n v_accvgpr_write_b32 a217, %0\n v_accvgpr_write_b32 a218, %0\n v_accvgpr_write_b32 a219, %0\n v_accvgpr_write_b32 a220, %0\n v_accvgpr_write_b32 a221, %0\n v_accvgpr_write_b32 a222, %0\n v_accvgpr_write_b32 a223, %0\n v_accvgpr_write_b32 a224, %0\n v_accvgpr_write_b32 a225, %0\n v_accvgpr_write_b32 a226, %0\n v_accvgpr_write_b32 a227, %0\n v_accvgpr_write_b32 a228, %0\n v_accvgpr_write_b32 a229, %0\n v_accvgpr_write_b32 a230, %0\n v_accvgpr_write_b32 a231, %0\n v_accvgpr_write_b32 a232, %0\n v_accvgpr_write_b32 a233, %0\n v_accvgpr_write_b32 a234, %0\n v_accvgpr_write_b32 a235, %0\n v_accvgpr_write_b32 a236, %0\n v_accvgpr_write_b32 a237, %0\n v_accvgpr_write_b32 a238, %0\n v_accvgpr_write_b32 a239, %0\n v_accvgpr_write_b32 a240, %0\n v_accvgpr_write_b32 a241, %0\n v_accvgpr_write_b32 a242, %0\n v_accvgpr_write_b32 a243, %0\n v_accvgpr_write_b32 a244, %0\n v_accvgpr_write_b32 a245, %0\n v_accvgpr_write_b32 a246, %0\n v_accvgpr_write_b32 a247, %0\n v_accvgpr_write_b32 a248, %0\n v_accvgpr_write_b32 a249, %0\n v_accvgpr_write_b32 a250, %0\n v_accvgpr_write_b32 a251, %0\n v_accvgpr_write_b32 a252, %0\n v_accvgpr_write_b32 a253, %0\n v_accvgpr_write_b32 a254, %0\n v_accvgpr_write_b32 a255, %0\n" :: "v"((pattern & 0xffffff00u) | 0x22u) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    __syncthreads();
    if (sink && pattern == 0x12345u)
        sink[threadIdx.x] = lds[threadIdx.x * 7] + priv[threadIdx.x & 511];
}

extern "C" int
lh_launch_poison(unsigned pattern, void *stream)
{
    /* 8 single-wave workgroups per CU (64 KiB LDS each: two fit per CU at a time) */
    hipLaunchKernelGGL(lh_poison_kernel, dim3(256 * 8 * 4), dim3(64), 0, (hipStream_t) stream, pattern,
                       (unsigned *) nullptr);
    return (int) hipGetLastError();
}
#else
extern "C" int lh_emu_encode_bytes(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const LhStreamDesc * descs,
                                   LhStreamState * states, LhFrameOut * out, uint8_t * bytes, int nstreams);
extern "C" int
lh_emu_encode(const LhConfig * cfg, const LhTables * T, const int16_t * pcm,
              const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, int nstreams)
{
    return lh_emu_encode_bytes(cfg, T, pcm, descs, states, out, (uint8_t *) 0, nstreams);
}

extern "C" int
lh_emu_encode_bytes(const LhConfig * cfg, const LhTables * T, const int16_t * pcm,
                    const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, uint8_t * bytes, int nstreams)
{
    hipemu_dim3 grid = { (unsigned) nstreams, 1, 1 }, block = { LH_NT, 1, 1 };
    hipemu_run(grid, block,[=] () {
               lh_encode_kernel(cfg, T, pcm, (const float *) 0, descs, states, out, bytes, nstreams);
               }
    );
    return 0;
}
#endif
