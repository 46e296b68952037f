/*
 * lh_kernels.hip -- the batched MP3 encode inner loop as one persistent HIP
 * kernel for gfx950 (MI355X).
 *
 * Decomposition: one workgroup of two wavefronts per
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

#include "lh_static_tables.h"
#include "lh_dev_common.h"
#include "lh_dev_psy.h"
#include "lh_dev_mdct.h"
#include "lh_dev_quant.h"

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

/* one frame of one stream; executed by the whole workgroup */
LH_DEVFN void
lh_encode_frame(LhCtx & c, LhLds & L, LhFrameOut * fo)
{
    const LhConfig *cfg = c.cfg;
    const LhTables *T = c.T;
    LhStreamState *st = c.st;
    int const w = c.wave, lane = c.lane, tid = c.tid;

    /* ---- polyphase priming on the first frame (reference encoder.c:189-236) ---- */
    if (!st->primed) {
        LhCtx   pc = c;
        pc.frame_base = c.frame_base - 1152;
        lh_polyphase(pc, w, L.u.mdct.sb[w]);
        for (int i = lane; i < 576; i += 64)
            st->sb_prev[w][i] = L.u.mdct.sb[w][2][i];
        LH_SYNC_WG();
        if (tid == 0)
            st->primed = 1;
    }
    LH_SYNC_WG();

    /* ---- padding (reference encoder.c:348-352) ---- */
    int     padding = 0;
    int     slot_lag = st->slot_lag - cfg->frac_SpF;
    if (slot_lag < 0) {
        slot_lag += cfg->samplerate;
        padding = 1;
    }

    /* ---- partition start tables (prefix sums of numlines) ---- */
    if (tid < 64) {
        int     a = 0, b = 0;
        for (int k = 0; k < tid; k++) {
            a += T->psy_l.numlines[k];
            b += T->psy_s.numlines[k];
        }
        L.pstart_l[tid] = a;
        L.pstart_s[tid] = b;
    }
    LH_SYNC_WG();

    /* ---- stage 1: psycho-acoustic model, two granules ---- */
    for (int gr = 0; gr < 2; gr++)
        lh_psy_granule(c, L, gr);

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
        loud[0][1] = L.loudness_sq[0][1];
        loud[1][0] = L.loudness_sq[1][0];
        loud[1][1] = L.loudness_sq[1][1];
        lh_adjust_ATH(T, loud, &factor, &limit);
        LH_SYNC_WG();
        if (tid == 0) {
            st->ath_adjust_factor = factor;
            st->ath_adjust_limit = limit;
        }
    }
    LH_SYNC_WG();

    /* ---- stage 2: polyphase + MDCT (reference encoder.c:405) ---- */
    for (int i = lane; i < 576; i += 64)
        L.u.mdct.sb[w][0][i] = st->sb_prev[w][i];
    lh_polyphase(c, w, L.u.mdct.sb[w]);
    lh_mdct_granules(c, L, w, L.u.mdct.sb[w]);
    for (int i = lane; i < 576; i += 64)
        st->sb_prev[w][i] = L.u.mdct.sb[w][2][i];
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
        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < 2; ch++) {
                pe_use[gr][ch] = L.pe[gr][msoff + ch];
                f += pe_use[gr][ch];
            }
        buf[18] = f;
        f = buf[9];
        for (int i = 0; i < 9; i++)
            f += (buf[i] + buf[18 - i]) * lh_pe_fir[i];
        f = (670 * 5 * 2 * 2) / f;
        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < 2; ch++)
                pe_use[gr][ch] *= f;
        LH_SYNC_WG();
        if (tid < 19)
            st->pefirbuf[tid] = buf[tid];
    }

    /* ---- stage 4: CBR iteration loop (reference quantize.c:1988-2050) ---- */
    int     ResvSize = st->ResvSize, ResvMax, mdb = st->main_data_begin;
    int     substep = st->substep_shaping;
    int const frame_bits = lh_frame_bits(cfg, cfg->bitrate_index, padding);
    int const mean_bits = (frame_bits - cfg->sideinfo_len * 8) / cfg->mode_gr;
    {
        /* ResvFrameBegin */
        int const resvLimit = (8 * 256) * cfg->mode_gr - 8;
        ResvMax = cfg->buffer_constraint - frame_bits;
        if (ResvMax > resvLimit)
            ResvMax = resvLimit;
        if (ResvMax < 0 || cfg->disable_reservoir)
            ResvMax = 0;
    }
    int     total_bits = 0;
    for (int gr = 0; gr < 2; gr++) {
        int     targ_bits[2];
        int     max_bits = lh_on_pe(cfg, ResvSize, ResvMax, &substep, pe_use[gr], targ_bits, mean_bits, gr);
        LH_SYNC_WG();
        if (mode_ext == LH_MPG_MD_MS_LR) {
            float const k = (float) (LH_SQRT2 * 0.5);
            for (int i = tid; i < 576; i += LH_NT) {
                float const l = L.xr[0][gr][i];
                float const r = L.xr[1][gr][i];
                L.xr[0][gr][i] = (l + r) * k;
                L.xr[1][gr][i] = (l - r) * k;
            }
            lh_reduce_side(targ_bits, ms_ener_ratio[gr], mean_bits, max_bits);
        }
        LH_SYNC_WG();
        {
            int const ch = w;
            LhChanLds & Q = L.u.quant.ch[ch];
            LhQR    R;
            LhGrR   g;
            float  *xr = L.xr[ch][gr];
            LhGranule *o = &fo->gr[gr][ch];
            lh_init_outer_loop(c, Q, R, g, xr, L.block_type[gr][ch], substep);
            if (lh_init_xrpow(c, Q, R, g, xr)) {
                lh_calc_xmin(c, Q, R, xr, L.ratio_en[gr][msoff + ch], L.ratio_thm[gr][msoff + ch]);
                (void) lh_outer_loop(c, Q, R, g, xr, ch, targ_bits[ch]);
            }
            lh_best_scalefac_store(c, Q, R, g, gr, fo->gr[0][ch].scalefac, L.block_type[0][ch],
                                   L.scfsi[ch]);
            if (cfg->use_best_huffman == 1)
                lh_best_huffman_divide(c, Q, R, g);
            lh_store_granule(c, Q, R, g, xr, o);
            if (lane == 0)
                L.bits_used[ch] = g.part2_3_length + g.part2_length;
        }
        LH_SYNC_WG();
        ResvSize -= L.bits_used[0] + L.bits_used[1];
        total_bits += L.bits_used[0] + L.bits_used[1];
        LH_SYNC_WG();
    }
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
    /* main_data_begin bookkeeping of format_bitstream (reference bitstream.c:917-935) */
    {
        int const bits = 8 * cfg->sideinfo_len + total_bits + drain_post;
        mdb += (frame_bits - bits) / 8;
    }
    if (tid == 0) {
        for (int ch = 0; ch < 2; ch++)
            for (int i = 0; i < 4; i++)
                fo->scfsi[ch][i] = (int8_t) L.scfsi[ch][i];
        fo->main_data_begin = (int16_t) mdb;
        fo->resvDrain_pre = (int16_t) drain_pre;
        fo->resvDrain_post = (int16_t) drain_post;
        fo->bitrate_index = (int8_t) cfg->bitrate_index;
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
        st->masking_lower = (L.block_type[1][1] != LH_SHORT_TYPE) ? cfg->masking_lower_long
            : cfg->masking_lower_short;
        st->frame_number = st->frame_number + 1;
        if (mdb * 8 != ResvSize)
            st->status |= 1;    /* reservoir inconsistency (reference bitstream.c:947) */
    }
    LH_SYNC_WG();
}

#ifndef LH_EMU
extern "C" __global__ void __launch_bounds__(LH_NT)
#else
void
#endif
lh_encode_kernel(const LhConfig * cfg, const LhTables * T, const int16_t * pcm,
                 const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out,
                 int nstreams)
{
    __shared__ LhLds L;
    int const sidx = (int) blockIdx.x;
    if (sidx >= nstreams)
        return;
    LhCtx   c;
    c.cfg = cfg;
    c.T = T;
    c.st = &states[sidx];
    c.pcm = pcm;
    c.d = descs[sidx];
    c.tid = (int) threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = c.tid >> 6;
    for (int f = c.d.frame_begin; f < c.d.frame_end; f++) {
        c.frame_base = 1152LL * f - LH_MF_START;
        lh_encode_frame(c, L, &out[c.d.out_index + (f - c.d.frame_begin)]);
    }
}

#ifndef LH_EMU
/* host-side launcher with a C ABI for lh_api.cpp */
extern "C" int
lh_launch_encode(const LhConfig * cfg, const LhTables * T, const int16_t * pcm,
                 const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out,
                 int nstreams, void *stream)
{
    if (nstreams <= 0)
        return 0;
    hipLaunchKernelGGL(lh_encode_kernel, dim3((unsigned) nstreams), dim3(LH_NT), 0,
                       (hipStream_t) stream, cfg, T, pcm, descs, states, out, nstreams);
    return (int) hipGetLastError();
}
#else
extern "C" int
lh_emu_encode(const LhConfig * cfg, const LhTables * T, const int16_t * pcm,
              const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, int nstreams)
{
    hipemu_dim3 grid = { (unsigned) nstreams, 1, 1 }, block = { LH_NT, 1, 1 };
    hipemu_run(grid, block,[=] () {
               lh_encode_kernel(cfg, T, pcm, descs, states, out, nstreams);
               }
    );
    return 0;
}
#endif
