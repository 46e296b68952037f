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

#ifdef LH_EMU
#include <string.h>
#if defined(LH_LSF) || defined(LH_SPLIT)
extern "C" { extern int lh_emu_poison_lds; }
#else
extern "C" { int lh_emu_poison_lds = 0; }
#endif
#endif
#include "lh_static_tables.h"
#include "lh_dev_common.h"
#include "lh_dev_psy.h"
#include "lh_dev_mdct.h"
#include "lh_dev_quant.h"
#include "lh_dev_qloop.h"

/* The ATH level follows the loudness of the frame (reference encoder.c:56-137, adjust_ATH; wave-uniform):
 * `factor' scales the threshold in quiet, `limit' is the floor it may not be raised above on the way
 * back up.  A loud frame (half the louder granule's loudness, times the sensitivity, above 1/32)
 * snaps the factor back into [limit, 1]; a quiet one pulls it towards 31.98 x that level + 0.000625 --
 * from above by 7.5 % of the way per frame, from below at once, within the previous limit. */
LH_DEVFN void
lh_adjust_ATH(const LhTables * T, const float loud[2][2], float *factor, float *limit)
{
    float const was = *factor, floor_was = *limit;
    float   level, now, floor_now;
    if (T->ath_use_adjust == 0) {
        *factor = 1.0;
        return;
    }
    {
        float const first = loud[0][0] + loud[0][1], second = loud[1][0] + loud[1][1];
        /* (MPEG-2 / 2.5: the frame's only granule, reference encoder.c:80-82: loud[1] is passed as loud[0] then) */
        level = (first > second) ? first : second;
        level = (float) (level * 0.5);
        level *= T->aa_sensitivity_p;
    }
    if (level > 0.03125) {
        now = (was >= 1.0) ? 1.0f : (was < floor_was) ? floor_was : was;
        floor_now = 1.0;
    }
    else {
        float const quiet = (float) (31.98 * level + 0.000625);
        if (was >= quiet) {
            float const eased = (float) (was * (quiet * 0.075 + 0.925));
            now = (eased < quiet) ? quiet : eased;
        }
        else
            now = (floor_was >= quiet) ? quiet : (was < floor_was) ? floor_was : was;
        floor_now = quiet;
    }
    *factor = now;
    *limit = floor_now;
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
        q.pow43h[i] = c.T->pow43[i];
        q.qthr[i] = c.T->qthr[i];
    }
    static_assert(sizeof(q.ht_len) % 4 == 0 && __builtin_offsetof(LhQTabs, ht_len) % 4 == 0, "ht_len is copied by words");
    for (int i = c.tid; i < (int) sizeof(q.ht_len) / 4; i += LH_NT)
        ((uint32_t *) q.ht_len)[i] = ((const uint32_t *) lh_ht_len)[i];
    {
        /* the region split by big_values, folded with the band edges by the host (LhTables.bvpack): three loads in
         * flight per thread instead of two dependent pairs per entry */
        uint32_t const b0 = c.T->bvpack[c.tid], b1 = c.T->bvpack[c.tid + LH_NT], b2 = c.T->bvpack[256 + (c.tid & 31)];
        q.bvpack[c.tid] = b0;
        q.bvpack[c.tid + LH_NT] = b1;
        if (c.tid < 32)
            q.bvpack[256 + c.tid] = b2;
    }
    if (c.tid == 0) {
        q.sfb_s3 = (uint16_t) c.T->sfb_s[3];
        q.pad = 0;
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
        {
            int const j = c.tid, idx = ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3);
            q.t3233p[j] = ((uint32_t) lh_t32l[idx] << 16) | lh_t33l[idx];
        }
    }
    if (c.tid < 24) {
        q.sfb_l[c.tid] = (uint16_t) ((c.tid < 23) ? c.T->sfb_l[c.tid] : 576);
        q.pretab[c.tid] = (c.tid < 22) ? lh_pretab[c.tid] : 0;
    }
}

/* what the iteration loops look up on their dependent chains, into LDS: the quantiser tables (behind xr, where the fused
 * kernel's PCM window lies earlier in a frame) and, over the second quantised image, the step tables of the VBR
 * scalefactor search or calc_noise's log table.  Whole workgroup; the caller's barrier follows.  The fused kernel does
 * this once per frame (its psy model and its window use the same LDS), the split pipeline's encode kernel once per launch. */
LH_DEVFN void
lh_stage_loop_tables(const LhCtx & c)
{
    const LhTables *T = c.T;
    lh_load_qtabs(c, lh_lds.qt);
    if (c.cfg->vbr == 1 || c.cfg->vbr == 4) {
        for (int i = c.tid; i < 256; i += LH_NT) {
            LH_VBR_IPOW20[i] = T->ipow20[i];
            LH_VBR_POW20[i] = T->pow20[i + LH_QMAX2];
        }
    }
    else {
        for (int i = c.tid; i < 513; i += LH_NT)
            LH_LOGT_LDS_W(i) = T->log_table[i];
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

#ifdef LH_DEBUG_DUMP
/* test build: see LhStreamState.dbg_* (lh_device.h) */
LH_DEVFN void
lh_dbg_xmin(const LhCtx & c, int gr, int ch, int rch, const LhChanLds & Q, int psymax)
{
    int const slot = (lh_uni_i(lh_lds.psy_slot) + gr) % 3;
    if (c.lane <= LH_SFBMAX)
        c.st->dbg_xmin[gr][ch][c.lane] = (c.lane < psymax) ? Q.l3_xmin[c.lane < LH_SFBMAX ? c.lane : 0] : 0.0f;
    if (c.lane < LH_XMIN_N) {
        c.st->dbg_en[gr][ch][c.lane] = lh_lds.psy_en[slot][rch][c.lane];
        c.st->dbg_thm[gr][ch][c.lane] = lh_lds.psy_thm[slot][rch][c.lane];
    }
}

LH_DEVFN void
lh_dbg_xr(const LhCtx & c, int gr, int ch, const float *xr)
{
    for (int i = c.lane; i < 576; i += 64)
        c.st->dbg_xr[gr][ch][i] = xr[i];
}
#define LH_DBG_XMIN(c, gr, ch, rch, Q, psymax) lh_dbg_xmin(c, gr, ch, rch, Q, psymax)
#define LH_DBG_XR(c, gr, ch, xr) lh_dbg_xr(c, gr, ch, xr)
#else
#define LH_DBG_XMIN(c, gr, ch, rch, Q, psymax) do { } while (0)
#define LH_DBG_XR(c, gr, ch, xr) do { } while (0)
#endif

#include "lh_dev_emit.h"
#include "lh_dev_vbr.h"
#ifdef LH_VBR_OLD
#include "lh_dev_vbrold.h"
#endif

/* what a granule's search starts from and what does not depend on its bit budget: geometry (init_outer_loop),
 * xrpow, the allowed noise per band (calc_xmin); R / g are left in the channel's LDS slot.  Returns 0 for an all-zero
 * granule.  substep: only bit 1 (a constant of the settings) is looked at downstream. */
LH_DEVFN int
lh_prepare_granule(const LhCtx & c, int ch, int gr, int msoff, int substep)
{
    LhLds & L = lh_lds;
    LhChanLds & Q = L.u.quant.ch[ch];
    LhQR    R;
    LhGrR   g;
    int     nonzero;
    int const usual = lh_uni_i(lh_granule_is_usual(c, L.block_type[gr][ch], substep));
    if (usual)
        lh_init_outer_loop_n(ch, gr, substep);
    else
        lh_init_outer_loop(ch, gr, L.block_type[gr][ch], substep);
    R = lh_uniform(L.rg[ch].R);
    g = lh_uniform(L.rg[ch].g);
    nonzero = lh_init_xrpow(c, Q, R, g, L.xr[ch][gr]);
    if (nonzero) {
        lh_rg_put(c, R, g);
        if (usual)
            lh_calc_xmin_n(ch, gr, msoff + ch);
        else
            lh_calc_xmin(ch, gr, msoff + ch);
        R = lh_uniform(L.rg[ch].R);
        LH_DBG_XMIN(c, gr, ch, msoff + ch, Q, R.psymax);
        lh_zero_tail(c, Q, R);
    }
    lh_rg_put(c, R, g);
    return lh_uni_i(nonzero);
}


#if !defined(LH_EMU)
/* Issue priority by progress and by role.  At 1024 streams every SIMD hosts two waves of two different streams for the whole
 * launch, and the SIMD's issue arbiter prefers the older of two ready waves: identical streams finish up to 14 % apart
 * depending on where the dispatcher put them (tools/stream_balance.py, tools/ubench/hwid.hip), streams that differ in content
 * further apart still -- and a launch ends with its last stream.  Each wave posts the frames its stream has left at its
 * (XCC, SE, SH, CU, SIMD, wave slot) and reads the other slot's figure once per frame: a wave whose stream is more than a
 * frame behind the other's takes the top priority (s_setprio), one that is more than a frame ahead the bottom one.  In
 * between the role decides: of a stream's two channels the one that reached the last granule barrier second -- the other
 * waited for it -- goes first.  Scheduling only: results cannot depend on it; a finished stream leaves 0. */
static __device__ int lh_prio_tab[8 * 8 * 2 * 16 * 4 * 2];
LH_DEVFN int
lh_prio_index()
{
    unsigned v, x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    /* HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13] */
    return (int) ((((((x & 7u) * 8u + ((v >> 13) & 7u)) * 2u + ((v >> 12) & 1u)) * 16u + ((v >> 8) & 15u)) * 4u + ((v >> 4) & 3u)) * 2u
                  + (v & 1u));
}
LH_DEVFN void
lh_prio_apply(int rel, int late)
{
    int const level = lh_uni_i(rel > 0 ? 3 : rel < 0 ? 0 : 1 + (late != 0));
    if (level == 3)
        __builtin_amdgcn_s_setprio(3);
    else if (level == 2)
        __builtin_amdgcn_s_setprio(2);
    else if (level == 1)
        __builtin_amdgcn_s_setprio(1);
    else
        __builtin_amdgcn_s_setprio(0);
}
/* frames left against the other slot's: +1 far behind, -1 far ahead, 0 within LH_PRIO_BAND frames */
#ifndef LH_PRIO_BAND
#define LH_PRIO_BAND 1
#endif
LH_DEVFN int
lh_prio_tick(int me, int left)
{
    if (lh_lane() == 0)
        __hip_atomic_store(&lh_prio_tab[me], left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int const other = lh_uni_i(__hip_atomic_load(&lh_prio_tab[me ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return (left > other + LH_PRIO_BAND) ? 1 : (left + LH_PRIO_BAND < other) ? -1 : 0;
}
#define LH_PRIO_INDEX() lh_prio_index()
#define LH_PRIO_TICK(me, left) lh_prio_tick(me, left)
#define LH_PRIO_APPLY(rel, late) lh_prio_apply(rel, late)
#else
#define LH_PRIO_INDEX() 0
#define LH_PRIO_TICK(me, left) 0
#define LH_PRIO_APPLY(rel, late) do { } while (0)
#endif

/* The frame and the stream loop are inlined into the kernel: out of line they saved and restored ~60 registers per
 * frame through scratch memory, 50 KB of HBM traffic per frame at no gain. */

/* The frame's barriers between stages that hand data on through LDS only (LH_SYNC_FRAME): an LDS-only fence.  A full
 * __syncthreads() also drains the wave's global stores -- the payload of the granule just finished -- which nobody in the
 * workgroup reads. */
#if defined(LH_BAR_LDS) && !defined(LH_EMU)
#define LH_SYNC_FRAME() LH_SYNC_WG_LDS()
#else
#define LH_SYNC_FRAME() LH_SYNC_WG()
#endif

/* one frame of one stream; executed by the whole workgroup */
LH_DEVFN void
lh_encode_frame(LhCtx & c, LhFrameOut * fo, LhWaveCarry & carry)
{
    LhLds & L = lh_lds;
    const LhConfig *cfg = c.cfg;
    const LhTables *T = c.T;
    LhStreamState *st = c.st;
    int const w = c.wave, lane = c.lane, tid = c.tid;
    /* mono: wave 1 has no channel; it runs the transforms on the duplicated PCM (never read) and
     * only keeps pace through the barriers of the quantisation stage */
    int const nch = cfg->channels;
    /* granules per frame: 2, or 1 for MPEG-2 / 2.5 (576 samples per frame; reference lame.c:797).  The transforms
     * below always run over two granules' worth of the staged window -- the second is the next frame's first and
     * is thrown away -- so that the one-granule frame needs no code of its own there. */
    constexpr int ngr = LH_NGR, fs = 576 * LH_NGR;

    /* ---- polyphase priming on the first frame (reference encoder.c:189-236) ---- */
#if defined(LH_PROF) && !defined(LH_EMU)
    if (lane < LH_NPROF)
        L.prof[w][lane] = 0;
#endif
    LH_PT(t_frame);
#ifdef LH_SPLIT
    /* the split pipeline: the PCM was read by the analysis kernels (lh_analysis.hip, lh_subband.hip), whose output for this
     * frame the stages below pick up from HBM (L.ctx.mid_*) */
    if (tid == 0)
        lh_lds.ss.primed = 1;
    {
        /* Everything the prologue needs from HBM in one batch, so that the frame waits for HBM once: the MDCT spectra of
         * both granules (L / R as lh_subband_kernel left them; the mid/side rotation follows below) to their place, the
         * small record and the long-block masking to the psy model's scratch. */
        const LhMidFrame *rec = LH_AS_GLOBAL(const LhMidFrame, L.ctx.mid);
        const lh_f32x4 *sx = (const lh_f32x4 *) rec->xr.xr, *sm = (const lh_f32x4 *) &rec->small;
        lh_f32x4 *dx = (lh_f32x4 *) L.xr, *dm = (lh_f32x4 *) &L.u.psy.mid.small;
        constexpr int NM = (int) ((sizeof(LhMidSmall) + sizeof(LhMidLong)) / 16);      /* 420 */
        static_assert(__builtin_offsetof(LhMidFrame, lng) == sizeof(LhMidSmall) && sizeof(LhMidSmall) % 16 == 0, "small and lng are one run");
        lh_f32x4 v[5], m[4];
#if !defined(LH_EMU) && defined(LH_MID_PREFETCH)
        /* (off since round 6: the touched lines did not survive a frame's time in the 4 MB of an XCD's L2 beside 128 streams' scratch
         * and payload lines and were fetched again -- 12.7 KB of the 31.6 KB the kernel read per frame -- for 0.3 % of its time:
         * profiles/r06_traffic_ab.txt)  The NEXT frame's record is asked for now, one word per cache line (a load nobody uses: it only brings the
         * lines into the L2 / the memory-side cache, a frame's time before the batch above is issued for them; the pool has
         * a margin of records behind the launch's last frame).  Issued first, so it has returned when the loads below have. */
        uint32_t touched;
        {
            int const line = tid < 53 ? tid : (tid < 125 ? tid - 53 : 0);
            const char *nx = (const char *) (rec + 1) + (tid < 53 ? 0 : __builtin_offsetof(LhMidFrame, xr)) + 128 * line;
            asm volatile("global_load_dword %0, %1, off" : "=v"(touched) : "v"(nx) : "memory");
        }
#endif
#pragma unroll
        for (int u = 0; u < 5; u++)
            v[u] = sx[(tid + LH_NT * u < 576) ? tid + LH_NT * u : 575];
#pragma unroll
        for (int u = 0; u < 4; u++)
            m[u] = sm[(tid + LH_NT * u < NM) ? tid + LH_NT * u : NM - 1];
#pragma unroll
        for (int u = 0; u < 5; u++)
            if (tid + LH_NT * u < 576)
                dx[tid + LH_NT * u] = v[u];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (tid + LH_NT * u < NM)
                dm[tid + LH_NT * u] = m[u];
#if !defined(LH_EMU) && defined(LH_MID_PREFETCH)
        asm volatile("" :: "v"(touched));       /* (its register is the load's until here) */
#endif
    }
    LH_SYNC_FRAME();
#else
    if (!lh_lds.ss.primed) {
        lh_stage_window(c, L.mf, c.frame_base - fs);
        LH_SYNC_WG();
        lh_polyphase(w);
#pragma unroll
        for (int k = 0; k < 9; k++)
            carry.sb[k] = L.u.mdct.sb[w][ngr][lane + 64 * k];
        LH_SYNC_WG();
        if (tid == 0)
            lh_lds.ss.primed = 1;
    }
    LH_SYNC_WG();

    lh_stage_window(c, L.mf, c.frame_base);
    LH_SYNC_WG();
#endif

    /* ---- padding (reference encoder.c:348-352) ---- */
    int     padding = 0;
    /* Frame-level scalars are the same in every lane; lh_uni_*() says so to the compiler, which then
     * keeps them in scalar registers across the stage calls instead of in vector registers that
     * the stages would have to save to scratch memory (scratch that falls out of the L2 is HBM traffic). */
    int     slot_lag = lh_uni_i(lh_lds.ss.slot_lag) - lh_uni_i(cfg->frac_SpF);
    if (slot_lag < 0) {
        slot_lag += cfg->samplerate;
        padding = 1;
    }


    LH_PA(24, t_frame);
    /* ---- stage 1: psycho-acoustic model, two granules ---- */
    LH_PT(t_psy);
    for (int gr = 0; gr < ngr; gr++)
        carry.nb = lh_psy_granule(gr, carry.nb);
    if (ngr == 1) {
        /* the transforms below also run over the window's second granule (thrown away): give it a block type */
        if (tid < 2)
            L.block_type[1][tid] = LH_NORM_TYPE;
        LH_SYNC_WG();
    }
    LH_PA(1, t_psy);

    float   ms_ener_ratio[2] = { .5f, .5f };
    if (cfg->mode == LH_MODE_JOINT_STEREO) {
        for (int gr = 0; gr < ngr; gr++) {
            float   r = L.tot_ener[gr][2] + L.tot_ener[gr][3];
            if (r > 0)
                r = L.tot_ener[gr][3] / r;
            ms_ener_ratio[gr] = lh_uni_f(r);
        }
    }

    /* ---- ATH auto adjustment (reference encoder.c:397) ---- */
    {
        float   factor = lh_lds.ss.ath_adjust_factor, limit = lh_lds.ss.ath_adjust_limit;
        float   loud[2][2];
        loud[0][0] = L.loudness_sq[0][0];
        loud[1][0] = L.loudness_sq[ngr - 1][0];
        /* one channel counts twice (reference encoder.c:72-79) */
        loud[0][1] = (cfg->channels == 2) ? L.loudness_sq[0][1] : loud[0][0];
        loud[1][1] = (cfg->channels == 2) ? L.loudness_sq[ngr - 1][1] : loud[1][0];
        lh_adjust_ATH(T, loud, &factor, &limit);
        LH_SYNC_FRAME();
        if (tid == 0) {
            lh_lds.ss.ath_adjust_factor = factor;
            lh_lds.ss.ath_adjust_limit = limit;
        }
    }
    LH_SYNC_FRAME();

    LH_PA(25, t_frame);
    /* ---- stage 2: polyphase + MDCT (reference encoder.c:405) ---- */
    LH_PT(t_mdct);
#ifndef LH_SPLIT                /* (split pipeline: the spectra are in place since the top of the frame) */
#pragma unroll
    for (int k = 0; k < 9; k++)
        L.u.mdct.sb[w][0][lane + 64 * k] = carry.sb[k];
    lh_polyphase(w);
    LH_SYNC_WG();               /* last read of mf (both channels) before xr overwrites it */
    lh_mdct_granules(w);
#pragma unroll
    for (int k = 0; k < 9; k++)
        carry.sb[k] = L.u.mdct.sb[w][ngr][lane + 64 * k];
    LH_SYNC_WG();
#endif
    LH_PA(2, t_mdct);
#ifndef LH_SPLIT                /* (split pipeline: staged once per launch, nothing overwrites them between frames) */
    lh_stage_loop_tables(c);    /* mf is dead; xr stays */
    LH_SYNC_WG();
#endif

    /* ---- stage 3: M/S decision (reference encoder.c:413-461) ---- */
    int     mode_ext = LH_MPG_MD_LR_LR;
    if (cfg->force_ms)
        mode_ext = LH_MPG_MD_MS_LR;
    else if (cfg->mode == LH_MODE_JOINT_STEREO) {
        float   sum_pe_MS = 0, sum_pe_LR = 0;
        for (int gr = 0; gr < ngr; gr++)
            for (int ch = 0; ch < 2; ch++) {
                sum_pe_MS += L.pe[gr][2 + ch];
                sum_pe_LR += L.pe[gr][ch];
            }
        if (sum_pe_MS <= 1.00 * sum_pe_LR) {
            if (L.block_type[0][0] == L.block_type[0][1] && L.block_type[ngr - 1][0] == L.block_type[ngr - 1][1])
                mode_ext = LH_MPG_MD_MS_LR;
        }
    }
    mode_ext = lh_uni_i(mode_ext);
    int const msoff = (mode_ext == LH_MPG_MD_MS_LR) ? 2 : 0;

    /* ---- PE smoothing FIR (reference encoder.c:489-518) ---- */
    float   pe_use[2][2];
    {
        float   buf[19], f;
        for (int i = 0; i < 18; i++)
            buf[i] = lh_lds.ss.pefirbuf[i + 1];
        f = 0.0;
        pe_use[1][0] = pe_use[1][1] = 0.0f;
        for (int gr = 0; gr < ngr; gr++) {
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
        f = (670 * 5 * ngr * nch) / f;
        for (int gr = 0; gr < ngr; gr++)
            for (int ch = 0; ch < nch; ch++)
                pe_use[gr][ch] = lh_uni_f(pe_use[gr][ch] * f);
        LH_SYNC_FRAME();
        if (tid < 19)
            lh_lds.ss.pefirbuf[tid] = buf[tid];
    }

    LH_PA(26, t_frame);
    /* ---- stage 4: CBR iteration loop (reference quantize.c:1988-2050) ---- */
    int     ResvSize = lh_uni_i(lh_lds.ss.ResvSize), ResvMax, mdb = lh_uni_i(lh_lds.ss.main_data_begin);
    int     substep = lh_uni_i(lh_lds.ss.substep_shaping);
    int     bitrate_index = lh_uni_i(cfg->bitrate_index);
    int     frame_bits = lh_uni_i(lh_frame_bits(cfg, bitrate_index, padding));
    int     mean_bits = lh_uni_i((frame_bits - cfg->sideinfo_len * 8) / LH_NGR);
    int     total_bits = 0;
#ifdef LH_VBR_OLD
    int const vbr_old = (cfg->vbr == 2);
#else
    int const vbr_old = 0;
#endif
    /* (below, vbr_new stands for both VBR loops: either does the whole iteration stage in a function of its own) */
    int const vbr_new = (cfg->vbr == 1 || cfg->vbr == 4 || vbr_old), abr = (cfg->vbr == 3);
    float   masking_lower_left = cfg->masking_lower_long;
    int     abr_targ[2][2] = { {0, 0}, {0, 0} }, analog_silence_bits = 0;
    if (vbr_new) {
        LH_SYNC_WG();
        if (tid == 0)
            for (int gr = 0; gr < 2; gr++)
                for (int ch = 0; ch < 2; ch++) {
                    L.pe_use[gr][ch] = pe_use[gr][ch];
#ifdef LH_DEBUG_DUMP
                    st->dbg_pe[gr][ch] = pe_use[gr][ch];
#endif
                }
        LH_SYNC_WG();
#ifdef LH_VBR_OLD
        if (vbr_old) {
            if (tid == 0) {
                L.ms_ener_ratio[0] = ms_ener_ratio[0];
                L.ms_ener_ratio[1] = ms_ener_ratio[1];
            }
            LH_SYNC_WG();
            lh_vbrold_frame(fo, mode_ext, msoff);
            masking_lower_left = lh_uni_f(L.pe_use[0][0]);
        }
        else
#endif
            lh_vbr_frame(fo, mode_ext, msoff);
        bitrate_index = lh_uni_i(L.frame_bits);
        total_bits = lh_uni_i(L.max_bits);
        ResvSize = lh_uni_i(L.mean_bits);
        substep = lh_uni_i(L.targ_bits[0]);
        frame_bits = lh_frame_bits(cfg, bitrate_index, 0);
        mean_bits = (frame_bits - cfg->sideinfo_len * 8) / LH_NGR;
    }
    if (abr) {
        int const bt[2][2] = { {L.block_type[0][0], L.block_type[0][1]}, {L.block_type[1][0], L.block_type[1][1]} };
        lh_abr_target_bits(cfg, ResvSize, substep, pe_use, ms_ener_ratio, bt, mode_ext, abr_targ, &analog_silence_bits);
    }
    {
        /* ResvFrameBegin */
        int const resvLimit = (8 * 256) * LH_NGR - 8;
        ResvMax = cfg->buffer_constraint - frame_bits;
        if (ResvMax > resvLimit)
            ResvMax = resvLimit;
        if (ResvMax < 0 || cfg->disable_reservoir)
            ResvMax = 0;
        ResvMax = lh_uni_i(ResvMax);
    }
    if (abr)
        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < 2; ch++)
                abr_targ[gr][ch] = lh_uni_i(abr_targ[gr][ch]);
    analog_silence_bits = lh_uni_i(analog_silence_bits);
    if (mode_ext == LH_MPG_MD_MS_LR && !vbr_new) {
        /* mid / side spectra of both granules at once (reference quantize.c:2006-2008 does it granule by granule;
         * nothing in between reads xr): granule 1's are then there for the channel that is done with granule 0
         * first (below) */
        float const k = (float) (LH_SQRT2 * 0.5);
        for (int i = tid; i < ngr * 576; i += LH_NT) {
            int const gr = i >= 576, at = i - 576 * gr;
            float const l = L.xr[0][gr][at];
            float const r = L.xr[1][gr][at];
            L.xr[0][gr][at] = (l + r) * k;
            L.xr[1][gr][at] = (l - r) * k;
        }
    }
    if (ngr == 1) {
        /* the payload's second granule does not exist: all zero (as the checkers leave it) */
        uint32_t *z = (uint32_t *) &fo->gr[1][w];
        for (int i = lane; i < (int) (sizeof(LhGranule) / 4); i += 64)
            z[i] = 0u;
    }
    int     prepared = 0, prepared_nonzero = 0;
    for (int gr = 0; gr < ngr && !vbr_new; gr++) {
        int     targ_bits[2] = { abr_targ[gr][0], abr_targ[gr][1] };
        int     max_bits = 0;
        if (!abr)
            max_bits = lh_on_pe(cfg, ResvSize, ResvMax, &substep, pe_use[gr], targ_bits, mean_bits, gr);
        LH_SYNC_FRAME();
        substep = lh_uni_i(substep);
        if (mode_ext == LH_MPG_MD_MS_LR && !abr)
            lh_reduce_side(targ_bits, ms_ener_ratio[gr], mean_bits, max_bits);
        targ_bits[0] = lh_uni_i(targ_bits[0]);
        targ_bits[1] = lh_uni_i(targ_bits[1]);
#ifdef LH_DEBUG_DUMP
        if (tid == 0) {
            for (int ch = 0; ch < 2; ch++) {
                st->dbg_pe[gr][ch] = pe_use[gr][ch];
                st->dbg_targ[gr][ch] = targ_bits[ch];
            }
            st->dbg_mean_bits = mean_bits;
        }
#endif
        LH_SYNC_FRAME();
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
            int const nonzero = prepared ? prepared_nonzero : lh_prepare_granule(c, ch, gr, msoff, substep);
            R = lh_uniform(L.rg[ch].R);
            g = lh_uniform(L.rg[ch].g);
            if (nonzero) {
                LH_PA(4, t_q);
                LH_PT(t_ol);
                if (abr && !R.ath_over)
                    targ_bits[ch] = analog_silence_bits;    /* reference quantize.c:1953-1954 */
                /* (ten stages of one source: five or four slots x {normal long block, short block} of the usual classes at noise shaping
                 * 2 or 1, or anything: lh_dev_qloop.h) */
                int const usual = lh_uni_i(lh_granule_is_usual(c, R.block_type, R.substep_shaping));
                int const ushort = lh_uni_i(lh_granule_is_usual_short(c, R.block_type, R.substep_shaping));
                if (lq_needs_tail(c, Q, R)) {
                    if (usual == 2)
                        lq_outer_loop_stage5n(ch, gr, targ_bits[ch]);
                    else if (usual == 1)
                        lq_outer_loop_stage5m(ch, gr, targ_bits[ch]);
                    else if (ushort == 2)
                        lq_outer_loop_stage5s(ch, gr, targ_bits[ch]);
                    else if (ushort == 1)
                        lq_outer_loop_stage5t(ch, gr, targ_bits[ch]);
                    else
                        lq_outer_loop_stage5(ch, gr, targ_bits[ch]);
                }
                else {
                    if (usual == 2)
                        lq_outer_loop_stage4n(ch, gr, targ_bits[ch]);
                    else if (usual == 1)
                        lq_outer_loop_stage4m(ch, gr, targ_bits[ch]);
                    else if (ushort == 2)
                        lq_outer_loop_stage4s(ch, gr, targ_bits[ch]);
                    else if (ushort == 1)
                        lq_outer_loop_stage4t(ch, gr, targ_bits[ch]);
                    else
                        lq_outer_loop_stage4(ch, gr, targ_bits[ch]);
                }
                R = lh_uniform(L.rg[ch].R);
                g = lh_uniform(L.rg[ch].g);
                LH_PA(5, t_ol);
            }
            LH_PT(t_fin);
            lh_rg_put(c, R, g);
            if (lh_uni_i(lh_granule_is_usual(c, R.block_type, R.substep_shaping))) {
                lh_best_scalefac_store_n(ch, gr, fo->gr[0][ch].scalefac, L.block_type[0][ch]);
                if (cfg->use_best_huffman == 1)
                    lh_best_huffman_divide_n(ch);
            }
            else {
                lh_best_scalefac_store(ch, gr, fo->gr[0][ch].scalefac, L.block_type[0][ch]);
                if (cfg->use_best_huffman == 1)
                    lh_best_huffman_divide(ch);
            }
            g = lh_uniform(L.rg[ch].g);
            LH_PA(6, t_fin);
            lh_store_granule(c, Q, R, g, xr, o);
            LH_DBG_XR(c, gr, ch, xr);
            if (lh_uni_i(L.ctx.bytes != nullptr))
                lh_emit_part_stage(ch, gr);     /* R / g are in the wave's LDS slot since the last stage call */
            LH_PA(3, t_q);
            if (lane == 0)
                L.bits_used[ch] = g.part2_3_length + g.part2_length;
            if (gr == 0 && nch == 2 && ngr == 2) {
                /* Granule 1's budget needs what BOTH channels spent on granule 0, but its geometry, xrpow and
                 * allowed noise do not: the channel that is done first prepares them while it would otherwise
                 * wait at the barrier for the other (the wait was 8 % of a frame); the one that is done last goes
                 * straight to the barrier.  "First" = the other channel has not yet left its mark for this frame. */
                int const stamp = lh_uni_i(lh_lds.ss.frame_number) + 1;
                int     first;
                LH_WAVE_SYNC();
                /* (lane 0's reading for the whole wave: the other wave may leave its mark at any moment) */
                first = (int) lh_bcast_u32((uint32_t) *(volatile int *) &L.gr0_done[1 - ch], 0) != stamp;
                if (lane == 0)
                    *(volatile int *) &L.gr0_done[ch] = stamp;
                carry.prio_late = !first;
                if (first) {
                    prepared_nonzero = lh_prepare_granule(c, ch, 1, msoff, substep);
                    prepared = 1;
                }
            }
            else if (nch == 2) {
                /* who is last at this barrier (for the issue priority only: see lh_prio_apply) */
                int const stamp = -(lh_uni_i(lh_lds.ss.frame_number) + 1);
                LH_WAVE_SYNC();
                carry.prio_late = (int) lh_bcast_u32((uint32_t) *(volatile int *) &L.gr0_done[1 - ch], 0) == stamp;
                if (lane == 0)
                    *(volatile int *) &L.gr0_done[ch] = stamp;
            }
        }
        LH_SYNC_FRAME();
        LH_PRIO_APPLY(carry.prio_rel, carry.prio_late);
        {
            int const used = lh_uni_i(L.bits_used[0] + L.bits_used[1]);
            ResvSize -= used;
            total_bits += used;
        }
        LH_SYNC_FRAME();
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
        ResvSize += mean_bits * LH_NGR;
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
        lh_lds.ss.slot_lag = slot_lag;
        lh_lds.ss.ResvSize = ResvSize;
        lh_lds.ss.ResvMax = ResvMax;
        lh_lds.ss.main_data_begin = mdb;
        lh_lds.ss.substep_shaping = substep;
        /* what the next frame's psy model finds in sv_qnt.masking_lower: the CBR loop leaves the value
         * of its last granule/channel (channel 0 for mono), the VBR loop always the long-block one (reference quantize.c:1622) */
        lh_lds.ss.masking_lower = vbr_new ? masking_lower_left
            : (L.block_type[ngr - 1][nch - 1] != LH_SHORT_TYPE) ? cfg->masking_lower_long : cfg->masking_lower_short;
        lh_lds.ss.frame_number = lh_lds.ss.frame_number + 1;
        if (mdb * 8 != ResvSize)
            lh_lds.ss.status |= 1;    /* reservoir inconsistency (reference bitstream.c:947) */
    }
    if (lh_uni_i(L.ctx.bytes != nullptr))
        lh_emit_frame(fo, drain_pre, drain_post, frame_bits / 8, mdb_header, bitrate_index, padding, mode_ext,
                      c.d.flush && (int) ((c.frame_base + LH_MF_START) / fs) == c.d.frame_end - 1);
    LH_PA(0, t_frame);
    LH_SYNC_FRAME();
#if defined(LH_PROF) && !defined(LH_EMU)
    if (lane < LH_NPROF)
        st->prof[w][lane] += L.prof[w][lane];
    LH_SYNC_WG();
#endif
}

#ifndef LH_WAVES_PER_EU
#define LH_WAVES_PER_EU 2
#endif
/* (the settings and the tables are never written by a kernel: with noalias the loads of their fields move out of the frame loop
 * and across the payload stores: +1 %) */
#if !defined(LH_EMU) && !defined(LH_NO_RESTRICT)
#define LH_RESTRICT __restrict__
#else
#define LH_RESTRICT
#endif
/* the LH_LSF build of this file (MPEG-2 / 2.5 streams) is a second object in the same library: its own names */
#if defined(LH_LSF) && !defined(LH_SPLIT)
#define lh_encode_kernel lh_encode_kernel_lsf
#define lh_launch_encode lh_launch_encode_lsf
#define lh_emu_encode lh_emu_encode_lsf
#define lh_emu_encode_bytes lh_emu_encode_bytes_lsf
#endif
/* and so is the LH_VBRK build: the same MPEG-1 source compiled with the scheduling strategy that suits the new VBR
 * loop (csrc/Makefile); lh_api.cpp launches it for vbr_mt / vbr_mtrh configurations */
#if defined(LH_VBRK) && !defined(LH_SPLIT)
#define lh_encode_kernel lh_encode_kernel_vbr
#define lh_launch_encode lh_launch_encode_vbr
#endif
/* and the LH_SPLIT builds of all three: the encode kernel of the split pipeline, which starts from the analysis kernels'
 * output (LhMidPools) instead of the PCM */
#ifdef LH_SPLIT
#if defined(LH_LSF)
#define lh_encode_kernel lh_encode_kernel_q_lsf
#define lh_launch_encode lh_launch_encode_q_lsf
#define lh_emu_encode_q lh_emu_encode_q_lsf
#elif defined(LH_VBRK)
#define lh_encode_kernel lh_encode_kernel_q_vbr
#define lh_launch_encode lh_launch_encode_q_vbr
#else
#define lh_encode_kernel lh_encode_kernel_q
#define lh_launch_encode lh_launch_encode_q
#endif
#define LH_MID_PARAM , LhMidPools mid
#define LH_MID_ARG , mid
#else
#define LH_MID_PARAM
#define LH_MID_ARG
#endif
/* all frames of one stream (the workgroup's whole job) */
LH_DEVFN void
lh_encode_stream(const LhConfig * LH_RESTRICT cfg, const LhTables * LH_RESTRICT T, const int16_t * pcm, const float *pcmf,
                 const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, uint8_t * bytes,
                 int nstreams LH_MID_PARAM)
{
    LhLds & L = lh_lds;
    int const sidx = (int) blockIdx.x;
    if (sidx >= nstreams)
        return;
#ifndef LH_VBR_OLD
    /* a build without the old VBR loop (the profiling variant: its cycle counters take that loop's LDS) must not fall
     * into the CBR loop with a vbr_rh configuration: status bit 32, nothing encoded */
    if (cfg->vbr == 2) {
        if (threadIdx.x == 0)
            states[sidx].status |= 32;
        return;
    }
#endif
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
    /* the descriptor is the same for the whole workgroup: scalar registers (see lh_encode_frame) */
    c.d.pcm_l = lh_uni_ll(c.d.pcm_l);
    c.d.pcm_r = lh_uni_ll(c.d.pcm_r);
    c.d.pcm_base = lh_uni_ll(c.d.pcm_base);
    c.d.nsamples = lh_uni_ll(c.d.nsamples);
    c.d.out_index = lh_uni_ll(c.d.out_index);
    c.d.frame_begin = lh_uni_i(c.d.frame_begin);
    c.d.frame_end = lh_uni_i(c.d.frame_end);
    c.d.bytes_base = lh_uni_ll(c.d.bytes_base);
    c.d.bytes_cap = lh_uni_ll(c.d.bytes_cap);
    c.d.flush = lh_uni_i(c.d.flush);
    c.d.mid_rel = lh_uni_i(c.d.mid_rel);
    c.tid = (int) threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = lh_uni_i(c.tid >> 6);      /* scalar: everything indexed by the wave id gets scalar addressing */
    lh_ctx_hot(c);
    if (c.tid == 0) {
        L.gr0_done[0] = L.gr0_done[1] = 0;
        L.ctx.cfg = c.cfg;
        L.ctx.T = c.T;
        L.ctx.st = c.st;
        L.ctx.pcm = c.pcm;
        L.ctx.pcmf = c.pcmf;
        L.ctx.bytes = bytes;
        L.ctx.d = c.d;
    }
    /* State that is rewritten every frame stays on the chip for the whole launch: the polyphase
     * overlap and the psy model's previous partition energies in registers (LhWaveCarry), its band
     * energies / thresholds in the LDS ring (LhLds.psy_en).  HBM sees them once per launch. */
    LhWaveCarry carry;
    carry.prio_rel = 0;
    carry.prio_late = 0;
    LhStreamState *st = c.st;
    for (int p = 0; p < 2; p++) {
        carry.nb.n1[p] = st->nb_l1[c.wave + 2 * p][c.lane];
        carry.nb.n2[p] = st->nb_l2[c.wave + 2 * p][c.lane];
    }
#ifndef LH_SPLIT
#pragma unroll
    for (int k = 0; k < 9; k++)
        carry.sb[k] = st->sb_prev[c.wave][c.lane + 64 * k];
#endif
    for (int t = c.tid; t < 4 * LH_XMIN_N; t += LH_NT) {
        int const chn = t / LH_XMIN_N, i = t - chn * LH_XMIN_N;
        L.psy_en[0][chn][i] = st->en[chn][i];
        L.psy_thm[0][chn][i] = st->thm[chn][i];
    }
    if (c.tid < LH_SS_WORDS_A)
        ((uint32_t *) &L.ss)[c.tid] = ((const uint32_t *) &st->loudness_sq_save[0])[c.tid];
    else if (c.tid < LH_SS_WORDS_A + LH_SS_WORDS_B)
        ((uint32_t *) &L.ss)[c.tid] = ((const uint32_t *) &st->pefirbuf[0])[c.tid - LH_SS_WORDS_A];
#ifdef LH_SPLIT
    lh_stage_loop_tables(c);
#endif
    LH_SYNC_WG();               /* the state words are read by every thread from here on */
    int     slot = 0;           /* ring slot holding the ratios of the frame's first granule */
    constexpr int ngr = LH_NGR, fs = 576 * LH_NGR;      /* granules / samples per frame (1 / 576: MPEG-2, 2.5) */
    int const prio_me = LH_PRIO_INDEX();
    for (int f = c.d.frame_begin; f < c.d.frame_end; f++) {
        carry.prio_rel = LH_PRIO_TICK(prio_me, c.d.frame_end - f);
        LH_PRIO_APPLY(carry.prio_rel, carry.prio_late);
        c.frame_base = (long long) fs * f - LH_MF_START;
        if (c.tid == 0) {
            L.ctx.frame_base = c.frame_base;    /* read by the stages after the next workgroup barrier */
            L.psy_slot = slot;
#ifdef LH_SPLIT
            L.ctx.mid = mid.frames + (c.d.out_index + c.d.mid_rel + (f - c.d.frame_begin));
#endif
        }
#ifdef LH_SPLIT
        LH_SYNC_WG();           /* (the fused kernel's first barrier of a frame follows the window's staging) */
#endif
        lh_encode_frame(c, &out[c.d.out_index + (f - c.d.frame_begin)], carry);
        slot = (slot + ngr) % 3;
    }
    (void) LH_PRIO_TICK(prio_me, 0);
    if (c.tid < LH_SS_WORDS_A)
        ((uint32_t *) &st->loudness_sq_save[0])[c.tid] = ((const uint32_t *) &L.ss)[c.tid];
    else if (c.tid < LH_SS_WORDS_A + LH_SS_WORDS_B)
        ((uint32_t *) &st->pefirbuf[0])[c.tid - LH_SS_WORDS_A] = ((const uint32_t *) &L.ss)[c.tid];
    for (int p = 0; p < 2; p++) {
        st->nb_l1[c.wave + 2 * p][c.lane] = carry.nb.n1[p];
        st->nb_l2[c.wave + 2 * p][c.lane] = carry.nb.n2[p];
    }
#ifndef LH_SPLIT                /* (split pipeline: lh_subband_kernel leaves the stream's polyphase overlap there) */
#pragma unroll
    for (int k = 0; k < 9; k++)
        st->sb_prev[c.wave][c.lane + 64 * k] = carry.sb[k];
#endif
    for (int t = c.tid; t < 4 * LH_XMIN_N; t += LH_NT) {
        int const chn = t / LH_XMIN_N, i = t - chn * LH_XMIN_N;
        st->en[chn][i] = L.psy_en[slot][chn][i];
        st->thm[chn][i] = L.psy_thm[slot][chn][i];
    }
}

#ifndef LH_EMU
extern "C" __global__ void __launch_bounds__(LH_BLOCK, LH_WAVES_PER_EU)
#else
void
#endif
lh_encode_kernel(const LhConfig * LH_RESTRICT cfg, const LhTables * LH_RESTRICT T, const int16_t * pcm, const float *pcmf,
                 const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, uint8_t * bytes,
                 int nstreams LH_MID_PARAM)
{
    lh_encode_stream(cfg, T, pcm, pcmf, descs, states, out, bytes, nstreams LH_MID_ARG);
}

#if !defined(LH_EMU) && !defined(LH_LSF) && !defined(LH_VBRK) && !defined(LH_SPLIT)
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
        {
            /* the four-word transposed sum of count_bits: three words of three 10-bit fields (values < 128 per
             * lane), one of two 16-bit fields */
            unsigned const p0 = (x & 127u) | (((x >> 7) & 127u) << 10) | (((x >> 14) & 127u) << 20);
            unsigned const p1 = ((x >> 3) & 127u) | (((x >> 11) & 127u) << 10) | (((x >> 17) & 127u) << 20);
            unsigned const p2 = ((x >> 5) & 127u) | (((x >> 13) & 127u) << 10) | (((x >> 21) & 127u) << 20);
            unsigned const q = ((x >> 2) & 255u) | (((x >> 9) & 255u) << 16);
            unsigned L = 0, H = 0, want_l = 0, want_h = 0, want_q;
            unsigned const qt = lh_wave_sum_regions(p0, p1, p2, q, &L, &H);
            unsigned const sh = (lane == 0) ? 0u : (lane == 1) ? 3u : 5u;
            unsigned const a = lh_wave_sum_u32((x >> sh) & 127u), b = lh_wave_sum_u32((x >> (sh + (lane == 0 ? 7u : 8u))) & 127u),
                cc = lh_wave_sum_u32((x >> (sh + (lane == 0 ? 14u : lane == 1 ? 14u : 16u))) & 127u);
            (void) a; (void) b; (void) cc;
            {
                /* per-region expectations, formed by every lane for all three regions */
                unsigned const f[3][3] = { {0u, 7u, 14u}, {3u, 11u, 17u}, {5u, 13u, 21u} };
                unsigned tot[3][3];
                for (int r = 0; r < 3; r++)
                    for (int k = 0; k < 3; k++)
                        tot[r][k] = lh_wave_sum_u32((x >> f[r][k]) & 127u);
                want_q = lh_wave_sum_u32((x >> 2) & 255u) | (lh_wave_sum_u32((x >> 9) & 255u) << 16);
                if (lane < 3) {
                    want_l = tot[lane][0] | (tot[lane][1] << 16);
                    want_h = tot[lane][2];
                    bad += (L != want_l) + (H != want_h);
                }
                bad += (qt != want_q);
            }
        }
        bad += (lh_wave_max_u32(x) != rmax);
        {
            /* eight maxima at once: lane k gets that of word k & 7 (word j of lane i: vals[i] rotated by 3 j bits) */
            unsigned w8[8], want = 0;
            for (int j = 0; j < 8; j++)
                w8[j] = (x >> (3 * j)) | (x << (32 - 3 * j) % 32);
            for (int i = 0; i < 64; i++) {
                unsigned const v = vals[i], j = lane & 7u;
                unsigned const r = (v >> (3 * j)) | (v << (32 - 3 * j) % 32);
                want = r > want ? r : want;
            }
            bad += (lh_wave_max8(w8) != want);
            bad += (lh_lane_minus_u32 < 2 > (x) != ((lane & 15u) >= 2 ? vals[lane - 2] : 0u));
            bad += (lh_lane_above_u32(x, x ^ 0x5a5a5a5au) != (lane < 63 ? vals[lane + 1] : (vals[0] ^ 0x5a5a5a5au)));
        }
        {
            /* sums over runs of equal keys inside the rows of 16 lanes (lh_row_seg_scan_addf, lh_dev_vbr.h): runs of 1 .. 40
             * lanes from the round's bits, small integers as floats (exact under any order); checked at the last lane every run
             * has in a row */
            __shared__ unsigned keys[64];
            unsigned const len = 1u + (seed + round * 7u) % 40u;
            unsigned const key = (lane + (round & 7u)) / len;
            float   f[2] = { (float) (x & 1023u), (float) ((x >> 10) & 255u) };
            keys[lane] = key;
            __syncthreads();
            lh_row_seg_scan_addf < 2 > (f, key);
            if ((lane & 15u) == 15u || keys[lane + 1] != key) {
                float   w0 = 0.0f, w1 = 0.0f;
                for (int i = (int) lane; i >= (int) (lane & 48u) && keys[i] == key; i--) {
                    w0 += (float) (vals[i] & 1023u);
                    w1 += (float) ((vals[i] >> 10) & 255u);
                }
                bad += (f[0] != w0) + (f[1] != w1);
            }
        }
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

/* Incremental batches (lamehip_batch_append / _encode_available): the chunks the host staged go to their
 * places in the PCM pool.  The chunks lie back to back in one pinned arena (a chunk = its left samples, then
 * its right samples) that reached HBM with a single copy of the bytes in use; seg[4 k .. 4 k + 3] = where
 * chunk k starts in the arena (samples), its stream, where it goes in the stream's pool rows, its length. */
extern "C" __global__ void __launch_bounds__(256)
lh_scatter_kernel(const int16_t * arena, int16_t * pool, long cap, const int *seg)
{
    int const k = (int) (blockIdx.x >> 1), ch = (int) (blockIdx.x & 1);
    long const from = (long) seg[4 * k], row = 2L * seg[4 * k + 1] + ch;
    int const at = seg[4 * k + 2], n = seg[4 * k + 3];
    for (int i = (int) threadIdx.x; i < n; i += 256)
        pool[row * cap + at + i] = arena[from + (long) ch * n + i];
}

/* Device-packed batches: what the host needs to know of each stream before it copies the bytes -- how many there
 * are (the packer's next header position) and the status word -- as two words per stream instead of the
 * 12 KB of LhStreamState. */
extern "C" __global__ void __launch_bounds__(256)
lh_summary_kernel(const LhStreamState * states, long long *sum, int nstreams)
{
    int const s = (int) (blockIdx.x * 256 + threadIdx.x);
    if (s < nstreams) {
        sum[2 * s] = states[s].em_next_header;
        sum[2 * s + 1] = (long long) states[s].status;
    }
}

extern "C" int
lh_launch_summary(const LhStreamState * states, long long *sum, int nstreams, void *stream)
{
    if (nstreams <= 0)
        return 0;
    hipLaunchKernelGGL(lh_summary_kernel, dim3((unsigned) ((nstreams + 255) / 256)), dim3(256), 0, (hipStream_t) stream, states,
                       sum, nstreams);
    return (int) hipGetLastError();
}

extern "C" int
lh_launch_scatter(const int16_t * arena, int16_t * pool, long cap, const int *seg, int nseg, void *stream)
{
    if (nseg <= 0)
        return 0;
    hipLaunchKernelGGL(lh_scatter_kernel, dim3((unsigned) (2 * nseg)), dim3(256), 0, (hipStream_t) stream, arena, pool, cap,
                       seg);
    return (int) hipGetLastError();
}

#endif
#if defined(LH_TRACE) && !defined(LH_EMU)
/* development aid (tools/trace_profile.py): the search's segment counters since the last call, [wave][cycles | visits][segment] */
extern "C" int
lh_trace_fetch(unsigned long long *dst)
{
    static unsigned long long zero[2][2][LH_NTRACE];
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(lh_trace_buf), sizeof(zero)) != hipSuccess)
        return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(lh_trace_buf), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
#ifndef LH_EMU
/* host-side launcher with a C ABI for lh_api.cpp */
extern "C" int
lh_launch_encode(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf,
                 const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, uint8_t * bytes,
                 int nstreams, void *stream LH_MID_PARAM)
{
    if (nstreams <= 0)
        return 0;
    hipLaunchKernelGGL(lh_encode_kernel, dim3((unsigned) nstreams), dim3(LH_BLOCK), 0,
                       (hipStream_t) stream, cfg, T, pcm, pcmf, descs, states, out, bytes, nstreams LH_MID_ARG);
    return (int) hipGetLastError();
}

#elif defined(LH_SPLIT)
/* (emulator, split pipeline: the encode kernel alone; tests/test_emulator.py runs the analysis kernels' twins first) */
extern "C" int
lh_emu_encode_q(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf,
                const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, uint8_t * bytes, int nstreams,
                const LhMidPools * pools)
{
    hipemu_dim3 grid = { (unsigned) nstreams, 1, 1 }, block = { LH_BLOCK, 1, 1 };
    LhMidPools const mid = *pools;
    hipemu_run(grid, block,[=] () {
               lh_encode_kernel(cfg, T, pcm, pcmf, descs, states, out, bytes, nstreams, mid);
               }
    );
    return 0;
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

#ifndef LH_LSF
extern "C" int lh_emu_encode_bytes_lsf(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const LhStreamDesc * descs,
                                       LhStreamState * states, LhFrameOut * out, uint8_t * bytes, int nstreams);
#endif
extern "C" int
lh_emu_encode_bytes(const LhConfig * cfg, const LhTables * T, const int16_t * pcm,
                    const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, uint8_t * bytes, int nstreams)
{
#ifndef LH_LSF
    if (cfg->mode_gr == 1)      /* an MPEG-2 / 2.5 stream: the other object's kernel (as lh_api.cpp picks the launcher) */
        return lh_emu_encode_bytes_lsf(cfg, T, pcm, descs, states, out, bytes, nstreams);
#endif
    hipemu_dim3 grid = { (unsigned) nstreams, 1, 1 }, block = { LH_BLOCK, 1, 1 };
    hipemu_run(grid, block,[=] () {
               lh_encode_kernel(cfg, T, pcm, (const float *) 0, descs, states, out, bytes, nstreams);
               }
    );
    return 0;
}
#endif
