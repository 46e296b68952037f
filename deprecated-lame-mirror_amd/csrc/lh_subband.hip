/*
 * lh_subband.hip -- polyphase filterbank + MDCT + alias reduction for all frames of a launch (gfx950).
 *
 * Reference newmdct.c:430-1039 (mdct_sub48) needs, besides the PCM, only the granules' block types, which
 * lh_attack_scan_kernel (lh_analysis.hip) has settled for the whole launch by the time this kernel starts.  A workgroup
 * (wave = channel, as in the fused encode kernel, same device functions: lh_dev_mdct.h) takes a run of LH_SB_RUN
 * consecutive frames of one stream: the polyphase output of the granule before the run -- the MDCT overlaps every granule
 * with its predecessor -- is recomputed from the PCM (what the fused kernel does once, on a stream's first frame), inside
 * the run it is carried in registers.  The spectra go to HBM (LhMidXr), where the encode kernel of the split pipeline
 * (lh_kernels.hip, -DLH_SPLIT) picks them up; the stream's overlap after the launch's last frame goes to
 * LhStreamState.sb_prev, so that the fused kernel could take the stream's next launch.
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

#define LH_CUSTOM_LDS "lh_lds_subband.h"
/* LDS bank conflicts were three quarters of this kernel's LDS time (profiles/r05*): the frame window is stored with bit 4 of
 * the sample index flipped in odd blocks of 32 (lh_dev_mdct.h: LH_MF_SWZ), the time slots of the sub-band samples 33 words apart */
#define LH_MF_SWZ(i) ((i) ^ (((i) >> 1) & 16))
#define LH_STAGE_IDX(i) LH_MF_SWZ(i)
#define LH_SB_STRIDE 33
#define LH_ENW_TAPS (lh_lds.enw)
#define LH_MDCT_WIN (lh_lds.mwin)
#define LH_MF_PADDED ((LH_MF_NEEDED + 63) / 64 * 64)
#include "lh_static_tables.h"
#include "lh_dev_common.h"
#if !defined(LH_EMU)
#define LH_KRESTRICT __restrict__
#else
#define LH_KRESTRICT
#endif

#include "lh_dev_mdct.h"

#if defined(LH_APROF) && !defined(LH_EMU)
#define LH_AP_T0() unsigned long long ap_t = clock64()
#define LH_AP(k) do { unsigned long long const n_ = clock64(); if (c.lane == 0) atomicAdd(&c.st->prof[c.wave][(k)], n_ - ap_t); ap_t = n_; } while (0)
#else
#define LH_AP_T0() do { } while (0)
#define LH_AP(k) do { } while (0)
#endif
#define LH_SB_CARRY ((LH_SB_GRANULE + 63) / 64)
#ifndef LH_SB_RUN
#define LH_SB_RUN 16            /* frames per workgroup: one recomputed granule per run */
#endif

#ifdef LH_LSF
#define lh_subband_kernel lh_subband_kernel_lsf
#define lh_launch_subband lh_launch_subband_lsf
#define lh_emu_subband lh_emu_subband_lsf
#endif

#ifndef LH_EMU
extern "C" __global__ void __launch_bounds__(LH_NT, 2)
#else
void
#endif
lh_subband_kernel(const LhConfig * LH_KRESTRICT cfg, const LhTables * LH_KRESTRICT T, const int16_t * LH_KRESTRICT pcm, const float *pcmf, const LhStreamDesc * descs,
                  LhStreamState * states, LhMidFrame * frames, int nstreams)
{
    LhLds & L = lh_lds;
    int const sidx = (int) blockIdx.y;
    constexpr int ngr = LH_NGR, fs = 576 * LH_NGR;
    LhCtx   c;
    c.cfg = cfg;
    c.T = T;
    c.st = &states[sidx];
    c.pcm = pcm;
    c.pcmf = pcmf;
    c.d = descs[sidx];
    c.d.pcm_l = lh_uni_ll(c.d.pcm_l);
    c.d.pcm_r = lh_uni_ll(c.d.pcm_r);
    c.d.pcm_base = lh_uni_ll(c.d.pcm_base);
    c.d.nsamples = lh_uni_ll(c.d.nsamples);
    c.d.out_index = lh_uni_ll(c.d.out_index);
    c.d.mid_rel = lh_uni_i(c.d.mid_rel);
    c.d.frame_begin = lh_uni_i(c.d.frame_begin);
    c.d.frame_end = lh_uni_i(c.d.frame_end);
    c.tid = (int) threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = lh_uni_i(c.tid >> 6);
    lh_ctx_hot(c);
    int const f0 = c.d.frame_begin + LH_SB_RUN * (int) blockIdx.x;
    int const f1 = (f0 + LH_SB_RUN < c.d.frame_end) ? f0 + LH_SB_RUN : c.d.frame_end;
    if (f0 >= c.d.frame_end)
        return;
    int const w = c.wave, lane = c.lane, tid = c.tid;
    if (tid == 0) {
        L.ctx.cfg = c.cfg;
        L.ctx.T = c.T;
        L.ctx.st = c.st;
        L.ctx.pcm = c.pcm;
        L.ctx.pcmf = c.pcmf;
        L.ctx.bytes = nullptr;
        L.ctx.d = c.d;
        L.ctx.frame_base = 0;
    }
    float   sb[LH_SB_CARRY];
    LH_AP_T0();
    for (int i = tid; i < 288; i += LH_NT)
        L.enw[i] = lh_enwindow[i < 285 ? i : 284];
    for (int i = tid; i < 4 * 36; i += LH_NT)
        L.mwin[i] = lh_mdct_win[i];
    /* the granule before the run (reference encoder.c:189-236 primes the filterbank the same way on a stream's first frame) */
    lh_stage_span < LH_MF_NEEDED, LH_NT > (c, L.mf[0], L.mf[1], (long long) fs * f0 - LH_MF_START - fs);
    LH_SYNC_WG_LDS();
    lh_polyphase(w);
#pragma unroll
    for (int k = 0; k < LH_SB_CARRY; k++)
        sb[k] = L.u.mdct.sb[w][ngr][(lane + 64 * k < LH_SB_GRANULE) ? lane + 64 * k : 0];
    LH_SYNC_WG_LDS();
    LH_AP(8);
    for (int f = f0; f < f1; f++) {
        long long const at = c.d.out_index + c.d.mid_rel + (f - c.d.frame_begin);
        lh_stage_span < LH_MF_NEEDED, LH_NT > (c, L.mf[0], L.mf[1], (long long) fs * f - LH_MF_START);
        if (tid < 4) {
            /* (a one-granule frame: the transforms also run over the window's second granule, which is thrown away) */
            int const gr = tid >> 1, ch = tid & 1;
            L.block_type[gr][ch] = (gr < ngr) ? (int) frames[at].small.gr[gr].block_type[ch] : LH_NORM_TYPE;
        }
#pragma unroll
        for (int k = 0; k < LH_SB_CARRY; k++)
            if (lane + 64 * k < LH_SB_GRANULE)
                L.u.mdct.sb[w][0][lane + 64 * k] = sb[k];
        LH_SYNC_WG_LDS();
        LH_AP(9);
        lh_polyphase(w);
        LH_SYNC_WG_LDS();           /* last read of mf (both channels) before xr overwrites it */
        LH_AP(10);
        lh_mdct_granules(w);
#pragma unroll
        for (int k = 0; k < LH_SB_CARRY; k++)
            sb[k] = L.u.mdct.sb[w][ngr][(lane + 64 * k < LH_SB_GRANULE) ? lane + 64 * k : 0];
        LH_SYNC_WG_LDS();
        LH_AP(11);
        {
            lh_f32x4 *dst = (lh_f32x4 *) frames[at].xr.xr;
            const lh_f32x4 *src = (const lh_f32x4 *) L.xr;
            for (int i = tid; i < 576; i += LH_NT)
                dst[i] = src[i];
        }
        LH_SYNC_WG_LDS();
        LH_AP(12);
    }
    if (f1 == c.d.frame_end) {
        /* (LhStreamState keeps the plain layout: slot * 32 + band) */
#pragma unroll
        for (int k = 0; k < LH_SB_CARRY; k++) {
            int const i = lane + 64 * k, slot = i / LH_SB_STRIDE, col = i - slot * LH_SB_STRIDE;
            if (i < LH_SB_GRANULE && col < 32)
                c.st->sb_prev[w][slot * 32 + col] = sb[k];
        }
    }
}

#ifndef LH_EMU
extern "C" int
lh_launch_subband(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf, const LhStreamDesc * descs,
                  LhStreamState * states, LhMidPools mid, int nstreams, int max_frames, void *stream)
{
    if (nstreams <= 0 || max_frames <= 0)
        return 0;
    hipLaunchKernelGGL(lh_subband_kernel, dim3((unsigned) ((max_frames + LH_SB_RUN - 1) / LH_SB_RUN), (unsigned) nstreams),
                       dim3(LH_NT), 0, (hipStream_t) stream, cfg, T, pcm, pcmf, descs, states, mid.frames, nstreams);
    return (int) hipGetLastError();
}
#else
extern "C" int
lh_emu_subband(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf, const LhStreamDesc * descs,
               LhStreamState * states, const LhMidPools * pools, int nstreams, int max_frames)
{
    LhMidPools const mid = *pools;
    hipemu_dim3 grid = { (unsigned) ((max_frames + LH_SB_RUN - 1) / LH_SB_RUN), (unsigned) nstreams, 1 }, block = { LH_NT, 1, 1 };
    hipemu_run(grid, block,[=] () {
               lh_subband_kernel(cfg, T, pcm, pcmf, descs, states, mid.frames, nstreams);
               }
    );
    return 0;
}
#endif
