/*
 * lh_analysis.hip -- the frame-parallel part of the psycho-acoustic model (gfx950).
 *
 * The encode kernel (lh_kernels.hip) walks a stream's frames in order because the bit reservoir chains them together.
 * Most of the psycho-acoustic model (reference psymodel.c:1397-1600) is not part of that chain: the high-pass filter and
 * sub-block peaks of the attack detection (:759-830), the windowed FHTs (fft.c:193-289), the power spectra and their
 * sums (:655-737, :213-226), and the partition energies, tonality and spreading convolution (:1031-1262 up to the
 * pre-echo clamp) depend on the PCM alone.  The kernels of this file compute them for ALL frames of a launch at once --
 * one workgroup per (stream, granule), many workgroups per CU instead of the encode kernel's two waves per SIMD -- and
 * park the results in HBM (lh_device.h: LhMidSmall / LhMidLong / LhMidShort); the encode kernel of the split pipeline
 * (-DLH_SPLIT) starts from there and keeps only the recurrences.
 *
 *   lh_attack_kernel        (stream, granule): high-pass FIR, the nine sub-block peaks of L, R, M, S
 *   lh_attack_scan_kernel   (stream): the attack verdicts, long / short decision and block types -- the one short
 *                           recurrence (last_attacks, blocktype_old) that the transforms below depend on
 *   lh_analysis_kernel      (stream, granule): FHT-1024, spectra, sums, long masking; for a granule with a short-block
 *                           channel also the three FHT-256 and their masking
 *
 * The arithmetic is the fused kernel's (same device functions, same evaluation order): the results are bit-identical.
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

#if defined(LH_APROF) && !defined(LH_EMU)
/* the shared device functions' own marks (lh_compute_masking: slots 14 .. 16) land 20 slots up in the stream's accumulators */
__shared__ unsigned long long *lh_ap_base;
#define LH_PT(var) unsigned long long var = clock64()
#define LH_PA(idx, var) do { if (c.lane == 0) atomicAdd(&lh_ap_base[LH_NPROF * c.wave + 20 + (idx)], clock64() - var); } while (0)
#define LH_PC(idx) do { } while (0)
#endif
#define LH_CUSTOM_LDS "lh_lds_analysis.h"
#define LH_CUSTOM_SMP
/* The transform buffers are stored bank-swizzled: word i of a channel's 1024 at i ^ X(i), X = bit 5 -> bit 1, bit 6 -> bit 3,
 * bit 7 -> bits 2 and 4.  The radix-4 passes walk them at strides of 8, 16 and 64 words and the first pass in bit-reversed
 * order; in the plain layout a wave's 64 addresses fell on 4 or 8 of the 32 banks, and with twelve workgroups per CU the
 * transforms were what the LDS pipe spent most of its time on (profiles/r05_*: three quarters of the kernel's LDS cycles were
 * bank conflicts).  Found by a search over linear swizzles of the passes' address patterns: 1280 -> 416 bank cycles per
 * transform (320 = no conflict at all).  Bit 0 is left alone (samples are staged in pairs), bits 8, 9 do not take part
 * (offsets of 256 words commute with it). */
#define LH_FZ_X(i) ((((i) >> 4) & 2) | (((i) >> 3) & 24) | (((i) >> 5) & 4))
#define LH_FZ(i) ((i) ^ LH_FZ_X(i))
#define LH_STAGE_IDX(i) LH_FZ(i)
#include "lh_static_tables.h"
#include "lh_dev_common.h"
#if !defined(LH_EMU)
#define LH_KRESTRICT __restrict__
#else
#define LH_KRESTRICT
#endif

/* The 1024 samples a granule's transforms read (from bufp = frame window + 576 gr + 304 on, reference psymodel.c:1420) are
 * staged as scaled floats in the work area, channel ch at work[1024 ch ..], and transformed in place; sample i of that span */
#define LH_SPAN (lh_lds.work)
LH_DEVFN float
lh_smp(const LhCtx & c, int ch, int i)
{
    return LH_SPAN[ch * LH_BLKSIZE + LH_FZ(i)];
}

#include "lh_dev_psy_core.h"

/* which granule of which frame a workgroup of the (granule, stream) grid works on */
struct LhGranuleAt {
    int     live;
    int     gr;                 /* granule of the frame */
    long long at;               /* the frame's record in the pools */
    long long frame_base;       /* stream sample index of the frame window's first sample (1152 f - 528) */
};

LH_DEVFN LhGranuleAt
lh_granule_at(const LhStreamDesc & d, int g)
{
    LhGranuleAt o;
    int const nf = d.frame_end - d.frame_begin;
    int const fr = g / LH_NGR;
    o.live = g < LH_NGR * nf;
    o.gr = g - fr * LH_NGR;
    o.at = d.out_index + d.mid_rel + fr;
    o.frame_base = (long long) (576 * LH_NGR) * (d.frame_begin + fr) - LH_MF_START;
    return o;
}

LH_DEVFN LhStreamDesc
lh_desc_uniform(const LhStreamDesc * descs, int sidx)
{
    LhStreamDesc d = descs[sidx];
    d.pcm_l = lh_uni_ll(d.pcm_l);
    d.pcm_r = lh_uni_ll(d.pcm_r);
    d.pcm_base = lh_uni_ll(d.pcm_base);
    d.nsamples = lh_uni_ll(d.nsamples);
    d.out_index = lh_uni_ll(d.out_index);
    d.mid_rel = lh_uni_i(d.mid_rel);
    d.frame_begin = lh_uni_i(d.frame_begin);
    d.frame_end = lh_uni_i(d.frame_end);
    return d;
}

#ifdef LH_LSF
#define lh_attack_kernel lh_attack_kernel_lsf
#define lh_attack_scan_kernel lh_attack_scan_kernel_lsf
#define lh_analysis_kernel lh_analysis_kernel_lsf
#define lh_launch_analysis lh_launch_analysis_lsf
#define lh_emu_analysis lh_emu_analysis_lsf
#endif

/* development aid (-DLH_APROF, tools/an_profile.py): cycles per phase, added up per stream in LhStreamState.prof[wave][k] */
#if defined(LH_APROF) && !defined(LH_EMU)
#define LH_AP_T0() unsigned long long ap_t = clock64()
#define LH_AP(k) do { unsigned long long const n_ = clock64(); if (c.lane == 0) atomicAdd(&aprof[64 * 0 + LH_NPROF * c.wave + (k)], n_ - ap_t); ap_t = n_; } while (0)
#else
#define LH_AP_T0() do { } while (0)
#define LH_AP(k) do { } while (0)
#endif
#define LH_FIR_SPAN 640         /* 576 + 21 samples are read; staged in pairs by 64 threads */

/* ---- attack detection, part 1: the high-passed granule and its sub-block peaks (reference psymodel.c:759-830) ---- */
#ifndef LH_EMU
extern "C" __global__ void __launch_bounds__(64)
#else
void
#endif
lh_attack_kernel(const LhConfig * LH_KRESTRICT cfg, const int16_t * LH_KRESTRICT pcm, const float *pcmf, const LhStreamDesc * descs, LhMidFrame * frames,
                 int nstreams)
{
    /* [2][LH_FIR_SPAN] samples, then (in place) [2][576] filtered ones: 5 KB of its own, not the analysis kernel's image --
     * one wave per workgroup, and what limits the waves per CU is the LDS a workgroup takes */
    __shared__ float span[2 * LH_FIR_SPAN];
    int const sidx = (int) blockIdx.y;
    LhCtx   c;
    c.cfg = cfg;
    c.T = nullptr;
    c.st = nullptr;
    c.pcm = pcm;
    c.pcmf = pcmf;
    c.d = lh_desc_uniform(descs, sidx);
    c.tid = (int) threadIdx.x;
    c.lane = c.tid;
    c.wave = 0;
    LhGranuleAt const ga = lh_granule_at(c.d, (int) blockIdx.x);
    if (!ga.live)
        return;
    int const lane = c.lane;
    int const n_chn_psy = (cfg->mode == LH_MODE_JOINT_STEREO) ? 4 : cfg->channels;
    int const bufbase = 576 + ga.gr * 576 - LH_FFTOFFSET;
    int const firbase = bufbase + 576 - 350 - LH_NSFIRLEN + 192;
    LhMidGr *mg = &frames[ga.at].small.gr[ga.gr];
    lh_stage_span < LH_FIR_SPAN, 64, 1 > (c, span, span + LH_FIR_SPAN, ga.frame_base + firbase);
    LH_WAVE_SYNC();
    {
        /* a lane filters nine consecutive samples of either channel (as the fused kernel does: lh_dev_psy.h); every lane
         * has its 30 samples in registers before the first filtered one replaces a sample */
        int const i0 = 9 * lane;
        float   x[2][30], y[2][9];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int t = 0; t < 30; t++)
                x[ch][t] = span[ch * LH_FIR_SPAN + i0 + t];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int m = 0; m < 9; m++) {
                float   sum1 = x[ch][m + 10], sum2 = 0.0;
#pragma unroll
                for (int j = 0; j < ((LH_NSFIRLEN - 1) / 2) - 1; j += 2) {
                    sum1 += lh_hp_fir[j] * (x[ch][m + j] + x[ch][m + LH_NSFIRLEN - j]);
                    sum2 += lh_hp_fir[j + 1] * (x[ch][m + j + 1] + x[ch][m + LH_NSFIRLEN - j - 1]);
                }
                y[ch][m] = sum1 + sum2;
            }
        LH_WAVE_SYNC();
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int m = 0; m < 9; m++)
                span[ch * LH_FIR_SPAN + i0 + m] = y[ch][m];
    }
    LH_WAVE_SYNC();
    float   fl[9], fr[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        fl[k] = span[lane + 64 * k];
        fr[k] = span[LH_FIR_SPAN + lane + 64 * k];
    }
    for (int chn = 0; chn < n_chn_psy; chn++) {
        /* peak k = the largest magnitude among samples 64 k .. 64 k + 63 (exact under any order) */
        uint32_t pk[8], pk8 = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            float const l = fl[k], r = fr[k];
            float const v = (chn == 0) ? l : (chn == 1) ? r : (chn == 2) ? l + r : l - r;
            if (k < 8)
                pk[k] = lh_f32_as_u32(lh_fabsf(v));
            else
                pk8 = lh_f32_as_u32(lh_fabsf(v));
        }
        {
            uint32_t const m8 = lh_wave_max8(pk);       /* lane k: peak k & 7 */
            uint32_t const m9 = lh_wave_max_u32(pk8);
            if (lane < 9)
                mg->peak[chn][lane] = lh_u32_as_f32(lane == 8 ? m9 : m8);
        }
    }
}

/* ---- attack detection, part 2: the verdicts (reference psymodel.c:806-933, 1265-1319); one wave per stream ----
 * The only part of the analysis that is a recurrence over a stream's granules (last_attacks, blocktype_old), and it is short:
 * the four pseudo-channels sit side by side in the wave -- lane 16 chn + i (i < 12) owns sub-block i of channel chn, every
 * lane of a group of 16 forms its channel's verdicts (the same in all of them) --, and the next granule's peaks are on their
 * way while this one's verdicts are formed. */
#ifndef LH_SCAN_BLOCK
#define LH_SCAN_BLOCK 4
#endif
#ifndef LH_EMU
extern "C" __global__ void __launch_bounds__(64)
#else
void
#endif
lh_attack_scan_kernel(const LhConfig * LH_KRESTRICT cfg, const LhTables * LH_KRESTRICT T, const LhStreamDesc * descs, const LhStreamState * states,
                      LhMidFrame * frames, int nstreams)
{
    int const sidx = (int) blockIdx.x;
    int const lane = (int) threadIdx.x;
    LhStreamDesc const d = lh_desc_uniform(descs, sidx);
    int const nf = d.frame_end - d.frame_begin;
    if (nf <= 0)
        return;
    const LhStreamState *st = &states[sidx];
    int const n_chn_psy = lh_uni_i((cfg->mode == LH_MODE_JOINT_STEREO) ? 4 : cfg->channels);
    int const channels = lh_uni_i(cfg->channels), short_blocks = lh_uni_i(cfg->short_blocks);
    int const chn = lane >> 4, i = lane & 15, grp = lane & 48;
    int const mine = chn < n_chn_psy;
    /* what the previous launch left: lane 16 chn + k (k < 9) holds last_en_subshort[chn][k]; the scalars in every lane of
     * their channel's group */
    float   le = st->last_en_subshort[chn][i < 9 ? i : 0];
    int     last_att = st->last_attacks[chn];
    float const thresh = T->attack_threshold[chn];
    int     bt_old0 = lh_uni_i(st->blocktype_old[0]), bt_old1 = lh_uni_i(st->blocktype_old[1]);
    /* Only the last few steps of a granule's verdicts depend on the granule before (last_attacks, the block types): everything
     * up to there -- the ratios, the votes, the 1.7 rule -- is formed for LH_SCAN_BLOCK granules side by side, so that the
     * cross-lane latencies of one granule hide behind the others', and the short dependent tails follow one another. */
    int const ng = LH_NGR * nf;
    float   pk_blk[LH_SCAN_BLOCK];
#pragma unroll
    for (int j = 0; j < LH_SCAN_BLOCK; j++) {
        int const g = (j < ng) ? j : 0;
        pk_blk[j] = frames[d.out_index + d.mid_rel + g / LH_NGR].small.gr[g % LH_NGR].peak[chn][i < 9 ? i : 0];
    }
    for (int g0 = 0; g0 < ng; g0 += LH_SCAN_BLOCK) {
        float   pk_cur[LH_SCAN_BLOCK], ssf_blk[LH_SCAN_BLOCK];
        int     nsa_blk[LH_SCAN_BLOCK][4];
#pragma unroll
        for (int j = 0; j < LH_SCAN_BLOCK; j++) {
            int const gn = g0 + LH_SCAN_BLOCK + j, g = (gn < ng) ? gn : 0;
            pk_cur[j] = pk_blk[j];
            pk_blk[j] = frames[d.out_index + d.mid_rel + g / LH_NGR].small.gr[g % LH_NGR].peak[chn][i < 9 ? i : 0];
        }
#pragma unroll
        for (int j = 0; j < LH_SCAN_BLOCK; j++) {
            float const pk = pk_cur[j];
            /* twelve sub-blocks per channel: three from the previous granule, nine new ones */
            float const fresh = pk < 1.0f ? 1.0f : pk;  /* lane k < 9 of the group: the new sub-block k */
            float const old = lh_shfl_f32(le, grp + ((i + 6) & 15)), older = lh_shfl_f32(le, grp + ((i + 4) & 15));
            float const got = lh_shfl_f32(fresh, grp + ((i - 3) & 15));
            float const e = (i < 3) ? old : got;
            float const e1 = lh_shfl_f32(e, grp + ((i - 1) & 15)), e0 = lh_shfl_f32(e, grp + ((i - 2) & 15));
            float const then = (i < 3) ? older : e0;
            float const ai = (i < 3) ? e / then : (e > then) ? e / then : (then > e * 10.0f) ? then / (e * 10.0f) : 0.0f;
            float const whole = e0 + e1 + e;
            int const tail_low = e * 6 < whole, mid_low = e1 * 6 < whole;
            uint64_t const over_all = lh_ballot(i < 12 && ai > thresh);
            unsigned const over = (unsigned) (over_all >> (16 * chn)) & 0xfffu;        /* this channel's twelve bits */
            float   en_short[4];
            en_short[0] = lh_shfl_f32(whole, grp + 2);
            en_short[1] = lh_shfl_f32(whole, grp + 5);
            en_short[2] = lh_shfl_f32(whole, grp + 8);
            en_short[3] = lh_shfl_f32(whole, grp + 11);
            ssf_blk[j] = tail_low ? (mid_low ? 0.25f : 0.5f) : 1.0f;   /* lanes 5, 8, 11 of a group: sub_short_factor[0..2] */
            int    *nsa = nsa_blk[j];
            le = fresh;
            for (int gq = 0; gq < 4; gq++) {
                /* the first sub-block of the short block whose ratio exceeds the threshold, 1-based */
                unsigned const bits = (over >> (3 * gq)) & 7u;
                nsa[gq] = bits ? ((bits & 1u) ? 1 : (bits & 2u) ? 2 : 3) : 0;
            }
            for (int q = 1; q < 4; q++) {
                float const u = en_short[q - 1];
                float const v = en_short[q];
                float const m = (u > v) ? u : v;
                if (m < 40000) {
                    if (u < 1.7f * v && v < 1.7f * u) {
                        if (q == 1 && nsa[0] <= nsa[q])
                            nsa[0] = 0;
                        nsa[q] = 0;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < LH_SCAN_BLOCK; j++) {
            int const g = g0 + j;
            if (g >= ng)
                break;
            int const fr = g / LH_NGR, gr = g - fr * LH_NGR;
            LhMidGr *mg = &frames[d.out_index + d.mid_rel + fr].small.gr[gr];
            int    *nsa = nsa_blk[j];
            float const ssf = ssf_blk[j];
            int     uselong = 1;
            if (nsa[0] <= last_att)
                nsa[0] = 0;
            if (last_att == 3 || nsa[0] + nsa[1] + nsa[2] + nsa[3]) {
                uselong = 0;
                if (nsa[1] && nsa[0])
                    nsa[1] = 0;
                if (nsa[2] && nsa[1])
                    nsa[2] = 0;
                if (nsa[3] && nsa[2])
                    nsa[3] = 0;
            }
            if (!mine) {
                nsa[0] = nsa[1] = nsa[2] = nsa[3] = 0;
                uselong = 1;
            }
            else
                last_att = nsa[2];
            if (i == 5 || i == 8 || i == 11)
                mg->sub_short_factor[chn][(i - 5) / 3] = mine ? ssf : 1.0f;
            if (i < 4)
                mg->ns_attacks[chn][i] = (int8_t) (i == 0 ? nsa[0] : i == 1 ? nsa[1] : i == 2 ? nsa[2] : nsa[3]);
            {
                /* uselongblock[] (reference psymodel.c:926-933, 1265-1286) and the block types (:1289-1319): wave-uniform */
                uint64_t const shorts = lh_ballot(!uselong);
                int const s0 = (int) (shorts & 1u), s1 = (int) ((shorts >> 16) & 1u), s23 = (int) ((shorts >> 32) & 0x10001u) != 0;
                int     ul0 = !s0, ul1 = (channels == 2) ? !s1 : 1;
                int     btd[2];
                if (s23)
                    ul0 = ul1 = 0;
                if (short_blocks == 1 && !(ul0 && ul1))
                    ul0 = ul1 = 0;
                if (short_blocks == 2)
                    ul0 = ul1 = 1;
                if (short_blocks == 3)
                    ul0 = ul1 = 0;
                for (int ch = 0; ch < 2; ch++) {
                    int     blocktype = LH_NORM_TYPE;
                    int     was = ch ? bt_old1 : bt_old0;
                    if (ch ? ul1 : ul0) {
                        if (was == LH_SHORT_TYPE)
                            blocktype = LH_STOP_TYPE;
                    }
                    else {
                        blocktype = LH_SHORT_TYPE;
                        if (was == LH_NORM_TYPE)
                            was = LH_START_TYPE;
                        if (was == LH_STOP_TYPE)
                            was = LH_SHORT_TYPE;
                    }
                    btd[ch] = was;
                    if (ch)
                        bt_old1 = lh_uni_i(blocktype);
                    else
                        bt_old0 = lh_uni_i(blocktype);
                }
                if (lane < 2) {
                    mg->uselong[lane] = (int8_t) (lane ? ul1 : ul0);
                    mg->block_type[lane] = (int8_t) (lane ? btd[1] : btd[0]);
                }
            }
        }
    }
}

/* One radix-4 butterfly unit (lh_fht_unit, lh_dev_psy_core.h) on a swizzled buffer.  The eight places of a unit are
 * lo + j K1 and hi + j K1; which bits above bit 4 an offset changes -- and with them the swizzle -- is known per pass. */
template < int K1 > LH_DEVFN void
lh_fht_unit_fz(lh_f32x4 tw, float *fz, int u)
{
    constexpr int kx = K1 >> 1, k2 = K1 << 1, k3 = k2 + K1, k4 = k2 << 1;
    int const blk = u / kx, i = u - blk * kx;
    int const axis = (i == 0);
    int const lo = blk * k4 + i;
    int const hi = blk * k4 + (axis ? kx : K1 - i);
    int     al[4], ah[4];
    if (K1 == 4) {
        /* a unit stays inside 16 words: one swizzle for all eight */
        int const x = LH_FZ_X(lo);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            al[j] = (lo + j * K1) ^ x;
            ah[j] = (hi + j * K1) ^ x;
        }
    }
    else if (K1 == 16) {
        /* inside 64 words: the upper two of either quadruple lie beyond bit 5 */
        int const x = LH_FZ_X(blk * k4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            al[j] = (lo + j * K1) ^ x ^ (j >= 2 ? 2 : 0);
            ah[j] = (hi + j * K1) ^ x ^ (j >= 2 ? 2 : 0);
        }
    }
    else if (K1 == 64) {
        /* lo below word 32 of its 256, hi in 32 .. 63: bits 6 and 7 are the offset's */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            al[j] = (lo + j * K1) ^ LH_FZ_X(64 * j);
            ah[j] = (hi + j * K1) ^ (2 ^ LH_FZ_X(64 * j));
        }
    }
    else {
        int const xl = LH_FZ_X(lo), xh = LH_FZ_X(hi);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            al[j] = (lo + j * K1) ^ xl;
            ah[j] = (hi + j * K1) ^ xh;
        }
    }
    float const p0 = fz[al[0]], p1 = fz[al[1]], p2 = fz[al[2]], p3 = fz[al[3]];
    float const q0 = fz[ah[0]], q1 = fz[ah[1]], q2 = fz[ah[2]], q3 = fz[ah[3]];
    /* on the axes: no rotation, the mirrored quarter only scales by sqrt 2 */
    float const s01 = p0 + p1, d01 = p0 - p1, s23 = p2 + p3, d23 = p2 - p3;
    float const r2 = (float) (LH_SQRT2 * q2), r3 = (float) (LH_SQRT2 * q3);
    float const t01 = q0 + q1, u01 = q0 - q1;
    float const a_l0 = s01 + s23, a_l1 = d01 + d23, a_l2 = s01 - s23, a_l3 = d01 - d23;
    float const a_h0 = t01 + r2, a_h1 = u01 + r3, a_h2 = t01 - r2, a_h3 = u01 - r3;
    /* off the axes: the second and fourth quarters turn by the double angle, then the two half-sums by the single angle */
    LhRot const rq1 = lh_rot(tw.z, tw.w, p1, q1);
    LhRot const rq3 = lh_rot(tw.z, tw.w, p3, q3);
    float const le = p0 + rq1.along, lm = p0 - rq1.along;
    float const he = q0 + rq1.across, hm = q0 - rq1.across;
    float const l2e = p2 + rq3.along, l2m = p2 - rq3.along;
    float const h2e = q2 + rq3.across, h2m = q2 - rq3.across;
    LhRot const ra = lh_rot(tw.x, tw.y, l2e, h2m);
    LhRot const rb = lh_rot(tw.y, tw.x, h2e, l2m);
    fz[al[0]] = axis ? a_l0 : le + ra.along;
    fz[al[1]] = axis ? a_l1 : lm + rb.across;
    fz[al[2]] = axis ? a_l2 : le - ra.along;
    fz[al[3]] = axis ? a_l3 : lm - rb.across;
    fz[ah[0]] = axis ? a_h0 : he + rb.along;
    fz[ah[1]] = axis ? a_h1 : hm + ra.across;
    fz[ah[2]] = axis ? a_h2 : he - rb.along;
    fz[ah[3]] = axis ? a_h3 : hm - ra.across;
}

/* The windowed long transform (reference fft.c:245-289; lh_fft_long in lh_dev_psy_core.h) IN PLACE on a swizzled buffer: a
 * lane reads the sixteen samples of its two trips before the first of its sums replaces a sample.  One wave, its own
 * channel's span x[1024]. */
LH_DEVFN void
lh_fft_long_inplace(const LhCtx & c, float *x)
{
    const float *w = c.T->fft_window;
    int const lane = c.lane;
    float   o[2][8];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int const jj = lane + 64 * q;
        int const i = (int) lh_rev8((unsigned) jj);
        const float *xs = x + (i ^ LH_FZ_X(i));         /* (the eight offsets touch bits 0, 8 and 9 only) */
        float   f0, f1, f2, f3, ww;
        f0 = w[i] * xs[0];
        ww = w[i + 0x200] * xs[0x200];
        f1 = f0 - ww;
        f0 = f0 + ww;
        f2 = w[i + 0x100] * xs[0x100];
        ww = w[i + 0x300] * xs[0x300];
        f3 = f2 - ww;
        f2 = f2 + ww;
        o[q][0] = f0 + f2;
        o[q][2] = f0 - f2;
        o[q][1] = f1 + f3;
        o[q][3] = f1 - f3;
        f0 = w[i + 0x001] * xs[0x001];
        ww = w[i + 0x201] * xs[0x201];
        f1 = f0 - ww;
        f0 = f0 + ww;
        f2 = w[i + 0x101] * xs[0x101];
        ww = w[i + 0x301] * xs[0x301];
        f3 = f2 - ww;
        f2 = f2 + ww;
        o[q][4] = f0 + f2;
        o[q][6] = f0 - f2;
        o[q][5] = f1 + f3;
        o[q][7] = f1 - f3;
    }
    {
        lh_f32x4 tw[4][2];
#pragma unroll
        for (int stage = 0; stage < 4; stage++) {
            int const kx = 2 << (2 * stage);
#pragma unroll
            for (int q = 0; q < 2; q++)
                tw[stage][q] = *(const lh_f32x4 *) c.T->fht_tw[stage][(lane + 64 * q) % kx];
        }
        LH_WAVE_SYNC_MEM();     /* every lane has read its samples */
#pragma unroll
        for (int q = 0; q < 2; q++) {
            int const at = 4 * (lane + 64 * q), xa = LH_FZ_X(at);   /* (the eight offsets touch bits 0, 1 and 9 only) */
#pragma unroll
            for (int k = 0; k < 4; k++) {
                x[(at + k) ^ xa] = o[q][k];
                x[(at + k + LH_BLKSIZE / 2) ^ xa] = o[q][4 + k];
            }
        }
        LH_WAVE_SYNC_MEM();
#pragma unroll
        for (int q = 0; q < 2; q++)
            lh_fht_unit_fz < 4 > (tw[0][q], x, lane + 64 * q);
        LH_WAVE_SYNC_MEM();
#pragma unroll
        for (int q = 0; q < 2; q++)
            lh_fht_unit_fz < 16 > (tw[1][q], x, lane + 64 * q);
        LH_WAVE_SYNC_MEM();
#pragma unroll
        for (int q = 0; q < 2; q++)
            lh_fht_unit_fz < 64 > (tw[2][q], x, lane + 64 * q);
        LH_WAVE_SYNC_MEM();
#pragma unroll
        for (int q = 0; q < 2; q++)
            lh_fht_unit_fz < 256 > (tw[3][q], x, lane + 64 * q);
        LH_WAVE_SYNC_MEM();
    }
}

/* the same for the three short transforms (reference fft.c:193-243; lh_fft_short): x[1024] holds the span and receives the
 * three transforms at x[0 / 256 / 512 ..] */
LH_DEVFN void
lh_fft_short_inplace(const LhCtx & c, float *x)
{
    const float *ws = c.T->fft_window_s;
    int const lane = c.lane;
    float   o[2][8];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int const t = lane + 64 * q, tt = t < 3 * (LH_BLKSIZE_S / 8) ? t : 0;
        int const b = tt >> 5, j = tt & 31;
        int const k = (576 / 3) * (b + 1);
        int const i = (int) lh_rev8((unsigned) (j << 2));
        float   f0, f1, f2, f3, w;
        f0 = ws[i] * x[LH_FZ(i + k)];
        w = ws[0x7f - i] * x[LH_FZ(i + k + 0x80)];
        f1 = f0 - w;
        f0 = f0 + w;
        f2 = ws[i + 0x40] * x[LH_FZ(i + k + 0x40)];
        w = ws[0x3f - i] * x[LH_FZ(i + k + 0xc0)];
        f3 = f2 - w;
        f2 = f2 + w;
        o[q][0] = f0 + f2;
        o[q][2] = f0 - f2;
        o[q][1] = f1 + f3;
        o[q][3] = f1 - f3;
        f0 = ws[i + 0x01] * x[LH_FZ(i + k + 0x01)];
        w = ws[0x7e - i] * x[LH_FZ(i + k + 0x81)];
        f1 = f0 - w;
        f0 = f0 + w;
        f2 = ws[i + 0x41] * x[LH_FZ(i + k + 0x41)];
        w = ws[0x3e - i] * x[LH_FZ(i + k + 0xc1)];
        f3 = f2 - w;
        f2 = f2 + w;
        o[q][4] = f0 + f2;
        o[q][6] = f0 - f2;
        o[q][5] = f1 + f3;
        o[q][7] = f1 - f3;
    }
    LH_WAVE_SYNC_MEM();         /* every lane has read its samples */
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int const t = lane + 64 * q;
        if (t < 3 * (LH_BLKSIZE_S / 8)) {
            int const at = (t >> 5) * LH_BLKSIZE_S + 4 * (t & 31);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                x[LH_FZ(at + k)] = o[q][k];
                x[LH_FZ(at + k + LH_BLKSIZE_S / 2)] = o[q][4 + k];
            }
        }
    }
    LH_WAVE_SYNC_MEM();
    for (int t = lane; t < 3 * (LH_BLKSIZE_S / 8); t += 64) {
        int const b = t >> 5, u = t & 31;
        lh_fht_unit_fz < 4 > (*(const lh_f32x4 *) c.T->fht_tw[0][u % 2], x + b * LH_BLKSIZE_S, u);
    }
    LH_WAVE_SYNC_MEM();
    for (int t = lane; t < 3 * (LH_BLKSIZE_S / 8); t += 64) {
        int const b = t >> 5, u = t & 31;
        lh_fht_unit_fz < 16 > (*(const lh_f32x4 *) c.T->fht_tw[1][u % 8], x + b * LH_BLKSIZE_S, u);
    }
    LH_WAVE_SYNC_MEM();
    for (int t = lane; t < 3 * (LH_BLKSIZE_S / 8); t += 64) {
        int const b = t >> 5, u = t & 31;
        lh_fht_unit_fz < 64 > (*(const lh_f32x4 *) c.T->fht_tw[2][u % 32], x + b * LH_BLKSIZE_S, u);
    }
    LH_WAVE_SYNC_MEM();
}

/* power spectrum of one or two pseudo-channels from two SWIZZLED transforms of n points at wl / wr + first (lh_fft_energy,
 * lh_fft_energy_pair: reference psymodel.c:664-688, 713-736): for the short transforms, whose spectra have a place of their own */
LH_DEVFN void
lh_an_spectra_short(const LhCtx & c, int w, int both, const float *wl, const float *wr, int first, float *out_own, float *out_ms)
{
    float const sqrt2_half = (float) (LH_SQRT2 * 0.5f);
    int const n = LH_BLKSIZE_S, h = n >> 1;
    for (int m = c.lane; m <= h; m += 64) {
        int const ire = first + m, iim = first + ((m == 0) ? 0 : (n - m));
        float const lre = wl[LH_FZ(ire)], lim = wl[LH_FZ(iim)], rre = wr[LH_FZ(ire)], rim = wr[LH_FZ(iim)];
        float const ore = w ? rre : lre, oim = w ? rim : lim;
        float const mre = (w ? lre - rre : lre + rre) * sqrt2_half, mim = (w ? lim - rim : lim + rim) * sqrt2_half;
        out_own[m] = (m == 0) ? ore * ore : (ore * ore + oim * oim) * 0.5f;
        if (both)
            out_ms[m] = (m == 0) ? mre * mre : (mre * mre + mim * mim) * 0.5f;
    }
}

/* The long power spectra (reference psymodel.c:664-688; lh_fft_energy / lh_fft_energy_pair) IN PLACE: rows of LH_AN_ROW
 * floats per pseudo-channel over the two transforms they are formed from.  Both waves read everything they need of both
 * transforms into registers, the workgroup meets, then the rows are written.  Wave w: its own channel, and with `ms' the mid
 * (w = 0) or side (w = 1) channel.  Every thread of the workgroup must call. */
LH_DEVFN void
lh_an_spectra(const LhCtx & c, int w, int own, int ms, float *work)
{
    float const sqrt2_half = (float) (LH_SQRT2 * 0.5f);
    const float *wl = work, *wr = work + LH_BLKSIZE;
    float   eo[9], em[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int const m0 = c.lane + 64 * k, m = m0 <= LH_BLKSIZE / 2 ? m0 : LH_BLKSIZE / 2;
        int const ire = m, iim = (m == 0) ? 0 : (LH_BLKSIZE - m);
        float const lre = wl[LH_FZ(ire)], lim = wl[LH_FZ(iim)], rre = wr[LH_FZ(ire)], rim = wr[LH_FZ(iim)];
        float const ore = w ? rre : lre, oim = w ? rim : lim;
        float const mre = (w ? lre - rre : lre + rre) * sqrt2_half, mim = (w ? lim - rim : lim + rim) * sqrt2_half;
        eo[k] = (m == 0) ? ore * ore : (ore * ore + oim * oim) * 0.5f;
        em[k] = (m == 0) ? mre * mre : (mre * mre + mim * mim) * 0.5f;
    }
    LH_SYNC_WG_LDS();           /* both waves have read both transforms */
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int const m = c.lane + 64 * k;
        if (m <= LH_BLKSIZE / 2) {
            if (own)
                work[w * LH_AN_ROW + m] = eo[k];
            if (ms)
                work[(w + 2) * LH_AN_ROW + m] = em[k];
        }
    }
}

/* ---- transforms, spectra, masking up to the recurrences; one workgroup (wave = channel) per (stream, granule) ---- */
#ifndef LH_EMU
extern "C" __global__ void __launch_bounds__(LH_NT, 6)
#else
void
#endif
lh_analysis_kernel(const LhConfig * LH_KRESTRICT cfg, const LhTables * LH_KRESTRICT T, const int16_t * LH_KRESTRICT pcm, const float *pcmf, const LhStreamDesc * descs,
                   LhMidFrame * frames, int nstreams, LhStreamState * states)
{
#if defined(LH_APROF) && !defined(LH_EMU)
    unsigned long long *aprof = &states[blockIdx.y].prof[0][0];
    if (threadIdx.x == 0)
        lh_ap_base = aprof;
#endif
    LH_AP_T0();
    LhLds & L = lh_lds;
    LhLds & P = L;              /* (eb / thr) */
    float  *const work = L.work;
#define LH_AN_E(chn) (work + (chn) * LH_AN_ROW)
    int const sidx = (int) blockIdx.y;
    LhCtx   c;
    c.cfg = cfg;
    c.T = T;
    c.st = nullptr;
    c.pcm = pcm;
    c.pcmf = pcmf;
    c.d = lh_desc_uniform(descs, sidx);
    c.tid = (int) threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = lh_uni_i(c.tid >> 6);
    LhGranuleAt const ga = lh_granule_at(c.d, (int) blockIdx.x);
    if (!ga.live)
        return;
    int const lane = c.lane, w = c.wave, gr = ga.gr;
    int const n_chn_psy = (cfg->mode == LH_MODE_JOINT_STEREO) ? 4 : cfg->channels;
    int const bufbase = 576 + gr * 576 - LH_FFTOFFSET;
    LhMidGr *mg = &frames[ga.at].small.gr[gr];
    if (c.tid < 2)
        L.uselong[c.tid] = mg->uselong[c.tid];
    LH_AP(0);
    lh_stage_span < LH_BLKSIZE, LH_NT > (c, LH_SPAN, LH_SPAN + LH_BLKSIZE, ga.frame_base + bufbase);
    LH_SYNC_WG_LDS();
    LH_AP(1);
    /* long FFTs of L (wave 0) and R (wave 1), in place over their spans */
    if (w < cfg->channels)
        lh_fft_long_inplace(c, work + w * LH_BLKSIZE);
    LH_SYNC_WG_LDS();
    LH_AP(2);
    /* power spectra of this wave's one or two pseudo-channels, in place over the transforms */
    lh_an_spectra(c, w, w < n_chn_psy, n_chn_psy == 4, work);
    LH_WAVE_SYNC_MEM();
    /* (the long-block spreading matrix is read where it lies, in the L2: with four waves per SIMD its look-ups hide behind
     * the other waves, and staging it cost a granule two barriers and an HBM round trip) */
    const float *stg_s3 = T->psy_l.s3;
    LH_WAVE_SYNC_MEM();
    LH_AP(3);
    LH_AP(4);
    /* serial sums: total energy (bins 11..512) of chn w (lane 0) and w + 2 (lane 1), loudness of channel w (lane 2), in
     * bin order (reference psymodel.c:213-226, 690-696); see the fused kernel for the layout */
    {
        int const chn = (lane == 1) ? w + 2 : w;
        int const summing = lane < 3 && chn < n_chn_psy;
        int const loud = (lane == 2);
        float  *prod = (w == 0) ? P.eb : P.thr; /* [256] per wave */
        const float *ew = T->ath_eql_w;
        const float *e = LH_AN_E(summing ? chn : w);
        float   acc = 0.0f;
        for (int h = 0; h < 2; h++) {
            int const j0 = 256 * h;
            LH_WAVE_SYNC_MEM();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int const j = j0 + lane + 64 * q;
                prod[j - j0] = LH_AN_E(w)[j] * ew[j];
            }
            LH_WAVE_SYNC_MEM();
            if (summing) {
                const lh_f32x4 *s4 = (const lh_f32x4 *) ((loud ? prod - j0 : e) + j0);
                lh_f32x4 a0 = s4[0], a1 = s4[1], a2 = s4[2], a3 = s4[3];
                if (h == 0 && !loud) {
                    /* bins 0..10 are not part of the total energy */
                    a0.x = a0.y = a0.z = a0.w = 0.0f;
                    a1.x = a1.y = a1.z = a1.w = 0.0f;
                    a2.x = a2.y = a2.z = 0.0f;
                }
                for (int g = 0; g < 64; g += 4) {
                    int const n = (g + 4 < 64) ? g + 4 : g;
                    lh_f32x4 const b0 = s4[n], b1 = s4[n + 1], b2 = s4[n + 2], b3 = s4[n + 3];
#ifndef LH_EMU
                    asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\t"
                                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %0, %0, %7\n\tv_add_f32 %0, %0, %8\n\t"
                                 "v_add_f32 %0, %0, %9\n\tv_add_f32 %0, %0, %10\n\tv_add_f32 %0, %0, %11\n\tv_add_f32 %0, %0, %12\n\t"
                                 "v_add_f32 %0, %0, %13\n\tv_add_f32 %0, %0, %14\n\tv_add_f32 %0, %0, %15\n\tv_add_f32 %0, %0, %16"
                                 : "+v"(acc)
                                 : "v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w),
                                   "v"(a2.x), "v"(a2.y), "v"(a2.z), "v"(a2.w), "v"(a3.x), "v"(a3.y), "v"(a3.z), "v"(a3.w));
#else
                    acc += a0.x; acc += a0.y; acc += a0.z; acc += a0.w;
                    acc += a1.x; acc += a1.y; acc += a1.z; acc += a1.w;
                    acc += a2.x; acc += a2.y; acc += a2.z; acc += a2.w;
                    acc += a3.x; acc += a3.y; acc += a3.z; acc += a3.w;
#endif
                    a0 = b0;
                    a1 = b1;
                    a2 = b2;
                    a3 = b3;
                }
            }
        }
        if (summing) {
            if (!loud) {
                acc += e[LH_BLKSIZE / 2];
                mg->tot_ener[chn] = acc;
            }
            else {
                acc = (float) (acc * LH_VO_SCALE);
                mg->loud[w] = acc;
            }
        }
        LH_WAVE_SYNC_MEM();
    }
    /* (the products a wave's loudness lane adds up lie in the partition arrays the OTHER wave's masking is about to write) */
    LH_SYNC_WG_LDS();
    LH_AP(5);
    /* masking, long blocks, up to the recurrences: the wave's one or two pseudo-channels together */
    {
        LhMidLong *ml = &frames[ga.at].lng;
        if (n_chn_psy == 4) {
            LhMaskChan const two[2] = {
                {w, LH_AN_E(w), &P.eb[w * 64], &P.thr[w * 64], nullptr, nullptr, &ml->m[gr][w]},
                {w + 2, LH_AN_E(w + 2), &P.eb[(w + 2) * 64], &P.thr[(w + 2) * 64], nullptr, nullptr, &ml->m[gr][w + 2]}
            };
            lh_compute_masking < 2, 1 > (c, 1, two, stg_s3);
        }
        else if (w < n_chn_psy) {
            LhMaskChan const one[1] = { {w, LH_AN_E(w), &P.eb[w * 64], &P.thr[w * 64], nullptr, nullptr, &ml->m[gr][w]} };
            lh_compute_masking < 1, 1 > (c, 1, one, stg_s3);
        }
    }
    /* short blocks (reference psymodel.c:1470-1500): only for a granule in which a channel switches */
    LH_AP(6);
    int const any_short = lh_uni_i(!(L.uselong[0] && L.uselong[1]));
    if (!any_short)
        return;
    LH_SYNC_WG_LDS();           /* the long spectra and the staged matrix are done with */
    lh_stage_span < LH_BLKSIZE, LH_NT > (c, LH_SPAN, LH_SPAN + LH_BLKSIZE, ga.frame_base + bufbase);
    LH_SYNC_WG_LDS();
    if (!L.uselong[w])
        lh_fft_short_inplace(c, work + w * LH_BLKSIZE);
    LH_SYNC_WG_LDS();
    for (int sblock = 0; sblock < 3; sblock++) {
        if (w < n_chn_psy && !L.uselong[w]) {
            LhMidShort *ms = &frames[ga.at].shrt;
            int const both = (n_chn_psy == 4);
            lh_an_spectra_short(c, w, both, work, work + LH_BLKSIZE, sblock * LH_BLKSIZE_S, L.eshort[w], L.eshort[w + 2 < 4 ? w + 2 : 3]);
            LH_WAVE_SYNC_MEM();
            if (both) {
                LhMaskChan const two[2] = {
                    {w, L.eshort[w], &P.eb[w * 64], &P.thr[w * 64], nullptr, nullptr, &ms->m[gr][sblock][w]},
                    {w + 2, L.eshort[w + 2], &P.eb[(w + 2) * 64], &P.thr[(w + 2) * 64], nullptr, nullptr, &ms->m[gr][sblock][w + 2]}
                };
                lh_compute_masking < 2, 1 > (c, 0, two, T->psy_s.s3);
            }
            else {
                LhMaskChan const one[1] = { {w, L.eshort[w], &P.eb[w * 64], &P.thr[w * 64], nullptr, nullptr, &ms->m[gr][sblock][w]} };
                lh_compute_masking < 1, 1 > (c, 0, one, T->psy_s.s3);
            }
        }
        LH_SYNC_WG_LDS();
    }
    LH_AP(7);
}

#ifndef LH_EMU
/* all three in order on one HIP stream; max_frames = the longest frame range of the launch */
extern "C" int
lh_launch_analysis(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf, const LhStreamDesc * descs,
                   const LhStreamState * states, LhMidPools mid, int nstreams, int max_frames, void *stream)
{
    if (nstreams <= 0 || max_frames <= 0)
        return 0;
    dim3 const grid((unsigned) (LH_NGR * max_frames), (unsigned) nstreams);
    hipLaunchKernelGGL(lh_attack_kernel, grid, dim3(64), 0, (hipStream_t) stream, cfg, pcm, pcmf, descs, mid.frames, nstreams);
    hipLaunchKernelGGL(lh_attack_scan_kernel, dim3((unsigned) nstreams), dim3(64), 0, (hipStream_t) stream, cfg, T, descs, states,
                       mid.frames, nstreams);
    hipLaunchKernelGGL(lh_analysis_kernel, grid, dim3(LH_NT), 0, (hipStream_t) stream, cfg, T, pcm, pcmf, descs, mid.frames, nstreams,
                       (LhStreamState *) states);
    return (int) hipGetLastError();
}
#else
extern "C" int
lh_emu_analysis(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf, const LhStreamDesc * descs,
                const LhStreamState * states, const LhMidPools * pools, int nstreams, int max_frames)
{
    LhMidPools const mid = *pools;
    hipemu_dim3 grid = { (unsigned) (LH_NGR * max_frames), (unsigned) nstreams, 1 }, b64 = { 64, 1, 1 }, b128 = { LH_NT, 1, 1 };
    hipemu_dim3 grid1 = { (unsigned) nstreams, 1, 1 };
    hipemu_run(grid, b64,[=] () {
               lh_attack_kernel(cfg, pcm, pcmf, descs, mid.frames, nstreams);
               }
    );
    hipemu_run(grid1, b64,[=] () {
               lh_attack_scan_kernel(cfg, T, descs, states, mid.frames, nstreams);
               }
    );
    hipemu_run(grid, b128,[=] () {
               lh_analysis_kernel(cfg, T, pcm, pcmf, descs, mid.frames, nstreams, (LhStreamState *) states);
               }
    );
    return 0;
}
#endif
