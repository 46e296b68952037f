/*
 * lh_dev_common.h -- shared definitions of the HIP kernels: the LDS image of a
 * workgroup (one stream, two waves = two channels) and the launch context.
 *
 * LDS budget (must stay <= 40 KB so that four workgroups share a CU's 160 KB and
 * a batch of 1024 streams is fully resident on 256 CUs): see LhLds below.
 */
#ifndef LH_DEV_COMMON_H
#define LH_DEV_COMMON_H

#include <stdint.h>
#include "lamehip_types.h"
#include "lh_device.h"
#include "lh_wave.h"
#include "lh_dev_math.h"

#define LH_NT 128               /* threads that walk the frame: wave 0 = left/mid, wave 1 = right/side */
/* The kernel file is compiled twice (csrc/Makefile): for MPEG-1 streams (two granules per frame; the default) and,
 * with -DLH_LSF, for MPEG-2 / 2.5 streams (one granule of 576 samples per frame, partitioned scalefactors, the 8 kHz
 * band limits).  The granule count and everything that hangs on it are compile-time constants in either object, so
 * the MPEG-1 kernel carries no test for the other case (as a run-time switch it cost the headline 5 %). */
#ifdef LH_LSF
#define LH_NGR 1
#define LH_IS_LSF 1
#define LH_RATE8K(c) ((c).rate8k)
#else
#define LH_NGR 2
#define LH_IS_LSF 0
#define LH_RATE8K(c) 0
#endif
#define LH_BLOCK LH_NT
#define LH_SQRT2 1.41421356237309504880

/* LH_SYNC_WG(): workgroup barrier with a full fence (LDS and HBM: __syncthreads()).
 * LH_SYNC_WG_LDS(): the same for phases whose waves exchange data through LDS only (the psy model,
 * the transforms): global loads in flight -- table look-ups, the next phase's prefetches -- are not
 * drained at the barrier. */
#ifdef LH_EMU
#define LH_SYNC_WG() do { __syncthreads(); } while (0)
#define LH_SYNC_WG_LDS() do { __syncthreads(); } while (0)
#else
#define LH_SYNC_WG() do { __syncthreads(); } while (0)
#define LH_SYNC_WG_LDS() do { \
                              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
                              __builtin_amdgcn_s_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#endif

#if defined(LH_PROF) && !defined(LH_EMU)
#define LH_PT(var) unsigned long long var = clock64()
#define LH_PA(idx, var) do { if (c.lane == 0 && c.wave < 2) lh_lds.prof[c.wave][idx] += (unsigned) (clock64() - var); } while (0)
#define LH_PC(idx) do { if (c.lane == 0 && c.wave < 2) lh_lds.prof[c.wave][idx] += 1; } while (0)
#elif !defined(LH_PT)           /* (lh_analysis.hip brings its own with -DLH_APROF) */
#define LH_PT(var) do { } while (0)
#define LH_PA(idx, var) do { } while (0)
#define LH_PC(idx) do { } while (0)
#endif

/* development aid: section markers in the generated assembly (tools/isa_sections.py) */
#if defined(LH_MARK) && !defined(LH_EMU)
#define LQ_MARK(name) asm volatile("; LQMARK " name)
#else
#define LQ_MARK(name) do { } while (0)
#endif

/* development aid (-DLH_TRACE, tools/trace_profile.py; profiles/r06_wait_breakdown.txt): where a search iteration's cycles go.
 * A mark reads the clock (s_memtime, then s_waitcnt lgkmcnt(0): every LDS operation in flight is waited for at a mark, which
 * is what makes the segments add up) and adds the cycles since the wave's previous mark -- whichever it was -- to the
 * segment that ENDS at this mark: lane `id' of one vector register holds segment id's cycles, lane id of a second its
 * visits; no memory is touched until the search stage returns (lh_tr_flush: one atomic per lane into lh_trace_buf).  A mark
 * costs ~75 cycles (measured by a pair of adjacent marks: segment 63); the tool takes that off per visit. */
#if defined(LH_TRACE) && !defined(LH_EMU)
#define LH_NTRACE 64
__device__ unsigned long long lh_trace_buf[2][2][LH_NTRACE];   /* [wave][cycles | visits][segment] over all streams */
struct LhTr {
    uint32_t acc, cnt, last;
};
LH_DEVFN void
lh_tr_mark(LhTr & t, int id)
{
    unsigned long long now;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) :: "memory");
    uint32_t const lo = (uint32_t) now;
    int const me = ((int) (threadIdx.x & 63u) == id);
    t.acc += me ? lo - t.last : 0u;
    t.cnt += me ? 1u : 0u;
    t.last = lo;
}
LH_DEVFN void
lh_tr_begin(LhTr & t)
{
    unsigned long long now;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) :: "memory");
    t.acc = 0;
    t.cnt = 0;
    t.last = (uint32_t) now;
}
LH_DEVFN void
lh_tr_flush(const LhTr & t, int wave)
{
    int const lane = (int) (threadIdx.x & 63u);
    if (wave < 2 && t.cnt) {
        atomicAdd(&lh_trace_buf[wave][0][lane], (unsigned long long) t.acc);
        atomicAdd(&lh_trace_buf[wave][1][lane], (unsigned long long) t.cnt);
    }
}
#define LQ_T(S, id) lh_tr_mark((S).tr, id)
#else
#define LQ_T(S, id) do { } while (0)
#endif

LH_DEVFN float
lh_fabsf(float x)
{
    return lh_u32_as_f32(lh_f32_as_u32(x) & 0x7fffffffu);
}

/* inclusive prefix sum over the wave */
LH_DEVFN uint32_t
lh_wave_scan_u32(uint32_t v)
{
#ifdef LH_EMU
    const uint64_t *x = hipemu_wave_exchange(v);
    uint32_t s = 0;
    int const me = lh_lane();
    for (int i = 0; i <= me; i++)
        s += (uint32_t) x[i];
    return s;
#else
    /* row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast 15 / 31 across them */
    v += lh_dpp < 0x111, 0u > (v);
    v += lh_dpp < 0x112, 0u > (v);
    v += lh_dpp < 0x114, 0u > (v);
    v += lh_dpp < 0x118, 0u > (v);
    v += lh_dpp_rows < 0x142, 0xa, 0u > (v);
    v += lh_dpp_rows < 0x143, 0xc, 0u > (v);
    return v;
#endif
}

/* inclusive running maximum over the wave (unsigned) */
LH_DEVFN uint32_t
lh_wave_scan_max_u32(uint32_t v)
{
#ifdef LH_EMU
    const uint64_t *x = hipemu_wave_exchange(v);
    uint32_t s = 0;
    int const me = lh_lane();
    for (int i = 0; i <= me; i++)
        s = (uint32_t) x[i] > s ? (uint32_t) x[i] : s;
    return s;
#else
    uint32_t t;
    t = lh_dpp < 0x111, 0u > (v);
    v = t > v ? t : v;
    t = lh_dpp < 0x112, 0u > (v);
    v = t > v ? t : v;
    t = lh_dpp < 0x114, 0u > (v);
    v = t > v ? t : v;
    t = lh_dpp < 0x118, 0u > (v);
    v = t > v ? t : v;
    t = lh_dpp_rows < 0x142, 0xa, 0u > (v);
    v = t > v ? t : v;
    t = lh_dpp_rows < 0x143, 0xc, 0u > (v);
    v = t > v ? t : v;
    return v;
#endif
}

/* ---- LDS layout ---------------------------------------------------- */
struct LhPsyLds {
    union {
        struct {
            float   wsamp[2][LH_BLKSIZE];       /* FHT work buffers of L and R (3x256 for short blocks) */
            union {
                struct {
                    float   hpf[2][576];        /* high-passed samples for attack detection */
                } a;
                struct {
                    float   energy[4][LH_HBLKSIZE + 3];   /* power spectra of L,R,M,S */
                } b;
            };
        };
        struct {                        /* split pipeline's encode kernel: the frame's small record and long-block masking, */
            float   keep[1728];         /* brought in from HBM in one batch at the top of the frame (lh_encode_frame); behind */
            LhMidSmall small;           /* channel 0's second quantised image, where the loop tables stay for the whole launch */
            LhMidLong lng;
        } mid;
    };
    float   eb[4 * 64];
    float   thr[4 * 64];        /* lh_compute_masking parks a channel's tonality indices here until its thresholds are known */
};

/* words between the time slots of the sub-band samples: 32 in the encode kernel; lh_subband.hip pads to 33, which spreads
 * the slots -- one per lane in the butterfly network -- over the LDS banks */
#ifndef LH_SB_STRIDE
#define LH_SB_STRIDE 32
#endif
#define LH_SB_GRANULE (18 * LH_SB_STRIDE)
struct LhMdctLds {
    float   sb[2][3][LH_SB_GRANULE];    /* [ch][0 = previous granule, 1 = gr0, 2 = gr1][slot * LH_SB_STRIDE + band] */
};

/* wave-uniform scalar image of gr_info (reference l3side.h:47-84): every lane holds
 * its own identical copy in registers, so updates need no cross-lane ordering */

struct LhGrR {
    int     part2_3_length, big_values, count1, global_gain, scalefac_compress;
    int     table_select[3], subblock_gain[4];
    int     region0_count, region1_count, preflag, scalefac_scale, count1table_select;
    int     part2_length, count1bits;
    float   xrpow_max;
};

/* subblock_gain[w] without a dynamically indexed register array (which would force the
 * whole struct into scratch memory) */
LH_DEVFN int
lh_sbg(const LhGrR & g, int w)
{
    /* the four values pass through readfirstlane first: a select between plain loads of the
     * struct's fields is turned into ONE load from a selected address by the optimiser, and that
     * dynamic address pins the whole struct (every use of it, everywhere) in scratch memory */
    int const a = lh_uni_i(g.subblock_gain[0]), b = lh_uni_i(g.subblock_gain[1]), c2 = lh_uni_i(g.subblock_gain[2]),
        d = lh_uni_i(g.subblock_gain[3]);
    return w == 0 ? a : w == 1 ? b : w == 2 ? c2 : d;
}

/* wave-uniform per-granule geometry + the scalar part of calc_noise_data */
struct LhQR {
    int     block_type, sfb_lmax, sfb_smin, psy_lmax, sfbmax, psymax, sfbdivide, mnc;
    int     pn_global_gain, pn_sfb_count1;
    int     substep_shaping;
    int     ath_over;           /* calc_xmin's return value != 0 (only the VBR loop asks: analog silence) */
    int     s_mnc;              /* band that holds line max_nonzero_coeff (set by the search loop) */
};

/* copies of R / g that came back from an out-of-line stage through per-lane memory */
LH_DEVFN LhQR
lh_uniform(const LhQR & r)
{
    LhQR    o;
    o.block_type = lh_uni_i(r.block_type);
    o.sfb_lmax = lh_uni_i(r.sfb_lmax);
    o.sfb_smin = lh_uni_i(r.sfb_smin);
    o.psy_lmax = lh_uni_i(r.psy_lmax);
    o.sfbmax = lh_uni_i(r.sfbmax);
    o.psymax = lh_uni_i(r.psymax);
    o.sfbdivide = lh_uni_i(r.sfbdivide);
    o.mnc = lh_uni_i(r.mnc);
    o.pn_global_gain = lh_uni_i(r.pn_global_gain);
    o.pn_sfb_count1 = lh_uni_i(r.pn_sfb_count1);
    o.substep_shaping = lh_uni_i(r.substep_shaping);
    o.ath_over = lh_uni_i(r.ath_over);
    o.s_mnc = lh_uni_i(r.s_mnc);
    return o;
}

LH_DEVFN LhGrR
lh_uniform(const LhGrR & a)
{
    LhGrR   o;
    o.part2_3_length = lh_uni_i(a.part2_3_length);
    o.big_values = lh_uni_i(a.big_values);
    o.count1 = lh_uni_i(a.count1);
    o.global_gain = lh_uni_i(a.global_gain);
    o.scalefac_compress = lh_uni_i(a.scalefac_compress);
    o.table_select[0] = lh_uni_i(a.table_select[0]);
    o.table_select[1] = lh_uni_i(a.table_select[1]);
    o.table_select[2] = lh_uni_i(a.table_select[2]);
    o.subblock_gain[0] = lh_uni_i(a.subblock_gain[0]);
    o.subblock_gain[1] = lh_uni_i(a.subblock_gain[1]);
    o.subblock_gain[2] = lh_uni_i(a.subblock_gain[2]);
    o.subblock_gain[3] = lh_uni_i(a.subblock_gain[3]);
    o.region0_count = lh_uni_i(a.region0_count);
    o.region1_count = lh_uni_i(a.region1_count);
    o.preflag = lh_uni_i(a.preflag);
    o.scalefac_scale = lh_uni_i(a.scalefac_scale);
    o.count1table_select = lh_uni_i(a.count1table_select);
    o.part2_length = lh_uni_i(a.part2_length);
    o.count1bits = lh_uni_i(a.count1bits);
    o.xrpow_max = lh_uni_f(a.xrpow_max);
    return o;
}

struct LhNoiseRes {             /* calc_noise_result */
    int     over_count, over_SSD, bits;
    float   over_noise, tot_noise, max_noise;
};

/* LDS image of one channel's quantiser working set.  Arrays indexed by band are
 * written by lane == band only (then LH_WAVE_SYNC); line arrays by the lane
 * owning the line. */
/* Huffman code lengths for the CBR search loop (lh_dev_qloop.h): 16 x 16 grids of dwords indexed
 * x * 16 + y, each entry = the lengths of a region's (up to) three candidate tables, 10 bits each.
 * Two big grids (ESC tables 16.. / 24.. + count of values == 15; tables 13 / 14 / 15) and one grid
 * that holds the small alphabets side by side (LQ_ORG_*: cell of a group's (0,0)). */
#define LQ_SMALL_CELLS 192
#define LQ_ORG_T10  0           /* tables 10 / 11 / 12: rows 0..7, columns 0..7 */
#define LQ_ORG_T7   8           /* tables 7 / 8 / 9: rows 0..5, columns 8..13 */
#define LQ_ORG_ZERO 104         /* row 6, column 8: an all-zero cell (region maximum 0) */
#define LQ_ORG_T5   128         /* tables 5 / 6: rows 8..11, columns 0..3 */
#define LQ_ORG_T2   132         /* tables 2 / 3: rows 8..10, columns 4..6 */
#define LQ_ORG_T1   136         /* table 1: rows 8..9, columns 8..9 */

struct LhChanLds {
    float   xrpow[576];         /* the CBR search keeps xrpow in registers and uses this as calc_noise's scratch */
    union {
        float   save_xrpow[576];        /* scratch of the VBR loop and of best_huffman_divide */
        uint32_t hl3_big[2][256];       /* while the CBR search runs: [0] = ESC grid, [1] = tables 13..15 */
    };
    int16_t ix[2][576];         /* [0] = best so far (cod_info), [1] = working copy (cod_info_w) */
    int     sf[2][LH_SFBMAX + 1];       /* scalefactors of the two images */
    uint16_t width[LH_SFBMAX + 1], start[LH_SFBMAX + 1];
    uint8_t window[LH_SFBMAX + 1];
    uint8_t sfb_mode[LH_SFBMAX + 1];    /* scratch: flags per band */
    uint8_t sfb_of_line[576];
    float   l3_xmin[LH_SFBMAX + 1];
    union {
        struct {
            float   distort[LH_SFBMAX + 1];
            /* calc_noise_data (reference quantize_pvt.h:75-82), per band */
            int     pn_step[LH_SFBMAX + 1];
            float   pn_noise[LH_SFBMAX + 1], pn_noise_log[LH_SFBMAX + 1];
            int     pseudohalf[LH_SFBMAX + 1];
        };
        uint32_t hl3_small[LQ_SMALL_CELLS];     /* while the CBR search runs (it keeps the five arrays in registers) */
    };
    /* scratch */
    float   sfb_f[LH_SFBMAX + 1];
    float   zero2[2];           /* two zeros on an 8-byte boundary: where masked-out term loads are redirected */
    float   pad2[2];            /* keeps sizeof a multiple of 16 */
};

/* hot lookup tables of the quantiser, staged into LDS for the iteration-loop phase
 * (they live behind xr in the region the PCM window occupied earlier in the frame) */
struct LhQTabs {
    uint32_t table23[9], table56[16];
    uint16_t sfb_l[24];
    uint8_t ht_len[1124];       /* code lengths of tables 1..15 back to back (lh_ht_off()); the ESC
                                 * tables are read through largetbl */
    uint32_t bvpack[288];       /* big_values/2 - 1 -> region0_count | region1_count << 4 | end of region 0 << 8
                                 * | end of region 1 << 18 (reference takehiro.c:1334-1375 folded with sfb_l) */
    uint16_t sfb_s3, pad;       /* sfb_s[3] */
    uint8_t t32l[16], t33l[16];
    uint32_t t3233[16];         /* t32l << 16 | t33l */
    uint32_t t3233p[16];        /* the same indexed by v | w << 1 | x << 2 | y << 3 (lq_count: quadruples from two packed pairs) */
    uint8_t pretab[24];
    uint32_t ctabA[32], ctabB[32];      /* lh_dev_qloop.h: per class of a region maximum, the grid origin of its
                                         * candidate tables / the tables and their linbits (lq_class_tabs) */
    float   pow43h[256];        /* heads of pow43 / adj43asm: nearly all quantised values are < 256 */
    float   qthr[256];          /* LhTables.qthr: the quantiser's second rounding as a comparison */
};


/* what the VBR loop keeps of a granule between its first pass and the (rare) second pass that
 * squeezes the frame into its bit budget (algo_t + sfwork / vbrsfmin, reference vbrquantize.c:47-55,
 * 1254-1256) */
struct LhVbrSave {
    uint8_t sfwork[LH_SFBMAX + 1];
    uint8_t sfmin[LH_SFBMAX + 1];
    int     mingain_l, mingain_s[3];
    int     global_gain, mnc, max_bits, use_bits;
    int     ath_over, nonzero, pad[2];
    /* what a bit count leaves alone when the granule has no big values (region counts) or a region is empty (its table):
     * the second pass finds them as the first pass left them (reference takehiro.c:700-760 returns early / skips) */
    int     table_select[3], region0_count, region1_count, pad2[3];
};

/* The profiling build's cycle counters take the LDS the old VBR loop's parking area needs. */
#if !(defined(LH_PROF) && !defined(LH_EMU))
#define LH_VBR_OLD 1
#endif

/* what the old VBR loop (vbr_rh) parks of a granule between its passes over a frame: the granule as the search left
 * it, before the finishing steps (lh_dev_vbrold.h) */
struct LhVbrOldSave {
    LhQR    R;
    LhGrR   g;
    uint8_t sf[LH_SFBMAX + 1];
    float   xmin[LH_SFBMAX + 1];        /* allowed noise, raised by every bit-pressure pass */
    int     used_bits, fin_bits;        /* part2_3 + part2 lengths after the search / after the finishing steps */
    int     ath_over, pad;
};

struct LhQuantLds {
    LhChanLds ch[2];
};
static_assert(__builtin_offsetof(LhTables, pow43) % 128 == 0 && __builtin_offsetof(LhTables, vqthr) % 128 == 0
              && __builtin_offsetof(LhTables, vq3) % 128 == 0,
              "the gathered tables start on cache lines (hipMalloc aligns the struct itself)");
static_assert(sizeof(LhChanLds) % 16 == 0, "both channels' float2/float4 accesses need 16-byte alignment");
static_assert(__builtin_offsetof(LhPsyLds, mid.small) >= __builtin_offsetof(LhChanLds, ix[1]) + sizeof(((LhChanLds *) 0)->ix[1])
              && __builtin_offsetof(LhPsyLds, mid.lng) + sizeof(LhMidLong) <= sizeof(LhChanLds) + __builtin_offsetof(LhChanLds, ix[1])
              && __builtin_offsetof(LhPsyLds, mid.small) % 16 == 0,
              "the staged mid record lies between the two channels' second images (log table / VBR step tables)");

/* VBR only: the step tables of the scalefactor search, ipow20[0..255] and pow20[116..371]
 * (= pow20[sf + Q_MAX2]), staged over the second quantised image, which that loop never uses */
#define LH_VBR_IPOW20 ((float *) lh_lds.u.quant.ch[0].ix[1])
#define LH_VBR_POW20  ((float *) lh_lds.u.quant.ch[1].ix[1])

/* CBR / ABR / old VBR: the 513 entries of LhTables.log_table that calc_noise's logarithm interpolates between
 * (reference util.c:976-1001), staged over the same unused second image once per frame -- 288 entries behind channel 0's
 * first image, the rest behind channel 1's -- so that the look-up on the search's dependent chain is an LDS round trip
 * (~64 cycles) and not one to the vector cache (~250 with two waves on the SIMD; 65 look-ups per channel and frame) */
#define LH_LOGT_SPLIT 288
#define LH_LOGT_LDS(m) (((const float *) lh_lds.u.quant.ch[0].ix[1]) \
                        [(m) + ((m) >= LH_LOGT_SPLIT ? (int) ((sizeof(LhChanLds) - 4 * LH_LOGT_SPLIT) / 4) : 0)])
#define LH_LOGT_LDS_W(m) (((float *) lh_lds.u.quant.ch[0].ix[1]) \
                          [(m) + ((m) >= LH_LOGT_SPLIT ? (int) ((sizeof(LhChanLds) - 4 * LH_LOGT_SPLIT) / 4) : 0)])
static_assert(sizeof(((LhChanLds *) 0)->ix[1]) == 4 * LH_LOGT_SPLIT && 2 * LH_LOGT_SPLIT >= 513, "log table over the second images");

/* launch context of the workgroup, written once per launch (frame_base per frame) by thread 0;
 * out-of-line stages read it from here instead of receiving a per-lane copy */
struct LhCtxShared {
    const LhConfig *cfg;
    const LhTables *T;
    LhStreamState *st;
    const int16_t *pcm;
    const float *pcmf;          /* not null: the window comes from this pool, already transformed (handle API) */
    uint8_t *bytes;             /* not null: finished MP3 bytes are assembled here (lh_dev_emit.h) */
    LhStreamDesc d;
    long long frame_base;       /* stream sample index of mfbuf[0] for the current frame: 1152 f - 528 */
#ifdef LH_SPLIT
    /* the encode kernel of the split pipeline: the current frame's record of the analysis kernels' output (lh_device.h) */
    const LhMidFrame *mid;
#endif
};

/* per-granule scalars on their way into / out of an out-of-line stage (one slot per wave) */
struct LhRgSlot {
    LhQR    R;
    LhGrR   g;
};

/* Per-stream state a wave carries in registers from frame to frame of a launch (loaded from
 * LhStreamState when the launch starts, stored when it ends: lh_encode_kernel).  Lane-local, so no
 * exchange is ever needed:
 *   n1 / n2 [pass]  the psy model's spread partition energies of the last two long-block calls for
 *                   pseudo-channel wave + 2 pass, lane = partition (lh_compute_masking);
 *   sb[k]           the polyphase output of the last granule of the previous frame, which the MDCT
 *                   overlaps with (reference sb_sample[ch][0], newmdct.c:943-1040): value
 *                   lane + 64 k of channel `wave'. */
struct LhPsyCarry {
    float   n1[2], n2[2];
};
struct LhWaveCarry {
    LhPsyCarry nb;
    float   sb[9];
    int     prio_rel, prio_late;        /* issue priority (lh_prio_apply in lh_kernels.hip): scheduling only */
};

/* The stream's small state words (the members of LhStreamState from loudness_sq_save to
 * ath_adjust_limit and from pefirbuf to status, same order): in LDS for the whole launch, so that
 * nothing on a frame's dependency chains waits for HBM (read at the start of the launch, written back
 * at its end by lh_encode_kernel, word for word). */
struct LhSmallState {
    float   loudness_sq_save[2];
    float   tot_ener[4];
    float   last_en_subshort[4][9];
    int     last_attacks[4];
    int     blocktype_old[2];
    float   ath_adjust_factor;
    float   ath_adjust_limit;
    /* -- second run of members -- */
    float   pefirbuf[19];
    int     slot_lag;
    int     ResvSize;
    int     ResvMax;
    int     main_data_begin;
    int     OldValue[2];
    int     CurrentStep[2];
    float   masking_lower;
    int     substep_shaping;
    int     frame_number;
    int     primed;
    int     status;
};
#define LH_SS_WORDS_A 50        /* loudness_sq_save .. ath_adjust_limit */
#define LH_SS_WORDS_B 32        /* pefirbuf .. status */
static_assert(sizeof(LhSmallState) == 4 * (LH_SS_WORDS_A + LH_SS_WORDS_B), "LhSmallState is two runs of words");
static_assert(__builtin_offsetof(LhStreamState, ath_adjust_limit) - __builtin_offsetof(LhStreamState, loudness_sq_save)
              == 4 * (LH_SS_WORDS_A - 1), "first run of LhStreamState's small members");
static_assert(__builtin_offsetof(LhStreamState, status) - __builtin_offsetof(LhStreamState, pefirbuf)
              == 4 * (LH_SS_WORDS_B - 1), "second run of LhStreamState's small members");

#ifdef LH_CUSTOM_LDS              /* (the analysis kernels' translation units bring their own, smaller image: a header's name) */
#include LH_CUSTOM_LDS
#else
struct LhLds {
    LhSmallState ss;
    /* Band energies / thresholds of the psy model, [L,R,M,S] each: a ring of three slots.  The model's
     * output is used one granule late (reference psymodel.c:1397-1460 hands back last call's values), so
     * with c = psy_slot at the start of a frame: slot c = what the previous frame's second granule
     * produced = the ratios of this frame's granule 0; granule 0 writes slot c + 1 = the ratios of
     * granule 1; granule 1 writes slot c + 2 = the next frame's slot c.  Nothing is copied, and the
     * values never leave LDS during a launch (LhStreamState.en / thm are read at the start of the
     * launch and written at its end). */
    float   psy_en[3][4][LH_XMIN_N];
    float   psy_thm[3][4][LH_XMIN_N];
    int     psy_slot;
    float   pe[2][4];
    float   tot_ener[2][4];
    float   loudness_sq[2][2];
    float   sub_short_factor[4][3];
    int8_t  ns_attacks[4][4];
    int     ns_uselong[4];
    int     uselongblock[2];
    int     next_blocktype[2];
    int     block_type[2][2];   /* [gr][ch] */
    /* frame scalars */
    int     mode_ext, padding, mean_bits, max_bits, frame_bits;
    int     targ_bits[2];
    int     bits_used[2];
    float   pe_use[2][2];
    float   ms_ener_ratio[2];
    int     scfsi[2][4];
    int     gr0_done[2];        /* frame number + 1 once the channel is through with granule 0 (lh_encode_frame) */
#if defined(LH_PROF) && !defined(LH_EMU)
    unsigned prof[2][LH_NPROF];         /* profiling builds only; cycles of one frame fit 32 bits */
#endif
    LhCtxShared ctx;
    LhRgSlot rg[2];
    /* both unions start on 16-byte boundaries: the kernels read float2 / float4 from the arrays
     * inside (ds_read_b64 / b128), and a misaligned wide LDS access is split by the hardware --
     * a layout change that shifted them by 4 bytes once cost 13 % of the whole kernel */
    union __attribute__((aligned(16))) {
        float   mf[2][LH_MF_NEEDED];    /* scaled float PCM window of the frame (psy, polyphase) */
        struct {
            float   xr[2][2][576];      /* [ch][gr] MDCT spectra; written after the last read of mf */
            LhQTabs qt;                 /* loaded after the MDCT, used by the iteration loop */
            union {                     /* [gr][ch]; in the tail the longer PCM window leaves free (and a little more) */
                LhVbrSave vbr[2][2];
#ifdef LH_VBR_OLD
                LhVbrOldSave old[2][2];
#endif
            };
        };
    };
    union __attribute__((aligned(16))) {
        LhPsyLds psy;
        LhMdctLds mdct;
        LhQuantLds quant;
    } u;
};

static_assert(sizeof(LhLds) <= 40960, "four workgroups per CU need <= 40 KiB of the 160 KiB LDS each");

/* The workgroup's LDS image (one stream), at file scope: every device function, in line or
 * not, addresses it as LDS with constant offsets (ds_* instructions).  Handing it to the
 * out-of-line stages by reference made all their accesses generic FLAT loads. */
__shared__ LhLds lh_lds __attribute__((aligned(16)));
#define LH_QT (&lh_lds.qt)
#endif                          /* !LH_CUSTOM_LDS */

/* ---- launch context -------------------------------------------------- */
struct LhCtx {
    const LhConfig *cfg;
    const LhTables *T;
    LhStreamState *st;
    const int16_t *pcm;
    const float *pcmf;
    LhStreamDesc d;
    long long frame_base;       /* stream sample index of mfbuf[0] for the current frame: 1152 f - 528 */
    int     lane, wave, tid;
    /* the settings the iteration loop tests again and again, read from HBM once (scalar
     * registers) instead of once per use */
    int     ns, ns_amp, sfb21_extra, full_outer_loop, subblock_gain;
    int     rate8k;             /* (LH_LSF build) an 8 kHz stream: 17 long / 9 short coded bands */
    int     l21;                /* sfb_l[21]: the first line of the last long band (the usual-case stages' padded sums fit up to 418) */
};

LH_DEVFN void
lh_ctx_hot(LhCtx & c)
{
    c.ns = lh_uni_i(c.cfg->noise_shaping);
    c.ns_amp = lh_uni_i(c.cfg->noise_shaping_amp);
    c.sfb21_extra = lh_uni_i(c.cfg->sfb21_extra);
    c.full_outer_loop = lh_uni_i(c.cfg->full_outer_loop);
    c.subblock_gain = lh_uni_i(c.cfg->subblock_gain);
    c.l21 = lh_uni_i(c.T->sfb_l[LH_SBPSY_L]);
#ifdef LH_LSF
    c.rate8k = lh_uni_i(c.cfg->samplerate <= 8000);
#else
    c.rate8k = 0;
#endif
}

/* The context reaches an out-of-line stage through per-lane memory, which hides from the
 * compiler that its pointers address HBM; routing them through the global address space once
 * lets it use global_/s_load instead of FLAT accesses for everything derived from them. */
#ifdef LH_EMU
#define LH_AS_GLOBAL(type, p) (p)
#else
/* p is wave-uniform and addresses HBM.  The value is rebuilt from two scalar registers behind an
 * empty asm statement, so that the optimiser cannot fold the address-space casts away; what
 * comes out is "a global pointer, cast to generic", which address-space inference then
 * propagates to every access made through it. */
template < typename X > __device__ __forceinline__ X *
lh_as_global(X * p)
{
    typedef __attribute__((address_space(1))) X GX;
    unsigned long long const u = (unsigned long long) p;
    unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) u);
    unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (u >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (X *) (GX *) (((unsigned long long) hi << 32) | lo);
}
#define LH_AS_GLOBAL(type, p) lh_as_global < type > (p)
#endif
/* the context of an out-of-line stage: shared part from LDS, pointers re-typed global */
LH_DEVFN LhCtx
lh_ctx_load(void)
{
    LhCtx   o;
    o.cfg = LH_AS_GLOBAL(const LhConfig, lh_lds.ctx.cfg);
    o.T = LH_AS_GLOBAL(const LhTables, lh_lds.ctx.T);
    o.st = LH_AS_GLOBAL(LhStreamState, lh_lds.ctx.st);
    o.pcm = LH_AS_GLOBAL(const int16_t, lh_lds.ctx.pcm);
    o.pcmf = LH_AS_GLOBAL(const float, lh_lds.ctx.pcmf);
    o.d = lh_lds.ctx.d;
    o.frame_base = lh_lds.ctx.frame_base;
    o.tid = (int) threadIdx.x;
    o.lane = o.tid & 63;
    o.wave = lh_uni_i(o.tid >> 6);
    lh_ctx_hot(o);
    return o;
}

/* The usual case (round 6): a long block of the normal type of a 44.1 / 48 kHz stream under the settings of the CBR / ABR
 * presets at the default quality -- noise shaping 1 or 2 (the presets up to 128 kb/s scale the scalefactors: 2) with
 * amplification rule 1, no full outer loop, no sfb21 band, no substep shaping.  The stages that run once or more per granule
 * and channel exist again with these as constants (suffix n, the search also m for noise shaping 1: the same source, the
 * fields below pinned before the body runs), which drops the other block types' and presets' code from them. */
LH_DEVFN int
lh_cfg_class(const LhCtx & c)
{
    /* (l21: the 44.1 and 48 kHz band tables; at 32 kHz the bands' padded squares -- LhQS.pad, lh_dev_qloop.h -- would not fit
     * the scratch: those streams take the general stages).  0: none; else the noise shaping of the class */
    int const ok = !LH_IS_LSF && (c.ns == 1 || c.ns == 2) && c.ns_amp == 1 && c.full_outer_loop == 0 && c.sfb21_extra == 0
        && c.l21 <= 418;
    return ok ? c.ns : 0;
}

LH_DEVFN int
lh_granule_is_usual(const LhCtx & c, int block_type, int substep)
{
    return (block_type == LH_NORM_TYPE && (substep & 2) == 0) ? lh_cfg_class(c) : 0;
}

/* the same for a short block (the search stages only; substep shaping and the subblock gains as the presets have them) */
LH_DEVFN int
lh_granule_is_usual_short(const LhCtx & c, int block_type, int substep)
{
    return (block_type == LH_SHORT_TYPE && (substep & 2) == 0 && c.subblock_gain == 1) ? lh_cfg_class(c) : 0;
}

/* nsh: the class's noise shaping (0: a stage that does not look at it) */
LH_DEVFN void
lh_pin_usual(LhCtx & c, int nsh)
{
    if (nsh)
        c.ns = nsh;
    c.ns_amp = 1;
    c.full_outer_loop = 0;
    c.sfb21_extra = 0;
    c.rate8k = 0;
    if (c.l21 > 418)
        __builtin_unreachable();
}

/* (the values lh_init_outer_loop_body gives these fields in the usual case) */
LH_DEVFN void
lh_pin_usual(LhQR & R)
{
    R.block_type = LH_NORM_TYPE;
    R.sfb_lmax = LH_SBPSY_L;
    R.sfb_smin = LH_SBPSY_S;
    R.psy_lmax = LH_SBPSY_L;
    R.psymax = LH_SBPSY_L;
    R.sfbmax = LH_SBPSY_L;
    R.sfbdivide = 11;
    if ((R.substep_shaping & 2) != 0)
        __builtin_unreachable();
}

/* (... in the short-block case of the same presets) */
LH_DEVFN void
lh_pin_usual_short(LhQR & R)
{
    R.block_type = LH_SHORT_TYPE;
    R.sfb_lmax = 0;
    R.sfb_smin = 0;
    R.psy_lmax = 0;
    R.psymax = 3 * LH_SBPSY_S;
    R.sfbmax = 3 * LH_SBPSY_S;
    R.sfbdivide = 3 * LH_SBPSY_S - 18;
    if ((R.substep_shaping & 2) != 0)
        __builtin_unreachable();
}

LH_DEVFN void
lh_rg_put(const LhCtx & c, const LhQR & R, const LhGrR & g)
{
    if (c.lane == 0) {
        lh_lds.rg[c.wave].R = R;
        lh_lds.rg[c.wave].g = g;
    }
    LH_WAVE_SYNC();
}

/* stage the 1904-sample window starting at stream sample `base' into LDS (whole workgroup) */
LH_DEVFN void
lh_stage_window(const LhCtx & c, float (*mf)[LH_MF_NEEDED], long long base)
{
    /* Loads are issued unconditionally at clamped positions and the out-of-stream samples
     * are zeroed afterwards, in batches: a load under a condition is a branch with its own
     * wait, and a frame's window would cost thirty serial HBM round trips per thread. */
    float const scale = c.cfg->pcm_scale;
    long long const last = c.d.nsamples - 1;
    float const mix = c.cfg->pcm_mix;
    float const scale_r = c.cfg->pcm_scale_r;
    if (c.pcmf) {
        /* float pool of the lame_encode_buffer* handle path: the host applied the sample type's
         * scale and the pcm_transform matrix (lame_copy_inbuffer, reference lame.c:1786-1834) */
        for (int t = c.tid; t < 2 * LH_MF_NEEDED; t += LH_NT) {
            int const ch = t >= LH_MF_NEEDED, i = t - ch * LH_MF_NEEDED;
            long long const p = base + i;
            float   v = 0.0f;
            if (p >= 0 && p <= last && p >= c.d.pcm_base)
                v = c.pcmf[(ch == 0 ? c.d.pcm_l : c.d.pcm_r) + (p - c.d.pcm_base)];
            mf[ch][i] = v;
        }
        return;
    }
    if (mix != 0.0f) {
        /* two channels mixed down to one (plain loop: not the configuration the batching below is for) */
        for (int i = c.tid; i < LH_MF_NEEDED; i += LH_NT) {
            long long const p = base + i;
            float   v = 0.0f;
            if (p >= 0 && p <= last && p >= c.d.pcm_base) {
                float const xl = (float) c.pcm[c.d.pcm_l + (p - c.d.pcm_base)];
                float const xr = (float) c.pcm[c.d.pcm_r + (p - c.d.pcm_base)];
                v = xl * scale + xr * mix;
            }
            mf[0][i] = v;
            mf[1][i] = 0.0f;
        }
        return;
    }
    {
        /* The usual frame: the whole window lies inside the stream and inside the pool, and both planes
         * start on an even element -- no clamping, two samples per load, all loads of a thread in
         * flight at once (16 dwords: 952 pairs per channel over 128 threads). */
        long long const lo = base, hi = base + LH_MF_NEEDED - 1;
        long long const ol = c.d.pcm_l + (base - c.d.pcm_base), orr = c.d.pcm_r + (base - c.d.pcm_base);
        int const inside = lo >= 0 && lo >= c.d.pcm_base && hi <= last && ((ol | orr) & 1) == 0;
        if (lh_uni_i(inside)) {
            const uint32_t *pl = (const uint32_t *) (c.pcm + ol), *pr = (const uint32_t *) (c.pcm + orr);
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int const j = c.tid + LH_NT * u;
                int const jj = j < LH_MF_NEEDED / 2 ? j : LH_MF_NEEDED / 2 - 1;
                v[u] = pl[jj];
                v[8 + u] = pr[jj];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int const j = c.tid + LH_NT * u;
                if (j < LH_MF_NEEDED / 2) {
                    lh_f32x2 a, b;
                    a.x = (float) (int16_t) (v[u] & 0xffffu) * scale;
                    a.y = (float) (int16_t) (v[u] >> 16) * scale;
                    b.x = (float) (int16_t) (v[8 + u] & 0xffffu) * scale_r;
                    b.y = (float) (int16_t) (v[8 + u] >> 16) * scale_r;
                    ((lh_f32x2 *) mf[0])[j] = a;
                    ((lh_f32x2 *) mf[1])[j] = b;
                }
            }
            return;
        }
    }
    for (int t0 = c.tid; t0 < 2 * LH_MF_NEEDED; t0 += 6 * LH_NT) {
        int16_t v[6];
#pragma unroll
        for (int u = 0; u < 6; u++) {
            int const t = t0 + u * LH_NT;
            int const tt = t < 2 * LH_MF_NEEDED ? t : 2 * LH_MF_NEEDED - 1;
            int const ch = tt >= LH_MF_NEEDED, i = tt - ch * LH_MF_NEEDED;
            long long p = base + i;
            p = p < 0 ? 0 : (p > last ? last : p);
            p = p < c.d.pcm_base ? c.d.pcm_base : p;    /* never before the pool (also nsamples == 0) */
            v[u] = (c.d.nsamples > 0) ? c.pcm[(ch == 0 ? c.d.pcm_l : c.d.pcm_r) + (p - c.d.pcm_base)] : (int16_t) 0;
        }
#pragma unroll
        for (int u = 0; u < 6; u++) {
            int const t = t0 + u * LH_NT;
            if (t < 2 * LH_MF_NEEDED) {
                int const ch = t >= LH_MF_NEEDED, i = t - ch * LH_MF_NEEDED;
                long long const p = base + i;
                mf[ch][i] = (p < 0 || p > last) ? 0.0f : (float) v[u] * (ch == 0 ? scale : scale_r);
            }
        }
    }
}

/* N samples of both channels from stream sample `base' on, scaled as lh_stage_window scales the frame window (same three
 * sources: s16 pool, float pool of the handle / resampling paths, two channels mixed down), zero outside the stream;
 * NT threads (the analysis kernels: lh_analysis.hip, lh_subband.hip).  LH_STAGE_IDX(i): where sample i of the span goes
 * (even i stay even: the bank swizzles of lh_subband.hip and of lh_analysis.hip's transform buffers); PLAIN = 1: as it comes. */
#ifndef LH_STAGE_IDX
#define LH_STAGE_IDX(i) (i)
#endif
template < int N, int NT, int PLAIN = 0 > LH_DEVFN void
lh_stage_span(const LhCtx & c, float *d0, float *d1, long long base)
{
#define LH_STAGE_AT(i) (PLAIN ? (i) : LH_STAGE_IDX(i))
    float const scale = c.cfg->pcm_scale;
    long long const last = c.d.nsamples - 1;
    float const mix = c.cfg->pcm_mix;
    float const scale_r = c.cfg->pcm_scale_r;
    if (c.pcmf) {
        for (int t = c.tid; t < 2 * N; t += NT) {
            int const ch = t >= N, i = t - ch * N;
            long long const p = base + i;
            float   v = 0.0f;
            if (p >= 0 && p <= last && p >= c.d.pcm_base)
                v = c.pcmf[(ch == 0 ? c.d.pcm_l : c.d.pcm_r) + (p - c.d.pcm_base)];
            (ch ? d1 : d0)[LH_STAGE_AT(i)] = v;
        }
        return;
    }
    if (mix != 0.0f) {
        for (int i = c.tid; i < N; i += NT) {
            long long const p = base + i;
            float   v = 0.0f;
            if (p >= 0 && p <= last && p >= c.d.pcm_base) {
                float const xl = (float) c.pcm[c.d.pcm_l + (p - c.d.pcm_base)];
                float const xr = (float) c.pcm[c.d.pcm_r + (p - c.d.pcm_base)];
                v = xl * scale + xr * mix;
            }
            d0[LH_STAGE_AT(i)] = v;
            d1[LH_STAGE_AT(i)] = 0.0f;
        }
        return;
    }
    {
        /* the usual span: inside the stream and the pool, both planes on an even element -- two samples per load */
        long long const lo = base, hi = base + N - 1;
        long long const ol = c.d.pcm_l + (base - c.d.pcm_base), orr = c.d.pcm_r + (base - c.d.pcm_base);
        int const inside = lo >= 0 && lo >= c.d.pcm_base && hi <= last && ((ol | orr) & 1) == 0;
        static_assert(N % 2 == 0, "pairs");
        constexpr int U = (N / 2 + NT - 1) / NT;
        if (lh_uni_i(inside)) {
            const uint32_t *pl = (const uint32_t *) (c.pcm + ol), *pr = (const uint32_t *) (c.pcm + orr);
            uint32_t v[2 * U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                int const j = c.tid + NT * u;
                int const jj = j < N / 2 ? j : N / 2 - 1;
                v[u] = pl[jj];
                v[U + u] = pr[jj];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                int const j = c.tid + NT * u;
                if (j < N / 2) {
                    lh_f32x2 a, b;
                    a.x = (float) (int16_t) (v[u] & 0xffffu) * scale;
                    a.y = (float) (int16_t) (v[u] >> 16) * scale;
                    b.x = (float) (int16_t) (v[U + u] & 0xffffu) * scale_r;
                    b.y = (float) (int16_t) (v[U + u] >> 16) * scale_r;
                    *(lh_f32x2 *) &d0[LH_STAGE_AT(2 * j)] = a;
                    *(lh_f32x2 *) &d1[LH_STAGE_AT(2 * j)] = b;
                }
            }
            return;
        }
    }
    for (int t = c.tid; t < 2 * N; t += NT) {
        int const ch = t >= N, i = t - ch * N;
        long long const p0 = base + i;
        long long p = p0 < 0 ? 0 : (p0 > last ? last : p0);
        int16_t v;
        p = p < c.d.pcm_base ? c.d.pcm_base : p;        /* never before the pool (also nsamples == 0) */
        v = (c.d.nsamples > 0) ? c.pcm[(ch == 0 ? c.d.pcm_l : c.d.pcm_r) + (p - c.d.pcm_base)] : (int16_t) 0;
        (ch ? d1 : d0)[LH_STAGE_AT(i)] = (p0 < 0 || p0 > last) ? 0.0f : (float) v * (ch == 0 ? scale : scale_r);
    }
}
#undef LH_STAGE_AT

/* sample i of the current frame window */
#ifndef LH_CUSTOM_SMP
LH_DEVFN float
lh_smp(const LhCtx & c, int ch, int i)
{
    return lh_lds.mf[ch][i];
}
#endif

#endif
