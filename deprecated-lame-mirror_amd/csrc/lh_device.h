/*
 * lh_device.h -- layouts shared between the host C++ layer and the HIP kernels:
 * the per-stream carried state that lives in HBM between launches
 * (SURVEY.md 8(a) "carried state"; reference util.h:219-338) and the launch
 * descriptors.
 */
#ifndef LH_DEVICE_H
#define LH_DEVICE_H

#include "lamehip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LH_NPROF 44             /* cycle accumulators per wave (profiling builds) */
#define LH_EMIT_HQ_MAX 256      /* frame headers that can be pending inside a stretch of main data (device bit packer; the host packer and the reference hold 256 as well: MAX_HEADER_BUF) */
#define LH_XMIN_N 61            /* 22 long + 13*3 short values of III_psy_xmin */

typedef struct LhStreamState {
    /* psycho-acoustic model (PsyStateVar_t) */
    float   nb_l1[4][LH_CBANDS];
    float   nb_l2[4][LH_CBANDS];
    float   en[4][LH_XMIN_N + 3];       /* l[22] then s[13][3]; padded to 64 */
    float   thm[4][LH_XMIN_N + 3];
    float   loudness_sq_save[2];
    float   tot_ener[4];
    float   last_en_subshort[4][9];
    int     last_attacks[4];
    int     blocktype_old[2];
    /* ATH auto adjustment */
    float   ath_adjust_factor;
    float   ath_adjust_limit;
    /* polyphase overlap: sub-band samples of the granule preceding the next frame */
    float   sb_prev[2][18 * 32];
    /* frame driver / reservoir / quantiser (EncStateVar_t, QntStateVar_t) */
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
    int     status;                     /* 0 ok; device-detected inconsistencies are reported here */
    int     pad[3];
    /* on-device bit packing (lh_dev_emit.h): where the next header / main-data byte of the stream go,
     * the header starts the cursor has not passed yet, the stuffing pattern's flag, and the packed
     * bits of the frame's granules until the frame is assembled */
    long long em_next_header;
    long long em_cursor;
    long long em_hq[LH_EMIT_HQ_MAX];    /* (16 used by MPEG-1 streams; MPEG-2 / 2.5: see lh_dev_emit.h) */
    int     em_nq;
    int     em_anc_flag;
    uint32_t em_part[2][2][132];
    /* per-wave cycle accumulators, only filled by builds with -DLH_PROF (profiling aid) */
    unsigned long long prof[2][LH_NPROF];
#ifdef LH_DEBUG_DUMP
    /* TEST BUILD ONLY (make dump -> liblamehip_dump.so, tests/test_stage_fixtures.py): what the stages of the last
     * frame handed on, [gr][ch] -- the spectra as the quantiser left them in place (after the mid/side rotation and the
     * short-block reordering), the allowed noise calc_xmin formed and the band energies / thresholds it formed it from,
     * the smoothed perceptual entropies and the bit budgets of the CBR loop */
    float   dbg_xr[2][2][576];
    float   dbg_xmin[2][2][LH_SFBMAX + 1];
    float   dbg_en[2][2][LH_XMIN_N + 3];
    float   dbg_thm[2][2][LH_XMIN_N + 3];
    float   dbg_pe[2][2];
    int     dbg_targ[2][2];
    int     dbg_mean_bits;
    int     dbg_pad[3];
#endif
} LhStreamState;

/* one stream's work for one launch */
typedef struct LhStreamDesc {
    long long pcm_l;            /* element offset of this stream's left/right planes in the PCM pool */
    long long pcm_r;
    long long pcm_base;         /* stream sample index held at pool offset 0 */
    long long nsamples;         /* samples of the stream that exist; beyond that the input is zero */
    long long out_index;        /* first LhFrameOut slot of this launch */
    int     frame_begin;        /* frames [frame_begin, frame_end) are encoded */
    int     frame_end;
    long long bytes_base;       /* device bit packing: this stream's slice of the byte pool ... */
    long long bytes_cap;        /* ... and its size */
    int     flush;              /* pad the last frame out after frame_end - 1 (end of the stream) */
    int     mid_rel;            /* split pipeline: frame f's analysis record is LhMidPools.frames[out_index + mid_rel + (f - frame_begin)];
                                 * 0 unless the launch runs in windows of frames (batch_launch, lh_api.cpp) */
} LhStreamDesc;

/* ---- what the analysis kernels hand to the encode kernel ("mid" data; lh_analysis.hip, lh_subband.hip) ----
 * Everything of the psycho-acoustic model, the polyphase filter and the MDCT that depends on the PCM alone is computed
 * for all frames of a launch at once, at high occupancy, and parked in HBM; the encode kernel's frame prologue only
 * runs the recurrences (pre-echo clamp against the previous granules, ATH level, thresholds under the masking adjustment
 * the last frame's loop left, partition -> band sums, perceptual entropy).  One record (LhMidFrame) per frame of the launch,
 * at LhStreamDesc.out_index + mid_rel + (frame - frame_begin): the payload's index unless the launch runs in windows. */
typedef struct LhMidGr {
    float   peak[4][12];        /* [chn L,R,M,S][k < 9]: largest |high-passed sample| of the granule's nine sub-blocks of 64 samples
                                 * (reference psymodel.c:806-830 before the clamp to 1) */
    float   tot_ener[4];        /* total energy of the long spectrum, bins 11..512 (psymodel.c:690-696) */
    float   loud[2];            /* loudness of L / R (psymodel.c:213-226) */
    float   sub_short_factor[4][3];
    int8_t  ns_attacks[4][4];   /* the attack scan's verdicts (psymodel.c:866-925) */
    int8_t  uselong[2];         /* uselongblock[ch] after the channels are coupled (psymodel.c:926-933, 1265-1286) */
    int8_t  block_type[2];      /* the granule's block type per channel (psymodel.c:1289-1319) */
    int8_t  pad[4];
} LhMidGr;

typedef struct LhMidSmall {
    LhMidGr gr[2];
} LhMidSmall;

/* masking of one pseudo-channel before the recurrences, lane = partition (psymodel.c:1134-1262 up to the pre-echo clamp):
 * v[0] = the partition's energy eb, v[1] = its spread energy x the tonality-dependent weight (ecb), v[2] = the cap
 * max x minval x weight */
typedef struct LhMidMask {
    float   v[3][LH_CBANDS];
} LhMidMask;

typedef struct LhMidLong {
    LhMidMask m[2][4];          /* [gr][chn] */
} LhMidLong;

typedef struct LhMidShort {
    LhMidMask m[2][3][4];       /* [gr][sub-block][chn]; only written for granules with a short-block channel */
} LhMidShort;

typedef struct LhMidXr {
    float   xr[2][2][576];      /* [ch][gr] MDCT spectra of L / R (before the mid/side rotation) */
} LhMidXr;

/* one frame's record: what the pools of a launch are made of */
typedef struct LhMidFrame {
    LhMidSmall small;
    LhMidLong lng;
    LhMidShort shrt;
    LhMidXr xr;
} LhMidFrame;

typedef struct LhMidPools {
    LhMidFrame *frames;         /* [frames of the launch] */
} LhMidPools;

void    lh_state_init(LhStreamState * s, const LhConfig * cfg);

#ifdef __cplusplus
}
#endif
#endif
