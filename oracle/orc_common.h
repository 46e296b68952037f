/*
 * orc_common.h -- CPU restatement of the reference's per-frame encode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by
 * or executed from the product (deprecated-lame-mirror_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status: PINNED against the real reference (oracle/_ref, built from
 * /root/reference by oracle/Makefile) by tests/test_oracle_vs_ref.py: per-frame
 * side information, spectra and the final byte stream are compared bit for bit.
 *
 * Plain C, strict IEEE (-fno-fast-math -ffp-contract=off); every float/double
 * expression keeps the reference's evaluation order and intermediate type.
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <stdint.h>
#include <string.h>
#include <math.h>
#include "lamehip_types.h"

#define ORC_SQRT2 1.41421356237309504880
#define ORC_LOG2  0.69314718055994530942
#define ORC_LOG10 2.30258509299404568402

typedef struct OrcXmin {        /* III_psy_xmin, reference l3side.h:37-40 */
    float   l[LH_SBMAX_L];
    float   s[LH_SBMAX_S][3];
} OrcXmin;

typedef struct OrcRatio {       /* III_psy_ratio, reference l3side.h:42-45 */
    OrcXmin thm;
    OrcXmin en;
} OrcRatio;

/* working image of one granule/channel, gr_info (reference l3side.h:47-84) */
typedef struct OrcGr {
    float   xr[576];
    int     l3_enc[576];
    int     scalefac[LH_SFBMAX];
    float   xrpow_max;
    int     part2_3_length;
    int     big_values;
    int     count1;
    int     global_gain;
    int     scalefac_compress;
    int     block_type;
    int     mixed_block_flag;
    int     table_select[3];
    int     subblock_gain[4];
    int     region0_count;
    int     region1_count;
    int     preflag;
    int     scalefac_scale;
    int     count1table_select;
    int     part2_length;
    int     sfb_lmax;
    int     sfb_smin;
    int     psy_lmax;
    int     sfbmax;
    int     psymax;
    int     sfbdivide;
    int     width[LH_SFBMAX];
    int     window[LH_SFBMAX];
    int     count1bits;
    int     max_nonzero_coeff;
    char    energy_above_cutoff[LH_SFBMAX];
} OrcGr;

typedef struct OrcNoiseResult { /* calc_noise_result, reference quantize_pvt.h:62-69 */
    int     over_count;
    int     tot_noise_i;
    float   over_noise;
    float   tot_noise;
    float   max_noise;
    int     over_SSD;
    int     bits;
} OrcNoiseResult;

typedef struct OrcNoiseData {   /* calc_noise_data, reference quantize_pvt.h:75-82 */
    int     global_gain;
    int     sfb_count1;
    int     step[39];
    float   noise[39];
    float   noise_log[39];
} OrcNoiseData;

/* per-stream carried state (SURVEY.md 8(a) "carried state") */
typedef struct OrcStream {
    const LhConfig *cfg;
    const LhTables *tab;
    /* psycho-acoustics, PsyStateVar_t (reference util.h:219-236) */
    float   nb_l1[4][LH_CBANDS], nb_l2[4][LH_CBANDS];
    float   nb_s1[4][LH_CBANDS], nb_s2[4][LH_CBANDS];
    OrcXmin thm[4];
    OrcXmin en[4];
    float   loudness_sq_save[2];
    float   tot_ener[4];
    float   last_en_subshort[4][9];
    int     last_attacks[4];
    int     blocktype_old[2];
    float   loudness_sq[2][2];
    /* ATH auto adjust */
    float   ath_adjust_factor, ath_adjust_limit;
    /* encoder state, EncStateVar_t (reference util.h:246-300) */
    float   sb_sample[2][2][18][32];
    float   pefirbuf[19];
    int     slot_lag;
    int     ResvSize, ResvMax;
    int     main_data_begin;
    /* quantiser state, QntStateVar_t (reference util.h:318-338) */
    int     OldValue[2], CurrentStep[2];
    float   masking_lower;
    int     substep_shaping;
    int     pseudohalf[LH_SFBMAX];
    int     sfb21_off;          /* VBR_encode_granule switches sfb21_extra off near the top of its range (quantize.c:1267-1270) */
    /* frame results */
    int     frame_init_done;
    int     frame_number;
    int     padding, mode_ext, bitrate_index;
    int     resvDrain_pre, resvDrain_post;
    int     scfsi[2][4];
    OrcGr   tt[2][2];
} OrcStream;

/* orc_psy.c */
int     orc_psycho_anal(OrcStream * S, const float *const buffer[2], int gr_out,
                        OrcRatio masking_ratio[2][2], OrcRatio masking_MS_ratio[2][2],
                        float percep_entropy[2], float percep_MS_entropy[2], float energy[4],
                        int blocktype_d[2]);
float   orc_fast_log2(const LhTables * t, float x);
void    orc_fft_long(const LhTables * t, float x[LH_BLKSIZE], int chn, const float *const buffer[2]);
void    orc_fft_short(const LhTables * t, float x_real[3][LH_BLKSIZE_S], int chn,
                      const float *const buffer[2]);

/* orc_mdct.c */
void    orc_mdct_sub48(OrcStream * S, const float *w0, const float *w1);

/* orc_quant.c */
void    orc_cbr_iteration_loop(OrcStream * S, float pe[2][2], const float ms_ener_ratio[2],
                               const OrcRatio ratio[2][2]);
void    orc_abr_iteration_loop(OrcStream * S, float pe[2][2], const float ms_ener_ratio[2], const OrcRatio ratio[2][2]);
void    orc_vbr_new_iteration_loop(OrcStream * S, float pe[2][2], const OrcRatio ratio[2][2]);
void    orc_vbr_old_iteration_loop(OrcStream * S, float pe[2][2], const float ms_ener_ratio[2], const OrcRatio ratio[2][2]);
float   orc_ath_adjust(const LhTables * t, float a, float x, float athFloor, float ATHfixpoint);

/* orc_frame.c */
void    orc_stream_init(OrcStream * S, const LhConfig * cfg, const LhTables * tab);
int     orc_encode_frame(OrcStream * S, const float *inbuf_l, const float *inbuf_r,
                         LhFrameOut * out);

#endif
