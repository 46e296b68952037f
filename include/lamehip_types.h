/*
 * lamehip_types.h -- plain-old-data layouts shared by the host C layer, the HIP
 * kernels, the CPU oracle (oracle/) and the reference harness (oracle/_ref).
 *
 * Every struct here is a flat, pointer-free image so that it can be uploaded to
 * HBM with one memcpy.  The reference keeps the same information scattered over
 * SessionConfig_t / PsyConst_t / ATH_t / QntStateVar_t / EncStateVar_t
 * (reference libmp3lame/util.h:166-459) and gr_info / III_side_info_t
 * (reference libmp3lame/l3side.h:47-93).
 */
#ifndef LAMEHIP_TYPES_H
#define LAMEHIP_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dimensions, reference libmp3lame/encoder.h:93-114 */
#define LH_SBMAX_L     22
#define LH_SBMAX_S     13
#define LH_SBPSY_L     21
#define LH_SBPSY_S     12
#define LH_PSFB21      6
#define LH_PSFB12      6
#define LH_SFBMAX      39
#define LH_CBANDS      64
#define LH_BLKSIZE     1024
#define LH_HBLKSIZE    513
#define LH_BLKSIZE_S   256
#define LH_HBLKSIZE_S  129
#define LH_PRECALC     8208      /* IXMAX_VAL+2, reference quantize_pvt.h:24-29 */
#define LH_IXMAX       8206
#define LH_QMAX        257
#define LH_QMAX2       116
#define LH_LARGE_BITS  100000
#define LH_MAX_BITS_PER_CHANNEL 4095
#define LH_MAX_BITS_PER_GRANULE 7680
#define LH_S3_MAX      1280      /* upper bound for the packed spreading matrix */

#define LH_NORM_TYPE   0
#define LH_START_TYPE  1
#define LH_SHORT_TYPE  2
#define LH_STOP_TYPE   3

#define LH_MPG_MD_LR_LR 0
#define LH_MPG_MD_MS_LR 2

/* MPEG modes, reference include/lame.h MPEG_mode */
#define LH_MODE_STEREO       0
#define LH_MODE_JOINT_STEREO 1
#define LH_MODE_DUAL         2
#define LH_MODE_MONO         3

/* polyphase / framing delays, reference libmp3lame/encoder.h:57-83 */
#define LH_ENCDELAY    576
#define LH_POSTDELAY   1152
#define LH_MDCTDELAY   48
#define LH_FFTOFFSET   (224 + LH_MDCTDELAY)
#define LH_MF_START    (LH_ENCDELAY - LH_MDCTDELAY)   /* 528, reference lame.c:2302 */
#define LH_MF_NEEDED   1904                          /* reference lame.c:1627-1648, MPEG-1 */

/* ------------------------------------------------------------------ */
/* resolved per-stream constants (subset of SessionConfig_t)           */
typedef struct LhConfig {
    int     version;              /* 1 = MPEG-1 */
    int     samplerate;           /* of the stream (output rate); another input rate is converted on the host first */
    int     samplerate_index;
    int     bitrate_index;
    int     avg_bitrate;          /* kbps */
    int     mode;                 /* LH_MODE_* */
    int     mode_gr;              /* 2 */
    int     channels;             /* 2 */
    int     vbr;                  /* 0 = vbr_off (CBR), 1 / 4 = vbr_mt / vbr_mtrh, 3 = vbr_abr */
    int     quality;
    int     noise_shaping;
    int     noise_shaping_amp;
    int     noise_shaping_stop;
    int     subblock_gain;
    int     use_best_huffman;
    int     full_outer_loop;
    int     substep_shaping;
    int     quant_comp;
    int     quant_comp_short;
    int     sfb21_extra;
    int     short_blocks;         /* 1 = coupled, 0 = allowed, 2 dispensed, 3 forced (reference lame.h short_block_t) */
    int     use_safe_joint_stereo;
    int     use_temporal_masking;
    int     force_ms;
    int     sideinfo_len;
    int     buffer_constraint;
    int     frac_SpF;
    int     disable_reservoir;
    int     error_protection;
    int     copyright, original, extension, emphasis;
    int     lowpassfreq;
    float   msfix;
    float   ATHfixpoint;
    float   ATH_offset_db;
    float   ATH_offset_factor;
    float   ATHcurve;
    int     ATHtype;
    float   minval;
    float   mask_adjust;
    float   mask_adjust_short;
    float   masking_lower_long;   /* pow(10, mask_adjust*0.1), reference quantize.c:2029 */
    float   masking_lower_short;
    float   pcm_scale;            /* pcm_transform diagonal, reference lame.c:1209-1234 */
    float   interChRatio;
    /* VBR (vbr_mt / vbr_mtrh), reference util.h SessionConfig_t */
    int     vbr_q;
    int     vbr_min_bitrate_index;
    int     vbr_max_bitrate_index;
    int     enforce_min_bitrate;
    /* ABR */
    int     vbr_avg_bitrate_kbps;
    float   compression_ratio;
    /* two input channels mixed down to one: sample = l * pcm_scale + r * pcm_mix (pcm_transform[0][],
     * reference lame.c:1209-1234); 0 otherwise */
    float   pcm_mix;
    float   pcm_scale_r;          /* right channel's factor (pcm_transform[1][1]); pcm_scale is the left one's */
    int     highpassfreq;         /* lame_set_highpassfreq: Hz, 0 = none, -1 = "off" (the tag's -k test) */
    int     ath_flags;            /* bit 0 noATH, bit 1 ATHonly, bit 2 ATHshort (reference util.h SessionConfig_t) */
} LhConfig;

/* partition -> scalefactor-band mapping, PsyConst_CB2SB_t (reference util.h:188-203) */
typedef struct LhPsyBand {
    float   masking_lower[LH_CBANDS];
    float   minval[LH_CBANDS];
    float   rnumlines[LH_CBANDS];
    float   mld_cb[LH_CBANDS];
    float   mld[LH_SBMAX_L];
    float   bo_weight[LH_SBMAX_L];
    int     s3ind[LH_CBANDS][2];
    int     numlines[LH_CBANDS];
    int     bm[LH_SBMAX_L];
    int     bo[LH_SBMAX_L];
    int     npart;
    int     n_sb;
    int     s3_count;
    int     s3_row[LH_CBANDS];    /* offset of row b in s3[] (derived; not in the reference) */
    float   s3[LH_S3_MAX];
} LhPsyBand;

typedef struct LhTables {
    /* scalefactor band boundaries, reference quantize_pvt.c:100-167, lame.c:927-946 */
    int     sfb_l[LH_SBMAX_L + 1];
    int     sfb_s[LH_SBMAX_S + 1];
    int     psfb21[LH_PSFB21 + 1];
    int     psfb12[LH_PSFB12 + 1];
    /* quantiser power tables, reference quantize_pvt.c:171-179, 350-366 */
    int     line_pad0[13];        /* pow43 and vqthr start on 128-byte lines of the device copy (checked in
                                   * lh_dev_common.h): the searches gather mostly small indices from them,
                                   * which then share one cache line instead of straddling two */
    float   pow43[LH_PRECALC];
    float   line_pad1[16];
    /* The second rounding of the quantiser as a comparison (see qthr below) for the VBR noise search, which
     * meets every k.  Offsets of large k round to small positive
     * numbers, so a class yields either {k - 1, k} or {k, k + 1}: the quantised value is
     * k - (a < |vqthr[k]|) + (vqthr[k] carries a minus sign), |vqthr[k]| being the first float of the class
     * that gives the higher value (0 when the whole class does).  lh_tables_init refuses a table where a
     * class would take three values. */
    float   vqthr[LH_PRECALC];
    float   line_pad2[16];
    /* vqthr and the two values of pow43 a class can take, side by side (one 16-byte look-up of the VBR noise search instead of
     * two dependent ones): { |vqthr[k]|, pow43[higher - 1], pow43[higher], 0 }, higher = k or k + 1 as vqthr's sign says --
     * the line's pow43 is [1] when a < [0], else [2] */
    float   vq3[LH_PRECALC][4];
    float   adj43asm[LH_PRECALC];
    float   ipow20[LH_QMAX];
    float   pow20[LH_QMAX + LH_QMAX2 + 1];
    int     bv_scf[576];          /* reference takehiro.c:1334-1375 */
    /* ATH, reference util.h:166-182 */
    float   ath_l[LH_SBMAX_L];
    float   ath_s[LH_SBMAX_S];
    float   ath_psfb21[LH_PSFB21];
    float   ath_psfb12[LH_PSFB12];
    float   ath_cb_l[LH_CBANDS];
    float   ath_cb_s[LH_CBANDS];
    float   ath_eql_w[LH_BLKSIZE / 2];
    float   ath_floor;
    float   ath_decay;
    float   aa_sensitivity_p;
    int     ath_use_adjust;
    float   longfact[LH_SBMAX_L];
    float   shortfact[LH_SBMAX_S];
    /* psycho-acoustic constants, reference psymodel.c:1867-2157 */
    LhPsyBand psy_l;
    LhPsyBand psy_s;
    LhPsyBand psy_l_to_s;
    float   attack_threshold[4];
    float   decay;
    float   ma_max_i1, ma_max_i2;
    /* FFT windows and the twiddle recurrence unrolled (reference fft.c:56-148, 296-310) */
    float   fft_window[LH_BLKSIZE];
    float   fft_window_s[LH_BLKSIZE_S / 2];
    float   fht_tw[4][128][4];    /* [stage][i] -> c1,s1,c2,s2 as produced by the recurrence fft.c:103-144 */
    float   amp_filter[32];       /* reference lame.c:103-190 */
    float   log_table[513];       /* reference util.c:954-972 */
    /* scalefactor band of each of the 576 lines as gr_info.width[] lays them out (derived from
     * sfb_l / sfb_s; long blocks: 22 bands, short blocks: 39 = 13 x 3 windows, window-major) */
    uint8_t sfb_line_l[576];
    uint8_t sfb_line_s[576];
    /* Huffman code lengths as the CBR search reads them (lh_dev_qloop.h): three 16 x 16 grids of words
     * indexed x * 16 + y, each entry the lengths of (up to) three candidate tables, 10 bits each:
     * [0..255] the ESC tables 16.. | 24.. and the count of values == 15, [256..511] tables 13 / 14 / 15,
     * [512..703] the small alphabets side by side (LQ_ORG_* in lh_dev_common.h).  Built by lh_tables_init. */
    uint32_t hgrid[704];
    /* The second rounding of the x^(3/4) quantiser (reference takehiro.c:144-200) as a comparison: for a
     * scaled line a whose first rounding gives k < 256, the quantised value is k - (a < qthr[k]).  The
     * reference's (float) ((double) a + 2^23 + adj43asm[k]) is non-decreasing in a and, with the offsets of
     * these k all negative, takes only the values k - 1 and k on the floats that round to k; qthr[k] is the
     * first float that gives k, found by lh_tables_init with that very expression. */
    float   qthr[256];
    /* The masking addition (reference psymodel.c:294-341) without its quotient.  The reference forms
     * ratio = larger / smaller in float and asks (a) near the diagonal: which of table2's cells
     * i = (int) (fast_log2(ratio) * 16 log10(2)) applies, unless ratio >= ma_max_i1, (b) elsewhere: whether
     * ratio < ma_max_i2.  The index never decreases with the ratio and steps at eight floats r_1 .. r_8
     * (tests/test_quantizer_identity.py walks every float below ma_max_i1), and a correctly rounded quotient
     * is >= a float r exactly when larger > midpoint(pred r, r) x smaller -- a product that is exact in
     * double.  mask_mid[0..7] are those midpoints for r_1 .. r_8, [8] for ma_max_i1, [9] for ma_max_i2;
     * lh_tables_init finds r_j by bisection with the reference's expression. */
    double  mask_mid[10];
    /* The region split of a long block by big_values (reference takehiro.c:1334-1375, bv_scf above) folded with the band
     * edges, as the CBR search reads it: entry big_values / 2 - 1 = region0_count | region1_count << 4 | end of region 0
     * << 8 | end of region 1 << 18 (lines).  Built by lh_tables_init; the kernel copies it into LDS once per frame. */
    uint32_t bvpack[288];
} LhTables;

/* ------------------------------------------------------------------ */
/* one granule of one channel as consumed by the bit packer            */
typedef struct LhGranule {
    int16_t l3_enc[576];          /* quantised magnitudes, sign bit carries (xr < 0) */
    int8_t  scalefac[LH_SFBMAX];  /* -1 = shared through scfsi */
    int8_t  pad0;
    int16_t part2_3_length;
    int16_t part2_length;
    int16_t big_values;
    int16_t count1;
    int16_t global_gain;
    int16_t scalefac_compress;
    int8_t  block_type;
    int8_t  mixed_block_flag;
    int8_t  table_select[3];
    int8_t  subblock_gain[3];
    int8_t  region0_count;
    int8_t  region1_count;
    int8_t  preflag;
    int8_t  scalefac_scale;
    int8_t  count1table_select;
    int8_t  sfbmax;
    int8_t  sfbdivide;
    int8_t  pad1;
    int16_t count1bits;
    int16_t pad2;
} LhGranule;

/* device -> host payload for one frame (what format_bitstream reads, reference bitstream.c:917-985) */
typedef struct LhFrameOut {
    LhGranule gr[2][2];
    int8_t  scfsi[2][4];
    int16_t main_data_begin;
    int16_t resvDrain_pre;
    int16_t resvDrain_post;
    int8_t  bitrate_index;
    int8_t  padding;
    int8_t  mode_ext;
    int8_t  pad[7];              /* explicit: keeps resv_size 4-aligned with no hidden padding */
    int32_t resv_size;            /* ResvSize after ResvFrameEnd: cross-check for the packer */
    int32_t frame_bits;
} LhFrameOut;

#ifdef __cplusplus
}
#endif
#endif
