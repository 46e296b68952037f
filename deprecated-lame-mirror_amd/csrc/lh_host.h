/*
 * lh_host.h -- internal host-side interfaces of liblamehip (plain C).
 */
#ifndef LH_HOST_H
#define LH_HOST_H

#include "lamehip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* what the lame_set_* calls of the public API collect before lame_init_params */
typedef struct LhUserParams {
    int     samplerate;
    int     channels;
    int     brate;               /* kbps */
    int     mode;                /* -1 = not set (joint stereo), else LH_MODE_* */
    int     quality;             /* -1 = default (3) */
    int     vbr;                 /* 0 = CBR, 1 / 4 = vbr_mt / vbr_mtrh (the same loop in the reference), 3 = ABR */
    int     vbr_q;               /* VBR quality 0..9 (lame_set_VBR_q), default 4 */
    int     samplerate_out;      /* 0 = let the encoder choose; != samplerate: the input is resampled (lh_resample.c) */
    int     abr_kbps;            /* ABR mean bitrate (lame_set_VBR_mean_bitrate_kbps), default 128 */
    /* frontend-level switches with the reference's defaults (lame.c:2280-2420) */
    int     force_ms, disable_reservoir, error_protection, copyright, original, emphasis, extension;
    int     short_blocks;        /* -1 not set, else short_block_t: 0 allowed, 1 coupled, 2 dispensed, 3 forced */
    int     strict_ISO;          /* 0 MDB_DEFAULT, 1 MDB_STRICT_ISO, 2 MDB_MAXIMUM (default) */
    int     lowpassfreq;         /* 0 = by bitrate / quality, -1 = none, else Hz */
    int     lowpasswidth;        /* -1 = default */
    float   scale, scale_left, scale_right;
    float   vbr_q_frac;          /* lame_set_VBR_quality: the fractional part of -V n.f */
    int     vbr_min_kbps, vbr_max_kbps, vbr_hard_min;   /* -b / -B / -F with VBR or ABR; 0 = not set */
    /* tuning switches of the frontend's "experimental" section, with the reference's "not set" values
     * (lame.c:2340-2400): a value the caller left alone takes the bitrate's / quality's preset value
     * (presets.c SET_OPTION) or the default lame_init_params falls back to (lame.c:1112-1180) */
    float   msfix;               /* -1 */
    int     ATHtype;             /* -1 */
    float   ATHcurve;            /* -1 */
    float   ATH_lower_db;        /* 0; lame_set_ATHlower */
    int     athaa_type;          /* -1 */
    float   athaa_sensitivity;   /* 0 */
    int     ATHonly, ATHshort, noATH;
    float   interChRatio;        /* -1 */
    int     useTemporal;         /* -1 */
    int     highpassfreq;        /* 0 = none, -1 = off */
    int     highpasswidth;       /* -1 = default */
    int     exp_nspsytune;       /* bit 1 safe joint stereo, bits 2.. the bass / alto / treble / sfb21 adjustments */
    int     experimentalY, experimentalZ;
    int     free_format;
    float   compression_ratio;   /* 0 = not set: CBR without a bitrate takes 11.025 (lame.c:622-644) */
    int     preset_kbps;         /* 0, or the bitrate lame_set_preset(8..320 / INSANE) applied its tuning row for at call time */
} LhUserParams;

/* values that only feed table generation */
typedef struct LhInitAux {
    float   lowpass1, lowpass2;
    float   attackthre, attackthre_s;
    int     vbr_q;               /* VBR_q / VBR_q_frac as psymodel_init sees them (4 / 0 for CBR) */
    float   vbr_q_frac;
    float   athaa_sensitivity;
    float   adjust_sfb21_db;     /* exp_nspsytune bits 20..25 (+ the treble adjustment), reference lame.c:1196-1203 */
    float   adjust_bass_db, adjust_alto_db, adjust_treble_db;    /* bits 2..7, 8..13, 14..19 */
    int     athaa_type;          /* ATH->use_adjust: 3 unless lame_set_athaa_type said otherwise */
    float   highpass1, highpass2;        /* edges of the polyphase high-pass, as fractions of the Nyquist rate */
    int     samplerate_in;       /* the caller's rate; LhConfig.samplerate is the output rate */
} LhInitAux;

void    lh_params_default(LhUserParams * p);
int     lh_config_resolve(const LhUserParams * p, LhConfig * c, LhInitAux * aux);
float   lh_abr_preset_scale(int kbps);
int     lh_tables_build(LhConfig * c, const LhInitAux * aux, LhTables * t);

/* number of MP3 frames the reference produces for n input samples per channel
 * when followed by lame_encode_flush (reference lame.c:1671-1775, 2041-2120) */
int     lh_total_frames(long nsamples);
int     lh_total_frames_fs(long nsamples, int samples_per_frame);       /* 576 for MPEG-2 / 2.5 */
const int16_t *lh_bitrate_row(int version);     /* the frame sizes' kb/s of an MPEG version (lh_host_init.c) */

/* ------------------------------------------------------------------ */
/* serial bit packer (reference bitstream.c), host only                */
#define LH_MAX_HEADER_BUF 256
#define LH_MAX_HEADER_LEN 40
#define LH_BS_BUFSIZE (16384 + 147456)   /* LAME_MAXMP3BUFFER, reference lame.h */

/* a frame header + side information waiting for its place in the byte stream */
typedef struct LhQueuedHeader {
    int     due;                 /* stream position (bits) at which it is inserted */
    unsigned char bytes[LH_MAX_HEADER_LEN];
} LhQueuedHeader;

typedef struct LhBitstream {
    unsigned char *buf;          /* finished bytes not yet handed to the caller */
    int     buf_size;
    int     fill;                /* how many there are */
    unsigned long long acc;      /* main-data bits on their way into whole bytes */
    int     acc_bits;
    int     stream_bits;         /* bits in the stream so far (whole bytes, headers included) */
    LhQueuedHeader queue[LH_MAX_HEADER_BUF];
    int     q_in, q_out;         /* ring: next free slot, oldest waiting header */
    int     hdr_len;             /* header + side information, bytes */
    int     stuff_bit;
    int     main_data_begin;     /* packer's own running value, cross-checked with the device's */
    int     error;
} LhBitstream;

int     lh_bs_init(LhBitstream * bs);
int     lh_bs_init_sized(LhBitstream * bs, int size);
/* CRC-16 of a frame's protected part: header bytes 2..3 and 6..len-1 (ISO/IEC 11172-3 2.4.3.1; reference
 * bitstream.c:304-318 CRC_writeheader); the caller stores it big-endian in bytes 4..5 */
unsigned lh_header_crc(const unsigned char *h, int len);
void    lh_bs_free(LhBitstream * bs);
/* appends one frame; returns 0, or <0 when the device payload is inconsistent */
int     lh_bs_format_frame(LhBitstream * bs, const LhConfig * c, const LhTables * t,
                           const LhFrameOut * fo);
void    lh_bs_flush(LhBitstream * bs, const LhConfig * c, const LhFrameOut * last);
/* moves the finished bytes out (reference copy_buffer); -1 if size!=0 and too small */
int     lh_bs_copy(LhBitstream * bs, unsigned char *out, int size);
/* bytes waiting to be copied out */
int     lh_bs_pending(const LhBitstream * bs);


/* ---- Xing/Info + LAME tag bookkeeping (lh_vbrtag.c; reference VbrTag.c) ---- */
#define LH_LAMEHEADERSIZE (4 + 4 + 4 + 4 + 100 + 4 + 9 + 1 + 1 + 8 + 1 + 1 + 3 + 1 + 1 + 2 + 4 + 2 + 2)
#define LH_TAG_BAG 400
typedef struct LhVbrTag {
    int     enabled;
    int     total_frame_size;
    int     sum, seen, want, pos, size;
    int     bag[LH_TAG_BAG];
    unsigned num_frames;
    unsigned long bytes_written;
    uint16_t music_crc;
    int     samplerate_in;       /* source rate for the tag's info byte (lh_tag_init: the output rate) */
    int     radio_gain_on, radio_gain;   /* lame_set_findReplayGain: the title's gain in tenths of a dB */
    int     nogap_total, nogap_current;  /* the frontend's --nogap: this title's place in the set */
} LhVbrTag;

int     lh_tag_init(LhVbrTag * v, const LhConfig * c);
void    lh_tag_add_frame(LhVbrTag * v, int kbps);
int     lh_tag_kbps(int version, int bitrate_index);
void    lh_tag_crc(LhVbrTag * v, const unsigned char *buf, long n);
int     lh_tag_placeholder(const LhVbrTag * v, const LhConfig * c, unsigned char *buf);
int     lh_tag_frame(const LhVbrTag * v, const LhConfig * c, int vbr_q, int enc_padding, int last_mode_ext,
                     unsigned char *buf, long size);
int     lh_end_padding(long nsamples);
int     lh_end_padding_fs(long nsamples, int samples_per_frame);

/* ---- sample rate conversion in front of the encoder (lh_resample.c; reference util.c:483-697) ---- */
#define LH_RS_MAXPHASES 320     /* BPC, reference util.h */
typedef struct LhResampler {
    int     rate_in, rate_out;
    double  ratio;               /* input samples per output sample */
    int     phases, taps;
    double  clock[2];            /* input time of the next block's first sample, per channel */
    float   history[2][34];      /* the last taps + 1 input samples of the previous blocks */
    float   bank[2 * LH_RS_MAXPHASES + 1][34];
} LhResampler;

int     lh_rs_needed(int rate_in, int rate_out);
void    lh_rs_init(LhResampler * r, int rate_in, int rate_out);
int     lh_rs_block(LhResampler * r, int ch, float *out, int want, const float *in, int len, int *used);

/* ---- ReplayGain "radio gain" for the LAME tag (lh_replaygain.c; reference gain_analysis.c) ---- */
#define LH_RG_BINS 12000        /* 0.01 dB steps up to 120 dB */
#define LH_RG_MAX_BLOCK 2404    /* a piece never exceeds a 50 ms window at 48 kHz */
typedef struct LhReplayGain {
    int     rate_index;          /* row of the filter tables; -1 = off */
    long    window, filled;      /* samples per 50 ms window / in the current one */
    double  lsum, rsum;          /* the window's energies so far */
    float   hist_in[2][10], hist_mid[2][10], hist_out[2][10];   /* the last ten samples before / between / after the filters */
    uint32_t bins[LH_RG_BINS];
    float   work[3][10 + LH_RG_MAX_BLOCK];
} LhReplayGain;

int     lh_rg_start(LhReplayGain * g, int samplerate);
int     lh_rg_block(LhReplayGain * g, const float *l, const float *r, int n, int channels);
int     lh_rg_finish(LhReplayGain * g);

#ifdef __cplusplus
}
#endif
#endif
