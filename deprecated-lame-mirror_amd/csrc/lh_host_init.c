/*
 * lh_host_init.c -- host side (plain C, glibc libm): resolve the user settings
 * into the immutable stream constants (LhConfig) and generate every table the
 * HIP kernels consume (LhTables).
 *
 * This is the part of the reference's lame_init_params() that fixes constants
 * for the hot path (reference libmp3lame/lame.c:537-1260, presets.c:215-317,
 * quantize_pvt.c:210-417, psymodel.c:1605-2157, fft.c:296-310, takehiro.c:1334,
 * util.c:197-282,954-972).  All of it uses host libm (pow, exp, cos, atan, log,
 * powf), so it runs on the host once per configuration and the result is
 * uploaded; nothing here is recomputed on the device.  Float/double evaluation
 * order follows the reference expression by expression because a one-ulp table
 * difference changes integer decisions downstream.
 *
 * Supported on this path: MPEG-1 (32/44.1/48 kHz), 2 channels, CBR, stereo or
 * joint stereo, quality 0..9.  Anything else is refused with -1 (never a silent
 * fallback).
 */
#include <math.h>
#include <stddef.h>
#include <string.h>
#include <float.h>

#include "lamehip_types.h"
#include "lh_host.h"
#include "lh_static_tables.h"

#define LH_PI    3.14159265358979323846
#define LH_LOG10 2.30258509299404568402
#define LH_LOG2  0.69314718055994530942
#define LH_DELBARK .34
#define LH_NSATHSCALE 100

/* ---------------------------------------------------------------------- */
/* bitrate helpers (reference util.c:344-382, 420-470)                      */
static const int full_bitrate_table[17] =
    { 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320 };

static int
nearest_full_index(int bitrate)
{
    int     lower = 16, upper = 16, lower_k = 320, upper_k = 320, b;
    for (b = 0; b < 16; b++) {
        int     m = bitrate > full_bitrate_table[b + 1] ? bitrate : full_bitrate_table[b + 1];
        if (m != bitrate) {
            upper_k = full_bitrate_table[b + 1];
            upper = b + 1;
            lower_k = full_bitrate_table[b];
            lower = b;
            break;
        }
    }
    if ((upper_k - bitrate) > (bitrate - lower_k))
        return lower;
    return upper;
}

/* the bitrate row of a stream: MPEG-1, MPEG-2, or -- only where a bitrate is LOOKED FOR -- the shorter MPEG-2.5 row
 * (reference util.c:376-440 FindNearestBitrate / BitrateIndex switch to it below 16 kHz; frame sizes always come from
 * the version's row, bitstream.c:60-88) */
const int16_t *
lh_bitrate_row(int version)
{
    return version ? lh_bitrate_mpeg1 : lh_bitrate_mpeg2;
}

static const int16_t *
bitrate_search_row(int version, int samplerate)
{
    return (samplerate < 16000) ? lh_bitrate_mpeg25 : lh_bitrate_row(version);
}

/* reference util.c:376-400 */
static int
find_nearest_bitrate(int b, int version, int samplerate)
{
    const int16_t *row = bitrate_search_row(version, samplerate);
    int     i, best = row[1];
    for (i = 2; i <= 14; i++) {
        if (row[i] > 0) {
            int     d0 = row[i] - b, d1 = best - b;
            if (d0 < 0)
                d0 = -d0;
            if (d1 < 0)
                d1 = -d1;
            if (d0 < d1)
                best = row[i];
        }
    }
    return best;
}

/* reference util.c:421-440: -1 when the row does not hold the rate */
static int
bitrate_index_of(int b, int version, int samplerate)
{
    const int16_t *row = bitrate_search_row(version, samplerate);
    int     i;
    for (i = 0; i <= 14; i++)
        if (row[i] > 0 && row[i] == b)
            return i;
    return -1;
}

/* ABR/CBR tuning table (reference presets.c:232-250); columns used by this path */
typedef struct {
    int     kbps;
    int     safejoint;
    float   nsmsfix, st_lrm, st_s, scale, masking_adj, ath_lower, ath_curve, interch;
    int     sfscale;
} LhAbrRow;

static const LhAbrRow abr_map[17] = {
    {8, 0, 0, 6.60, 145, 0.95, 0, -30.0, 11, 0.0012, 1},
    {16, 0, 0, 6.60, 145, 0.95, 0, -25.0, 11, 0.0010, 1},
    {24, 0, 0, 6.60, 145, 0.95, 0, -20.0, 11, 0.0010, 1},
    {32, 0, 0, 6.60, 145, 0.95, 0, -15.0, 11, 0.0010, 1},
    {40, 0, 0, 6.60, 145, 0.95, 0, -10.0, 11, 0.0009, 1},
    {48, 0, 0, 6.60, 145, 0.95, 0, -10.0, 11, 0.0009, 1},
    {56, 0, 0, 6.60, 145, 0.95, 0, -6.0, 11, 0.0008, 1},
    {64, 0, 0, 6.60, 145, 0.95, 0, -2.0, 11, 0.0008, 1},
    {80, 0, 0, 6.60, 145, 0.95, 0, .0, 8, 0.0007, 1},
    {96, 0, 2.50, 6.60, 145, 0.95, 0, 1.0, 5.5, 0.0006, 1},
    {112, 0, 2.25, 6.60, 145, 0.95, 0, 2.0, 4.5, 0.0005, 1},
    {128, 0, 1.95, 6.40, 140, 0.95, 0, 3.0, 4, 0.0002, 1},
    {160, 1, 1.79, 6.00, 135, 0.95, -2, 5.0, 3.5, 0, 1},
    {192, 1, 1.49, 5.60, 125, 0.97, -4, 7.0, 3, 0, 0},
    {224, 1, 1.25, 5.20, 125, 0.98, -6, 9.0, 2, 0, 0},
    {256, 1, 0.97, 5.20, 125, 1.00, -8, 10.0, 1, 0, 0},
    {320, 1, 0.90, 5.20, 125, 1.00, -10, 12.0, 0, 0, 0}
};

/* lowpass by bitrate (reference lame.c:211-229) */
static const int lowpass_map[17] = {
    2000, 3700, 3900, 5500, 7000, 7500, 10000, 11000, 13500, 15100, 15600, 17000, 17500, 18600,
    19400, 19700, 20500
};

void
lh_params_default(LhUserParams * p)
{
    memset(p, 0, sizeof(*p));
    p->samplerate = 44100;
    p->channels = 2;
    p->brate = 0;               /* reference lame.c:2290: the bitrate then follows from the compression ratio (11.025) */
    p->mode = -1;
    p->quality = -1;
    p->vbr = 0;
    p->vbr_q = 4;               /* reference lame.c:2360 */
    p->abr_kbps = 128;          /* reference lame.c:2361 */
    p->original = 1;
    p->short_blocks = -1;
    p->strict_ISO = 2;          /* MDB_MAXIMUM, reference lame.c:2341 */
    p->lowpasswidth = -1;
    p->scale = p->scale_left = p->scale_right = 1.0f;
    p->samplerate_out = 0;
    /* "not set" values of the tuning switches (reference lame.c:2355-2384) */
    p->msfix = -1;
    p->ATHtype = -1;
    p->ATHcurve = -1;
    p->athaa_type = -1;
    p->interChRatio = -1;
    p->useTemporal = -1;
    p->highpasswidth = -1;
}

/* quality -> algorithm switches (reference lame.c:362-477) */
static int
apply_quality(LhConfig * c, int noise_shaping_in, int *attack_unused)
{
    (void) attack_unused;
    c->noise_shaping = noise_shaping_in;
    c->subblock_gain = -1;
    c->substep_shaping = 0;
    switch (c->quality) {
    default:
    case 9:
        c->noise_shaping = 0;
        c->noise_shaping_amp = 0;
        c->noise_shaping_stop = 0;
        c->use_best_huffman = 0;
        c->full_outer_loop = 0;
        break;
    case 8:
        c->quality = 7;
        /* fall through */
    case 7:
        c->noise_shaping = 0;
        c->noise_shaping_amp = 0;
        c->noise_shaping_stop = 0;
        c->use_best_huffman = 0;
        c->full_outer_loop = 0;
        if (c->vbr == 1 || c->vbr == 4)
            c->full_outer_loop = -1;    /* selects the guessed scalefactor search, reference lame.c:387 */
        break;
    case 6:
    case 5:
        if (c->noise_shaping == 0)
            c->noise_shaping = 1;
        c->noise_shaping_amp = 0;
        c->noise_shaping_stop = 0;
        if (c->subblock_gain == -1)
            c->subblock_gain = 1;
        c->use_best_huffman = 0;
        c->full_outer_loop = 0;
        break;
    case 4:
        if (c->noise_shaping == 0)
            c->noise_shaping = 1;
        c->noise_shaping_amp = 0;
        c->noise_shaping_stop = 0;
        if (c->subblock_gain == -1)
            c->subblock_gain = 1;
        c->use_best_huffman = 1;
        c->full_outer_loop = 0;
        break;
    case 3:
        if (c->noise_shaping == 0)
            c->noise_shaping = 1;
        c->noise_shaping_amp = 1;
        c->noise_shaping_stop = 1;
        if (c->subblock_gain == -1)
            c->subblock_gain = 1;
        c->use_best_huffman = 1;
        c->full_outer_loop = 0;
        break;
    case 2:
        if (c->noise_shaping == 0)
            c->noise_shaping = 1;
        if (c->substep_shaping == 0)
            c->substep_shaping = 2;
        c->noise_shaping_amp = 1;
        c->noise_shaping_stop = 1;
        if (c->subblock_gain == -1)
            c->subblock_gain = 1;
        c->use_best_huffman = 1;
        c->full_outer_loop = 0;
        break;
    case 1:
        if (c->noise_shaping == 0)
            c->noise_shaping = 1;
        if (c->substep_shaping == 0)
            c->substep_shaping = 2;
        c->noise_shaping_amp = 2;
        c->noise_shaping_stop = 1;
        if (c->subblock_gain == -1)
            c->subblock_gain = 1;
        c->use_best_huffman = 1;
        c->full_outer_loop = 0;
        break;
    case 0:
        if (c->noise_shaping == 0)
            c->noise_shaping = 1;
        if (c->substep_shaping == 0)
            c->substep_shaping = 2;
        c->noise_shaping_amp = 2;
        c->noise_shaping_stop = 1;
        if (c->subblock_gain == -1)
            c->subblock_gain = 1;
        c->use_best_huffman = 1;
        c->full_outer_loop = 1;
        break;
    }
    return 0;
}


/* VBR (vbr_mt / vbr_mtrh) preset rows, reference presets.c:106-126 (vbr_mt_psy_switch_map) */
typedef struct {
    int     expY;
    float   st_lrm, st_s, masking_adj, masking_adj_short, ath_lower, ath_curve, ath_sensitivity, interch;
    int     safejoint, sfb21mod;
    float   msfix, minval, ath_fixpoint;
} LhVbrPreset;

static const LhVbrPreset vbr_mt_map[11] = {
    {0, 4.20, 25.0, -6.8, -6.8, 7.1, 1, 0, 0, 2, 31, 1.000, 5, 100},
    {0, 4.20, 25.0, -4.8, -4.8, 5.4, 1.4, -1, 0, 2, 27, 1.122, 5, 98},
    {0, 4.20, 25.0, -2.6, -2.6, 3.7, 2.0, -3, 0, 2, 23, 1.288, 5, 97},
    {1, 4.20, 25.0, -1.6, -1.6, 2.0, 2.0, -5, 0, 2, 18, 1.479, 5, 96},
    {1, 4.20, 25.0, -0.0, -0.0, 0.0, 2.0, -8, 0, 2, 12, 1.698, 5, 95},
    {1, 4.20, 25.0, 1.3, 1.3, -6, 3.5, -11, 0, 2, 8, 1.950, 5, 94.2},
    {1, 4.50, 100.0, 2.2, 2.3, -12.0, 6.0, -14, 0, 2, 4, 2.239, 3, 93.9},
    {1, 4.80, 200.0, 2.7, 2.7, -18.0, 9.0, -17, 0, 2, 0, 2.570, 1, 93.6},
    {1, 5.30, 300.0, 2.8, 2.8, -21.0, 10.0, -23, 0.0002, 0, 0, 2.951, 0, 93.3},
    {1, 6.60, 300.0, 2.8, 2.8, -23.0, 11.0, -25, 0.0006, 0, 0, 3.388, 0, 93.3},
    {1, 25.00, 300.0, 2.8, 2.8, -25.0, 12.0, -27, 0.0025, 0, 0, 3.500, 0, 93.3}
};

/* the same for vbr_rh, reference presets.c:90-104 (vbr_old_switch_map) */
static const LhVbrPreset vbr_old_map[11] = {
    {0, 5.20, 125.0, -4.2, -6.3, 4.8, 1, 0, 0, 2, 21, 0.97, 5, 100},
    {0, 5.30, 125.0, -3.6, -5.6, 4.5, 1.5, 0, 0, 2, 21, 1.35, 5, 100},
    {0, 5.60, 125.0, -2.2, -3.5, 2.8, 2, 0, 0, 2, 21, 1.49, 5, 100},
    {1, 5.80, 130.0, -1.8, -2.8, 2.6, 3, -4, 0, 2, 20, 1.64, 5, 100},
    {1, 6.00, 135.0, -0.7, -1.1, 1.1, 3.5, -8, 0, 2, 0, 1.79, 5, 100},
    {1, 6.40, 140.0, 0.5, 0.4, -7.5, 4, -12, 0.0002, 0, 0, 1.95, 5, 100},
    {1, 6.60, 145.0, 0.67, 0.65, -14.7, 6.5, -19, 0.0004, 0, 0, 2.30, 5, 100},
    {1, 6.60, 145.0, 0.8, 0.75, -19.7, 8, -22, 0.0006, 0, 0, 2.70, 5, 100},
    {1, 6.60, 145.0, 1.2, 1.15, -27.5, 10, -23, 0.0007, 0, 0, 0, 5, 100},
    {1, 6.60, 145.0, 1.6, 1.6, -36, 11, -25, 0.0008, 0, 0, 0, 5, 100},
    {1, 6.60, 145.0, 2.0, 2.0, -36, 12, -25, 0.0008, 0, 0, 0, 5, 100}
};

/* output rate the reference picks for a lowpass and an input rate (reference lame.c:273-345):
 * the MPEG rate that suits the input, lowered as far as the lowpass allows, but never below the
 * next MPEG rate above the input */
static int
suggested_samplerate(int lp, int samplerate_in)
{
    static const int mpeg_rates[9] = { 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000 };
    /* a lowpass at or below edge[i] moves the suggestion to mpeg_rates[i + 1] */
    static const int edge[8] = { 15960, 15250, 11220, 9970, 7230, 5420, 4510, 3970 };
    int     suggested = 44100, i;
    for (i = 0; i < 9; i++)
        if (samplerate_in >= mpeg_rates[i]) {
            suggested = mpeg_rates[i];
            break;
        }
    if (lp == -1)
        return suggested;
    for (i = 0; i < 8; i++)
        if (lp <= edge[i])
            suggested = mpeg_rates[i + 1];
    if (samplerate_in < suggested) {
        /* the smallest MPEG rate that is not below the input */
        for (i = 8; i >= 0; i--)
            if (samplerate_in <= mpeg_rates[i])
                return mpeg_rates[i];
        return 48000;
    }
    return suggested;
}

/* the stream's rate: version, index into the header's rate field and granules per frame (reference util.c:443-487
 * SmpFrqIndex, lame.c:797); sideinfo_len follows it (lame.c:949-952) */
static int
set_output_rate(LhConfig * c, int rate)
{
    switch (rate) {
    case 44100: c->version = 1; c->samplerate_index = 0; break;
    case 48000: c->version = 1; c->samplerate_index = 1; break;
    case 32000: c->version = 1; c->samplerate_index = 2; break;
    case 22050: c->version = 0; c->samplerate_index = 0; break;
    case 24000: c->version = 0; c->samplerate_index = 1; break;
    case 16000: c->version = 0; c->samplerate_index = 2; break;
    case 11025: c->version = 0; c->samplerate_index = 0; break;
    case 12000: c->version = 0; c->samplerate_index = 1; break;
    case 8000:  c->version = 0; c->samplerate_index = 2; break;
    default:
        return -1;
    }
    c->samplerate = rate;
    c->mode_gr = (rate <= 24000) ? 1 : 2;
    if (c->mode_gr == 2)
        c->sideinfo_len = (c->channels == 1) ? 4 + 17 : 4 + 32;
    else
        c->sideinfo_len = (c->channels == 1) ? 4 + 9 : 4 + 17;
    return 0;
}

static void
lowpass_edges(LhConfig * c, LhInitAux * aux, int width)
{
    int const lp = c->lowpassfreq;
    aux->lowpass1 = 0;
    aux->lowpass2 = 0;
    if (lp > 0 && lp < c->samplerate / 2) {
        aux->lowpass2 = 2. * lp;
        if (width >= 0) {       /* lame_set_lowpasswidth, reference lame.c:880-884 */
            aux->lowpass1 = 2. * (lp - width);
            if (aux->lowpass1 < 0)
                aux->lowpass1 = 0;
        }
        else
            aux->lowpass1 = (1 - 0.00) * 2. * lp;
        aux->lowpass1 /= c->samplerate;
        aux->lowpass2 /= c->samplerate;
    }
}

/* input scale of the ABR / CBR tuning row for a bitrate (lame_set_preset applies it at call time on top of
 * lame_init_params, as the reference does: presets.c:296) */
float
lh_abr_preset_scale(int kbps)
{
    return abr_map[nearest_full_index(kbps)].scale;
}

/* smallest / largest frame size VBR and ABR may pick (-b / -B / -F; reference lame.c:1064-1085) */
static int
vbr_bitrate_limits(const LhUserParams * p, LhConfig * c)
{
    int     r;
    c->vbr_min_bitrate_index = 1;
    c->vbr_max_bitrate_index = 14;
    if (c->samplerate < 16000)
        c->vbr_max_bitrate_index = 8;   /* 64 kb/s: MPEG-2.5 (reference lame.c:1068-1069) */
    if (p->vbr_min_kbps) {
        r = bitrate_index_of(find_nearest_bitrate(p->vbr_min_kbps, c->version, c->samplerate), c->version, c->samplerate);
        if (r < 0)
            return -1;
        c->vbr_min_bitrate_index = r;
    }
    if (p->vbr_max_kbps) {
        r = bitrate_index_of(find_nearest_bitrate(p->vbr_max_kbps, c->version, c->samplerate), c->version, c->samplerate);
        if (r < 0)
            return -1;
        c->vbr_max_bitrate_index = r;
    }
    c->enforce_min_bitrate = p->vbr_hard_min;
    /* -b above -B: the reference takes it and, in ABR, then ends every frame with a mean of 0 bits (the loop that picks the
     * frame size never runs, quantize.c:1963-1968), i.e. with a reservoir that only shrinks -- not a stream; refused here */
    if (c->vbr_min_bitrate_index > c->vbr_max_bitrate_index)
        return -1;
    return 0;
}

/* The caller's tuning switches on top of what the bitrate's / quality's preset row put into c / aux
 * (reference presets.c:34-42: a preset value only fills an option the caller left at its "not set" value;
 * lame.c:1112-1203: what remains unset falls back to a default).  `nspsytune' comes in with the preset's
 * bits and goes out with the caller's merged in; vbr_new: vbr_mt / vbr_mtrh force ATH type 5 and leave the
 * temporal masking effect off unless asked for. */
static int
config_apply_tuning(const LhUserParams * p, LhConfig * c, LhInitAux * aux, int preset_nspsytune, int preset_sfb21mod,
                    int vbr_new)
{
    int     nsp = p->exp_nspsytune | (preset_nspsytune & 2);
    if (preset_sfb21mod > 0 && ((nsp >> 20) & 63) == 0)
        nsp |= preset_sfb21mod << 20;
    nsp |= 1;
    if (p->free_format || p->experimentalZ || p->ATHonly)
        return -1;              /* free format frames, forced short-block analysis, ATH-only thresholds: not on this path */
    if (fabs(p->msfix - (-1)) > 0)
        c->msfix = p->msfix;
    if (fabs(p->ATH_lower_db - 0) > 0) {
        c->ATH_offset_db = 0 - p->ATH_lower_db;
        c->ATH_offset_factor = powf(10.f, c->ATH_offset_db * 0.1f);
    }
    if (fabs(p->ATHcurve - (-1)) > 0)
        c->ATHcurve = p->ATHcurve;
    if (fabs(p->athaa_sensitivity - 0) > 0)
        aux->athaa_sensitivity = p->athaa_sensitivity;
    if (fabs(p->interChRatio - (-1)) > 0)
        c->interChRatio = p->interChRatio;
    if (c->interChRatio < 0)
        c->interChRatio = 0;
    if (!vbr_new && p->ATHtype >= 0)
        c->ATHtype = p->ATHtype;
    if (p->useTemporal >= 0)
        c->use_temporal_masking = p->useTemporal;
    aux->athaa_type = (p->athaa_type < 0) ? 3 : p->athaa_type;
    c->ath_flags = (p->noATH ? 1 : 0) | (p->ATHshort ? 4 : 0);
    c->use_safe_joint_stereo = nsp & 2;
    {
        /* six bits each, two's complement, quarter dB (reference lame.c:1181-1203) */
        float   db[4];
        int     k;
        for (k = 0; k < 4; k++) {
            db[k] = (nsp >> (2 + 6 * k)) & 63;
            if (db[k] >= 32.f)
                db[k] -= 64.f;
            db[k] *= 0.25f;
        }
        aux->adjust_bass_db = db[0];
        aux->adjust_alto_db = db[1];
        aux->adjust_treble_db = db[2];
        aux->adjust_sfb21_db = db[3] + db[2];
    }
    /* the polyphase high-pass (reference lame.c:859-874) */
    c->highpassfreq = p->highpassfreq;
    aux->highpass1 = aux->highpass2 = 0;
    if (c->highpassfreq > 0) {
        aux->highpass1 = 2. * c->highpassfreq;
        if (p->highpasswidth >= 0)
            aux->highpass2 = 2. * (c->highpassfreq + p->highpasswidth);
        else
            aux->highpass2 = (1 + 0.00) * 2. * c->highpassfreq;
        aux->highpass1 /= c->samplerate;
        aux->highpass2 /= c->samplerate;
    }
    return 0;
}

/* vbr_mt / vbr_mtrh settings (reference lame.c:661-692, 730-744, 770-776, 972-1004, 1064-1094;
 * presets.c:146-213 apply_vbr_preset with every option still at its default); vbr_rh (c->vbr == 2) differs
 * in its tables (lame.c:717-729, presets.c:90-104), takes neither the quality -> output rate mapping nor ATH
 * type 5, keeps the temporal masking effect on, and its lowpass stops at 20.5 kHz (lame.c:773-775) */
static int
config_resolve_vbr(const LhUserParams * p, LhConfig * c, LhInitAux * aux)
{
    static const int lp_by_q_new[11] = { 24000, 19500, 18500, 18000, 17500, 17000, 16500, 15600, 15200, 7230, 3950 };
    static const int lp_by_q_old[11] = { 19500, 19000, 18600, 18000, 17500, 16000, 15600, 14900, 12500, 10000, 3950 };
    int const old = (c->vbr == 2);
    const int *const lp_by_q = old ? lp_by_q_old : lp_by_q_new;
    const LhVbrPreset *const map = old ? vbr_old_map : vbr_mt_map;
    int     vbr_q = p->vbr_q, samplerate_out = p->samplerate_out, lowpassfreq = p->lowpassfreq, i;
    float   vbr_q_frac = p->vbr_q_frac;
    LhVbrPreset P, Q;
    float   x;
    int     nspsytune = 1;

    if (vbr_q < 0)
        vbr_q = 0;
    if (vbr_q > 9)
        vbr_q = 9;
    if (samplerate_out == 0 && !old) {
        /* VBR scale -> internal quality + output rate (reference lame.c:661-692); rows 2.. of its table */
        static const struct { int sr_a; float qa, qb, ta, tb; } m[7] = {
            {32000, 6.5, 8.0, 5.2, 6.5}, {24000, 8.0, 8.5, 5.2, 6.0}, {22050, 8.5, 9.01, 5.2, 6.5},
            {16000, 9.01, 9.4, 4.9, 6.5}, {12000, 9.4, 9.6, 4.5, 6.0}, {11025, 9.6, 9.9, 5.1, 6.5},
            {8000, 9.9, 10., 4.9, 6.5}
        };
        /* Above a rate's knee (qa) the quality scale is stretched: [0, qa) keeps the input rate with
         * qualities 0 .. ta, [qa, qb) switches the output rate to sr_a with qualities ta .. tb.  The
         * first row whose interval holds the value decides. */
        float const qval = vbr_q + vbr_q_frac;
        for (i = 0; i < 7; ++i) {
            double  stretched = -1;
            int const here = (p->samplerate == m[i].sr_a), above = (p->samplerate >= m[i].sr_a);
            int const inside = (m[i].qa <= qval && qval < m[i].qb);
            if (here && qval < m[i].qa) {
                stretched = qval / m[i].qa;
                stretched = stretched * m[i].ta;
            }
            if (above && inside) {
                float const dq = m[i].qb - m[i].qa, dt = m[i].tb - m[i].ta;
                stretched = m[i].ta + dt * (qval - m[i].qa) / dq;
                samplerate_out = m[i].sr_a;
                lowpassfreq = lowpassfreq == 0 ? -1 : lowpassfreq;
            }
            if (stretched >= 0) {
                vbr_q = (int) stretched;
                vbr_q_frac = stretched - vbr_q;
            }
            if (above && inside)
                break;
        }
    }
    if (lowpassfreq == 0) {
        double  a = lp_by_q[vbr_q], b = lp_by_q[vbr_q + 1], mm = vbr_q_frac;
        double  lowpass = a + mm * (b - a);
        lowpassfreq = lowpass;
    }
    if (samplerate_out == 0) {
        if (2 * lowpassfreq > p->samplerate)
            lowpassfreq = p->samplerate / 2;
        samplerate_out = suggested_samplerate(lowpassfreq, p->samplerate);
    }
    if (set_output_rate(c, samplerate_out) != 0)
        return -1;
    {
        int const top = old ? 20500 : 24000;
        lowpassfreq = (top < lowpassfreq) ? top : lowpassfreq;
    }
    lowpassfreq = (samplerate_out / 2 < lowpassfreq) ? samplerate_out / 2 : lowpassfreq;
    c->lowpassfreq = lowpassfreq;
    lowpass_edges(c, aux, p->lowpasswidth);

    c->bitrate_index = 1;
    c->avg_bitrate = 0;         /* gfp->brate stays 0 in VBR mode */
    c->buffer_constraint = 7680 * (c->version + 1);     /* strict_ISO = MDB_MAXIMUM */
    c->use_temporal_masking = old ? 1 : 0;

    /* preset row, interpolated by the fractional quality (reference presets.c:146-171) */
    P = map[vbr_q];
    Q = map[vbr_q + 1];
    x = vbr_q_frac;
    {
        /* every tuning value moves towards the next row's by the fractional quality */
        static const size_t field[11] = {
            offsetof(LhVbrPreset, st_lrm), offsetof(LhVbrPreset, st_s), offsetof(LhVbrPreset, masking_adj),
            offsetof(LhVbrPreset, masking_adj_short), offsetof(LhVbrPreset, ath_lower), offsetof(LhVbrPreset, ath_curve),
            offsetof(LhVbrPreset, ath_sensitivity), offsetof(LhVbrPreset, interch),
            offsetof(LhVbrPreset, msfix), offsetof(LhVbrPreset, minval), offsetof(LhVbrPreset, ath_fixpoint)
        };
        int     k;
        for (k = 0; k < 11; k++) {
            float  *mine = (float *) ((char *) &P + field[k]);
            float const next = *(const float *) ((const char *) &Q + field[k]);
            *mine = *mine + x * (next - *mine);
        }
        /* the one integer among them (dB tenths for the band above sfb 21): truncated after the blend */
        P.sfb21mod = (int) (P.sfb21mod + x * (Q.sfb21mod - P.sfb21mod));
    }
    c->quant_comp = 9;
    c->quant_comp_short = 9;
    aux->attackthre = P.st_lrm;
    aux->attackthre_s = P.st_s;
    c->mask_adjust = P.masking_adj;
    c->mask_adjust_short = P.masking_adj_short;
    c->ATHtype = old ? 4 : 5;
    c->ATH_offset_db = 0 - P.ath_lower;
    c->ATH_offset_factor = powf(10.f, c->ATH_offset_db * 0.1f);
    c->ATHcurve = P.ath_curve;
    aux->athaa_sensitivity = P.ath_sensitivity;
    c->interChRatio = (P.interch > 0) ? P.interch : 0;
    if (P.safejoint > 0)
        nspsytune |= 2;
    c->msfix = P.msfix;
    c->minval = P.minval;
    {
        double const xs = fabs(p->scale);
        double const y = (xs > 0.f) ? (10.f * log10(xs)) : 0.f;
        c->ATHfixpoint = P.ath_fixpoint - y;
    }
    if (config_apply_tuning(p, c, aux, nspsytune, P.sfb21mod, !old) != 0)
        return -1;
    {
        float   db = c->mask_adjust - 0;
        c->masking_lower_long = pow(10.0, db * 0.1);
        db = c->mask_adjust_short - 0;
        c->masking_lower_short = pow(10.0, db * 0.1);
    }
    /* quality levels of the new VBR code (reference lame.c:985-992) */
    c->quality = (p->quality < 0) ? 3 : p->quality;
    if (old) {
        /* the old loop needs the noise measurements: level 6 at least (reference lame.c:1017-1027) */
        if (c->quality > 6)
            c->quality = 6;
    }
    else {
        if (c->quality < 5)
            c->quality = 0;
        if (c->quality > 7)
            c->quality = 7;
    }
    apply_quality(c, 0, 0);
    c->sfb21_extra = (P.expY || p->experimentalY) ? 0 : (samplerate_out > 44000);
    c->short_blocks = (c->mode == LH_MODE_MONO || c->mode == LH_MODE_DUAL) ? 0 : 1;
    c->pcm_scale = p->scale * p->scale_left;
    c->pcm_scale_r = p->scale * p->scale_right;
    c->disable_reservoir = 0;
    c->frac_SpF = 0;
    c->vbr_q = vbr_q;
    aux->vbr_q = vbr_q;
    aux->vbr_q_frac = vbr_q_frac;
    if (vbr_bitrate_limits(p, c) != 0)
        return -1;
    {
        /* reference lame.c:828-836 */
        static const float cmp[10] = { 5.7, 6.5, 7.3, 8.2, 10, 11.9, 13, 14, 15, 16.5 };
        c->compression_ratio = cmp[vbr_q];
    }
    c->vbr_avg_bitrate_kbps = 128;      /* VBR_mean_bitrate_kbps default, kept inside the limits (lame.c:1086-1091) */
    if (c->vbr_avg_bitrate_kbps > lh_bitrate_row(c->version)[c->vbr_max_bitrate_index])
        c->vbr_avg_bitrate_kbps = lh_bitrate_row(c->version)[c->vbr_max_bitrate_index];
    if (c->vbr_avg_bitrate_kbps < lh_bitrate_row(c->version)[c->vbr_min_bitrate_index])
        c->vbr_avg_bitrate_kbps = lh_bitrate_row(c->version)[c->vbr_min_bitrate_index];
    return 0;
}

static int config_resolve_inner(const LhUserParams * p, LhConfig * c, LhInitAux * aux);

/* the frontend-level switches that only overwrite resolved constants */
static int
config_apply_switches(const LhUserParams * p, LhConfig * c, LhInitAux * aux)
{
    (void) aux;
    c->copyright = p->copyright != 0;
    c->original = p->original != 0;
    c->extension = p->extension != 0;
    c->emphasis = p->emphasis & 3;
    c->disable_reservoir = p->disable_reservoir != 0;
    if (p->force_ms) {
        if (c->mode != LH_MODE_JOINT_STEREO)
            return -1;          /* the frontend's -m f: joint stereo with every frame M/S */
        c->force_ms = 1;
    }
    if (p->error_protection) {
        c->error_protection = 1;
        c->sideinfo_len += 2;
    }
    if (p->short_blocks >= 0) {
        /* reference lame.c:1113-1130: "allowed" becomes "coupled" for the stereo modes */
        int     sb = p->short_blocks;
        if (sb == 0 && (c->mode == LH_MODE_JOINT_STEREO || c->mode == LH_MODE_STEREO))
            sb = 1;
        c->short_blocks = sb;
    }
    switch (p->strict_ISO) {    /* get_max_frame_buffer_size_by_constraint, reference bitstream.c:91-131 */
    case 0:
        c->buffer_constraint = 8 * 1440;
        break;
    case 1:
        c->buffer_constraint = 8 * ((c->version + 1) * 72000 * 320 / c->samplerate);
        break;
    default:
        c->buffer_constraint = 7680 * (c->version + 1);
        break;
    }
    return 0;
}

int
lh_config_resolve(const LhUserParams * p, LhConfig * c, LhInitAux * aux)
{
    int     rc = config_resolve_inner(p, c, aux);
    if (rc == 0)
        rc = config_apply_switches(p, c, aux);
    if (rc == 0 && p->channels == 2 && c->channels == 1) {
        /* two channels in, one out: the transform's first row averages them (reference lame.c:1224-1229) */
        float const m00 = c->pcm_scale, m01 = 0.0f * c->pcm_scale, m10 = 0.0f * c->pcm_scale_r, m11 = c->pcm_scale_r;
        c->pcm_scale = 0.5f * (m00 + m10);
        c->pcm_mix = 0.5f * (m01 + m11);
        c->pcm_scale_r = 0;
    }
    return rc;
}

static int
config_resolve_inner(const LhUserParams * p, LhConfig * c, LhInitAux * aux)
{
    int     r;
    float   scale, ath_lower_db, maskingadjust, maskingadjust_short;
    int     noise_shaping = 0, ratio_kbps = 128, pinned_out = 0, brate_asked = 128;

    memset(c, 0, sizeof(*c));
    memset(aux, 0, sizeof(*aux));
    if (p->channels != 1 && p->channels != 2)
        return -1;
    if (p->vbr < 0 || p->vbr > 4)
        return -1;
    if (p->samplerate <= 0)
        return -1;
    /* the output rate follows from the lowpass below when the caller left it open; the input rate
     * only matters to that choice and to the resampler (lh_resample.c) */
    aux->samplerate_in = p->samplerate;
    c->version = 1;
    c->mode_gr = 2;
    c->vbr = p->vbr;
    c->mode = (p->mode < 0) ? LH_MODE_JOINT_STEREO : p->mode;
    if (p->channels == 1)
        c->mode = LH_MODE_MONO; /* one input channel: reference lame.c:598-601 */
    if (c->mode != LH_MODE_JOINT_STEREO && c->mode != LH_MODE_STEREO && c->mode != LH_MODE_MONO
        && c->mode != LH_MODE_DUAL)
        return -1;
    c->channels = (c->mode == LH_MODE_MONO) ? 1 : 2;
    c->force_ms = 0;
    c->original = 1;
    /* VBR_q also reaches the CBR path (psymodel_init's masking_lower slope, the tag's quality byte) */
    aux->vbr_q = (p->vbr_q < 0) ? 0 : (p->vbr_q > 9 ? 9 : p->vbr_q);
    aux->vbr_q_frac = p->vbr_q_frac;    /* (-V 4.7 before --preset 128: lame_set_VBR_quality's fraction stays) */
    aux->athaa_sensitivity = 0;
    aux->adjust_sfb21_db = 0;
    c->vbr_q = aux->vbr_q;
    if (c->vbr == 1 || c->vbr == 2 || c->vbr == 4)
        return config_resolve_vbr(p, c, aux);

    {
        /* The reference's order (lame.c:606-660, 700-778, 904-915, 1064-1091): what was asked for -- a bitrate, a mean, or
         * a compression ratio, which also pins the output rate -- gives the lowpass, the lowpass the output rate the caller
         * left open, the output rate the MPEG version, and only then is the bitrate rounded to a frame size of that version's
         * row and are the VBR limits looked up. */
        int     mean = p->abr_kbps, brate = 0, asked, lp;
        double  lowpass;
        if (c->vbr == 3) {
            /* ABR: any mean; apply_abr_preset clamps it to 8..320 (reference presets.c:268-272) */
            if (p->samplerate_out == 0 && (mean < 8 || mean > 320))
                return -1;      /* the reference applies no preset at all there (presets.c:411-416) */
            mean = mean > 320 ? 320 : mean;
            mean = mean < 8 ? 8 : mean;
        }
        else {
            double  ratio = p->compression_ratio;
            brate = (p->brate == 0 && p->abr_kbps != 128) ? p->abr_kbps : p->brate;     /* lame.c:606-607 */
            if (brate == 0 && ratio == 0)
                ratio = 11.025; /* lame.c:622-626 */
            if (ratio > 0) {
                /* the bitrate that gives the ratio at the MPEG rate just above 97 % of the input rate (lame.c:629-644) */
                static const int mpeg_rates[9] = { 8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000 };
                int     out = p->samplerate_out, i;
                LhConfig tmp = *c;
                if (out == 0) {
                    int const f = (int) (0.97 * p->samplerate);
                    out = 48000;
                    for (i = 0; i < 9; i++)
                        if (f <= mpeg_rates[i]) {
                            out = mpeg_rates[i];
                            break;
                        }
                }
                if (set_output_rate(&tmp, out) != 0)
                    return -1;
                brate = out * 16 * c->channels / (1.e3 * ratio);
                brate = find_nearest_bitrate(brate, tmp.version, out);  /* (this route rounds at once, lame.c:642-643) */
                pinned_out = out;
            }
        }
        {
            /* an output rate that is known already bounds the mean (lame.c:645-660; VBR_mean_bitrate_kbps is 128 unless
             * the caller set it, so this only ever shows in ABR) */
            int const out = p->samplerate_out ? p->samplerate_out : pinned_out;
            if (out && c->vbr == 3) {
                int const lo = (out < 32000) ? 8 : 32, hi = (out < 16000) ? 64 : (out < 32000) ? 160 : 320;
                mean = mean < lo ? lo : (mean > hi ? hi : mean);
            }
        }
        asked = (c->vbr == 3) ? mean : brate;
        ratio_kbps = asked;     /* (lame.c:778-784 come before the rounding) */
        /* lowpass (reference lame.c:194-260, 700-760, 846-862) */
        lowpass = lowpass_map[nearest_full_index(asked)];
        if (c->mode == LH_MODE_MONO)
            lowpass *= 1.5;     /* reference lame.c:758-759 */
        lp = (p->lowpassfreq != 0) ? p->lowpassfreq : (int) lowpass;   /* lame_set_lowpassfreq: Hz, -1 = none */
        /* an output rate the caller left open follows from the lowpass (optimum_samplefreq, reference lame.c:273-345,
         * 762-767); when it differs from the input rate the input is resampled in front of the encoder */
        {
            int     out = p->samplerate_out ? p->samplerate_out : pinned_out;
            if (out == 0) {
                if (2 * lp > p->samplerate)
                    lp = p->samplerate / 2;
                out = suggested_samplerate(lp, p->samplerate);
            }
            if (set_output_rate(c, out) != 0)
                return -1;
        }
        if (ratio_kbps <= 0)
            return -1;
        c->compression_ratio = c->samplerate * 16 * c->channels / (1.e3 * ratio_kbps);
        if (lp > 20500)
            lp = 20500;
        if (lp > c->samplerate / 2)
            lp = c->samplerate / 2;
        c->lowpassfreq = lp;
        lowpass_edges(c, aux, p->lowpasswidth);
        if (c->vbr == 3) {
            c->avg_bitrate = mean;
            c->bitrate_index = 1;
            c->vbr_avg_bitrate_kbps = mean;
            if (vbr_bitrate_limits(p, c) != 0)
                return -1;
            /* the mean stays inside the limits (reference lame.c:1086-1091; after compression_ratio was formed) */
            if (c->vbr_avg_bitrate_kbps > lh_bitrate_row(c->version)[c->vbr_max_bitrate_index])
                c->vbr_avg_bitrate_kbps = lh_bitrate_row(c->version)[c->vbr_max_bitrate_index];
            if (c->vbr_avg_bitrate_kbps < lh_bitrate_row(c->version)[c->vbr_min_bitrate_index])
                c->vbr_avg_bitrate_kbps = lh_bitrate_row(c->version)[c->vbr_min_bitrate_index];
        }
        else {
            /* lame.c:904-915 */
            c->avg_bitrate = find_nearest_bitrate(brate, c->version, c->samplerate);
            c->bitrate_index = bitrate_index_of(c->avg_bitrate, c->version, c->samplerate);
            c->vbr_avg_bitrate_kbps = c->avg_bitrate;   /* lame_set_VBR_mean_bitrate_kbps(brate), lame.c:1043 */
        }
        (void) brate_asked;
    }
    if (c->bitrate_index <= 0)
        return -1;

    c->buffer_constraint = 7680 * (c->version + 1);     /* MDB_MAXIMUM, reference bitstream.c:91-131 */

    /* preset for the bitrate (reference presets.c:215-317) */
    r = nearest_full_index(c->avg_bitrate);
    {
        /* lame_set_preset(n) applied row n's values at call time already; lame_init_params then applies the final
         * bitrate's row, which only fills what is still at its "not set" value (SET_OPTION, presets.c:34-42): msfix, the
         * short-block thresholds, the ATH curve and the inter-channel ratio stay the first row's, masking adjustment and
         * ATH lowering too unless the first row's are 0 (= not set); safe joint stereo and sfscale are only ever switched
         * on; minval and the input scale follow the final row (`--preset insane -b 96', `--preset cbr 160 --comp 11') */
        int const r1 = p->preset_kbps ? nearest_full_index(p->preset_kbps) : r;
        int const rm = (abr_map[r1].masking_adj != 0) ? r1 : r;
        int const ra = (abr_map[r1].ath_lower != 0) ? r1 : r;
        if (abr_map[r].safejoint > 0 || abr_map[r1].safejoint > 0)
            c->use_safe_joint_stereo = 2;
        if (abr_map[r].sfscale > 0 || abr_map[r1].sfscale > 0)
            noise_shaping = 2;
        c->quant_comp = 9;
        c->quant_comp_short = 9;
        c->msfix = abr_map[r1].nsmsfix;
        aux->attackthre = abr_map[r1].st_lrm;
        aux->attackthre_s = abr_map[r1].st_s;
        scale = p->scale * abr_map[r].scale;
        maskingadjust = abr_map[rm].masking_adj;
        if (abr_map[rm].masking_adj > 0)
            maskingadjust_short = abr_map[rm].masking_adj * .9;
        else
            maskingadjust_short = abr_map[rm].masking_adj * 1.1;
        ath_lower_db = abr_map[ra].ath_lower;
        c->ATHcurve = abr_map[r1].ath_curve;
        c->interChRatio = abr_map[r1].interch;
        c->minval = 5. * (abr_map[r].kbps / 320.);
    }

    c->mask_adjust = maskingadjust;
    c->mask_adjust_short = maskingadjust_short;
    {
        /* reference quantize.c:2016-2029: FLOAT db = mask_adjust - 0; pow(10.0, db * 0.1) */
        float   db = c->mask_adjust - 0;
        c->masking_lower_long = pow(10.0, db * 0.1);
        db = c->mask_adjust_short - 0;
        c->masking_lower_short = pow(10.0, db * 0.1);
    }

    c->quality = (p->quality < 0) ? 3 : p->quality;
    if (c->quality > 9)
        c->quality = 9;
    apply_quality(c, noise_shaping, 0);
    c->sfb21_extra = 0;
    c->short_blocks = (c->mode == LH_MODE_MONO || c->mode == LH_MODE_DUAL) ? 0 : 1;  /* coupled for stereo / joint stereo, reference lame.c:1134-1137 */
    c->use_temporal_masking = 1;
    c->ATHtype = 4;
    c->ATH_offset_db = 0 - ath_lower_db;
    c->ATH_offset_factor = powf(10.f, c->ATH_offset_db * 0.1f);
    c->ATHfixpoint = 0;
    if (config_apply_tuning(p, c, aux, c->use_safe_joint_stereo, 0, 0) != 0)
        return -1;
    c->pcm_scale = scale * p->scale_left;
    c->pcm_scale_r = scale * p->scale_right;
    c->disable_reservoir = 0;
    c->frac_SpF = (c->vbr == 0) ? (int) (((c->version + 1) * 72000L * c->avg_bitrate) % c->samplerate) : 0;
    return 0;
}

/* ====================================================================== */
/* Tables, organised per table.  Every value must equal the reference's to the last bit (a one-ulp
 * difference changes integer decisions downstream), so each expression keeps the reference's mix
 * of float and double operands and its order of operations; what is this file's own is the
 * decomposition: small pure functions for the curves, one builder per table, shared walkers for
 * "minimum over the lines of a band" and "partition of the spectrum". */

/* ---- the threshold in quiet ------------------------------------------------------------ */

/* one of the reference's curve variants (util.c:197-264): a shape parameter, the frequency range
 * it is clamped to (kHz) and a level shift */
typedef struct QuietCurve {
    float   shape, khz_lo, khz_hi;
    int     lift_db;
} QuietCurve;

static QuietCurve
quiet_curve(const LhConfig * c)
{
    QuietCurve q = { 0.f, 0.1f, 24.0f, 0 };
    switch (c->ATHtype) {
    case 0:
        q.shape = 9;
        break;
    case 1:
        q.shape = -1;
        break;
    case 3:
        q.shape = 1;
        q.lift_db = 6;
        break;
    case 4:
        q.shape = c->ATHcurve;
        break;
    case 5:
        q.shape = c->ATHcurve;
        q.khz_lo = 3.41f;
        q.khz_hi = 16.1f;
        break;
    default:                   /* 2 and anything unknown: shape 0 */
        break;
    }
    return q;
}

/* level of the threshold in quiet at `hz' in dB (negative hz: the level at the curve's minimum,
 * 3.41 kHz); four terms in double: the low-frequency rise, the ear-canal dip, a bump near 8.7 kHz
 * and the high-frequency rise whose weight is the shape parameter */
static float
quiet_level_db(const LhConfig * c, float hz)
{
    QuietCurve const q = quiet_curve(c);
    float   khz = hz, level;
    double  low, dip, bump, high;
    if (khz < -.3)
        khz = 3410;
    khz /= 1000;
    khz = (q.khz_lo > khz) ? q.khz_lo : khz;
    khz = (q.khz_hi < khz) ? q.khz_hi : khz;
    low = 3.640 * pow(khz, -0.8);
    dip = 6.800 * exp(-0.6 * pow(khz - 3.4, 2.0));
    bump = 6.000 * exp(-0.15 * pow(khz - 8.7, 2.0));
    high = (0.6 + 0.04 * q.shape) * 0.001 * pow(khz, 4.0);
    level = low - dip + bump + high;
    if (q.lift_db)
        level = level + q.lift_db;
    return level;
}

/* the same as the energy the quantiser compares MDCT lines with (reference quantize_pvt.c:210-228) */
static float
quiet_energy(const LhConfig * c, float hz)
{
    float   db = quiet_level_db(c, hz);
    db -= (c->ATHfixpoint > 0) ? c->ATHfixpoint : LH_NSATHSCALE;
    db += c->ATH_offset_db;
    return powf(10.0f, db * 0.1f);
}

/* the quietest line of [from, to) when `lines' MDCT lines span half the sampling rate */
static float
quietest_line(const LhConfig * c, int from, int to, int lines)
{
    float const rate = c->samplerate;
    float   least = 1e37f;     /* (an empty band keeps it: reference machine.h:135 FLOAT_MAX, as the reference is built here) */
    int     k;
    for (k = from; k < to; k++) {
        float const e = quiet_energy(c, k * rate / (2 * lines));
        least = (least < e) ? least : e;
    }
    return least;
}

/* ATH per scalefactor band: long, short (times the band width: three windows share it) and the
 * partial bands above sfb21 / sfb12 (reference quantize_pvt.c:230-321) */
static void
build_band_thresholds(const LhConfig * c, LhTables * t)
{
    int     b;
    for (b = 0; b < LH_SBMAX_L; b++)
        t->ath_l[b] = quietest_line(c, t->sfb_l[b], t->sfb_l[b + 1], 576);
    for (b = 0; b < LH_PSFB21; b++)
        t->ath_psfb21[b] = quietest_line(c, t->psfb21[b], t->psfb21[b + 1], 576);
    for (b = 0; b < LH_SBMAX_S; b++) {
        t->ath_s[b] = quietest_line(c, t->sfb_s[b], t->sfb_s[b + 1], 192);
        t->ath_s[b] *= (t->sfb_s[b + 1] - t->sfb_s[b]);
    }
    for (b = 0; b < LH_PSFB12; b++) {
        t->ath_psfb12[b] = quietest_line(c, t->psfb12[b], t->psfb12[b + 1], 192);
        t->ath_psfb12[b] *= (t->sfb_s[13] - t->sfb_s[12]);
    }
    t->ath_floor = 10. * log10(quiet_energy(c, -1.));
    if (c->ath_flags & 1) {
        /* lame_set_noATH: every band's threshold in quiet at -200 dB (reference quantize_pvt.c:294-307) */
        for (b = 0; b < LH_SBMAX_L; b++)
            t->ath_l[b] = 1E-20;
        for (b = 0; b < LH_PSFB21; b++)
            t->ath_psfb21[b] = 1E-20;
        for (b = 0; b < LH_SBMAX_S; b++)
            t->ath_s[b] = 1E-20;
        for (b = 0; b < LH_PSFB12; b++)
            t->ath_psfb12[b] = 1E-20;
    }
}

/* ---- quantiser tables ------------------------------------------------------------------- */

/* x^(4/3), the rounding offsets of the x^(3/4) quantiser and the two step tables
 * (reference quantize_pvt.c:350-366) */
static void
build_power_tables(LhTables * t)
{
    int     k;
    t->pow43[0] = 0.0;
    t->adj43asm[0] = 0.0;
    for (k = 1; k < LH_PRECALC; k++)
        t->pow43[k] = pow((float) k, 4.0 / 3.0);
    for (k = 1; k < LH_PRECALC; k++)
        t->adj43asm[k] = k - 0.5 - pow(0.5 * (t->pow43[k - 1] + t->pow43[k]), 0.75);
    for (k = 0; k < LH_QMAX; k++)
        t->ipow20[k] = pow(2.0, (double) (k - 210) * -0.1875);
    for (k = 0; k <= LH_QMAX + LH_QMAX2; k++)
        t->pow20[k] = pow(2.0, (double) (k - 210 - LH_QMAX2) * 0.25);
}

/* The device derives steps from a few mantissas and a power of two instead of reading these
 * tables (lh_dev_qloop.h): that needs pow20[k + 4] == 2 pow20[k] and ipow20[k + 16] == ipow20[k] / 8
 * to hold exactly, which they do whenever the host's pow() is correctly rounded here.  Checked
 * once per table build; a libm that breaks it makes lame_init_params fail instead of producing
 * different bytes. */
static int
power_tables_scale_exactly(const LhTables * t)
{
    int     k;
    for (k = 0; k + 4 <= LH_QMAX + LH_QMAX2; k++)
        if (t->pow20[k + 4] != 2.0f * t->pow20[k])
            return 0;
    for (k = 0; k + 16 < LH_QMAX; k++)
        if (t->ipow20[k + 16] != t->ipow20[k] * 0.125f)
            return 0;
    return 1;
}

/* Region split of a long block by big_values (reference takehiro.c:38-88, 1334-1375): per number of
 * bands below big_values, a first guess for the band counts of regions 0 and 1, which is then
 * lowered until the region ends at or below big_values. */
static const signed char region_guess[23][2] = {
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 1}, {1, 1}, {1, 1}, {1, 2}, {2, 2}, {2, 3},
    {2, 3}, {3, 4}, {3, 4}, {3, 4}, {4, 5}, {4, 5}, {4, 6}, {5, 6}, {5, 6}, {5, 7}, {6, 7},
    {6, 7}
};

/* Code lengths of tables t, t + 1, t + 2 at alphabet position idx, 10 bits each; n = how many of them
 * are candidates for the same region maximum (the others repeat the first, which a strict comparison
 * never prefers). */
static uint32_t
grid_entry(int t, int n, int idx)
{
    uint32_t const first = lh_ht_len[lh_ht_offset[t] + idx];
    uint32_t const second = (n >= 2) ? lh_ht_len[lh_ht_offset[t + 1] + idx] : first;
    uint32_t const third = (n >= 3) ? lh_ht_len[lh_ht_offset[t + 2] + idx] : first;
    return first | (second << 10) | (third << 20);
}

/* The Huffman length grids of the CBR search (LhTables.hgrid).  The small alphabets share one 16-column
 * grid: tables 10-12 in rows 0..7 / columns 0..7, 7-9 in rows 0..5 / columns 8..13, 5-6 in rows 8..11 /
 * columns 0..3, 2-3 in rows 8..10 / columns 4..6, table 1 in rows 8..9 / columns 8..9; every other
 * cell is zero (one of them serves regions whose maximum is 0). */
static void
build_huffman_grids(LhTables * t)
{
    int     i;
    for (i = 0; i < 256; i++) {
        uint32_t const esc = lh_largetbl[i];    /* lengths with tables 16.. << 16 | with tables 24.. */
        uint32_t const fifteens = (uint32_t) (((i >> 4) == 15) + ((i & 15) == 15));
        t->hgrid[i] = (esc >> 16) | ((esc & 0xffffu) << 10) | (fifteens << 20);
        t->hgrid[256 + i] = grid_entry(13, 3, i);
    }
    for (i = 0; i < 192; i++) {
        int const row = i >> 4, col = i & 15;
        uint32_t v = 0;
        if (row < 8 && col < 8)
            v = grid_entry(10, 3, row * 8 + col);
        else if (row < 6 && col >= 8 && col < 14)
            v = grid_entry(7, 3, row * 6 + (col - 8));
        else if (row >= 8 && row < 12 && col < 4)
            v = grid_entry(5, 2, (row - 8) * 4 + col);
        else if (row >= 8 && row < 11 && col >= 4 && col < 7)
            v = grid_entry(2, 2, (row - 8) * 3 + (col - 4));
        else if (row >= 8 && row < 10 && col >= 8 && col < 10)
            v = grid_entry(1, 1, (row - 8) * 2 + (col - 8));
        t->hgrid[512 + i] = v;
    }
}

/* The quantiser's second rounding for a fixed offset: what the reference computes from a scaled line a
 * once the first rounding has picked adj43asm[k] (reference takehiro.c:166-190) */
static int
second_rounding(float a, float offset)
{
    double const shifted = (double) a + 8388608.0;
    union { float f; uint32_t u; } r;
    r.f = (float) (shifted + offset);
    return (int) (r.u - 0x4B000000u);
}

/* LhTables.qthr: per first rounding k the smallest float whose second rounding reaches k.  Positive
 * floats ascend with their bit patterns and second_rounding() never decreases with a, so the first
 * pattern is found by bisection between k - 0.75 (gives k - 1) and k + 0.75 (gives at least k). */
static void
build_quant_thresholds(LhTables * t)
{
    int     k;
    t->qthr[0] = 0.0f;          /* offset 0: every line of the class quantises to 0 */
    for (k = 1; k < 256; k++) {
        union { float f; uint32_t u; } below, reach, mid;
        below.f = (float) k - 0.75f;
        reach.f = (float) k + 0.75f;
        while (reach.u - below.u > 1u) {
            mid.u = below.u + (reach.u - below.u) / 2u;
            if (second_rounding(mid.f, t->adj43asm[k]) >= k)
                reach = mid;
            else
                below = mid;
        }
        t->qthr[k] = reach.f;
    }
}

/* first rounding of the quantiser (reference takehiro.c:166-190) */
static int
first_rounding(float a)
{
    union { float f; uint32_t u; } r;
    r.f = (float) ((double) a + 8388608.0);
    return (int) (r.u - 0x4B000000u);
}

/* LhTables.vqthr: the floats whose first rounding is k form a run of bit patterns [first, last]; the
 * second rounding ascends over it.  Returns 0 when a class takes more than two values or leaves
 * {k - 1, k, k + 1}. */
static int
build_vbr_quant_thresholds(LhTables * t)
{
    int     k;
    for (k = 0; k < LH_PRECALC; k++) {
        union { float f; uint32_t u; } first, last, below, reach, mid;
        int     v_lo, v_hi, upper;
        /* ends of the class: bisect the first rounding around k - 0.5 and k + 0.5 */
        if (k == 0)
            first.u = 0u;
        else {
            below.f = (float) k - 0.75f;
            reach.f = (float) k + 0.25f;
            while (reach.u - below.u > 1u) {
                mid.u = below.u + (reach.u - below.u) / 2u;
                if (first_rounding(mid.f) >= k)
                    reach = mid;
                else
                    below = mid;
            }
            first = reach;
        }
        below.f = (float) k + 0.25f;
        reach.f = (float) k + 0.75f;
        while (reach.u - below.u > 1u) {
            mid.u = below.u + (reach.u - below.u) / 2u;
            if (first_rounding(mid.f) > k)
                reach = mid;
            else
                below = mid;
        }
        last = below;
        v_lo = second_rounding(first.f, t->adj43asm[k]);
        v_hi = second_rounding(last.f, t->adj43asm[k]);
        if (v_hi > v_lo + 1 || v_lo < k - 1 || v_hi > k + 1 || v_hi < k)
            return 0;
        upper = (v_hi == k + 1);        /* the class takes {k, k + 1}, else {k - 1, k} */
        if (v_lo == v_hi)
            reach.u = 0u;       /* no line lies below 0: the whole class takes the higher value */
        else {
            below = first;
            reach = last;
            while (reach.u - below.u > 1u) {
                mid.u = below.u + (reach.u - below.u) / 2u;
                if (second_rounding(mid.f, t->adj43asm[k]) >= v_hi)
                    reach = mid;
                else
                    below = mid;
            }
        }
        t->vq3[k][0] = reach.f;
        t->vq3[k][1] = (v_hi >= 1 && v_hi - 1 < LH_PRECALC) ? t->pow43[v_hi - 1] : 0.0f;
        t->vq3[k][2] = (v_hi < LH_PRECALC) ? t->pow43[v_hi] : 0.0f;
        t->vq3[k][3] = 0.0f;
        if (upper)
            reach.u |= 0x80000000u;
        t->vqthr[k] = reach.f;
    }
    return 1;
}

/* cell of the masking-addition table for a ratio of two maskers: the reference's table-driven log2
 * (util.c:976-1001) scaled to sixteenths of a decade and truncated (psymodel.c:327-333) */
static int
mask_table_cell(const LhTables * t, float ratio)
{
    union { float f; uint32_t u; } r;
    int     mantissa, slot;
    float   whole, along, lg;
    r.f = ratio;
    mantissa = (int) (r.u & 0x7fffffu);
    whole = (float) ((int) ((r.u >> 23) & 0xffu) - 127);
    along = (float) (mantissa & 16383);
    along *= 1.0f / 16384;
    slot = mantissa >> 14;
    lg = whole;
    lg += t->log_table[slot] * (1.0f - along) + t->log_table[slot + 1] * along;
    return (int) (lg * (0.69314718055994530942 / 2.30258509299404568402 * 16.0f));
}

/* midpoint of a float and its predecessor: quotients above it round to the float or beyond */
static double
rounding_boundary(float v)
{
    union { float f; uint32_t u; } below;
    below.f = v;
    below.u -= 1u;
    return 0.5 * ((double) v + (double) below.f);
}

/* LhTables.mask_mid (needs log_table, ma_max_i1 / _i2).  Returns 0 when the cells do not step up one
 * at a time between ratio 1 (cell 0) and ma_max_i1 (cell 8 just below it). */
static int
build_mask_boundaries(LhTables * t)
{
    int     j;
    union { float f; uint32_t u; } below, reach, mid, top;
    top.f = t->ma_max_i1;
    top.u -= 1u;
    if (mask_table_cell(t, 1.0f) != 0 || mask_table_cell(t, top.f) != 8)
        return 0;
    for (j = 1; j <= 8; j++) {
        below.f = 1.0f;
        reach = top;
        while (reach.u - below.u > 1u) {
            mid.u = below.u + (reach.u - below.u) / 2u;
            if (mask_table_cell(t, mid.f) >= j)
                reach = mid;
            else
                below = mid;
        }
        if (mask_table_cell(t, reach.f) != j || mask_table_cell(t, below.f) != j - 1)
            return 0;
        t->mask_mid[j - 1] = rounding_boundary(reach.f);
    }
    t->mask_mid[8] = rounding_boundary(t->ma_max_i1);
    t->mask_mid[9] = rounding_boundary(t->ma_max_i2);
    return 1;
}

/* largest count <= guess with edge[base + count] <= limit; the guess itself when there is none */
static int
lower_until_inside(const int *edge, int base, int guess, int limit)
{
    int     n = guess;
    while (n >= 0 && edge[base + n] > limit)
        n--;
    return n < 0 ? guess : n;
}

/* Returns 0 when a table breaks what the device count relies on (lq_count, lh_dev_qloop.h): the first region of a granule
 * ends within the first 64 pairs -- one register slot of a wave -- for every block type, and with the long-block band edges
 * of 44.1 / 48 kHz (edge 21 at or below line 418) the second region of a normal block ends within the first 128. */
static int
build_region_split(LhTables * t)
{
    int     bv, ok = 1;
    for (bv = 2; bv <= 576; bv += 2) {
        int     below = 1, r0;
        while (t->sfb_l[below] < bv)
            below++;            /* bands that start below big_values */
        r0 = lower_until_inside(t->sfb_l, 1, region_guess[below][0], bv);
        t->bv_scf[bv - 2] = r0;
        t->bv_scf[bv - 1] = lower_until_inside(t->sfb_l, r0 + 2, region_guess[below][1], bv);
    }
    for (bv = 2; bv <= 576; bv += 2) {
        /* the same folded with the band edges (LhTables.bvpack) */
        int const r0 = t->bv_scf[bv - 2], r1 = t->bv_scf[bv - 1];
        int const a1 = t->sfb_l[r0 + 1], a2 = t->sfb_l[(r0 + r1 + 2 < LH_SBMAX_L) ? r0 + r1 + 2 : LH_SBMAX_L];
        t->bvpack[bv / 2 - 1] = (uint32_t) r0 | ((uint32_t) r1 << 4) | ((uint32_t) a1 << 8) | ((uint32_t) a2 << 18);
        if (a1 > 128 || (t->sfb_l[LH_SBPSY_L] <= 418 && a2 > 256))
            ok = 0;
    }
    /* (start / stop blocks: region 0 ends at band edge 8; short blocks: at three times short edge 3) */
    if (t->sfb_l[7 + 1] > 128 || 3 * t->sfb_s[3] > 128)
        ok = 0;
    return ok;
}

/* per-band weights on the masking threshold: four groups of bands (bass, alto, treble, the band
 * above the last scalefactor band), each 10^(dB / 10) (reference quantize_pvt.c:375-417), shifted by
 * the bass / alto / treble / sfb21 adjustments of lame_set_exp_nspsytune (the VBR presets set the last) */
static void
build_band_weights(const LhInitAux * aux, LhTables * t)
{
    static float const group_db_long[4] = { -0.500f, -0.250f, -0.025f, +0.500f };
    static float const group_db_short[4] = { -2.000f, -1.000f, -0.050f, +0.500f };
    static int const last_long[4] = { 6, 13, 20, LH_SBMAX_L - 1 };
    static int const last_short[4] = { 2, 6, 11, LH_SBMAX_S - 1 };
    float const tune[4] = { aux->adjust_bass_db, aux->adjust_alto_db, aux->adjust_treble_db, aux->adjust_sfb21_db };
    int     g, b;
    for (g = 0, b = 0; g < 4; g++) {
        float const db = tune[g] + group_db_long[g];
        float const w = powf(10.f, db * 0.1f);
        for (; b <= last_long[g]; b++)
            t->longfact[b] = w;
    }
    for (g = 0, b = 0; g < 4; g++) {
        float const db = tune[g] + group_db_short[g];
        float const w = powf(10.f, db * 0.1f);
        for (; b <= last_short[g]; b++)
            t->shortfact[b] = w;
    }
}

/* ---- psycho-acoustic constants (reference psymodel.c:1605-2157) --------------------------- */

/* critical-band rate of a frequency (Hz -> Bark; reference util.c:268-282) */
static float
bark_of(float hz)
{
    float   khz = hz < 0 ? 0 : hz;
    khz = khz * 0.001;
    return 13.0 * atan(.76 * khz) + 3.5 * atan(khz * khz / (7.5 * 7.5));
}

/* masking level difference between mid/side and left/right at a frequency */
static float
mld_at(double hz)
{
    double  z = bark_of(hz);
    z = ((z < 15.5 ? z : 15.5) / 15.5);
    return pow(10.0, 1.25 * (1 - cos(LH_PI * z)) - 2.5);
}

/* the spreading function at a distance of `dz' Bark from the masker: a slope pair with a dip
 * between 0.5 and 2.5 (scaled) Bark above the masker, cut at -60 dB, normalised */
static float
spread_at(float dz)
{
    float   z = dz, dip = 0.0, slopes;
    if (z >= 0)
        z *= 3;
    else
        z *= 1.5;
    if (z >= 0.5 && z <= 2.5) {
        float const u = z - 0.5;
        dip = 8.0 * (u * u - 2.0 * u);
    }
    z += 0.474;
    slopes = 15.811389 + 7.5 * z - 17.5 * sqrt(1.0 + z * z);
    if (slopes <= -60.0)
        return 0.0;
    z = exp((dip + slopes) * (LH_LOG10 / 10));
    z /= .6609193;
    return z;
}

/* Partition of the FFT lines 0 .. n/2 into bands about a third of a Bark wide: first[p] is the
 * first line of partition p, first[count] one past the last line (it can be n/2 + 1). */
typedef struct Partitioning {
    int     count;
    int     first[LH_CBANDS + 1];
    int     of_line[LH_HBLKSIZE + 1];
    float   hz_per_line;
} Partitioning;

static void
partition_spectrum(Partitioning * P, float rate, int n)
{
    int const half = n / 2;
    float   z[LH_HBLKSIZE + 2];
    int     line, p;
    P->hz_per_line = rate / n;
    for (line = 0; line <= half + 1; line++)
        z[line] = bark_of(P->hz_per_line * line);
    memset(P->of_line, 0, sizeof(P->of_line));
    P->first[0] = 0;
    for (p = 0; p < LH_CBANDS;) {
        int     end = P->first[p];
        while (z[end] - z[P->first[p]] < LH_DELBARK && end <= half)
            end++;
        for (line = P->first[p]; line < end; line++)
            P->of_line[line] = p;
        P->first[++p] = end;
        if (end > half)
            break;
    }
    P->count = p;
}

/* numlines, their reciprocals, the M/S masking level difference per partition, and the mapping
 * of partitions to the scalefactor bands whose edges are edge[0..nbands] on a grid of `lines'
 * MDCT lines: last partition of a band (bo), a middle one (bm), and how far into partition bo the
 * band's upper edge reaches (bo_weight) */
static void
build_partition_tables(LhPsyBand * d, float rate, int n, int lines, int nbands, int const *edge)
{
    Partitioning P;
    int const half = n / 2;
    float const line_ratio = n / (2.0f * lines);        /* FFT lines per MDCT line */
    float const hz_per_mdct_line = rate / (2.0f * lines);
    float   hz_first[LH_CBANDS + 1];
    int     p, b;
    partition_spectrum(&P, rate, n);
    d->npart = P.count;
    d->n_sb = nbands;
    memset(hz_first, 0, sizeof(hz_first));
    for (p = 0; p <= P.count; p++)
        hz_first[p] = P.hz_per_line * ((p == P.count && P.first[p] > half) ? half : P.first[p]);
    for (p = 0; p < P.count; p++) {
        int const nl = P.first[p + 1] - P.first[p];
        d->numlines[p] = nl;
        d->rnumlines[p] = (nl > 0) ? (1.0f / nl) : 0;
        d->mld_cb[p] = mld_at(P.hz_per_line * (P.first[p] + nl / 2));
    }
    for (; p < LH_CBANDS; p++)
        d->mld_cb[p] = 1;
    for (b = 0; b < nbands; b++) {
        int     lo = floor(.5 + line_ratio * (edge[b] - .5));
        int     hi = floor(.5 + line_ratio * (edge[b + 1] - .5));
        int     last;
        float   reach;
        if (lo < 0)
            lo = 0;
        if (hi > half)
            hi = half;
        last = P.of_line[hi];
        d->bo[b] = last;
        d->bm[b] = (P.of_line[lo] + last) / 2;
        reach = (hz_per_mdct_line * edge[b + 1] - hz_first[last]) / (hz_first[last + 1] - hz_first[last]);
        d->bo_weight[b] = reach < 0 ? 0 : (reach > 1 ? 1 : reach);
        d->mld[b] = mld_at(hz_per_mdct_line * edge[b]);
    }
}

/* centre and width of each partition on the Bark scale */
static void
partition_barks(const LhPsyBand * d, float rate, int n, float *centre, float *width)
{
    float const hz_per_line = rate / n;
    int     p, line = 0;
    for (p = 0; p < d->npart; p++) {
        int const nl = d->numlines[p];
        float const z_lo = bark_of(hz_per_line * (line)), z_hi = bark_of(hz_per_line * (line + nl - 1));
        float const e_lo = bark_of(hz_per_line * (line - .5)), e_hi = bark_of(hz_per_line * (line + nl - .5));
        centre[p] = .5 * (z_lo + z_hi);
        width[p] = e_hi - e_lo;
        line += nl;
    }
}

/* spreading matrix, stored as the run of positive entries of each row: s3[s3_row[i] ..] holds the
 * columns s3ind[i][0] .. s3ind[i][1] of row i = spread(centre_i - centre_j) width_j gain_i */
static int
build_spreading(LhPsyBand * d, float const *centre, float const *width, float const *gain)
{
    int const np = d->npart;
    int     i, j, used = 0;
    for (i = 0; i < np; i++) {
        float   row[LH_CBANDS];
        int     lo, hi;
        for (j = 0; j < np; j++) {
            float const v = spread_at(centre[i] - centre[j]) * width[j];
            row[j] = v * gain[i];
        }
        for (lo = 0; lo < np && !(row[lo] > 0.0f); lo++);
        for (hi = np - 1; hi > 0 && !(row[hi] > 0.0f); hi--);
        d->s3ind[i][0] = lo;
        d->s3ind[i][1] = hi;
        d->s3_row[i] = used;
        if (used + (hi - lo + 1) > LH_S3_MAX)
            return -1;
        for (j = lo; j <= hi; j++)
            d->s3[used++] = row[j];
    }
    d->s3_count = used;
    return 0;
}

/* signal-to-mask gain of a partition: `below' dB up to 13 Bark, then a line to `above' dB at 24 */
static float
snr_gain(float z, float below, float above)
{
    float const z0 = 13, z1 = 24;
    double  snr = below;
    if (z >= z0)
        snr = above * (z - z0) / (z1 - z0) + below * (z1 - z) / (z1 - z0);
    return pow(10.0, snr / 10.0);
}

/* threshold in quiet per partition: the quietest FFT line's level, 20 dB down, times the line
 * count; lines are spaced rate / n apart */
static void
build_partition_quiet(const LhConfig * c, const LhPsyBand * d, int n, float *out)
{
    float const rate = c->samplerate;
    int     p, k, line = 0;
    for (p = 0; p < d->npart; p++) {
        double  least = FLT_MAX;
        for (k = 0; k < d->numlines[p]; k++, line++) {
            float const khz = rate * line / (1000.0 * n);
            float   e = quiet_level_db(c, khz * 1000) - 20;
            e = pow(10., 0.1 * e);
            e *= d->numlines[p];
            if (least > e)
                least = e;
        }
        out[p] = least;
    }
}

/* floor of the masking threshold relative to the partition's energy: from a tonality estimate
 * that follows the Bark value (`shape': 0 = long blocks, 1 = short blocks), limited by the
 * preset's minval */
static float
masking_floor(const LhConfig * c, float z, int nl, int shape)
{
    float const lowest = (0.f - c->minval);
    double  x;
    if (shape == 0)
        x = 20.0 * (z / 10 - 1.0);
    else {
        float const knee = 12;
        x = 7.0 * (z / knee - 1.0);
        if (z > knee)
            x *= 1 + log(1 + x) * 3.1;
        if (z < knee)
            x *= 1 + log(1 - x) * 2.3;
    }
    if (x > 6)
        x = 30;
    if (x < lowest)
        x = lowest;
    if (c->samplerate < 44000)
        x = 30;
    x -= 8.;
    return pow(10.0, x / 10.) * nl;
}

/* lowering of the masking threshold towards high partitions (the VBR quality's "sk" slope) */
static void
build_masking_lower(LhPsyBand * d, float slope)
{
    int     p;
    for (p = 0; p < d->npart; p++) {
        float const m = (float) (d->npart - p) / d->npart;
        d->masking_lower[p] = powf(10.f, slope * m * 0.1f);
    }
    for (; p < LH_CBANDS; p++)
        d->masking_lower[p] = 1.f;
}

/* loudness weights of the 512 FFT lines (inverse threshold in quiet), normalised to sum 1 */
static void
build_loudness_weights(const LhConfig * c, LhTables * t)
{
    float const step = (float) c->samplerate / (float) (LH_BLKSIZE);
    float   hz = 0.0, total = 0.0;
    int     k;
    for (k = 0; k < LH_BLKSIZE / 2; ++k) {
        hz += step;
        t->ath_eql_w[k] = 1. / pow(10, quiet_level_db(c, hz) / 10);
        total += t->ath_eql_w[k];
    }
    total = 1.0 / total;
    for (k = LH_BLKSIZE / 2; --k >= 0;)
        t->ath_eql_w[k] *= total;
}

static int
psymodel_tables(LhConfig * c, const LhInitAux * aux, LhTables * t)
{
    float const rate = c->samplerate;
    float   centre[LH_CBANDS], width[LH_CBANDS], gain[LH_CBANDS];
    LhPsyBand *L = &t->psy_l, *S = &t->psy_s;
    int     p;

    /* long blocks: 1024-point FFT over 576 MDCT lines */
    memset(gain, 0, sizeof(gain));
    build_partition_tables(L, rate, LH_BLKSIZE, 576, LH_SBMAX_L, t->sfb_l);
    partition_barks(L, rate, LH_BLKSIZE, centre, width);
    for (p = 0; p < L->npart; p++)
        gain[p] = snr_gain(centre[p], 0, 0);
    if (build_spreading(L, centre, width, gain))
        return -1;
    build_partition_quiet(c, L, LH_BLKSIZE, t->ath_cb_l);
    for (p = 0; p < L->npart; p++)
        L->minval[p] = masking_floor(c, centre[p], L->numlines[p], 0);

    /* short blocks: 256-point FFT over 192 MDCT lines */
    build_partition_tables(S, rate, LH_BLKSIZE_S, 192, LH_SBMAX_S, t->sfb_s);
    partition_barks(S, rate, LH_BLKSIZE_S, centre, width);
    for (p = 0; p < S->npart; p++)
        gain[p] = snr_gain(centre[p], -8.25, -4.5);
    build_partition_quiet(c, S, LH_BLKSIZE_S, t->ath_cb_s);
    for (p = 0; p < S->npart; p++)
        S->minval[p] = masking_floor(c, centre[p], S->numlines[p], 1);
    if (build_spreading(S, centre, width, gain))
        return -1;

    /* limits of the masking-addition table indices, temporal decay of the short-block thresholds */
    t->ma_max_i1 = pow(10, (8 + 1) / 16.0);
    t->ma_max_i2 = pow(10, (23 + 1) / 16.0);
    t->decay = exp(-1.0 * LH_LOG10 / (0.01 * rate / 192.0));
    {
        float   msfix = 3.5;    /* NS_MSFIX */
        if (c->use_safe_joint_stereo)
            msfix = 1.0;
        if (fabs(c->msfix) > 0.0)
            msfix = c->msfix;
        c->msfix = msfix;
    }
    for (p = 0; p < L->npart; p++)
        if (L->s3ind[p][1] > L->npart - 1)
            L->s3ind[p][1] = L->npart - 1;
    /* ATH auto-adjustment and the loudness measure it follows */
    t->ath_decay = pow(10., -12. / 10. * (576. * c->mode_gr / rate));
    t->ath_use_adjust = aux->athaa_type;
    t->aa_sensitivity_p = pow(10.0, aux->athaa_sensitivity / -10.0);
    build_loudness_weights(c, t);
    /* attack detection thresholds: the three sub-windows share one, the fourth has its own */
    {
        float const usual = aux->attackthre < 0 ? (float) 4.4 : aux->attackthre;
        float const late = aux->attackthre_s < 0 ? (float) 25 : aux->attackthre_s;
        t->attack_threshold[0] = t->attack_threshold[1] = t->attack_threshold[2] = usual;
        t->attack_threshold[3] = late;
    }
    {
        /* slope by VBR quality (4 on the CBR path), interpolated for fractional qualities */
        static float const sk[] = { -7.4, -7.4, -7.4, -9.5, -7.4, -6.1, -5.5, -4.7, -4.7, -4.7, -4.7 };
        float const slope = (aux->vbr_q < 4) ? sk[0]
            : sk[aux->vbr_q] + aux->vbr_q_frac * (sk[aux->vbr_q] - sk[aux->vbr_q + 1]);
        build_masking_lower(S, slope);
        build_masking_lower(L, slope);
    }
    /* long-block partitions mapped onto the short scalefactor bands (for the thresholds a
     * short granule inherits from a long analysis) */
    memcpy(&t->psy_l_to_s, L, sizeof(LhPsyBand));
    build_partition_tables(&t->psy_l_to_s, rate, LH_BLKSIZE, 192, LH_SBMAX_S, t->sfb_s);
    return 0;
}

/* FFT windows + the twiddle recurrence of the reference FHT, tabulated
 * (reference fft.c:56-148, 296-310).  tw[stage][i] = {c1, s1, c2, s2} exactly
 * as the in-loop float recurrence produces them, so that butterflies can be
 * evaluated in any order on the device. */
static void
fft_tables(LhTables * t)
{
    static const float costab[8] = {
        9.238795325112867e-01, 3.826834323650898e-01,
        9.951847266721969e-01, 9.801714032956060e-02,
        9.996988186962042e-01, 2.454122852291229e-02,
        9.999811752826011e-01, 6.135884649154475e-03
    };
    int     i, stage, k4;
    for (i = 0; i < LH_BLKSIZE; i++)
        t->fft_window[i] = 0.42 - 0.5 * cos(2 * LH_PI * (i + .5) / LH_BLKSIZE) +
            0.08 * cos(4 * LH_PI * (i + .5) / LH_BLKSIZE);
    for (i = 0; i < LH_BLKSIZE_S / 2; i++)
        t->fft_window_s[i] = 0.5 * (1.0 - cos(2.0 * LH_PI * (i + 0.5) / LH_BLKSIZE_S));
    memset(t->fht_tw, 0, sizeof(t->fht_tw));
    k4 = 4;
    for (stage = 0; stage < 4; stage++) {
        const float *tri = costab + 2 * stage;
        int     kx = k4 >> 1;
        float   c1 = tri[0], s1 = tri[1];
        for (i = 1; i < kx; i++) {
            float   c2, s2;
            c2 = 1 - (2 * s1) * s1;
            s2 = (2 * s1) * c1;
            t->fht_tw[stage][i][0] = c1;
            t->fht_tw[stage][i][1] = s1;
            t->fht_tw[stage][i][2] = c2;
            t->fht_tw[stage][i][3] = s2;
            c2 = c1;
            c1 = c2 * tri[0] - s1 * tri[1];
            s1 = c2 * tri[1] + s1 * tri[0];
        }
        k4 <<= 2;
    }
}

/* polyphase lowpass (reference lame.c:91-190) */
static float
filter_coef(float x)
{
    if (x > 1.0)
        return 0.0;
    if (x <= 0.0)
        return 1.0;
    return cos(LH_PI / 2 * x);
}

static void
ppflt_tables(const LhInitAux * aux, LhTables * t)
{
    int     band, minband;
    float   freq;
    int     lowpass_band = 32;
    float   lowpass1 = aux->lowpass1, lowpass2 = aux->lowpass2;
    float   highpass1 = aux->highpass1, highpass2 = aux->highpass2;

    if (lowpass1 > 0) {
        minband = 999;
        for (band = 0; band <= 31; band++) {
            freq = band / 31.0;
            if (freq >= lowpass2)
                lowpass_band = lowpass_band < band ? lowpass_band : band;
            if (lowpass1 < freq && freq < lowpass2)
                minband = minband < band ? minband : band;
        }
        if (minband == 999)
            lowpass1 = (lowpass_band - .75) / 31.0;
        else
            lowpass1 = (minband - .75) / 31.0;
        lowpass2 = lowpass_band / 31.0;
    }
    /* lame_set_highpassfreq: the bands the filter really has (reference lame.c:140-171) */
    if (highpass2 > 0 && highpass2 < .9 * (.75 / 31.0))
        highpass1 = highpass2 = 0;
    if (highpass2 > 0) {
        int     maxband = -1, highpass_band = -1;
        for (band = 0; band <= 31; band++) {
            freq = band / 31.0;
            if (freq <= highpass1)
                highpass_band = highpass_band > band ? highpass_band : band;
            if (highpass1 < freq && freq < highpass2)
                maxband = maxband > band ? maxband : band;
        }
        highpass1 = highpass_band / 31.0;
        if (maxband == -1)
            highpass2 = (highpass_band + .75) / 31.0;
        else
            highpass2 = (maxband + .75) / 31.0;
    }
    for (band = 0; band < 32; band++) {
        float   fc1, fc2;
        freq = band / 31.0f;
        if (highpass2 > highpass1)
            fc1 = filter_coef((highpass2 - freq) / (highpass2 - highpass1 + 1e-20));
        else
            fc1 = 1.0f;
        if (lowpass2 > lowpass1)
            fc2 = filter_coef((freq - lowpass1) / (lowpass2 - lowpass1 + 1e-20));
        else
            fc2 = 1.0f;
        t->amp_filter[band] = fc1 * fc2;
    }
}

int
lh_tables_build(LhConfig * c, const LhInitAux * aux, LhTables * t)
{
    int     i;
    const int16_t *sl, *ss;

    memset(t, 0, sizeof(*t));
    {
        /* the rate's row of the nine scalefactor band tables (reference lame.c:922) */
        int const j = c->samplerate_index + 3 * c->version + 6 * (c->samplerate < 16000);
        sl = &lh_sfb_l_all[23 * j];
        ss = &lh_sfb_s_all[14 * j];
    }
    for (i = 0; i < LH_SBMAX_L + 1; i++)
        t->sfb_l[i] = sl[i];
    for (i = 0; i < LH_PSFB21 + 1; i++) {
        int const size = (t->sfb_l[22] - t->sfb_l[21]) / LH_PSFB21;
        t->psfb21[i] = t->sfb_l[21] + i * size;
    }
    t->psfb21[LH_PSFB21] = 576;
    for (i = 0; i < LH_SBMAX_S + 1; i++)
        t->sfb_s[i] = ss[i];
    for (i = 0; i < LH_PSFB12 + 1; i++) {
        int const size = (t->sfb_s[13] - t->sfb_s[12]) / LH_PSFB12;
        t->psfb12[i] = t->sfb_s[12] + i * size;
    }
    t->psfb12[LH_PSFB12] = 192;

    ppflt_tables(aux, t);
    build_band_thresholds(c, t);
    build_power_tables(t);
    if (!power_tables_scale_exactly(t))
        return -1;
    if (!build_region_split(t))
        return -1;
    build_huffman_grids(t);
    build_quant_thresholds(t);
    if (!build_vbr_quant_thresholds(t))
        return -1;
    build_band_weights(aux, t);
    if (psymodel_tables(c, aux, t))
        return -1;
    fft_tables(t);
    /* reference util.c:960-972 */
    for (i = 0; i < 513; i++)
        t->log_table[i] = log(1.0f + i / (float) 512) / log(2.0f);
    if (!build_mask_boundaries(t))
        return -1;
    {
        /* band of every line (derived; long: largest sfb with sfb_l[sfb] <= i, short: window-major) */
        int     i, k;
        for (i = 0; i < 576; i++) {
            int     bl = 0, bs = 0;
            for (k = 1; k < LH_SBMAX_L; k++)
                if (t->sfb_l[k] <= i)
                    bl = k;
            for (k = 1; k < 3 * LH_SBMAX_S; k++) {
                int const sfb = k / 3, win = k - 3 * sfb;
                int const wd = t->sfb_s[sfb + 1] - t->sfb_s[sfb];
                if (3 * t->sfb_s[sfb] + win * wd <= i)
                    bs = k;
            }
            t->sfb_line_l[i] = (uint8_t) bl;
            t->sfb_line_s[i] = (uint8_t) bs;
        }
    }
    return 0;
}

/* initial per-stream carried state (reference psymodel.c:1897-1922, 2075-2076;
 * lame.c:962-963, 2285-2302) */
#include "lh_device.h"
void
lh_state_init(LhStreamState * s, const LhConfig * cfg)
{
    int     i, j;
    memset(s, 0, sizeof(*s));
    for (i = 0; i < 4; ++i) {
        for (j = 0; j < LH_CBANDS; ++j) {
            s->nb_l1[i][j] = 1e20;
            s->nb_l2[i][j] = 1e20;
        }
        for (j = 0; j < LH_XMIN_N; j++) {
            s->en[i][j] = 1e20;
            s->thm[i][j] = 1e20;
        }
        s->last_attacks[i] = 0;
        for (j = 0; j < 9; j++)
            s->last_en_subshort[i][j] = 10.;
    }
    s->blocktype_old[0] = s->blocktype_old[1] = LH_NORM_TYPE;
    s->ath_adjust_factor = 0.01;
    s->ath_adjust_limit = 1.0;
    for (i = 0; i < 19; i++)
        s->pefirbuf[i] = 700 * cfg->mode_gr * cfg->channels;
    s->slot_lag = cfg->frac_SpF;
    s->OldValue[0] = s->OldValue[1] = 180;
    s->CurrentStep[0] = s->CurrentStep[1] = 4;
    s->masking_lower = 1;
    s->substep_shaping = cfg->substep_shaping;
}
