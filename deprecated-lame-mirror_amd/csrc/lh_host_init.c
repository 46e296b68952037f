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

static int
find_nearest_bitrate_mpeg1(int b)
{
    int     i, best = lh_bitrate_mpeg1[1];
    for (i = 1; i <= 14; i++) {
        int     d0 = lh_bitrate_mpeg1[i] - b, d1 = best - b;
        if (d0 < 0)
            d0 = -d0;
        if (d1 < 0)
            d1 = -d1;
        if (d0 < d1)
            best = lh_bitrate_mpeg1[i];
    }
    return best;
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
    p->brate = 128;
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

/* the stream's rate: MPEG-1 only on this path (MPEG-2 / 2.5 frames have one granule) */
static int
set_output_rate(LhConfig * c, int rate)
{
    switch (rate) {
    case 44100:
        c->samplerate_index = 0;
        break;
    case 48000:
        c->samplerate_index = 1;
        break;
    case 32000:
        c->samplerate_index = 2;
        break;
    default:
        return -1;
    }
    c->version = 1;
    c->samplerate = rate;
    c->mode_gr = 2;
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
    if (p->vbr_min_kbps) {
        int const k = find_nearest_bitrate_mpeg1(p->vbr_min_kbps);
        for (r = 1; r <= 14; r++)
            if (lh_bitrate_mpeg1[r] == k)
                c->vbr_min_bitrate_index = r;
    }
    if (p->vbr_max_kbps) {
        int const k = find_nearest_bitrate_mpeg1(p->vbr_max_kbps);
        for (r = 1; r <= 14; r++)
            if (lh_bitrate_mpeg1[r] == k)
                c->vbr_max_bitrate_index = r;
    }
    c->enforce_min_bitrate = p->vbr_hard_min;
    return 0;
}

/* vbr_mt / vbr_mtrh settings (reference lame.c:661-692, 730-744, 770-776, 972-1004, 1064-1094;
 * presets.c:146-213 apply_vbr_preset with every option still at its default) */
static int
config_resolve_vbr(const LhUserParams * p, LhConfig * c, LhInitAux * aux)
{
    static const int lp_by_q[11] = { 24000, 19500, 18500, 18000, 17500, 17000, 16500, 15600, 15200, 7230, 3950 };
    int     vbr_q = p->vbr_q, samplerate_out = p->samplerate_out, lowpassfreq = p->lowpassfreq, i;
    float   vbr_q_frac = p->vbr_q_frac;
    LhVbrPreset P, Q;
    float   x;
    int     nspsytune = 1;

    if (vbr_q < 0)
        vbr_q = 0;
    if (vbr_q > 9)
        vbr_q = 9;
    if (samplerate_out == 0) {
        /* VBR scale -> internal quality + output rate (reference lame.c:661-692); rows 2.. of its table */
        static const struct { int sr_a; float qa, qb, ta, tb; } m[7] = {
            {32000, 6.5, 8.0, 5.2, 6.5}, {24000, 8.0, 8.5, 5.2, 6.0}, {22050, 8.5, 9.01, 5.2, 6.5},
            {16000, 9.01, 9.4, 4.9, 6.5}, {12000, 9.4, 9.6, 4.5, 6.0}, {11025, 9.6, 9.9, 5.1, 6.5},
            {8000, 9.9, 10., 4.9, 6.5}
        };
        float const qval = vbr_q + vbr_q_frac;
        for (i = 0; i < 7; ++i) {
            if (p->samplerate == m[i].sr_a) {
                if (qval < m[i].qa) {
                    double  d = qval / m[i].qa;
                    d = d * m[i].ta;
                    vbr_q = (int) d;
                    vbr_q_frac = d - vbr_q;
                }
            }
            if (p->samplerate >= m[i].sr_a) {
                if (m[i].qa <= qval && qval < m[i].qb) {
                    float const q_ = m[i].qb - m[i].qa;
                    float const t_ = m[i].tb - m[i].ta;
                    double  d = m[i].ta + t_ * (qval - m[i].qa) / q_;
                    vbr_q = (int) d;
                    vbr_q_frac = d - vbr_q;
                    samplerate_out = m[i].sr_a;
                    if (lowpassfreq == 0)
                        lowpassfreq = -1;
                    break;
                }
            }
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
        return -1;              /* MPEG-2 / 2.5 output rates are outside this path */
    lowpassfreq = (24000 < lowpassfreq) ? 24000 : lowpassfreq;
    lowpassfreq = (samplerate_out / 2 < lowpassfreq) ? samplerate_out / 2 : lowpassfreq;
    c->lowpassfreq = lowpassfreq;
    lowpass_edges(c, aux, p->lowpasswidth);

    c->bitrate_index = 1;
    c->avg_bitrate = 0;         /* gfp->brate stays 0 in VBR mode */
    c->sideinfo_len = (c->channels == 1) ? 4 + 17 : 4 + 32;
    c->buffer_constraint = 7680 * (c->version + 1);     /* strict_ISO = MDB_MAXIMUM */
    c->use_temporal_masking = 0;

    /* preset row, interpolated by the fractional quality (reference presets.c:146-171) */
    P = vbr_mt_map[vbr_q];
    Q = vbr_mt_map[vbr_q + 1];
    x = vbr_q_frac;
#define LERP(f) (P.f = P.f + x * (Q.f - P.f))
    LERP(st_lrm);
    LERP(st_s);
    LERP(masking_adj);
    LERP(masking_adj_short);
    LERP(ath_lower);
    LERP(ath_curve);
    LERP(ath_sensitivity);
    LERP(interch);
    LERP(sfb21mod);
    LERP(msfix);
    LERP(minval);
    LERP(ath_fixpoint);
#undef LERP
    c->quant_comp = 9;
    c->quant_comp_short = 9;
    aux->attackthre = P.st_lrm;
    aux->attackthre_s = P.st_s;
    c->mask_adjust = P.masking_adj;
    c->mask_adjust_short = P.masking_adj_short;
    c->ATHtype = 5;
    c->ATH_offset_db = 0 - P.ath_lower;
    c->ATH_offset_factor = powf(10.f, c->ATH_offset_db * 0.1f);
    c->ATHcurve = P.ath_curve;
    aux->athaa_sensitivity = P.ath_sensitivity;
    c->interChRatio = (P.interch > 0) ? P.interch : 0;
    if (P.safejoint > 0)
        nspsytune |= 2;
    if (P.sfb21mod > 0)
        nspsytune |= P.sfb21mod << 20;
    c->msfix = P.msfix;
    c->minval = P.minval;
    {
        double const xs = fabs(p->scale);
        double const y = (xs > 0.f) ? (10.f * log10(xs)) : 0.f;
        c->ATHfixpoint = P.ath_fixpoint - y;
    }
    c->use_safe_joint_stereo = nspsytune & 2;
    {
        float   db = (nspsytune >> 20) & 63;
        if (db >= 32.f)
            db -= 64.f;
        db *= 0.25f;
        aux->adjust_sfb21_db = db + 0.f;        /* + adjust_treble_db */
    }
    {
        float   db = c->mask_adjust - 0;
        c->masking_lower_long = pow(10.0, db * 0.1);
        db = c->mask_adjust_short - 0;
        c->masking_lower_short = pow(10.0, db * 0.1);
    }
    /* quality levels of the new VBR code (reference lame.c:985-992) */
    c->quality = (p->quality < 0) ? 3 : p->quality;
    if (c->quality < 5)
        c->quality = 0;
    if (c->quality > 7)
        c->quality = 7;
    apply_quality(c, 0, 0);
    c->sfb21_extra = P.expY ? 0 : (samplerate_out > 44000);
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
    c->vbr_avg_bitrate_kbps = 128;      /* VBR_mean_bitrate_kbps default, inside the MPEG-1 range */
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
    int     noise_shaping = 0, ratio_kbps = 128;

    memset(c, 0, sizeof(*c));
    memset(aux, 0, sizeof(*aux));
    if (p->channels != 1 && p->channels != 2)
        return -1;
    if (p->vbr != 0 && p->vbr != 1 && p->vbr != 3 && p->vbr != 4)
        return -1;              /* the old VBR loop (vbr_rh) is outside this path */
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
    aux->vbr_q_frac = 0;
    aux->athaa_sensitivity = 0;
    aux->adjust_sfb21_db = 0;
    c->vbr_q = aux->vbr_q;
    if (c->vbr == 1 || c->vbr == 4)
        return config_resolve_vbr(p, c, aux);

    if (c->vbr == 3) {
        /* ABR: any mean bitrate; apply_abr_preset clamps it to 8..320 and leaves it in brate
         * (reference presets.c:268-272), lame_init_params to the MPEG-1 table range (lame.c:1088-1093) */
        int     mean = p->abr_kbps;
        if (p->samplerate_out == 0 && (mean < 8 || mean > 320))
            return -1;          /* the reference applies no preset at all there (presets.c:411-416) */
        if (p->samplerate_out)
            mean = mean < 32 ? 32 : (mean > 320 ? 320 : mean);  /* reference lame.c:654-657 */
        mean = mean > 320 ? 320 : mean;
        mean = mean < 8 ? 8 : mean;
        c->avg_bitrate = mean;
        c->bitrate_index = 1;
        c->vbr_avg_bitrate_kbps = mean < 32 ? 32 : mean;
        ratio_kbps = c->vbr_avg_bitrate_kbps;
        if (vbr_bitrate_limits(p, c) != 0)
            return -1;
        /* the mean stays inside the limits (reference lame.c:1086-1091; after compression_ratio was formed) */
        if (c->vbr_avg_bitrate_kbps > lh_bitrate_mpeg1[c->vbr_max_bitrate_index])
            c->vbr_avg_bitrate_kbps = lh_bitrate_mpeg1[c->vbr_max_bitrate_index];
        if (c->vbr_avg_bitrate_kbps < lh_bitrate_mpeg1[c->vbr_min_bitrate_index])
            c->vbr_avg_bitrate_kbps = lh_bitrate_mpeg1[c->vbr_min_bitrate_index];
    }
    else {
        /* bitrate (reference lame.c:904-915) */
        c->avg_bitrate = find_nearest_bitrate_mpeg1(p->brate > 0 ? p->brate : 128);
        for (r = 1; r <= 14; r++)
            if (lh_bitrate_mpeg1[r] == c->avg_bitrate)
                c->bitrate_index = r;
        c->vbr_avg_bitrate_kbps = c->avg_bitrate;       /* lame_set_VBR_mean_bitrate_kbps(brate), lame.c:1043 */
        ratio_kbps = c->avg_bitrate;
    }
    if (c->bitrate_index <= 0)
        return -1;

    /* lowpass (reference lame.c:194-260, 700-760, 846-862) */
    {
        double  lowpass = lowpass_map[nearest_full_index(c->avg_bitrate)];
        int     lp;
        if (c->mode == LH_MODE_MONO)
            lowpass *= 1.5;     /* reference lame.c:758-759 */
        lp = (p->lowpassfreq != 0) ? p->lowpassfreq : (int) lowpass;   /* lame_set_lowpassfreq: Hz, -1 = none */
        /* an output rate the caller left open follows from the lowpass (optimum_samplefreq,
         * reference lame.c:273-345, 762-767); when it differs from the input rate the input is
         * resampled in front of the encoder */
        {
            int     out = p->samplerate_out;
            if (out == 0) {
                if (2 * lp > p->samplerate)
                    lp = p->samplerate / 2;
                out = suggested_samplerate(lp, p->samplerate);
            }
            if (set_output_rate(c, out) != 0)
                return -1;      /* MPEG-2 / 2.5 output rates are outside this path */
        }
        c->compression_ratio = c->samplerate * 16 * c->channels / (1.e3 * ratio_kbps);
        if (lp > 20500)
            lp = 20500;
        if (lp > c->samplerate / 2)
            lp = c->samplerate / 2;
        c->lowpassfreq = lp;
        lowpass_edges(c, aux, p->lowpasswidth);
    }

    c->sideinfo_len = (c->channels == 1) ? 4 + 17 : 4 + 32;
    c->buffer_constraint = 7680 * (c->version + 1);     /* MDB_MAXIMUM, reference bitstream.c:91-131 */

    /* preset for the bitrate (reference presets.c:215-317) */
    r = nearest_full_index(c->avg_bitrate);
    if (abr_map[r].safejoint > 0)
        c->use_safe_joint_stereo = 2;
    if (abr_map[r].sfscale > 0)
        noise_shaping = 2;
    c->quant_comp = 9;
    c->quant_comp_short = 9;
    c->msfix = abr_map[r].nsmsfix;
    aux->attackthre = abr_map[r].st_lrm;
    aux->attackthre_s = abr_map[r].st_s;
    scale = p->scale * abr_map[r].scale;
    maskingadjust = abr_map[r].masking_adj;
    if (abr_map[r].masking_adj > 0)
        maskingadjust_short = abr_map[r].masking_adj * .9;
    else
        maskingadjust_short = abr_map[r].masking_adj * 1.1;
    ath_lower_db = abr_map[r].ath_lower;
    c->ATHcurve = abr_map[r].ath_curve;
    c->interChRatio = abr_map[r].interch;
    c->minval = 5. * (abr_map[r].kbps / 320.);

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
    c->pcm_scale = scale * p->scale_left;
    c->pcm_scale_r = scale * p->scale_right;
    c->disable_reservoir = 0;
    c->frac_SpF = (c->vbr == 0) ? (int) (((c->version + 1) * 72000L * c->avg_bitrate) % c->samplerate) : 0;
    return 0;
}

/* ---------------------------------------------------------------------- */
/* ATH formula (reference util.c:197-264) */
static float
ath_formula_gb(float f, float value, float f_min, float f_max)
{
    float   ath;
    if (f < -.3)
        f = 3410;
    f /= 1000;
    f = (f_min > f) ? f_min : f;
    f = (f_max < f) ? f_max : f;
    ath = 3.640 * pow(f, -0.8)
        - 6.800 * exp(-0.6 * pow(f - 3.4, 2.0))
        + 6.000 * exp(-0.15 * pow(f - 8.7, 2.0))
        + (0.6 + 0.04 * value) * 0.001 * pow(f, 4.0);
    return ath;
}

static float
ath_formula(const LhConfig * c, float f)
{
    switch (c->ATHtype) {
    case 0:
        return ath_formula_gb(f, 9, 0.1f, 24.0f);
    case 1:
        return ath_formula_gb(f, -1, 0.1f, 24.0f);
    case 2:
        return ath_formula_gb(f, 0, 0.1f, 24.0f);
    case 3:
        return ath_formula_gb(f, 1, 0.1f, 24.0f) + 6;
    case 4:
        return ath_formula_gb(f, c->ATHcurve, 0.1f, 24.0f);
    case 5:
        return ath_formula_gb(f, c->ATHcurve, 3.41f, 16.1f);
    default:
        return ath_formula_gb(f, 0, 0.1f, 24.0f);
    }
}

/* reference util.c:268-282 */
static float
freq2bark(float freq)
{
    if (freq < 0)
        freq = 0;
    freq = freq * 0.001;
    return 13.0 * atan(.76 * freq) + 3.5 * atan(freq * freq / (7.5 * 7.5));
}

/* reference quantize_pvt.c:210-228 */
static float
ath_mdct(const LhConfig * c, float f)
{
    float   ath = ath_formula(c, f);
    if (c->ATHfixpoint > 0)
        ath -= c->ATHfixpoint;
    else
        ath -= LH_NSATHSCALE;
    ath += c->ATH_offset_db;
    ath = powf(10.0f, ath * 0.1f);
    return ath;
}

/* reference quantize_pvt.c:230-321 */
static void
compute_ath(const LhConfig * c, LhTables * t)
{
    int     sfb, i, start, end;
    float   ath_f;
    float const samp_freq = c->samplerate;

    for (sfb = 0; sfb < LH_SBMAX_L; sfb++) {
        start = t->sfb_l[sfb];
        end = t->sfb_l[sfb + 1];
        t->ath_l[sfb] = FLT_MAX;
        for (i = start; i < end; i++) {
            float const freq = i * samp_freq / (2 * 576);
            ath_f = ath_mdct(c, freq);
            t->ath_l[sfb] = (t->ath_l[sfb] < ath_f) ? t->ath_l[sfb] : ath_f;
        }
    }
    for (sfb = 0; sfb < LH_PSFB21; sfb++) {
        start = t->psfb21[sfb];
        end = t->psfb21[sfb + 1];
        t->ath_psfb21[sfb] = FLT_MAX;
        for (i = start; i < end; i++) {
            float const freq = i * samp_freq / (2 * 576);
            ath_f = ath_mdct(c, freq);
            t->ath_psfb21[sfb] = (t->ath_psfb21[sfb] < ath_f) ? t->ath_psfb21[sfb] : ath_f;
        }
    }
    for (sfb = 0; sfb < LH_SBMAX_S; sfb++) {
        start = t->sfb_s[sfb];
        end = t->sfb_s[sfb + 1];
        t->ath_s[sfb] = FLT_MAX;
        for (i = start; i < end; i++) {
            float const freq = i * samp_freq / (2 * 192);
            ath_f = ath_mdct(c, freq);
            t->ath_s[sfb] = (t->ath_s[sfb] < ath_f) ? t->ath_s[sfb] : ath_f;
        }
        t->ath_s[sfb] *= (t->sfb_s[sfb + 1] - t->sfb_s[sfb]);
    }
    for (sfb = 0; sfb < LH_PSFB12; sfb++) {
        start = t->psfb12[sfb];
        end = t->psfb12[sfb + 1];
        t->ath_psfb12[sfb] = FLT_MAX;
        for (i = start; i < end; i++) {
            float const freq = i * samp_freq / (2 * 192);
            ath_f = ath_mdct(c, freq);
            t->ath_psfb12[sfb] = (t->ath_psfb12[sfb] < ath_f) ? t->ath_psfb12[sfb] : ath_f;
        }
        t->ath_psfb12[sfb] *= (t->sfb_s[13] - t->sfb_s[12]);
    }
    t->ath_floor = 10. * log10(ath_mdct(c, -1.));
}

/* region split lookup (reference takehiro.c:38-88 subdv_table, 1334-1375 huffman_init) */
static const signed char subdv[23][2] = {
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 1}, {1, 1}, {1, 1}, {1, 2}, {2, 2}, {2, 3},
    {2, 3}, {3, 4}, {3, 4}, {3, 4}, {4, 5}, {4, 5}, {4, 6}, {5, 6}, {5, 6}, {5, 7}, {6, 7},
    {6, 7}
};

static void
build_bv_scf(LhTables * t)
{
    int     i;
    for (i = 2; i <= 576; i += 2) {
        int     scfb_anz = 0, bv_index;
        while (t->sfb_l[++scfb_anz] < i);
        bv_index = subdv[scfb_anz][0];
        while (t->sfb_l[bv_index + 1] > i)
            bv_index--;
        if (bv_index < 0)
            bv_index = subdv[scfb_anz][0];
        t->bv_scf[i - 2] = bv_index;
        bv_index = subdv[scfb_anz][1];
        while (t->sfb_l[bv_index + t->bv_scf[i - 2] + 2] > i)
            bv_index--;
        if (bv_index < 0)
            bv_index = subdv[scfb_anz][1];
        t->bv_scf[i - 1] = bv_index;
    }
}

/* reference quantize_pvt.c:336-417 */
static void
iteration_tables(const LhConfig * c, const LhInitAux * aux, LhTables * t)
{
    static float const payload_long[4] = { -0.500f, -0.250f, -0.025f, +0.500f };
    static float const payload_short[4] = { -2.000f, -1.000f, -0.050f, +0.500f };
    /* adjust_{bass,alto,treble}_db are 0 on this path (exp_nspsytune bits 2..19 are clear); the VBR
     * presets set the sfb21 bits */
    float const adj_bass = 0.f, adj_alto = 0.f, adj_treble = 0.f, adj_sfb21 = aux->adjust_sfb21_db;
    float   adjust, db;
    int     i;

    compute_ath(c, t);
    t->pow43[0] = 0.0;
    for (i = 1; i < LH_PRECALC; i++)
        t->pow43[i] = pow((float) i, 4.0 / 3.0);
    t->adj43asm[0] = 0.0;
    for (i = 1; i < LH_PRECALC; i++)
        t->adj43asm[i] = i - 0.5 - pow(0.5 * (t->pow43[i - 1] + t->pow43[i]), 0.75);
    for (i = 0; i < LH_QMAX; i++)
        t->ipow20[i] = pow(2.0, (double) (i - 210) * -0.1875);
    for (i = 0; i <= LH_QMAX + LH_QMAX2; i++)
        t->pow20[i] = pow(2.0, (double) (i - 210 - LH_QMAX2) * 0.25);
    build_bv_scf(t);

    db = adj_bass + payload_long[0];
    adjust = powf(10.f, db * 0.1f);
    for (i = 0; i <= 6; ++i)
        t->longfact[i] = adjust;
    db = adj_alto + payload_long[1];
    adjust = powf(10.f, db * 0.1f);
    for (; i <= 13; ++i)
        t->longfact[i] = adjust;
    db = adj_treble + payload_long[2];
    adjust = powf(10.f, db * 0.1f);
    for (; i <= 20; ++i)
        t->longfact[i] = adjust;
    db = adj_sfb21 + payload_long[3];
    adjust = powf(10.f, db * 0.1f);
    for (; i < LH_SBMAX_L; ++i)
        t->longfact[i] = adjust;

    db = adj_bass + payload_short[0];
    adjust = powf(10.f, db * 0.1f);
    for (i = 0; i <= 2; ++i)
        t->shortfact[i] = adjust;
    db = adj_alto + payload_short[1];
    adjust = powf(10.f, db * 0.1f);
    for (; i <= 6; ++i)
        t->shortfact[i] = adjust;
    db = adj_treble + payload_short[2];
    adjust = powf(10.f, db * 0.1f);
    for (; i <= 11; ++i)
        t->shortfact[i] = adjust;
    db = adj_sfb21 + payload_short[3];
    adjust = powf(10.f, db * 0.1f);
    for (; i < LH_SBMAX_S; ++i)
        t->shortfact[i] = adjust;
}

/* ---------------------------------------------------------------------- */
/* psycho-acoustic constants (reference psymodel.c:1605-2157)               */

static float
s3_func(float bark)
{
    float   tempx, x, tempy, temp;
    tempx = bark;
    if (tempx >= 0)
        tempx *= 3;
    else
        tempx *= 1.5;

    if (tempx >= 0.5 && tempx <= 2.5) {
        temp = tempx - 0.5;
        x = 8.0 * (temp * temp - 2.0 * temp);
    }
    else
        x = 0.0;
    tempx += 0.474;
    tempy = 15.811389 + 7.5 * tempx - 17.5 * sqrt(1.0 + tempx * tempx);
    if (tempy <= -60.0)
        return 0.0;
    tempx = exp((x + tempy) * (LH_LOG10 / 10));
    tempx /= .6609193;
    return tempx;
}

static float
stereo_demask(double f)
{
    double  arg = freq2bark(f);
    arg = ((arg < 15.5 ? arg : 15.5) / 15.5);
    return pow(10.0, 1.25 * (1 - cos(LH_PI * arg)) - 2.5);
}

static void
init_numline(LhPsyBand * gd, float sfreq, int fft_size, int mdct_size, int sbmax,
             int const *scalepos)
{
    float   b_frq[LH_CBANDS + 1];
    float const mdct_freq_frac = sfreq / (2.0f * mdct_size);
    float const deltafreq = fft_size / (2.0f * mdct_size);
    int     partition[LH_HBLKSIZE];
    int     i, j, ni, sfb;

    memset(partition, 0, sizeof(partition));
    memset(b_frq, 0, sizeof(b_frq));
    sfreq /= fft_size;
    j = 0;
    ni = 0;
    for (i = 0; i < LH_CBANDS; i++) {
        float   bark1;
        int     j2, nl;
        bark1 = freq2bark(sfreq * j);
        b_frq[i] = sfreq * j;
        for (j2 = j; freq2bark(sfreq * j2) - bark1 < LH_DELBARK && j2 <= fft_size / 2; j2++);
        nl = j2 - j;
        gd->numlines[i] = nl;
        gd->rnumlines[i] = (nl > 0) ? (1.0f / nl) : 0;
        ni = i + 1;
        while (j < j2)
            partition[j++] = i;
        if (j > fft_size / 2) {
            j = fft_size / 2;
            ++i;
            break;
        }
    }
    b_frq[i] = sfreq * j;
    gd->n_sb = sbmax;
    gd->npart = ni;
    j = 0;
    for (i = 0; i < gd->npart; i++) {
        int const nl = gd->numlines[i];
        float const freq = sfreq * (j + nl / 2);
        gd->mld_cb[i] = stereo_demask(freq);
        j += nl;
    }
    for (; i < LH_CBANDS; ++i)
        gd->mld_cb[i] = 1;
    for (sfb = 0; sfb < sbmax; sfb++) {
        int     i1, i2, bo;
        int     start = scalepos[sfb];
        int     end = scalepos[sfb + 1];
        i1 = floor(.5 + deltafreq * (start - .5));
        if (i1 < 0)
            i1 = 0;
        i2 = floor(.5 + deltafreq * (end - .5));
        if (i2 > fft_size / 2)
            i2 = fft_size / 2;
        bo = partition[i2];
        gd->bm[sfb] = (partition[i1] + partition[i2]) / 2;
        gd->bo[sfb] = bo;
        {
            float const f_tmp = mdct_freq_frac * end;
            float   bo_w = (f_tmp - b_frq[bo]) / (b_frq[bo + 1] - b_frq[bo]);
            if (bo_w < 0)
                bo_w = 0;
            else if (bo_w > 1)
                bo_w = 1;
            gd->bo_weight[sfb] = bo_w;
        }
        gd->mld[sfb] = stereo_demask(mdct_freq_frac * start);
    }
}

static void
compute_bark_values(LhPsyBand const *gd, float sfreq, int fft_size, float *bval, float *bval_width)
{
    int     k, j = 0, ni = gd->npart;
    sfreq /= fft_size;
    for (k = 0; k < ni; k++) {
        int const w = gd->numlines[k];
        float   bark1, bark2;
        bark1 = freq2bark(sfreq * (j));
        bark2 = freq2bark(sfreq * (j + w - 1));
        bval[k] = .5 * (bark1 + bark2);
        bark1 = freq2bark(sfreq * (j - .5));
        bark2 = freq2bark(sfreq * (j + w - .5));
        bval_width[k] = bark2 - bark1;
        j += w;
    }
}

static int
init_s3_values(LhPsyBand * gd, float const *bval, float const *bval_width, float const *norm)
{
    static float s3[LH_CBANDS][LH_CBANDS];
    int     i, j, k, npart = gd->npart;
    int     nonzero = 0;

    memset(&s3[0][0], 0, sizeof(s3));
    for (i = 0; i < npart; i++) {
        for (j = 0; j < npart; j++) {
            float   v = s3_func(bval[i] - bval[j]) * bval_width[j];
            s3[i][j] = v * norm[i];
        }
    }
    for (i = 0; i < npart; i++) {
        for (j = 0; j < npart; j++)
            if (s3[i][j] > 0.0f)
                break;
        gd->s3ind[i][0] = j;
        for (j = npart - 1; j > 0; j--)
            if (s3[i][j] > 0.0f)
                break;
        gd->s3ind[i][1] = j;
        nonzero += (gd->s3ind[i][1] - gd->s3ind[i][0] + 1);
    }
    if (nonzero > LH_S3_MAX)
        return -1;
    k = 0;
    for (i = 0; i < npart; i++) {
        gd->s3_row[i] = k;
        for (j = gd->s3ind[i][0]; j <= gd->s3ind[i][1]; j++)
            gd->s3[k++] = s3[i][j];
    }
    gd->s3_count = k;
    return 0;
}

static int
psymodel_tables(LhConfig * c, const LhInitAux * aux, LhTables * t)
{
    int     i, j, b, k;
    float   bvl_a = 13, bvl_b = 24;
    float   snr_l_a = 0, snr_l_b = 0;
    float   snr_s_a = -8.25, snr_s_b = -4.5;
    float   bval[LH_CBANDS], bval_width[LH_CBANDS], norm[LH_CBANDS];
    float const sfreq = c->samplerate;
    float   xav = 10, xbv = 12;
    float const minval_low = (0.f - c->minval);
    LhPsyBand *gl = &t->psy_l, *gs = &t->psy_s;

    memset(norm, 0, sizeof(norm));
    init_numline(gl, sfreq, LH_BLKSIZE, 576, LH_SBMAX_L, t->sfb_l);
    compute_bark_values(gl, sfreq, LH_BLKSIZE, bval, bval_width);
    for (i = 0; i < gl->npart; i++) {
        double  snr = snr_l_a;
        if (bval[i] >= bvl_a) {
            snr = snr_l_b * (bval[i] - bvl_a) / (bvl_b - bvl_a)
                + snr_l_a * (bvl_b - bval[i]) / (bvl_b - bvl_a);
        }
        norm[i] = pow(10.0, snr / 10.0);
    }
    if (init_s3_values(gl, bval, bval_width, norm))
        return -1;
    j = 0;
    for (i = 0; i < gl->npart; i++) {
        double  x;
        x = FLT_MAX;
        for (k = 0; k < gl->numlines[i]; k++, j++) {
            float const freq = sfreq * j / (1000.0 * LH_BLKSIZE);
            float   level;
            level = ath_formula(c, freq * 1000) - 20;
            level = pow(10., 0.1 * level);
            level *= gl->numlines[i];
            if (x > level)
                x = level;
        }
        t->ath_cb_l[i] = x;
        x = 20.0 * (bval[i] / xav - 1.0);
        if (x > 6)
            x = 30;
        if (x < minval_low)
            x = minval_low;
        if (c->samplerate < 44000)
            x = 30;
        x -= 8.;
        gl->minval[i] = pow(10.0, x / 10.) * gl->numlines[i];
    }

    init_numline(gs, sfreq, LH_BLKSIZE_S, 192, LH_SBMAX_S, t->sfb_s);
    compute_bark_values(gs, sfreq, LH_BLKSIZE_S, bval, bval_width);
    j = 0;
    for (i = 0; i < gs->npart; i++) {
        double  x;
        double  snr = snr_s_a;
        if (bval[i] >= bvl_a) {
            snr = snr_s_b * (bval[i] - bvl_a) / (bvl_b - bvl_a)
                + snr_s_a * (bvl_b - bval[i]) / (bvl_b - bvl_a);
        }
        norm[i] = pow(10.0, snr / 10.0);
        x = FLT_MAX;
        for (k = 0; k < gs->numlines[i]; k++, j++) {
            float const freq = sfreq * j / (1000.0 * LH_BLKSIZE_S);
            float   level;
            level = ath_formula(c, freq * 1000) - 20;
            level = pow(10., 0.1 * level);
            level *= gs->numlines[i];
            if (x > level)
                x = level;
        }
        t->ath_cb_s[i] = x;
        x = 7.0 * (bval[i] / xbv - 1.0);
        if (bval[i] > xbv)
            x *= 1 + log(1 + x) * 3.1;
        if (bval[i] < xbv)
            x *= 1 + log(1 - x) * 2.3;
        if (x > 6)
            x = 30;
        if (x < minval_low)
            x = minval_low;
        if (c->samplerate < 44000)
            x = 30;
        x -= 8;
        gs->minval[i] = pow(10.0, x / 10) * gs->numlines[i];
    }
    if (init_s3_values(gs, bval, bval_width, norm))
        return -1;

    t->ma_max_i1 = pow(10, (8 + 1) / 16.0);
    t->ma_max_i2 = pow(10, (23 + 1) / 16.0);

    t->decay = exp(-1.0 * LH_LOG10 / (0.01 * sfreq / 192.0));
    {
        float   msfix = 3.5;    /* NS_MSFIX */
        if (c->use_safe_joint_stereo)
            msfix = 1.0;
        if (fabs(c->msfix) > 0.0)
            msfix = c->msfix;
        c->msfix = msfix;
        for (b = 0; b < gl->npart; b++)
            if (gl->s3ind[b][1] > gl->npart - 1)
                gl->s3ind[b][1] = gl->npart - 1;
    }
    t->ath_decay = pow(10., -12. / 10. * (576. * c->mode_gr / sfreq));
    t->ath_use_adjust = 3;
    t->aa_sensitivity_p = pow(10.0, aux->athaa_sensitivity / -10.0);
    {
        float   freq;
        float const freq_inc = (float) c->samplerate / (float) (LH_BLKSIZE);
        float   eql_balance = 0.0;
        freq = 0.0;
        for (i = 0; i < LH_BLKSIZE / 2; ++i) {
            freq += freq_inc;
            t->ath_eql_w[i] = 1. / pow(10, ath_formula(c, freq) / 10);
            eql_balance += t->ath_eql_w[i];
        }
        eql_balance = 1.0 / eql_balance;
        for (i = LH_BLKSIZE / 2; --i >= 0;)
            t->ath_eql_w[i] *= eql_balance;
    }
    {
        float   x = aux->attackthre;
        float   y = aux->attackthre_s;
        if (x < 0)
            x = 4.4;
        if (y < 0)
            y = 25;
        t->attack_threshold[0] = t->attack_threshold[1] = t->attack_threshold[2] = x;
        t->attack_threshold[3] = y;
    }
    {
        /* VBR_q stays at its default 4 on the CBR path */
        float   sk_s, sk_l;
        static float const sk[] =
            { -7.4, -7.4, -7.4, -9.5, -7.4, -6.1, -5.5, -4.7, -4.7, -4.7, -4.7 };
        if (aux->vbr_q < 4)
            sk_l = sk_s = sk[0];
        else
            sk_l = sk_s = sk[aux->vbr_q] + aux->vbr_q_frac * (sk[aux->vbr_q] - sk[aux->vbr_q + 1]);
        b = 0;
        for (; b < gs->npart; b++) {
            float   m = (float) (gs->npart - b) / gs->npart;
            gs->masking_lower[b] = powf(10.f, sk_s * m * 0.1f);
        }
        for (; b < LH_CBANDS; ++b)
            gs->masking_lower[b] = 1.f;
        b = 0;
        for (; b < gl->npart; b++) {
            float   m = (float) (gl->npart - b) / gl->npart;
            gl->masking_lower[b] = powf(10.f, sk_l * m * 0.1f);
        }
        for (; b < LH_CBANDS; ++b)
            gl->masking_lower[b] = 1.f;
    }
    memcpy(&t->psy_l_to_s, gl, sizeof(LhPsyBand));
    init_numline(&t->psy_l_to_s, sfreq, LH_BLKSIZE, 192, LH_SBMAX_S, t->sfb_s);
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
    for (band = 0; band < 32; band++) {
        float   fc1, fc2;
        freq = band / 31.0f;
        fc1 = 1.0f;             /* no highpass on this path */
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
    switch (c->samplerate_index) {
    case 0:
        sl = lh_sfb_l_0;
        ss = lh_sfb_s_0;
        break;
    case 1:
        sl = lh_sfb_l_1;
        ss = lh_sfb_s_1;
        break;
    default:
        sl = lh_sfb_l_2;
        ss = lh_sfb_s_2;
        break;
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
    iteration_tables(c, aux, t);
    if (psymodel_tables(c, aux, t))
        return -1;
    fft_tables(t);
    /* reference util.c:960-972 */
    for (i = 0; i < 513; i++)
        t->log_table[i] = log(1.0f + i / (float) 512) / log(2.0f);
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
