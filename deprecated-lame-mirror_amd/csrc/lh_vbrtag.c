/*
 * lh_vbrtag.c -- host side: the Xing/"Info" + LAME tag frame of a CBR stream
 * (SURVEY.md 8(f) row 3).  Behaviour of the reference's libmp3lame/VbrTag.c:
 *   - lame_init_params reserves a frame-sized placeholder at the start of the stream
 *     (InitVbrTag :492-581; all zero after the 4 header bytes),
 *   - every encoded frame adds its bit rate to a bounded seek-point bag (AddVbrFrame :196,
 *     addVbr :124-148),
 *   - the audio bytes handed to the caller feed a CRC-16 and a byte count
 *     (bitstream.c:1078-1090),
 *   - lame_get_lametag_frame (:900-1018) builds the final frame: "Info", frame and byte
 *     counts, a 100-entry seek table (Xing_seek_table :151-172) and the LAME extension
 *     (PutLameVBR :615-861) with two CRCs.
 * Everything here is byte bookkeeping on the host; nothing touches the GPU.
 */
#include <math.h>
#include <string.h>
#include "lh_host.h"

static uint16_t lh_crc16_tab[256];
static int lh_crc16_ready = 0;

static void
crc16_init(void)
{
    /* CRC-16 with the reflected polynomial 0xA001, the table the reference spells out
     * (VbrTag.c:73-106) */
    int     i, k;
    for (i = 0; i < 256; i++) {
        uint16_t c = (uint16_t) i;
        for (k = 0; k < 8; k++)
            c = (uint16_t) ((c & 1) ? (c >> 1) ^ 0xA001 : (c >> 1));
        lh_crc16_tab[i] = c;
    }
    lh_crc16_ready = 1;
}

static uint16_t
crc16_update(uint16_t value, uint16_t crc)
{
    uint16_t const tmp = (uint16_t) (crc ^ value);
    return (uint16_t) ((crc >> 8) ^ lh_crc16_tab[tmp & 0xff]);
}

void
lh_tag_crc(LhVbrTag * v, const unsigned char *buf, long n)
{
    long    i;
    if (!lh_crc16_ready)
        crc16_init();
    for (i = 0; i < n; i++)
        v->music_crc = crc16_update(buf[i], v->music_crc);
    v->bytes_written += (unsigned long) n;
}

/* kbps of a bitrate index in the stream's MPEG version (what AddVbrFrame accumulates, reference VbrTag.c:195-201) */
int
lh_tag_kbps(int version, int bitrate_index)
{
    return lh_bitrate_row(version)[bitrate_index & 15];
}

/* the bit rate of the tag frame itself (reference VbrTag.c:515-529, 287-300): the stream's own for CBR, else 128 / 64 /
 * 32 kb/s for MPEG-1 / 2 / 2.5, and its index in the version's row (2.5: the MPEG-2 row holds 32 at the same index) */
static int
tag_frame_kbps(const LhConfig * c)
{
    if (c->vbr == 0)
        return c->avg_bitrate;
    return c->version == 1 ? 128 : (c->samplerate < 16000 ? 32 : 64);
}

static int
tag_frame_bitrate_index(const LhConfig * c)
{
    int const kbps = tag_frame_kbps(c);
    int     i;
    if (c->vbr == 0)
        return c->bitrate_index;
    for (i = 1; i <= 14; i++)
        if (lh_bitrate_row(c->version)[i] == kbps)
            return i;
    return 0;
}

/* reference VbrTag.c:492-559: 0 = the tag does not fit, stays off */
int
lh_tag_init(LhVbrTag * v, const LhConfig * c)
{
    /* CBR: the stream's own frame size; VBR: a 128 / 64 / 32 kbps frame (XING_BITRATE1 / 2 / 25, reference VbrTag.c:515-529) */
    int const kbps_header = tag_frame_kbps(c);
    int const total = ((c->version + 1) * 72000 * kbps_header) / c->samplerate;
    int const header_size = c->sideinfo_len + LH_LAMEHEADERSIZE;
    memset(v, 0, sizeof(*v));
    if (!lh_crc16_ready)
        crc16_init();
    if (total < header_size || total > 2880)
        return 0;
    v->total_frame_size = total;
    v->want = 1;
    v->size = LH_TAG_BAG;
    v->enabled = 1;
    v->samplerate_in = c->samplerate;
    return total;
}

/* Seek-point bookkeeping (reference VbrTag.c:124-148): the running sum of the frames' bit rates is
 * sampled every `want' frames into a bag of fixed size; a full bag keeps every second sample and
 * the sampling distance doubles, so the bag always spans the whole stream evenly. */
void
lh_tag_add_frame(LhVbrTag * v, int kbps)
{
    v->num_frames++;
    v->sum += kbps;
    if (++v->seen < v->want)
        return;
    if (v->pos < v->size) {
        v->bag[v->pos++] = v->sum;
        v->seen = 0;
    }
    if (v->pos == v->size) {
        int     keep;
        for (keep = 0; 2 * keep + 1 < v->size; keep++)
            v->bag[keep] = v->bag[2 * keep + 1];
        v->pos /= 2;
        v->want += v->want;
    }
}

/* the 4 header bytes, reference VbrTag.c:257-323 (MPEG-1, CBR: the stream's own bit rate) */
static void
tag_frame_header(const LhConfig * c, int mode_ext, unsigned char *b)
{
    b[0] = 0xff;
    b[1] = (unsigned char) (0xe0 | ((c->samplerate < 16000 ? 0 : 1) << 4) | (c->version << 3) | (1 << 1)
                            | (c->error_protection ? 0 : 1));
    b[1] = (unsigned char) ((b[1] & 0xf1) | (c->version == 1 ? 0x0a : 0x02));
    b[2] = (unsigned char) (((c->samplerate_index << 2) | (c->extension & 1)) & 0x0d);
    /* bitrate field: the CBR rate, or the tag frame's own for VBR streams (reference VbrTag.c:286-303) */
    b[2] = (unsigned char) (b[2] | (16 * tag_frame_bitrate_index(c)));
    b[3] = (unsigned char) ((c->mode << 6) | ((mode_ext & 3) << 4) | ((c->copyright & 1) << 3)
                            | ((c->original & 1) << 2) | (c->emphasis & 3));
}

/* the placeholder that opens the stream; returns its size */
int
lh_tag_placeholder(const LhVbrTag * v, const LhConfig * c, unsigned char *buf)
{
    memset(buf, 0, (size_t) v->total_frame_size);
    tag_frame_header(c, 0, buf);
    return v->total_frame_size;
}

static void
put_i4(unsigned char *b, uint32_t x)
{
    b[0] = (unsigned char) (x >> 24);
    b[1] = (unsigned char) (x >> 16);
    b[2] = (unsigned char) (x >> 8);
    b[3] = (unsigned char) x;
}

static void
put_i2(unsigned char *b, int x)
{
    b[0] = (unsigned char) ((x >> 8) & 0xff);
    b[1] = (unsigned char) (x & 0xff);
}

/* final tag frame; returns the frame size, 0 when there is no tag (reference VbrTag.c:900-1018).
 * user_quality / vbr_q are the caller-level settings the "quality" byte is made of. */
int
lh_tag_frame(const LhVbrTag * v, const LhConfig * c, int vbr_q, int enc_padding, int last_mode_ext,
             unsigned char *buf, long size)
{
    unsigned char toc[100];
    int     i, n;
    uint16_t crc = 0;
    if (!v->enabled || v->pos <= 0)
        return 0;
    if (size < v->total_frame_size)
        return v->total_frame_size;
    if (!buf)
        return 0;
    memset(buf, 0, (size_t) v->total_frame_size);
    tag_frame_header(c, last_mode_ext, buf);
    memset(toc, 0, sizeof(toc));
    for (i = 1; i < 100; ++i) {
        /* entry i = where i percent of the playing time lies, in 1/256 of the byte count: the
         * bag sample nearest below, relative to the final sum (float index and ratio, double
         * scaling, as the reference's Xing_seek_table evaluates them) */
        float const percent = i / (float) 100;
        int     slot = (int) (floor(percent * v->pos)), where;
        float   upto, all;
        slot = slot > v->pos - 1 ? v->pos - 1 : slot;
        upto = (float) v->bag[slot];
        all = (float) v->sum;
        where = (int) (256. * upto / all);
        toc[i] = (unsigned char) (where > 255 ? 255 : where);
    }
    n = c->sideinfo_len;
    if (c->error_protection)
        n -= 2;
    memcpy(buf + n, c->vbr == 0 ? "Info" : "Xing", 4);   /* reference VbrTag.c:964-977 */
    n += 4;
    put_i4(buf + n, 0x0001 | 0x0002 | 0x0004 | 0x0008);    /* frames, bytes, TOC, quality */
    n += 4;
    put_i4(buf + n, (uint32_t) v->num_frames);
    n += 4;
    put_i4(buf + n, (uint32_t) (v->bytes_written + (unsigned long) v->total_frame_size));
    n += 4;
    memcpy(buf + n, toc, sizeof(toc));
    n += (int) sizeof(toc);
    if (c->error_protection) {
        /* the tag frame carries a header CRC like any other frame (reference VbrTag.c:996-999); it
         * covers the first bytes of the tag, which start two bytes early in this case */
        unsigned const hc = lh_header_crc(buf, c->sideinfo_len);
        buf[4] = (unsigned char) (hc >> 8);
        buf[5] = (unsigned char) (hc & 255u);
    }
    for (i = 0; i < n; i++)
        crc = crc16_update(buf[i], crc);
    {
        /* LAME extension, reference PutLameVBR */
        unsigned char *p = buf + n;
        int     k = 0;
        int     quality = 100 - 10 * vbr_q - c->quality;
        double const lp = (c->lowpassfreq / 100.0) + .5;
        unsigned char const lowpass = (unsigned char) (lp > 255 ? 255 : lp);
        int const ath_type = c->ATHtype;
        int const safe_joint = (c->use_safe_joint_stereo != 0);
        int     stereo_mode, source_freq, non_optimal = 0;
        unsigned long const music_length = v->bytes_written + (unsigned long) v->total_frame_size;
        if (quality < 0)
            quality = 0;
        switch (c->mode) {
        case LH_MODE_MONO:
            stereo_mode = 0;
            break;
        case LH_MODE_STEREO:
            stereo_mode = 1;
            break;
        case LH_MODE_DUAL:
            stereo_mode = 2;
            break;
        case LH_MODE_JOINT_STEREO:
            stereo_mode = c->force_ms ? 4 : 3;
            break;
        default:
            stereo_mode = 7;
        }
        if (v->samplerate_in <= 32000)
            source_freq = 0;
        else if (v->samplerate_in == 48000)
            source_freq = 2;
        else if (v->samplerate_in > 48000)
            source_freq = 3;
        else
            source_freq = 1;
        /* short_blocks: 2 = dispensed, 3 = forced (lame.h short_block_t) */
        if (c->short_blocks == 3 || c->short_blocks == 2
            || (c->lowpassfreq == -1 && c->highpassfreq == -1) || (c->disable_reservoir && c->avg_bitrate < 320)
            || (c->ath_flags & 3) || ath_type == 0 || v->samplerate_in <= 32000)
            non_optimal = 1;
        put_i4(p + k, (uint32_t) quality);
        k += 4;
        memcpy(p + k, "LAME3.99r", 9);  /* get_lame_tag_encoder_short_version() of 3.99.5 */
        k += 9;
        /* tag revision 0 + method (vbr_type_translator, reference VbrTag.c:646): 1 CBR, 5 vbr_mt, 3 vbr_rh, 2 ABR, 4 vbr_mtrh */
        p[k++] = (unsigned char) (c->vbr == 0 ? 1 : (c->vbr == 1 ? 5 : (c->vbr == 2 ? 3 : (c->vbr == 3 ? 2 : 4))));
        p[k++] = lowpass;
        put_i4(p + k, 0);               /* peak signal amplitude: not measured */
        k += 4;
        {
            /* radio ReplayGain (reference VbrTag.c:704-721): name code 1, originator "automatic", sign, |gain| in
             * tenths of a dB; all zero when it was not measured */
            unsigned field = 0;
            if (v->radio_gain_on) {
                int     rg = v->radio_gain;
                rg = rg > 0x1FE ? 0x1FE : (rg < -0x1FE ? -0x1FE : rg);
                field = 0x2000u | 0xC00u | (rg >= 0 ? (unsigned) rg : (0x200u | (unsigned) -rg));
            }
            put_i2(p + k, (uint16_t) field);
        }
        k += 2;
        put_i2(p + k, 0);               /* audiophile ReplayGain */
        k += 2;
        {
            /* --nogap: more titles follow / came before (reference VbrTag.c:728-735) */
            int const more = (v->nogap_total != -1 && v->nogap_current < v->nogap_total - 1);
            int const prev = (v->nogap_total != -1 && v->nogap_current > 0);
            p[k++] = (unsigned char) (ath_type + (1 << 4) + (safe_joint << 5) + (more << 6) + (prev << 7));
        }
        {
            /* CBR: the bit rate; ABR: the mean; VBR: the lowest allowed one (reference VbrTag.c:679-693) */
            int const abr = (c->vbr == 0) ? c->avg_bitrate : (c->vbr == 3) ? c->vbr_avg_bitrate_kbps
                : lh_bitrate_row(c->version)[c->vbr_min_bitrate_index];
            p[k++] = (unsigned char) (abr >= 255 ? 0xFF : abr);
        }
        p[k] = (unsigned char) (LH_ENCDELAY >> 4);
        p[k + 1] = (unsigned char) ((LH_ENCDELAY << 4) + (enc_padding >> 8));
        p[k + 2] = (unsigned char) enc_padding;
        k += 3;
        p[k++] = (unsigned char) (c->noise_shaping + (stereo_mode << 2) + (non_optimal << 5) + (source_freq << 6));
        p[k++] = 0;                     /* MP3 gain */
        /* preset: apply_preset(brate) leaves the bit rate here (presets.c:361, lame.c:1045);
         * the VBR path applies V0..V9 = 500 - 10 q (lame.c:983) */
        put_i2(p + k, (c->vbr == 0 || c->vbr == 3) ? c->avg_bitrate : 500 - 10 * c->vbr_q);
        k += 2;
        put_i4(p + k, (uint32_t) music_length);
        k += 4;
        put_i2(p + k, v->music_crc);
        k += 2;
        for (i = 0; i < k; i++)
            crc = crc16_update(p[i], crc);
        put_i2(p + k, crc);
    }
    return v->total_frame_size;
}
