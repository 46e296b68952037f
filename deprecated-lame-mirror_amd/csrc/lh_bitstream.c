/*
 * lh_bitstream.c -- serial MPEG-1 Layer III bit packer, host side (plain C).
 *
 * north_star keeps this stage on the host: it consumes the per-frame payload
 * the HIP kernels leave in HBM (LhFrameOut = what the reference's
 * format_bitstream reads from gfc->l3_side, reference bitstream.c:917-985) and
 * produces the byte stream.  Frame headers + side info are built ahead of time
 * into a small ring and spliced into the main-data bit stream when the running
 * bit count reaches their slot, exactly as the standard's bit reservoir
 * requires (reference bitstream.c:133-185, 320-485).
 *
 * l3_enc arrives as signed 16-bit values: magnitude = quantised line, sign =
 * sign of the spectral line (the reference reads it from xr[], bitstream.c:511,584).
 */
#include <stdlib.h>
#include <string.h>

#include "lh_host.h"
#include "lh_static_tables.h"

static const int slen1_tab[16] = { 0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4 };
static const int slen2_tab[16] = { 0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3 };
static const char lh_short_version[] = "3.99.5";   /* reference version.c:86-110: spliced into ancillary padding */

int
lh_bs_init(LhBitstream * bs)
{
    memset(bs, 0, sizeof(*bs));
    bs->buf_size = LH_BS_BUFSIZE;
    bs->buf = (unsigned char *) calloc((size_t) bs->buf_size, 1);
    if (!bs->buf)
        return -2;
    bs->buf_byte_idx = -1;
    bs->buf_bit_idx = 0;
    bs->totbit = 0;
    bs->h_ptr = bs->w_ptr = 0;
    bs->header[0].write_timing = 0;
    return 0;
}

void
lh_bs_free(LhBitstream * bs)
{
    free(bs->buf);
    bs->buf = 0;
}

static void
splice_header(LhBitstream * bs, int sideinfo_len)
{
    memcpy(&bs->buf[bs->buf_byte_idx], bs->header[bs->w_ptr].buf, (size_t) sideinfo_len);
    bs->buf_byte_idx += sideinfo_len;
    bs->totbit += sideinfo_len * 8;
    bs->w_ptr = (bs->w_ptr + 1) & (LH_MAX_HEADER_BUF - 1);
}

/* append j bits of val (reference bitstream.c:152-185) */
static void
putbits(LhBitstream * bs, int sideinfo_len, int val, int j)
{
    while (j > 0) {
        int     k;
        if (bs->buf_bit_idx == 0) {
            bs->buf_bit_idx = 8;
            bs->buf_byte_idx++;
            if (bs->buf_byte_idx + LH_MAX_HEADER_LEN + 1 >= bs->buf_size) {
                bs->error = 1;  /* caller did not drain the buffer */
                return;
            }
            if (bs->header[bs->w_ptr].write_timing == bs->totbit)
                splice_header(bs, sideinfo_len);
            bs->buf[bs->buf_byte_idx] = 0;
        }
        k = (j < bs->buf_bit_idx) ? j : bs->buf_bit_idx;
        j -= k;
        bs->buf_bit_idx -= k;
        bs->buf[bs->buf_byte_idx] |= ((val >> j) << bs->buf_bit_idx);
        bs->totbit += k;
    }
}

/* ancillary stuffing (reference bitstream.c:223-267) */
static void
drain_into_ancillary(LhBitstream * bs, const LhConfig * c, int remainingBits)
{
    int     i;
    int const sl = c->sideinfo_len;
    if (remainingBits >= 8) {
        putbits(bs, sl, 0x4c, 8);
        remainingBits -= 8;
    }
    if (remainingBits >= 8) {
        putbits(bs, sl, 0x41, 8);
        remainingBits -= 8;
    }
    if (remainingBits >= 8) {
        putbits(bs, sl, 0x4d, 8);
        remainingBits -= 8;
    }
    if (remainingBits >= 8) {
        putbits(bs, sl, 0x45, 8);
        remainingBits -= 8;
    }
    if (remainingBits >= 32) {
        for (i = 0; i < (int) strlen(lh_short_version) && remainingBits >= 8; ++i) {
            remainingBits -= 8;
            putbits(bs, sl, lh_short_version[i], 8);
        }
    }
    for (; remainingBits >= 1; remainingBits -= 1) {
        putbits(bs, sl, bs->ancillary_flag, 1);
        bs->ancillary_flag ^= !c->disable_reservoir;
    }
}

/* header + side info into the ring (reference bitstream.c:270-285, 320-485) */
static void
hdr_bits(LhBitstream * bs, int val, int j)
{
    int     ptr = bs->header[bs->h_ptr].ptr;
    while (j > 0) {
        int const k = (j < 8 - (ptr & 7)) ? j : 8 - (ptr & 7);
        j -= k;
        bs->header[bs->h_ptr].buf[ptr >> 3] |= ((val >> j)) << (8 - (ptr & 7) - k);
        ptr += k;
    }
    bs->header[bs->h_ptr].ptr = ptr;
}

static int
tsel(int t)
{
    return (t == 14) ? 16 : t;  /* table 14 is only a length estimate; 16 carries the same code book */
}

static void
encode_side_info(LhBitstream * bs, const LhConfig * c, const LhFrameOut * fo, int mdb,
                 int bitsPerFrame)
{
    int     gr, ch, band;
    bs->header[bs->h_ptr].ptr = 0;
    memset(bs->header[bs->h_ptr].buf, 0, (size_t) c->sideinfo_len);
    hdr_bits(bs, 0xfff, 12);
    hdr_bits(bs, c->version, 1);
    hdr_bits(bs, 4 - 3, 2);
    hdr_bits(bs, !c->error_protection, 1);
    hdr_bits(bs, fo->bitrate_index, 4);
    hdr_bits(bs, c->samplerate_index, 2);
    hdr_bits(bs, fo->padding, 1);
    hdr_bits(bs, c->extension, 1);
    hdr_bits(bs, c->mode, 2);
    hdr_bits(bs, fo->mode_ext, 2);
    hdr_bits(bs, c->copyright, 1);
    hdr_bits(bs, c->original, 1);
    hdr_bits(bs, c->emphasis, 2);
    if (c->error_protection)
        hdr_bits(bs, 0, 16);    /* CRC word, filled in below */
    hdr_bits(bs, mdb, 9);
    hdr_bits(bs, 0, c->channels == 2 ? 3 : 5);  /* private bits (reference bitstream.c:357-360) */
    for (ch = 0; ch < c->channels; ch++)
        for (band = 0; band < 4; band++)
            hdr_bits(bs, fo->scfsi[ch][band], 1);
    for (gr = 0; gr < 2; gr++) {
        for (ch = 0; ch < c->channels; ch++) {
            const LhGranule *gi = &fo->gr[gr][ch];
            hdr_bits(bs, gi->part2_3_length + gi->part2_length, 12);
            hdr_bits(bs, gi->big_values / 2, 9);
            hdr_bits(bs, gi->global_gain, 8);
            hdr_bits(bs, gi->scalefac_compress, 4);
            if (gi->block_type != LH_NORM_TYPE) {
                hdr_bits(bs, 1, 1);
                hdr_bits(bs, gi->block_type, 2);
                hdr_bits(bs, gi->mixed_block_flag, 1);
                hdr_bits(bs, tsel(gi->table_select[0]), 5);
                hdr_bits(bs, tsel(gi->table_select[1]), 5);
                hdr_bits(bs, gi->subblock_gain[0], 3);
                hdr_bits(bs, gi->subblock_gain[1], 3);
                hdr_bits(bs, gi->subblock_gain[2], 3);
            }
            else {
                hdr_bits(bs, 0, 1);
                hdr_bits(bs, tsel(gi->table_select[0]), 5);
                hdr_bits(bs, tsel(gi->table_select[1]), 5);
                hdr_bits(bs, tsel(gi->table_select[2]), 5);
                hdr_bits(bs, gi->region0_count, 4);
                hdr_bits(bs, gi->region1_count, 3);
            }
            hdr_bits(bs, gi->preflag, 1);
            hdr_bits(bs, gi->scalefac_scale, 1);
            hdr_bits(bs, gi->count1table_select, 1);
        }
    }
    if (c->error_protection) {
        /* CRC-16 (polynomial 0x8005, start 0xffff) over header bytes 2, 3 and the side information
         * (reference bitstream.c:287-318) */
        unsigned char *h = (unsigned char *) bs->header[bs->h_ptr].buf;
        int     crc = 0xffff, i, k;
        for (i = 2; i < c->sideinfo_len; i++) {
            int     value;
            if (i == 4 || i == 5)
                continue;
            value = h[i] << 8;
            for (k = 0; k < 8; k++) {
                value <<= 1;
                crc <<= 1;
                if ((crc ^ value) & 0x10000)
                    crc ^= 0x8005;
            }
        }
        h[4] = (unsigned char) (crc >> 8);
        h[5] = (unsigned char) (crc & 255);
    }
    {
        int const old = bs->h_ptr;
        bs->h_ptr = (old + 1) & (LH_MAX_HEADER_BUF - 1);
        bs->header[bs->h_ptr].write_timing = bs->header[old].write_timing + bitsPerFrame;
        if (bs->h_ptr == bs->w_ptr)
            bs->error = 2;      /* header ring overflow */
    }
}

/* big-value pairs (reference bitstream.c:560-631) */
static int
huffman_pairs(LhBitstream * bs, int sl, unsigned int tableindex, int start, int end,
              const LhGranule * gi)
{
    unsigned int const linbits = lh_ht_xlen[tableindex];
    const uint16_t *codes;
    const uint8_t *lens;
    int     i, bits = 0;
    if (!tableindex)
        return bits;
    codes = lh_ht_code + lh_ht_offset[tableindex];
    lens = lh_ht_len + lh_ht_offset[tableindex];
    for (i = start; i < end; i += 2) {
        int16_t cbits = 0;
        uint16_t xbits = 0;
        unsigned int xlen = lh_ht_xlen[tableindex];
        unsigned int ext = 0;
        int const v1 = gi->l3_enc[i], v2 = gi->l3_enc[i + 1];
        unsigned int x1 = (unsigned int) (v1 < 0 ? -v1 : v1);
        unsigned int x2 = (unsigned int) (v2 < 0 ? -v2 : v2);
        if (x1 != 0u) {
            if (v1 < 0)
                ext++;
            cbits--;
        }
        if (tableindex > 15u) {
            if (x1 >= 15u) {
                uint16_t const linbits_x1 = (uint16_t) (x1 - 15u);
                ext |= (unsigned int) linbits_x1 << 1u;
                xbits = (uint16_t) linbits;
                x1 = 15u;
            }
            if (x2 >= 15u) {
                uint16_t const linbits_x2 = (uint16_t) (x2 - 15u);
                ext <<= linbits;
                ext |= linbits_x2;
                xbits = (uint16_t) (xbits + linbits);
                x2 = 15u;
            }
            xlen = 16;
        }
        if (x2 != 0u) {
            ext <<= 1;
            if (v2 < 0)
                ext++;
            cbits--;
        }
        x1 = x1 * xlen + x2;
        xbits = (uint16_t) (xbits - cbits);
        cbits = (int16_t) (cbits + lens[x1]);
        putbits(bs, sl, codes[x1], cbits);
        putbits(bs, sl, (int) ext, xbits);
        bits += cbits + xbits;
    }
    return bits;
}

/* count1 quadruples (reference bitstream.c:490-551) */
static int
huffman_quads(LhBitstream * bs, int sl, const LhGranule * gi)
{
    int const t = gi->count1table_select + 32;
    const uint16_t *codes = lh_ht_code + lh_ht_offset[t];
    const uint8_t *lens = lh_ht_len + lh_ht_offset[t];
    int     i, bits = 0;
    const int16_t *ix = &gi->l3_enc[gi->big_values];
    for (i = (gi->count1 - gi->big_values) / 4; i > 0; --i) {
        int     huffbits = 0;
        int     p = 0;
        if (ix[0]) {
            p += 8;
            if (ix[0] < 0)
                huffbits++;
        }
        if (ix[1]) {
            p += 4;
            huffbits *= 2;
            if (ix[1] < 0)
                huffbits++;
        }
        if (ix[2]) {
            p += 2;
            huffbits *= 2;
            if (ix[2] < 0)
                huffbits++;
        }
        if (ix[3]) {
            p++;
            huffbits *= 2;
            if (ix[3] < 0)
                huffbits++;
        }
        ix += 4;
        putbits(bs, sl, huffbits + codes[p], lens[p]);
        bits += lens[p];
    }
    return bits;
}

/* scalefactors + Huffman data of one frame (reference bitstream.c:685-790, MPEG-1) */
static int
write_main_data(LhBitstream * bs, const LhConfig * c, const LhTables * t, const LhFrameOut * fo)
{
    int     gr, ch, sfb, data_bits, tot_bits = 0;
    int const sl = c->sideinfo_len;
    for (gr = 0; gr < 2; gr++) {
        for (ch = 0; ch < c->channels; ch++) {
            const LhGranule *gi = &fo->gr[gr][ch];
            int const slen1 = slen1_tab[gi->scalefac_compress];
            int const slen2 = slen2_tab[gi->scalefac_compress];
            int     bigvalues = gi->big_values;
            data_bits = 0;
            for (sfb = 0; sfb < gi->sfbdivide; sfb++) {
                if (gi->scalefac[sfb] == -1)
                    continue;
                putbits(bs, sl, gi->scalefac[sfb], slen1);
                data_bits += slen1;
            }
            for (; sfb < gi->sfbmax; sfb++) {
                if (gi->scalefac[sfb] == -1)
                    continue;
                putbits(bs, sl, gi->scalefac[sfb], slen2);
                data_bits += slen2;
            }
            if (gi->block_type == LH_SHORT_TYPE) {
                int     region1Start = 3 * t->sfb_s[3];
                if (region1Start > bigvalues)
                    region1Start = bigvalues;
                data_bits += huffman_pairs(bs, sl, (unsigned) tsel(gi->table_select[0]), 0, region1Start, gi);
                data_bits += huffman_pairs(bs, sl, (unsigned) tsel(gi->table_select[1]), region1Start, bigvalues, gi);
            }
            else {
                int     i = gi->region0_count + 1;
                int     region1Start = t->sfb_l[i], region2Start;
                i += gi->region1_count + 1;
                region2Start = t->sfb_l[i];
                if (region1Start > bigvalues)
                    region1Start = bigvalues;
                if (region2Start > bigvalues)
                    region2Start = bigvalues;
                data_bits += huffman_pairs(bs, sl, (unsigned) tsel(gi->table_select[0]), 0, region1Start, gi);
                data_bits += huffman_pairs(bs, sl, (unsigned) tsel(gi->table_select[1]), region1Start, region2Start, gi);
                data_bits += huffman_pairs(bs, sl, (unsigned) tsel(gi->table_select[2]), region2Start, bigvalues, gi);
            }
            data_bits += huffman_quads(bs, sl, gi);
            /* the quantiser's bit count must agree with what was written (reference bitstream.c:728) */
            if (data_bits != gi->part2_3_length + gi->part2_length)
                bs->error = 3;
            tot_bits += data_bits;
        }
    }
    return tot_bits;
}

/* reference bitstream.c:804-858 */
static int
compute_flushbits(const LhBitstream * bs, const LhConfig * c, int frame_bits)
{
    int     flushbits, remaining_headers;
    int     last_ptr, first_ptr;
    first_ptr = bs->w_ptr;
    last_ptr = bs->h_ptr - 1;
    if (last_ptr == -1)
        last_ptr = LH_MAX_HEADER_BUF - 1;
    flushbits = bs->header[last_ptr].write_timing - bs->totbit;
    if (flushbits >= 0) {
        remaining_headers = 1 + last_ptr - first_ptr;
        if (last_ptr < first_ptr)
            remaining_headers = 1 + last_ptr - first_ptr + LH_MAX_HEADER_BUF;
        flushbits -= remaining_headers * 8 * c->sideinfo_len;
    }
    flushbits += frame_bits;
    return flushbits;
}

int
lh_bs_format_frame(LhBitstream * bs, const LhConfig * c, const LhTables * t, const LhFrameOut * fo)
{
    int     bits, mdb;
    int const bitsPerFrame = fo->frame_bits;

    /* refuse a payload whose fields would index outside the Huffman tables */
    {
        int     gr, ch, k;
        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < c->channels; ch++) {
                const LhGranule *gi = &fo->gr[gr][ch];
                int     bad = gi->big_values < 0 || gi->big_values > 576 || gi->count1 < gi->big_values
                    || gi->count1 > 576 || (gi->big_values & 1) || ((gi->count1 - gi->big_values) & 3)
                    || gi->region0_count < 0 || gi->region0_count > 15 || gi->region1_count < 0
                    || gi->region1_count > 15 || (unsigned) gi->count1table_select > 1u
                    || (unsigned) gi->block_type > 3u || (unsigned) gi->global_gain > 255u;
                for (k = 0; k < 3; k++)
                    bad |= (unsigned) gi->table_select[k] > 31u || gi->table_select[k] == 4;
                if (bad) {
                    bs->error = 6;
                    return -1;
                }
            }
        if (bitsPerFrame <= 0 || bitsPerFrame > 8 * 2880 || fo->resvDrain_pre < 0 || fo->resvDrain_post < 0) {
            bs->error = 6;
            return -1;
        }
    }
    drain_into_ancillary(bs, c, fo->resvDrain_pre);
    /* ResvFrameEnd moved resvDrain_pre/8 bytes out of the reservoir before the
     * header was built (reference reservoir.c:279-289) */
    mdb = bs->main_data_begin - fo->resvDrain_pre / 8;
    encode_side_info(bs, c, fo, mdb, bitsPerFrame);
    bits = 8 * c->sideinfo_len;
    bits += write_main_data(bs, c, t, fo);
    drain_into_ancillary(bs, c, fo->resvDrain_post);
    bits += fo->resvDrain_post;
    bs->main_data_begin = mdb + (bitsPerFrame - bits) / 8;
    /* consistency with the device-side reservoir (reference bitstream.c:940-972) */
    if (bs->main_data_begin != fo->main_data_begin || bs->main_data_begin * 8 != fo->resv_size)
        bs->error = 4;
    if (compute_flushbits(bs, c, bitsPerFrame) != fo->resv_size)
        bs->error = 5;
    if (bs->totbit > 1000000000) {
        int     i;
        for (i = 0; i < LH_MAX_HEADER_BUF; ++i)
            bs->header[i].write_timing -= bs->totbit;
        bs->totbit = 0;
    }
    return bs->error ? -bs->error : 0;
}

/* reference bitstream.c:863-889 */
void
lh_bs_flush(LhBitstream * bs, const LhConfig * c, const LhFrameOut * last)
{
    int     flushbits;
    int     frame_bits;
    if (last)
        frame_bits = last->frame_bits;
    else
        frame_bits = 8 * ((c->version + 1) * 72000 * c->avg_bitrate / c->samplerate);
    if ((flushbits = compute_flushbits(bs, c, frame_bits)) < 0)
        return;
    drain_into_ancillary(bs, c, flushbits);
    bs->main_data_begin = 0;
}

/* reference bitstream.c:1045-1060 */
int
lh_bs_copy(LhBitstream * bs, unsigned char *out, int size)
{
    int const minimum = bs->buf_byte_idx + 1;
    if (minimum <= 0)
        return 0;
    if (size != 0 && minimum > size)
        return -1;
    memcpy(out, bs->buf, (size_t) minimum);
    bs->buf_byte_idx = -1;
    bs->buf_bit_idx = 0;
    return minimum;
}

/* reference lame.c:1671-1775 + 2041-2120: frames produced for n samples followed by a flush */
int
lh_total_frames(long n)
{
    long    mf_size = LH_MF_START, to_encode = LH_ENCDELAY + LH_POSTDELAY;
    long    frames = 0, fed = 0;
    int     end_padding, frames_left;
    to_encode += n;
    if (n > 0) {
        /* each fill of <=1152 samples is followed by one frame whenever 1904 are buffered */
        long    total = mf_size + n;
        if (total >= LH_MF_NEEDED)
            frames = (total - LH_MF_NEEDED) / 1152 + 1;
        (void) fed;
        to_encode -= 1152 * frames;
    }
    to_encode -= LH_POSTDELAY;
    end_padding = 1152 - (int) (to_encode % 1152);
    if (end_padding < 576)
        end_padding += 1152;
    frames_left = (int) ((to_encode + end_padding) / 1152);
    return (int) (frames + frames_left);
}

/* end padding that lame_encode_flush adds after n input samples (reference lame.c:2077-2091);
 * the same arithmetic as in lh_total_frames */
int
lh_end_padding(long n)
{
    long    to_encode = LH_ENCDELAY + LH_POSTDELAY + n;
    int     end_padding;
    if (n > 0) {
        long const total = LH_MF_START + n;
        if (total >= LH_MF_NEEDED)
            to_encode -= 1152 * ((total - LH_MF_NEEDED) / 1152 + 1);
    }
    to_encode -= LH_POSTDELAY;
    end_padding = 1152 - (int) (to_encode % 1152);
    if (end_padding < 576)
        end_padding += 1152;
    return end_padding;
}
