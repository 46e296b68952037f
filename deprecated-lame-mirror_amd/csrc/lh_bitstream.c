/*
 * lh_bitstream.c -- MPEG-1 Layer III frame assembly on the host (plain C).
 *
 * Input: the per-frame payload the HIP kernels leave in HBM (LhFrameOut: quantised lines with
 * signs, scalefactors, side-information fields, reservoir drains -- what the reference's
 * format_bitstream reads from gfc->l3_side, reference bitstream.c:917-985).  Output: the byte
 * stream, identical to the reference's.  The optional device packer (lh_dev_emit.h) produces the
 * same bytes on the GPU; this file serves the lame_encode_buffer handle path, the batch pack
 * calls and, in the tests, the cross-check of the device packer.
 *
 * Organisation (this file's own):
 *   sink    main-data bits are shifted into a 64-bit accumulator and leave it a byte at a time;
 *           every byte passes sink_byte(), the one place that knows about frame headers;
 *   queue   a frame's header + side information is finished before its main data and waits in a
 *           ring until the byte stream reaches the position the bit reservoir assigned to it
 *           (`due', counted in stream bits); sink_byte() drops it in front of the byte that
 *           arrives at that position;
 *   frame   lh_bs_format_frame() = stuffing owed to the previous frame, queue the header, main
 *           data (scalefactors, big-value pairs, count1 quadruples), stuffing, bookkeeping of the
 *           back pointer, and the consistency checks against the device's reservoir.
 * The bit layouts are those of ISO/IEC 11172-3 section 2.4.1; the stuffing bytes and the
 * order of the drains follow the reference (bitstream.c:223-267, 917-985), because its bytes are
 * the acceptance test.
 */
#include <stdlib.h>
#include <string.h>

#include "lh_host.h"
#include "lh_static_tables.h"

/* error codes left in LhBitstream.error */
enum { BS_FULL = 1, BS_QUEUE = 2, BS_COUNT = 3, BS_BACKPTR = 4, BS_FLUSH = 5, BS_PAYLOAD = 6, BS_ALIGN = 7 };

int
lh_bs_init(LhBitstream * bs)
{
    return lh_bs_init_sized(bs, LH_BS_BUFSIZE);
}

/* with room for `size' finished bytes between two lh_bs_copy calls (a stream of a batch is drained
 * after every few frames and needs far less than a handle that may be fed a minute at once) */
int
lh_bs_init_sized(LhBitstream * bs, int size)
{
    memset(bs, 0, sizeof(*bs));
    bs->buf_size = size;
    bs->buf = (unsigned char *) calloc((size_t) bs->buf_size, 1);
    return bs->buf ? 0 : -2;
}

void
lh_bs_free(LhBitstream * bs)
{
    free(bs->buf);
    bs->buf = 0;
}

int
lh_bs_pending(const LhBitstream * bs)
{
    return bs->fill;
}

/* ---- sink ------------------------------------------------------------------------------ */

/* one finished main-data byte enters the stream; a header whose position has come goes first */
static void
sink_byte(LhBitstream * bs, unsigned byte)
{
    LhQueuedHeader const *h = &bs->queue[bs->q_out];
    if (bs->fill + LH_MAX_HEADER_LEN + 2 >= bs->buf_size) {
        bs->error = BS_FULL;    /* the caller did not drain the buffer */
        return;
    }
    if (h->due == bs->stream_bits) {
        memcpy(bs->buf + bs->fill, h->bytes, (size_t) bs->hdr_len);
        bs->fill += bs->hdr_len;
        bs->stream_bits += 8 * bs->hdr_len;
        bs->q_out = (bs->q_out + 1) % LH_MAX_HEADER_BUF;
    }
    bs->buf[bs->fill++] = (unsigned char) byte;
    bs->stream_bits += 8;
}

/* append the low n bits of v, most significant first (n <= 32) */
static void
sink(LhBitstream * bs, unsigned v, int n)
{
    if (n <= 0)
        return;
    bs->acc = (bs->acc << n) | ((unsigned long long) v & ((1ull << n) - 1ull));
    bs->acc_bits += n;
    while (bs->acc_bits >= 8) {
        bs->acc_bits -= 8;
        sink_byte(bs, (unsigned) (bs->acc >> bs->acc_bits) & 0xffu);
    }
}

/* stream position in bits, counting the bits still in the accumulator */
static int
sink_position(const LhBitstream * bs)
{
    return bs->stream_bits + bs->acc_bits;
}

/* Reservoir stuffing: the encoder's signature first ("LAME" + version, as far as whole bytes
 * fit), then single bits that alternate unless the reservoir is disabled. */
static void
sink_stuffing(LhBitstream * bs, const LhConfig * c, int nbits)
{
    static const char mark[] = "LAME", version[] = "3.99.5";
    int     i;
    for (i = 0; i < 4 && nbits >= 8; i++, nbits -= 8)
        sink(bs, (unsigned char) mark[i], 8);
    if (nbits >= 32)
        for (i = 0; version[i] && nbits >= 8; i++, nbits -= 8)
            sink(bs, (unsigned char) version[i], 8);
    while (nbits-- > 0) {
        sink(bs, (unsigned) bs->stuff_bit, 1);
        if (!c->disable_reservoir)
            bs->stuff_bit ^= 1;
    }
}

/* ---- header queue ------------------------------------------------------------------------ */

typedef struct {
    unsigned char *at;
    int     used;               /* bits */
} HdrWriter;

static void
hdr(HdrWriter * w, unsigned v, int n)
{
    while (n-- > 0) {
        if ((v >> n) & 1u)
            w->at[w->used >> 3] |= (unsigned char) (0x80u >> (w->used & 7));
        w->used++;
    }
}

/* the encoder counts table 14 as an estimate of 16: what is signalled is 16 */
static unsigned
signalled_table(int t)
{
    return (unsigned) (t == 14 ? 16 : t);
}

/* CRC-16 of the protected part: header bytes 2..3 and the side information (polynomial 0x8005,
 * preset 0xffff, ISO/IEC 11172-3 section 2.4.3.1) */
unsigned
lh_header_crc(const unsigned char *h, int len)
{
    unsigned crc = 0xffffu;
    int     i, b;
    for (i = 2; i < len; i++) {
        if (i == 4 || i == 5)
            continue;           /* the CRC word itself */
        for (b = 7; b >= 0; b--) {
            unsigned const in = (h[i] >> b) & 1u, top = (crc >> 15) & 1u;
            crc = (crc << 1) & 0xffffu;
            if (in ^ top)
                crc ^= 0x8005u;
        }
    }
    return crc;
}

/* build frame header + side information in the ring's next slot; it becomes due one frame length
 * after its predecessor */
static void
queue_header(LhBitstream * bs, const LhConfig * c, const LhFrameOut * fo, int back_pointer)
{
    LhQueuedHeader *slot = &bs->queue[bs->q_in];
    int const nch = c->channels;
    HdrWriter w;
    int     gr, ch, k, next;
    memset(slot->bytes, 0, sizeof(slot->bytes));
    w.at = slot->bytes;
    w.used = 0;
    /* header: syncword, ID, layer III, protection, bitrate, sampling frequency, padding, private,
     * mode, mode extension, copyright, original, emphasis */
    hdr(&w, (c->samplerate < 16000) ? 0xffeu : 0xfffu, 12);     /* MPEG-2.5: the shortened syncword */
    hdr(&w, (unsigned) c->version, 1);
    hdr(&w, 1u, 2);
    hdr(&w, c->error_protection ? 0u : 1u, 1);
    hdr(&w, (unsigned) fo->bitrate_index, 4);
    hdr(&w, (unsigned) c->samplerate_index, 2);
    hdr(&w, (unsigned) fo->padding, 1);
    hdr(&w, (unsigned) c->extension, 1);
    hdr(&w, (unsigned) c->mode, 2);
    hdr(&w, (unsigned) fo->mode_ext, 2);
    hdr(&w, (unsigned) c->copyright, 1);
    hdr(&w, (unsigned) c->original, 1);
    hdr(&w, (unsigned) c->emphasis, 2);
    if (c->error_protection)
        w.used += 16;           /* room for the CRC word */
    /* side information: main_data_begin, private bits, (MPEG-1) scfsi, then the granules.  MPEG-2 / 2.5 (reference
     * bitstream.c:420-467): an 8-bit back pointer, one private bit per channel, one granule, a 9-bit scalefac_compress
     * and no preflag bit (it is folded into scalefac_compress >= 500) */
    if (c->version == 1) {
        hdr(&w, (unsigned) back_pointer, 9);
        w.used += (nch == 2) ? 3 : 5;
        for (ch = 0; ch < nch; ch++)
            for (k = 0; k < 4; k++)
                hdr(&w, (unsigned) fo->scfsi[ch][k], 1);
    }
    else {
        hdr(&w, (unsigned) back_pointer, 8);
        w.used += nch;
    }
    for (gr = 0; gr < c->mode_gr; gr++)
        for (ch = 0; ch < nch; ch++) {
            const LhGranule *g = &fo->gr[gr][ch];
            int const switched = (g->block_type != LH_NORM_TYPE);
            hdr(&w, (unsigned) (g->part2_3_length + g->part2_length), 12);
            hdr(&w, (unsigned) (g->big_values / 2), 9);
            hdr(&w, (unsigned) g->global_gain, 8);
            hdr(&w, (unsigned) g->scalefac_compress, (c->version == 1) ? 4 : 9);
            hdr(&w, (unsigned) switched, 1);
            if (switched) {
                hdr(&w, (unsigned) g->block_type, 2);
                hdr(&w, (unsigned) g->mixed_block_flag, 1);
                for (k = 0; k < 2; k++)
                    hdr(&w, signalled_table(g->table_select[k]), 5);
                for (k = 0; k < 3; k++)
                    hdr(&w, (unsigned) g->subblock_gain[k], 3);
            }
            else {
                for (k = 0; k < 3; k++)
                    hdr(&w, signalled_table(g->table_select[k]), 5);
                hdr(&w, (unsigned) g->region0_count, 4);
                hdr(&w, (unsigned) g->region1_count, 3);
            }
            if (c->version == 1)
                hdr(&w, (unsigned) g->preflag, 1);
            hdr(&w, (unsigned) g->scalefac_scale, 1);
            hdr(&w, (unsigned) g->count1table_select, 1);
        }
    if (c->error_protection) {
        unsigned const crc = lh_header_crc(slot->bytes, c->sideinfo_len);
        slot->bytes[4] = (unsigned char) (crc >> 8);
        slot->bytes[5] = (unsigned char) (crc & 0xffu);
    }
    bs->hdr_len = c->sideinfo_len;
    next = (bs->q_in + 1) % LH_MAX_HEADER_BUF;
    bs->queue[next].due = slot->due + fo->frame_bits;
    bs->q_in = next;
    if (next == bs->q_out)
        bs->error = BS_QUEUE;   /* more headers waiting than the ring holds */
}

/* bits that still have to be produced before the stream ends on a frame boundary: up to the
 * position of the last queued header, plus the headers still waiting, plus that frame */
static int
bits_to_frame_end(const LhBitstream * bs, int frame_bits)
{
    int const newest = (bs->q_in + LH_MAX_HEADER_BUF - 1) % LH_MAX_HEADER_BUF;
    int     gap = bs->queue[newest].due - sink_position(bs);
    if (gap >= 0) {
        int const waiting = (newest - bs->q_out + LH_MAX_HEADER_BUF) % LH_MAX_HEADER_BUF + 1;
        gap -= waiting * 8 * bs->hdr_len;
    }
    return gap + frame_bits;
}

/* ---- main data ----------------------------------------------------------------------------- */

static const unsigned char slen_bits[2][16] = {
    {0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4},
    {0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3}
};

/* scalefactors of one granule: the first `divide' bands with slen1 bits, the rest with slen2;
 * -1 marks a band shared with granule 0 through scfsi */
static int
put_scalefactors(LhBitstream * bs, const LhGranule * g)
{
    int     band, written = 0;
    for (band = 0; band < g->sfbmax; band++) {
        int const width = slen_bits[band >= g->sfbdivide][g->scalefac_compress];
        if (g->scalefac[band] < 0)
            continue;
        sink(bs, (unsigned) g->scalefac[band], width);
        written += width;
    }
    return written;
}

/* MPEG-2 / 2.5: the scalefactors go in four partitions, each with its own width; widths and partition sizes follow from
 * scalefac_compress the way a decoder reads them (ISO 13818-3 2.4.3.2; what the reference keeps in gr_info.slen[] and
 * sfb_partition_table, takehiro.c:1266-1296, and writes in bitstream.c:735-770).  Bands shared through scfsi do not
 * exist here; a negative entry is written as 0. */
static int
put_scalefactors_lsf(LhBitstream * bs, const LhGranule * g)
{
    int const sc = g->scalefac_compress;
    int const is_short = (g->block_type == LH_SHORT_TYPE);
    int     slen[4], table, part, band = 0, written = 0, i;
    if (sc < 400) {
        table = 0;
        slen[0] = (sc >> 4) / 5;
        slen[1] = (sc >> 4) % 5;
        slen[2] = (sc >> 2) & 3;
        slen[3] = sc & 3;
    }
    else if (sc < 500) {
        table = 1;
        slen[0] = ((sc - 400) >> 2) / 5;
        slen[1] = ((sc - 400) >> 2) % 5;
        slen[2] = (sc - 400) & 3;
        slen[3] = 0;
    }
    else {
        table = 2;
        slen[0] = (sc - 500) / 3;
        slen[1] = (sc - 500) % 3;
        slen[2] = slen[3] = 0;
    }
    for (part = 0; part < 4; part++) {
        int const n = lh_nr_of_sfb_block[(table * 3 + is_short) * 4 + part];
        for (i = 0; i < n && band < LH_SFBMAX; i++, band++) {
            int const v = g->scalefac[band] < 0 ? 0 : g->scalefac[band];
            sink(bs, (unsigned) v, slen[part]);
            written += slen[part];
        }
    }
    return written;
}

/* big-value pairs [from, to) with code book `book': code word, then per non-zero value its
 * linbits (ESC books, values >= 15) and its sign, x before y */
static int
put_pairs(LhBitstream * bs, unsigned book, int from, int to, const LhGranule * g)
{
    int     i, written = 0;
    if (book == 0 || from >= to)
        return 0;
    {
        unsigned const esc = book > 15u;
        unsigned const linbits = esc ? lh_ht_xlen[book] : 0u;
        unsigned const side = esc ? 16u : lh_ht_xlen[book];
        const uint16_t *code = lh_ht_code + lh_ht_offset[book];
        const uint8_t *length = lh_ht_len + lh_ht_offset[book];       /* code length + one per non-zero value */
        for (i = from; i < to; i += 2) {
            int const sx = g->l3_enc[i], sy = g->l3_enc[i + 1];
            unsigned const ax = (unsigned) (sx < 0 ? -sx : sx), ay = (unsigned) (sy < 0 ? -sy : sy);
            unsigned const cx = (esc && ax > 15u) ? 15u : ax, cy = (esc && ay > 15u) ? 15u : ay;
            unsigned const cell = cx * side + cy;
            int const signs = (ax != 0u) + (ay != 0u);
            sink(bs, code[cell], (int) length[cell] - signs);
            written += length[cell];
            if (ax != 0u) {
                if (esc && ax >= 15u) {
                    sink(bs, ax - 15u, (int) linbits);
                    written += (int) linbits;
                }
                sink(bs, sx < 0, 1);
            }
            if (ay != 0u) {
                if (esc && ay >= 15u) {
                    sink(bs, ay - 15u, (int) linbits);
                    written += (int) linbits;
                }
                sink(bs, sy < 0, 1);
            }
        }
    }
    return written;
}

/* count1 region: quadruples of 0 / +-1 with book A or B; the table entry leaves room for the
 * signs of the non-zero values below the code word */
static int
put_quadruples(LhBitstream * bs, const LhGranule * g)
{
    int const book = 32 + g->count1table_select;
    const uint16_t *code = lh_ht_code + lh_ht_offset[book];
    const uint8_t *length = lh_ht_len + lh_ht_offset[book];
    int     i, k, written = 0;
    for (i = g->big_values; i + 4 <= g->count1; i += 4) {
        unsigned pattern = 0, signs = 0;
        for (k = 0; k < 4; k++) {
            int const v = g->l3_enc[i + k];
            pattern <<= 1;
            if (v != 0) {
                pattern |= 1u;
                signs = (signs << 1) | (unsigned) (v < 0);
            }
        }
        sink(bs, code[pattern] + signs, length[pattern]);
        written += length[pattern];
    }
    return written;
}

/* where the big-value regions of a granule end (in lines), clipped to big_values */
static void
region_ends(const LhGranule * g, const LhTables * t, int end[3])
{
    int     k;
    if (g->block_type == LH_SHORT_TYPE) {
        end[0] = 3 * t->sfb_s[3];
        end[1] = end[2] = g->big_values;
    }
    else {
        int const b1 = g->region0_count + 1, b2 = b1 + g->region1_count + 1;
        end[0] = t->sfb_l[b1];
        end[1] = t->sfb_l[b2];
        end[2] = g->big_values;
    }
    for (k = 0; k < 2; k++)
        if (end[k] > g->big_values)
            end[k] = g->big_values;
}

static int
put_main_data(LhBitstream * bs, const LhConfig * c, const LhTables * t, const LhFrameOut * fo)
{
    int     gr, ch, total = 0;
    for (gr = 0; gr < c->mode_gr; gr++)
        for (ch = 0; ch < c->channels; ch++) {
            const LhGranule *g = &fo->gr[gr][ch];
            int     end[3], bits, k, from = 0;
            bits = (c->version == 1) ? put_scalefactors(bs, g) : put_scalefactors_lsf(bs, g);
            region_ends(g, t, end);
            for (k = 0; k < 3; k++) {
                /* switched blocks signal two books; their third region is empty by construction */
                if (k < 2 || g->block_type == LH_NORM_TYPE)
                    bits += put_pairs(bs, signalled_table(g->table_select[k]), from, end[k], g);
                from = end[k];
            }
            bits += put_quadruples(bs, g);
            /* what was written must be what the quantiser counted */
            if (bits != g->part2_3_length + g->part2_length)
                bs->error = BS_COUNT;
            total += bits;
        }
    return total;
}

/* ---- frame ----------------------------------------------------------------------------------- */

/* a payload whose fields would index outside the code books or the band tables is refused */
static int
payload_plausible(const LhConfig * c, const LhFrameOut * fo)
{
    int     gr, ch, k;
    if (fo->frame_bits <= 0 || fo->frame_bits > 8 * 2880 || fo->resvDrain_pre < 0 || fo->resvDrain_post < 0)
        return 0;
    for (gr = 0; gr < c->mode_gr; gr++)
        for (ch = 0; ch < c->channels; ch++) {
            const LhGranule *g = &fo->gr[gr][ch];
            if (g->big_values < 0 || g->big_values > 576 || (g->big_values & 1))
                return 0;
            if (g->count1 < g->big_values || g->count1 > 576 || ((g->count1 - g->big_values) & 3))
                return 0;
            if ((unsigned) g->region0_count > 15u || (unsigned) g->region1_count > 15u)
                return 0;
            if ((unsigned) g->count1table_select > 1u || (unsigned) g->block_type > 3u || (unsigned) g->global_gain > 255u)
                return 0;
            if ((unsigned) g->scalefac_compress > ((c->version == 1) ? 15u : 511u) || g->sfbmax < 0 || g->sfbmax > LH_SFBMAX || g->sfbdivide < 0)
                return 0;
            for (k = 0; k < 3; k++)
                if ((unsigned) g->table_select[k] > 31u || g->table_select[k] == 4)
                    return 0;
        }
    return 1;
}

/* appends one frame; returns 0, or the negated error code */
int
lh_bs_format_frame(LhBitstream * bs, const LhConfig * c, const LhTables * t, const LhFrameOut * fo)
{
    int     back_pointer, produced;
    if (!payload_plausible(c, fo)) {
        bs->error = BS_PAYLOAD;
        return -1;
    }
    bs->hdr_len = c->sideinfo_len;
    /* whole bytes of the first drain were taken out of the reservoir before this frame's header was
     * built (reference reservoir.c:279-289), so the back pointer shrinks by them */
    sink_stuffing(bs, c, fo->resvDrain_pre);
    back_pointer = bs->main_data_begin - fo->resvDrain_pre / 8;
    queue_header(bs, c, fo, back_pointer);
    produced = 8 * c->sideinfo_len + put_main_data(bs, c, t, fo);
    sink_stuffing(bs, c, fo->resvDrain_post);
    produced += fo->resvDrain_post;
    bs->main_data_begin = back_pointer + (fo->frame_bits - produced) / 8;
    /* the device kept the same books (reference bitstream.c:940-972) */
    if (bs->main_data_begin != fo->main_data_begin || bs->main_data_begin * 8 != fo->resv_size)
        bs->error = BS_BACKPTR;
    if (bits_to_frame_end(bs, fo->frame_bits) != fo->resv_size)
        bs->error = BS_FLUSH;
    if (bs->acc_bits != 0)
        bs->error = BS_ALIGN;   /* a frame's data ends on a byte */
    if (bs->stream_bits > 1000000000) {
        /* keep the position counters small on very long streams */
        int     i;
        for (i = 0; i < LH_MAX_HEADER_BUF; i++)
            bs->queue[i].due -= bs->stream_bits;
        bs->stream_bits = 0;
    }
    return bs->error ? -bs->error : 0;
}

/* end of stream: stuff up to the end of the last frame (reference bitstream.c:863-889) */
void
lh_bs_flush(LhBitstream * bs, const LhConfig * c, const LhFrameOut * last)
{
    int const frame_bits = last ? last->frame_bits
        : 8 * ((c->version + 1) * 72000 * c->avg_bitrate / c->samplerate);
    int const missing = bits_to_frame_end(bs, frame_bits);
    if (missing < 0)
        return;
    bs->hdr_len = c->sideinfo_len;
    sink_stuffing(bs, c, missing);
    bs->main_data_begin = 0;
}

/* hands the finished bytes over; -1 if size != 0 and too small (then nothing is taken) */
int
lh_bs_copy(LhBitstream * bs, unsigned char *out, int size)
{
    int const n = bs->fill;
    if (n <= 0)
        return 0;
    if (size != 0 && n > size)
        return -1;
    memcpy(out, bs->buf, (size_t) n);
    bs->fill = 0;
    return n;
}

/* reference lame.c:1671-1775 + 2041-2120: frames produced for n samples followed by a flush; fs = samples per frame
 * (1152, or 576 for MPEG-2 / 2.5): each fill of <= fs samples is followed by one frame whenever BLKSIZE + fs - FFTOFFSET
 * samples are buffered */
int
lh_total_frames_fs(long n, int fs)
{
    long    mf_size = LH_MF_START, to_encode = LH_ENCDELAY + LH_POSTDELAY;
    long    frames = 0;
    int     end_padding, frames_left;
    int const needed = LH_BLKSIZE + fs - LH_FFTOFFSET;
    to_encode += n;
    if (n > 0) {
        long    total = mf_size + n;
        if (total >= needed)
            frames = (total - needed) / fs + 1;
        to_encode -= fs * frames;
    }
    to_encode -= LH_POSTDELAY;
    end_padding = fs - (int) (to_encode % fs);
    if (end_padding < 576)
        end_padding += fs;
    frames_left = (int) ((to_encode + end_padding) / fs);
    return (int) (frames + frames_left);
}

int
lh_total_frames(long n)
{
    return lh_total_frames_fs(n, 1152);
}

/* end padding that lame_encode_flush adds after n input samples (reference lame.c:2077-2091);
 * the same arithmetic as in lh_total_frames */
int
lh_end_padding_fs(long n, int fs)
{
    long    to_encode = LH_ENCDELAY + LH_POSTDELAY + n;
    int     end_padding;
    int const needed = LH_BLKSIZE + fs - LH_FFTOFFSET;
    if (n > 0) {
        long const total = LH_MF_START + n;
        if (total >= needed)
            to_encode -= fs * ((total - needed) / fs + 1);
    }
    to_encode -= LH_POSTDELAY;
    end_padding = fs - (int) (to_encode % fs);
    if (end_padding < 576)
        end_padding += fs;
    return end_padding;
}

int
lh_end_padding(long n)
{
    return lh_end_padding_fs(n, 1152);
}
