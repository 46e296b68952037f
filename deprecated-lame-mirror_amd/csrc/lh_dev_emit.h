/*
 * lh_dev_emit.h -- the bit packer on the device (optional).
 *
 * What it produces is format_bitstream's output (reference libmp3lame/bitstream.c:133-985):
 * header + side information of every frame at the frame's byte position, and the main data
 * (scalefactors, Huffman codes, ancillary stuffing) as one continuous bit stream that flows
 * around the headers, back into the space earlier frames left (the bit reservoir).  The
 * reference writes that stream bit by bit through putbits(), splicing a header in whenever the
 * cursor reaches its write_timing.  Here
 *   - each granule/channel packs its own bits right after it was quantised: one lane per
 *     scalefactor band / Huffman pair / count1 quadruple computes its code word and length,
 *     a wave prefix sum gives the bit position, LDS atomic ORs place the words;
 *   - at the end of the frame, when the reservoir arithmetic has fixed the stuffing, the four
 *     parts and the stuffing are shifted into one frame-long bit string in LDS, and its bytes
 *     are scattered to their final positions in the stream's output buffer: position = cursor +
 *     index + side-info length x (headers crossed), with the few pending header positions kept
 *     in the stream state;
 *   - the header itself (a few hundred bits of fields, CRC-16 when asked for) is built by one lane.
 * The host then copies finished MP3 bytes out of HBM (lamehip_batch_get_bytes) instead of
 * LhFrameOut records (4.9 KB per frame) that it would have to Huffman-code itself.
 */
#ifndef LH_DEV_EMIT_H
#define LH_DEV_EMIT_H

#include "lh_dev_quant.h"

#define LH_EMIT_WORDS 132       /* 4095 bits of one granule/channel + slack */

/* out of line and marked cold: with the packer switched off nothing of it should sit between the
 * stages of the encode loop */
#ifdef LH_EMU
#define LH_COLDFN static
#else
#define LH_COLDFN __device__ __attribute__((noinline, cold))
#endif

LH_DEVFN void
lh_lds_or(uint32_t * p, uint32_t v)
{
#ifdef LH_EMU
    *p |= v;
#else
    (void) __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

/* n (<= 32) bits of v, most significant first, at bit position pos of a word array whose bit 31 comes first */
LH_DEVFN void
lh_put_bits(uint32_t * buf, int pos, uint32_t v, int n)
{
    if (n > 0) {
        int const w = pos >> 5, off = pos & 31, room = 32 - off;
        if (n <= room)
            lh_lds_or(&buf[w], (n == 32) ? v : (v << (room - n)));
        else {
            lh_lds_or(&buf[w], v >> (n - room));
            lh_lds_or(&buf[w + 1], v << (32 - (n - room)));
        }
    }
}

/* scalefactors + Huffman data of the granule that sits in Q (reference bitstream.c:490-631,
 * 685-790, MPEG-1) into words[LH_EMIT_WORDS] in HBM; returns the bits written */
LH_DEVFN int
lh_emit_part(const LhCtx & c, LhChanLds & Q, const LhQR & R, const LhGrR & g, const float *xr, uint32_t * words)
{
    const LhQTabs *qt = LH_QT;
    uint32_t *buf = (uint32_t *) Q.xrpow;       /* dead after the quantisation loop */
    const uint32_t *ix2 = (const uint32_t *) Q.ix[0];
    int const lane = c.lane;
    int     pos = 0;
    LH_WAVE_SYNC();
    for (int i = lane; i < LH_EMIT_WORDS; i += 64)
        buf[i] = 0u;
    LH_WAVE_SYNC();
    if (LH_IS_LSF) {
        /* part 2, MPEG-2 / 2.5 (reference bitstream.c:735-770): four partitions with their own widths, both read from
         * scalefac_compress the way lh_scale_bitcount_lsf wrote it; every scalefactor of a partition is written (a
         * negative one as 0) */
        int const sc = g.scalefac_compress, pre = sc >= 500, sh = (R.block_type == LH_SHORT_TYPE);
        uint32_t const ends = pre ? (sh ? 0x24242412u : 0x1515150bu) : (sh ? 0x241b1209u : 0x15100b06u);
        int const w0 = pre ? (sc - 500) / 3 : (sc >> 4) / 5, w1 = pre ? (sc - 500) % 3 : (sc >> 4) % 5;
        int const w2 = pre ? 0 : (sc >> 2) & 3, w3 = pre ? 0 : sc & 3;
        int const part = (lane >= (int) (ends & 255u)) + (lane >= (int) ((ends >> 8) & 255u)) + (lane >= (int) ((ends >> 16) & 255u));
        int const inp = lane < (int) (ends >> 24);
        int const sfv = (inp && lane < LH_SFBMAX) ? Q.sf[0][lane < LH_SFBMAX ? lane : 0] : 0;
        int const sf = sfv < 0 ? 0 : sfv;
        int const len = inp ? (part == 0 ? w0 : part == 1 ? w1 : part == 2 ? w2 : w3) : 0;
        uint32_t const incl = lh_wave_scan_u32((uint32_t) len);
        lh_put_bits(buf, (int) incl - len, (uint32_t) sf, len);
        pos = (int) lh_bcast_u32(incl, 63);
    }
    else {
        /* part 2: one lane per scalefactor band; -1 = shared with granule 0 through scfsi */
        int const s1 = (int) ((0x4433322211130000ull >> (4 * g.scalefac_compress)) & 15u);
        int const s2 = (int) ((0x3232132132103210ull >> (4 * g.scalefac_compress)) & 15u);
        int const sf = (lane < R.sfbmax) ? Q.sf[0][lane < LH_SFBMAX ? lane : 0] : -1;
        int const len = (sf < 0) ? 0 : (lane < R.sfbdivide ? s1 : s2);
        uint32_t const incl = lh_wave_scan_u32((uint32_t) len);
        lh_put_bits(buf, (int) incl - len, (uint32_t) sf, len);
        pos = (int) lh_bcast_u32(incl, 63);
    }
    {
        /* part 3a: big values, one lane per pair, five rounds */
        int const bv2 = g.big_values >> 1;
        int     r1, r2;
        if (R.block_type == LH_SHORT_TYPE) {
            r1 = 3 * (int) qt->sfb_s3;
            r2 = g.big_values;
        }
        else {
            r1 = qt->sfb_l[g.region0_count + 1];
            r2 = qt->sfb_l[g.region0_count + g.region1_count + 2];
        }
        r1 = (r1 > g.big_values ? g.big_values : r1) >> 1;
        r2 = (r2 > g.big_values ? g.big_values : r2) >> 1;
        /* The code tables are in HBM: what depends on a look-up is kept apart from what does not, so that the look-ups of all
         * five rounds (and of the three rounds of quadruples behind them) are in flight together -- round by round every
         * round waited for two dependent ones, 16 trips per granule.  The three regions' tables first (wave-uniform). */
        int     tsel[3], toff[3];
        unsigned tlin[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            int const t = g.table_select[r];
            tsel[r] = (t == 14) ? 16 : t;       /* table 14 is only a length estimate; 16 carries the code book */
            tlin[r] = lh_ht_xlen[tsel[r]];
            toff[r] = lh_ht_offset[tsel[r]];
        }
        int const nq = (g.count1 - g.big_values) >> 2;
        int const t1 = g.count1table_select + 32;
        int const off1 = lh_ht_offset[t1];
        unsigned ext[5], code[5], idx[5];
        int     cbits[5], xbits[5], live[5];
        int     qlen[3], qidx[3], qhb[3];
        uint32_t qval[3];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int const p = lane + 64 * k;
            int const pc = (k < 4 || p < 288) ? p : 287;
            uint32_t const pair = ix2[pc];
            int const in = p < bv2;
            int const reg = (p < r1) ? 0 : (p < r2) ? 1 : 2;
            int const t = reg == 0 ? tsel[0] : reg == 1 ? tsel[1] : tsel[2];
            unsigned const linbits = reg == 0 ? tlin[0] : reg == 1 ? tlin[1] : tlin[2];
            int const off = reg == 0 ? toff[0] : reg == 1 ? toff[1] : toff[2];
            unsigned x1 = pair & 0xffffu, x2 = pair >> 16;
            int const n1 = xr[2 * pc] < 0.0f, n2 = xr[2 * pc + 1] < 0.0f;
            unsigned xlen = linbits;
            ext[k] = 0;
            cbits[k] = 0;
            xbits[k] = 0;
            live[k] = in && t != 0;
            if (live[k]) {
                if (x1 != 0u) {
                    ext[k] = (unsigned) n1;
                    cbits[k]--;
                }
                if (t > 15) {
                    if (x1 >= 15u) {
                        ext[k] |= (x1 - 15u) << 1;
                        xbits[k] = (int) linbits;
                        x1 = 15u;
                    }
                    if (x2 >= 15u) {
                        ext[k] <<= linbits;
                        ext[k] |= (x2 - 15u);
                        xbits[k] += (int) linbits;
                        x2 = 15u;
                    }
                    xlen = 16;
                }
                if (x2 != 0u) {
                    ext[k] <<= 1;
                    ext[k] |= (unsigned) n2;
                    cbits[k]--;
                }
            }
            idx[k] = live[k] ? (unsigned) off + x1 * xlen + x2 : 0u;
        }
        {
            /* part 3b's indices: count1 quadruples, three rounds */
            const int16_t *ix = Q.ix[0];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                int const qd = lane + 64 * k;
                int const in = qd < nq;
                int const base = in ? g.big_values + 4 * qd : 0;
                int     p = 0, hb = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    int const v = ix[base + u];
                    if (v) {
                        p += 8 >> u;
                        hb = hb * 2 + (xr[base + u] < 0.0f ? 1 : 0);
                    }
                }
                qidx[k] = in ? off1 + p : -1;
                qhb[k] = hb;
            }
        }
        {
            /* every look-up of the granule */
            unsigned hl[5], hc[5];
            int     ql[3];
            unsigned qc[3];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                hl[k] = lh_ht_len[idx[k]];
                hc[k] = lh_ht_code[idx[k]];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                ql[k] = lh_ht_len[qidx[k] < 0 ? 0 : qidx[k]];
                qc[k] = lh_ht_code[qidx[k] < 0 ? 0 : qidx[k]];
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                code[k] = 0;
                if (live[k]) {
                    xbits[k] -= cbits[k];
                    cbits[k] += (int) hl[k];
                    code[k] = hc[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                qlen[k] = qidx[k] < 0 ? 0 : ql[k];
                qval[k] = qidx[k] < 0 ? 0u : (uint32_t) qhb[k] + qc[k];
            }
        }
        /* part 3a: big values, one lane per pair, five rounds */
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int const len = cbits[k] + xbits[k];
            uint32_t const incl = lh_wave_scan_u32((uint32_t) len);
            int const at = pos + (int) incl - len;
            lh_put_bits(buf, at, code[k], cbits[k]);
            lh_put_bits(buf, at + cbits[k], ext[k], xbits[k]);
            pos += (int) lh_bcast_u32(incl, 63);
        }
        /* part 3b: count1 quadruples, three rounds */
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint32_t const incl = lh_wave_scan_u32((uint32_t) qlen[k]);
            lh_put_bits(buf, pos + (int) incl - qlen[k], qval[k], qlen[k]);
            pos += (int) lh_bcast_u32(incl, 63);
        }
    }
    LH_WAVE_SYNC();
    for (int i = lane; i < LH_EMIT_WORDS; i += 64)
        words[i] = buf[i];
    return pos;
}

/* bit i of `remaining' bits of ancillary stuffing (drain_into_ancillary, reference bitstream.c:223-267):
 * "LAME", then the version string when at least 32 bits are left after it, then a bit pattern */
LH_DEVFN int
lh_anc_bit(int i, int remaining, int flag, int toggling)
{
    int const nl = remaining >= 32 ? 4 : (remaining >> 3);              /* bytes of "LAME" */
    int const rem = remaining - 8 * nl;
    int const nv = (rem >= 32) ? ((rem >> 3) < 6 ? (rem >> 3) : 6) : 0; /* bytes of "3.99.5" */
    int const byte = i >> 3;
    if (byte < nl + nv) {
        unsigned const ch = (byte < nl) ? ((0x454d414cu >> (8 * byte)) & 255u)         /* 'L' 'A' 'M' 'E' */
            : ((unsigned) ((0x352e39392e33ull >> (8 * (byte - nl))) & 255ull));         /* '3' '.' '9' '9' '.' '5' */
        return (int) ((ch >> (7 - (i & 7))) & 1u);
    }
    return flag ^ (toggling ? ((i - 8 * (nl + nv)) & 1) : 0);
}

/* the flag after `remaining' bits of stuffing */
LH_DEVFN int
lh_anc_flag_after(int remaining, int flag, int toggling)
{
    int const nl = remaining >= 32 ? 4 : (remaining >> 3);
    int const rem = remaining - 8 * nl;
    int const nv = (rem >= 32) ? ((rem >> 3) < 6 ? (rem >> 3) : 6) : 0;
    int const tail = remaining - 8 * (nl + nv);
    return flag ^ (toggling ? (tail & 1) : 0);
}

/* the side information of a granule as LhGranule carries it behind the quantised lines and the scalefactors: eight words,
 * which lh_emit_frame brings into LDS for all granules of the frame at once (the header's one lane then reads LDS, not ~60
 * fields of HBM one after the other) */
struct LhSideInfo {
    int16_t part2_3_length, part2_length, big_values, count1, global_gain, scalefac_compress;
    int8_t  block_type, mixed_block_flag, table_select[3], subblock_gain[3];
    int8_t  region0_count, region1_count, preflag, scalefac_scale, count1table_select, sfbmax, sfbdivide, pad1;
    int16_t count1bits, pad2;
};
#define LH_SIDE_OFF ((int) __builtin_offsetof(LhGranule, part2_3_length))
static_assert(sizeof(LhSideInfo) == 32 && LH_SIDE_OFF % 4 == 0 && sizeof(LhGranule) == LH_SIDE_OFF + 32 && sizeof(LhGranule) % 4 == 0
              && __builtin_offsetof(LhGranule, count1bits) - LH_SIDE_OFF == __builtin_offsetof(LhSideInfo, count1bits)
              && __builtin_offsetof(LhGranule, region0_count) - LH_SIDE_OFF == __builtin_offsetof(LhSideInfo, region0_count),
              "LhSideInfo mirrors the tail of LhGranule");

/* header + side information of one frame (reference bitstream.c:320-487) into hw[] (big-endian words: byte i of the header is
 * bits 31 - 8 (i & 3) .. of hw[i >> 2]); one lane, serial.  The fields run through a 64-bit accumulator in registers and leave
 * it a word at a time -- a byte array indexed by the running bit position sits in scratch memory, where every field was a
 * dependent load + store through the vector memory path: 80 k cycles per frame, nine tenths of what the packer cost. */
LH_DEVFN void
lh_emit_header(const LhConfig * cfg, const LhSideInfo * si, int mdb, int bitrate_index, int padding, int mode_ext,
               const int8_t * scfsi, uint32_t * hw)
{
    unsigned long long acc = 0;
    int     nacc = 0, wi = 0;
    int const sl = cfg->sideinfo_len;
#define LH_HB(val, n) do { int const n_ = (n); \
        if (n_ > 0) { acc = (acc << n_) | ((unsigned long long) (unsigned) (val) & ((1ull << n_) - 1ull)); nacc += n_; \
            if (nacc >= 32) { hw[wi++] = (uint32_t) (acc >> (nacc - 32)); nacc -= 32; } } } while (0)
    LH_HB(cfg->samplerate < 16000 ? 0xffe : 0xfff, 12);
    LH_HB(cfg->version, 1);
    LH_HB(4 - 3, 2);
    LH_HB(!cfg->error_protection, 1);
    LH_HB(bitrate_index, 4);
    LH_HB(cfg->samplerate_index, 2);
    LH_HB(padding, 1);
    LH_HB(cfg->extension, 1);
    LH_HB(cfg->mode, 2);
    LH_HB(mode_ext, 2);
    LH_HB(cfg->copyright, 1);
    LH_HB(cfg->original, 1);
    LH_HB(cfg->emphasis, 2);
    if (cfg->error_protection)
        LH_HB(0, 16);
    if (cfg->version == 1) {
        LH_HB(mdb, 9);
        LH_HB(0, cfg->channels == 2 ? 3 : 5);
        for (int ch = 0; ch < cfg->channels; ch++)
            for (int band = 0; band < 4; band++)
                LH_HB(scfsi[4 * ch + band], 1);
    }
    else {
        /* MPEG-2 / 2.5 (reference bitstream.c:420-467) */
        LH_HB(mdb, 8);
        LH_HB(0, cfg->channels);
    }
    for (int gr = 0; gr < LH_NGR; gr++)
        for (int ch = 0; ch < cfg->channels; ch++) {
            const LhSideInfo *gi = &si[gr * cfg->channels + ch];
            LH_HB(gi->part2_3_length + gi->part2_length, 12);
            LH_HB(gi->big_values / 2, 9);
            LH_HB(gi->global_gain, 8);
            LH_HB(gi->scalefac_compress, cfg->version == 1 ? 4 : 9);
            if (gi->block_type != LH_NORM_TYPE) {
                LH_HB(1, 1);
                LH_HB(gi->block_type, 2);
                LH_HB(gi->mixed_block_flag, 1);
                LH_HB(gi->table_select[0] == 14 ? 16 : gi->table_select[0], 5);
                LH_HB(gi->table_select[1] == 14 ? 16 : gi->table_select[1], 5);
                LH_HB(gi->subblock_gain[0], 3);
                LH_HB(gi->subblock_gain[1], 3);
                LH_HB(gi->subblock_gain[2], 3);
            }
            else {
                LH_HB(0, 1);
                LH_HB(gi->table_select[0] == 14 ? 16 : gi->table_select[0], 5);
                LH_HB(gi->table_select[1] == 14 ? 16 : gi->table_select[1], 5);
                LH_HB(gi->table_select[2] == 14 ? 16 : gi->table_select[2], 5);
                LH_HB(gi->region0_count, 4);
                LH_HB(gi->region1_count, 3);
            }
            if (cfg->version == 1)
                LH_HB(gi->preflag, 1);
            LH_HB(gi->scalefac_scale, 1);
            LH_HB(gi->count1table_select, 1);
        }
#undef LH_HB
    if (nacc > 0)
        hw[wi++] = (uint32_t) (acc << (32 - nacc));
    while (wi < (sl + 3) / 4 + 1)
        hw[wi++] = 0u;
    if (cfg->error_protection) {
        int     crc = 0xffff;
        for (int i = 2; i < sl; i++) {
            int     value;
            if (i == 4 || i == 5)
                continue;
            value = (int) ((hw[i >> 2] >> (24 - 8 * (i & 3))) & 255u) << 8;
            for (int k = 0; k < 8; k++) {
                value <<= 1;
                crc <<= 1;
                if ((crc ^ value) & 0x10000)
                    crc ^= 0x8005;
            }
        }
        hw[1] = (hw[1] & 0x0000ffffu) | ((uint32_t) (crc & 0xffff) << 16);    /* bytes 4 and 5 */
    }
}

/* output position of main-data byte j counted from the cursor: every pending header start that is
 * reached pushes the rest back by the side-info length */
LH_DEVFN long long
lh_emit_pos(long long cursor, long long j, const long long *hq, int nq, int sl)
{
    long long p = cursor + j;
    for (int k = 0; k < nq; k++)
        if (p >= hq[k])
            p += sl;
    return p;
}

/* out-of-line entry for the quantisation stages: the granule's R / g come through the wave's LDS slot
 * (keeps the packer's registers and code out of the loops that run when it is switched off) */
LH_COLDFN void
lh_emit_part_stage(int qch, int gr)
{
    LhCtx const c = lh_ctx_load();
    LhQR const R = lh_uniform(lh_lds.rg[qch].R);
    LhGrR const g = lh_uniform(lh_lds.rg[qch].g);
    int const nb = lh_emit_part(c, lh_lds.u.quant.ch[qch], R, g, lh_lds.xr[qch][lh_uni_i(gr)],
                                c.st->em_part[lh_uni_i(gr)][qch]);
    if (nb != g.part2_3_length + g.part2_length && c.lane == 0)
        lh_lds_or((uint32_t *) &lh_lds.ss.status, 4u);      /* the packed bits disagree with the quantiser's count */
}

/* One frame's bytes, whole workgroup.  `nbits' = pre + sum of the parts + post (a multiple of 8 by
 * the reservoir's construction); part k of `np' parts has plen[k] bits in st->em_part[...]. */
LH_COLDFN void
lh_emit_frame(const LhFrameOut * fo_in, int drain_pre, int drain_post, int frame_bytes,
              int mdb, int bitrate_index, int padding, int mode_ext, int flush)
{
    LhCtx const c = lh_ctx_load();
    const LhFrameOut *fo = LH_AS_GLOBAL(const LhFrameOut, fo_in);
    uint8_t *bytes = LH_AS_GLOBAL(uint8_t, lh_lds.ctx.bytes);
    LhLds & L = lh_lds;
    const LhConfig *cfg = c.cfg;
    LhStreamState *st = c.st;
    uint32_t *fb = (uint32_t *) &L.u.quant.ch[0];      /* the quantiser's working set is dead: 4.6 KB of xrpow + save_xrpow */
    uint32_t *hw = (uint32_t *) &L.u.quant.ch[1];      /* (the other channel's: the header's words, the frame's side information) */
    uint32_t *hs = hw + 16;
    const LhSideInfo *si = (const LhSideInfo *) hs;
    int const tid = c.tid, nch = cfg->channels, sl = cfg->sideinfo_len;
    int const toggling = !cfg->disable_reservoir;
    int     plen[4], poff[4], np = 0, nbits, flag;
    /* Frame headers the main data still has to jump over.  MPEG-1: a frame holds at least 60 bytes of main data and the
     * back pointer reaches 511 bytes: never more than 10.  MPEG-2 / 2.5: frames of 24 bytes with 21 of side information
     * exist (8 kb/s at 24 kHz) while the back pointer reaches 255 bytes: up to 85 + the frames in flight. */
    constexpr int NHQ = LH_IS_LSF ? LH_EMIT_HQ_MAX : 16;
    long long cursor, hq[NHQ];
    int     nq;
    uint8_t *out = bytes + c.d.bytes_base;
    drain_pre = lh_uni_i(drain_pre);
    drain_post = lh_uni_i(drain_post);
    frame_bytes = lh_uni_i(frame_bytes);
    flush = lh_uni_i(flush);

    LH_PT(t_em);
    LH_SYNC_WG();
    LH_PA(14, t_em);
    /* the frame's side information (eight words per granule and channel) and scfsi, from the record the granules' last stage
     * wrote, into LDS */
    if (tid < 8 * LH_NGR * nch) {
        int const k = tid >> 3, gr = k / nch, ch = k - gr * nch;
        hs[tid] = ((const uint32_t *) ((const char *) &fo->gr[gr][ch] + LH_SIDE_OFF))[tid & 7];
    }
    else if (tid < 8 * LH_NGR * nch + 2)
        hs[32 + (tid - 8 * LH_NGR * nch)] = ((const uint32_t *) &fo->scfsi[0][0])[tid - 8 * LH_NGR * nch];
    LH_SYNC_WG();
    nbits = drain_pre;
    for (int gr = 0; gr < LH_NGR; gr++)
        for (int ch = 0; ch < nch; ch++) {
            const LhSideInfo *gi = &si[gr * nch + ch];
            poff[np] = nbits;
            plen[np] = gi->part2_3_length + gi->part2_length;
            nbits += plen[np];
            np++;
        }
    nbits += drain_post;
    cursor = st->em_cursor;
    nq = st->em_nq;
    for (int k = 0; k < NHQ; k++)
        hq[k] = (k < nq) ? st->em_hq[k] : 0;
    flag = st->em_anc_flag;
    /* this frame's header joins the pending ones */
    {
        long long const here = st->em_next_header;
        int const queue_full = nq >= NHQ;
        if (!queue_full)
            hq[nq++] = here;
        LH_SYNC_WG();           /* everyone has read the state */
        if (tid == 0) {
            /* status bit 8: something could not be written (slice of the byte pool too small, or more
             * headers pending than the queue holds): the host refuses the stream's bytes */
            if (queue_full || here + sl > c.d.bytes_cap)
                lh_lds_or((uint32_t *) &lh_lds.ss.status, 8u);
            lh_emit_header(cfg, si, mdb, bitrate_index, padding, mode_ext, (const int8_t *) (hs + 32), hw);
            st->em_next_header = here + frame_bytes;
        }
        LH_SYNC_WG();
        /* a lane per header byte */
        if (tid < sl && here + sl <= c.d.bytes_cap)
            out[here + tid] = (uint8_t) ((hw[tid >> 2] >> (24 - 8 * (tid & 3))) & 255u);
    }
    LH_PA(15, t_em);
    for (int round = 0; round < (flush ? 2 : 1); round++) {
        int     nbytes;
        if (round == 1) {
            /* flush_bitstream (reference bitstream.c:863-889): stuffing up to the end of the last frame */
            long long const end = st->em_next_header;
            nbits = (int) (8 * (end - cursor - (long long) sl * nq));
            drain_pre = nbits;
            drain_post = 0;
            np = 0;
            if (nbits <= 0)
                break;
        }
        nbytes = nbits >> 3;
        for (int i = tid; i < ((nbits + 31) >> 5) + 1; i += LH_NT)
            fb[i] = 0u;
        LH_SYNC_WG();
        /* stuffing before and after the parts; the flag of the bit pattern runs through both */
        for (int i = tid; i < drain_pre; i += LH_NT)
            if (lh_anc_bit(i, drain_pre, flag, toggling))
                lh_lds_or(&fb[i >> 5], 1u << (31 - (i & 31)));
        {
            int const f1 = lh_anc_flag_after(drain_pre, flag, toggling);
            int const at = nbits - drain_post;
            for (int i = tid; i < drain_post; i += LH_NT)
                if (lh_anc_bit(i, drain_post, f1, toggling))
                    lh_lds_or(&fb[(at + i) >> 5], 1u << (31 - ((at + i) & 31)));
            flag = lh_anc_flag_after(drain_post, f1, toggling);
        }
        for (int k = 0; k < np; k++) {
            const uint32_t *w = st->em_part[k / nch][k % nch];
            int const nw = (plen[k] + 31) >> 5;
            for (int i = tid; i < nw; i += LH_NT) {
                int const n = (plen[k] - 32 * i) < 32 ? (plen[k] - 32 * i) : 32;
                uint32_t const v = w[i];
                lh_put_bits(fb, poff[k] + 32 * i, (n == 32) ? v : (v >> (32 - n)), n);
            }
        }
        LH_SYNC_WG();
        LH_PA(16, t_em);
        for (int j = tid; j < nbytes; j += LH_NT) {
            long long const p = lh_emit_pos(cursor, j, hq, nq, sl);
            if (p < c.d.bytes_cap)
                out[p] = (uint8_t) ((fb[j >> 2] >> (24 - 8 * (j & 3))) & 255u);
            else
                lh_lds_or((uint32_t *) &lh_lds.ss.status, 8u);
        }
        /* cursor and the pending headers behind it */
        cursor = lh_emit_pos(cursor, nbytes, hq, nq, sl);
        {
            int     keep = 0;
            for (int k = 0; k < nq; k++)
                if (hq[k] >= cursor)
                    hq[keep++] = hq[k];
            nq = keep;
        }
        if ((nbits & 7) != 0 && tid == 0)
            lh_lds_or((uint32_t *) &lh_lds.ss.status, 2u);    /* the reservoir arithmetic should make every frame's main data whole bytes */
        LH_SYNC_WG();
    }
    LH_PA(17, t_em);
    if (tid == 0) {
        st->em_cursor = cursor;
        st->em_nq = nq;
        for (int k = 0; k < NHQ; k++)
            st->em_hq[k] = (k < nq) ? hq[k] : 0;
        st->em_anc_flag = flag;
    }
    LH_SYNC_WG();
}

#endif
