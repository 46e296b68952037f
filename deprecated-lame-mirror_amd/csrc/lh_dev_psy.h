/*
 * lh_dev_psy.h -- psycho-acoustic model of the encode kernel: masking with its recurrences, mid/side thresholds,
 * perceptual entropy and the granule driver (the transforms and table walks are in lh_dev_psy_core.h).
 */
#ifndef LH_DEV_PSY_H
#define LH_DEV_PSY_H

#include "lh_dev_psy_core.h"

/* reference psymodel.c:1326-1388; one lane per partition, eb/thr are [4][64] in LDS */
LH_DEVFN void
lh_ms_thresholds(const LhCtx & c, const float *eb, float *thr, const float *cb_mld,
                 const float *ath_cb, float athlower, float msfix, int n)
{
    int const b = c.lane;
    if (b < n) {
        float const msfix2 = msfix * 2.f;
        float   rside, rmid;
        float const ebM = eb[2 * 64 + b];
        float const ebS = eb[3 * 64 + b];
        float const thmL = thr[0 * 64 + b];
        float const thmR = thr[1 * 64 + b];
        float   thmM = thr[2 * 64 + b];
        float   thmS = thr[3 * 64 + b];
        if (thmL <= 1.58f * thmR && thmR <= 1.58f * thmL) {
            float const mld_m = cb_mld[b] * ebS;
            float const mld_s = cb_mld[b] * ebM;
            float const tmp_m = __builtin_fminf(thmS, mld_m);
            float const tmp_s = __builtin_fminf(thmM, mld_s);
            rmid = __builtin_fmaxf(thmM, tmp_m);
            rside = __builtin_fmaxf(thmS, tmp_s);
        }
        else {
            rmid = thmM;
            rside = thmS;
        }
        if (msfix > 0.f) {
            float   thmLR, thmMS;
            float const ath = ath_cb[b] * athlower;
            float const tmp_l = __builtin_fmaxf(thmL, ath);
            float const tmp_r = __builtin_fmaxf(thmR, ath);
            thmLR = __builtin_fminf(tmp_l, tmp_r);
            thmM = __builtin_fmaxf(rmid, ath);
            thmS = __builtin_fmaxf(rside, ath);
            thmMS = thmM + thmS;
            if (thmMS > 0.f && (thmLR * msfix2) < thmMS) {
                float const f = thmLR * msfix2 / thmMS;
                thmM *= f;
                thmS *= f;
            }
            rmid = __builtin_fminf(thmM, rmid);
            rside = __builtin_fminf(thmS, rside);
        }
        if (rmid > ebM)
            rmid = ebM;
        if (rside > ebS)
            rside = ebS;
        thr[2 * 64 + b] = rmid;
        thr[3 * 64 + b] = rside;
    }
}

/* perceptual entropy of one pseudo-channel (reference psymodel.c:458-553); serial, one lane.
 * en/thm point at a 61-entry III_psy_xmin image (l[22], s[13][3]) */
LH_DEVFN float
lh_pecalc(const LhTables * T, const float *en, const float *thm, float masking_lower, int is_short)
{
    float   pe;
    if (is_short) {
        pe = 1236.28f / 4;
        for (int sb = 0; sb < LH_SBMAX_S - 1; sb++) {
            for (int sblock = 0; sblock < 3; sblock++) {
                float const t = thm[22 + sb * 3 + sblock];
                if (t > 0.0f) {
                    float const x = t * masking_lower;
                    float const e = en[22 + sb * 3 + sblock];
                    if (e > x) {
                        if (e > x * 1e10f)
                            pe = (float) (pe + lh_regcoef_s[sb] * (10.0f * 2.30258509299404568402));
                        else
                            pe = (float) (pe + lh_regcoef_s[sb] *
                                          (lh_fast_log2(T->log_table, e / x) * LH_LOG2_OVER_LOG10));
                    }
                }
            }
        }
    }
    else {
        pe = 1124.23f / 4;
        for (int sb = 0; sb < LH_SBMAX_L - 1; sb++) {
            float const t = thm[sb];
            if (t > 0.0f) {
                float const x = t * masking_lower;
                float const e = en[sb];
                if (e > x) {
                    if (e > x * 1e10f)
                        pe = (float) (pe + lh_regcoef_l[sb] * (10.0f * 2.30258509299404568402));
                    else
                        pe = (float) (pe + lh_regcoef_l[sb] *
                                      (lh_fast_log2(T->log_table, e / x) * LH_LOG2_OVER_LOG10));
                }
            }
        }
    }
    return pe;
}

#ifdef LH_SPLIT
/* The encode kernel of the split pipeline (lh_kernels.hip with -DLH_SPLIT): the analysis kernels (lh_analysis.hip) have
 * done everything of a granule's masking that depends on the PCM alone -- lh_compute_masking < NC, 1 > left the partition
 * energies, the spread energies x weight and the caps in HBM (LhMidMask) --, and what is left are the recurrences: the
 * pre-echo clamp against the two previous long calls, the cap, and the masking adjustment the last frame's loop left
 * (reference psymodel.c:1208-1262 / :1100-1131).  One lane per partition, straight from HBM into registers. */
struct LhTailChan {
    int     chn;
    const LhMidMask *raw;
    float  *eb, *thr;           /* partition energies / thresholds out (LDS, 64 each) */
    float  *nb1, *nb2;
};

template < int NC > LH_DEVFN void
lh_masking_tail(const LhCtx & c, int is_long, const LhTailChan (&ch)[NC])
{
    LhPsyBand const *gd = is_long ? &c.T->psy_l : &c.T->psy_s;
    int const b = c.lane;
    int const np = gd->npart;
    int const on = b < np;
    float const t_mlow = gd->masking_lower[on ? b : 0];
    float   ebb[NC], ecb[NC], cap[NC], th[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) {
        ebb[q] = ch[q].raw->v[0][b];
        ecb[q] = ch[q].raw->v[1][b];
        cap[q] = ch[q].raw->v[2][b];
        th[q] = 0;
    }
    if (on) {
        float const masking_lower = t_mlow * lh_lds.ss.masking_lower;
#pragma unroll
        for (int q = 0; q < NC; q++) {
            float   x, e = ecb[q], t;
            if (is_long) {
                int const bt_old = lh_lds.ss.blocktype_old[ch[q].chn & 1];
                float const n1v = *ch[q].nb1, n2v = *ch[q].nb2;
                if (bt_old == LH_SHORT_TYPE) {
                    float const ecb_limit = LH_RPELEV * n1v;
                    if (ecb_limit > 0)
                        t = __builtin_fminf(e, ecb_limit);
                    else {
                        float const alt = (float) (ebb[q] * LH_PREECHO_ATT2);
                        t = __builtin_fminf(e, alt);
                    }
                }
                else {
                    float   lim2 = LH_RPELEV2 * n2v;
                    float   lim1 = LH_RPELEV * n1v;
                    float   lim;
                    if (lim2 <= 0)
                        lim2 = e;
                    if (lim1 <= 0)
                        lim1 = e;
                    if (bt_old == LH_NORM_TYPE)
                        lim = __builtin_fminf(lim1, lim2);
                    else
                        lim = lim1;
                    t = __builtin_fminf(e, lim);
                }
                *ch[q].nb2 = n1v;
                *ch[q].nb1 = e;
            }
            else
                t = e;
            x = cap[q];
            if (t > x)
                t = x;
            if (masking_lower > 1)
                t *= masking_lower;
            if (t > ebb[q])
                t = ebb[q];
            if (masking_lower < 1)
                t *= masking_lower;
            th[q] = t;
        }
    }
#pragma unroll
    for (int q = 0; q < NC; q++) {
        ch[q].eb[b] = ebb[q];   /* 0 above the last partition */
        ch[q].thr[b] = th[q];
    }
    LH_WAVE_SYNC_MEM();
}
#endif

/* ------------------------------------------------------------------ */
/* One granule of the psycho-acoustic model for the whole workgroup     */
/* (reference L3psycho_anal_vbr, psymodel.c:1397-1597).                 */
LH_STAGEFN LhPsyCarry
lh_psy_granule(int gr, LhPsyCarry nb)
{
    LhCtx const c = lh_ctx_load();
    LhLds & L = lh_lds;
    const LhConfig *cfg = c.cfg;
    const LhTables *T = c.T;
    LhStreamState *st = c.st;
    LhPsyLds & P = L.u.psy;
    int const lane = c.lane, w = c.wave;
    /* mono: one channel, wave 1 only keeps pace through the workgroup barriers (its per-wave work on
     * the duplicated PCM is never read) */
    int const n_chn_psy = (cfg->mode == LH_MODE_JOINT_STEREO) ? 4 : cfg->channels;
    int const bufbase = 576 + gr * 576 - LH_FFTOFFSET;      /* bufp[ch] = &inbuf[ch][bufbase] */

    /* (1) one-granule delay: what the last call computed (ring slot `was') are this granule's ratios
     * and its last_thm; this call's band energies / thresholds go to slot `now' (see LhLds.psy_en).
     * Every band of a channel the model runs on is rewritten by each call (22 long bands, then the
     * long->short estimate or the short-block values for the other 39), so nothing needs copying. */
    int const was = (lh_uni_i(L.psy_slot) + gr) % 3, now = (was + 1) % 3;
    if (c.tid < 4)
        L.tot_ener[gr][c.tid] = lh_lds.ss.tot_ener[c.tid];

    LH_PT(t_psy0);
#ifdef LH_SPLIT
    /* (2) - (5) attack detection, the long transforms, spectra and their sums: done by the analysis kernels; what this
     * granule has of them is in its LhMidGr record.  The state words the fused kernel keeps up to date are kept up to date
     * here as well (either kernel may take the stream's next launch). */
    const LhMidFrame *mid = LH_AS_GLOBAL(const LhMidFrame, lh_lds.ctx.mid);
    const LhMidGr *mg = &P.mid.small.gr[gr];            /* (staged into LDS at the top of the frame) */
    const LhMidLong *mlong = &P.mid.lng;
    const LhMidShort *mshort = &mid->shrt;              /* (HBM: short-block granules only) */
    LH_SYNC_WG_LDS();           /* the previous call's total energies are read */
    for (int pass = 0; pass < 2; pass++) {
        int const chn = w + 2 * pass;
        if (chn < n_chn_psy && lane >= 3 && lane < 12) {
            float const pk = mg->peak[chn][lane - 3];
            lh_lds.ss.last_en_subshort[chn][lane - 3] = pk < 1.0f ? 1.0f : pk;
        }
    }
    if (c.tid < 16)
        L.ns_attacks[c.tid >> 2][c.tid & 3] = mg->ns_attacks[c.tid >> 2][c.tid & 3];
    else if (c.tid < 28)
        L.sub_short_factor[(c.tid - 16) / 3][(c.tid - 16) % 3] = mg->sub_short_factor[(c.tid - 16) / 3][(c.tid - 16) % 3];
    else if (c.tid < 30)
        L.uselongblock[c.tid - 28] = mg->uselong[c.tid - 28];
    else if (c.tid >= 32 && c.tid < 32 + n_chn_psy)
        lh_lds.ss.tot_ener[c.tid - 32] = mg->tot_ener[c.tid - 32];
    if (lane == 63 && w < n_chn_psy) {
        L.loudness_sq[gr][w] = lh_lds.ss.loudness_sq_save[w];
        lh_lds.ss.loudness_sq_save[w] = mg->loud[w];
    }
    LH_SYNC_WG_LDS();
#else
    LQ_MARK("ps_attack");
    /* (2) attack detection (reference psymodel.c:759-940) */
    {
        int const firbase = bufbase + 576 - 350 - LH_NSFIRLEN + 192;
        {
            /* A lane filters NINE CONSECUTIVE samples: their 22-sample windows overlap in all but one place, so the lane reads
             * 30 samples once (15 two-word reads, conflict-free at a stride of nine words) instead of 21 per output -- 189 --,
             * and every output's additions keep the reference's order. */
            int const i0 = 9 * lane;
            float   x[30];
#pragma unroll
            for (int t = 0; t < 30; t++)
                x[t] = lh_smp(c, w, firbase + i0 + t);
#pragma unroll
            for (int m = 0; m < 9; m++) {
                float   sum1 = x[m + 10], sum2 = 0.0;
#pragma unroll
                for (int j = 0; j < ((LH_NSFIRLEN - 1) / 2) - 1; j += 2) {
                    sum1 += lh_hp_fir[j] * (x[m + j] + x[m + LH_NSFIRLEN - j]);
                    sum2 += lh_hp_fir[j + 1] * (x[m + j + 1] + x[m + LH_NSFIRLEN - j - 1]);
                }
                P.a.hpf[w][i0 + m] = sum1 + sum2;
            }
        }
    }
    LH_SYNC_WG_LDS();
    for (int pass = 0; pass < 2; pass++) {
        int const chn = w + 2 * pass;
        if (chn < n_chn_psy) {
            /* Twelve sub-blocks of 64 samples: three from the previous call, nine new ones (reference
             * psymodel.c:806-925).  Lane i (< 12) owns sub-block i: its peak, the ratio against the one
             * two earlier, the sums over its short block.  The nine new peaks are wave maxima: eight come
             * out of one transposed reduction with peak k in lane k (and move up three lanes), the ninth
             * from a plain one. */
            uint32_t pk[8], pk8 = 0;
            float   e, then, ai;
            for (int k = 0; k < 9; k++) {
                int const i = lane + 64 * k;
                float   v;
                if (pass == 0)
                    v = P.a.hpf[w][i];
                else if (w == 0)
                    v = P.a.hpf[0][i] + P.a.hpf[1][i];
                else
                    v = P.a.hpf[0][i] - P.a.hpf[1][i];
                /* p = max(1, max |x|): exact under any evaluation order */
                if (k < 8)
                    pk[k] = lh_f32_as_u32(lh_fabsf(v));
                else
                    pk8 = lh_f32_as_u32(lh_fabsf(v));
            }
            {
                uint32_t const m8 = lh_wave_max8(pk);           /* lane k: peak k & 7 */
                uint32_t const m9 = lh_wave_max_u32(pk8);
                uint32_t const up = lh_lane_minus_u32 < 3 > (m8);       /* lane i: peak i - 3 (i = 3..10) */
                float const mine = lh_u32_as_f32((lane == 11) ? m9 : up);
                float const old = lh_lds.ss.last_en_subshort[chn][(lane < 3) ? lane + 6 : 0];
                float const older = lh_lds.ss.last_en_subshort[chn][(lane < 3) ? lane + 4 : 0];
                float const fresh = mine < 1.0f ? 1.0f : mine;
                e = (lane < 3) ? old : fresh;                   /* en_subshort[lane], lanes 0..11 */
                {
                    float const two_back = lh_u32_as_f32(lh_lane_minus_u32 < 2 > (lh_f32_as_u32(e)));  /* (every lane takes part) */
                    then = (lane < 3) ? older : two_back;
                }
            }
            LH_WAVE_SYNC_MEM();         /* the old values are read: the new ones may replace them */
            if (lane >= 3 && lane < 12)
                lh_lds.ss.last_en_subshort[chn][lane - 3] = e;
            /* a sub-block's peak against the one two sub-blocks earlier: a rise counts by its ratio, a
             * fall only beyond 10 x (reference psymodel.c:836-850); the three old ones: plain ratios */
            ai = (lane < 3) ? e / then : (e > then) ? e / then : (then > e * 10.0f) ? then / (e * 10.0f) : 0.0f;
            {
                /* sums over the short blocks (sub-blocks 3 g .. 3 g + 2, added in that order) arrive in
                 * lane 3 g + 2; a short block whose energy sits in its first sub-blocks gets a lower
                 * threshold (reference psymodel.c:853-864): halve once, or twice, when a later sub-block
                 * holds less than a sixth of the block */
                float const e1 = lh_u32_as_f32(lh_lane_minus_u32 < 1 > (lh_f32_as_u32(e)));
                float const e0 = lh_u32_as_f32(lh_lane_minus_u32 < 2 > (lh_f32_as_u32(e)));
                float const whole = e0 + e1 + e;
                int const tail_low = e * 6 < whole, mid_low = e1 * 6 < whole;
                float const x = T->attack_threshold[chn];
                uint64_t const over = lh_ballot(lane < 12 && ai > x);
                float const en_short0 = lh_u32_as_f32(lh_bcast_u32(lh_f32_as_u32(whole), 2));
                float const en_short1 = lh_u32_as_f32(lh_bcast_u32(lh_f32_as_u32(whole), 5));
                float const en_short2 = lh_u32_as_f32(lh_bcast_u32(lh_f32_as_u32(whole), 8));
                float const en_short3 = lh_u32_as_f32(lh_bcast_u32(lh_f32_as_u32(whole), 11));
                float const en_short[4] = { en_short0, en_short1, en_short2, en_short3 };
                int     nsa[4];
                int     ns_uselongblock = 1;
                int const last_att = lh_uni_i(lh_lds.ss.last_attacks[chn]);
                if (lane == 5 || lane == 8 || lane == 11)
                    L.sub_short_factor[chn][(lane - 5) / 3] = tail_low ? (mid_low ? 0.25f : 0.5f) : 1.0f;
                for (int gq = 0; gq < 4; gq++) {
                    /* the first sub-block of the short block whose ratio exceeds the threshold, 1-based */
                    unsigned const bits = (unsigned) (over >> (3 * gq)) & 7u;
                    nsa[gq] = bits ? ((bits & 1u) ? 1 : (bits & 2u) ? 2 : 3) : 0;
                }
                for (int i = 1; i < 4; i++) {
                    float const u = en_short[i - 1];
                    float const v = en_short[i];
                    float const m = (u > v) ? u : v;
                    if (m < 40000) {
                        if (u < 1.7f * v && v < 1.7f * u) {
                            if (i == 1 && nsa[0] <= nsa[i])
                                nsa[0] = 0;
                            nsa[i] = 0;
                        }
                    }
                }
                if (nsa[0] <= last_att)
                    nsa[0] = 0;
                if (last_att == 3 || nsa[0] + nsa[1] + nsa[2] + nsa[3]) {
                    ns_uselongblock = 0;
                    if (nsa[1] && nsa[0])
                        nsa[1] = 0;
                    if (nsa[2] && nsa[1])
                        nsa[2] = 0;
                    if (nsa[3] && nsa[2])
                        nsa[3] = 0;
                }
                if (lane == 0) {
                    for (int i = 0; i < 4; i++)
                        L.ns_attacks[chn][i] = (int8_t) nsa[i];
                    L.ns_uselong[chn] = ns_uselongblock;
                }
            }
        }
        LH_WAVE_SYNC_MEM();
    }
    LH_SYNC_WG_LDS();
    {
        /* uselongblock[] resolution (reference psymodel.c:926-933, 1265-1286) */
        int     ul0 = L.ns_uselong[0], ul1 = (cfg->channels == 2) ? L.ns_uselong[1] : 1;
        for (int chn = 2; chn < n_chn_psy; chn++)
            if (L.ns_uselong[chn] == 0)
                ul0 = ul1 = 0;
        if (cfg->short_blocks == 1 && !(ul0 && ul1))
            ul0 = ul1 = 0;
        if (cfg->short_blocks == 2)
            ul0 = ul1 = 1;
        if (cfg->short_blocks == 3)
            ul0 = ul1 = 0;
        LH_SYNC_WG_LDS();
        if (c.tid == 0) {
            L.uselongblock[0] = ul0;
            L.uselongblock[1] = ul1;
        }
    }
    LH_SYNC_WG_LDS();

    LH_PA(29, t_psy0);
    LQ_MARK("ps_fft");
    /* (3) long FFTs of L (wave 0) and R (wave 1) */
    lh_fft_long(c, w, bufbase, P.wsamp[w]);
    LH_SYNC_WG_LDS();
    LH_PA(30, t_psy0);
    LQ_MARK("ps_energy");
    /* (4) power spectra of this wave's two pseudo-channels */
    if (n_chn_psy == 4)
        lh_fft_energy_pair(c, w, P.wsamp[0], P.wsamp[1], LH_BLKSIZE, P.b.energy[w], P.b.energy[w + 2]);
    else if (w < n_chn_psy)
        lh_fft_energy(c, w, P.wsamp[0], P.wsamp[1], LH_BLKSIZE, P.b.energy[w]);
    /* The FHT buffers are free until the short FFTs: stage the long-block spreading matrix and
     * the tables of the masking addition there (the spreading loop of stage 6 makes three to
     * four dependent look-ups per step; from HBM each costs a few hundred cycles). */
    float  *stg_s3 = &P.wsamp[0][0];                    /* [LH_S3_MAX] */
    LH_SYNC_WG_LDS();
    for (int i = c.tid; i < T->psy_l.s3_count; i += LH_NT)
        stg_s3[i] = T->psy_l.s3[i];
    LH_SYNC_WG_LDS();
    LH_PA(31, t_psy0);
    LQ_MARK("ps_sums");
    /* (5) serial sums: total energy (bins 11..512) and loudness (reference psymodel.c:213-226,
     * 690-696): lane 0/1 = tot_ener of chn w / w+2, lane 2 = loudness of channel w */
    {
        /* Lane 0 / 1 add up the power spectrum of chn w / w + 2 in bin order, lane 2 the products
         * energy[w][j] * eql_w[j].  The products are independent of each other: all lanes form them
         * (256 at a time, into the partition arrays, which are idle until stage 6) so that the three
         * chains have nothing but LDS reads and their own additions on the critical path.  Terms
         * outside a sum's range are replaced by +0.0f, which leaves a non-negative sum unchanged. */
        int const chn = (lane == 1) ? w + 2 : w;
        int const summing = lane < 3 && chn < n_chn_psy;
        int const loud = (lane == 2);
        int const lo = loud ? 0 : 11;
        (void) lo;
        float  *prod = (w == 0) ? P.eb : P.thr;         /* [256] per wave */
        const float *ew = T->ath_eql_w;
        const float *e = P.b.energy[summing ? chn : w];
        float   acc = 0.0f;
        for (int h = 0; h < 2; h++) {
            int const j0 = 256 * h;
            LH_WAVE_SYNC_MEM();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int const j = j0 + lane + 64 * q;
                prod[j - j0] = P.b.energy[w][j] * ew[j];
            }
            LH_WAVE_SYNC_MEM();
            if (summing) {
                /* sixteen terms per trip; the next sixteen are read before this trip's additions, so the chain waits
                 * for its own additions only (a read issued when its terms are due costs the chain ~20 cycles a term) */
                const lh_f32x4 *s4 = (const lh_f32x4 *) ((loud ? prod - j0 : e) + j0);
                lh_f32x4 a0 = s4[0], a1 = s4[1], a2 = s4[2], a3 = s4[3];
                if (h == 0 && !loud) {
                    /* bins 0..10 are not part of the total energy */
                    a0.x = a0.y = a0.z = a0.w = 0.0f;
                    a1.x = a1.y = a1.z = a1.w = 0.0f;
                    a2.x = a2.y = a2.z = 0.0f;
                }
                for (int g = 0; g < 64; g += 4) {
                    int const n = (g + 4 < 64) ? g + 4 : g;
                    lh_f32x4 const b0 = s4[n], b1 = s4[n + 1], b2 = s4[n + 2], b3 = s4[n + 3];
#ifndef LH_EMU
                    /* (one statement, so that the four reads above stay ahead of the sixteen additions) */
                    asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\t"
                                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %0, %0, %7\n\tv_add_f32 %0, %0, %8\n\t"
                                 "v_add_f32 %0, %0, %9\n\tv_add_f32 %0, %0, %10\n\tv_add_f32 %0, %0, %11\n\tv_add_f32 %0, %0, %12\n\t"
                                 "v_add_f32 %0, %0, %13\n\tv_add_f32 %0, %0, %14\n\tv_add_f32 %0, %0, %15\n\tv_add_f32 %0, %0, %16"
                                 : "+v"(acc)
                                 : "v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w),
                                   "v"(a2.x), "v"(a2.y), "v"(a2.z), "v"(a2.w), "v"(a3.x), "v"(a3.y), "v"(a3.z), "v"(a3.w));
#else
                    acc += a0.x; acc += a0.y; acc += a0.z; acc += a0.w;
                    acc += a1.x; acc += a1.y; acc += a1.z; acc += a1.w;
                    acc += a2.x; acc += a2.y; acc += a2.z; acc += a2.w;
                    acc += a3.x; acc += a3.y; acc += a3.z; acc += a3.w;
#endif
                    a0 = b0;
                    a1 = b1;
                    a2 = b2;
                    a3 = b3;
                }
            }
        }
        if (summing) {
            if (!loud) {
                acc += e[LH_BLKSIZE / 2];
                lh_lds.ss.tot_ener[chn] = acc;
            }
            else {
                acc = (float) (acc * LH_VO_SCALE);
                L.loudness_sq[gr][w] = lh_lds.ss.loudness_sq_save[w];
                lh_lds.ss.loudness_sq_save[w] = acc;
            }
        }
        LH_WAVE_SYNC_MEM();
    }
    /* (the products a wave's loudness lane adds up lie in the partition arrays the OTHER wave's masking is about to write:
     * found when the same code ran at four waves per SIMD in lh_analysis.hip, where the two waves drift far enough apart) */
    LH_SYNC_WG_LDS();
#endif                          /* !LH_SPLIT */
    LH_PA(32, t_psy0);
    LQ_MARK("ps_mask");
    /* (6) masking thresholds, long blocks: the wave's one or two pseudo-channels together */
#ifdef LH_SPLIT
    if (n_chn_psy == 4) {
        LhTailChan const two[2] = {
            {w, &mlong->m[gr][w], &P.eb[w * 64], &P.thr[w * 64], &nb.n1[0], &nb.n2[0]},
            {w + 2, &mlong->m[gr][w + 2], &P.eb[(w + 2) * 64], &P.thr[(w + 2) * 64], &nb.n1[1], &nb.n2[1]}
        };
        lh_masking_tail < 2 > (c, 1, two);
    }
    else if (w < n_chn_psy) {
        LhTailChan const one[1] = { {w, &mlong->m[gr][w], &P.eb[w * 64], &P.thr[w * 64], &nb.n1[0], &nb.n2[0]} };
        lh_masking_tail < 1 > (c, 1, one);
    }
#else
    if (n_chn_psy == 4) {
        LhMaskChan const two[2] = {
            {w, P.b.energy[w], &P.eb[w * 64], &P.thr[w * 64], &nb.n1[0], &nb.n2[0]},
            {w + 2, P.b.energy[w + 2], &P.eb[(w + 2) * 64], &P.thr[(w + 2) * 64], &nb.n1[1], &nb.n2[1]}
        };
        lh_compute_masking < 2 > (c, 1, two, stg_s3);
    }
    else if (w < n_chn_psy) {
        LhMaskChan const one[1] = { {w, P.b.energy[w], &P.eb[w * 64], &P.thr[w * 64], &nb.n1[0], &nb.n2[0]} };
        lh_compute_masking < 1 > (c, 1, one, stg_s3);
    }
#endif
    LH_SYNC_WG_LDS();
    if (cfg->mode == LH_MODE_JOINT_STEREO && (L.uselongblock[0] + L.uselongblock[1]) == 2) {
        float const ath_factor =
            (cfg->msfix > 0.f) ? (cfg->ATH_offset_factor * lh_lds.ss.ath_adjust_factor) : 1.f;
        if (w == 0)
            lh_ms_thresholds(c, P.eb, P.thr, T->psy_l.mld_cb, T->ath_cb_l, ath_factor, cfg->msfix,
                             T->psy_l.npart);
    }
    LH_SYNC_WG_LDS();
    LH_PA(33, t_psy0);
    LQ_MARK("ps_p2sfb");
    /* (7) partitions -> scalefactor bands, long and long->short estimates
     * (reference psymodel.c:411-439): both tables and the wave's pseudo-channels in one pass */
    if (n_chn_psy == 4) {
        LhSfbChan const two[2] = {
            {&P.eb[w * 64], &P.thr[w * 64], &L.psy_en[now][w][0], &L.psy_thm[now][w][0]},
            {&P.eb[(w + 2) * 64], &P.thr[(w + 2) * 64], &L.psy_en[now][w + 2][0], &L.psy_thm[now][w + 2][0]}
        };
        lh_partition2sfb_long < 2 > (T, two, lane);
    }
    else if (w < n_chn_psy) {
        LhSfbChan const one[1] = { {&P.eb[w * 64], &P.thr[w * 64], &L.psy_en[now][w][0], &L.psy_thm[now][w][0]} };
        lh_partition2sfb_long < 1 > (T, one, lane);
    }
    LH_SYNC_WG_LDS();
    LH_PA(34, t_psy0);
    LQ_MARK("ps_short");
    /* (8) short blocks (reference psymodel.c:1470-1500) */
    /* (nothing of it runs when both channels keep long blocks -- the usual granule: the values the
     * short transforms would replace are the long->short estimates of stage 7) */
    int const any_short = lh_uni_i(!(L.uselongblock[0] && L.uselongblock[1]));
#ifndef LH_SPLIT
    if (any_short) {
        if (!L.uselongblock[w])
            lh_fft_short(c, w, bufbase, &P.wsamp[w][0]);
        LH_SYNC_WG_LDS();
    }
#endif
    for (int sblock = 0; any_short && sblock < 3; sblock++) {
        if (w < n_chn_psy && !L.uselongblock[w]) {
            /* (wave w's pseudo-channels w and w + 2 share uselongblock[w]) */
#ifdef LH_SPLIT
            if (n_chn_psy == 4) {
                LhTailChan const two[2] = {
                    {w, &mshort->m[gr][sblock][w], &P.eb[w * 64], &P.thr[w * 64], &nb.n1[0], &nb.n2[0]},
                    {w + 2, &mshort->m[gr][sblock][w + 2], &P.eb[(w + 2) * 64], &P.thr[(w + 2) * 64], &nb.n1[1], &nb.n2[1]}
                };
                lh_masking_tail < 2 > (c, 0, two);
            }
            else {
                LhTailChan const one[1] = { {w, &mshort->m[gr][sblock][w], &P.eb[w * 64], &P.thr[w * 64], &nb.n1[0], &nb.n2[0]} };
                lh_masking_tail < 1 > (c, 0, one);
            }
        }
#else
            int const both = (n_chn_psy == 4);
            if (both)
                lh_fft_energy_pair(c, w, &P.wsamp[0][sblock * LH_BLKSIZE_S], &P.wsamp[1][sblock * LH_BLKSIZE_S],
                                   LH_BLKSIZE_S, P.b.energy[w], P.b.energy[w + 2]);
            else
                lh_fft_energy(c, w, &P.wsamp[0][sblock * LH_BLKSIZE_S], &P.wsamp[1][sblock * LH_BLKSIZE_S], LH_BLKSIZE_S,
                              P.b.energy[w]);
            LH_WAVE_SYNC_MEM();
            if (both) {
                LhMaskChan const two[2] = {
                    {w, P.b.energy[w], &P.eb[w * 64], &P.thr[w * 64], &nb.n1[0], &nb.n2[0]},
                    {w + 2, P.b.energy[w + 2], &P.eb[(w + 2) * 64], &P.thr[(w + 2) * 64], &nb.n1[1], &nb.n2[1]}
                };
                lh_compute_masking < 2 > (c, 0, two, T->psy_s.s3);    /* short: nb left alone */
            }
            else {
                LhMaskChan const one[1] = { {w, P.b.energy[w], &P.eb[w * 64], &P.thr[w * 64], &nb.n1[0], &nb.n2[0]} };
                lh_compute_masking < 1 > (c, 0, one, T->psy_s.s3);
            }
        }
#endif
        LH_SYNC_WG_LDS();
        if (cfg->mode == LH_MODE_JOINT_STEREO && (L.uselongblock[0] + L.uselongblock[1]) == 0) {
            float const ath_factor =
                (cfg->msfix > 0.f) ? (cfg->ATH_offset_factor * lh_lds.ss.ath_adjust_factor) : 1.f;
            if (w == 0)
                lh_ms_thresholds(c, P.eb, P.thr, T->psy_s.mld_cb, T->ath_cb_s, ath_factor,
                                 cfg->msfix, T->psy_s.npart);
        }
        LH_SYNC_WG_LDS();
        for (int t = 0; t < 2; t++) {
            int const chn = w + 2 * t;
            int const act = chn < n_chn_psy && !L.uselongblock[chn & 1];
            int const cc = (chn < n_chn_psy) ? chn : w;
            lh_partition2sfb_wave(&T->psy_s, &P.eb[cc * 64], &P.thr[cc * 64], &L.psy_en[now][cc][22 + sblock],
                                  &L.psy_thm[now][cc][22 + sblock], 3, -1.0f, 0, lane, act);
        }
        LH_SYNC_WG_LDS();
    }
    LH_PA(35, t_psy0);
    LQ_MARK("ps_preecho");
    /* (9) short block pre-echo control (reference psymodel.c:1502-1553): one lane per (chn, sb) */
    {
        float const pcfact = 0.6f;
        for (int t = c.tid; t < n_chn_psy * LH_SBMAX_S; t += LH_NT) {
            int const chn = t / LH_SBMAX_S, sb = t - chn * LH_SBMAX_S;
            const float *last_thm = &L.psy_thm[was][chn][22 + sb * 3];
            float   new_thmm[3], prev_thm, t1, t2, thmm;
            int const last_att = lh_lds.ss.last_attacks[chn];
            for (int sblock = 0; sblock < 3; sblock++) {
                int const a0 = L.ns_attacks[chn][sblock], a1 = L.ns_attacks[chn][sblock + 1];
                thmm = L.psy_thm[now][chn][22 + sb * 3 + sblock];
                thmm = (float) (thmm * LH_PREECHO_ATT0);
                t1 = t2 = thmm;
                if (sblock > 0)
                    prev_thm = new_thmm[sblock - 1];
                else
                    prev_thm = last_thm[2];
                if (a0 >= 2 || a1 == 1)
                    t1 = lh_ns_interp(prev_thm, thmm, (float) (LH_PREECHO_ATT1 * pcfact));
                thmm = (t1 < thmm) ? t1 : thmm;
                if (a0 == 1)
                    t2 = lh_ns_interp(prev_thm, thmm, (float) (LH_PREECHO_ATT2 * pcfact));
                else if ((sblock == 0 && last_att == 3)
                         || (sblock > 0 && L.ns_attacks[chn][sblock - 1] == 3)) {
                    switch (sblock) {
                    case 0:
                        prev_thm = last_thm[1];
                        break;
                    case 1:
                        prev_thm = last_thm[2];
                        break;
                    default:
                        prev_thm = new_thmm[0];
                        break;
                    }
                    t2 = lh_ns_interp(prev_thm, thmm, (float) (LH_PREECHO_ATT2 * pcfact));
                }
                thmm = (t1 < thmm) ? t1 : thmm;
                thmm = (t2 < thmm) ? t2 : thmm;
                thmm *= L.sub_short_factor[chn][sblock];
                new_thmm[sblock] = thmm;
            }
            for (int sblock = 0; sblock < 3; sblock++)
                L.psy_thm[now][chn][22 + sb * 3 + sblock] = new_thmm[sblock];
        }
    }
    LH_SYNC_WG_LDS();
    LH_PA(36, t_psy0);
    LQ_MARK("ps_pe");
    /* (10) block type state machine (reference psymodel.c:1289-1319) + PE (:1568-1595) */
    {
        int     btd[2];
        for (int chn = 0; chn < 2; chn++) {
            int     blocktype = LH_NORM_TYPE;
            int     old = lh_lds.ss.blocktype_old[chn];
            if (L.uselongblock[chn]) {
                if (old == LH_SHORT_TYPE)
                    blocktype = LH_STOP_TYPE;
            }
            else {
                blocktype = LH_SHORT_TYPE;
                if (old == LH_NORM_TYPE)
                    old = LH_START_TYPE;
                if (old == LH_STOP_TYPE)
                    old = LH_SHORT_TYPE;
            }
            btd[chn] = old;
            L.next_blocktype[chn] = blocktype;
        }
        LH_SYNC_WG_LDS();
        if (c.tid < 2) {
            lh_lds.ss.blocktype_old[c.tid] = L.next_blocktype[c.tid];
            L.block_type[gr][c.tid] = btd[c.tid];
        }
        /* Perceptual entropy (reference psymodel.c:458-553): the terms coef * log10(en / thr)
         * are independent of each other -- one lane per term, in double as the reference forms
         * them -- and only the accumulation pe = (float) (pe + term) runs in band order (a band
         * that contributes nothing adds 0.0, which leaves pe unchanged).  Wave w: channels w, w+2. */
#ifdef LH_SPLIT
        int const logt_lds = lh_uni_i(!(cfg->vbr == 1 || cfg->vbr == 4));
#endif
        for (int pass = 0; pass < 2; pass++) {
            int const chn = w + 2 * pass;
            int     type, is_short, nterms;
            if (chn >= n_chn_psy)
                continue;
            if (chn > 1) {
                type = LH_NORM_TYPE;
                if (btd[0] == LH_SHORT_TYPE || btd[1] == LH_SHORT_TYPE)
                    type = LH_SHORT_TYPE;
            }
            else
                type = btd[chn];
            is_short = (type == LH_SHORT_TYPE);
            nterms = is_short ? 3 * (LH_SBMAX_S - 1) : LH_SBMAX_L - 1;
            LH_WAVE_SYNC_MEM();
            {
                /* The terms stay in their lanes' registers (0.0 beyond the last one) and the accumulation -- a chain of
                 * float <- double additions in band order, the same in every lane -- takes term i from lane i with two
                 * v_readlane: as one lane walking an LDS array the chain paid a round trip per term (21 or 36 of them, twice
                 * per granule and wave).  Adding 0.0 leaves pe unchanged, so the long case may stop at 21. */
                double  term = 0.0;

                if (lane < nterms) {
                    int const idx = is_short ? 22 + lane : lane;
                    float const coef = is_short ? lh_regcoef_s[lane / 3] : lh_regcoef_l[lane];
                    float const t = L.psy_thm[was][chn][idx];
                    if (t > 0.0f) {
                        float const x = t * lh_lds.ss.masking_lower;
                        float const e = L.psy_en[was][chn][idx];
                        if (e > x) {
                            if (e > x * 1e10f)
                                term = coef * (10.0f * 2.30258509299404568402);
                            else {
#ifdef LH_SPLIT
                                /* (the split pipeline's encode kernel keeps calc_noise's copy of the table in LDS for the
                                 * whole launch -- except in the new VBR loop, whose step tables lie there) */
                                float   l2;
                                if (logt_lds) {
                                    LH_FAST_LOG2_VIA(LH_LOGT_LDS, e / x, l2);
                                }
                                else
                                    l2 = lh_fast_log2(T->log_table, e / x);
                                term = coef * (l2 * LH_LOG2_OVER_LOG10);
#else
                                term = coef * (lh_fast_log2(T->log_table, e / x) * LH_LOG2_OVER_LOG10);
#endif
                            }
                        }
                    }
                }
                uint64_t const tb = lh_f64_as_u64(term);
                uint32_t const tlo = (uint32_t) tb, thi = (uint32_t) (tb >> 32);
                float   pe = is_short ? 1236.28f / 4 : 1124.23f / 4;
#pragma unroll
                for (int i = 0; i < LH_SBMAX_L - 1; i++)
                    pe = (float) (pe + lh_u64_as_f64((uint64_t) lh_bcast_u32(tlo, i) | ((uint64_t) lh_bcast_u32(thi, i) << 32)));
                if (is_short) {
#pragma unroll
                    for (int i = LH_SBMAX_L - 1; i < 3 * (LH_SBMAX_S - 1); i++)
                        pe = (float) (pe + lh_u64_as_f64((uint64_t) lh_bcast_u32(tlo, i) | ((uint64_t) lh_bcast_u32(thi, i) << 32)));
                }
                if (lane == 0) {
                    L.pe[gr][chn] = pe;
                    lh_lds.ss.last_attacks[chn] = L.ns_attacks[chn][2];
                }
            }
        }
    }
    LH_SYNC_WG_LDS();
    return nb;
}

#endif
