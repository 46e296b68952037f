/*
 * orc_psy.c -- CPU restatement of the psycho-acoustic stage
 * (reference libmp3lame/psymodel.c:213-1597, fft.c:63-289, util.c:976-1001).
 * TEST INFRASTRUCTURE ONLY -- see orc_common.h.
 */
#include "orc_common.h"

#define NSFIRLEN 21
#define RPELEV  2
#define RPELEV2 16
#define NS_PREECHO_ATT0 0.8
#define NS_PREECHO_ATT1 0.6
#define NS_PREECHO_ATT2 0.3
#define VO_SCALE (1./( 14752*14752 )/(LH_BLKSIZE/2))

/* reference util.c:976-1001 */
float
orc_fast_log2(const LhTables * t, float x)
{
    float   log2val, partial;
    union {
        float   f;
        int     i;
    } fi;
    int     mantisse;
    fi.f = x;
    mantisse = fi.i & 0x7fffff;
    log2val = ((fi.i >> 23) & 0xFF) - 0x7f;
    partial = (mantisse & ((1 << (23 - 9)) - 1));
    partial *= 1.0f / ((1 << (23 - 9)));
    mantisse >>= (23 - 9);
    log2val += t->log_table[mantisse] * (1.0f - partial) + t->log_table[mantisse + 1] * partial;
    return log2val;
}

#define FAST_LOG10(t,x)      (orc_fast_log2(t,x)*(ORC_LOG2/ORC_LOG10))
#define FAST_LOG10_X(t,x,y)  (orc_fast_log2(t,x)*(ORC_LOG2/ORC_LOG10*(y)))

/* ---------------------------------------------------------------------- */
/* FHT with the twiddle recurrence tabulated (reference fft.c:63-148).     */
/* Butterflies of one pass are independent, so any visiting order gives    */
/* the reference's bits.                                                    */
static void
orc_fht(const LhTables * t, float *fz, int n /* full length: 1024 or 256 */ )
{
    int     k4 = 4, stage = 0;
    float  *fi, *gi;
    float const *fn = fz + n;
    do {
        int     i, k1, k2, k3, kx;
        kx = k4 >> 1;
        k1 = k4;
        k2 = k4 << 1;
        k3 = k2 + k1;
        k4 = k2 << 1;
        fi = fz;
        gi = fi + kx;
        do {
            float   f0, f1, f2, f3;
            f1 = fi[0] - fi[k1];
            f0 = fi[0] + fi[k1];
            f3 = fi[k2] - fi[k3];
            f2 = fi[k2] + fi[k3];
            fi[k2] = f0 - f2;
            fi[0] = f0 + f2;
            fi[k3] = f1 - f3;
            fi[k1] = f1 + f3;
            f1 = gi[0] - gi[k1];
            f0 = gi[0] + gi[k1];
            f3 = ORC_SQRT2 * gi[k3];
            f2 = ORC_SQRT2 * gi[k2];
            gi[k2] = f0 - f2;
            gi[0] = f0 + f2;
            gi[k3] = f1 - f3;
            gi[k1] = f1 + f3;
            gi += k4;
            fi += k4;
        } while (fi < fn);
        for (i = 1; i < kx; i++) {
            float const c1 = t->fht_tw[stage][i][0], s1 = t->fht_tw[stage][i][1];
            float const c2 = t->fht_tw[stage][i][2], s2 = t->fht_tw[stage][i][3];
            fi = fz + i;
            gi = fz + k1 - i;
            do {
                float   a, b, g0, f0, f1, g1, f2, g2, f3, g3;
                b = s2 * fi[k1] - c2 * gi[k1];
                a = c2 * fi[k1] + s2 * gi[k1];
                f1 = fi[0] - a;
                f0 = fi[0] + a;
                g1 = gi[0] - b;
                g0 = gi[0] + b;
                b = s2 * fi[k3] - c2 * gi[k3];
                a = c2 * fi[k3] + s2 * gi[k3];
                f3 = fi[k2] - a;
                f2 = fi[k2] + a;
                g3 = gi[k2] - b;
                g2 = gi[k2] + b;
                b = s1 * f2 - c1 * g3;
                a = c1 * f2 + s1 * g3;
                fi[k2] = f0 - a;
                fi[0] = f0 + a;
                gi[k3] = g1 - b;
                gi[k1] = g1 + b;
                b = c1 * g2 - s1 * f3;
                a = s1 * g2 + c1 * f3;
                gi[k2] = g0 - a;
                gi[0] = g0 + a;
                fi[k3] = f1 - b;
                fi[k1] = f1 + b;
                gi += k4;
                fi += k4;
            } while (fi < fn);
        }
        stage++;
    } while (k4 < n);
}

static unsigned
rev8(unsigned v)
{
    unsigned r = 0;
    int     i;
    for (i = 0; i < 8; i++)
        if (v & (1u << i))
            r |= 0x80u >> i;
    return r;
}

/* reference fft.c:245-289 */
void
orc_fft_long(const LhTables * t, float x[LH_BLKSIZE], int chn, const float *const buffer[2])
{
    const float *w = t->fft_window;
    const float *b = buffer[chn];
    int     jj;
    for (jj = LH_BLKSIZE / 8 - 1; jj >= 0; --jj) {
        float   f0, f1, f2, f3, ww;
        float  *o = x + 4 * jj;
        int     i = (int) rev8((unsigned) jj);
        f0 = w[i] * b[i];
        ww = w[i + 0x200] * b[i + 0x200];
        f1 = f0 - ww;
        f0 = f0 + ww;
        f2 = w[i + 0x100] * b[i + 0x100];
        ww = w[i + 0x300] * b[i + 0x300];
        f3 = f2 - ww;
        f2 = f2 + ww;
        o[0] = f0 + f2;
        o[2] = f0 - f2;
        o[1] = f1 + f3;
        o[3] = f1 - f3;
        f0 = w[i + 0x001] * b[i + 0x001];
        ww = w[i + 0x201] * b[i + 0x201];
        f1 = f0 - ww;
        f0 = f0 + ww;
        f2 = w[i + 0x101] * b[i + 0x101];
        ww = w[i + 0x301] * b[i + 0x301];
        f3 = f2 - ww;
        f2 = f2 + ww;
        o[LH_BLKSIZE / 2 + 0] = f0 + f2;
        o[LH_BLKSIZE / 2 + 2] = f0 - f2;
        o[LH_BLKSIZE / 2 + 1] = f1 + f3;
        o[LH_BLKSIZE / 2 + 3] = f1 - f3;
    }
    orc_fht(t, x, LH_BLKSIZE);
}

/* reference fft.c:193-243 */
void
orc_fft_short(const LhTables * t, float x_real[3][LH_BLKSIZE_S], int chn,
              const float *const buffer[2])
{
    const float *ws = t->fft_window_s;
    const float *bf = buffer[chn];
    int     b, j;
    for (b = 0; b < 3; b++) {
        int const k = (576 / 3) * (b + 1);
        for (j = LH_BLKSIZE_S / 8 - 1; j >= 0; --j) {
            float   f0, f1, f2, f3, w;
            float  *o = &x_real[b][4 * j];
            int     i = (int) rev8((unsigned) (j << 2));
            f0 = ws[i] * bf[i + k];
            w = ws[0x7f - i] * bf[i + k + 0x80];
            f1 = f0 - w;
            f0 = f0 + w;
            f2 = ws[i + 0x40] * bf[i + k + 0x40];
            w = ws[0x3f - i] * bf[i + k + 0xc0];
            f3 = f2 - w;
            f2 = f2 + w;
            o[0] = f0 + f2;
            o[2] = f0 - f2;
            o[1] = f1 + f3;
            o[3] = f1 - f3;
            f0 = ws[i + 0x01] * bf[i + k + 0x01];
            w = ws[0x7e - i] * bf[i + k + 0x81];
            f1 = f0 - w;
            f0 = f0 + w;
            f2 = ws[i + 0x41] * bf[i + k + 0x41];
            w = ws[0x3e - i] * bf[i + k + 0xc1];
            f3 = f2 - w;
            f2 = f2 + w;
            o[LH_BLKSIZE_S / 2 + 0] = f0 + f2;
            o[LH_BLKSIZE_S / 2 + 2] = f0 - f2;
            o[LH_BLKSIZE_S / 2 + 1] = f1 + f3;
            o[LH_BLKSIZE_S / 2 + 3] = f1 - f3;
        }
        orc_fht(t, x_real[b], LH_BLKSIZE_S);
    }
}

/* ---------------------------------------------------------------------- */
static const float tab[9] = {
    1.0, 0.79433, 0.63096, 0.63096, 0.63096, 0.63096, 0.63096, 0.25119, 0.11749
};
static const int tab_mask_add_delta[9] = { 2, 2, 2, 1, 1, 1, 0, 0, -1 };

/* reference psymodel.c:294-341 */
static float
mask_add(const LhTables * t, float m1, float m2, int b, int delta)
{
    static const float table2[] = {
        1.33352 * 1.33352, 1.35879 * 1.35879, 1.38454 * 1.38454, 1.39497 * 1.39497,
        1.40548 * 1.40548, 1.3537 * 1.3537, 1.30382 * 1.30382, 1.22321 * 1.22321,
        1.14758 * 1.14758,
        1
    };
    float   ratio;
    if (m1 < 0)
        m1 = 0;
    if (m2 < 0)
        m2 = 0;
    if (m1 <= 0)
        return m2;
    if (m2 <= 0)
        return m1;
    if (m2 > m1)
        ratio = m2 / m1;
    else
        ratio = m1 / m2;
    if (b < 0)
        b = -b;
    if (b <= delta) {
        if (ratio >= t->ma_max_i1)
            return m1 + m2;
        else {
            int     i = (int) (FAST_LOG10_X(t, ratio, 16.0f));
            return (m1 + m2) * table2[i];
        }
    }
    if (ratio < t->ma_max_i2)
        return m1 + m2;
    if (m1 < m2)
        m1 = m2;
    return m1;
}

/* reference psymodel.c:350-393 */
static void
convert_partition2scalefac(LhPsyBand const *gd, float const *eb, float const *thr,
                           float enn_out[], float thm_out[])
{
    float   enn, thmm;
    int     sb, b, n = gd->n_sb;
    enn = thmm = 0.0f;
    for (sb = b = 0; sb < n; ++b, ++sb) {
        int const bo_sb = gd->bo[sb];
        int const npart = gd->npart;
        int const b_lim = bo_sb < npart ? bo_sb : npart;
        while (b < b_lim) {
            enn += eb[b];
            thmm += thr[b];
            b++;
        }
        if (b >= npart) {
            enn_out[sb] = enn;
            thm_out[sb] = thmm;
            ++sb;
            break;
        }
        {
            float const w_curr = gd->bo_weight[sb];
            float const w_next = 1.0f - w_curr;
            enn += w_curr * eb[b];
            thmm += w_curr * thr[b];
            enn_out[sb] = enn;
            thm_out[sb] = thmm;
            enn = w_next * eb[b];
            thmm = w_next * thr[b];
        }
    }
    for (; sb < n; ++sb) {
        enn_out[sb] = 0;
        thm_out[sb] = 0;
    }
}

/* reference psymodel.c:443-454 */
static float
ns_interp(float x, float y, float r)
{
    if (r >= 1.0f)
        return x;
    if (r <= 0.0f)
        return y;
    if (y > 0.0f)
        return powf(x / y, r) * y;
    return 0.0f;
}

/* reference psymodel.c:458-501 */
static float
pecalc_s(const LhTables * t, OrcRatio const *mr, float masking_lower)
{
    float   pe_s;
    static const float regcoef_s[] = {
        11.8, 13.6, 17.2, 32, 46.5, 51.3, 57.5, 67.1, 71.5, 84.6, 97.6, 130,
    };
    unsigned int sb, sblock;
    pe_s = 1236.28f / 4;
    for (sb = 0; sb < LH_SBMAX_S - 1; sb++) {
        for (sblock = 0; sblock < 3; sblock++) {
            float const thm = mr->thm.s[sb][sblock];
            if (thm > 0.0f) {
                float const x = thm * masking_lower;
                float const en = mr->en.s[sb][sblock];
                if (en > x) {
                    if (en > x * 1e10f)
                        pe_s += regcoef_s[sb] * (10.0f * ORC_LOG10);
                    else
                        pe_s += regcoef_s[sb] * FAST_LOG10(t, en / x);
                }
            }
        }
    }
    return pe_s;
}

/* reference psymodel.c:503-553 */
static float
pecalc_l(const LhTables * t, OrcRatio const *mr, float masking_lower)
{
    float   pe_l;
    static const float regcoef_l[] = {
        6.8, 5.8, 5.8, 6.4, 6.5, 9.9, 12.1, 14.4, 15, 18.9, 21.6, 26.9, 34.2, 40.2, 46.8, 56.5,
        60.7, 73.9, 85.7, 93.4, 126.1,
    };
    unsigned int sb;
    pe_l = 1124.23f / 4;
    for (sb = 0; sb < LH_SBMAX_L - 1; sb++) {
        float const thm = mr->thm.l[sb];
        if (thm > 0.0f) {
            float const x = thm * masking_lower;
            float const en = mr->en.l[sb];
            if (en > x) {
                if (en > x * 1e10f)
                    pe_l += regcoef_l[sb] * (10.0f * ORC_LOG10);
                else
                    pe_l += regcoef_l[sb] * FAST_LOG10(t, en / x);
            }
        }
    }
    return pe_l;
}

/* tonality index from the 3-partition peak/average (reference psymodel.c:583-652, 958-1028;
 * the long and short variants are the same arithmetic on different tables) */
static void
calc_mask_index(LhPsyBand const *gd, float const *max, float const *avg, unsigned char *mask_idx)
{
    float   m, a;
    int     b, k;
    int const last_tab_entry = 8;
    b = 0;
    a = avg[b] + avg[b + 1];
    if (a > 0.0f) {
        m = max[b];
        if (m < max[b + 1])
            m = max[b + 1];
        a = 20.0f * (m * 2.0f - a) / (a * (gd->numlines[b] + gd->numlines[b + 1] - 1));
        k = (int) a;
        if (k > last_tab_entry)
            k = last_tab_entry;
        mask_idx[b] = k;
    }
    else
        mask_idx[b] = 0;
    for (b = 1; b < gd->npart - 1; b++) {
        a = avg[b - 1] + avg[b] + avg[b + 1];
        if (a > 0.0f) {
            m = max[b - 1];
            if (m < max[b])
                m = max[b];
            if (m < max[b + 1])
                m = max[b + 1];
            a = 20.0f * (m * 3.0f - a)
                / (a * (gd->numlines[b - 1] + gd->numlines[b] + gd->numlines[b + 1] - 1));
            k = (int) a;
            if (k > last_tab_entry)
                k = last_tab_entry;
            mask_idx[b] = k;
        }
        else
            mask_idx[b] = 0;
    }
    a = avg[b - 1] + avg[b];
    if (a > 0.0f) {
        m = max[b - 1];
        if (m < max[b])
            m = max[b];
        a = 20.0f * (m * 2.0f - a) / (a * (gd->numlines[b - 1] + gd->numlines[b] - 1));
        k = (int) a;
        if (k > last_tab_entry)
            k = last_tab_entry;
        mask_idx[b] = k;
    }
    else
        mask_idx[b] = 0;
}

/* reference psymodel.c:655-704 */
static void
compute_fft_l(OrcStream * S, const float *const buffer[2], int chn, float fftenergy[LH_HBLKSIZE],
              float (*wsamp_l)[LH_BLKSIZE])
{
    int     j;
    if (chn < 2)
        orc_fft_long(S->tab, *wsamp_l, chn, buffer);
    else if (chn == 2) {
        float const sqrt2_half = ORC_SQRT2 * 0.5f;
        for (j = LH_BLKSIZE - 1; j >= 0; --j) {
            float const l = wsamp_l[0][j];
            float const r = wsamp_l[1][j];
            wsamp_l[0][j] = (l + r) * sqrt2_half;
            wsamp_l[1][j] = (l - r) * sqrt2_half;
        }
    }
    fftenergy[0] = wsamp_l[0][0];
    fftenergy[0] *= fftenergy[0];
    for (j = LH_BLKSIZE / 2 - 1; j >= 0; --j) {
        float const re = (*wsamp_l)[LH_BLKSIZE / 2 - j];
        float const im = (*wsamp_l)[LH_BLKSIZE / 2 + j];
        fftenergy[LH_BLKSIZE / 2 - j] = (re * re + im * im) * 0.5f;
    }
    {
        float   totalenergy = 0.0f;
        for (j = 11; j < LH_HBLKSIZE; j++)
            totalenergy += fftenergy[j];
        S->tot_ener[chn] = totalenergy;
    }
}

/* reference psymodel.c:707-737 */
static void
compute_fft_s(OrcStream * S, const float *const buffer[2], int chn, int sblock,
              float (*fftenergy_s)[LH_HBLKSIZE_S], float (*wsamp_s)[3][LH_BLKSIZE_S])
{
    int     j;
    if (sblock == 0 && chn < 2)
        orc_fft_short(S->tab, *wsamp_s, chn, buffer);
    if (chn == 2) {
        float const sqrt2_half = ORC_SQRT2 * 0.5f;
        for (j = LH_BLKSIZE_S - 1; j >= 0; --j) {
            float const l = wsamp_s[0][sblock][j];
            float const r = wsamp_s[1][sblock][j];
            wsamp_s[0][sblock][j] = (l + r) * sqrt2_half;
            wsamp_s[1][sblock][j] = (l - r) * sqrt2_half;
        }
    }
    fftenergy_s[sblock][0] = (*wsamp_s)[sblock][0];
    fftenergy_s[sblock][0] *= fftenergy_s[sblock][0];
    for (j = LH_BLKSIZE_S / 2 - 1; j >= 0; --j) {
        float const re = (*wsamp_s)[sblock][LH_BLKSIZE_S / 2 - j];
        float const im = (*wsamp_s)[sblock][LH_BLKSIZE_S / 2 + j];
        fftenergy_s[sblock][LH_BLKSIZE_S / 2 - j] = (re * re + im * im) * 0.5f;
    }
}

/* reference psymodel.c:759-940 */
static void
attack_detection(OrcStream * S, const float *const buffer[2], int gr_out,
                 OrcRatio masking_ratio[2][2], OrcRatio masking_MS_ratio[2][2], float energy[4],
                 float sub_short_factor[4][3], int ns_attacks[4][4], int uselongblock[2])
{
    float   ns_hpfsmpl[2][576];
    int const n_chn_out = S->cfg->channels;
    int const n_chn_psy = (S->cfg->mode == LH_MODE_JOINT_STEREO) ? 4 : n_chn_out;
    int     chn, i, j;

    memset(&ns_hpfsmpl[0][0], 0, sizeof(ns_hpfsmpl));
    for (chn = 0; chn < n_chn_out; chn++) {
        static const float fircoef[] = {
            -8.65163e-18 * 2, -0.00851586 * 2, -6.74764e-18 * 2, 0.0209036 * 2,
            -3.36639e-17 * 2, -0.0438162 * 2, -1.54175e-17 * 2, 0.0931738 * 2,
            -5.52212e-17 * 2, -0.313819 * 2
        };
        const float *const firbuf = &buffer[chn][576 - 350 - NSFIRLEN + 192];
        for (i = 0; i < 576; i++) {
            float   sum1, sum2;
            sum1 = firbuf[i + 10];
            sum2 = 0.0;
            for (j = 0; j < ((NSFIRLEN - 1) / 2) - 1; j += 2) {
                sum1 += fircoef[j] * (firbuf[i + j] + firbuf[i + NSFIRLEN - j]);
                sum2 += fircoef[j + 1] * (firbuf[i + j + 1] + firbuf[i + NSFIRLEN - j - 1]);
            }
            ns_hpfsmpl[chn][i] = sum1 + sum2;
        }
        masking_ratio[gr_out][chn].en = S->en[chn];
        masking_ratio[gr_out][chn].thm = S->thm[chn];
        if (n_chn_psy > 2) {
            masking_MS_ratio[gr_out][chn].en = S->en[chn + 2];
            masking_MS_ratio[gr_out][chn].thm = S->thm[chn + 2];
        }
    }
    for (chn = 0; chn < n_chn_psy; chn++) {
        float   attack_intensity[12];
        float   en_subshort[12];
        float   en_short[4] = { 0, 0, 0, 0 };
        float const *pf = ns_hpfsmpl[chn & 1];
        int     ns_uselongblock = 1;

        if (chn == 2) {
            for (i = 0, j = 576; j > 0; ++i, --j) {
                float const l = ns_hpfsmpl[0][i];
                float const r = ns_hpfsmpl[1][i];
                ns_hpfsmpl[0][i] = l + r;
                ns_hpfsmpl[1][i] = l - r;
            }
        }
        for (i = 0; i < 3; i++) {
            en_subshort[i] = S->last_en_subshort[chn][i + 6];
            attack_intensity[i] = en_subshort[i] / S->last_en_subshort[chn][i + 4];
            en_short[0] += en_subshort[i];
        }
        for (i = 0; i < 9; i++) {
            float const *const pfe = pf + 576 / 9;
            float   p = 1.;
            for (; pf < pfe; pf++)
                if (p < fabs(*pf))
                    p = fabs(*pf);
            S->last_en_subshort[chn][i] = en_subshort[i + 3] = p;
            en_short[1 + i / 3] += p;
            if (p > en_subshort[i + 3 - 2])
                p = p / en_subshort[i + 3 - 2];
            else if (en_subshort[i + 3 - 2] > p * 10.0f)
                p = en_subshort[i + 3 - 2] / (p * 10.0f);
            else
                p = 0.0;
            attack_intensity[i + 3] = p;
        }
        for (i = 0; i < 3; ++i) {
            float const enn =
                en_subshort[i * 3 + 3] + en_subshort[i * 3 + 4] + en_subshort[i * 3 + 5];
            float   factor = 1.f;
            if (en_subshort[i * 3 + 5] * 6 < enn) {
                factor *= 0.5f;
                if (en_subshort[i * 3 + 4] * 6 < enn)
                    factor *= 0.5f;
            }
            sub_short_factor[chn][i] = factor;
        }
        {
            float   x = S->tab->attack_threshold[chn];
            for (i = 0; i < 12; i++)
                if (ns_attacks[chn][i / 3] == 0)
                    if (attack_intensity[i] > x)
                        ns_attacks[chn][i / 3] = (i % 3) + 1;
        }
        for (i = 1; i < 4; i++) {
            float const u = en_short[i - 1];
            float const v = en_short[i];
            float const m = (u > v) ? u : v;
            if (m < 40000) {
                if (u < 1.7f * v && v < 1.7f * u) {
                    if (i == 1 && ns_attacks[chn][0] <= ns_attacks[chn][i])
                        ns_attacks[chn][0] = 0;
                    ns_attacks[chn][i] = 0;
                }
            }
        }
        if (ns_attacks[chn][0] <= S->last_attacks[chn])
            ns_attacks[chn][0] = 0;
        if (S->last_attacks[chn] == 3 ||
            ns_attacks[chn][0] + ns_attacks[chn][1] + ns_attacks[chn][2] + ns_attacks[chn][3]) {
            ns_uselongblock = 0;
            if (ns_attacks[chn][1] && ns_attacks[chn][0])
                ns_attacks[chn][1] = 0;
            if (ns_attacks[chn][2] && ns_attacks[chn][1])
                ns_attacks[chn][2] = 0;
            if (ns_attacks[chn][3] && ns_attacks[chn][2])
                ns_attacks[chn][3] = 0;
        }
        if (chn < 2)
            uselongblock[chn] = ns_uselongblock;
        else if (ns_uselongblock == 0)
            uselongblock[0] = uselongblock[1] = 0;
        energy[chn] = S->tot_ener[chn];
    }
}

/* reference psymodel.c:1031-1131 */
static void
compute_masking_s(OrcStream * S, const float (*fftenergy_s)[LH_HBLKSIZE_S], float *eb, float *thr,
                  int chn, int sblock)
{
    LhPsyBand const *const gds = &S->tab->psy_s;
    float   max[LH_CBANDS], avg[LH_CBANDS];
    int     i, j, b;
    unsigned char mask_idx_s[LH_CBANDS];

    memset(max, 0, sizeof(max));
    memset(avg, 0, sizeof(avg));
    for (b = j = 0; b < gds->npart; ++b) {
        float   ebb = 0, m = 0;
        int const n = gds->numlines[b];
        for (i = 0; i < n; ++i, ++j) {
            float const el = fftenergy_s[sblock][j];
            ebb += el;
            if (m < el)
                m = el;
        }
        eb[b] = ebb;
        max[b] = m;
        avg[b] = ebb * gds->rnumlines[b];
    }
    calc_mask_index(gds, max, avg, mask_idx_s);
    for (j = b = 0; b < gds->npart; b++) {
        int     kk = gds->s3ind[b][0];
        int const last = gds->s3ind[b][1];
        int const delta = tab_mask_add_delta[mask_idx_s[b]];
        int     dd, dd_n;
        float   x, ecb, avg_mask;
        float const masking_lower = gds->masking_lower[b] * S->masking_lower;

        dd = mask_idx_s[kk];
        dd_n = 1;
        ecb = gds->s3[j] * eb[kk] * tab[mask_idx_s[kk]];
        ++j, ++kk;
        while (kk <= last) {
            dd += mask_idx_s[kk];
            dd_n += 1;
            x = gds->s3[j] * eb[kk] * tab[mask_idx_s[kk]];
            ecb = mask_add(S->tab, ecb, x, kk - b, delta);
            ++j, ++kk;
        }
        dd = (1 + 2 * dd) / (2 * dd_n);
        avg_mask = tab[dd] * 0.5f;
        ecb *= avg_mask;
        thr[b] = ecb;
        S->nb_s2[chn][b] = S->nb_s1[chn][b];
        S->nb_s1[chn][b] = ecb;
        x = max[b];
        x *= gds->minval[b];
        x *= avg_mask;
        if (thr[b] > x)
            thr[b] = x;
        if (masking_lower > 1)
            thr[b] *= masking_lower;
        if (thr[b] > eb[b])
            thr[b] = eb[b];
        if (masking_lower < 1)
            thr[b] *= masking_lower;
    }
    for (; b < LH_CBANDS; ++b) {
        eb[b] = 0;
        thr[b] = 0;
    }
}

/* reference psymodel.c:1134-1262 */
static void
compute_masking_l(OrcStream * S, const float fftenergy[LH_HBLKSIZE], float eb_l[LH_CBANDS],
                  float thr[LH_CBANDS], int chn)
{
    LhPsyBand const *const gdl = &S->tab->psy_l;
    float   max[LH_CBANDS], avg[LH_CBANDS];
    unsigned char mask_idx_l[LH_CBANDS + 2];
    int     k, b, j, i;

    /* calc_energy, reference psymodel.c:556-580 */
    for (b = j = 0; b < gdl->npart; ++b) {
        float   ebb = 0, m = 0;
        for (i = 0; i < gdl->numlines[b]; ++i, ++j) {
            float const el = fftenergy[j];
            ebb += el;
            if (m < el)
                m = el;
        }
        eb_l[b] = ebb;
        max[b] = m;
        avg[b] = ebb * gdl->rnumlines[b];
    }
    calc_mask_index(gdl, max, avg, mask_idx_l);

    k = 0;
    for (b = 0; b < gdl->npart; b++) {
        float   x, ecb, avg_mask, t;
        float const masking_lower = gdl->masking_lower[b] * S->masking_lower;
        int     kk = gdl->s3ind[b][0];
        int const last = gdl->s3ind[b][1];
        int const delta = tab_mask_add_delta[mask_idx_l[b]];
        int     dd = 0, dd_n = 0;

        dd = mask_idx_l[kk];
        dd_n += 1;
        ecb = gdl->s3[k] * eb_l[kk] * tab[mask_idx_l[kk]];
        ++k, ++kk;
        while (kk <= last) {
            dd += mask_idx_l[kk];
            dd_n += 1;
            x = gdl->s3[k] * eb_l[kk] * tab[mask_idx_l[kk]];
            t = mask_add(S->tab, ecb, x, kk - b, delta);
            ecb = t;
            ++k, ++kk;
        }
        dd = (1 + 2 * dd) / (2 * dd_n);
        avg_mask = tab[dd] * 0.5f;
        ecb *= avg_mask;

        if (S->blocktype_old[chn & 0x01] == LH_SHORT_TYPE) {
            float const ecb_limit = RPELEV * S->nb_l1[chn][b];
            if (ecb_limit > 0)
                thr[b] = (ecb < ecb_limit) ? ecb : ecb_limit;
            else {
                float const alt = eb_l[b] * NS_PREECHO_ATT2;
                thr[b] = (ecb < alt) ? ecb : alt;
            }
        }
        else {
            float   ecb_limit_2 = RPELEV2 * S->nb_l2[chn][b];
            float   ecb_limit_1 = RPELEV * S->nb_l1[chn][b];
            float   ecb_limit;
            if (ecb_limit_2 <= 0)
                ecb_limit_2 = ecb;
            if (ecb_limit_1 <= 0)
                ecb_limit_1 = ecb;
            if (S->blocktype_old[chn & 0x01] == LH_NORM_TYPE)
                ecb_limit = (ecb_limit_1 < ecb_limit_2) ? ecb_limit_1 : ecb_limit_2;
            else
                ecb_limit = ecb_limit_1;
            thr[b] = (ecb < ecb_limit) ? ecb : ecb_limit;
        }
        S->nb_l2[chn][b] = S->nb_l1[chn][b];
        S->nb_l1[chn][b] = ecb;
        x = max[b];
        x *= gdl->minval[b];
        x *= avg_mask;
        if (thr[b] > x)
            thr[b] = x;
        if (masking_lower > 1)
            thr[b] *= masking_lower;
        if (thr[b] > eb_l[b])
            thr[b] = eb_l[b];
        if (masking_lower < 1)
            thr[b] *= masking_lower;
    }
    for (; b < LH_CBANDS; ++b) {
        eb_l[b] = 0;
        thr[b] = 0;
    }
}

/* reference psymodel.c:1326-1388 */
static void
compute_MS_thresholds(const float eb[4][LH_CBANDS], float thr[4][LH_CBANDS],
                      const float cb_mld[LH_CBANDS], const float ath_cb[LH_CBANDS], float athlower,
                      float msfix, int n)
{
    float const msfix2 = msfix * 2.f;
    float   rside, rmid;
    int     b;
    for (b = 0; b < n; ++b) {
        float const ebM = eb[2][b];
        float const ebS = eb[3][b];
        float const thmL = thr[0][b];
        float const thmR = thr[1][b];
        float   thmM = thr[2][b];
        float   thmS = thr[3][b];
        if (thmL <= 1.58f * thmR && thmR <= 1.58f * thmL) {
            float const mld_m = cb_mld[b] * ebS;
            float const mld_s = cb_mld[b] * ebM;
            float const tmp_m = (thmS < mld_m) ? thmS : mld_m;
            float const tmp_s = (thmM < mld_s) ? thmM : mld_s;
            rmid = (thmM > tmp_m) ? thmM : tmp_m;
            rside = (thmS > tmp_s) ? thmS : tmp_s;
        }
        else {
            rmid = thmM;
            rside = thmS;
        }
        if (msfix > 0.f) {
            float   thmLR, thmMS;
            float const ath = ath_cb[b] * athlower;
            float const tmp_l = (thmL > ath) ? thmL : ath;
            float const tmp_r = (thmR > ath) ? thmR : ath;
            thmLR = (tmp_l < tmp_r) ? tmp_l : tmp_r;
            thmM = (rmid > ath) ? rmid : ath;
            thmS = (rside > ath) ? rside : ath;
            thmMS = thmM + thmS;
            if (thmMS > 0.f && (thmLR * msfix2) < thmMS) {
                float const f = thmLR * msfix2 / thmMS;
                thmM *= f;
                thmS *= f;
            }
            rmid = (thmM < rmid) ? thmM : rmid;
            rside = (thmS < rside) ? thmS : rside;
        }
        if (rmid > ebM)
            rmid = ebM;
        if (rside > ebS)
            rside = ebS;
        thr[2][b] = rmid;
        thr[3][b] = rside;
    }
}

/* reference psymodel.c:1397-1597 */
int
orc_psycho_anal(OrcStream * S, const float *const buffer[2], int gr_out,
                OrcRatio masking_ratio[2][2], OrcRatio masking_MS_ratio[2][2],
                float percep_entropy[2], float percep_MS_entropy[2], float energy[4],
                int blocktype_d[2])
{
    const LhConfig *cfg = S->cfg;
    const LhTables *T = S->tab;
    LhPsyBand const *const gdl = &T->psy_l;
    LhPsyBand const *const gds = &T->psy_s;
    OrcXmin last_thm[4];
    float   (*wsamp_l)[LH_BLKSIZE];
    float   (*wsamp_s)[3][LH_BLKSIZE_S];
    float   fftenergy[LH_HBLKSIZE];
    float   fftenergy_s[3][LH_HBLKSIZE_S];
    static float wsamp_L[2][LH_BLKSIZE];
    static float wsamp_S[2][3][LH_BLKSIZE_S];
    float   eb[4][LH_CBANDS], thr[4][LH_CBANDS];
    float   sub_short_factor[4][3];
    float   thmm;
    float const pcfact = 0.6f;
    float const ath_factor = (cfg->msfix > 0.f) ? (cfg->ATH_offset_factor * S->ath_adjust_factor) : 1.f;
    int     ns_attacks[4][4] = { {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0} };
    int     uselongblock[2];
    int     chn, sb, sblock;
    int const n_chn_out = cfg->channels;
    int const n_chn_psy = (cfg->mode == LH_MODE_JOINT_STEREO) ? 4 : n_chn_out;

    memcpy(&last_thm[0], &S->thm[0], sizeof(last_thm));
    attack_detection(S, buffer, gr_out, masking_ratio, masking_MS_ratio, energy, sub_short_factor,
                     ns_attacks, uselongblock);
    /* vbrpsy_compute_block_type, reference psymodel.c:1265-1286 */
    if (cfg->short_blocks == 1 && !(uselongblock[0] && uselongblock[1]))
        uselongblock[0] = uselongblock[1] = 0;
    for (chn = 0; chn < n_chn_out; chn++) {
        if (cfg->short_blocks == 2)
            uselongblock[chn] = 1;
        if (cfg->short_blocks == 3)
            uselongblock[chn] = 0;
    }

    for (chn = 0; chn < n_chn_psy; chn++) {
        int const ch01 = chn & 0x01;
        wsamp_l = wsamp_L + ch01;
        compute_fft_l(S, buffer, chn, fftenergy, wsamp_l);
        /* loudness approximation, reference psymodel.c:213-226, 743-752 */
        if (chn < 2) {
            int     i;
            float   loudness_power = 0.0;
            S->loudness_sq[gr_out][chn] = S->loudness_sq_save[chn];
            for (i = 0; i < LH_BLKSIZE / 2; ++i)
                loudness_power += fftenergy[i] * T->ath_eql_w[i];
            loudness_power *= VO_SCALE;
            S->loudness_sq_save[chn] = loudness_power;
        }
        compute_masking_l(S, fftenergy, eb[chn], thr[chn], chn);
    }
    if (cfg->mode == LH_MODE_JOINT_STEREO) {
        if ((uselongblock[0] + uselongblock[1]) == 2)
            compute_MS_thresholds((const float (*)[LH_CBANDS]) eb, thr, gdl->mld_cb, T->ath_cb_l,
                                  ath_factor, cfg->msfix, gdl->npart);
    }
    for (chn = 0; chn < n_chn_psy; chn++) {
        float   enn[LH_SBMAX_S], thm[LH_SBMAX_S];
        convert_partition2scalefac(gdl, eb[chn], thr[chn], &S->en[chn].l[0], &S->thm[chn].l[0]);
        /* convert_partition2scalefac_l_to_s, reference psymodel.c:421-439 */
        convert_partition2scalefac(&T->psy_l_to_s, eb[chn], thr[chn], enn, thm);
        for (sb = 0; sb < LH_SBMAX_S; ++sb) {
            float const scale = 1. / 64.f;
            float const tmp_enn = enn[sb];
            float const tmp_thm = thm[sb] * scale;
            for (sblock = 0; sblock < 3; ++sblock) {
                S->en[chn].s[sb][sblock] = tmp_enn;
                S->thm[chn].s[sb][sblock] = tmp_thm;
            }
        }
    }
    for (sblock = 0; sblock < 3; sblock++) {
        for (chn = 0; chn < n_chn_psy; ++chn) {
            int const ch01 = chn & 0x01;
            if (uselongblock[ch01]) {
                /* vbrpsy_skip_masking_s, reference psymodel.c:943-955 */
                if (sblock == 0) {
                    int     b;
                    for (b = 0; b < gds->npart; b++)
                        S->nb_s2[chn][b] = S->nb_s1[chn][b];
                }
            }
            else {
                wsamp_s = wsamp_S + ch01;
                compute_fft_s(S, buffer, chn, sblock, fftenergy_s, wsamp_s);
                compute_masking_s(S, (const float (*)[LH_HBLKSIZE_S]) fftenergy_s, eb[chn],
                                  thr[chn], chn, sblock);
            }
        }
        if (cfg->mode == LH_MODE_JOINT_STEREO) {
            if ((uselongblock[0] + uselongblock[1]) == 0)
                compute_MS_thresholds((const float (*)[LH_CBANDS]) eb, thr, gds->mld_cb,
                                      T->ath_cb_s, ath_factor, cfg->msfix, gds->npart);
        }
        for (chn = 0; chn < n_chn_psy; ++chn) {
            int const ch01 = chn & 0x01;
            if (!uselongblock[ch01]) {
                float   enn[LH_SBMAX_S], thm[LH_SBMAX_S];
                convert_partition2scalefac(gds, eb[chn], thr[chn], enn, thm);
                for (sb = 0; sb < LH_SBMAX_S; ++sb) {
                    S->en[chn].s[sb][sblock] = enn[sb];
                    S->thm[chn].s[sb][sblock] = thm[sb];
                }
            }
        }
    }
    /* short block pre-echo control, reference psymodel.c:1502-1553 */
    for (chn = 0; chn < n_chn_psy; chn++) {
        for (sb = 0; sb < LH_SBMAX_S; sb++) {
            float   new_thmm[3], prev_thm, t1, t2;
            for (sblock = 0; sblock < 3; sblock++) {
                thmm = S->thm[chn].s[sb][sblock];
                thmm *= NS_PREECHO_ATT0;
                t1 = t2 = thmm;
                if (sblock > 0)
                    prev_thm = new_thmm[sblock - 1];
                else
                    prev_thm = last_thm[chn].s[sb][2];
                if (ns_attacks[chn][sblock] >= 2 || ns_attacks[chn][sblock + 1] == 1)
                    t1 = ns_interp(prev_thm, thmm, NS_PREECHO_ATT1 * pcfact);
                thmm = (t1 < thmm) ? t1 : thmm;
                if (ns_attacks[chn][sblock] == 1)
                    t2 = ns_interp(prev_thm, thmm, NS_PREECHO_ATT2 * pcfact);
                else if ((sblock == 0 && S->last_attacks[chn] == 3)
                         || (sblock > 0 && ns_attacks[chn][sblock - 1] == 3)) {
                    switch (sblock) {
                    case 0:
                        prev_thm = last_thm[chn].s[sb][1];
                        break;
                    case 1:
                        prev_thm = last_thm[chn].s[sb][2];
                        break;
                    case 2:
                        prev_thm = new_thmm[0];
                        break;
                    }
                    t2 = ns_interp(prev_thm, thmm, NS_PREECHO_ATT2 * pcfact);
                }
                thmm = (t1 < thmm) ? t1 : thmm;
                thmm = (t2 < thmm) ? t2 : thmm;
                thmm *= sub_short_factor[chn][sblock];
                new_thmm[sblock] = thmm;
            }
            for (sblock = 0; sblock < 3; sblock++)
                S->thm[chn].s[sb][sblock] = new_thmm[sblock];
        }
    }
    for (chn = 0; chn < n_chn_psy; chn++)
        S->last_attacks[chn] = ns_attacks[chn][2];

    /* vbrpsy_apply_block_type, reference psymodel.c:1289-1319 */
    for (chn = 0; chn < n_chn_out; chn++) {
        int     blocktype = LH_NORM_TYPE;
        if (uselongblock[chn]) {
            if (S->blocktype_old[chn] == LH_SHORT_TYPE)
                blocktype = LH_STOP_TYPE;
        }
        else {
            blocktype = LH_SHORT_TYPE;
            if (S->blocktype_old[chn] == LH_NORM_TYPE)
                S->blocktype_old[chn] = LH_START_TYPE;
            if (S->blocktype_old[chn] == LH_STOP_TYPE)
                S->blocktype_old[chn] = LH_SHORT_TYPE;
        }
        blocktype_d[chn] = S->blocktype_old[chn];
        S->blocktype_old[chn] = blocktype;
    }
    for (chn = 0; chn < n_chn_psy; chn++) {
        float  *ppe;
        int     type;
        OrcRatio const *mr;
        if (chn > 1) {
            ppe = percep_MS_entropy - 2;
            type = LH_NORM_TYPE;
            if (blocktype_d[0] == LH_SHORT_TYPE || blocktype_d[1] == LH_SHORT_TYPE)
                type = LH_SHORT_TYPE;
            mr = &masking_MS_ratio[gr_out][chn - 2];
        }
        else {
            ppe = percep_entropy;
            type = blocktype_d[chn];
            mr = &masking_ratio[gr_out][chn];
        }
        if (type == LH_SHORT_TYPE)
            ppe[chn] = pecalc_s(T, mr, S->masking_lower);
        else
            ppe[chn] = pecalc_l(T, mr, S->masking_lower);
    }
    return 0;
}
