/*
 * orc_resample.c -- CPU restatement of the reference's input-rate conversion and of the call
 * pattern it depends on (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).
 *
 * Follows reference util.c:483-505 (blackman), 520-654 (fill_buffer_resample), 656-697
 * (isResamplingNecessary, fill_buffer), lame.c:1708-1772 (the loop of lame_encode_buffer_sample_t)
 * and lame.c:2075-2120 (lame_encode_flush feeding zeros through the converter).
 * Parity: pinned against the compiled reference by tests/test_resample.py (bytes of whole
 * streams, several rate pairs and call patterns).
 */
#include <float.h>

#define ORC_RS_BPC 320

typedef struct {
    double  ratio;              /* samplerate_in / samplerate_out */
    int     bpc, filter_l;
    double  itime[2];
    float   old[2][40];
    float   filt[2 * ORC_RS_BPC + 1][40];
} OrcResample;

static float
orc_rs_blackman(float x, float fcn, int l)
{
    const double pi = 3.14159265358979323846;
    float   bkwn, x2;
    float const wcn = (float) (pi * fcn);
    x /= l;
    if (x < 0)
        x = 0;
    if (x > 1)
        x = 1;
    x2 = (float) (x - .5);
    bkwn = (float) (0.42 - 0.5 * cos(2 * x * pi) + 0.08 * cos(4 * x * pi));
    if (fabs(x2) < 1e-9)
        return (float) (wcn / pi);
    return (float) (bkwn * sin(l * wcn * x2) / (pi * l * x2));
}

int
orc_rs_needed(int in, int out)
{
    int const l = out * 0.9995f;
    int const h = out * 1.0005f;
    return (in < l) || (h < in) ? 1 : 0;
}

void
orc_rs_setup(OrcResample * R, int in, int out)
{
    int     a = out, b = in, i, j, intratio;
    float   fcn;
    memset(R, 0, sizeof(*R));
    while (b) {
        int const t = a % b;
        a = b;
        b = t;
    }
    R->ratio = (double) in / (double) out;
    R->bpc = out / a;
    if (R->bpc > ORC_RS_BPC)
        R->bpc = ORC_RS_BPC;
    intratio = (fabs(R->ratio - floor(.5 + R->ratio)) < FLT_EPSILON);
    fcn = (float) (1.00 / R->ratio);
    if (fcn > 1.00)
        fcn = 1.00;
    R->filter_l = 31 + intratio;
    for (j = 0; j <= 2 * R->bpc; j++) {
        float   sum = 0.;
        float const offset = (float) ((j - R->bpc) / (2. * R->bpc));
        for (i = 0; i <= R->filter_l; i++)
            sum += R->filt[j][i] = orc_rs_blackman(i - offset, fcn, R->filter_l);
        for (i = 0; i <= R->filter_l; i++)
            R->filt[j][i] /= sum;
    }
}

/* one fill_buffer_resample call */
int
orc_rs_fill(OrcResample * R, float *outbuf, int desired_len, const float *inbuf, int len, int *num_used, int ch)
{
    int const filter_l = R->filter_l, BLACKSIZE = filter_l + 1;
    float  *inbuf_old = R->old[ch];
    int     i, j = 0, k;
    for (k = 0; k < desired_len; k++) {
        double const time0 = k * R->ratio;
        float   offset, xvalue;
        int     joff;
        j = (int) floor(time0 - R->itime[ch]);
        if ((filter_l + j - filter_l / 2) >= len)
            break;
        offset = (float) (time0 - R->itime[ch] - (j + .5 * (filter_l % 2)));
        joff = (int) floor((offset * 2 * R->bpc) + R->bpc + .5);
        xvalue = 0.;
        for (i = 0; i <= filter_l; ++i) {
            int const j2 = i + j - filter_l / 2;
            float const y = (j2 < 0) ? inbuf_old[BLACKSIZE + j2] : inbuf[j2];
            xvalue += y * R->filt[joff][i];
        }
        outbuf[k] = xvalue;
    }
    *num_used = (len < filter_l + j - filter_l / 2) ? len : (filter_l + j - filter_l / 2);
    R->itime[ch] += *num_used - k * R->ratio;
    if (*num_used >= BLACKSIZE) {
        for (i = 0; i < BLACKSIZE; i++)
            inbuf_old[i] = inbuf[*num_used + i - BLACKSIZE];
    }
    else {
        int const n_shift = BLACKSIZE - *num_used;
        for (i = 0; i < n_shift; ++i)
            inbuf_old[i] = inbuf_old[i + *num_used];
        for (j = 0; i < BLACKSIZE; ++i, ++j)
            inbuf_old[i] = inbuf[j];
    }
    return k;
}

/* Whole stream: s16 input fed in calls of calls[0..ncalls) samples (0-terminated usage: the last
 * entry repeats until the input is used up), then the flush.  The converted signal (after the
 * input scaling of cfg, before the encoder) goes to out[0], out[1] (cap samples each); returns its
 * length, *nframes = frames the reference encodes, *padding = its encoder_padding. */
long
orc_resample_stream(const LhConfig * cfg, int rate_in, const short *l, const short *r, long n,
                    const int *calls, int ncalls, float *out0, float *out1, long cap, int *nframes, int *padding)
{
    OrcResample *R = (OrcResample *) malloc(sizeof(OrcResample));
    float  *in[2];
    long    fed = 0, pos = 0, mf_size = LH_MF_START, to_encode = LH_ENCDELAY + LH_POSTDELAY;
    int     frames = 0, c = 0, nch = cfg->channels, flushing = 0, frames_left = 0;
    float  *out[2];
    out[0] = out0;
    out[1] = out1;
    in[0] = (float *) malloc(sizeof(float) * 1152 * 64);
    in[1] = (float *) malloc(sizeof(float) * 1152 * 64);
    orc_rs_setup(R, rate_in, cfg->samplerate);
    for (;;) {
        int     m, i, at = 0;
        int const before = frames;
        if (!flushing && pos >= n) {
            int     samples_to_encode = (int) (to_encode - LH_POSTDELAY);
            int     end_padding;
            samples_to_encode += 16. / R->ratio;
            end_padding = 1152 - (samples_to_encode % 1152);
            if (end_padding < 576)
                end_padding += 1152;
            *padding = end_padding;
            frames_left = (samples_to_encode + end_padding) / 1152;
            flushing = 1;
        }
        if (flushing) {
            int     bunch;
            if (frames_left <= 0)
                break;
            bunch = (int) (LH_MF_NEEDED - mf_size);
            bunch *= R->ratio;
            if (bunch > 1152)
                bunch = 1152;
            if (bunch < 1)
                bunch = 1;
            m = bunch;
            for (i = 0; i < m; i++)
                in[0][i] = in[1][i] = 0;
        }
        else {
            m = calls[c < ncalls ? c : ncalls - 1];
            c++;
            if (m > n - pos)
                m = (int) (n - pos);
            if (m > 1152 * 64)
                m = 1152 * 64;
            for (i = 0; i < m; i++) {
                /* lame_copy_inbuffer, reference lame.c:1802-1834 */
                float const xl = l[pos + i], xr = r[pos + i];
                in[0][i] = xl * cfg->pcm_scale + xr * cfg->pcm_mix;
                in[1][i] = xl * (0.0f * cfg->pcm_scale) + xr * cfg->pcm_scale_r;
            }
            pos += m;
        }
        /* lame_encode_buffer_sample_t */
        while (m > 0) {
            int     n_in = 0, n_out = 0, ch;
            float   blk[2][1152];
            for (ch = 0; ch < nch; ch++)
                n_out = orc_rs_fill(R, blk[ch], 1152, in[ch] + at, m, &n_in, ch);
            for (i = 0; i < n_out && fed + i < cap; i++) {
                out[0][fed + i] = blk[0][i];
                out[1][fed + i] = (nch == 2) ? blk[1][i] : 0.f;
            }
            fed += n_out;
            m -= n_in;
            at += n_in;
            mf_size += n_out;
            to_encode += n_out;
            if (mf_size >= LH_MF_NEEDED) {
                frames++;
                mf_size -= 1152;
                to_encode -= 1152;
            }
        }
        if (flushing)
            frames_left -= (frames != before) ? 1 : 0;
    }
    *nframes = frames;
    free(in[0]);
    free(in[1]);
    free(R);
    return fed;
}

/* encode an already converted float stream: frame f reads x[1152 f - 528 ...), zero outside [0, n) */
int
orc_encode_stream_f(const LhConfig * cfg, const LhTables * tab, const float *l, const float *r, long n, int nf,
                    LhFrameOut * frames, int max_frames)
{
    OrcStream *S = (OrcStream *) malloc(sizeof(OrcStream));
    static float mf[2][LH_MF_NEEDED];
    int     f, i;
    orc_stream_init(S, cfg, tab);
    for (f = 0; f < nf; f++) {
        LhFrameOut tmp;
        long const base = 1152L * f - LH_MF_START;
        for (i = 0; i < LH_MF_NEEDED; i++) {
            long const p = base + i;
            mf[0][i] = (p >= 0 && p < n) ? l[p] : 0.f;
            mf[1][i] = (p >= 0 && p < n) ? r[p] : 0.f;
        }
        orc_encode_frame(S, mf[0], mf[1], &tmp);
        if (f < max_frames && frames)
            frames[f] = tmp;
    }
    free(S);
    return nf;
}
