/*
 * lh_resample.c -- input rate -> output rate conversion in front of the batched encoder (host C).
 *
 * The reference converts inside lame_encode_buffer, one block of at most one frame of output
 * per fill_buffer call (reference util.c:531-697: a bank of Blackman-windowed sinc filters, one per
 * fractional offset, picked per output sample).  Which filter a sample gets and how the running
 * input time is rounded depend on where those blocks start, so the resampled signal is only
 * reproducible call by call: lh_rs_block is one such block, and the handle API strings the
 * blocks together exactly the way the reference's loop does (lh_api.cpp: encode_buffer_any).
 *
 * Arithmetic types follow the reference's FLOAT (= float) / double split line by line, because the
 * bytes downstream are compared with the reference's; built with -ffp-contract=off.
 */
#include <math.h>
#include <float.h>
#include <string.h>
#include "lh_host.h"

#define RS_PI 3.14159265358979323846

/* reference util.c:658: input within +-0.05 % of the output rate is passed through */
int
lh_rs_needed(int rate_in, int rate_out)
{
    int const lo = rate_out * 0.9995f;
    int const hi = rate_out * 1.0005f;
    return (rate_in < lo) || (hi < rate_in);
}

static int
common_divisor(int a, int b)
{
    while (b) {
        int const t = a % b;
        a = b;
        b = t;
    }
    return a;
}

/* one tap of the windowed sinc (reference util.c:483-505); x in taps from the window's start */
static float
window_tap(float x, float cutoff, int taps)
{
    float const wc = (float) (RS_PI * cutoff);
    float   win, d;
    x /= taps;
    if (x < 0)
        x = 0;
    if (x > 1)
        x = 1;
    d = (float) (x - .5);
    win = (float) (0.42 - 0.5 * cos(2 * x * RS_PI) + 0.08 * cos(4 * x * RS_PI));
    if (fabs(d) < 1e-9)
        return (float) (wc / RS_PI);
    return (float) (win * sin(taps * wc * d) / (RS_PI * taps * d));
}

void
lh_rs_init(LhResampler * r, int rate_in, int rate_out)
{
    int     i, j;
    memset(r, 0, sizeof(*r));
    r->rate_in = rate_in;
    r->rate_out = rate_out;
    r->ratio = (double) rate_in / (double) rate_out;
    r->phases = rate_out / common_divisor(rate_out, rate_in);
    if (r->phases > LH_RS_MAXPHASES)
        r->phases = LH_RS_MAXPHASES;
    /* 31 taps, 32 when the ratio is a whole number (the window is then centred on a sample) */
    r->taps = 31 + ((fabs(r->ratio - floor(.5 + r->ratio)) < FLT_EPSILON) ? 1 : 0);
    {
        float   cutoff = (float) (1.00 / r->ratio);
        if (cutoff > 1.00)
            cutoff = 1.00;
        for (j = 0; j <= 2 * r->phases; j++) {
            float   sum = 0.f;
            float const shift = (float) ((j - r->phases) / (2. * r->phases));
            for (i = 0; i <= r->taps; i++)
                sum += r->bank[j][i] = window_tap(i - shift, cutoff, r->taps);
            for (i = 0; i <= r->taps; i++)
                r->bank[j][i] /= sum;
        }
    }
}

/* One block: up to `want' output samples of channel ch from in[0..len), continuing after the
 * samples of the previous blocks (history[ch]).  Returns the number written, *used = input consumed. */
int
lh_rs_block(LhResampler * r, int ch, float *out, int want, const float *in, int len, int *used)
{
    int const taps = r->taps, keep = taps + 1, half = taps / 2;
    float  *hist = r->history[ch];
    int     k, j = 0, i;

    for (k = 0; k < want; k++) {
        double const t = k * r->ratio;  /* when output sample k is due, in input samples */
        float   shift, acc;
        int     phase;
        j = (int) floor(t - r->clock[ch]);
        if (taps + j - half >= len)
            break;              /* the window reaches past the input at hand */
        shift = (float) (t - r->clock[ch] - (j + .5 * (taps % 2)));
        phase = (int) floor((shift * 2 * r->phases) + r->phases + .5);
        acc = 0.f;
        for (i = 0; i <= taps; ++i) {
            int const at = i + j - half;
            float const y = (at < 0) ? hist[keep + at] : in[at];
            acc += y * r->bank[phase][i];
        }
        out[k] = acc;
    }
    *used = (len < taps + j - half) ? len : taps + j - half;
    /* the next block's output 0 is due at time 0; its input starts at clock[ch] */
    r->clock[ch] += *used - k * r->ratio;
    if (*used >= keep) {
        for (i = 0; i < keep; i++)
            hist[i] = in[*used + i - keep];
    }
    else {
        int const stay = keep - *used;
        for (i = 0; i < stay; ++i)
            hist[i] = hist[i + *used];
        for (j = 0; i < keep; ++i, ++j)
            hist[i] = in[j];
    }
    return k;
}
