/*
 * lh_resample.c -- input rate -> output rate conversion in front of the batched encoder (host C).
 *
 * What has to come out: exactly the float samples the reference's converter produces (reference
 * util.c:483-697 -- a polyphase bank of Blackman-windowed sinc kernels, the kernel of an output
 * sample picked by the fractional part of its input time), because the bytes downstream are
 * compared with the reference's.  The arithmetic (which products are float, which double, and the
 * order of every sum) is therefore fixed by the reference; the organisation here is this file's own:
 *
 *   - a converter is a `kernel bank' (designed once: lh_rs_init) plus, per channel, a `tail' of the
 *     last input samples and the input time at which the next block starts;
 *   - a block (lh_rs_block) sees the channel's signal as ONE virtual sequence, tail followed by the
 *     new input (rs_at), locates each output sample on it (rs_locate), gathers the span of input
 *     the kernel covers and takes the ordered dot product (rs_dot);
 *   - afterwards the tail is simply the last keep samples of that same virtual sequence.
 *
 * The reference re-bases its input time once per fill_buffer block, and which kernel a sample gets
 * depends on that rounding, so blocks must be cut where the reference cuts them: the handle API
 * strings them together the way lame_encode_buffer / lame_encode_flush do (lh_api.cpp).
 * Built with -ffp-contract=off.
 */
#include <math.h>
#include <float.h>
#include <string.h>
#include "lh_host.h"

static const double rs_pi = 3.14159265358979323846;

/* inputs within +-0.05 % of the output rate pass through unconverted (reference util.c:658) */
int
lh_rs_needed(int rate_in, int rate_out)
{
    int const band_lo = rate_out * 0.9995f, band_hi = rate_out * 1.0005f;
    return !(band_lo <= rate_in && rate_in <= band_hi);
}

/* kernels per unit of input time = rate_out / gcd(rate_out, rate_in), at most LH_RS_MAXPHASES */
static int
rs_phase_count(int rate_in, int rate_out)
{
    int     a = rate_out, b = rate_in, n;
    while (b != 0) {
        int const rest = a % b;
        a = b;
        b = rest;
    }
    n = rate_out / a;
    return n < LH_RS_MAXPHASES ? n : LH_RS_MAXPHASES;
}

/* Design kernel `ph' of the bank: tap i sits at i - shift on a window of `span' input samples,
 * shift = (ph - phases) / (2 phases) in [-1/2, +1/2].  Window and sinc are evaluated in the
 * reference's precision mix (position and cut-off are floats, the trigonometry is double), and
 * the taps are normalised by their float sum in tap order. */
static void
rs_design_kernel(LhResampler * r, int ph, float cutoff)
{
    int const span = r->taps;
    float const shift = (float) ((ph - r->phases) / (2. * r->phases));
    float const wc = (float) (rs_pi * cutoff);
    float  *tap = r->bank[ph];
    float   total = 0.f;
    int     i;
    for (i = 0; i <= span; i++) {
        float   u = i - shift, centre, value;
        u /= span;
        u = (u < 0) ? 0 : (u > 1) ? 1 : u;     /* position on the window, 0..1 */
        centre = (float) (u - .5);
        if (fabs(centre) < 1e-9)
            value = (float) (wc / rs_pi);
        else {
            float const blackman = (float) (0.42 - 0.5 * cos(2 * u * rs_pi) + 0.08 * cos(4 * u * rs_pi));
            value = (float) (blackman * sin(span * wc * centre) / (rs_pi * span * centre));
        }
        tap[i] = value;
        total += value;
    }
    for (i = 0; i <= span; i++)
        tap[i] /= total;
}

void
lh_rs_init(LhResampler * r, int rate_in, int rate_out)
{
    float   cutoff;
    int     whole, ph;
    memset(r, 0, sizeof(*r));
    r->rate_in = rate_in;
    r->rate_out = rate_out;
    r->ratio = (double) rate_in / (double) rate_out;
    r->phases = rs_phase_count(rate_in, rate_out);
    /* an odd span of 31, or 32 when every output sample falls on an input sample */
    whole = fabs(r->ratio - floor(.5 + r->ratio)) < FLT_EPSILON;
    r->taps = whole ? 32 : 31;
    cutoff = (float) (1.00 / r->ratio);
    if (cutoff > 1.00)
        cutoff = 1.00;
    for (ph = 0; ph <= 2 * r->phases; ph++)
        rs_design_kernel(r, ph, cutoff);
}

/* sample `at' of the channel's virtual sequence: negative positions are the tail kept from the
 * previous blocks (its last sample is position -1) */
static float
rs_at(const float *tail, int keep, const float *in, int at)
{
    return at < 0 ? tail[keep + at] : in[at];
}

/* where output sample k of the block sits: first input position of its span and the kernel */
typedef struct {
    int     first;              /* position of tap 0 on the virtual sequence */
    int     kernel;
} RsSpot;

static RsSpot
rs_locate(const LhResampler * r, double start, int k)
{
    RsSpot  s;
    double const due = k * r->ratio;            /* input time of the output sample, from the block's start */
    int const whole = (int) floor(due - start);
    float const frac = (float) (due - start - (whole + .5 * (r->taps % 2)));
    s.first = whole - r->taps / 2;
    s.kernel = (int) floor((frac * 2 * r->phases) + r->phases + .5);
    return s;
}

/* dot product of the span starting at `first' with one kernel, in tap order */
static float
rs_dot(const LhResampler * r, const float *tail, const float *in, RsSpot s)
{
    const float *tap = r->bank[s.kernel];
    float   acc = 0.f;
    int     i;
    for (i = 0; i <= r->taps; ++i)
        acc += rs_at(tail, r->taps + 1, in, s.first + i) * tap[i];
    return acc;
}

/* One block: up to `want' output samples of channel ch from in[0..len), continuing after the
 * previous blocks.  Returns the number written; *used = input samples consumed. */
int
lh_rs_block(LhResampler * r, int ch, float *out, int want, const float *in, int len, int *used)
{
    int const keep = r->taps + 1;
    float  *tail = r->history[ch];
    double const start = r->clock[ch];
    int     made = 0, reach = r->taps - r->taps / 2, taken, i;

    while (made < want) {
        RsSpot const s = rs_locate(r, start, made);
        reach = s.first + r->taps;      /* last input position the span touches */
        if (reach >= len)
            break;              /* the kernel reaches past the input at hand */
        out[made++] = rs_dot(r, tail, in, s);
    }
    /* The reference measures consumption by the span of the sample it stopped at (made or not);
     * with no sample attempted (want == 0) that is the span of position 0. */
    taken = (len < reach) ? len : reach;
    *used = taken;
    /* next block: its output 0 is due at time 0, its input begins at clock[ch] */
    r->clock[ch] = start + (taken - made * r->ratio);
    /* new tail = the last `keep' samples of [tail | in[0..taken)] */
    {
        float   next[34];
        for (i = 0; i < keep; i++)
            next[i] = rs_at(tail, keep, in, taken - keep + i);
        memcpy(tail, next, (size_t) keep * sizeof(float));
    }
    return made;
}
