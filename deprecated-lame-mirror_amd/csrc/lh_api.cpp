/*
 * lh_api.cpp -- C-ABI layer of liblamehip (declared in include/lamehip.h).
 *
 * Host side only: parameter collection (the lame_set_* subset of the reference,
 * set_get.c), lame_init_params -> constants + tables -> HBM, PCM staging, kernel
 * launches (lh_kernels.hip), D2H of the side-info payload and the serial bit
 * packer (lh_bitstream.c).  The per-frame arithmetic of the hot path runs only
 * in the HIP kernels; there is no CPU implementation of it in this library.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>
#include <mutex>
#include <thread>
#include <atomic>

#include "lamehip.h"
#include "lamehip_types.h"
#include "lh_host.h"
#include "lh_device.h"

extern "C" int lh_launch_encode(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf,
                                const LhStreamDesc * descs, LhStreamState * states,
                                LhFrameOut * out, uint8_t * bytes, int nstreams, void *stream);

/* the same kernel compiled for MPEG-2 / 2.5 streams (lh_kernels.hip with -DLH_LSF: one granule per frame) */
extern "C" int lh_launch_encode_lsf(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf,
                                    const LhStreamDesc * descs, LhStreamState * states,
                                    LhFrameOut * out, uint8_t * bytes, int nstreams, void *stream);

/* the MPEG-1 kernel once more, compiled for the new VBR loop (lh_kernels.hip with -DLH_VBRK: same source, the
 * instruction scheduling strategy that loop runs best with; csrc/Makefile) */
extern "C" int lh_launch_encode_vbr(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf,
                                    const LhStreamDesc * descs, LhStreamState * states,
                                    LhFrameOut * out, uint8_t * bytes, int nstreams, void *stream);

/* the split pipeline: the analysis kernels (lh_analysis.hip, lh_subband.hip), which do everything of a
 * frame that depends on the PCM alone for all frames of a launch at once, and the encode kernels compiled to start from
 * their output (lh_kernels.hip with -DLH_SPLIT); one set per frame geometry / scheduling variant as above */
#define LH_DECL_SPLIT(sfx) \
    extern "C" int lh_launch_analysis##sfx(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf, \
                                           const LhStreamDesc * descs, const LhStreamState * states, LhMidPools mid, \
                                           int nstreams, int max_frames, void *stream); \
    extern "C" int lh_launch_subband##sfx(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf, \
                                          const LhStreamDesc * descs, LhStreamState * states, LhMidPools mid, \
                                          int nstreams, int max_frames, void *stream);
LH_DECL_SPLIT()
LH_DECL_SPLIT(_lsf)
#define LH_DECL_Q(sfx) \
    extern "C" int lh_launch_encode_q##sfx(const LhConfig * cfg, const LhTables * T, const int16_t * pcm, const float *pcmf, \
                                           const LhStreamDesc * descs, LhStreamState * states, LhFrameOut * out, \
                                           uint8_t * bytes, int nstreams, void *stream, LhMidPools mid);
LH_DECL_Q()
LH_DECL_Q(_vbr)
LH_DECL_Q(_lsf)

extern "C" int lh_launch_selftest(unsigned *d_out, unsigned seed, void *stream);
extern "C" int lh_launch_summary(const LhStreamState * states, long long *sum, int nstreams, void *stream);
extern "C" int lh_launch_scatter(const int16_t * arena, int16_t * pool, long cap, const int *seg, int nseg, void *stream);

#define LAME_ID 0xFFF88E3Bu     /* reference util.h:482 */

static thread_local char g_err[512] = "";

static int
set_err(const char *what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return LAMEHIP_ERR_DEVICE;
}

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err(#call, e_); } while (0)

extern "C" const char *
lamehip_last_error(void)
{
    return g_err;
}

extern "C" int
lamehip_device_count(void)
{
    int     n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

/* samples per frame (1152; 576 for MPEG-2 / 2.5: one granule) and the samples that have to be buffered before a frame can
 * be encoded (BLKSIZE + framesize - FFTOFFSET: 1904 / 1328; reference lame.c:1627-1648) */
static inline int
fs_of(const LhConfig & c)
{
    return 576 * c.mode_gr;
}

static inline int
mfn_of(const LhConfig & c)
{
    return LH_BLKSIZE + 576 * c.mode_gr - LH_FFTOFFSET;
}

/* per device: the end of the last launch that filled it (lamehip_batch_encode) */
struct LhLaunchSerial {
    std::mutex lock;
    hipEvent_t ev = nullptr;
};

static LhLaunchSerial &
launch_serial(int device)
{
    static LhLaunchSerial per_device[64];
    return per_device[(device >= 0 && device < 64) ? device : 0];
}

/* device-resident constants shared by a handle or a batch */
struct LhDeviceConst {
    LhConfig *d_cfg = nullptr;
    LhTables *d_tab = nullptr;
    int     lsf = 0;            /* an MPEG-2 / 2.5 stream: the kernel object compiled for one granule per frame */
    int     vbrk = 0;           /* an MPEG-1 stream in the new VBR loop: the object scheduled for that loop */
    int launch(const int16_t * pcm, const float *pcmf, const LhStreamDesc * descs, LhStreamState * states,
               LhFrameOut * out, uint8_t * bytes, int nstreams, void *stream) const {
        return (lsf ? lh_launch_encode_lsf : vbrk ? lh_launch_encode_vbr : lh_launch_encode)
            (d_cfg, d_tab, pcm, pcmf, descs, states, out, bytes, nstreams, stream);
    }
    /* The split pipeline: analysis kernels for every frame of the launch, then the encode kernel that starts from what they
     * left in `mid'.  ev[0..1], when given, are recorded behind the analysis and the sub-band kernels (per-kernel times). */
    int launch_split(const int16_t * pcm, const float *pcmf, const LhStreamDesc * descs, LhStreamState * states,
                     LhFrameOut * out, uint8_t * bytes, int nstreams, int max_frames, const LhMidPools & mid, void *stream,
                     hipEvent_t * ev) const {
        int     rc;
        rc = (lsf ? lh_launch_analysis_lsf : lh_launch_analysis) (d_cfg, d_tab, pcm, pcmf, descs, states, mid, nstreams, max_frames, stream);
        if (rc)
            return rc;
        if (ev) {
            hipError_t const e = hipEventRecord(ev[0], (hipStream_t) stream);
            if (e != hipSuccess)
                return (int) e;
        }
        rc = (lsf ? lh_launch_subband_lsf : lh_launch_subband) (d_cfg, d_tab, pcm, pcmf, descs, states, mid, nstreams, max_frames, stream);
        if (rc)
            return rc;
        if (ev) {
            hipError_t const e = hipEventRecord(ev[1], (hipStream_t) stream);
            if (e != hipSuccess)
                return (int) e;
        }
        return (lsf ? lh_launch_encode_q_lsf : vbrk ? lh_launch_encode_q_vbr : lh_launch_encode_q)
            (d_cfg, d_tab, pcm, pcmf, descs, states, out, bytes, nstreams, stream, mid);
    }
    int upload(const LhConfig & cfg, const LhTables & tab) {
        lsf = (cfg.mode_gr == 1);
        vbrk = !lsf && (cfg.vbr == 1 || cfg.vbr == 4);
        HIPCHK(hipMalloc((void **) &d_cfg, sizeof(LhConfig)));
        HIPCHK(hipMalloc((void **) &d_tab, sizeof(LhTables)));
        HIPCHK(hipMemcpy(d_cfg, &cfg, sizeof(LhConfig), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_tab, &tab, sizeof(LhTables), hipMemcpyHostToDevice));
        return 0;
    }
    void release() {
        if (d_cfg)
            (void) hipFree(d_cfg);
        if (d_tab)
            (void) hipFree(d_tab);
        d_cfg = nullptr;
        d_tab = nullptr;
    }
};

/* Every handle and batch belongs to one HIP device: the one that was current when it was set up, or
 * the one named by lamehip_set_device / lamehip_batch_create_on.  Entry points that touch the device
 * make it current for the duration of the call and put the caller's device back afterwards. */
struct LhDeviceScope {
    int     prev = -1, mine = -1;
    explicit LhDeviceScope(int dev) {
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev && hipSetDevice(dev) == hipSuccess)
            mine = dev;
    }
    ~LhDeviceScope() {
        if (mine >= 0)
            (void) hipSetDevice(prev);
    }
};

struct lame_global_struct {
    unsigned class_id;
    int     device;             /* -1 until lame_init_params or lamehip_set_device fixes it */
    int     init_rc;            /* what lame_init_params returned */
    /* message callbacks (reference lame.h:346-348, util.c:707-760): errors of this library's calls on the
     * handle go to report_err; nullptr silences them */
    lame_report_function report_err, report_dbg, report_msg;
    LhUserParams p;
    int     out_samplerate;
    int     write_vbr_tag;
    int     inited;
    int     have_device;
    LhConfig cfg;
    LhTables *tab;              /* host copy */
    LhDeviceConst dc;
    /* streaming state of the single-handle path */
    std::vector < float >hl, hr; /* transformed samples [hist_base, fed) kept on the host (in_buffer_0/1) */
    long long hist_base;
    long long fed;
    int     frames_done;
    int     flushed;
    LhStreamState *d_state;
    float  *d_pcm;
    long long d_pcm_cap;
    LhStreamDesc *d_desc;
    LhFrameOut *d_out;
    int     d_out_cap;
    std::vector < LhFrameOut > h_out;
    LhFrameOut last_frame;
    int     have_last;
    LhBitstream bs;
    hipStream_t stream;
    int     nogap_total, nogap_current;         /* the frontend's --nogap bookkeeping (lame_set_nogap_*) */
    int     find_replaygain;    /* lame_set_findReplayGain: the title's radio gain goes into the LAME tag */
    LhReplayGain *rg;
    /* Xing/Info + LAME tag (host bookkeeping, lh_vbrtag.c) */
    LhVbrTag tag;
    int     tag_placeholder_pending;
    int     enc_padding;
    /* input rate != output rate: the transformed samples pass through this first (lh_resample.c) */
    LhResampler *rs;
    std::vector < float >tl, tr;
    /* what the frontend's progress display asks for (reference encoder.c:156-184 updateStats) */
    unsigned long num_samples;
    int     preset_vbr;         /* lame_set_preset chose a V0..V9 preset */
    int     frame_num_base;     /* frames before the last lame_init_bitstream */
    int     hist_mode[16][5];   /* [bitrate index | 15 = all][mode extension | 4 = frames] */
    int     hist_block[16][6];  /* [bitrate index | 15 = all][block type, 4 = mixed | 5 = granules] */
};

static int
valid(const lame_t g)
{
    return g && g->class_id == LAME_ID;
}

/* the reference's default message sink (util.c:707-716) */
static void
report_to_stderr(const char *format, va_list ap)
{
    (void) vfprintf(stderr, format, ap);
    fflush(stderr);
}

static void
report_through(lame_report_function f, const char *format, ...)
{
    va_list ap;
    if (!f)
        return;
    va_start(ap, format);
    f(format, ap);
    va_end(ap);
}

/* a failed call on a handle: what lamehip_last_error() holds also goes to the handle's errorf */
static int
report_failure(lame_t g, int rc)
{
    if (rc < 0 && g && g_err[0])
        report_through(g->report_err, "lamehip: %s\n", g_err);
    return rc;
}

extern "C" int
lame_set_errorf(lame_t g, lame_report_function f)
{
    if (!valid(g))
        return -1;
    g->report_err = f;
    return 0;
}

extern "C" int
lame_set_debugf(lame_t g, lame_report_function f)
{
    if (!valid(g))
        return -1;
    g->report_dbg = f;
    return 0;
}

extern "C" int
lame_set_msgf(lame_t g, lame_report_function f)
{
    if (!valid(g))
        return -1;
    g->report_msg = f;
    return 0;
}

extern "C" lame_t
lame_init(void)
{
    lame_t  g = new(std::nothrow) lame_global_struct();
    if (!g)
        return nullptr;
    g->class_id = LAME_ID;
    g->device = -1;
    g->init_rc = 0;
    g->report_err = g->report_dbg = g->report_msg = report_to_stderr;
    lh_params_default(&g->p);
    g->out_samplerate = 0;
    g->num_samples = 0xFFFFFFFFul;       /* MAX_U_32_NUM, reference lame.c:2336 */
    g->write_vbr_tag = 1;       /* reference default (lame.c:2340) */
    g->inited = 0;
    g->have_device = 0;
    g->tab = nullptr;
    g->hist_base = 0;
    g->fed = 0;
    g->frames_done = 0;
    g->flushed = 0;
    g->d_state = nullptr;
    g->d_pcm = nullptr;
    g->d_pcm_cap = 0;
    g->d_desc = nullptr;
    g->d_out = nullptr;
    g->d_out_cap = 0;
    g->have_last = 0;
    g->stream = nullptr;
    memset(&g->bs, 0, sizeof(g->bs));
    memset(&g->tag, 0, sizeof(g->tag));
    g->tag_placeholder_pending = 0;
    g->enc_padding = 0;
    g->nogap_total = g->nogap_current = 0;
    g->find_replaygain = 0;
    g->rg = nullptr;
    return g;
}

#define SETTER(name, field, type) \
    extern "C" int name(lame_t g, type v) { if (!valid(g)) return -1; g->field = (int) v; return 0; }
#define GETTER(name, expr, type) \
    extern "C" type name(const lame_t g) { if (!valid(g)) return (type) 0; return (type) (expr); }

SETTER(lame_set_in_samplerate, p.samplerate, int)
GETTER(lame_get_in_samplerate, g->p.samplerate, int)
SETTER(lame_set_num_channels, p.channels, int)
GETTER(lame_get_num_channels, g->p.channels, int)
SETTER(lame_set_out_samplerate, out_samplerate, int)
GETTER(lame_get_out_samplerate, g->inited ? g->cfg.samplerate : g->out_samplerate, int)
SETTER(lame_set_brate, p.brate, int)
GETTER(lame_get_brate, g->inited ? g->cfg.avg_bitrate : g->p.brate, int)
SETTER(lame_set_quality, p.quality, int)
GETTER(lame_get_quality, g->inited ? g->cfg.quality : g->p.quality, int)
SETTER(lame_set_bWriteVbrTag, write_vbr_tag, int)
GETTER(lame_get_bWriteVbrTag, g->write_vbr_tag, int)

/* ---- the frontend's tuning switches (reference set_get.c; semantics in lh_host_init.c:config_apply_tuning) ---- */
#define FSETTER(name, field) \
    extern "C" int name(lame_t g, float v) { if (!valid(g)) return -1; g->p.field = v; return 0; }
#define FGETTER(name, field) \
    extern "C" float name(const lame_t g) { if (!valid(g)) return 0; return g->p.field; }
SETTER(lame_set_ATHtype, p.ATHtype, int)                /* lame.h:502 */
GETTER(lame_get_ATHtype, g->inited ? g->cfg.ATHtype : g->p.ATHtype, int)
FSETTER(lame_set_ATHcurve, ATHcurve)
FGETTER(lame_get_ATHcurve, ATHcurve)
FSETTER(lame_set_ATHlower, ATH_lower_db)                /* lame.h:506 */
FGETTER(lame_get_ATHlower, ATH_lower_db)
SETTER(lame_set_athaa_type, p.athaa_type, int)          /* lame.h:510 */
GETTER(lame_get_athaa_type, g->p.athaa_type, int)
FSETTER(lame_set_athaa_sensitivity, athaa_sensitivity)  /* lame.h:521 */
FGETTER(lame_get_athaa_sensitivity, athaa_sensitivity)
SETTER(lame_set_ATHonly, p.ATHonly, int)                /* lame.h:490 */
GETTER(lame_get_ATHonly, g->p.ATHonly, int)
SETTER(lame_set_ATHshort, p.ATHshort, int)              /* lame.h:494 */
GETTER(lame_get_ATHshort, g->p.ATHshort, int)
SETTER(lame_set_noATH, p.noATH, int)                    /* lame.h:498 */
GETTER(lame_get_noATH, g->p.noATH, int)
SETTER(lame_set_highpassfreq, p.highpassfreq, int)      /* lame.h:477 */
GETTER(lame_get_highpassfreq, g->p.highpassfreq, int)
SETTER(lame_set_highpasswidth, p.highpasswidth, int)    /* lame.h:480 */
GETTER(lame_get_highpasswidth, g->p.highpasswidth, int)
SETTER(lame_set_exp_nspsytune, p.exp_nspsytune, int)    /* lame.h:421 */
GETTER(lame_get_exp_nspsytune, g->p.exp_nspsytune, int)
SETTER(lame_set_experimentalY, p.experimentalY, int)    /* lame.h:413 */
GETTER(lame_get_experimentalY, g->p.experimentalY, int)
SETTER(lame_set_experimentalZ, p.experimentalZ, int)    /* lame.h:417 */
GETTER(lame_get_experimentalZ, g->p.experimentalZ, int)
FSETTER(lame_set_compression_ratio, compression_ratio)  /* lame.h:355 */
extern "C" float
lame_get_compression_ratio(const lame_t g)
{
    if (!valid(g))
        return 0;
    return g->inited ? g->cfg.compression_ratio : g->p.compression_ratio;
}

extern "C" void
lame_set_msfix(lame_t g, double msfix)                  /* lame.h:424 */
{
    if (valid(g))
        g->p.msfix = (float) msfix;
}

extern "C" float
lame_get_msfix(const lame_t g)
{
    return valid(g) ? g->p.msfix : 0;
}

extern "C" int
lame_set_interChRatio(lame_t g, float ratio)            /* lame.h:543: 0 .. 1 */
{
    if (!valid(g) || !(0 <= ratio && ratio <= 1.0))
        return -1;
    g->p.interChRatio = ratio;
    return 0;
}
FGETTER(lame_get_interChRatio, interChRatio)

extern "C" int
lame_set_useTemporal(lame_t g, int on)                  /* lame.h:539: 0 / 1 */
{
    if (!valid(g) || on < 0 || on > 1)
        return -1;
    g->p.useTemporal = on;
    return 0;
}
GETTER(lame_get_useTemporal, g->p.useTemporal, int)

extern "C" int
lame_set_free_format(lame_t g, int on)                  /* lame.h:292: accepted here, refused by lame_init_params */
{
    if (!valid(g) || on < 0 || on > 1)
        return -1;
    g->p.free_format = on;
    return 0;
}
GETTER(lame_get_free_format, g->p.free_format, int)

/* switches of the reference that have nothing to act on in this library: the decoder (decode_only,
 * decode_on_the_fly), ReplayGain analysis, assembler variants.  Their setters take the "off" value and refuse
 * the "on" value; the getters report "off" / 0 like a reference build without those parts. */
extern "C" int lame_set_decode_only(lame_t g, int v) { return (valid(g) && v == 0) ? 0 : -1; }       /* lame.h:244 */
extern "C" int lame_get_decode_only(const lame_t) { return 0; }
extern "C" int lame_set_decode_on_the_fly(lame_t, int) { return -1; }       /* (a reference without DECODE_ON_THE_FLY) */
extern "C" int lame_get_decode_on_the_fly(const lame_t) { return 0; }
/* the frontend's default (--replaygain-fast): the input's radio gain is measured on the host beside the encode
 * (lh_replaygain.c) and stored in the LAME tag */
extern "C" int
lame_set_findReplayGain(lame_t g, int on)               /* lame.h:296 */
{
    if (!valid(g) || on < 0 || on > 1)
        return -1;
    g->find_replaygain = on;
    return 0;
}
GETTER(lame_get_findReplayGain, g->find_replaygain, int)
GETTER(lame_get_RadioGain, g->tag.radio_gain, int)
extern "C" int lame_get_AudiophileGain(const lame_t) { return 0; }
extern "C" float lame_get_PeakSample(const lame_t) { return 0; }
extern "C" int lame_get_noclipGainChange(const lame_t) { return 0; }
extern "C" float lame_get_noclipScale(const lame_t) { return 0; }
extern "C" int lame_set_asm_optimizations(lame_t g, int optim, int) { return valid(g) ? optim : -1; } /* lame.h:360 */
SETTER(lame_set_nogap_total, nogap_total, int)          /* lame.h:326: bookkeeping of the frontend's --nogap */
GETTER(lame_get_nogap_total, g->nogap_total, int)
SETTER(lame_set_nogap_currentindex, nogap_current, int)
GETTER(lame_get_nogap_currentindex, g->nogap_current, int)

/* reference lame.c: bitrate_table[version][index] */
extern "C" int
lame_get_bitrate(int mpeg_version, int table_index)     /* lame.h:1290 */
{
    static const int t[3][16] = {
        {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, -1},
        {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, -1},
        {0, 8, 16, 24, 32, 40, 48, 56, 64, -1, -1, -1, -1, -1, -1, -1}
    };
    if (0 <= mpeg_version && mpeg_version <= 2 && 0 <= table_index && table_index <= 15)
        return t[mpeg_version][table_index];
    return -1;
}

extern "C" int
lame_get_samplerate(int mpeg_version, int table_index)  /* lame.h:1291 */
{
    static const int t[3][4] = { {22050, 24000, 16000, -1}, {44100, 48000, 32000, -1}, {11025, 12000, 8000, -1} };
    if (0 <= mpeg_version && mpeg_version <= 2 && 0 <= table_index && table_index <= 3)
        return t[mpeg_version][table_index];
    return -1;
}

/* what this library answers where the reference names itself (version.c): the reference's numbers, since the
 * streams -- tag frame included -- are the reference's */
extern "C" const char *get_lame_version(void) { return "3.99.5"; }
extern "C" const char *get_lame_short_version(void) { return "3.99.5"; }
extern "C" const char *get_lame_very_short_version(void) { return "LAME3.99r"; }
extern "C" const char *get_psy_version(void) { return "1.0"; }
extern "C" const char *get_lame_url(void) { return "http://lame.sf.net"; }
extern "C" const char *get_lame_os_bitness(void) { return sizeof(void *) == 8 ? "64bits" : sizeof(void *) == 4 ? "32bits" : ""; }

/* lame_print_config / lame_print_internals (lame.h:678, 679): the settings as this library resolved them, through
 * the handle's msgf (the reference's wording is not reproduced; the frontend only passes the text on) */
extern "C" void
lame_print_config(const lame_t g)
{
    if (!valid(g) || !g->inited)
        return;
    report_through(g->report_msg, "liblamehip (MI355X): %d Hz -> %d Hz, %s, %s, quality %d, lowpass %d Hz\n", g->p.samplerate,
                   g->cfg.samplerate, g->cfg.mode == LH_MODE_MONO ? "mono" : g->cfg.mode == LH_MODE_JOINT_STEREO ? "joint stereo" :
                   g->cfg.mode == LH_MODE_DUAL ? "dual channel" : "stereo",
                   g->cfg.vbr == 0 ? "CBR" : g->cfg.vbr == 3 ? "ABR" : g->cfg.vbr == 2 ? "VBR (old)" : "VBR (new)", g->cfg.quality, g->cfg.lowpassfreq);
}

extern "C" void
lame_print_internals(const lame_t g)
{
    if (!valid(g) || !g->inited)
        return;
    report_through(g->report_msg, "liblamehip internals: bitrate %d kb/s (index %d), noise shaping %d / amp %d / stop %d, "
                   "best huffman %d, msfix %g, ATH type %d curve %g offset %g dB, temporal masking %d, short blocks %d\n",
                   g->cfg.avg_bitrate, g->cfg.bitrate_index, g->cfg.noise_shaping, g->cfg.noise_shaping_amp,
                   g->cfg.noise_shaping_stop, g->cfg.use_best_huffman, (double) g->cfg.msfix, g->cfg.ATHtype,
                   (double) g->cfg.ATHcurve, (double) g->cfg.ATH_offset_db, g->cfg.use_temporal_masking, g->cfg.short_blocks);
}

extern "C" int
lame_set_mode(lame_t g, MPEG_mode m)
{
    if (!valid(g))
        return -1;
    if ((int) m < 0 || m >= MAX_INDICATOR)
        return -1;
    g->p.mode = (m == NOT_SET) ? -1 : (int) m;
    return 0;
}

extern "C" MPEG_mode
lame_get_mode(const lame_t g)
{
    if (!valid(g))
        return NOT_SET;
    if (g->inited)
        return (MPEG_mode) g->cfg.mode;
    return g->p.mode < 0 ? NOT_SET : (MPEG_mode) g->p.mode;
}

extern "C" int
lame_set_VBR(lame_t g, vbr_mode v)
{
    if (!valid(g))
        return -1;
    g->p.vbr = (int) v;
    return 0;
}

extern "C" vbr_mode
lame_get_VBR(const lame_t g)
{
    return valid(g) ? (vbr_mode) g->p.vbr : vbr_off;
}

extern "C" int
lame_set_VBR_q(lame_t g, int q)
{
    int     ret = 0;
    if (!valid(g))
        return -1;
    if (q < 0) {                /* reference set_get.c:1127-1141: clamps and reports -1 */
        ret = -1;
        q = 0;
    }
    if (q > 9) {
        ret = -1;
        q = 9;
    }
    g->p.vbr_q = q;
    g->p.vbr_q_frac = 0;
    return ret;
}

/* --preset / lame_set_preset (reference set_get.c:2158-2166, presets.c:319-420): the named presets
 * and V0..V9 select the new VBR loop's quality, 8..320 an ABR mean, INSANE is CBR 320.  Applied at
 * call time like the reference, so later lame_set_* calls still override. */
extern "C" int
lame_set_preset(lame_t g, int preset)
{
    if (!valid(g))
        return -1;
    switch (preset) {
    case 1000:                 /* R3MIX */
        preset = 470;
        g->p.vbr = 4;
        break;
    case 1006:                 /* MEDIUM, MEDIUM_FAST */
    case 1007:
        preset = 460;
        g->p.vbr = 4;
        break;
    case 1001:                 /* STANDARD, STANDARD_FAST */
    case 1004:
        preset = 480;
        g->p.vbr = 4;
        break;
    case 1002:                 /* EXTREME, EXTREME_FAST */
    case 1005:
        preset = 500;
        g->p.vbr = 4;
        break;
    case 1003:                 /* INSANE */
        g->p.vbr = 0;
        g->p.brate = g->p.abr_kbps = 320;
        g->p.preset_kbps = (!g->p.preset_kbps || g->p.preset_kbps == 320) ? 320 : -1;   /* -1: two different ones */
        g->p.scale *= lh_abr_preset_scale(320);
        g->preset_vbr = 0;
        return 320;
    }
    if (preset >= 410 && preset <= 500 && preset % 10 == 0) {   /* V9 .. V0 */
        g->p.vbr_q = (500 - preset) / 10;
        g->p.vbr_q_frac = 0;
        g->preset_vbr = 1;      /* its tunings are the VBR loop's: not combined with CBR / ABR here */
        return preset;
    }
    if (8 <= preset && preset <= 320) {
        g->p.vbr = 3;
        g->p.abr_kbps = g->p.brate = preset;
        /* (its row's tuning values stay when the bitrate changes afterwards: lh_host_init.c) */
        g->p.preset_kbps = (!g->p.preset_kbps || g->p.preset_kbps == preset) ? preset : -1;
        g->p.scale *= lh_abr_preset_scale(preset);      /* and once more in lame_init_params, like the reference */
        g->preset_vbr = 0;
    }
    return preset;
}

/* -V n.f (reference set_get.c:1155-1175) */
extern "C" int
lame_set_VBR_quality(lame_t g, float q)
{
    int     ret = 0;
    if (!valid(g))
        return -1;
    if (0 > q) {
        ret = -1;
        q = 0;
    }
    if (9.999 < q) {
        ret = -1;
        q = 9.999;
    }
    g->p.vbr_q = (int) q;
    g->p.vbr_q_frac = q - g->p.vbr_q;
    return ret;
}

extern "C" float
lame_get_VBR_quality(const lame_t g)
{
    return valid(g) ? g->p.vbr_q + g->p.vbr_q_frac : 0;
}

SETTER(lame_set_VBR_min_bitrate_kbps, p.vbr_min_kbps, int)
GETTER(lame_get_VBR_min_bitrate_kbps, g->inited && g->cfg.vbr ? lh_tag_kbps(g->cfg.version, g->cfg.vbr_min_bitrate_index) : g->p.vbr_min_kbps, int)
SETTER(lame_set_VBR_max_bitrate_kbps, p.vbr_max_kbps, int)
GETTER(lame_get_VBR_max_bitrate_kbps, g->inited && g->cfg.vbr ? lh_tag_kbps(g->cfg.version, g->cfg.vbr_max_bitrate_index) : g->p.vbr_max_kbps, int)
SETTER(lame_set_VBR_hard_min, p.vbr_hard_min, int)
GETTER(lame_get_VBR_hard_min, g->p.vbr_hard_min, int)

GETTER(lame_get_VBR_q, g->inited ? g->cfg.vbr_q : g->p.vbr_q, int)
SETTER(lame_set_force_ms, p.force_ms, int)
GETTER(lame_get_force_ms, g->p.force_ms, int)
SETTER(lame_set_disable_reservoir, p.disable_reservoir, int)
GETTER(lame_get_disable_reservoir, g->p.disable_reservoir, int)
SETTER(lame_set_error_protection, p.error_protection, int)
GETTER(lame_get_error_protection, g->p.error_protection, int)
SETTER(lame_set_copyright, p.copyright, int)
GETTER(lame_get_copyright, g->p.copyright, int)
SETTER(lame_set_original, p.original, int)
GETTER(lame_get_original, g->p.original, int)
SETTER(lame_set_emphasis, p.emphasis, int)
GETTER(lame_get_emphasis, g->p.emphasis, int)
SETTER(lame_set_extension, p.extension, int)
GETTER(lame_get_extension, g->p.extension, int)
SETTER(lame_set_strict_ISO, p.strict_ISO, int)
GETTER(lame_get_strict_ISO, g->p.strict_ISO, int)
SETTER(lame_set_lowpassfreq, p.lowpassfreq, int)
GETTER(lame_get_lowpassfreq, g->inited ? g->cfg.lowpassfreq : g->p.lowpassfreq, int)
SETTER(lame_set_lowpasswidth, p.lowpasswidth, int)
GETTER(lame_get_lowpasswidth, g->p.lowpasswidth, int)

extern "C" int
lame_set_scale(lame_t g, float v)
{
    if (!valid(g))
        return -1;
    g->p.scale = v;
    return 0;
}

extern "C" int
lame_set_scale_left(lame_t g, float v)
{
    if (!valid(g))
        return -1;
    g->p.scale_left = v;
    return 0;
}

extern "C" int
lame_set_scale_right(lame_t g, float v)
{
    if (!valid(g))
        return -1;
    g->p.scale_right = v;
    return 0;
}

/* short block switches (reference set_get.c:1650-1846) */
extern "C" int
lame_set_allow_diff_short(lame_t g, int v)
{
    if (!valid(g))
        return -1;
    g->p.short_blocks = v ? 0 : 1;
    return 0;
}

extern "C" int
lame_set_no_short_blocks(lame_t g, int v)
{
    if (!valid(g) || v < 0 || v > 1)
        return -1;
    g->p.short_blocks = v ? 2 : 0;
    return 0;
}

extern "C" int
lame_set_force_short_blocks(lame_t g, int v)
{
    if (!valid(g) || v < 0 || v > 1)
        return -1;
    if (v == 1)
        g->p.short_blocks = 3;
    else if (g->p.short_blocks == 3)
        g->p.short_blocks = 0;
    return 0;
}

SETTER(lame_set_VBR_mean_bitrate_kbps, p.abr_kbps, int)
GETTER(lame_get_VBR_mean_bitrate_kbps, g->inited ? g->cfg.vbr_avg_bitrate_kbps : g->p.abr_kbps, int)

GETTER(lame_get_framesize, g->inited ? fs_of(g->cfg) : 576 * 2, int)
GETTER(lame_get_frameNum, g->frames_done - g->frame_num_base, int)
GETTER(lame_get_encoder_delay, LH_ENCDELAY, int)
GETTER(lame_get_encoder_padding, g->enc_padding, int)
/* ENCDELAY + POSTDELAY + samples taken in - samples encoded; 0 after the flush (reference lame.c:1737-1766, 2117) */
GETTER(lame_get_mf_samples_to_encode, (!g->inited || g->flushed) ? 0 : (int) (LH_ENCDELAY + LH_POSTDELAY + g->fed - (long long) fs_of(g->cfg) * g->frames_done), int)

extern "C" int
lame_set_num_samples(lame_t g, unsigned long n)
{
    if (!valid(g))
        return -1;
    g->num_samples = n;
    return 0;
}

extern "C" unsigned long
lame_get_num_samples(const lame_t g)
{
    return valid(g) ? g->num_samples : 0;
}

/* frames the stream will have, from the announced sample count (reference set_get.c:2120-2152) */
extern "C" int
lame_get_totalframes(const lame_t g)
{
    unsigned long n, padding;
    if (!valid(g) || !g->inited)
        return 0;
    n = g->num_samples;
    if (n == (0ul - 1ul))
        return 0;
    if (g->p.samplerate != g->cfg.samplerate && g->p.samplerate > 0) {
        double const q = (double) g->cfg.samplerate / g->p.samplerate;
        n *= q;
    }
    n += 576;
    padding = (unsigned long) fs_of(g->cfg) - (n % (unsigned long) fs_of(g->cfg));
    if (padding < 576)
        padding += (unsigned long) fs_of(g->cfg);
    n += padding;
    return (int) (n / (unsigned long) fs_of(g->cfg));
}

/* histograms over the frames encoded so far (reference lame.c:2461-2610) */
extern "C" void
lame_bitrate_kbps(const lame_t g, int bitrate_kbps[14])
{
    if (valid(g) && g->inited)
        for (int i = 0; i < 14; i++)
            bitrate_kbps[i] = lh_tag_kbps(g->inited ? g->cfg.version : 1, i + 1);
}

extern "C" void
lame_bitrate_hist(const lame_t g, int bitrate_count[14])
{
    if (valid(g) && g->inited)
        for (int i = 0; i < 14; i++)
            bitrate_count[i] = g->hist_mode[i + 1][4];
}

extern "C" void
lame_stereo_mode_hist(const lame_t g, int stmode_count[4])
{
    if (valid(g) && g->inited)
        for (int i = 0; i < 4; i++)
            stmode_count[i] = g->hist_mode[15][i];
}

extern "C" void
lame_bitrate_stereo_mode_hist(const lame_t g, int bitrate_stmode_count[14][4])
{
    if (valid(g) && g->inited)
        for (int j = 0; j < 14; j++)
            for (int i = 0; i < 4; i++)
                bitrate_stmode_count[j][i] = g->hist_mode[j + 1][i];
}

extern "C" void
lame_block_type_hist(const lame_t g, int btype_count[6])
{
    if (valid(g) && g->inited)
        for (int i = 0; i < 6; i++)
            btype_count[i] = g->hist_block[15][i];
}

extern "C" void
lame_bitrate_block_type_hist(const lame_t g, int bitrate_btype_count[14][6])
{
    if (valid(g) && g->inited)
        for (int j = 0; j < 14; j++)
            for (int i = 0; i < 6; i++)
                bitrate_btype_count[j][i] = g->hist_block[j + 1][i];
}
GETTER(lame_get_version, g->inited ? g->cfg.version : 1, int)

static int init_params_once(lame_t g);

extern "C" int
lame_init_params(lame_t g)
{
    int     rc;
    if (!valid(g))
        return -1;
    if (g->inited)
        return g->init_rc;      /* a second call reports what the first one found (also its failure) */
    g_err[0] = 0;               /* what the report callback prints is this call's, never an earlier call's */
    rc = init_params_once(g);
    if (g->inited)
        g->init_rc = rc;
    return report_failure(g, rc);
}

static int
init_params_once(lame_t g)
{
    LhInitAux aux;
    g->p.samplerate_out = g->out_samplerate;
    if (g->preset_vbr && g->p.vbr != 1 && g->p.vbr != 2 && g->p.vbr != 4) {
        snprintf(g_err, sizeof(g_err), "a V0..V9 preset without lame_set_VBR(vbr_mtrh / vbr_mt / vbr_rh) is outside the accelerated path");
        return -1;
    }
    if (g->p.preset_kbps < 0) {
        snprintf(g_err, sizeof(g_err), "two different bitrate presets (lame_set_preset 8..320 / INSANE) on one handle are outside the accelerated path");
        return -1;
    }
    if (g->p.preset_kbps && (g->preset_vbr || g->p.vbr == 1 || g->p.vbr == 2 || g->p.vbr == 4)) {
        /* e.g. --preset insane --vbr-old, --preset 192 --preset extreme: the reference then runs a VBR loop with what the
         * bitrate preset's row left in its tuning options -- a combination nobody asks for, not rebuilt here */
        snprintf(g_err, sizeof(g_err), "a bitrate preset (lame_set_preset 8..320 / INSANE) followed by a VBR mode is outside the accelerated path");
        return -1;
    }
    if (lh_config_resolve(&g->p, &g->cfg, &aux) != 0) {
        snprintf(g_err, sizeof(g_err),
                 "unsupported settings for the MI355X path (need an MPEG output rate, 1 or 2 input channels)");
        return -1;
    }
    /* (a call that failed before g->inited may be repeated: what it had allocated is reused) */
    if (!g->tab)
        g->tab = (LhTables *) malloc(sizeof(LhTables));
    if (!g->tab) {
        snprintf(g_err, sizeof(g_err), "out of memory (tables)");
        return -2;
    }
    if (lh_tables_build(&g->cfg, &aux, g->tab) != 0) {
        snprintf(g_err, sizeof(g_err), "table generation failed");
        return -1;
    }
    if (!g->bs.buf && lh_bs_init(&g->bs) != 0) {
        snprintf(g_err, sizeof(g_err), "out of memory (bit stream buffer)");
        return -2;
    }
    if (lh_rs_needed(g->p.samplerate, g->cfg.samplerate)) {
        if (!g->rs)
            g->rs = (LhResampler *) malloc(sizeof(LhResampler));
        if (!g->rs) {
            snprintf(g_err, sizeof(g_err), "out of memory (sample rate converter)");
            return -2;
        }
        lh_rs_init(g->rs, g->p.samplerate, g->cfg.samplerate);
    }
    if (g->find_replaygain) {
        if (!g->rg)
            g->rg = (LhReplayGain *) malloc(sizeof(LhReplayGain));
        if (!g->rg || lh_rg_start(g->rg, g->cfg.samplerate) != 0) {
            snprintf(g_err, sizeof(g_err), "ReplayGain analysis could not be set up");
            return -6;          /* the reference's code for it (lame.c:1264-1268) */
        }
    }
    g->inited = 1;              /* host constants are valid from here on (lamehip_get_*) */
    /* the tag frame is reserved at the head of the stream (reference InitVbrTag); when it does
     * not fit the reference silently switches it off */
    if (g->write_vbr_tag && lh_tag_init(&g->tag, &g->cfg) > 0)
        g->tag_placeholder_pending = 1;
    else
        g->write_vbr_tag = 0;
    g->tag.samplerate_in = g->p.samplerate;
    g->tag.radio_gain_on = g->find_replaygain;
    g->tag.radio_gain = 0;
    g->tag.nogap_total = g->nogap_total;
    g->tag.nogap_current = g->nogap_current;
    if (lamehip_device_count() <= 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device: liblamehip has no CPU encode path");
        g->have_device = 0;
        return LAMEHIP_ERR_NODEVICE;
    }
    if (g->device < 0 && hipGetDevice(&g->device) != hipSuccess)
        g->device = 0;
    if (g->device >= lamehip_device_count()) {
        snprintf(g_err, sizeof(g_err), "lamehip_set_device: no HIP device %d", g->device);
        return LAMEHIP_ERR_NODEVICE;
    }
    {
        LhDeviceScope const on_device(g->device);
        int     rc = g->dc.upload(g->cfg, *g->tab);
        LhStreamState s0;
        if (rc)
            return rc;
        HIPCHK(hipStreamCreate(&g->stream));
        HIPCHK(hipMalloc((void **) &g->d_state, sizeof(LhStreamState)));
        HIPCHK(hipMalloc((void **) &g->d_desc, sizeof(LhStreamDesc)));
        lh_state_init(&s0, &g->cfg);
        HIPCHK(hipMemcpy(g->d_state, &s0, sizeof(s0), hipMemcpyHostToDevice));
    }
    g->have_device = 1;
    return 0;
}

/* the reserved tag frame leaves with the first bytes the caller gets (reference lame.c:1744-1748:
 * copy_buffer(.., 0) at the start of every lame_encode_buffer call) */
static int
emit_tag_placeholder(lame_t g, unsigned char *mp3buf, int mp3buf_size, int *written)
{
    if (!g->tag_placeholder_pending)
        return 0;
    if (mp3buf_size != 0 && mp3buf_size - *written < g->tag.total_frame_size)
        return -1;
    {
        unsigned char *h = mp3buf + *written;
        *written += lh_tag_placeholder(&g->tag, &g->cfg, h);
        if (g->have_last)       /* a later file of a --nogap run: the header carries the last frame's mode extension */
            h[3] = (unsigned char) ((h[3] & 0xcf) | ((g->last_frame.mode_ext & 3) << 4));
    }
    g->tag_placeholder_pending = 0;
    return 0;
}

/* encode frames [frames_done, upto) of the single-handle stream and append the packed bytes */
static int
handle_encode_frames(lame_t g, int upto, unsigned char *mp3buf, int mp3buf_size, int *written)
{
    int const f0 = g->frames_done, nf = upto - f0;
    long long p0, p1, n;
    LhStreamDesc d;
    if (nf <= 0)
        return 0;
    /* samples touched: priming of frame 0 reaches 1152 further back (all zero there) */
    p0 = (long long) fs_of(g->cfg) * f0 - LH_MF_START - fs_of(g->cfg);
    if (p0 < g->hist_base)
        p0 = g->hist_base;
    p1 = (long long) fs_of(g->cfg) * (upto - 1) - LH_MF_START + LH_MF_NEEDED;
    if (p1 > g->fed)
        p1 = g->fed;
    n = p1 > p0 ? p1 - p0 : 0;
    if (n > g->d_pcm_cap) {
        /* the new buffer first: a failed allocation leaves the old one (and its size) in place */
        float  *bigger = nullptr;
        HIPCHK(hipMalloc((void **) &bigger, (size_t) (n + 4096) * 2 * sizeof(float)));
        if (g->d_pcm)
            (void) hipFree(g->d_pcm);
        g->d_pcm = bigger;
        g->d_pcm_cap = n + 4096;
    }
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(g->d_pcm, &g->hl[(size_t) (p0 - g->hist_base)], (size_t) n * sizeof(float),
                              hipMemcpyHostToDevice, g->stream));
        HIPCHK(hipMemcpyAsync(g->d_pcm + g->d_pcm_cap, &g->hr[(size_t) (p0 - g->hist_base)],
                              (size_t) n * sizeof(float), hipMemcpyHostToDevice, g->stream));
    }
    if (nf > g->d_out_cap) {
        LhFrameOut *bigger = nullptr;
        HIPCHK(hipMalloc((void **) &bigger, (size_t) (nf + 8) * sizeof(LhFrameOut)));
        if (g->d_out)
            (void) hipFree(g->d_out);
        g->d_out = bigger;
        g->d_out_cap = nf + 8;
    }
    d.pcm_l = 0;
    d.pcm_r = g->d_pcm_cap;
    d.pcm_base = p0;
    d.nsamples = g->fed;
    d.out_index = 0;
    d.frame_begin = f0;
    d.frame_end = upto;
    d.bytes_base = d.bytes_cap = 0;
    d.flush = d.mid_rel = 0;
    HIPCHK(hipMemcpyAsync(g->d_desc, &d, sizeof(d), hipMemcpyHostToDevice, g->stream));
    {
        int     rc = g->dc.launch((const int16_t *) 0, g->d_pcm, g->d_desc, g->d_state, g->d_out, (uint8_t *) 0, 1,
                                  (void *) g->stream);
        if (rc)
            return set_err("kernel launch", (hipError_t) rc);
    }
    g->h_out.resize((size_t) nf);
    HIPCHK(hipMemcpyAsync(g->h_out.data(), g->d_out, (size_t) nf * sizeof(LhFrameOut),
                          hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    for (int i = 0; i < nf; i++) {
        int     k;
        if (lh_bs_format_frame(&g->bs, &g->cfg, g->tab, &g->h_out[(size_t) i]) != 0) {
            int     n = snprintf(g_err, sizeof(g_err), "inconsistent device payload (packer check %d) at frame %d",
                                 g->bs.error, f0 + i);
            for (int q = 0; q < 4 && n > 0 && n < (int) sizeof(g_err); q++) {
                const LhGranule *gi = &g->h_out[(size_t) i].gr[q >> 1][q & 1];
                n += snprintf(g_err + n, sizeof(g_err) - (size_t) n, " [bv %d c1 %d gg %d bt %d ts %d/%d/%d r %d/%d]",
                              gi->big_values, gi->count1, gi->global_gain, gi->block_type, gi->table_select[0],
                              gi->table_select[1], gi->table_select[2], gi->region0_count, gi->region1_count);
            }
            return LAMEHIP_ERR_PAYLOAD;
        }
        k = lh_bs_copy(&g->bs, mp3buf + *written, mp3buf_size ? mp3buf_size - *written : 0);
        if (k < 0)
            return -1;
        {
            const LhFrameOut & fo = g->h_out[(size_t) i];
            int const bi = fo.bitrate_index & 15, me = fo.mode_ext & 3;
            g->hist_mode[bi][4]++;
            g->hist_mode[15][4]++;
            if (g->cfg.channels == 2) {
                g->hist_mode[bi][me]++;
                g->hist_mode[15][me]++;
            }
            for (int gr = 0; gr < 2; gr++)
                for (int ch = 0; ch < g->cfg.channels; ch++) {
                    int const bt = fo.gr[gr][ch].mixed_block_flag ? 4 : (fo.gr[gr][ch].block_type & 3);
                    g->hist_block[bi][bt]++;
                    g->hist_block[bi][5]++;
                    g->hist_block[15][bt]++;
                    g->hist_block[15][5]++;
                }
        }
        if (g->write_vbr_tag) {
            lh_tag_add_frame(&g->tag, lh_tag_kbps(g->cfg.version, g->h_out[(size_t) i].bitrate_index));  /* reference encoder.c:550-551 */
            lh_tag_crc(&g->tag, mp3buf + *written, k);          /* reference bitstream.c:1082-1088 */
        }
        *written += k;
    }
    g->last_frame = g->h_out[(size_t) nf - 1];
    g->have_last = 1;
    g->frames_done = upto;
    /* drop history that no later frame (nor its priming) can touch */
    {
        long long keep = (long long) fs_of(g->cfg) * g->frames_done - LH_MF_START - 64;
        if (keep > g->hist_base) {
            size_t  drop = (size_t) (keep - g->hist_base);
            if (drop > g->hl.size())
                drop = g->hl.size();
            g->hl.erase(g->hl.begin(), g->hl.begin() + (long) drop);
            g->hr.erase(g->hr.begin(), g->hr.begin() + (long) drop);
            g->hist_base += (long long) drop;
        }
    }
    return 0;
}

/* lame_encode_buffer_template + lame_copy_inbuffer (reference lame.c:1786-1872): samples of any
 * type become sample_t through the transform matrix scaled by the type's norm; the frames that
 * became complete are encoded */
template < typename T > static int encode_buffer_impl(lame_t g, const T * l, const T * r, int nsamples, int jump,
                                                      float norm, unsigned char *mp3buf, int mp3buf_size);

template < typename T > static int
encode_buffer_any(lame_t g, const T * l, const T * r, int nsamples, int jump, float norm, unsigned char *mp3buf,
                  int mp3buf_size)
{
    g_err[0] = 0;
    return report_failure(g, encode_buffer_impl(g, l, r, nsamples, jump, norm, mp3buf, mp3buf_size));
}

template < typename T > static int
encode_buffer_impl(lame_t g, const T * l, const T * r, int nsamples, int jump, float norm, unsigned char *mp3buf,
                   int mp3buf_size)
{
    LhDeviceScope const on_device(valid(g) ? g->device : -1);
    int     written = 0, rc, avail;
    if (!valid(g) || !g->inited)
        return -3;
    if (!g->have_device)
        return LAMEHIP_ERR_NODEVICE;
    if (nsamples == 0)
        return 0;
    if (nsamples < 0)
        return -1;
    if (g->p.channels == 1)
        r = l;                  /* one input channel: buffer_r is not read (reference lame.c:1855-1866) */
    if (!l || !r)
        return 0;
    {
        /* pcm_transform: [0] = { pcm_scale, pcm_mix }, [1] = { 0, pcm_scale_r } */
        float const m00 = norm * g->cfg.pcm_scale, m01 = norm * g->cfg.pcm_mix;
        float const m10 = norm * (0.0f * g->cfg.pcm_scale), m11 = norm * g->cfg.pcm_scale_r;
        std::vector < float >&dl = g->rs ? g->tl : g->hl;
        std::vector < float >&dr = g->rs ? g->tr : g->hr;
        size_t const at = g->rs ? 0 : g->hl.size();
        dl.resize(at + (size_t) nsamples);
        dr.resize(at + (size_t) nsamples);
        for (int i = 0; i < nsamples; i++) {
            float const xl = (float) l[(size_t) i * (size_t) jump];
            float const xr = (float) r[(size_t) i * (size_t) jump];
            dl[at + (size_t) i] = xl * m00 + xr * m01;
            dr[at + (size_t) i] = xl * m10 + xr * m11;
        }
    }
    if (g->rs) {
        /* the reference's loop (lame.c:1708-1772, fill_buffer util.c:665-697): blocks of at most one
         * frame of output until the call's input is used up; every output channel sees the same
         * block boundaries */
        int     pos = 0, left = nsamples;
        while (left > 0) {
            float   blk[2][1152];
            int     used = 0, made = 0;
            for (int ch = 0; ch < g->cfg.channels; ch++)
                made = lh_rs_block(g->rs, ch, blk[ch], fs_of(g->cfg), (ch ? g->tr.data() : g->tl.data()) + pos, left, &used);
            if (g->cfg.channels == 1)
                memset(blk[1], 0, sizeof(blk[1]));
            g->hl.insert(g->hl.end(), blk[0], blk[0] + made);
            g->hr.insert(g->hr.end(), blk[1], blk[1] + made);
            if (g->rg)
                (void) lh_rg_block(g->rg, blk[0], blk[1], made, g->cfg.channels);       /* reference lame.c:1715-1720 */
            g->fed += made;
            pos += used;
            left -= used;
        }
    }
    else {
        /* (the reference's loop hands the analysis what one fill_buffer call took in: at most a frame's samples) */
        if (g->rg) {
            size_t const at = g->hl.size() - (size_t) nsamples;
            for (int pos = 0; pos < nsamples; pos += fs_of(g->cfg)) {
                int const m = nsamples - pos > fs_of(g->cfg) ? fs_of(g->cfg) : nsamples - pos;
                (void) lh_rg_block(g->rg, g->hl.data() + at + (size_t) pos, g->hr.data() + at + (size_t) pos, m, g->cfg.channels);
            }
        }
        g->fed += nsamples;
    }
    g->flushed = 0;
    /* a frame is encoded whenever 1904 samples are buffered behind the 528-sample
     * lead-in (reference lame.c:1737-1769) */
    avail = (LH_MF_START + g->fed >= mfn_of(g->cfg))
        ? (int) ((LH_MF_START + g->fed - mfn_of(g->cfg)) / fs_of(g->cfg) + 1) : 0;
    if (emit_tag_placeholder(g, mp3buf, mp3buf_size, &written))
        return -1;
    rc = handle_encode_frames(g, avail, mp3buf, mp3buf_size, &written);
    if (rc)
        return rc;
    return written;
}

extern "C" int
lame_encode_buffer(lame_t g, const short int l[], const short int r[], const int nsamples,
                   unsigned char *mp3buf, const int mp3buf_size)
{
    return encode_buffer_any(g, l, r, nsamples, 1, 1.0f, mp3buf, mp3buf_size);
}

extern "C" int
lame_encode_buffer_interleaved(lame_t g, short int pcm[], int num_samples, unsigned char *mp3buf,
                               int mp3buf_size)
{
    return encode_buffer_any(g, pcm, pcm + 1, num_samples, 2, 1.0f, mp3buf, mp3buf_size);
}

/* +/- 32768 full scale (reference lame.c:1884-1890) */
extern "C" int
lame_encode_buffer_float(lame_t g, const float l[], const float r[], const int nsamples, unsigned char *mp3buf,
                         const int mp3buf_size)
{
    return encode_buffer_any(g, l, r, nsamples, 1, 1.0f, mp3buf, mp3buf_size);
}

/* +/- 1.0 full scale (reference lame.c:1894-1930) */
extern "C" int
lame_encode_buffer_ieee_float(lame_t g, const float l[], const float r[], const int nsamples, unsigned char *mp3buf,
                              const int mp3buf_size)
{
    return encode_buffer_any(g, l, r, nsamples, 1, 32767.0f, mp3buf, mp3buf_size);
}

extern "C" int
lame_encode_buffer_interleaved_ieee_float(lame_t g, const float pcm[], const int nsamples, unsigned char *mp3buf,
                                          const int mp3buf_size)
{
    return encode_buffer_any(g, pcm, pcm + 1, nsamples, 2, 32767.0f, mp3buf, mp3buf_size);
}

extern "C" int
lame_encode_buffer_ieee_double(lame_t g, const double l[], const double r[], const int nsamples,
                               unsigned char *mp3buf, const int mp3buf_size)
{
    return encode_buffer_any(g, l, r, nsamples, 1, 32767.0f, mp3buf, mp3buf_size);
}

extern "C" int
lame_encode_buffer_interleaved_ieee_double(lame_t g, const double pcm[], const int nsamples, unsigned char *mp3buf,
                                           const int mp3buf_size)
{
    return encode_buffer_any(g, pcm, pcm + 1, nsamples, 2, 32767.0f, mp3buf, mp3buf_size);
}

/* +/- MAX_INT full scale (reference lame.c:1934-1942) */
extern "C" int
lame_encode_buffer_int(lame_t g, const int l[], const int r[], const int nsamples, unsigned char *mp3buf,
                       const int mp3buf_size)
{
    float const norm = (float) (1.0 / (1L << (8 * sizeof(int) - 16)));
    return encode_buffer_any(g, l, r, nsamples, 1, norm, mp3buf, mp3buf_size);
}

/* +/- MAX_LONG full scale (reference lame.c:1945-1953) */
extern "C" int
lame_encode_buffer_long2(lame_t g, const long l[], const long r[], const int nsamples, unsigned char *mp3buf,
                         const int mp3buf_size)
{
    float const norm = (float) (1.0 / (1L << (8 * sizeof(long) - 16)));
    return encode_buffer_any(g, l, r, nsamples, 1, norm, mp3buf, mp3buf_size);
}

/* +/- 32768 full scale in a long (reference lame.c:1956-1962) */
extern "C" int
lame_encode_buffer_long(lame_t g, const long l[], const long r[], const int nsamples, unsigned char *mp3buf,
                        const int mp3buf_size)
{
    return encode_buffer_any(g, l, r, nsamples, 1, 1.0f, mp3buf, mp3buf_size);
}

static int finish_stream(lame_t g, unsigned char *mp3buf, int size, int written);

/* lame_encode_flush with the resampler in the way (reference lame.c:2075-2120): the number of
 * frames still owed follows from the samples buffered plus the resampler's delay, and zeros are fed
 * through lame_encode_buffer -- resampler included -- in bunches sized to complete one frame at a
 * time until those frames have come out */
static int
flush_resampled(lame_t g, unsigned char *mp3buf, int size)
{
    static const short zeros[1152] = { 0 };
    double const ratio = g->rs->ratio;
    /* mf_samples_to_encode - POSTDELAY, with mf_samples_to_encode = ENCDELAY + POSTDELAY + fed - 1152 frames */
    int     owed = (int) (576 + g->fed - (long long) fs_of(g->cfg) * g->frames_done);
    int     padding, frames_left, written = 0;
    owed += 16. / ratio;
    padding = fs_of(g->cfg) - (owed % fs_of(g->cfg));
    if (padding < 576)
        padding += fs_of(g->cfg);
    g->enc_padding = padding;
    frames_left = (owed + padding) / fs_of(g->cfg);
    while (frames_left > 0) {
        int const before = g->frames_done;
        int     bunch = mfn_of(g->cfg) - (int) (LH_MF_START + g->fed - (long long) fs_of(g->cfg) * g->frames_done);
        int     k;
        bunch *= ratio;
        if (bunch > 1152)
            bunch = 1152;
        if (bunch < 1)
            bunch = 1;
        k = encode_buffer_any(g, zeros, zeros, bunch, 1, 1.0f, mp3buf + written, size ? size - written : 0);
        if (k < 0)
            return k;
        written += k;
        frames_left -= (g->frames_done != before) ? 1 : 0;
    }
    return finish_stream(g, mp3buf, size, written);
}

extern "C" int
lame_encode_flush(lame_t g, unsigned char *mp3buf, int size)
{
    LhDeviceScope const on_device(valid(g) ? g->device : -1);
    int     written = 0, rc, total;
    if (!valid(g) || !g->inited)
        return -3;
    if (!g->have_device)
        return LAMEHIP_ERR_NODEVICE;
    if (g->flushed)
        return 0;               /* reference lame.c:2076-2079 */
    if (g->rs)
        return flush_resampled(g, mp3buf, size);
    if (g->rg) {
        /* the reference flushes by feeding zeros through lame_encode_buffer, (mf_needed - mf_size) <= 1152 at a time,
         * until the frames it owes are out (lame.c:2093-2117); the analysis hears those zeros */
        static const float zeros[1152] = { 0 };
        long long fed = g->fed;
        int     frames = g->frames_done;
        int const owed = (int) (576 + fed - (long long) fs_of(g->cfg) * frames);
        int     padding = fs_of(g->cfg) - (owed % fs_of(g->cfg)), frames_left;
        if (padding < 576)
            padding += fs_of(g->cfg);
        frames_left = (owed + padding) / fs_of(g->cfg);
        while (frames_left > 0) {
            int     bunch = mfn_of(g->cfg) - (int) (LH_MF_START + fed - (long long) fs_of(g->cfg) * frames);
            bunch = bunch > 1152 ? 1152 : (bunch < 1 ? 1 : bunch);
            (void) lh_rg_block(g->rg, zeros, zeros, bunch, g->cfg.channels);
            fed += bunch;
            if (LH_MF_START + fed - (long long) fs_of(g->cfg) * frames >= mfn_of(g->cfg)) {
                frames++;
                frames_left--;
            }
        }
    }
    total = lh_total_frames_fs((long) g->fed, fs_of(g->cfg));
    g->enc_padding = lh_end_padding_fs((long) g->fed, fs_of(g->cfg));     /* reference lame.c:2088-2091 */
    if (emit_tag_placeholder(g, mp3buf, size, &written))
        return -1;
    rc = handle_encode_frames(g, total, mp3buf, size, &written);
    if (rc)
        return rc;
    return finish_stream(g, mp3buf, size, written);
}

/* pad the last frame out and hand over the rest of the bytes (reference lame.c:2122-2160) */
static int
finish_stream(lame_t g, unsigned char *mp3buf, int size, int written)
{
    int     k;
    lh_bs_flush(&g->bs, &g->cfg, g->have_last ? &g->last_frame : nullptr);
    k = lh_bs_copy(&g->bs, mp3buf + written, size ? size - written : 0);
    if (k < 0)
        return -1;
    if (g->rg)
        g->tag.radio_gain = lh_rg_finish(g->rg);        /* save_gain_values, reference lame.c:1565-1580 */
    if (g->write_vbr_tag)
        lh_tag_crc(&g->tag, mp3buf + written, k);
    written += k;
    g->flushed = 1;
    /* the reference zeroes the reservoir after padding out the last frame (bitstream.c:886-888) */
    {
        LhStreamState s;
        if (hipMemcpy(&s, g->d_state, sizeof(s), hipMemcpyDeviceToHost) == hipSuccess) {
            s.ResvSize = 0;
            s.main_data_begin = 0;
            (void) hipMemcpy(g->d_state, &s, sizeof(s), hipMemcpyHostToDevice);
        }
    }
    return written;
}

/* --nogap: close the bitstream at a file boundary without draining the sample buffers (reference
 * lame.c:1988-2001): the pending frames are padded out and the reservoir starts from zero, so the
 * pieces decode on their own and, concatenated, without a gap. */
extern "C" int
lame_encode_flush_nogap(lame_t g, unsigned char *mp3buf, int size)
{
    LhDeviceScope const on_device(valid(g) ? g->device : -1);
    int     was, k;
    if (!valid(g) || !g->inited)
        return -3;
    if (!g->have_device)
        return LAMEHIP_ERR_NODEVICE;
    was = g->flushed;
    k = finish_stream(g, mp3buf, size, 0);
    g->flushed = was;
    return k;
}

/* start the next file of a --nogap run: frame counter, histograms and a fresh tag frame
 * (reference lame.c:2006-2035) */
extern "C" int
lame_init_bitstream(lame_t g)
{
    if (!valid(g) || !g->inited)
        return -3;
    g->frame_num_base = g->frames_done;
    memset(g->hist_mode, 0, sizeof(g->hist_mode));
    memset(g->hist_block, 0, sizeof(g->hist_block));
    {
        uint16_t const crc = g->tag.music_crc;  /* the reference's music CRC runs on across the files */
        if (g->write_vbr_tag && lh_tag_init(&g->tag, &g->cfg) > 0) {
            g->tag.samplerate_in = g->p.samplerate;
            g->tag.music_crc = crc;
            g->tag_placeholder_pending = 1;
        }
    }
    return 0;
}

/* reference lame.h:970, VbrTag.c:900: the final tag frame that replaces the placeholder at the
 * head of the stream; 0 when the tag is off or nothing was encoded; the needed size when `size'
 * is too small */
extern "C" size_t
lame_get_lametag_frame(const lame_t g, unsigned char *buffer, size_t size)
{
    if (!valid(g) || !g->inited || !g->write_vbr_tag)
        return 0;
    g->tag.nogap_total = g->nogap_total;       /* (the frontend sets these per file, after lame_init_params) */
    g->tag.nogap_current = g->nogap_current;
    return (size_t) lh_tag_frame(&g->tag, &g->cfg, g->cfg.vbr_q, g->enc_padding,
                                 g->have_last ? g->last_frame.mode_ext : 0, buffer, (long) size);
}

extern "C" int
lame_close(lame_t g)
{
    LhDeviceScope const on_device(valid(g) ? g->device : -1);
    if (!valid(g))
        return -3;
    g->class_id = 0;
    if (g->d_state)
        (void) hipFree(g->d_state);
    if (g->d_pcm)
        (void) hipFree(g->d_pcm);
    if (g->d_desc)
        (void) hipFree(g->d_desc);
    if (g->d_out)
        (void) hipFree(g->d_out);
    if (g->stream)
        (void) hipStreamDestroy(g->stream);
    g->dc.release();
    if (g->bs.buf)
        lh_bs_free(&g->bs);
    free(g->tab);
    free(g->rs);
    free(g->rg);
    delete  g;
    return 0;
}

/* runs the device self-test of the wave primitives; returns the number of
 * mismatches (0 = pass) or a negative error */
extern "C" int
lamehip_selftest(void)
{
    unsigned *d = nullptr, h = 0xffffffffu;
    if (lamehip_device_count() <= 0)
        return LAMEHIP_ERR_NODEVICE;
    HIPCHK(hipMalloc((void **) &d, sizeof(unsigned)));
    for (unsigned seed = 1; seed <= 4; seed++) {
        int     rc = lh_launch_selftest(d, seed * 7919u, nullptr);
        if (rc)
            return set_err("selftest launch", (hipError_t) rc);
        HIPCHK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
        if (h != 0)
            break;
    }
    (void) hipFree(d);
    return (int) h;
}

/* sizes of the POD layouts this build was compiled with (ABI check for bindings) */
extern "C" int
lamehip_abi_sizeof(int which)
{
    switch (which) {
    case 0:
        return (int) sizeof(LhConfig);
    case 1:
        return (int) sizeof(LhTables);
    case 2:
        return (int) sizeof(LhFrameOut);
    case 3:
        return (int) sizeof(LhGranule);
    case 4:
        return (int) sizeof(LhStreamState);
    case 5:
        return (int) sizeof(LhStreamDesc);
    }
    return -1;
}

extern "C" int
lamehip_get_config(const lame_t g, void *out, int size)
{
    if (!valid(g) || !g->inited || size < (int) sizeof(LhConfig))
        return -1;
    memcpy(out, &g->cfg, sizeof(LhConfig));
    return (int) sizeof(LhConfig);
}

extern "C" int
lamehip_get_tables(const lame_t g, void *out, int size)
{
    if (!valid(g) || !g->inited || !g->tab || size < (int) sizeof(LhTables))
        return -1;
    memcpy(out, g->tab, sizeof(LhTables));
    return (int) sizeof(LhTables);
}

/* ====================================================================== */
/* batch extension                                                          */

/* Batches go through the split pipeline unless LAMEHIP_FUSED=1 asks for the single fused kernel (A/B measurements; the
 * handle API always uses the fused kernel: one frame per launch has nothing to analyse ahead). */
static int
batch_use_split(void)
{
    const char *e = getenv("LAMEHIP_FUSED");
    return !(e && e[0] == '1');
}

/* LAMEHIP_SPLIT_DENY=1, looked at before every launch: this launch takes the fused kernel as if the pools could not be had
 * (test aid: a batch whose launches change kernels mid-stream, tests/test_gpu_parity.py) */
static int
batch_split_denied(void)
{
    const char *e = getenv("LAMEHIP_SPLIT_DENY");
    return e && e[0] == '1';
}

struct lamehip_batch {
    int     device;
    LhConfig cfg;
    LhTables *tab;
    LhDeviceConst dc;
    int     B;
    long    cap;
    int16_t *d_pcm;             /* [B][2][cap] */
    LhStreamState *d_state;
    LhStreamState *d_state0;    /* the streams' initial states (batch_reset_states) */
    LhStreamDesc *d_desc;
    LhFrameOut *d_out;
    long long out_cap;
    std::vector < long >len;
    std::vector < int >nframes;
    std::vector < long long >out_off;
    /* device bit packing (lamehip_batch_set_device_packing) */
    int     dev_pack;
    uint8_t *d_bytes;
    long long bytes_cap;
    std::vector < long long >bytes_off;
    std::vector < LhStreamDesc > h_desc;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    float   last_ms;
    int     encoded;
    /* input rate != output rate: set_pcm converts on the host (as the reference's frontend would have
     * it: lame_encode_buffer calls of 1152 input samples, then the flush) into a float pool */
    int     rate_in;
    LhResampler *rs;
    float  *d_pcmf;             /* [B][2][capf] */
    long    capf;
    std::vector < int >padding; /* encoder_padding per stream (tag frame) */
    /* incremental use (lamehip_batch_append ...): samples in the pool / frames encoded per stream, a
     * packer and the bytes not yet drained per stream, and the pinned staging area of the next
     * lamehip_batch_encode_available (see h_stage below) */
    int     incremental;
    std::vector < long >fed;
    std::vector < int >done;
    std::vector < int >staged;
    std::vector < LhBitstream > packer;
    std::vector < std::vector < unsigned char > >pending;
    std::vector < LhFrameOut > last;
    std::vector < char >have_last;
    /* pinned / HBM: [B descriptors][LH_STAGE_SEGS x 4 ints: the chunks][arena of staged samples, back to back] */
    unsigned char *h_stage, *d_stage;
    long    stage_arena_at;     /* byte offset of the arena */
    long    stage_cap;          /* samples the arena holds */
    long    stage_used;         /* samples staged */
    int     stage_nseg;         /* chunks staged */
    int     finished;           /* lamehip_batch_finish has run: the streams are closed */
    std::vector < LhFrameOut > h_new;
    /* pinned host side of a pipelined batch (lamehip_batch_pcm_host_ptr / _upload / _fetch): the mirror of the
     * s16 pool the caller (or lamehip_batch_set_pcm) writes, which reaches HBM with one asynchronous copy on
     * the batch's stream; the device packer's bytes and a two-word summary per stream (bytes, status) on the
     * way back.  Another batch's copies and kernel run meanwhile (each batch has its own stream). */
    int16_t *h_pcm;
    std::vector < char >row_dirty;      /* streams whose mirror rows are newer than the pool */
    int     n_dirty;
    unsigned char *h_bytes;
    long long h_bytes_cap;
    long long *d_sum, *h_sum;
    int     fetched;            /* the bytes of the last encode are in (or on their way into) h_bytes */
    /* the copies of a pipelined batch have streams of their own (different hardware queues from the kernel's, and
     * from each other: an upload queued behind the previous round's download on one stream cost 12 % of the
     * pipeline's throughput), tied to the kernel's stream by events */
    hipStream_t up_stream, down_stream;
    hipEvent_t ev_up, ev_sum, ev_down;
    int     up_pending, down_pending;
    int     up_inflight;        /* an H2D copy out of the pinned mirror may still be running (host view: cleared only after ev_up) */
    int     launched;           /* a kernel was launched on this batch and ev1 recorded (survives lamehip_batch_reset) */
    /* the split pipeline's pools (one record of each per frame of a launch, like d_out) and the events between its kernels */
    LhMidPools mid;
    long long mid_cap;
    int     split;              /* this batch's launches go through the split pipeline (batch_use_split) */
    hipEvent_t ev_part[2];
    hipEvent_t ev_wait;         /* behind the launch; only ever polled (hipEventQuery between short sleeps in lamehip_batch_sync): the
                                 * host thread sleeps instead of spinning on the stream (a rank per GPU must not burn a CPU per rank
                                 * while its kernel runs) */
    float   part_ms[3];         /* analysis, sub-band, encode kernel of the last launch (0: fused launch) */
    int     last_split;
    /* a launch in windows of frames (batch_plan): the windows' descriptors (host, then HBM: [window][stream]) and three events
     * per window (its start, behind its analysis kernels, behind its sub-band kernel) */
    std::vector < LhStreamDesc > h_wdesc;
    LhStreamDesc *d_wdesc;
    long long wdesc_cap;
    std::vector < hipEvent_t > ev_win;
    int     last_windows;       /* sub-launches of the last launch (1: the whole launch at once) */
};

/* what batch_plan decides about a launch (outside the devices' launch order: it may allocate) and batch_launch carries out */
struct LhLaunchPlan {
    long long total;            /* frames of the launch */
    int     max_frames;         /* of its longest stream */
    int     split;              /* the split pipeline (else the fused kernel) */
    int     window;             /* frames per stream and sub-launch; 0: the whole launch at once */
    int     nwin;
};

/* room for `need' records in the split pipeline's pool (0), or not (-1: the pool is gone, *free_records says how many would
 * fit and lamehip_last_error() why) */
static int
batch_mid_reserve(lamehip_batch * b, long long need, long long *free_records = nullptr)
{
    /* (the encode kernel touches the record BEHIND the one it works on, the launch's last frame included: one spare
     * record has to exist whatever the launch's total is -- a later launch whose total equals the capacity must not
     * read past the pool) */
    if (free_records)
        *free_records = 0;
    if (need + 1 <= b->mid_cap)
        return 0;
    size_t  free_b = 0, total_b = 0;
    long long const want = need + 64;
    if (b->mid.frames)
        (void) hipFree(b->mid.frames);
    b->mid.frames = nullptr;
    b->mid_cap = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
        free_b = 0;
    {
        /* LAMEHIP_MID_BUDGET_MB=n: no more than n MB count as free (test aid: the window size as a device with little
         * memory left would choose it) */
        const char *e = getenv("LAMEHIP_MID_BUDGET_MB");
        long const mb = e ? strtol(e, nullptr, 10) : 0;
        if (mb > 0 && (size_t) mb * 1000000u < free_b)
            free_b = (size_t) mb * 1000000u;
    }
    if (free_records)
        *free_records = (long long) (0.8 * (double) free_b / (double) sizeof(LhMidFrame)) - 64;
    if ((double) want * (double) sizeof(LhMidFrame) > 0.8 * (double) free_b) {
        /* (not an error: the launch runs in windows or takes the fused kernel; lamehip_last_error() says why) */
        snprintf(g_err, sizeof(g_err), "split pipeline: %lld frames need %.1f GB of analysis records, %.1f GB free",
                 need, (double) want * (double) sizeof(LhMidFrame) / 1e9, (double) free_b / 1e9);
        return -1;
    }
    if (hipMalloc((void **) &b->mid.frames, (size_t) want * sizeof(LhMidFrame)) != hipSuccess) {
        (void) hipGetLastError();
        b->mid.frames = nullptr;
        snprintf(g_err, sizeof(g_err), "split pipeline: hipMalloc of %.1f GB of analysis records failed",
                 (double) want * (double) sizeof(LhMidFrame) / 1e9);
        return -1;
    }
    b->mid_cap = want;
    return 0;
}

/* LAMEHIP_MID_WINDOW=n, looked at before every launch: the split pipeline works through a launch in windows of n frames per
 * stream whatever the memory would allow (tuning / test aid; 0 or unset: windows only when the records of the whole launch
 * do not fit) */
static int
batch_window_env(void)
{
    const char *e = getenv("LAMEHIP_MID_WINDOW");
    long const v = e ? strtol(e, nullptr, 10) : 0;
    return (v > 0 && v < (1l << 30)) ? (int) v : 0;
}

#define LH_MID_WINDOW_MIN 64    /* frames: below that the fused kernel is the better launch (a sub-launch ends with its slowest stream) */

/* records a launch in windows of w frames needs at once (every stream's first window is its largest) */
static long long
batch_window_records(const lamehip_batch * b, const LhStreamDesc * h_descs, int w)
{
    long long n = 0;
    for (int s = 0; s < b->B; s++) {
        int const nf = h_descs[s].frame_end - h_descs[s].frame_begin;
        if (nf > 0)
            n += nf < w ? nf : w;
    }
    return n;
}

/* What the launch of the frames `h_descs' names will be: the split pipeline over the whole launch when its analysis records
 * fit the device (34 KB per frame: 81 GB at 1024 x 60 s), else the split pipeline over windows of as many frames per stream as
 * do fit -- each window a launch of its own, analysis kernels then encode kernel, the streams' state carried in
 * LhStreamState as between two launches of an incremental batch --, else (under LH_MID_WINDOW_MIN frames per window, more
 * than 65 535 streams, LAMEHIP_FUSED / LAMEHIP_SPLIT_DENY) the fused kernel.  The windows' descriptors go to HBM here, on the
 * batch's stream.  Allocates: call it before the devices' launch order is taken. */
static int
batch_plan(lamehip_batch * b, const LhStreamDesc * h_descs, LhLaunchPlan * p)
{
    long long free_records = 0;
    int     nactive = 0;
    memset(p, 0, sizeof(*p));
    for (int s = 0; s < b->B; s++) {
        int const nf = h_descs[s].frame_end - h_descs[s].frame_begin;
        if (nf > 0) {
            p->total += nf;
            nactive++;
            if (nf > p->max_frames)
                p->max_frames = nf;
        }
    }
    p->nwin = 1;
    /* (the analysis and sub-band kernels index the stream by blockIdx.y, which ends at 65535: a larger batch keeps the
     * fused kernel, whose grid is one-dimensional) */
    if (!(b->split && p->total > 0 && b->B <= 65535 && !batch_split_denied()))
        return 0;
    int     w = batch_window_env();
    if (w >= p->max_frames)
        w = 0;
    if (w == 0) {
        if (batch_mid_reserve(b, p->total, &free_records) == 0) {
            p->split = 1;
            return 0;
        }
        w = nactive ? (int) (free_records / nactive < p->max_frames ? free_records / nactive : p->max_frames) : 0;
        if (w < LH_MID_WINDOW_MIN) {
            size_t const n = strlen(g_err);
            snprintf(g_err + n, sizeof(g_err) - n, ": fused kernel");
            return 0;
        }
    }
    if (batch_mid_reserve(b, batch_window_records(b, h_descs, w)) != 0)
        return 0;
    p->split = 1;
    p->window = w;
    p->nwin = (p->max_frames + w - 1) / w;
    /* window k of a stream: its frames [begin + k w, begin + (k + 1) w) at the payload's places, their records from the
     * start of the pool on, stream after stream; the flush goes with the stream's last frame (a stream without frames
     * keeps its descriptor in window 0: what an incremental batch's launch may hold) */
    b->h_wdesc.resize((size_t) p->nwin * (size_t) b->B);
    for (int k = 0; k < p->nwin; k++) {
        long long at = 0;
        for (int s = 0; s < b->B; s++) {
            LhStreamDesc d = h_descs[s];
            int const nf = d.frame_end > d.frame_begin ? d.frame_end - d.frame_begin : 0;
            long long const lo = (long long) k * w < nf ? (long long) k * w : nf, hi = lo + w < nf ? lo + w : nf;
            if (nf > 0) {
                d.out_index = h_descs[s].out_index + lo;
                d.frame_begin = h_descs[s].frame_begin + (int) lo;
                d.frame_end = h_descs[s].frame_begin + (int) hi;
                d.flush = h_descs[s].flush && hi == nf && lo < hi;
                d.mid_rel = (int) (at - d.out_index);
                at += hi - lo;
            }
            else if (k > 0)
                d.flush = 0;
            b->h_wdesc[(size_t) k * (size_t) b->B + (size_t) s] = d;
        }
    }
    if ((long long) b->h_wdesc.size() > b->wdesc_cap) {
        LhStreamDesc *bigger = nullptr;
        HIPCHK(hipMalloc((void **) &bigger, b->h_wdesc.size() * sizeof(LhStreamDesc)));
        if (b->d_wdesc)
            (void) hipFree(b->d_wdesc);
        b->d_wdesc = bigger;
        b->wdesc_cap = (long long) b->h_wdesc.size();
    }
    HIPCHK(hipMemcpyAsync(b->d_wdesc, b->h_wdesc.data(), b->h_wdesc.size() * sizeof(LhStreamDesc), hipMemcpyHostToDevice, b->stream));
    while (b->ev_win.size() < 3 * (size_t) p->nwin) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess)
            return set_err("hipEventCreate", hipGetLastError());
        b->ev_win.push_back(e);
    }
    return 0;
}

/* one launch of the batch's frames [frame_begin, frame_end) per stream, as `descs' (device copy) / `h_descs' say, between
 * ev0 and ev1 on the batch's stream, the way batch_plan decided */
static int
batch_launch(lamehip_batch * b, const int16_t * pcm, const float *pcmf, const LhStreamDesc * descs, const LhLaunchPlan & p, uint8_t * bytes)
{
    int     rc = 0, split = p.split;
    if (split && !p.window && !b->ev_part[0]) {
        if (hipEventCreate(&b->ev_part[0]) != hipSuccess || hipEventCreate(&b->ev_part[1]) != hipSuccess)
            return set_err("hipEventCreate", hipGetLastError());
    }
    HIPCHK(hipEventRecord(b->ev0, b->stream));
    if (split && p.window) {
        for (int k = 0; k < p.nwin && !rc; k++) {
            int const left = p.max_frames - k * p.window;
            HIPCHK(hipEventRecord(b->ev_win[3 * (size_t) k], b->stream));
            rc = b->dc.launch_split(pcm, pcmf, b->d_wdesc + (size_t) k * (size_t) b->B, b->d_state, b->d_out, bytes, b->B,
                                    left < p.window ? left : p.window, b->mid, (void *) b->stream, &b->ev_win[3 * (size_t) k + 1]);
        }
    }
    else if (split)
        rc = b->dc.launch_split(pcm, pcmf, descs, b->d_state, b->d_out, bytes, b->B, p.max_frames, b->mid, (void *) b->stream, b->ev_part);
    else
        rc = b->dc.launch(pcm, pcmf, descs, b->d_state, b->d_out, bytes, b->B, (void *) b->stream);
    if (rc)
        return set_err("kernel launch", (hipError_t) rc);
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    if (!b->ev_wait && hipEventCreateWithFlags(&b->ev_wait, hipEventDisableTiming) != hipSuccess) {
        (void) hipGetLastError();
        b->ev_wait = nullptr;
    }
    if (b->ev_wait)
        HIPCHK(hipEventRecord(b->ev_wait, b->stream));
    b->last_split = split;
    b->last_windows = (split && p.window) ? p.nwin : 1;
    return 0;
}

static int
batch_padding(const lamehip_batch * b, int s)
{
    return b->rate_in ? b->padding[(size_t) s] : lh_end_padding_fs(b->len[(size_t) s], fs_of(b->cfg));
}

/* every stream back to its initial state: a device-to-device copy of the pristine image on the batch's own
 * stream (a host copy on the null stream would wait for every other batch's work in flight) */
static int
batch_reset_states(lamehip_batch * b)
{
    if (!b->d_state0) {
        std::vector < LhStreamState > s((size_t) b->B);
        for (int i = 0; i < b->B; i++)
            lh_state_init(&s[(size_t) i], &b->cfg);
        HIPCHK(hipMalloc((void **) &b->d_state0, s.size() * sizeof(LhStreamState)));
        HIPCHK(hipMemcpy(b->d_state0, s.data(), s.size() * sizeof(LhStreamState), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpyAsync(b->d_state, b->d_state0, (size_t) b->B * sizeof(LhStreamState), hipMemcpyDeviceToDevice, b->stream));
    b->encoded = 0;
    return 0;
}

/* the handle's device for its own launches; call before lame_init_params (reference: none -- the
 * reference has no devices; the frontend's handle simply lives on the current device by default) */
extern "C" int
lamehip_set_device(lame_t g, int device)
{
    if (!valid(g) || g->inited || device < 0)
        return -1;
    g->device = device;
    return 0;
}

extern "C" lamehip_batch *
lamehip_batch_create(const lame_t proto, int nstreams, long capacity_samples)
{
    int     dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        dev = 0;
    return lamehip_batch_create_on(dev, proto, nstreams, capacity_samples);
}

/* a batch on HIP device `device' (its pools, state, stream and launches live there whatever device
 * is current in the calling thread); the handle only provides the settings and may belong to
 * another device */
extern "C" lamehip_batch *
lamehip_batch_create_on(int device, const lame_t proto, int nstreams, long capacity_samples)
{
    lamehip_batch *b;
    if (device < 0 || device >= lamehip_device_count()) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_create_on: no HIP device %d", device);
        return nullptr;
    }
    LhDeviceScope const on_device(device);
    if (!valid(proto) || !proto->inited || !proto->have_device || nstreams <= 0
        || capacity_samples <= 0) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_create: need an initialised handle on a HIP device");
        return nullptr;
    }
    b = new(std::nothrow) lamehip_batch();
    if (!b)
        return nullptr;
    b->device = device;
    b->cfg = proto->cfg;
    b->tab = (LhTables *) malloc(sizeof(LhTables));
    if (!b->tab) {
        delete  b;
        return nullptr;
    }
    memcpy(b->tab, proto->tab, sizeof(LhTables));
    b->B = nstreams;
    b->cap = capacity_samples;
    b->d_pcm = nullptr;
    b->d_state = nullptr;
    b->d_state0 = nullptr;
    b->d_desc = nullptr;
    b->d_out = nullptr;
    b->out_cap = 0;
    b->len.assign((size_t) nstreams, 0);
    b->nframes.assign((size_t) nstreams, 0);
    b->out_off.assign((size_t) nstreams, 0);
    b->h_desc.resize((size_t) nstreams);
    b->last_ms = 0;
    b->encoded = 0;
    b->dev_pack = 0;
    b->d_bytes = nullptr;
    b->bytes_cap = 0;
    b->bytes_off.assign((size_t) nstreams, 0);
    b->rate_in = 0;
    b->rs = nullptr;
    b->d_pcmf = nullptr;
    b->capf = 0;
    b->padding.assign((size_t) nstreams, 0);
    b->incremental = 0;
    b->h_stage = b->d_stage = nullptr;
    b->stage_arena_at = b->stage_cap = b->stage_used = 0;
    b->stage_nseg = 0;
    b->finished = 0;
    b->h_pcm = nullptr;
    b->row_dirty.assign((size_t) nstreams, 0);
    b->n_dirty = 0;
    b->h_bytes = nullptr;
    b->h_bytes_cap = 0;
    b->d_sum = b->h_sum = nullptr;
    b->fetched = 0;
    b->up_stream = b->down_stream = nullptr;
    b->ev_up = b->ev_sum = b->ev_down = nullptr;
    b->up_pending = b->down_pending = 0;
    b->up_inflight = 0;
    b->launched = 0;
    b->mid.frames = nullptr;
    b->mid_cap = 0;
    b->split = batch_use_split();
    b->ev_part[0] = b->ev_part[1] = nullptr;
    b->d_wdesc = nullptr;
    b->wdesc_cap = 0;
    b->last_windows = 1;
    b->ev_wait = nullptr;
    b->part_ms[0] = b->part_ms[1] = b->part_ms[2] = 0;
    b->last_split = 0;
    if (proto->rs) {
        /* the s16 pool shrinks to nothing, the converted signal (plus the flush's tail) lives in a float pool */
        b->rate_in = proto->p.samplerate;
        b->rs = (LhResampler *) malloc(sizeof(LhResampler));
        b->capf = (long) ((double) capacity_samples / proto->rs->ratio) + 4 * 1152 + 64;
        capacity_samples = 1;
        if (!b->rs || hipMalloc((void **) &b->d_pcmf, (size_t) nstreams * 2 * (size_t) b->capf * sizeof(float)) != hipSuccess) {
            snprintf(g_err, sizeof(g_err), "lamehip_batch_create: device allocation failed");
            lamehip_batch_destroy(b);
            return nullptr;
        }
    }
    if (b->dc.upload(b->cfg, *b->tab) != 0
        || hipMalloc((void **) &b->d_pcm, (size_t) nstreams * 2 * (size_t) capacity_samples * 2) != hipSuccess
        || hipMalloc((void **) &b->d_state, (size_t) nstreams * sizeof(LhStreamState)) != hipSuccess
        || hipMalloc((void **) &b->d_desc, (size_t) nstreams * sizeof(LhStreamDesc)) != hipSuccess
        || hipStreamCreate(&b->stream) != hipSuccess
        || hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess
        || hipMemset(b->d_pcm, 0, (size_t) nstreams * 2 * (size_t) capacity_samples * 2) != hipSuccess
        || batch_reset_states(b) != 0) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_create: device allocation failed");
        lamehip_batch_destroy(b);
        return nullptr;
    }
    return b;
}

extern "C" void
lamehip_batch_destroy(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    if (!b)
        return;
    if (b->d_pcm)
        (void) hipFree(b->d_pcm);
    if (b->d_state)
        (void) hipFree(b->d_state);
    if (b->d_state0)
        (void) hipFree(b->d_state0);
    if (b->d_desc)
        (void) hipFree(b->d_desc);
    if (b->d_out)
        (void) hipFree(b->d_out);
    if (b->d_bytes)
        (void) hipFree(b->d_bytes);
    if (b->d_pcmf)
        (void) hipFree(b->d_pcmf);
    if (b->mid.frames)
        (void) hipFree(b->mid.frames);
    if (b->ev_wait)
        (void) hipEventDestroy(b->ev_wait);
    if (b->ev_part[0])
        (void) hipEventDestroy(b->ev_part[0]);
    if (b->ev_part[1])
        (void) hipEventDestroy(b->ev_part[1]);
    for (hipEvent_t e : b->ev_win)
        (void) hipEventDestroy(e);
    if (b->d_wdesc)
        (void) hipFree(b->d_wdesc);
    free(b->rs);
    if (b->h_stage)
        (void) hipHostFree(b->h_stage);
    if (b->d_stage)
        (void) hipFree(b->d_stage);
    if (b->up_stream)
        (void) hipStreamDestroy(b->up_stream);
    if (b->down_stream)
        (void) hipStreamDestroy(b->down_stream);
    if (b->ev_up)
        (void) hipEventDestroy(b->ev_up);
    if (b->ev_sum)
        (void) hipEventDestroy(b->ev_sum);
    if (b->ev_down)
        (void) hipEventDestroy(b->ev_down);
    if (b->h_pcm)
        (void) hipHostFree(b->h_pcm);
    if (b->h_bytes)
        (void) hipHostFree(b->h_bytes);
    if (b->h_sum)
        (void) hipHostFree(b->h_sum);
    if (b->d_sum)
        (void) hipFree(b->d_sum);
    for (size_t i = 0; i < b->packer.size(); i++)
        lh_bs_free(&b->packer[i]);
    if (b->stream)
        (void) hipStreamDestroy(b->stream);
    if (b->ev0)
        (void) hipEventDestroy(b->ev0);
    if (b->ev1)
        (void) hipEventDestroy(b->ev1);
    b->dc.release();
    free(b->tab);
    delete  b;
}

extern "C" int
lamehip_batch_set_length(lamehip_batch * b, int s, long n)
{
    if (!b || s < 0 || s >= b->B || n < 0 || n > b->cap || b->rate_in)
        return -1;              /* (a converting batch needs the samples themselves: lamehip_batch_set_pcm) */
    b->len[(size_t) s] = n;
    b->nframes[(size_t) s] = lh_total_frames_fs(n, fs_of(b->cfg));
    return 0;
}

/* A stream of a converting batch: what the reference makes of it when its frontend feeds
 * lame_encode_buffer 1152 input samples at a time and then flushes (lame.c:1708-1772, 2075-2120;
 * the same bookkeeping as the handle path above, without a device in the loop). */
static int
batch_convert_stream(lamehip_batch * b, int s, const short *l, const short *r, long n)
{
    std::vector < float >ol, orr;
    float   il[1152], ir[1152], blk[2][1152];
    double const ratio = (double) b->rate_in / (double) b->cfg.samplerate;
    long    fed = 0, mf_size = LH_MF_START;
    int     frames = 0, owed, padding, frames_left;
    float const m00 = b->cfg.pcm_scale, m01 = b->cfg.pcm_mix, m10 = 0.0f * b->cfg.pcm_scale, m11 = b->cfg.pcm_scale_r;
    auto    feed =[&](int m) {
        int     at = 0;
        while (m > 0) {
            int     used = 0, made = 0;
            for (int ch = 0; ch < b->cfg.channels; ch++)
                made = lh_rs_block(b->rs, ch, blk[ch], fs_of(b->cfg), (ch ? ir : il) + at, m, &used);
            if (b->cfg.channels == 1)
                memset(blk[1], 0, sizeof(blk[1]));
            ol.insert(ol.end(), blk[0], blk[0] + made);
            orr.insert(orr.end(), blk[1], blk[1] + made);
            fed += made;
            mf_size += made;
            if (mf_size >= mfn_of(b->cfg)) {
                frames++;
                mf_size -= fs_of(b->cfg);
            }
            at += used;
            m -= used;
        }
    };
    lh_rs_init(b->rs, b->rate_in, b->cfg.samplerate);
    ol.reserve((size_t) ((double) n / ratio) + 4096);
    orr.reserve((size_t) ((double) n / ratio) + 4096);
    for (long pos = 0; pos < n; pos += fs_of(b->cfg)) {
        int const m = (n - pos) > fs_of(b->cfg) ? fs_of(b->cfg) : (int) (n - pos);
        for (int i = 0; i < m; i++) {
            float const xl = (float) l[pos + i], xr = (float) r[pos + i];
            il[i] = xl * m00 + xr * m01;
            ir[i] = xl * m10 + xr * m11;
        }
        feed(m);
    }
    owed = (int) (576 + fed - (long) fs_of(b->cfg) * frames);
    owed += 16. / ratio;
    padding = fs_of(b->cfg) - (owed % fs_of(b->cfg));
    if (padding < 576)
        padding += fs_of(b->cfg);
    frames_left = (owed + padding) / fs_of(b->cfg);
    memset(il, 0, sizeof(il));
    memset(ir, 0, sizeof(ir));
    while (frames_left > 0) {
        int const before = frames;
        int     bunch = (int) (mfn_of(b->cfg) - mf_size);
        bunch *= ratio;
        if (bunch > 1152)
            bunch = 1152;
        if (bunch < 1)
            bunch = 1;
        feed(bunch);
        frames_left -= (frames != before) ? 1 : 0;
    }
    if ((long) ol.size() > b->capf) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_set_pcm: converted stream (%ld samples) exceeds the pool", (long) ol.size());
        return -1;
    }
    b->len[(size_t) s] = (long) ol.size();
    b->nframes[(size_t) s] = frames;
    b->padding[(size_t) s] = padding;
    if (!ol.empty()) {
        HIPCHK(hipMemcpy(b->d_pcmf + ((size_t) s * 2) * (size_t) b->capf, ol.data(), ol.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b->d_pcmf + ((size_t) s * 2 + 1) * (size_t) b->capf, orr.data(), orr.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" short *lamehip_batch_pcm_host_ptr(lamehip_batch * b);

/* the mirror for lamehip_batch_set_pcm: made on first use unless the pool is larger than LAMEHIP_PINNED_MAX_MB
 * (default 4096) of pinned host memory */
static int16_t *
batch_host_pool(lamehip_batch * b)
{
    if (!b->h_pcm && !b->rate_in) {
        const char *e = getenv("LAMEHIP_PINNED_MAX_MB");
        double const limit = (e ? atof(e) : 4096.0) * 1048576.0;
        if ((double) b->B * 4.0 * (double) b->cap > limit)
            return nullptr;
        (void) lamehip_batch_pcm_host_ptr(b);
    }
    return b->h_pcm;
}

/* the pinned mirror is about to be rewritten by the host: an upload out of it that is still on its way must have
 * finished (the device-side wait in lamehip_batch_encode says nothing to the host) */
static int
batch_mirror_quiesce(lamehip_batch * b)
{
    if (b->up_inflight) {
        HIPCHK(hipEventSynchronize(b->ev_up));
        b->up_inflight = 0;
    }
    return 0;
}

extern "C" int
lamehip_batch_set_pcm(lamehip_batch * b, int s, const short *l, const short *r, long n)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    if (b && b->rate_in) {
        if (s < 0 || s >= b->B || n < 0)
            return -1;
        if (b->cfg.channels == 1 && b->cfg.pcm_mix == 0.0f)
            r = l;
        return batch_convert_stream(b, s, l, r, n);
    }
    if (lamehip_batch_set_length(b, s, n) != 0)
        return -1;
    if (b->cfg.channels == 1 && b->cfg.pcm_mix == 0.0f)
        r = l;                  /* mono: the second plane mirrors the first, the kernel never uses it */
    if (batch_host_pool(b) != nullptr) {
        if (batch_mirror_quiesce(b) != 0)
            return LAMEHIP_ERR_DEVICE;
        /* into the pinned mirror; the rows travel with the next lamehip_batch_upload / _encode, all streams'
         * in one asynchronous copy (the reference's seam: lame_encode_buffer copies into mfbuf, lame.c:1672) */
        memcpy(b->h_pcm + ((size_t) s * 2) * (size_t) b->cap, l, (size_t) n * 2);
        memcpy(b->h_pcm + ((size_t) s * 2 + 1) * (size_t) b->cap, r, (size_t) n * 2);
        if (!b->row_dirty[(size_t) s]) {
            b->row_dirty[(size_t) s] = 1;
            b->n_dirty++;
        }
        return 0;
    }
    /* the pool is too large to mirror in pinned memory: straight to HBM, stream by stream */
    HIPCHK(hipMemcpy(b->d_pcm + ((size_t) s * 2) * (size_t) b->cap, l, (size_t) n * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->d_pcm + ((size_t) s * 2 + 1) * (size_t) b->cap, r, (size_t) n * 2, hipMemcpyHostToDevice));
    return 0;
}

/* The pinned mirror of the batch's s16 pool, [stream][2][capacity] like the pool itself: the caller may
 * decode straight into it (then lamehip_batch_set_length + lamehip_batch_mark_pcm, or lamehip_batch_set_pcm,
 * which copies into it).  NULL for a converting batch or when the mirror cannot be had. */
extern "C" short *
lamehip_batch_pcm_host_ptr(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    if (!b)
        return nullptr;
    if (!b->h_pcm && !b->rate_in
        && hipHostMalloc((void **) &b->h_pcm, (size_t) b->B * 2 * (size_t) b->cap * 2, 0) != hipSuccess) {
        b->h_pcm = nullptr;
        (void) hipGetLastError();
    }
    /* whoever asks for the pointer is about to write through it: no upload may still be reading the mirror.  (A caller
     * that keeps the pointer across rounds asks again -- or calls lamehip_batch_set_pcm -- before it rewrites rows that
     * an asynchronous lamehip_batch_upload / _encode has taken.) */
    if (b->h_pcm && batch_mirror_quiesce(b) != 0)
        return nullptr;
    return b->h_pcm;
}

/* stream s's rows of the mirror were written by the caller: they travel with the next upload */
extern "C" int
lamehip_batch_mark_pcm(lamehip_batch * b, int s)
{
    if (!b || s < 0 || s >= b->B || !b->h_pcm)
        return -1;
    if (!b->row_dirty[(size_t) s]) {
        b->row_dirty[(size_t) s] = 1;
        b->n_dirty++;
    }
    return 0;
}

/* Asynchronous H2D of everything that changed in the mirror, on the batch's stream (lamehip_batch_encode does
 * this itself when something is pending).  One copy when every stream changed and the streams fill their rows,
 * else one per row. */
static int
batch_copy_streams(lamehip_batch * b)
{
    if (!b->up_stream) {
        HIPCHK(hipStreamCreateWithFlags(&b->up_stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&b->down_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->ev_sum, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->ev_down, hipEventDisableTiming));
    }
    return 0;
}

extern "C" int
lamehip_batch_upload(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    if (!b)
        return -1;
    if (b->n_dirty == 0 || !b->h_pcm)
        return 0;
    if (batch_copy_streams(b) != 0)
        return LAMEHIP_ERR_DEVICE;
    /* the pool may still be read by the kernel of the previous round (also after a reset, which clears `encoded') */
    if (b->launched)
        HIPCHK(hipStreamWaitEvent(b->up_stream, b->ev1, 0));
    {
        long long used = 0;
        for (int s = 0; s < b->B; s++)
            used += b->len[(size_t) s];
        if (b->n_dirty == b->B && used * 10 >= (long long) b->B * b->cap * 9)
            HIPCHK(hipMemcpyAsync(b->d_pcm, b->h_pcm, (size_t) b->B * 2 * (size_t) b->cap * 2, hipMemcpyHostToDevice, b->up_stream));
        else
            for (int s = 0; s < b->B; s++) {
                size_t const n = (size_t) b->len[(size_t) s] * 2;
                if (!b->row_dirty[(size_t) s] || n == 0)
                    continue;
                for (int ch = 0; ch < 2; ch++) {
                    size_t const at = ((size_t) s * 2 + (size_t) ch) * (size_t) b->cap;
                    HIPCHK(hipMemcpyAsync(b->d_pcm + at, b->h_pcm + at, n, hipMemcpyHostToDevice, b->up_stream));
                }
            }
    }
    HIPCHK(hipEventRecord(b->ev_up, b->up_stream));
    b->up_pending = 1;
    b->up_inflight = 1;
    b->row_dirty.assign((size_t) b->B, 0);
    b->n_dirty = 0;
    return 0;
}

extern "C" int
lamehip_batch_set_pcm_device(lamehip_batch * b, int s, const void *dl, const void *dr, long n)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    if (lamehip_batch_set_length(b, s, n) != 0)
        return -1;
    if (b->cfg.channels == 1 && b->cfg.pcm_mix == 0.0f)
        dr = dl;
    if (b->up_pending || b->up_inflight) {      /* an upload of the mirror is on its way into the same pool */
        HIPCHK(hipStreamSynchronize(b->up_stream));
        b->up_inflight = 0;
    }
    if (b->row_dirty[(size_t) s]) {     /* what the mirror holds for this stream is superseded */
        b->row_dirty[(size_t) s] = 0;
        b->n_dirty--;
    }
    HIPCHK(hipMemcpy(b->d_pcm + ((size_t) s * 2) * (size_t) b->cap, dl, (size_t) n * 2, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(b->d_pcm + ((size_t) s * 2 + 1) * (size_t) b->cap, dr, (size_t) n * 2, hipMemcpyDeviceToDevice));
    return 0;
}

extern "C" void *
lamehip_batch_pcm_device_ptr(lamehip_batch * b)
{
    return (b && !b->rate_in) ? (void *) b->d_pcm : nullptr;
}


/* ---- incremental use of a batch: lame_encode_buffer semantics for many streams at once -------
 * (reference lame.c:1672-1775: every call appends samples to a stream and encodes the frames that
 * became complete; the output lags the input by the priming).  lamehip_batch_append stages a chunk
 * per stream in pinned host memory, lamehip_batch_encode_available moves all staged chunks to HBM
 * with ONE asynchronous copy, encodes every stream's newly complete frames with ONE launch and packs
 * them, lamehip_batch_drain hands a stream's new bytes over -- the bytes lame_encode_buffer would
 * have returned for the same calls --, lamehip_batch_finish is lame_encode_flush for all streams. */
static int
batch_incremental_begin(lamehip_batch * b)
{
    if (b->incremental)
        return 0;
    if (b->rate_in || b->dev_pack) {
        snprintf(g_err, sizeof(g_err), "incremental batches take the encoder's own input rate and the host packer");
        return -1;
    }
    if (batch_reset_states(b) != 0)
        return LAMEHIP_ERR_DEVICE;
    b->fed.assign((size_t) b->B, 0);
    b->done.assign((size_t) b->B, 0);
    b->staged.assign((size_t) b->B, 0);
    b->pending.assign((size_t) b->B, std::vector < unsigned char >());
    b->last.resize((size_t) b->B);
    b->have_last.assign((size_t) b->B, 0);
    b->packer.resize((size_t) b->B);
    for (int s = 0; s < b->B; s++)
        if (lh_bs_init_sized(&b->packer[(size_t) s], 65536) != 0)
            return -2;
    b->incremental = 1;
    return 0;
}

#define LH_STAGE_SEGS 4096       /* chunks per trip to HBM */
#define LH_STAGE_SAMPLES (8L << 20)     /* the arena: 8 M samples = 16 MB; what is staged beyond goes to HBM at once */

/* the staging area, made once: descriptors, chunk table, arena */
static int
batch_stage_reserve(lamehip_batch * b)
{
    long const arena_at = ((long) b->B * (long) sizeof(LhStreamDesc) + (long) LH_STAGE_SEGS * 16 + 63) & ~63L;
    long const bytes = arena_at + LH_STAGE_SAMPLES * 2;
    unsigned char *h = nullptr, *d = nullptr;
    if (b->h_stage)
        return 0;
    if (hipHostMalloc((void **) &h, (size_t) bytes, hipHostMallocDefault) != hipSuccess
        || hipMalloc((void **) &d, (size_t) bytes) != hipSuccess) {
        if (h)
            (void) hipHostFree(h);
        return set_err("staging allocation", hipErrorOutOfMemory);
    }
    b->h_stage = h;
    b->d_stage = d;
    b->stage_arena_at = arena_at;
    b->stage_cap = LH_STAGE_SAMPLES;
    b->stage_used = 0;
    b->stage_nseg = 0;
    return 0;
}

/* what is staged goes to the pool: one copy of the chunk table and of the arena's bytes in use, one scatter
 * launch; wait = the arena is free again on return (it is about to be refilled) */
static int
batch_stage_flush(lamehip_batch * b, int wait)
{
    size_t const segs_at = (size_t) b->B * sizeof(LhStreamDesc);
    if (b->stage_nseg > 0) {
        int     rc;
        HIPCHK(hipMemcpyAsync(b->d_stage + segs_at, b->h_stage + segs_at, (size_t) b->stage_nseg * 16, hipMemcpyHostToDevice,
                              b->stream));
        HIPCHK(hipMemcpyAsync(b->d_stage + b->stage_arena_at, b->h_stage + b->stage_arena_at, (size_t) b->stage_used * 2,
                              hipMemcpyHostToDevice, b->stream));
        rc = lh_launch_scatter((const int16_t *) (b->d_stage + b->stage_arena_at), b->d_pcm, b->cap,
                               (const int *) (b->d_stage + segs_at), b->stage_nseg, (void *) b->stream);
        if (rc)
            return set_err("scatter launch", (hipError_t) rc);
        for (int s = 0; s < b->B; s++) {
            b->fed[(size_t) s] += b->staged[(size_t) s];
            b->staged[(size_t) s] = 0;
        }
        b->stage_nseg = 0;
        b->stage_used = 0;
        if (wait)
            HIPCHK(hipStreamSynchronize(b->stream));
    }
    return 0;
}

extern "C" int
lamehip_batch_append(lamehip_batch * b, int s, const short *l, const short *r, int n)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    int     rc;
    if (!b || s < 0 || s >= b->B || n < 0 || (n > 0 && !l))
        return -1;
    if (b->finished) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_append: the batch is finished (lamehip_batch_reset starts it over)");
        return -1;
    }
    if ((rc = batch_incremental_begin(b)) != 0)
        return rc;
    if (n == 0)
        return 0;
    if (b->cfg.channels == 1 && b->cfg.pcm_mix == 0.0f)
        r = l;
    if (!r)
        return -1;
    if (b->fed[(size_t) s] + b->staged[(size_t) s] + n > b->cap) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_append: stream %d would exceed the batch's capacity of %ld samples", s, b->cap);
        return -1;
    }
    if ((rc = batch_stage_reserve(b)) != 0)
        return rc;
    /* pieces of at most half the arena; when the arena or the chunk table is full, what is staged leaves for HBM */
    while (n > 0) {
        int const piece = ((long) n > b->stage_cap / 4) ? (int) (b->stage_cap / 4) : n;
        int    *seg;
        int16_t *arena = (int16_t *) (b->h_stage + b->stage_arena_at);
        if (b->stage_used + 2L * piece > b->stage_cap || b->stage_nseg == LH_STAGE_SEGS)
            if ((rc = batch_stage_flush(b, 1)) != 0)
                return rc;
        seg = (int *) (b->h_stage + (size_t) b->B * sizeof(LhStreamDesc)) + 4 * b->stage_nseg;
        seg[0] = (int) b->stage_used;
        seg[1] = s;
        seg[2] = (int) (b->fed[(size_t) s] + b->staged[(size_t) s]);
        seg[3] = piece;
        memcpy(arena + b->stage_used, l, (size_t) piece * 2);
        memcpy(arena + b->stage_used + piece, r, (size_t) piece * 2);
        b->stage_used += 2L * piece;
        b->stage_nseg++;
        b->staged[(size_t) s] += piece;
        l += piece;
        r += piece;
        n -= piece;
    }
    return 0;
}

/* frames of a stream that are complete once `fed' samples are in: frame f reads 1904 samples from
 * 1152 f - 528 on (reference lame.c:1737-1766) */
static int
frames_complete(long fed, const LhConfig & c)
{
    long const have = fed + LH_MF_START;
    return have >= mfn_of(c) ? (int) ((have - mfn_of(c)) / fs_of(c) + 1) : 0;
}

/* encode frames [done, upto[s]) of every stream and pack them into pending[]; `end' marks the
 * streams' last frames (flush) */
static int
batch_encode_range(lamehip_batch * b, const std::vector < int >&upto, int end)
{
    LhStreamDesc *descs = (LhStreamDesc *) b->h_stage;
    long long total = 0;
    int     rc;
    for (int s = 0; s < b->B; s++) {
        LhStreamDesc & d = descs[s];
        int const nf = upto[(size_t) s] - b->done[(size_t) s];
        memset(&d, 0, sizeof(d));
        d.pcm_l = ((long long) s * 2) * b->cap;
        d.pcm_r = ((long long) s * 2 + 1) * b->cap;
        d.pcm_base = 0;
        d.nsamples = b->fed[(size_t) s] + b->staged[(size_t) s];
        d.out_index = total;
        d.frame_begin = b->done[(size_t) s];
        d.frame_end = upto[(size_t) s];
        d.flush = end;
        total += nf > 0 ? nf : 0;
    }
    if (total > b->out_cap) {
        /* the new buffer first: a failed allocation leaves the old one (and its size) in place */
        LhFrameOut *bigger = nullptr;
        HIPCHK(hipMalloc((void **) &bigger, (size_t) (total + 1024) * sizeof(LhFrameOut)));
        if (b->d_out)
            (void) hipFree(b->d_out);
        b->d_out = bigger;
        b->out_cap = total + 1024;
    }
    /* descriptors, then what is staged (chunk table + the arena's bytes in use), then the scatter */
    HIPCHK(hipMemcpyAsync(b->d_stage, b->h_stage, (size_t) b->B * sizeof(LhStreamDesc), hipMemcpyHostToDevice, b->stream));
    if ((rc = batch_stage_flush(b, 0)) != 0)
        return rc;
    if (total == 0) {
        HIPCHK(hipStreamSynchronize(b->stream));
        return 0;
    }
    {
        LhLaunchPlan plan;
        if ((rc = batch_plan(b, descs, &plan)) != 0
            || (rc = batch_launch(b, b->d_pcm, (const float *) 0, (const LhStreamDesc *) b->d_stage, plan, (uint8_t *) 0)) != 0)
            return rc;
    }
    b->launched = 1;
    b->h_new.resize((size_t) total);
    HIPCHK(hipMemcpyAsync(b->h_new.data(), b->d_out, (size_t) total * sizeof(LhFrameOut), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int s = 0; s < b->B; s++) {
        LhBitstream *bs = &b->packer[(size_t) s];
        std::vector < unsigned char >&out = b->pending[(size_t) s];
        for (int f = b->done[(size_t) s]; f < upto[(size_t) s]; f++) {
            const LhFrameOut & fo = b->h_new[(size_t) (descs[s].out_index + (f - descs[s].frame_begin))];
            size_t  at;
            int     k;
            if (lh_bs_format_frame(bs, &b->cfg, b->tab, &fo) != 0) {
                snprintf(g_err, sizeof(g_err), "inconsistent device payload (packer check %d) stream %d frame %d", bs->error, s, f);
                return LAMEHIP_ERR_PAYLOAD;
            }
            at = out.size();
            out.resize(at + (size_t) lh_bs_pending(bs));
            k = lh_bs_copy(bs, out.data() + at, 0);
            out.resize(at + (size_t) (k > 0 ? k : 0));
            b->last[(size_t) s] = fo;
            b->have_last[(size_t) s] = 1;
        }
        if (upto[(size_t) s] > b->done[(size_t) s])
            b->done[(size_t) s] = upto[(size_t) s];
    }
    return (int) total;
}

extern "C" int
lamehip_batch_encode_available(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    std::vector < int >upto;
    int     rc;
    if (!b)
        return -1;
    if (b->finished) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_encode_available: the batch is finished (lamehip_batch_reset starts it over)");
        return -1;
    }
    if ((rc = batch_incremental_begin(b)) != 0)
        return rc;
    if ((rc = batch_stage_reserve(b)) != 0)
        return rc;
    upto.resize((size_t) b->B);
    for (int s = 0; s < b->B; s++)
        upto[(size_t) s] = frames_complete(b->fed[(size_t) s] + b->staged[(size_t) s], b->cfg);
    return batch_encode_range(b, upto, 0);
}

/* lame_encode_flush for every stream of an incremental batch: the frames still owed for the samples
 * fed (with the reference's end padding), then the stuffing that completes the last frame */
extern "C" int
lamehip_batch_finish(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    std::vector < int >upto;
    int     rc, n;
    if (!b)
        return -1;
    if (b->finished) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_finish: the batch is finished already (lamehip_batch_reset starts it over)");
        return -1;
    }
    if ((rc = batch_incremental_begin(b)) != 0)
        return rc;
    if ((rc = batch_stage_reserve(b)) != 0)
        return rc;
    upto.resize((size_t) b->B);
    for (int s = 0; s < b->B; s++) {
        long const len = b->fed[(size_t) s] + b->staged[(size_t) s];
        b->len[(size_t) s] = len;
        b->nframes[(size_t) s] = lh_total_frames_fs(len, fs_of(b->cfg));
        upto[(size_t) s] = b->nframes[(size_t) s];
    }
    n = batch_encode_range(b, upto, 1);
    if (n < 0)
        return n;
    for (int s = 0; s < b->B; s++) {
        LhBitstream *bs = &b->packer[(size_t) s];
        std::vector < unsigned char >&out = b->pending[(size_t) s];
        size_t  at = out.size();
        int     k;
        lh_bs_flush(bs, &b->cfg, b->have_last[(size_t) s] ? &b->last[(size_t) s] : nullptr);
        out.resize(at + (size_t) lh_bs_pending(bs));
        k = lh_bs_copy(bs, out.data() + at, 0);
        out.resize(at + (size_t) (k > 0 ? k : 0));
    }
    b->finished = 1;
    return n;
}

/* bytes of stream s produced since the last drain; -1 (and nothing taken) when they do not fit */
extern "C" int
lamehip_batch_drain(lamehip_batch * b, int s, unsigned char *out, int cap)
{
    int     n;
    if (!b || !b->incremental || s < 0 || s >= b->B)
        return -1;
    n = (int) b->pending[(size_t) s].size();
    if (n == 0)
        return 0;
    if (!out || cap < n)
        return -1;
    memcpy(out, b->pending[(size_t) s].data(), (size_t) n);
    b->pending[(size_t) s].clear();
    return n;
}

extern "C" int
lamehip_batch_frames(lamehip_batch * b, int s)
{
    if (!b || s < 0 || s >= b->B)
        return -1;
    return b->nframes[(size_t) s];
}

extern "C" int
lamehip_batch_reset(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    if (!b)
        return -1;
    if (b->incremental) {
        /* an incremental batch starts over: nothing fed, nothing staged, fresh packers, nothing left to drain */
        for (int s = 0; s < b->B; s++) {
            lh_bs_free(&b->packer[(size_t) s]);
            if (lh_bs_init_sized(&b->packer[(size_t) s], 65536) != 0)
                return -2;
            b->pending[(size_t) s].clear();
        }
        b->fed.assign((size_t) b->B, 0);
        b->done.assign((size_t) b->B, 0);
        b->staged.assign((size_t) b->B, 0);
        b->have_last.assign((size_t) b->B, 0);
        b->stage_used = 0;
        b->stage_nseg = 0;
        b->finished = 0;
    }
    return batch_reset_states(b);
}

extern "C" int
lamehip_batch_encode(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    long long total = 0, bytes_total = 0;
    int     max_frame_bytes;
    if (!b)
        return -1;
    if (b->incremental) {
        snprintf(g_err, sizeof(g_err), "lamehip_batch_encode: this batch is fed with lamehip_batch_append (incremental use)");
        return -1;
    }
    /* a batch that has been encoded starts over: every call encodes the streams from their first
     * sample, so the carried state must be the initial one */
    if (b->encoded && batch_reset_states(b) != 0)
        return LAMEHIP_ERR_DEVICE;
    {
        int const top = (b->cfg.vbr == 0) ? b->cfg.bitrate_index : b->cfg.vbr_max_bitrate_index;
        /* (the row of the stream's MPEG version: an MPEG-2 / 2.5 index stands for half the MPEG-1 rate or less) */
        max_frame_bytes = (b->cfg.version + 1) * 72000 * lh_tag_kbps(b->cfg.version, top & 15) / b->cfg.samplerate + 1;
    }
    for (int s = 0; s < b->B; s++) {
        LhStreamDesc & d = b->h_desc[(size_t) s];
        b->out_off[(size_t) s] = total;
        d.pcm_l = ((long long) s * 2) * (b->rate_in ? b->capf : b->cap);
        d.pcm_r = ((long long) s * 2 + 1) * (b->rate_in ? b->capf : b->cap);
        d.pcm_base = 0;
        d.nsamples = b->len[(size_t) s];
        d.out_index = total;
        d.frame_begin = 0;
        d.frame_end = b->nframes[(size_t) s];
        d.flush = 1;
        d.mid_rel = 0;
        d.bytes_base = bytes_total;
        /* room for every frame at the largest frame size the settings allow (+1 for CBR padding) */
        d.bytes_cap = b->dev_pack ? (long long) b->nframes[(size_t) s] * max_frame_bytes : 0;
        b->bytes_off[(size_t) s] = bytes_total;
        bytes_total += d.bytes_cap;
        total += b->nframes[(size_t) s];
    }
    /* (a new buffer first: a failed allocation leaves the old one and its size in place) */
    if (b->dev_pack && bytes_total > b->bytes_cap) {
        uint8_t *bigger = nullptr;
        HIPCHK(hipMalloc((void **) &bigger, (size_t) bytes_total));
        if (b->d_bytes)
            (void) hipFree(b->d_bytes);
        b->d_bytes = bigger;
        b->bytes_cap = bytes_total;
    }
    if (total > b->out_cap) {
        LhFrameOut *bigger = nullptr;
        HIPCHK(hipMalloc((void **) &bigger, (size_t) total * sizeof(LhFrameOut)));
        if (b->d_out)
            (void) hipFree(b->d_out);
        b->d_out = bigger;
        b->out_cap = total;
    }
    if (b->n_dirty && lamehip_batch_upload(b) != 0)
        return LAMEHIP_ERR_DEVICE;
    if (b->up_pending) {
        HIPCHK(hipStreamWaitEvent(b->stream, b->ev_up, 0));
        b->up_pending = 0;
    }
    if (b->down_pending) {      /* the previous round's bytes must have left d_bytes */
        HIPCHK(hipStreamWaitEvent(b->stream, b->ev_down, 0));
        b->down_pending = 0;
    }
    HIPCHK(hipMemcpyAsync(b->d_desc, b->h_desc.data(), (size_t) b->B * sizeof(LhStreamDesc),
                          hipMemcpyHostToDevice, b->stream));
    /* (what the launch will be -- and the pool it needs -- before the device's launch order is taken: an allocation of tens
     * of GB must not keep other batches' launches waiting) */
    LhLaunchPlan plan;
    {
        int const rc = batch_plan(b, b->h_desc.data(), &plan);
        if (rc)
            return rc;
    }
    {
        /* Launches that fill the device run one after the other, in launch order, whatever HIP streams their batches
         * own: a launch of >= 512 streams keeps every SIMD's register file and every CU's LDS (2 x 256 VGPRs, 4 x 40 KB),
         * so a second one has nothing to gain from being dispatched early -- and measured on the MI355X it loses: dispatched
         * while the first still runs, its workgroups end up resident in two rounds (kernel 176 ms alone, 329 ms behind
         * another launch, every stream's own cycle count unchanged; tools/e2e_diag2.py), which cost the two-batch pipeline a
         * third of its rate.  Copies on the batches' copy streams overlap the kernels as before. */
        LhLaunchSerial & ser = launch_serial(b->device);
        std::lock_guard < std::mutex > hold(ser.lock);
        int const big = (b->B >= 512);
        if (big && ser.ev)
            HIPCHK(hipStreamWaitEvent(b->stream, ser.ev, 0));
        int     rc = batch_launch(b, b->rate_in ? (const int16_t *) 0 : b->d_pcm, b->rate_in ? b->d_pcmf : (const float *) 0,
                                  b->d_desc, plan, b->dev_pack ? b->d_bytes : (uint8_t *) 0);
        if (rc)
            return rc;
        if (b->dev_pack) {
            /* the two words per stream lamehip_batch_fetch copies first: gathered here, inside the serial order -- as a
             * launch of its own behind the NEXT batch's kernel it found no free register file until that kernel was over
             * (all of a launch's streams end within a frame or two of each other), and its batch came home one kernel late */
            if (batch_copy_streams(b) != 0)
                return LAMEHIP_ERR_DEVICE;
            if (!b->d_sum)
                HIPCHK(hipMalloc((void **) &b->d_sum, (size_t) b->B * 2 * sizeof(long long)));
            rc = lh_launch_summary(b->d_state, b->d_sum, b->B, (void *) b->stream);
            if (rc)
                return set_err("summary launch", (hipError_t) rc);
            HIPCHK(hipEventRecord(b->ev_sum, b->stream));
        }
        if (big) {
            if (!ser.ev)
                HIPCHK(hipEventCreateWithFlags(&ser.ev, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ser.ev, b->stream));
        }
    }
    b->launched = 1;
    b->encoded = 1;
    b->fetched = 0;
    return 0;
}

/* Device-packed batches: start the way back -- per stream (bytes, status), then the bytes themselves, into
 * pinned host memory, asynchronously on the batch's stream (behind the kernel).  lamehip_batch_bytes_ptr /
 * lamehip_batch_get_bytes_all wait for it.  Calling it right after lamehip_batch_encode lets the copies of this
 * batch run under the next batch's kernel. */
extern "C" int
lamehip_batch_fetch(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    long long total = 0;
    if (!b || !b->encoded || !b->dev_pack)
        return -1;
    if (b->fetched)
        return 0;
    for (int s = 0; s < b->B; s++)
        total += b->h_desc[(size_t) s].bytes_cap;
    if (!b->d_sum)
        HIPCHK(hipMalloc((void **) &b->d_sum, (size_t) b->B * 2 * sizeof(long long)));
    if (!b->h_sum)
        HIPCHK(hipHostMalloc((void **) &b->h_sum, (size_t) b->B * 2 * sizeof(long long), 0));
    if (total > b->h_bytes_cap) {
        unsigned char *nb = nullptr;
        HIPCHK(hipHostMalloc((void **) &nb, (size_t) (total > 0 ? total : 1), 0));
        if (b->h_bytes)
            (void) hipHostFree(b->h_bytes);
        b->h_bytes = nb;
        b->h_bytes_cap = total;
    }
    if (batch_copy_streams(b) != 0)
        return LAMEHIP_ERR_DEVICE;
    /* (the summary words were gathered right behind the kernel: lamehip_batch_encode) */
    HIPCHK(hipStreamWaitEvent(b->down_stream, b->ev_sum, 0));
    HIPCHK(hipMemcpyAsync(b->h_sum, b->d_sum, (size_t) b->B * 2 * sizeof(long long), hipMemcpyDeviceToHost, b->down_stream));
    if (total > 0)
        HIPCHK(hipMemcpyAsync(b->h_bytes, b->d_bytes, (size_t) total, hipMemcpyDeviceToHost, b->down_stream));
    HIPCHK(hipEventRecord(b->ev_down, b->down_stream));
    b->down_pending = 1;
    b->fetched = 1;
    return 0;
}

/* stream s's finished bytes in the batch's pinned buffer (valid until the batch is encoded again): returns
 * their number, or a negative code */
extern "C" long
lamehip_batch_bytes_ptr(lamehip_batch * b, int s, const unsigned char **p)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    long    n;
    if (!b || s < 0 || s >= b->B || !b->encoded || !b->dev_pack || !p)
        return -1;
    if (!b->fetched && lamehip_batch_fetch(b) != 0)
        return LAMEHIP_ERR_DEVICE;
    HIPCHK(hipStreamSynchronize(b->down_stream));
    if (b->h_sum[2 * s + 1] != 0) {
        snprintf(g_err, sizeof(g_err), "device bit packer reported status %d for stream %d", (int) b->h_sum[2 * s + 1], s);
        return LAMEHIP_ERR_PAYLOAD;
    }
    n = (b->nframes[(size_t) s] == 0) ? 0 : (long) b->h_sum[2 * s];
    *p = b->h_bytes + b->bytes_off[(size_t) s];
    return n;
}

/* Device bit packing: the kernel also assembles each stream's finished MP3 bytes in HBM
 * (lh_dev_emit.h); lamehip_batch_get_bytes then copies them out, no host packer involved.
 * Set before lamehip_batch_encode. */
extern "C" int
lamehip_batch_set_device_packing(lamehip_batch * b, int on)
{
    if (!b)
        return -1;
    b->dev_pack = on != 0;
    return 0;
}

/* bytes of one stream as the device packed them (audio frames incl. the final padding, no tag) */
extern "C" long
lamehip_batch_get_bytes(lamehip_batch * b, int s, unsigned char *out, long out_size)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    LhStreamState st;
    long    n;
    if (!b || s < 0 || s >= b->B || !b->encoded || !b->dev_pack)
        return -1;
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(&st, b->d_state + s, sizeof(st), hipMemcpyDeviceToHost));
    if (st.status != 0) {
        snprintf(g_err, sizeof(g_err), "device bit packer reported status %d for stream %d", st.status, s);
        return LAMEHIP_ERR_PAYLOAD;
    }
    n = (long) st.em_next_header;
    if (b->nframes[(size_t) s] == 0)
        n = 0;
    if (n > out_size)
        return -1;
    if (n > 0)
        HIPCHK(hipMemcpy(out, b->d_bytes + b->bytes_off[(size_t) s], (size_t) n, hipMemcpyDeviceToHost));
    return n;
}

/* device-packed stream as a complete file image: the final Xing/Info + LAME tag frame, then the audio
 * frames.  The tag's bookkeeping (frame count, bitrate table of contents, music CRC, mode extension of
 * the last frame) is read back from the frame headers of the bytes themselves. */
extern "C" long
lamehip_batch_get_bytes_tagged(lamehip_batch * b, int s, unsigned char *out, long out_size)
{
    LhVbrTag v;
    int     total, last_mode_ext = 0;
    long    k, pos = 0;
    if (!b || s < 0 || s >= b->B || !b->encoded || !b->dev_pack)
        return -1;
    total = lh_tag_init(&v, &b->cfg);
    if (b->rate_in)
        v.samplerate_in = b->rate_in;
    if (out_size < total)
        return -1;
    k = lamehip_batch_get_bytes(b, s, out + total, out_size - total);
    if (k < 0 || total == 0)
        return k;
    while (pos + 4 <= k) {
        const unsigned char *h = out + total + pos;
        int const bi = h[2] >> 4, pad = (h[2] >> 1) & 1;
        int const kbps = lh_tag_kbps(b->cfg.version, bi);
        int const size = (b->cfg.version + 1) * 72000 * kbps / b->cfg.samplerate + pad;
        if (h[0] != 0xff || (h[1] & 0xe0) != 0xe0 || kbps <= 0 || size <= 0) {
            snprintf(g_err, sizeof(g_err), "device-packed stream %d: lost frame sync at byte %ld", s, pos);
            return LAMEHIP_ERR_PAYLOAD;
        }
        lh_tag_add_frame(&v, kbps);
        last_mode_ext = (h[3] >> 4) & 3;
        pos += size;
    }
    lh_tag_crc(&v, out + total, k);
    if (lh_tag_frame(&v, &b->cfg, b->cfg.vbr_q, batch_padding(b, s), last_mode_ext, out, total) != total)
        memset(out, 0, (size_t) total);         /* no frames: the reference leaves the placeholder */
    return k + total;
}

/* all streams: stream s at out + s * out_stride, sizes[s] = bytes or a negative code */
extern "C" int
lamehip_batch_get_bytes_all(lamehip_batch * b, unsigned char *out, long out_stride, long *sizes)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    int     bad = 0;
    if (!b || !b->encoded || !b->dev_pack || !out || !sizes)
        return -1;
    if (lamehip_batch_fetch(b) != 0)
        return LAMEHIP_ERR_DEVICE;
    HIPCHK(hipStreamSynchronize(b->down_stream));
    for (int s = 0; s < b->B; s++) {
        long    n = (b->nframes[(size_t) s] == 0) ? 0 : (long) b->h_sum[2 * s];
        if (b->h_sum[2 * s + 1] != 0)
            n = LAMEHIP_ERR_PAYLOAD;
        else if (n > out_stride)
            n = -1;
        sizes[s] = n;
        if (n < 0)
            bad++;
        else if (n > 0)
            memcpy(out + (size_t) s * (size_t) out_stride, b->h_bytes + b->bytes_off[(size_t) s], (size_t) n);
    }
    return bad ? -1 : 0;
}

/* The buffers a launch of the batch's present streams needs -- payload, the analysis kernels' pool (split pipeline), the byte
 * pool of the device packer -- allocated now instead of by the first lamehip_batch_encode (81 GB of pool at 1024 x 60 s: a
 * caller that times its first launch calls this first). */
extern "C" int
lamehip_batch_reserve(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    long long total = 0, bytes_total = 0;
    int     max_frame_bytes;
    if (!b)
        return -1;
    if (b->incremental)
        return 0;               /* (incremental batches launch what has arrived: sized per launch) */
    {
        int const top = (b->cfg.vbr == 0) ? b->cfg.bitrate_index : b->cfg.vbr_max_bitrate_index;
        max_frame_bytes = (b->cfg.version + 1) * 72000 * lh_tag_kbps(b->cfg.version, top & 15) / b->cfg.samplerate + 1;
    }
    for (int s = 0; s < b->B; s++) {
        total += b->nframes[(size_t) s];
        bytes_total += b->dev_pack ? (long long) b->nframes[(size_t) s] * max_frame_bytes : 0;
    }
    if (b->dev_pack && bytes_total > b->bytes_cap) {
        uint8_t *bigger = nullptr;
        HIPCHK(hipMalloc((void **) &bigger, (size_t) bytes_total));
        if (b->d_bytes)
            (void) hipFree(b->d_bytes);
        b->d_bytes = bigger;
        b->bytes_cap = bytes_total;
    }
    if (total > b->out_cap) {
        LhFrameOut *bigger = nullptr;
        HIPCHK(hipMalloc((void **) &bigger, (size_t) total * sizeof(LhFrameOut)));
        if (b->d_out)
            (void) hipFree(b->d_out);
        b->d_out = bigger;
        b->out_cap = total;
    }
    if (b->split && total > 0 && !batch_window_env())
        (void) batch_mid_reserve(b, total);     /* (no room: the launch will run in windows, or take the fused kernel) */
    return 0;
}

extern "C" int
lamehip_batch_sync(lamehip_batch * b)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    if (!b)
        return -1;
    if (b->ev_wait && b->launched) {
        /* The wait for the launch: the runtime's own waits spin on the stream's signal whatever the event's flags say
         * (measured: one CPU per rank for the whole launch), so the event is polled between short sleeps -- at most 100 us
         * late on a launch of tens to hundreds of milliseconds, and the CPU is free meanwhile. */
        hipError_t q;
        struct timespec nap = { 0, 100000 };
        while ((q = hipEventQuery(b->ev_wait)) == hipErrorNotReady)
            nanosleep(&nap, nullptr);
        if (q != hipSuccess)
            return set_err("hipEventQuery", q);
    }
    if (b->up_stream) {
        HIPCHK(hipStreamSynchronize(b->up_stream));
        HIPCHK(hipStreamSynchronize(b->down_stream));
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->encoded) {
        float   ms = 0;
        if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess)
            b->last_ms = ms;
        b->part_ms[0] = b->part_ms[1] = b->part_ms[2] = 0;
        if (b->last_split && b->last_windows > 1) {
            /* a launch in windows: the three parts summed over the windows (a window ends where the next one starts) */
            for (int k = 0; k < b->last_windows; k++) {
                hipEvent_t const e0 = b->ev_win[3 * (size_t) k], e1 = b->ev_win[3 * (size_t) k + 1], e2 = b->ev_win[3 * (size_t) k + 2];
                hipEvent_t const e3 = (k + 1 < b->last_windows) ? b->ev_win[3 * (size_t) k + 3] : b->ev1;
                float   a = 0, s_ = 0, q = 0;
                if (hipEventElapsedTime(&a, e0, e1) != hipSuccess || hipEventElapsedTime(&s_, e1, e2) != hipSuccess
                    || hipEventElapsedTime(&q, e2, e3) != hipSuccess) {
                    (void) hipGetLastError();
                    b->part_ms[0] = b->part_ms[1] = b->part_ms[2] = 0.0f;
                    break;
                }
                b->part_ms[0] += a;
                b->part_ms[1] += s_;
                b->part_ms[2] += q;
            }
        }
        else if (b->last_split) {
            if (hipEventElapsedTime(&b->part_ms[0], b->ev0, b->ev_part[0]) != hipSuccess
                || hipEventElapsedTime(&b->part_ms[1], b->ev_part[0], b->ev_part[1]) != hipSuccess
                || hipEventElapsedTime(&b->part_ms[2], b->ev_part[1], b->ev1) != hipSuccess) {
                (void) hipGetLastError();
                b->part_ms[0] = b->part_ms[1] = b->part_ms[2] = 0.0f;   /* (never a stale figure of an earlier launch) */
            }
        }
    }
    return 0;
}

extern "C" float
lamehip_batch_last_kernel_ms(lamehip_batch * b)
{
    return b ? b->last_ms : 0.0f;
}

/* the last launch kernel by kernel (split pipeline): analysis kernels, sub-band kernel, encode kernel, in ms; returns 1 when
 * the launch went through the split pipeline, 0 for the fused kernel (all of lamehip_batch_last_kernel_ms is that one kernel) */
extern "C" int
lamehip_batch_last_kernel_parts_ms(lamehip_batch * b, float *parts3)
{
    if (!b || !parts3)
        return -1;
    parts3[0] = b->part_ms[0];
    parts3[1] = b->part_ms[1];
    parts3[2] = b->part_ms[2];
    return b->last_split;
}

/* sub-launches of the last launch: 1, or the number of frame windows the split pipeline worked through (batch_plan) */
extern "C" int
lamehip_batch_last_windows(lamehip_batch * b)
{
    return b ? b->last_windows : 0;
}

extern "C" int
lamehip_batch_kernel_waves(lamehip_batch * b)
{
    return b ? 2 : 0;
}

extern "C" int
lamehip_batch_get_frames(lamehip_batch * b, int s, void *frames_out, int max_frames)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    int     n;
    if (!b || s < 0 || s >= b->B || !b->encoded)
        return -1;
    n = b->nframes[(size_t) s];
    if (n > max_frames)
        n = max_frames;
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(frames_out, b->d_out + b->out_off[(size_t) s], (size_t) n * sizeof(LhFrameOut),
                     hipMemcpyDeviceToHost));
    return n;
}


/* debug aid: raw LhStreamState carried by a single-stream handle between launches */
extern "C" int
lamehip_get_state(const lame_t g, void *out, int size)
{
    LhDeviceScope const on_device(valid(g) ? g->device : -1);
    if (!g || !g->have_device || size < (int) sizeof(LhStreamState))
        return -1;
    HIPCHK(hipStreamSynchronize(g->stream));
    HIPCHK(hipMemcpy(out, g->d_state, sizeof(LhStreamState), hipMemcpyDeviceToHost));
    return (int) sizeof(LhStreamState);
}

/* debug / profiling aid: raw LhStreamState of one stream */
extern "C" int
lamehip_batch_get_state(lamehip_batch * b, int s, void *out, int size)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    if (!b || s < 0 || s >= b->B || size < (int) sizeof(LhStreamState))
        return -1;
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(out, b->d_state + s, sizeof(LhStreamState), hipMemcpyDeviceToHost));
    return (int) sizeof(LhStreamState);
}

/* host bit packing of one stream from its frames (already on the host) */
static long
pack_stream(const lamehip_batch * b, int s, const LhFrameOut * fr, int n, unsigned char *out, long out_size)
{
    LhBitstream bs;
    long    pos = 0;
    if (lh_bs_init(&bs) != 0)
        return -2;
    for (int i = 0; i < n; i++) {
        int     k;
        if (lh_bs_format_frame(&bs, &b->cfg, b->tab, &fr[i]) != 0) {
            snprintf(g_err, sizeof(g_err), "inconsistent device payload (packer check %d) stream %d frame %d",
                     bs.error, s, i);
            lh_bs_free(&bs);
            return LAMEHIP_ERR_PAYLOAD;
        }
        if (lh_bs_pending(&bs) > out_size - pos) {
            lh_bs_free(&bs);
            return -1;
        }
        k = lh_bs_copy(&bs, out + pos, 0);
        pos += k;
    }
    lh_bs_flush(&bs, &b->cfg, n > 0 ? &fr[n - 1] : nullptr);
    {
        int     k;
        if (lh_bs_pending(&bs) > out_size - pos) {
            lh_bs_free(&bs);
            return -1;
        }
        k = lh_bs_copy(&bs, out + pos, 0);
        pos += k;
    }
    lh_bs_free(&bs);
    return pos;
}

extern "C" long
lamehip_batch_pack(lamehip_batch * b, int s, unsigned char *out, long out_size)
{
    std::vector < LhFrameOut > fr;
    int     n;
    if (!b || s < 0 || s >= b->B || !b->encoded)
        return -1;
    n = b->nframes[(size_t) s];
    fr.resize((size_t) n);
    if (lamehip_batch_get_frames(b, s, fr.data(), n) != n)
        return LAMEHIP_ERR_DEVICE;
    return pack_stream(b, s, fr.data(), n, out, out_size);
}

/* One stream as a complete file image: the final Xing/Info + LAME tag frame followed by the audio
 * frames (what the reference's frontend leaves on disk after lame_mp3_tags_fid).  When the tag
 * does not fit the frame size the audio alone is returned, as the reference would. */
extern "C" long
lamehip_batch_pack_tagged(lamehip_batch * b, int s, unsigned char *out, long out_size)
{
    LhVbrTag v;
    std::vector < LhFrameOut > fr;
    int     n, total;
    long    k;
    if (!b || s < 0 || s >= b->B || !b->encoded)
        return -1;
    total = lh_tag_init(&v, &b->cfg);
    if (b->rate_in)
        v.samplerate_in = b->rate_in;
    if (out_size < total)
        return -1;
    n = b->nframes[(size_t) s];
    fr.resize((size_t) n);
    if (lamehip_batch_get_frames(b, s, fr.data(), n) != n)
        return LAMEHIP_ERR_DEVICE;
    k = pack_stream(b, s, fr.data(), n, out + total, out_size - total);
    if (k < 0 || total == 0)
        return k;
    for (int i = 0; i < n; i++)
        lh_tag_add_frame(&v, lh_tag_kbps(b->cfg.version, fr[(size_t) i].bitrate_index));
    lh_tag_crc(&v, out + total, k);
    if (lh_tag_frame(&v, &b->cfg, b->cfg.vbr_q, batch_padding(b, s), n > 0 ? fr[(size_t) n - 1].mode_ext : 0,
                     out, total) != total)
        memset(out, 0, (size_t) total);         /* no frames: the reference leaves the placeholder */
    return k + total;
}

/* All streams, `nthreads' host threads: thread t takes streams t, t + nthreads, ...; each copies
 * a stream's payload D2H into its own pinned buffer and packs it (the packer is serial per
 * stream -- reference bitstream.c -- but streams are independent).  Stream s is written at
 * out + s * out_stride; sizes[s] = bytes, or a negative error code. */
extern "C" int
lamehip_batch_pack_all(lamehip_batch * b, int nthreads, unsigned char *out, long out_stride, long *sizes)
{
    LhDeviceScope const on_device(b ? b->device : -1);
    int     dev = 0, maxf = 0;
    std::atomic < int >failed(0);
    /* the first failure of any worker: its code and its message (g_err is per thread) reach the caller */
    std::mutex first_lock;
    int     first_code = 0;
    char    first_text[sizeof(g_err)] = "";
    if (!b || !b->encoded || !out || !sizes || out_stride <= 0)
        return -1;
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > b->B)
        nthreads = b->B;
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetDevice(&dev));
    for (int s = 0; s < b->B; s++)
        maxf = b->nframes[(size_t) s] > maxf ? b->nframes[(size_t) s] : maxf;
    {
        std::vector < std::thread > pool;
        for (int t = 0; t < nthreads; t++)
            pool.emplace_back([=, &failed, &first_lock, &first_code, &first_text] () {
                auto note = [&](long code, const char *text) {
                    std::lock_guard < std::mutex > hold(first_lock);
                    if (first_code == 0) {
                        first_code = (int) code;
                        snprintf(first_text, sizeof(first_text), "%s", text);
                    }
                    failed = 1;
                };
                LhFrameOut *h = nullptr;
                hipStream_t st = nullptr;
                if (hipSetDevice(dev) != hipSuccess
                    || hipHostMalloc((void **) &h, (size_t) (maxf > 0 ? maxf : 1) * sizeof(LhFrameOut), 0) != hipSuccess
                    || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
                    note(LAMEHIP_ERR_DEVICE, "pack_all: a worker could not set up its device staging");
                    for (int s = t; s < b->B; s += nthreads)
                        sizes[s] = LAMEHIP_ERR_DEVICE;
                    if (h)
                        (void) hipHostFree(h);
                    return;
                }
                for (int s = t; s < b->B; s += nthreads) {
                    int const n = b->nframes[(size_t) s];
                    long    r;
                    if (n > 0 && (hipMemcpyAsync(h, b->d_out + b->out_off[(size_t) s], (size_t) n * sizeof(LhFrameOut),
                                                 hipMemcpyDeviceToHost, st) != hipSuccess
                                  || hipStreamSynchronize(st) != hipSuccess))
                    {
                        r = LAMEHIP_ERR_DEVICE;
                        snprintf(g_err, sizeof(g_err), "pack_all: copying stream %d's frames from the device failed", s);
                    }
                    else
                        r = pack_stream(b, s, h, n, out + (size_t) s * (size_t) out_stride, out_stride);
                    sizes[s] = r;
                    if (r < 0) {
                        if (r == -1)
                            snprintf(g_err, sizeof(g_err), "pack_all: out_stride %ld is too small for stream %d", out_stride, s);
                        note(r, g_err);
                    }
                }
                (void) hipStreamDestroy(st);
                (void) hipHostFree(h);
            });
        for (auto & th:pool)
            th.join();
    }
    if (failed) {
        snprintf(g_err, sizeof(g_err), "%s", first_text);
        return first_code;
    }
    return 0;
}
