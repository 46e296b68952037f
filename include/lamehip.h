/*
 * lamehip.h -- C ABI of liblamehip: the MI355X-native MP3 encode inner loop
 * behind the reference's public API.
 *
 * Part 1 mirrors the subset of the reference's include/lame.h that its frontend
 * uses for CBR encoding; each entry point cites the reference declaration it
 * replaces (file:line under /root/reference).  Names, argument meaning, return
 * codes and the output lag (first call returns 0 bytes) are the reference's.
 * Part 2 is the batch extension the GPU needs: a single lame_encode_buffer call
 * carries 26 ms of one stream, a batch call carries whole streams for thousands
 * of handles.
 *
 * Everything here is plain C: pointers and sizes only, no torch / HIP types.
 * Unsupported settings (MPEG-2 / 2.5 output rates, free format) make
 * lame_init_params() return -1 instead of silently taking another path, and
 * every call fails with LAMEHIP_ERR_NODEVICE when no HIP device is present --
 * there is no CPU fallback inside this library.
 */
#ifndef LAMEHIP_H
#define LAMEHIP_H

#include <stdint.h>
#include <stddef.h>
#include <stdarg.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAMEHIP_ERR_NODEVICE (-10)
#define LAMEHIP_ERR_DEVICE   (-11)     /* a HIP call failed; see lamehip_last_error() */
#define LAMEHIP_ERR_PAYLOAD  (-12)     /* device payload failed the packer's consistency checks */

/* ------------------------------------------------------------------ */
/* Part 1: lame.h-shaped handle API                                    */

struct lame_global_struct;
typedef struct lame_global_struct lame_global_flags;
typedef lame_global_flags *lame_t;

/* MPEG_mode, reference include/lame.h:61-69 */
typedef enum MPEG_mode_e { STEREO = 0, JOINT_STEREO, DUAL_CHANNEL, MONO, NOT_SET, MAX_INDICATOR } MPEG_mode;
/* vbr_mode, reference include/lame.h:49-58 */
typedef enum vbr_mode_e { vbr_off = 0, vbr_mt, vbr_rh, vbr_abr, vbr_mtrh, vbr_max_indicator,
    vbr_default = vbr_mtrh } vbr_mode;

lame_t  lame_init(void);                                            /* lame.h:168 */
int     lame_set_in_samplerate(lame_t, int);                         /* lame.h:188 */
int     lame_get_in_samplerate(const lame_t);                        /* lame.h:189 */
int     lame_set_num_channels(lame_t, int);                          /* lame.h:192 (2, or 1 = mono: only buffer_l is read) */
int     lame_get_num_channels(const lame_t);                         /* lame.h:193 */
int     lame_set_out_samplerate(lame_t, int);                        /* lame.h:224 (must equal the input rate) */
int     lame_get_out_samplerate(const lame_t);                       /* lame.h:225 */
int     lame_set_brate(lame_t, int);                                 /* lame.h:353 */
int     lame_get_brate(const lame_t);                                /* lame.h:354 */
int     lame_set_mode(lame_t, MPEG_mode);                            /* lame.h:270 */
MPEG_mode lame_get_mode(const lame_t);                               /* lame.h:271 */
int     lame_set_quality(lame_t, int);                               /* lame.h:263 */
int     lame_get_quality(const lame_t);                              /* lame.h:264 */
int     lame_set_VBR(lame_t, vbr_mode);                              /* lame.h:432 (vbr_off, vbr_abr, vbr_mt / vbr_mtrh) */
vbr_mode lame_get_VBR(const lame_t);                                 /* lame.h:433 */
int     lame_set_VBR_q(lame_t, int);                                 /* lame.h:436 (0 best .. 9; default 4) */
int     lame_get_VBR_q(const lame_t);                                /* lame.h:437 */
int     lame_set_preset(lame_t, int);                     /* --preset: V0..V9, named presets, 8..320 = ABR; lame.h:359 */
int     lame_set_VBR_quality(lame_t, float);              /* -V n.f, lame.h:393 */
float   lame_get_VBR_quality(const lame_t);
int     lame_set_VBR_min_bitrate_kbps(lame_t, int);       /* -b with VBR / ABR, lame.h:403 */
int     lame_get_VBR_min_bitrate_kbps(const lame_t);
int     lame_set_VBR_max_bitrate_kbps(lame_t, int);       /* -B, lame.h:406 */
int     lame_get_VBR_max_bitrate_kbps(const lame_t);
int     lame_set_VBR_hard_min(lame_t, int);               /* -F: the minimum also holds for digital silence, lame.h:413 */
int     lame_get_VBR_hard_min(const lame_t);
int     lame_set_VBR_mean_bitrate_kbps(lame_t, int);                 /* lame.h:444 (ABR mean, with lame_set_VBR(vbr_abr)) */
int     lame_get_VBR_mean_bitrate_kbps(const lame_t);                /* lame.h:445 */
int     lame_set_bWriteVbrTag(lame_t, int);                          /* lame.h:240 (default 1, as in the reference) */
int     lame_get_bWriteVbrTag(const lame_t);                         /* lame.h:241 */
int     lame_set_findReplayGain(lame_t, int);                        /* lame.h:296 (accepted, ignored) */
/* frontend switches (-m f, --nores, -p, -c, -o, -e, --strictly-enforce-ISO, --lowpass, --scale*,
 * --noshort, --shortblocks); values and defaults are the reference's */
int     lame_set_force_ms(lame_t, int);                              /* lame.h:288 (joint stereo only) */
int     lame_get_force_ms(const lame_t);
int     lame_set_disable_reservoir(lame_t, int);                     /* lame.h:400 */
int     lame_get_disable_reservoir(const lame_t);
int     lame_set_error_protection(lame_t, int);                      /* lame.h:376 (CRC-16 after the header) */
int     lame_get_error_protection(const lame_t);
int     lame_set_copyright(lame_t, int);                             /* lame.h:368 */
int     lame_get_copyright(const lame_t);
int     lame_set_original(lame_t, int);                              /* lame.h:372 */
int     lame_get_original(const lame_t);
int     lame_set_emphasis(lame_t, int);                              /* lame.h:558 */
int     lame_get_emphasis(const lame_t);
int     lame_set_extension(lame_t, int);                             /* lame.h:387 */
int     lame_get_extension(const lame_t);
int     lame_set_strict_ISO(lame_t, int);                            /* lame.h:391 (0 default, 1 strict, 2 maximum) */
int     lame_get_strict_ISO(const lame_t);
int     lame_set_lowpassfreq(lame_t, int);                           /* lame.h:470 (Hz; 0 default, -1 none) */
int     lame_get_lowpassfreq(const lame_t);
int     lame_set_lowpasswidth(lame_t, int);                          /* lame.h:473 */
int     lame_get_lowpasswidth(const lame_t);
int     lame_set_scale(lame_t, float);                               /* lame.h:199 */
int     lame_set_scale_left(lame_t, float);                          /* lame.h:206 */
int     lame_set_scale_right(lame_t, float);                         /* lame.h:213 */
int     lame_set_allow_diff_short(lame_t, int);                      /* lame.h:535 */
int     lame_set_no_short_blocks(lame_t, int);                       /* lame.h:547 */
int     lame_set_force_short_blocks(lame_t, int);                    /* lame.h:551 */
int     lame_init_params(lame_t);                                    /* lame.h:636 */
int     lame_get_framesize(const lame_t);                            /* lame.h:582 */
int     lame_get_frameNum(const lame_t);                             /* lame.h:597 */
int     lame_get_encoder_delay(const lame_t);                        /* lame.h:571 */
int     lame_get_encoder_padding(const lame_t);                      /* lame.h:579, known after lame_encode_flush */
int     lame_get_mf_samples_to_encode(const lame_t);                 /* lame.h:585 */
int     lame_set_num_samples(lame_t, unsigned long);                 /* lame.h:184 */
unsigned long lame_get_num_samples(const lame_t);
int     lame_get_totalframes(const lame_t);                          /* lame.h:603 */
/* histograms over the frames encoded so far, for a frontend's progress display (lame.h:893-960) */
void    lame_bitrate_kbps(const lame_t, int bitrate_kbps[14]);
void    lame_bitrate_hist(const lame_t, int bitrate_count[14]);
void    lame_stereo_mode_hist(const lame_t, int stereo_mode_count[4]);
void    lame_bitrate_stereo_mode_hist(const lame_t, int bitrate_stmode_count[14][4]);
void    lame_block_type_hist(const lame_t, int btype_count[6]);
void    lame_bitrate_block_type_hist(const lame_t, int bitrate_btype_count[14][6]);
/* message callbacks (lame.h:346-348): a failed call on the handle hands its explanation -- the text
 * lamehip_last_error() returns -- to errorf; the default prints to stderr like the reference's,
 * NULL silences.  debugf / msgf are stored for the frontend's sake (this library is quiet). */
typedef void (*lame_report_function)(const char *format, va_list ap);
int     lame_set_errorf(lame_t, lame_report_function);                   /* lame.h:346 */
int     lame_set_debugf(lame_t, lame_report_function);                   /* lame.h:347 */
int     lame_set_msgf(lame_t, lame_report_function);                     /* lame.h:348 */
int     lame_get_version(const lame_t);                              /* lame.h:568 */
/* The frontend's tuning switches (its "experimental" section).  A value left alone takes the bitrate's / quality's
 * preset value or the default of lame_init_params, exactly as in the reference (presets.c SET_OPTION, lame.c:1112-1203);
 * every one only changes constants and tables that reach the kernels.  lame_set_free_format(1), lame_set_experimentalZ
 * and lame_set_ATHonly(1) are accepted here and make lame_init_params fail (not on this path). */
void    lame_set_msfix(lame_t, double);                              /* lame.h:424 */
float   lame_get_msfix(const lame_t);
int     lame_set_ATHtype(lame_t, int);                               /* lame.h:502 */
int     lame_get_ATHtype(const lame_t);
int     lame_set_ATHlower(lame_t, float);                            /* lame.h:506 */
float   lame_get_ATHlower(const lame_t);
int     lame_set_athaa_type(lame_t, int);                            /* lame.h:510 */
int     lame_get_athaa_type(const lame_t);
int     lame_set_athaa_sensitivity(lame_t, float);                   /* lame.h:521 */
float   lame_get_athaa_sensitivity(const lame_t);
int     lame_set_ATHonly(lame_t, int);                               /* lame.h:490 */
int     lame_set_ATHshort(lame_t, int);                              /* lame.h:494 */
int     lame_set_noATH(lame_t, int);                                 /* lame.h:498 */
int     lame_set_interChRatio(lame_t, float);                        /* lame.h:543 */
int     lame_set_useTemporal(lame_t, int);                           /* lame.h:539 */
int     lame_set_highpassfreq(lame_t, int);                          /* lame.h:477 */
int     lame_set_highpasswidth(lame_t, int);                         /* lame.h:480 */
int     lame_set_exp_nspsytune(lame_t, int);                         /* lame.h:421 */
int     lame_get_exp_nspsytune(const lame_t);
int     lame_set_experimentalY(lame_t, int);                         /* lame.h:413 */
int     lame_set_experimentalZ(lame_t, int);                         /* lame.h:417 */
int     lame_set_free_format(lame_t, int);                           /* lame.h:292 */
int     lame_get_free_format(const lame_t);
int     lame_set_compression_ratio(lame_t, float);                   /* lame.h:355 */
float   lame_get_compression_ratio(const lame_t);
/* ReplayGain of the input for the LAME tag (the frontend's default): measured on the host, lh_replaygain.c */
int     lame_set_findReplayGain(lame_t, int);                        /* lame.h:296 */
int     lame_get_RadioGain(const lame_t);                            /* lame.h:606 */
/* parts of the reference this library does not have (decoder, assembler variants): "off" is accepted */
int     lame_set_decode_only(lame_t, int);                           /* lame.h:244 */
int     lame_set_asm_optimizations(lame_t, int, int);                /* lame.h:360 */
int     lame_set_nogap_total(lame_t, int);                           /* lame.h:326 */
int     lame_set_nogap_currentindex(lame_t, int);                    /* lame.h:329 */
void    lame_print_config(const lame_t);                             /* lame.h:678 */
void    lame_print_internals(const lame_t);                          /* lame.h:680 */
int     lame_get_bitrate(int mpeg_version, int table_index);         /* lame.h:1290 */
int     lame_get_samplerate(int mpeg_version, int table_index);      /* lame.h:1291 */
const char *get_lame_version(void);                                  /* lame.h:644 */
const char *get_lame_short_version(void);
const char *get_lame_very_short_version(void);
const char *get_psy_version(void);
const char *get_lame_url(void);
const char *get_lame_os_bitness(void);

/* return: bytes written to mp3buf (may be 0); -1 mp3buf too small; -2 alloc;
 * -3 lame_init_params not called; mp3buf_size == 0 disables the size check
 * (reference lame.h:687-722) */
int     lame_encode_buffer(lame_t, const short int buffer_l[], const short int buffer_r[],
                           const int nsamples, unsigned char *mp3buf, const int mp3buf_size);   /* lame.h:715 */
int     lame_encode_buffer_interleaved(lame_t, short int pcm[], int num_samples,
                                       unsigned char *mp3buf, int mp3buf_size);                  /* lame.h:730 */
/* the other sample types of the reference (same return codes); each sample goes through the
 * type's scale and the pcm_transform matrix exactly as lame_copy_inbuffer does */
int     lame_encode_buffer_float(lame_t, const float l[], const float r[], const int nsamples,
                                 unsigned char *mp3buf, const int mp3buf_size);                  /* lame.h:746 (+/-32768) */
int     lame_encode_buffer_ieee_float(lame_t, const float l[], const float r[], const int nsamples,
                                      unsigned char *mp3buf, const int mp3buf_size);             /* lame.h:758 (+/-1.0) */
int     lame_encode_buffer_interleaved_ieee_float(lame_t, const float pcm[], const int nsamples,
                                                  unsigned char *mp3buf, const int mp3buf_size); /* lame.h:765 */
int     lame_encode_buffer_ieee_double(lame_t, const double l[], const double r[], const int nsamples,
                                       unsigned char *mp3buf, const int mp3buf_size);            /* lame.h:776 */
int     lame_encode_buffer_interleaved_ieee_double(lame_t, const double pcm[], const int nsamples,
                                                   unsigned char *mp3buf, const int mp3buf_size);        /* lame.h:783 */
int     lame_encode_buffer_long(lame_t, const long l[], const long r[], const int nsamples,
                                unsigned char *mp3buf, const int mp3buf_size);                   /* lame.h:799 (+/-32768) */
int     lame_encode_buffer_long2(lame_t, const long l[], const long r[], const int nsamples,
                                 unsigned char *mp3buf, const int mp3buf_size);                  /* lame.h:813 (+/-2^63) */
int     lame_encode_buffer_int(lame_t, const int l[], const int r[], const int nsamples,
                               unsigned char *mp3buf, const int mp3buf_size);                    /* lame.h:831 (+/-2^31) */
int     lame_encode_flush(lame_t, unsigned char *mp3buf, int size);                              /* lame.h:856 */
/* --nogap: end a file without draining the sample buffers, then start the next one (lame.h:870, 886) */
int     lame_encode_flush_nogap(lame_t, unsigned char *mp3buf, int size);
int     lame_init_bitstream(lame_t);
/* final Xing/Info + LAME tag frame that replaces the placeholder at the head of the stream */
size_t  lame_get_lametag_frame(const lame_t, unsigned char *buffer, size_t size);                /* lame.h:970 */
int     lame_close(lame_t);                                                                      /* lame.h:977 */

/* ------------------------------------------------------------------ */
/* Part 2: batch extension (not in lame.h)                             */

typedef struct lamehip_batch lamehip_batch;

/* B independent streams that share the settings of `proto' (which must have
 * passed lame_init_params); capacity = samples per channel per stream. */
lamehip_batch *lamehip_batch_create(const lame_t proto, int nstreams, long capacity_samples);
/* the same on HIP device `device' (0 .. lamehip_device_count() - 1) instead of the calling thread's
 * current one.  A batch keeps its device: every later call on it runs there and leaves the
 * caller's current device as it was.  This is what one host thread per GPU uses to shard a batch
 * of independent streams over the GPUs of a node (no inter-device traffic). */
lamehip_batch *lamehip_batch_create_on(int device, const lame_t proto, int nstreams, long capacity_samples);
/* device of a handle's own (per-frame) launches; before lame_init_params.  Default: the device
 * that is current when lame_init_params runs. */
int     lamehip_set_device(lame_t, int device);
void    lamehip_batch_destroy(lamehip_batch *);
/* copy one stream's planar s16 PCM host -> HBM (H2D, synchronous).  When proto's input rate differs from
 * its output rate the stream is converted on the host first, exactly as the reference converts it when
 * lame_encode_buffer is fed 1152 input samples per call (util.c:520-697; capacity then counts input
 * samples, and the _device variants below are refused) */
int     lamehip_batch_set_pcm(lamehip_batch *, int stream, const short *l, const short *r, long nsamples);
/* same, from planar s16 buffers that already live in HBM (D2D) */
int     lamehip_batch_set_pcm_device(lamehip_batch *, int stream, const void *dev_l, const void *dev_r, long nsamples);
/* device pointer of the PCM pool: int16 [stream][2][capacity]; lets a producer
 * that already lives on the GPU fill it in place (then declare lengths) */
void   *lamehip_batch_pcm_device_ptr(lamehip_batch *);
int     lamehip_batch_set_length(lamehip_batch *, int stream, long nsamples);

/* Incremental use -- lame_encode_buffer (lame.h:715-722, lame.c:1672-1775) for all streams of a batch at once:
 *   lamehip_batch_append            stage n more samples of one stream (host, pinned memory; any n >= 0, ragged
 *                                   across streams; several calls per stream may precede one encode)
 *   lamehip_batch_encode_available  ONE asynchronous H2D copy of everything staged, ONE launch that encodes
 *                                   every stream's newly complete frames, packing; returns the frame count
 *   lamehip_batch_drain             the bytes a stream has produced since its last drain: what the
 *                                   reference's lame_encode_buffer returns for the same calls (the output
 *                                   lags the input by the priming: the first 1152 samples yield 0 bytes)
 *   lamehip_batch_finish            lame_encode_flush (lame.h:869, lame.c:2041-2175) for every stream
 * The first append switches the batch to this mode (its capacity still bounds a stream's total length).
 * Return codes as lame_encode_buffer's: >= 0, -1 a buffer or the capacity is too small, -2 allocation. */
int     lamehip_batch_append(lamehip_batch *, int stream, const short *l, const short *r, int nsamples);
int     lamehip_batch_encode_available(lamehip_batch *);
int     lamehip_batch_drain(lamehip_batch *, int stream, unsigned char *out, int out_size);
int     lamehip_batch_finish(lamehip_batch *);
/* encode every stream completely (all frames incl. the flush frames), payload
 * stays in HBM; asynchronous on the batch's HIP stream */
int     lamehip_batch_encode(lamehip_batch *);
int     lamehip_batch_sync(lamehip_batch *);
/* total frames of a stream (known after set_pcm / set_length) */
int     lamehip_batch_frames(lamehip_batch *, int stream);
/* D2H of one stream's payload + host bit packing; returns bytes or <0 */
long    lamehip_batch_pack(lamehip_batch *, int stream, unsigned char *out, long out_size);
/* one stream as a complete file image: final Xing/Info + LAME tag frame, then the audio frames */
long    lamehip_batch_pack_tagged(lamehip_batch *, int stream, unsigned char *out, long out_size);
/* the same for all streams with `nthreads' host threads (streams are independent; the packer is
 * serial per stream): stream s at out + s * out_stride, sizes[s] = bytes or a negative code */
int     lamehip_batch_pack_all(lamehip_batch *, int nthreads, unsigned char *out, long out_stride, long *sizes);
/* Bit packing on the device: with this switched on (before lamehip_batch_encode) the kernel also
 * assembles every stream's MP3 bytes in HBM -- headers, side information, Huffman data, reservoir
 * back pointers, final padding (reference bitstream.c:format_bitstream / flush_bitstream) -- and
 * lamehip_batch_get_bytes copies them out; the host packer is not involved.  Same bytes as
 * lamehip_batch_pack. */
int     lamehip_batch_set_device_packing(lamehip_batch *, int on);
long    lamehip_batch_get_bytes(lamehip_batch *, int stream, unsigned char *out, long out_size);
int     lamehip_batch_get_bytes_all(lamehip_batch *, unsigned char *out, long out_stride, long *sizes);
/* Pipelined use (several batches in flight, each on its own HIP stream; replaces the seam of the reference's
 * frontend loop lame_encode_buffer -> fwrite, frontend/lame_main.c:449-520, for many streams at once):
 *   lamehip_batch_pcm_host_ptr   pinned mirror of the s16 pool, short[stream][2][capacity]; write the samples there
 *                                (then lamehip_batch_set_length + lamehip_batch_mark_pcm) or let lamehip_batch_set_pcm
 *                                copy them there -- NULL for a batch that converts the sample rate;
 *   lamehip_batch_upload         one asynchronous H2D copy of what changed (lamehip_batch_encode does it if pending);
 *   lamehip_batch_fetch          after lamehip_batch_encode of a device-packed batch: bytes and per-stream sizes start
 *                                their way to pinned host memory behind the kernel, asynchronously;
 *   lamehip_batch_bytes_ptr      waits for them; stream's bytes in place (valid until the next encode), returns their
 *                                number or a negative code. */
short  *lamehip_batch_pcm_host_ptr(lamehip_batch *);
int     lamehip_batch_mark_pcm(lamehip_batch *, int stream);
int     lamehip_batch_upload(lamehip_batch *);
int     lamehip_batch_fetch(lamehip_batch *);
long    lamehip_batch_bytes_ptr(lamehip_batch *, int stream, const unsigned char **bytes);
/* tag frame + audio frames, like lamehip_batch_pack_tagged, from the device-packed bytes */
long    lamehip_batch_get_bytes_tagged(lamehip_batch *, int stream, unsigned char *out, long out_size);
/* raw payload access for tests: copies frames [0, n) of a stream (LhFrameOut[]) */
int     lamehip_batch_get_frames(lamehip_batch *, int stream, void *frames_out, int max_frames);
/* debug aid: raw per-stream carried state (LhStreamState, csrc/lh_device.h) */
int     lamehip_batch_get_state(lamehip_batch *, int stream, void *out, int size);
int     lamehip_get_state(const lame_t, void *out, int size);   /* same for a single-stream handle */
/* allocate what a launch of the batch's present streams needs (payload, analysis pool, byte pool) now rather than in the first
 * lamehip_batch_encode */
int     lamehip_batch_reserve(lamehip_batch *);
/* elapsed GPU time of the last lamehip_batch_encode in ms (HIP events on the batch stream) */
float   lamehip_batch_last_kernel_ms(lamehip_batch *);
/* the same launch kernel by kernel: a batch's frames go through the split pipeline -- analysis kernels over all frames at
 * once (attack detection, FHTs, spectra, masking up to the recurrences: csrc/lh_analysis.hip), the sub-band kernel
 * (polyphase + MDCT: csrc/lh_subband.hip), then the per-stream encode kernel -- parts3[0..2] = their times in ms.  Returns 1,
 * or 0 when the launch was the single fused kernel (environment LAMEHIP_FUSED=1, or no memory for the pools). */
int     lamehip_batch_last_kernel_parts_ms(lamehip_batch *, float *parts3);
/* The analysis kernels leave 34 KB per frame for the encode kernel (81 GB at 1024 streams x 60 s).  A launch whose records do not
 * fit the device's free memory is worked through in WINDOWS of as many frames per stream as do fit -- analysis kernels, then
 * the encode kernel, window after window on the batch's HIP stream, the streams' state carried between them as between two
 * launches of an incremental batch; results are the same bytes.  Returns the number of windows of the last launch (1: all at
 * once; parts3 above are sums over the windows).  Environment LAMEHIP_MID_WINDOW=n forces windows of n frames (tuning, tests);
 * under 64 frames per window the launch takes the fused kernel instead and lamehip_last_error() says so. */
int     lamehip_batch_last_windows(lamehip_batch *);
/* waves per stream of the kernel this batch encodes with: 2 (one per channel) */
int     lamehip_batch_kernel_waves(lamehip_batch *);
int     lamehip_batch_reset(lamehip_batch *);   /* re-initialise all stream states for another run */

/* device self-test of the wave-level primitives the kernels rely on: 0 = pass */
int     lamehip_selftest(void);
const char *lamehip_last_error(void);
/* sizeof() of the POD layouts (0 LhConfig, 1 LhTables, 2 LhFrameOut, 3 LhGranule,
 * 4 LhStreamState, 5 LhStreamDesc) this library was built with */
int     lamehip_abi_sizeof(int which);
int     lamehip_device_count(void);
/* table / config access for tests (fills LhConfig / LhTables images) */
int     lamehip_get_config(const lame_t, void *cfg_out, int size);
int     lamehip_get_tables(const lame_t, void *tab_out, int size);

#ifdef __cplusplus
}
#endif
#endif
