/*
 * radiorust_amd.h — C ABI of the MI355X (gfx950) IQ-stream DSP backend.
 *
 * The reference (JanBeh/radiorust v0.5.0, 100 % Rust) has no FFI seam of its
 * own; the seam this library replaces is the body of a block's task between
 * `receiver.recv()` and `sender.send()` (template: NopSignal,
 * src/blocks/mod.rs:206-238, the line marked "no operation here").  One handle
 * = the per-task state of one reference block.  Every entry point cites the
 * reference code it stands in for (paths relative to /root/reference).
 *
 * Conventions
 *   - rr_c32 / rr_c64 are layout-identical to num_complex::Complex<f32/f64>
 *     (#[repr(C)] {re, im}; re-exported at src/numbers.rs:10).
 *   - dtype: RR_F32 or RR_F64 = the reference's generic `Flt` (numbers.rs:23-42).
 *     `in`/`out` point to rr_c32 or rr_c64 arrays accordingly.
 *   - Every function returns an int status (RR_OK = 0).  Nothing unwinds across
 *     the boundary.  rr_last_error_string() describes the last failure on the
 *     calling thread.  Contract violations that `panic!`/`assert!` in the
 *     reference return RR_ERR_CONTRACT; the Rust shim re-raises them as panics.
 *   - A handle is used by one thread at a time (its state lives in one task
 *     closure in the reference, e.g. transform.rs:307-310) but may move between
 *     threads between calls: every entry point selects the handle's device.
 *   - Input is borrowed and never written (a Chunk is a shared Arc<Vec<T>>,
 *     bufferpool.rs:44-48); output is caller-allocated (ChunkBuf from the
 *     block's own pool, bufferpool.rs:213-222).
 *   - `*_process`      : host pointers, blocking (H2D, kernels, D2H, sync).
 *     `*_enqueue`      : host pointers (ideally pinned, see rr_host_*), returns
 *                        after queueing on the handle's stream; *n_out is final
 *                        on return (all schedules are computed on the host),
 *                        data is valid after rr_wait()/rr_query()==RR_OK.
 *     `*_process_dev`  : device pointers, asynchronous on the handle's stream.
 *   - There is no CPU fallback: with no usable HIP device every create fails
 *     with RR_ERR_HIP.
 */
#ifndef RADIORUST_AMD_H
#define RADIORUST_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } rr_c32;
typedef struct { double re, im; } rr_c64;

enum rr_status {
    RR_OK = 0,
    RR_ERR_BAD_ARG = 1,   /* null pointer, unknown dtype, ...                  */
    RR_ERR_CAPACITY = 2,  /* out_cap too small; nothing was consumed           */
    RR_ERR_HIP = 3,       /* HIP runtime error / no device                     */
    RR_ERR_CONTRACT = 4,  /* the reference would panic (assert!/expect)        */
    RR_ERR_NEED_DESIGN = 5, /* Filter: (rate, chunk length) not designed yet   */
    RR_ERR_NOT_READY = 6  /* rr_query: work still in flight                    */
};

enum rr_dtype { RR_F32 = 0, RR_F64 = 1 };

/* Window trait objects stay on the host (windowing.rs:6-10).  Built-ins are
 * described by value; anything else is sampled by the caller. */
enum rr_window_kind { RR_WIN_RECTANGULAR = 0, RR_WIN_KAISER = 1, RR_WIN_SAMPLED = 2 };
typedef struct {
    int kind;      /* rr_window_kind                                            */
    double beta;   /* Kaiser beta (windowing.rs:24-51)                          */
} rr_window;

typedef struct rr_block rr_block; /* base of every handle below */
typedef struct rr_freqshifter rr_freqshifter;
typedef struct rr_filter rr_filter;
typedef struct rr_downsampler rr_downsampler;
typedef struct rr_fourier rr_fourier;
typedef struct rr_chain rr_chain;

/* ------------------------------------------------------------------------ */
/* library / device                                                         */
/* ------------------------------------------------------------------------ */
int rr_version(void);
const char *rr_last_error_string(void);
int rr_device_count(int *count);
/* PCI address of a device ("0000:c1:00.0"): what a host layer needs to place its feeder thread and its pinned pool
 * on the GPU's NUMA node; out_cap >= 16. */
int rr_device_pci_bus_id(int device, char *out, size_t out_cap);

/* Any handle may be passed as rr_block*.  stream = hipStream_t (NULL restores
 * the handle's own stream).  The handle never owns a caller's stream. */
int rr_set_stream(rr_block *h, void *hip_stream);
int rr_wait(rr_block *h);   /* block until everything queued on h is done       */
int rr_query(rr_block *h);  /* RR_OK = idle, RR_ERR_NOT_READY = in flight       */

/* bufferpool.rs:187-222 re-backed by pinned memory: page-locked allocations
 * for ChunkBufPool, or registration of an existing Vec's storage. */
int rr_host_alloc(size_t bytes, void **out);
int rr_host_free(void *p);
int rr_host_register(void *p, size_t bytes);
int rr_host_unregister(void *p);

/* ------------------------------------------------------------------------ */
/* design math — src/math.rs:7-49, src/windowing.rs (host, f64, no GPU)     */
/* ------------------------------------------------------------------------ */
double rr_bessel_i0(double x);                          /* math.rs:7-20  */
double rr_kaiser_rel_with_beta(double beta, double x);  /* math.rs:26-28 */
double rr_kaiser_alpha_to_beta(double alpha);           /* math.rs:31-33 */
double rr_kaiser_null_at_bin_to_beta(double n);         /* math.rs:37-39 */
double rr_sinc(double x);                               /* math.rs:42-49 */
/* blocks::filters::deemphasis_factor (filters.rs:20-27): the complex amplification 1 / (1 + j tau 2 pi f) of a passive
 * first-order RC low-pass (tau e.g. 50e-6), what examples/relm_app/simple_receiver.rs:43-49 builds its second Filter from. */
int rr_deemphasis_factor(double tau, double frequency, rr_c64 *out);
/* out[i] = window.relative_value_at(2 (i + 0.5) / n - 1) for a built-in window
 * (the sampling positions of filters.rs:209-212 and analysis.rs:93-94). */
int rr_window_sample(const rr_window *w, size_t n, double *out);

/* FreqShifter::with_precision_and_shift's `freq_to_ratio` closure + num's
 * Ratio::new reduction (transform.rs:298-302). */
int rr_freqshifter_ratio(double sample_rate, double precision, double shift,
                         int64_t *numer, int64_t *denom);
/* The phase table of transform.rs:326-340 for `Flt` = dtype; table has room for
 * denom complex entries of that dtype. */
int rr_freqshifter_table(int dtype, int64_t numer, int64_t denom,
                         double start_phase, void *table);

/* Filter design, filters.rs:184-225: `resp[i]` is what the reference's
 * `response` vector holds before the division by `scale`, i.e.
 * resp[i] = freq_resp(i, i*fs/n) for i <= (n-1)/2, resp[n-i] = freq_resp(-i,
 * -i*fs/n), everything else 0 (the host layer evaluates the closure);
 * `window_rel[i]` = window.relative_value_at(2(i+.5)/n-1).  Writes the n
 * equivalent causal FIR taps g[k] = 2n*h[k] (f64): out[t] = sum_k g[k] x[t-k]. */
int rr_filter_design_taps(size_t n, const rr_c64 *resp, const double *window_rel,
                          rr_c64 *taps);

/* Downsampler design, resampling.rs:82-99: *ir_len = L; ir (capacity ir_cap,
 * may be NULL to query L) receives the unit-energy f64 impulse response. */
int rr_downsampler_design(double input_rate, double output_rate, double bandwidth,
                          double quality, size_t *ir_len, double *ir, size_t ir_cap);

/* The decimation schedule of resampling.rs:110-112 run over n_in inputs starting
 * from *pos (0.0 for a fresh block): emit[k] (capacity emit_cap, may be NULL)
 * receives the 0-based index of the input after which output k is produced;
 * *count the number of outputs; *pos the carried-over position. */
int rr_downsampler_schedule(double input_rate, double output_rate, size_t n_in,
                            double *pos, uint32_t *emit, size_t emit_cap, size_t *count);

/* The interpolation schedule of resampling.rs:248-265 run over n_in inputs starting
 * from *pos (0.0 for a fresh block): before[t] (capacity before_cap, may be NULL)
 * receives the number of outputs released before input t is added; *count the
 * number of outputs; *pos the carried-over position.  Without a list, rates that
 * are whole multiples of 2^-s take the closed form the device kernel uses. */
int rr_upsampler_schedule(double input_rate, double output_rate, size_t n_in,
                          double *pos, int32_t *before, size_t before_cap, size_t *count);

/* Fourier window, analysis.rs:88-101: values[i] = rel[i] * sqrt(n / sum rel^2). */
int rr_fourier_design_window(size_t n, const double *window_rel, double *values);
/* Which kernels transform a chunk of n samples (host only, no device needed; the same decision rr_fourier_process takes):
 * writes a short description into buf - "pow2", "pow2 two passes 128 x 256", "pow2 five launches 1024 x 1024",
 * "mixed 5 5 5 4 2" (the radices of the passes), "mixed two passes 125 x 160", "bluestein wave M=1024",
 * "bluestein one kernel M=4096" (M up to 8192 in f32, 4096 in f64), "bluestein four launches M=65536" (five / many for other M), "direct".  analysis.rs:82-115 accepts any length. */
int rr_fourier_route(int dtype, size_t n, char *buf, size_t cap);

/* ------------------------------------------------------------------------ */
/* FreqShifter — src/blocks/transform.rs:266-391                            */
/* ------------------------------------------------------------------------ */
/* FreqShifter::with_precision_and_shift (transform.rs:297) */
int rr_freqshifter_create(int dtype, double precision, double shift, int device,
                          rr_freqshifter **out);
/* FreqShifter::set_shift (transform.rs:384-386): applies at the next process,
 * keeping the phase continuous (transform.rs:322-325). */
int rr_freqshifter_set_shift(rr_freqshifter *h, double shift);
int rr_freqshifter_shift(const rr_freqshifter *h, double *shift);         /* :380 */
int rr_freqshifter_precision(const rr_freqshifter *h, double *precision); /* :376 */
/* One Signal::Samples message (transform.rs:314-355); n_out = n_in. */
int rr_freqshifter_process(rr_freqshifter *h, double sample_rate, const void *in,
                           size_t n_in, void *out, size_t out_cap, size_t *n_out);
int rr_freqshifter_enqueue(rr_freqshifter *h, double sample_rate, const void *in,
                           size_t n_in, void *out, size_t out_cap, size_t *n_out);
int rr_freqshifter_process_dev(rr_freqshifter *h, double sample_rate,
                               const void *d_in, size_t n_in, void *d_out,
                               size_t out_cap, size_t *n_out);
int rr_freqshifter_destroy(rr_freqshifter *h);

/* ------------------------------------------------------------------------ */
/* Filter — src/blocks/filters.rs:110-298                                   */
/* ------------------------------------------------------------------------ */
int rr_filter_create(int dtype, int device, rr_filter **out);
/* 1 when the reference would `recalculate` for this message (filters.rs:178-
 * 183): never designed, params updated, sample rate or chunk length changed. */
int rr_filter_needs_design(const rr_filter *h, double sample_rate, size_t n,
                           int *needed);
/* Filter::update / update_with_window (filters.rs:279-297): marks the params
 * as changed so that needs_design reports 1 for the next message. */
int rr_filter_mark_params_changed(rr_filter *h);
/* The `if recalculate { ... }` body (filters.rs:184-239); arguments as for
 * rr_filter_design_taps.  Drops the history like `previous_chunk = None`. */
int rr_filter_design(rr_filter *h, double sample_rate, size_t n,
                     const rr_c64 *resp, const double *window_rel);
/* Event::is_interrupt() handling (filters.rs:262-265). */
/* A GainControl (src/blocks/transform.rs:29-92: `sample * gain`, gain cast to Flt) wired BEHIND this block, as the
 * reference's receiver does behind its last Downsampler (examples/relm_app/simple_receiver.rs:52-56), without a pass of its
 * own: the gain is folded into the block's response tables on the host (Filter, Downsampler: the device does nothing for it;
 * results differ from the two-block composition by rounding only) or rides on the store (FmDemod: bit-equal to the
 * composition).  Takes effect at the next process call (`watch` semantics, transform.rs:64-66); histories are kept. */
int rr_filter_set_gain(rr_filter *h, double gain);
int rr_filter_reset(rr_filter *h);
/* One Signal::Samples message of exactly the designed length n
 * (filters.rs:240-260): n_out = 0 for the first chunk after create/design/
 * reset, n afterwards.  RR_ERR_NEED_DESIGN if (sample_rate, n_in) differ from
 * the design. */
int rr_filter_process(rr_filter *h, double sample_rate, const void *in, size_t n_in,
                      void *out, size_t out_cap, size_t *n_out);
int rr_filter_enqueue(rr_filter *h, double sample_rate, const void *in, size_t n_in,
                      void *out, size_t out_cap, size_t *n_out);
/* Batched form for device-resident streams: n_in = k*n consecutive chunks
 * (what a Rechunker(n) in front of the Filter would deliver, chunks.rs:42-177);
 * n_out = n_in - n if the history was empty, n_in otherwise. */
int rr_filter_process_dev(rr_filter *h, double sample_rate, const void *d_in,
                          size_t n_in, void *d_out, size_t out_cap, size_t *n_out);
/* SURVEY 8(d) cfg5's reduced-traffic points (no reference counterpart): as
 * rr_filter_process_dev, but d_out_f16 receives out_cap {half re, half im} pairs
 * (12 instead of 16 algorithmic bytes per sample); response_f16 != 0 also reads the
 * filter's frequency response from a table rounded to half.  Complex<f32> handles with
 * n = 129 .. 2048 (the overlap-save kernel); RR_ERR_BAD_ARG otherwise. */
int rr_filter_process_dev_f16(rr_filter *h, double sample_rate, const void *d_in,
                              size_t n_in, void *d_out_f16, size_t out_cap, size_t *n_out,
                              int response_f16);
/* Which kernel the last process call ran: 0 k_fir (direct form, any n, any dtype), 1 k_filter_ols
 * (2n-point overlap-save, the reference's recipe), 2 k_filter_blk4096 (4096-point blocks),
 * 3 k_filter_wave (a wave per 1024-sample block; f32, n <= 385, calls of >= 16384 outputs),
 * 4 the blocks of 2^14 .. 2^18 points through the tile transform (n >= 16384), 5 k_filter_blk16k
 * (16384-point blocks in LDS; f32, n = 2049 .. 8192), 6 k_ols4096_f64 (Complex<f64>, n <= 2049, calls of >= 4096 outputs).
 * RR_FILTER_KERNEL=ols4096|fir|parts in the environment (read at design time) keeps the older kernels. */
int rr_filter_last_kernel(const rr_filter *h, int *kernel);
int rr_filter_destroy(rr_filter *h);

/* ------------------------------------------------------------------------ */
/* Downsampler — src/blocks/resampling.rs:14-146                            */
/* ------------------------------------------------------------------------ */
/* Downsampler::with_quality (resampling.rs:45-56).  `output_chunk_len` stays
 * with the host layer, which regroups outputs into chunks (resampling.rs:121-
 * 131); the handle produces the raw decimated stream. */
int rr_downsampler_create(int dtype, double output_rate, double bandwidth,
                          double quality, int device, rr_downsampler **out);
/* Outputs this message will produce (the `pos` schedule of resampling.rs:110-
 * 112 run ahead without consuming anything). */
int rr_downsampler_set_gain(rr_downsampler *h, double gain); /* see rr_filter_set_gain */
int rr_downsampler_peek(rr_downsampler *h, double input_rate, size_t n_in,
                        size_t *n_out);
int rr_downsampler_process(rr_downsampler *h, double input_rate, const void *in,
                           size_t n_in, void *out, size_t out_cap, size_t *n_out);
int rr_downsampler_enqueue(rr_downsampler *h, double input_rate, const void *in,
                           size_t n_in, void *out, size_t out_cap, size_t *n_out);
int rr_downsampler_process_dev(rr_downsampler *h, double input_rate,
                               const void *d_in, size_t n_in, void *d_out,
                               size_t out_cap, size_t *n_out);
int rr_downsampler_ir_len(const rr_downsampler *h, size_t *ir_len);
/* Which kernel the last process call ran: 0 = k_fir (any ratio, any dtype); integer ratios 2, 4, 8
 * in f32 and calls of >= 4096 samples take the chain's fused kernels with an all-ones NCO table
 * (1 k_mix_fir_decim, 2 k_ols_decim4, 3 the wave / workgroup overlap-save kernels: k_ols_wave at 2 and 4 : 1,
 * k_ols_wave2k at 8 : 1, k_ols_wg at the ratios 5 .. 64 with long responses); every other integer ratio P : 1 and rational
 * ratios P : Q with Q <= 8 (both rates integral; e.g. 10 : 1, 8 : 3) take 5 = k_decim_poly (direct form,
 * polyphase LDS layout): same result within rounding (tests: 1e-5 RMS against the f64 oracle).
 * RR_DOWNSAMPLER_GENERIC=1 in the environment keeps k_fir. */
int rr_downsampler_last_kernel(const rr_downsampler *h, int *kernel);
int rr_downsampler_destroy(rr_downsampler *h);

/* ------------------------------------------------------------------------ */
/* Fourier — src/blocks/analysis.rs:26-133                                  */
/* ------------------------------------------------------------------------ */
/* Fourier::new / new_center_dc / with_window / with_window_center_dc
 * (analysis.rs:39-59).  With RR_WIN_SAMPLED the caller supplies the window via
 * rr_fourier_set_sampled_window whenever the chunk length changes. */
int rr_fourier_create(int dtype, const rr_window *window, int center_dc, int device,
                      rr_fourier **out);
int rr_fourier_set_sampled_window(rr_fourier *h, size_t n, const double *window_rel);
/* One Signal::Samples message (analysis.rs:77-121): n_out = n_in, any n_in >= 1. */
int rr_fourier_process(rr_fourier *h, const void *in, size_t n_in, void *out,
                       size_t out_cap, size_t *n_out);
int rr_fourier_enqueue(rr_fourier *h, const void *in, size_t n_in, void *out,
                       size_t out_cap, size_t *n_out);
/* Batched: n_in = k * chunk_len consecutive chunks, each transformed on its own. */
int rr_fourier_process_dev(rr_fourier *h, size_t chunk_len, const void *d_in,
                           size_t n_in, void *d_out, size_t out_cap, size_t *n_out);
int rr_fourier_destroy(rr_fourier *h);

/* ------------------------------------------------------------------------ */
/* Chain — FreqShifter -> Filter -> Downsampler -> Fourier wired as in       */
/* examples/bandwidth_meter/main.rs:51-72, on one device without host hops.  */
/* The stream entering the Filter is cut into chunks of `filter_len`          */
/* (a Rechunker(filter_len), chunks.rs:42-177); the Downsampler's             */
/* output_chunk_len is `fft_len`.                                             */
/* ------------------------------------------------------------------------ */
typedef struct {
    int dtype;              /* RR_F32 (fused fast path) or RR_F64               */
    double precision;       /* FreqShifter precision                            */
    double shift;           /* FreqShifter shift                                */
    size_t filter_len;      /* Filter chunk length = tap count n                */
    double output_rate;     /* Downsampler                                      */
    double bandwidth;
    double quality;
    size_t fft_len;         /* Downsampler output_chunk_len = Fourier length    */
    rr_window fft_window;   /* built-in kinds only                              */
    int center_dc;
    int allow_fused;        /* 0 forces the block-by-block path (for parity)    */
} rr_chain_params;

int rr_chain_create(const rr_chain_params *p, int device, rr_chain **out);
int rr_chain_set_shift(rr_chain *h, double shift);
int rr_chain_filter_needs_design(const rr_chain *h, double sample_rate, int *needed);
int rr_chain_filter_mark_params_changed(rr_chain *h);
int rr_chain_filter_design(rr_chain *h, double sample_rate, const rr_c64 *resp,
                           const double *window_rel);
/* An Event with is_interrupt() travelling down the chain: Filter drops its
 * history (filters.rs:262-265); FreqShifter, Downsampler and Fourier keep their
 * state (transform.rs:357-359, resampling.rs:135-137, analysis.rs:122-124). */
int rr_chain_interrupt(rr_chain *h);
/* Samples the Rechunker in front of the Filter holds (its patchwork, chunks.rs:62-64).  The reference's Rechunker
 * drops them and sends a SamplesLost event - which is an interrupt for the Filter behind it - when an event arrives
 * or the sample rate changes while it holds some (chunks.rs:72-92): the host layer asks here, calls
 * rr_chain_interrupt and emits the event.  rr_chain_filter_design for another sample rate drops them too. */
int rr_chain_pending(const rr_chain *h, size_t *n);
/* Spectra this call will emit for n_in more input samples. */
int rr_chain_peek(rr_chain *h, double sample_rate, size_t n_in, size_t *n_frames);
/* Consumes n_in samples; writes n_frames*fft_len spectrum bins (n_out). */
int rr_chain_process(rr_chain *h, double sample_rate, const void *in, size_t n_in,
                     void *out, size_t out_cap, size_t *n_out);
/* The same without waiting: copies and kernels are queued on the handle's stream
 * (true overlap needs pinned / registered host buffers, rr_host_*); n_out is known at
 * once, `out` is valid after rr_wait(h) or once rr_query(h) returns RR_OK. */
int rr_chain_enqueue(rr_chain *h, double sample_rate, const void *in, size_t n_in,
                     void *out, size_t out_cap, size_t *n_out);
int rr_chain_process_dev(rr_chain *h, double sample_rate, const void *d_in,
                         size_t n_in, void *d_out, size_t out_cap, size_t *n_out);
/* Which kernels the last process call ran: 0 = block-by-block; non-zero = fused
 * (1 direct-form k_mix_fir_decim, 2 overlap-save k_ols_decim4, 3 overlap-save
 * k_ols_wave, each followed by k_fft4096; 4 k_ols_frame: both stages in one kernel; 6 / 7 k_ols_frame / k_ols_wave
 * with the mixer folded into the response tables - NCO periods that divide 8; 8 / 9 k_ols_frame / k_ols_wave with the
 * mixer moved behind the filter - every other NCO period; 5 k_decim_poly(_f64); 11 k_ols4096_f64: Complex<f64> overlap-save). */
/* as rr_stft_set_metering (the rate of the spectra is the chain's output_rate): bandwidth and energy of every spectrum the
 * chain produces, computed in the kernel that makes it (k_ols_frame / k_fft4096 for 4096-point spectra) */
int rr_chain_set_metering(rr_chain *h, double double_percentile, double *d_bandwidth,
                          double *d_energy, size_t cap_frames, int store_spectra);
int rr_chain_last_path(const rr_chain *h, int *fused);
int rr_chain_destroy(rr_chain *h);

/* Measurement aid (no reference counterpart): with timing on, every kernel the
 * chain launches is bracketed by hipEvents on the chain's stream.  Stages:
 * see rr_chain_timing_stage_name(); read returns the accumulated device time
 * and launch count of one stage since the last reset and waits for the events
 * it needs.  on = 1: every stage; on = 2: only the fused mix + FIR + decimate
 * stage (a timed launch costs about 5 us of stream time; the benchmark times only
 * its dominant kernel inside the timed region); on = 0: off. */
int rr_chain_timing_enable(rr_chain *h, int on);
/* With on = 2, time one launch in `every` (default 1 = all of them): a launch that records its own start and end
 * neither overlaps its predecessor's tail nor lets its successor start early; timing every launch of a back-to-back
 * stream was measured to cost it about 1 % of its rate. */
int rr_chain_timing_every(rr_chain *h, unsigned every);
int rr_chain_timing_reset(rr_chain *h);
int rr_chain_timing_read(rr_chain *h, int stage, double *total_ms, uint64_t *launches);
const char *rr_chain_timing_stage_name(int stage); /* NULL past the last stage */

/* ------------------------------------------------------------------------ */
/* Polyphase FFT channelizer (BASELINE configs[2]) = the reference composition */
/*   Rechunker(bins) -> Overlapper(taps_per_branch) -> Fourier::with_window     */
/*   -> every taps_per_branch-th bin                                             */
/* (chunks.rs:42-242, analysis.rs:60-132) as one fold + bins-point FFT per hop. */
/* Input: whole chunks of `bins` samples; each chunk after the first             */
/* taps_per_branch-1 yields one frame of `bins` outputs (critically sampled).    */
/* ------------------------------------------------------------------------ */
typedef struct rr_channelizer rr_channelizer;
int rr_channelizer_create(int dtype, size_t bins, size_t taps_per_branch,
                          const rr_window *window, int device, rr_channelizer **out);
/* The general form: any number of bins (the bins-point transforms run through the Fourier machinery: fast kernels,
 * four-step or Bluestein) and an OVERSAMPLED filterbank with `hop` < bins samples between frames (hop must divide
 * bins * taps_per_branch; 0 = bins) = the composition Rechunker(hop) -> Overlapper(bins * taps_per_branch / hop) ->
 * Fourier::with_window -> every taps_per_branch-th bin.  Input: whole chunks of `hop` samples. */
int rr_channelizer_create_ex(int dtype, size_t bins, size_t taps_per_branch, size_t hop,
                             const rr_window *window, int device, rr_channelizer **out);
/* Any event makes the Overlapper drop its history (chunks.rs:225-233). */
int rr_channelizer_reset(rr_channelizer *h);
int rr_channelizer_peek(const rr_channelizer *h, size_t n_in, size_t *n_out);
int rr_channelizer_process(rr_channelizer *h, const void *in, size_t n_in, void *out,
                           size_t out_cap, size_t *n_out);
int rr_channelizer_process_dev(rr_channelizer *h, const void *d_in, size_t n_in,
                               void *d_out, size_t out_cap, size_t *n_out);
int rr_channelizer_destroy(rr_channelizer *h);

/* ------------------------------------------------------------------------ */
/* ChainBank — K chains with the same parameters whose streams advance in        */
/* LOCKSTEP (the antennas of an array, the sub-bands of a channelized receiver:   */
/* the reference runs one tokio task per block and channel, flow.rs:233-267, at   */
/* chunk sizes of 10^3 .. 10^5 samples, examples/bandwidth_meter/main.rs:56 — a    */
/* size at which a single channel's call on a GPU is launch-bound).  Every         */
/* channel is a full rr_chain with its own state (rr_chainbank_channel hands it    */
/* out; it may be driven alone at any time).  A call gives every channel n_in      */
/* samples: channel k reads d_in + k * in_stride, writes d_out + k * out_stride    */
/* (strides in samples).  While all channels are in the steady fused state at the  */
/* same stream position, the call is TWO launches for all of them (channel =       */
/* blockIdx.y); otherwise (stream start, after an interrupt or a retune, ragged    */
/* calls) the channels run one after the other.  Either way each channel's         */
/* spectra are bit-identical to those of a stand-alone rr_chain fed the same       */
/* samples in the same calls.  n_out: bins per channel.  All channels share the    */
/* bank's stream (rr_set_stream).  Complex<f32>, fft_len 4096 for the lockstep     */
/* step; other parameters run lane by lane.                                        */
/* ------------------------------------------------------------------------ */
typedef struct rr_chainbank rr_chainbank;
int rr_chainbank_create(const rr_chain_params *p, size_t channels, int device,
                        rr_chainbank **out);
int rr_chainbank_channels(const rr_chainbank *h, size_t *channels);
int rr_chainbank_channel(rr_chainbank *h, size_t k, rr_chain **lane); /* owned by the bank */
int rr_chainbank_set_shift(rr_chainbank *h, double shift);
int rr_chainbank_filter_needs_design(const rr_chainbank *h, double sample_rate, int *needed);
int rr_chainbank_filter_mark_params_changed(rr_chainbank *h);
int rr_chainbank_filter_design(rr_chainbank *h, double sample_rate, const rr_c64 *resp,
                               const double *window_rel);
int rr_chainbank_interrupt(rr_chainbank *h);
int rr_chainbank_peek(rr_chainbank *h, double sample_rate, size_t n_in, size_t *n_frames);
int rr_chainbank_process_dev(rr_chainbank *h, double sample_rate, const void *d_in,
                             size_t in_stride, size_t n_in, void *d_out, size_t out_stride,
                             size_t out_cap, size_t *n_out);
/* 1 when the last call ran in lockstep (two launches for all channels), 0: lane by lane */
int rr_chainbank_last_path(const rr_chainbank *h, int *lockstep);
int rr_chainbank_destroy(rr_chainbank *h);

/* ------------------------------------------------------------------------ */
/* Overlapped Fourier analysis: the composition            */
/*   Rechunker(chunk_len) -> Overlapper(chunk_count) -> Fourier::with_window     */
/* (src/blocks/chunks.rs:42-242, analysis.rs:26-133; the wiring of               */
/* examples/bandwidth_meter/main.rs:66-69) on the device without materialising   */
/* the overlapping chunks: every chunk_len new samples one windowed transform    */
/* over the last chunk_count * chunk_len samples, all bins.  Input: any number of */
/* samples (the Rechunker's patchwork is kept in the handle, chunks.rs:62-64);    */
/* output: n_out = frames * chunk_len * chunk_count bins.                        */
/* chunk_len * chunk_count: a power of two (<= 8192 f32, <= 4096 f64) or, for
 * Complex<f32>, any length 32 .. 4096 (Bluestein over the power-of-two kernels). */
/* ------------------------------------------------------------------------ */
typedef struct rr_stft rr_stft;
int rr_stft_create(int dtype, size_t chunk_len, size_t chunk_count, const rr_window *window,
                   int center_dc, int device, rr_stft **out);
/* Any event (and a change of sample rate) makes the Rechunker drop its patchwork
 * and the Overlapper its history (chunks.rs:72-88, 225-233). */
int rr_stft_reset(rr_stft *h);
/* Samples the Rechunker holds (chunks.rs:62-64); see rr_chain_pending. */
int rr_stft_pending(const rr_stft *h, size_t *n);
int rr_stft_peek(const rr_stft *h, size_t n_in, size_t *n_out);
int rr_stft_process(rr_stft *h, const void *in, size_t n_in, void *out, size_t out_cap,
                    size_t *n_out);
int rr_stft_process_dev(rr_stft *h, const void *d_in, size_t n_in, void *d_out,
                        size_t out_cap, size_t *n_out);
/* metering::bandwidth (src/metering.rs:41-80) as the pipeline's LAST STEP, the way the reference's only hot-path caller uses
 * the spectra (examples/bandwidth_meter/main.rs:75-78: `metering::bandwidth(0.01, sample_rate, &chunk)` per Fourier
 * output).  While set, every process call also writes one f64 per produced spectrum to d_bandwidth[0 .. frames) (and the
 * spectrum's energy, sum |X|^2, to d_energy if non-NULL), computed behind the transform while the bins are still in
 * registers (Complex<f32>, 4096-point spectra; other lengths and Complex<f64>: a parallel scan right behind the transform).
 * store_spectra = 0: the spectra themselves are NOT written - d_out may be NULL, out_cap is ignored, n_out still counts
 * them.  d_bandwidth = NULL switches the step off.  More spectra in a call than cap_frames: RR_ERR_CAPACITY.
 * (RR_METER_SERIAL=1 in the environment: the reference's sequential summation order in a kernel of its own, bit-equal to
 *  rr_bandwidth_dev; the fused form differs from it in the last bits of its f64 sums.) */
int rr_stft_set_metering(rr_stft *h, double double_percentile, double sample_rate,
                         double *d_bandwidth, double *d_energy, size_t cap_frames,
                         int store_spectra);
int rr_stft_destroy(rr_stft *h);

/* ------------------------------------------------------------------------ */
/* Meter — the reference's own hot-path caller in ITS order                       */
/* (examples/bandwidth_meter/main.rs:53-69):                                      */
/*   FreqShifter -> Downsampler::with_quality(chunk_len, output_rate, bandwidth,  */
/*   quality) -> Filter (it receives the Downsampler's chunks: chunk_len samples   */
/*   at output_rate) -> Overlapper(overlap) -> Fourier::with_window               */
/* on one device without host hops; output: frames of chunk_len * overlap bins    */
/* (feed them to rr_bandwidth_dev for the example's last step).  Any input length; */
/* any integer or short-period rational decimation runs the fast kernels.          */
/* ------------------------------------------------------------------------ */
typedef struct {
    int dtype;
    double precision;      /* FreqShifter                                        */
    double shift;
    double output_rate;    /* Downsampler                                        */
    double bandwidth;
    double quality;
    size_t chunk_len;      /* Downsampler output_chunk_len = Filter chunk length  */
    size_t overlap;        /* Overlapper chunk count                              */
    rr_window fft_window;  /* built-in kinds only                                 */
    int center_dc;
} rr_meter_params;
typedef struct rr_meter rr_meter;
int rr_meter_create(const rr_meter_params *p, int device, rr_meter **out);
int rr_meter_set_shift(rr_meter *h, double shift);
/* The Filter always sees (output_rate, chunk_len): one design, arguments as for rr_filter_design_taps with
 * n = chunk_len and the closure sampled at i * output_rate / chunk_len. */
int rr_meter_filter_design(rr_meter *h, const rr_c64 *resp, const double *window_rel);
/* An event travelling down the pipeline: an interrupting one resets the Filter (filters.rs:262-265), every event
 * the Overlapper (chunks.rs:225-233; the host layer sends its SamplesLost); the Downsampler's partly filled output
 * chunk stays (resampling.rs:135-137). */
int rr_meter_event(rr_meter *h, int is_interrupt);
int rr_meter_peek(rr_meter *h, double sample_rate, size_t n_in, size_t *n_frames);
int rr_meter_process(rr_meter *h, double sample_rate, const void *in, size_t n_in, void *out,
                     size_t out_cap, size_t *n_out);
int rr_meter_process_dev(rr_meter *h, double sample_rate, const void *d_in, size_t n_in,
                         void *d_out, size_t out_cap, size_t *n_out);
/* 1 when the last process call ran FreqShifter and Downsampler as ONE kernel (Complex<f32>, calls of >= 4096 samples,
 * any integer or short-period rational ratio: k_decim_poly with the phase table riding along), else 0. */
/* as rr_stft_set_metering; the rate of the spectra is the Meter's output_rate */
int rr_meter_set_metering(rr_meter *h, double double_percentile, double *d_bandwidth,
                          double *d_energy, size_t cap_frames, int store_spectra);
/* The example's loop body as ONE call (main.rs:75-78): host samples in, one bandwidth per spectrum out (host, blocking);
 * the spectra are never written anywhere. */
int rr_meter_process_bandwidth(rr_meter *h, double sample_rate, const void *in, size_t n_in,
                               double double_percentile, double *bandwidth_out,
                               size_t cap_frames, size_t *n_frames);
int rr_meter_last_path(const rr_meter *h, int *front_fused);
int rr_meter_destroy(rr_meter *h);

/* ------------------------------------------------------------------------ */
/* Upsampler — src/blocks/resampling.rs:147-280.          */
/* The reference adds every input, scaled by the impulse response, into a ring  */
/* buffer and releases output_rate / input_rate outputs per input (:237-267);   */
/* here each output gathers its inputs in the same order with the same         */
/* roundings (f32 results are bit-equal to the reference's algorithm).  The     */
/* impulse response is redesigned and the ring cleared when the input rate      */
/* changes (:203-236); events pass (:269-271).  Regrouping into chunks of       */
/* output_chunk_len (:251-261) is the caller's, as for the Downsampler.         */
/* ------------------------------------------------------------------------ */
typedef struct rr_upsampler rr_upsampler;
/* Upsampler::with_quality (resampling.rs:179-186); asserts -> RR_ERR_CONTRACT */
int rr_upsampler_create(int dtype, double output_rate, double bandwidth, double quality,
                        int device, rr_upsampler **out);
/* Outputs the next n_in input samples at input_rate will release. */
int rr_upsampler_peek(rr_upsampler *h, double input_rate, size_t n_in, size_t *n_out);
int rr_upsampler_process(rr_upsampler *h, double input_rate, const void *in, size_t n_in,
                         void *out, size_t out_cap, size_t *n_out);
int rr_upsampler_enqueue(rr_upsampler *h, double input_rate, const void *in, size_t n_in,
                         void *out, size_t out_cap, size_t *n_out);
int rr_upsampler_process_dev(rr_upsampler *h, double input_rate, const void *d_in,
                             size_t n_in, void *d_out, size_t out_cap, size_t *n_out);
int rr_upsampler_ir_len(const rr_upsampler *h, size_t *ir_len);
int rr_upsampler_destroy(rr_upsampler *h);
/* The f64 impulse response of resampling.rs:215-234 (ir may be null to query the length). */
int rr_upsampler_design(double input_rate, double output_rate, double bandwidth,
                        double quality, size_t *ir_len, double *ir, size_t cap);

/* ------------------------------------------------------------------------ */
/* FmDemod — src/blocks/modulation.rs:83-158.             */
/* out[t] = (arg(x[t] * conj(x[t-1])) * sample_rate / deviation / TAU, 0); the  */
/* previous sample carries over from call to call; without one (first call, or  */
/* after rr_fmdemod_reset = an interrupting event, :145-149) the first output   */
/* repeats the last output (zero initially, :107).                             */
/* ------------------------------------------------------------------------ */
typedef struct rr_fmdemod rr_fmdemod;
int rr_fmdemod_create(int dtype, double deviation, int device, rr_fmdemod **out);
/* set_deviation / deviation (modulation.rs:163-170); effective from the next process call */
int rr_fmdemod_set_deviation(rr_fmdemod *h, double deviation);
int rr_fmdemod_set_gain(rr_fmdemod *h, double gain); /* see rr_filter_set_gain */
int rr_fmdemod_deviation(const rr_fmdemod *h, double *deviation);
int rr_fmdemod_reset(rr_fmdemod *h);
int rr_fmdemod_process(rr_fmdemod *h, double sample_rate, const void *in, size_t n_in,
                       void *out, size_t out_cap, size_t *n_out);
int rr_fmdemod_enqueue(rr_fmdemod *h, double sample_rate, const void *in, size_t n_in,
                       void *out, size_t out_cap, size_t *n_out);
int rr_fmdemod_process_dev(rr_fmdemod *h, double sample_rate, const void *d_in, size_t n_in,
                           void *d_out, size_t out_cap, size_t *n_out);
int rr_fmdemod_destroy(rr_fmdemod *h);

/* ------------------------------------------------------------------------ */
/* Consumers of the chain output, over batches of frames   */
/* of n samples so that spectra need not leave the device.  `_dev`: device       */
/* pointers, asynchronous on `hip_stream`; the others: host pointers, blocking.  */
/* ------------------------------------------------------------------------ */
/* metering::level (src/metering.rs:21-30): mean |x|^2, f64; out[count] */
int rr_level_dev(int dtype, int device, void *hip_stream, const void *d_frames, size_t n,
                 size_t count, double *d_out);
int rr_level(int dtype, int device, const void *chunk, size_t n, double *out);
/* metering::bandwidth (src/metering.rs:41-80) */
int rr_bandwidth_dev(int dtype, int device, void *hip_stream, double double_percentile,
                     double sample_rate, const void *d_frames, size_t n, size_t count,
                     double *d_out);
int rr_bandwidth(int dtype, int device, double double_percentile, double sample_rate,
                 const void *bins, size_t n, double *out);
/* the same as a workgroup-wide parallel scan (256 lanes per frame, any n; d_energy may be NULL): what the metered
 * pipelines run; differs from rr_bandwidth_dev in the last bits of its f64 sums */
int rr_bandwidth_fast_dev(int dtype, int device, void *hip_stream, double double_percentile,
                          double sample_rate, const void *d_frames, size_t n, size_t count,
                          double *d_bandwidth, double *d_energy);
/* metering::rescale_energy (src/metering.rs:89-109): out = count x resolution Flt */
int rr_rescale_energy_dev(int dtype, int device, void *hip_stream, const void *d_frames,
                          size_t n, size_t count, size_t resolution, void *d_out);
int rr_rescale_energy(int dtype, int device, const void *input, size_t n, size_t resolution,
                      void *output);
/* GainControl's `sample * gain` (src/blocks/transform.rs:62-72), gain cast to Flt */
int rr_gain_dev(int dtype, int device, void *hip_stream, double gain, const void *d_in,
                size_t n, void *d_out);
int rr_gain(int dtype, int device, double gain, const void *in, size_t n, void *out);

/* ------------------------------------------------------------------------ */
/* Synthetic IQ source generated on the device; the test      */
/* harness's stand-in for an SDR source block.  d_out: n rr_c32.             */
/* ------------------------------------------------------------------------ */
int rr_synth_iq_dev(int device, void *hip_stream, uint64_t seed, uint64_t t0,
                    size_t n, void *d_out);

#ifdef __cplusplus
}
#endif
#endif /* RADIORUST_AMD_H */
