/*
 * rr_oracle.h — CPU restatement of the radiorust IQ hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under radiorust_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * What it restates (all file:line relative to /root/reference):
 *   math      src/math.rs:7-49            bessel_I0, kaiser_*, sinc
 *   windows   src/windowing.rs:6-67       Rectangular, Kaiser, CustomWindow
 *   mixer     src/blocks/transform.rs:297-362   FreqShifter
 *   filter    src/blocks/filters.rs:153-277     Filter (overlap-save)
 *   decimator src/blocks/resampling.rs:45-145   Downsampler
 *   fourier   src/blocks/analysis.rs:60-132     Fourier
 *
 * Third-party arithmetic that is NOT in /root/reference and is restated from
 * its published contract: rustfft ^6.0.1 (Cargo.toml:19; forward kernel
 * e^{-j2πkn/N}, inverse = conjugate kernel, both unnormalised, any N) and
 * num ^0.4.0 (Cargo.toml:18; Complex mul/arg, Ratio::new gcd reduction).
 *
 * PARITY PINNING.  The reference is Rust and cannot be built in this image
 * (no cargo/rustc).  Pinned by the reference's own known-answer tests:
 *   Fourier   analysis.rs:139-209 (test_fourier)
 *   bessel_I0 math.rs:55-69, sinc math.rs:70-85
 * Filter, FreqShifter and Downsampler have EMPTY test modules in the
 * reference (filters.rs:378-379, resampling.rs:282-283, transform.rs:393-417
 * covers GainControl only): for those three blocks PARITY IS UNPINNED — the
 * oracle is a line-by-line restatement cross-checked against an independent
 * numpy f64 formulation (oracle/oracle_np.py), nothing more.
 *
 * Two instantiations of the data path: suffix _f32 (Flt = f32, what the GPU
 * computes in) and _f64 (Flt = f64, used as "truth" for RMS error).
 * Compile with -ffp-contract=off: Rust never contracts a*b+c into an fma.
 */
#ifndef RR_ORACLE_H
#define RR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- design math (math.rs) ------------------------------------------- */
double rro_bessel_i0(double x);                         /* math.rs:7-20  */
double rro_kaiser_rel_with_beta(double beta, double x); /* math.rs:26-28 */
double rro_kaiser_alpha_to_beta(double alpha);          /* math.rs:31-33 */
double rro_kaiser_null_at_bin_to_beta(double n);        /* math.rs:37-39 */
double rro_sinc(double x);                              /* math.rs:42-49 */
void rro_deemphasis_factor(double tau, double frequency, double *out); /* filters.rs:20-27, out = {re, im} */

/* ---- windows (windowing.rs) ------------------------------------------ */
enum { RRO_WIN_RECT = 0, RRO_WIN_KAISER = 1, RRO_WIN_CUSTOM = 2 };
typedef struct {
    int kind;
    double beta;                       /* Kaiser */
    double (*fn)(double x, void *ud);  /* CustomWindow closure */
    void *ud;
} rro_window;
double rro_window_value(const rro_window *w, double x); /* windowing.rs:6-67 */

/* ---- FreqShifter ratio (transform.rs:298-302 + num Ratio::new) -------- */
void rro_freq_to_ratio(double sample_rate, double precision, double frequency,
                       int64_t *numer, int64_t *denom);

/* ---- FFT restating the rustfft contract (f64; f32 twins in _impl) ----- */
/* in-place, interleaved re/im; inverse!=0 -> conjugate kernel; unnormalised */
void rro_fft_f64(double *data, size_t n, int inverse);
void rro_fft_f32(float *data, size_t n, int inverse);

/* ---- per-precision block state + run --------------------------------- */
/* the user closure of Filter: freq_resp(bin, freq) -> out[2] = {re, im} */
typedef void (*rro_freq_resp_fn)(int64_t bin, double freq, double *out,
                                 void *ud);
#define RRO_DECL(SUF, FLT)                                                    \
    typedef struct rro_freqshifter_##SUF rro_freqshifter_##SUF;               \
    rro_freqshifter_##SUF *rro_freqshifter_new_##SUF(double precision,        \
                                                     double shift);           \
    void rro_freqshifter_set_shift_##SUF(rro_freqshifter_##SUF *, double);    \
    void rro_freqshifter_process_##SUF(rro_freqshifter_##SUF *, double rate,  \
                                       const FLT *in, size_t n, FLT *out);    \
    size_t rro_freqshifter_table_##SUF(rro_freqshifter_##SUF *, FLT *out,     \
                                       size_t cap);                           \
    void rro_freqshifter_free_##SUF(rro_freqshifter_##SUF *);                 \
    typedef struct rro_filter_##SUF rro_filter_##SUF;                         \
    rro_filter_##SUF *rro_filter_new_##SUF(rro_freq_resp_fn fn, void *ud,     \
                                           const rro_window *w);              \
    void rro_filter_update_##SUF(rro_filter_##SUF *, rro_freq_resp_fn fn,     \
                                 void *ud, const rro_window *w);              \
    size_t rro_filter_process_##SUF(rro_filter_##SUF *, double rate,          \
                                    const FLT *in, size_t n, FLT *out);       \
    void rro_filter_interrupt_##SUF(rro_filter_##SUF *);                      \
    size_t rro_filter_response_##SUF(rro_filter_##SUF *, double *out,         \
                                     size_t cap);                             \
    void rro_filter_free_##SUF(rro_filter_##SUF *);                           \
    typedef struct rro_downsampler_##SUF rro_downsampler_##SUF;               \
    rro_downsampler_##SUF *rro_downsampler_new_##SUF(double output_rate,      \
                                                     double bandwidth,        \
                                                     double quality);         \
    size_t rro_downsampler_process_##SUF(rro_downsampler_##SUF *,             \
                                         double input_rate, const FLT *in,    \
                                         size_t n, FLT *out, size_t cap);     \
    size_t rro_downsampler_ir_##SUF(rro_downsampler_##SUF *, FLT *out,        \
                                    size_t cap);                              \
    void rro_downsampler_free_##SUF(rro_downsampler_##SUF *);                 \
    typedef struct rro_fourier_##SUF rro_fourier_##SUF;                       \
    rro_fourier_##SUF *rro_fourier_new_##SUF(const rro_window *w,             \
                                             int center_dc);                  \
    void rro_fourier_process_##SUF(rro_fourier_##SUF *, const FLT *in,        \
                                   size_t n, FLT *out);                       \
    size_t rro_fourier_window_##SUF(rro_fourier_##SUF *, FLT *out,            \
                                    size_t cap);                              \
    void rro_fourier_free_##SUF(rro_fourier_##SUF *);                         \
    size_t rro_chain_run_##SUF(const FLT *x, size_t n, double fs,             \
                               double precision, double shift,                \
                               size_t filter_len, rro_freq_resp_fn fn,        \
                               void *ud, const rro_window *filter_window,     \
                               double output_rate, double bandwidth,          \
                               double quality, size_t fft_len,                \
                               const rro_window *fft_window, int center_dc,   \
                               FLT *out, size_t out_cap_frames);           \
    /* the same with one thread per block and capacity-1 hand-off (messages  \
     * of `batch` Filter chunks); spectra bit-equal to rro_chain_run's */     \
    size_t rro_chain_run_mt_##SUF(const FLT *x, size_t n, double fs,          \
                                  double precision, double shift,             \
                                  size_t filter_len, rro_freq_resp_fn fn,     \
                                  void *ud, const rro_window *filter_window,  \
                                  double output_rate, double bandwidth,       \
                                  double quality, size_t fft_len,             \
                                  const rro_window *fft_window, int center_dc,\
                                  FLT *out, size_t out_cap_frames,            \
                                  size_t batch);                              \
    double rro_level_##SUF(const FLT *chunk, size_t n);                       \
    double rro_bandwidth_##SUF(double double_percentile, double sample_rate,  \
                               const FLT *bins, size_t n);                    \
    int rro_rescale_energy_##SUF(FLT *output, size_t resolution,              \
                                 const FLT *input, size_t n);                 \
    void rro_gain_##SUF(double gain, const FLT *in, size_t n, FLT *out);      \
    /* Upsampler (resampling.rs:147-280); process returns the output count,   \
     * (size_t)-1 for a contract violation, (size_t)-2 if cap is too small */  \
    typedef struct rro_upsampler_##SUF rro_upsampler_##SUF;                   \
    rro_upsampler_##SUF *rro_upsampler_new_##SUF(double output_rate,          \
                                                 double bandwidth,            \
                                                 double quality);             \
    size_t rro_upsampler_process_##SUF(rro_upsampler_##SUF *,                 \
                                       double input_rate, const FLT *in,      \
                                       size_t n, FLT *out, size_t cap);       \
    size_t rro_upsampler_ir_##SUF(rro_upsampler_##SUF *, FLT *out,            \
                                  size_t cap);                                \
    void rro_upsampler_free_##SUF(rro_upsampler_##SUF *);                     \
    /* FmDemod (modulation.rs:83-158) */                                      \
    typedef struct rro_fmdemod_##SUF rro_fmdemod_##SUF;                       \
    rro_fmdemod_##SUF *rro_fmdemod_new_##SUF(double deviation);               \
    void rro_fmdemod_set_deviation_##SUF(rro_fmdemod_##SUF *, double);        \
    void rro_fmdemod_interrupt_##SUF(rro_fmdemod_##SUF *);                    \
    void rro_fmdemod_process_##SUF(rro_fmdemod_##SUF *, double sample_rate,   \
                                   const FLT *in, size_t n, FLT *out);        \
    void rro_fmdemod_free_##SUF(rro_fmdemod_##SUF *);

RRO_DECL(f32, float)
RRO_DECL(f64, double)

/* ---- counter-based synthetic IQ; host twin of the device
 *      generator.  out: n interleaved complex f32 starting at sample t0 --- */
void rro_synth_iq_f32(uint64_t seed, uint64_t t0, size_t n, float *out);

#ifdef __cplusplus
}
#endif
#endif
