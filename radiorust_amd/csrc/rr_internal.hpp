// rr_internal.hpp — shared declarations of the MI355X IQ-stream backend.
// Host side is C++17; device code lives in rr_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <complex>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/radiorust_amd.h"

namespace rr {

// ---- errors --------------------------------------------------------------
void set_error(const char *fmt, ...);
const char *last_error();

#define RR_FAIL(code, ...)          \
    do {                            \
        ::rr::set_error(__VA_ARGS__); \
        return (code);              \
    } while (0)

#define RR_HIP(expr)                                                          \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            ::rr::set_error("%s failed: %s (%s:%d)", #expr,                   \
                            hipGetErrorString(e_), __FILE__, __LINE__);       \
            return RR_ERR_HIP;                                                \
        }                                                                     \
    } while (0)

#define RR_TRY(expr)              \
    do {                          \
        int s_ = (expr);          \
        if (s_ != RR_OK) return s_; \
    } while (0)

inline size_t elem_size(int dtype) { return dtype == RR_F64 ? 16 : 8; }  // complex

// ---- device memory -------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;  // bytes
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // grow-only; contents are NOT preserved
    int reserve(size_t bytes) {
        if (bytes <= cap) return RR_OK;
        release();
        size_t want = bytes + bytes / 8 + 256;
        RR_HIP(hipMalloc(&p, want));
        cap = want;
        return RR_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// ---- host design math (rr_design.cpp) --------------------------------------
using cd = std::complex<double>;
double bessel_i0(double x);
double kaiser_rel_with_beta(double beta, double x);
double sinc(double x);
void fft_f64(std::vector<cd> &x, bool inverse);  // any n, unnormalised
int freq_to_ratio(double sample_rate, double precision, double shift, int64_t *numer, int64_t *denom);
template <class T> void nco_table(int64_t numer, int64_t denom, T start_phase, T *table /*2*denom*/);
int window_sample(const rr_window *w, size_t n, double *out);
int filter_design_taps(size_t n, const rr_c64 *resp, const double *window_rel, cd *taps);
int downsampler_design(double input_rate, double output_rate, double bandwidth, double quality,
                       std::vector<double> &ir);
int fourier_design_window(size_t n, const double *window_rel, double *values);
int upsampler_design(double input_rate, double output_rate, double bandwidth, double quality,
                     std::vector<double> &ir);

// The decimation schedule of resampling.rs:110-112, run on the host.
struct Schedule {
    double input_rate = 0, output_rate = 0;
    double pos = 0;
    // integer fast path: both rates integral and input_rate % output_rate == 0
    bool integer_ratio = false;
    uint64_t D = 0;      // input_rate / output_rate
    uint64_t phase = 0;  // inputs still to consume before the next emit, minus 1... see .cpp
    void configure(double in_rate, double out_rate);
    // Advances over n_in inputs.  If `emit` is non-null it receives the 0-based
    // input index after which each output is emitted.  Returns the count.
    size_t advance(size_t n_in, std::vector<uint32_t> *emit);
    size_t count(size_t n_in) const;  // like advance() without changing state
    // integer path only: index of the first emit for the next n_in inputs
    uint64_t first_emit() const { return phase; }
    // Both rates integral (every f64 operation of resampling.rs:110-112 is then exact): the schedule is periodic,
    // every P = in / g inputs release Q = out / g outputs (g = gcd).  count() and advance() without an emit list run
    // in closed form; first_emits() gives the 0-based input indices that trigger the next `count` outputs.
    // The same holds for rates that are whole multiples of 2^-s as long as (in + out) 2^s <= 2^53 (48000 -> 44100.5: s = 1):
    // ra, rb are the rates times `scale` = 2^s, `pos` stays in the reference's units (pos_units() = pos * scale, exact).
    bool periodic = false;
    uint64_t ra = 0, rb = 0, P = 0, Q = 0;
    double scale = 1.0;
    uint64_t pos_units() const { return static_cast<uint64_t>(pos * scale); }
    void first_emits(size_t count, int64_t *e) const;
};

// The interpolation schedule of resampling.rs:248-265, run on the host: how many outputs each
// input sample releases.
struct UpSchedule {
    double input_rate = 0, output_rate = 0;
    double pos = 0;
    bool integer_ratio = false;  // both rates integral and output_rate % input_rate == 0
    uint64_t U = 0;              // output_rate / input_rate
    // Both rates whole multiples of 2^-s with (in + out) 2^s <= 2^53: every sum of resampling.rs:252-265 is exact in f64 and the
    // schedule has a closed form - with pos in [0, ra) in units of 2^-s, the outputs released before input t of a call are
    // before[t] = ceil((t rb - pos) / ra)  (the smallest c with pos + c ra - t rb >= 0), ra / rb = the rates times `scale`.
    bool closed = false;
    uint64_t ra = 0, rb = 0;
    double scale = 1.0;
    uint64_t pos_units() const { return static_cast<uint64_t>(pos * scale); }
    void configure(double in_rate, double out_rate);
    size_t count(size_t n_in) const;
    // Advances over n_in inputs; if `before` is non-null it receives, per input, the number of
    // outputs of this call released before that input was added.  Returns the output count.
    size_t advance(size_t n_in, std::vector<int32_t> *before);
};

// ---- block base ------------------------------------------------------------
enum Kind { K_FREQSHIFTER = 1, K_FILTER, K_DOWNSAMPLER, K_FOURIER, K_CHAIN, K_CHANNELIZER, K_UPSAMPLER, K_FMDEMOD, K_STFT, K_METER, K_CHAINBANK };

}  // namespace rr

struct rr_block {
    int kind = 0;
    int dtype = RR_F32;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the one in use
    rr::DevBuf stage_in, stage_out;  // for host-pointer entry points
    virtual ~rr_block();
    int init_base(int kind_, int dtype_, int device_);
    int select() const;  // hipSetDevice
};
