// rr_design.cpp — host-side (f64) design math of the backend: everything the
// reference computes when a block is (re)configured, i.e. outside the hot loop.
//
//   math.rs:7-49          bessel_I0 / kaiser / sinc
//   windowing.rs:6-51     Rectangular, Kaiser
//   transform.rs:298-340  freq_to_ratio + phase table
//   filters.rs:184-225    frequency response -> windowed impulse response
//   resampling.rs:82-99   Kaiser-windowed sinc, unit energy
//   analysis.rs:88-101    mean-square-1 window
//   resampling.rs:110-112 the `pos` decimation schedule
//
// No HIP in this file; it is exported through the C ABI (rr_*_design*, rr_bessel_i0,
// ...) so that the host logic is testable on a machine without a GPU.
#include "rr_internal.hpp"

#include <cmath>
#include <limits>

namespace rr {

// ---- errors ----------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

// ---- math.rs ---------------------------------------------------------------
// Power series of I0; terminates when an added term no longer changes the sum
// (or the sum overflowed), which is the reference's stopping rule (math.rs:14-17).
double bessel_i0(double x) {
    const double q = x * x / 4.0;
    double term = 1.0, total = 1.0;
    for (int k = 1;; ++k) {
        term *= q / static_cast<double>(k * k);
        const double before = total;
        total += term;
        if (total == before || !std::isfinite(total)) return total;
    }
}

double kaiser_rel_with_beta(double beta, double x) { return bessel_i0(beta * std::sqrt(1.0 - x * x)); }

double sinc(double x) {
    if (x == 0.0) return 1.0;
    const double t = x * M_PI;
    return std::sin(t) / t;
}

int window_sample(const rr_window *w, size_t n, double *out) {
    if (!w || (!out && n)) RR_FAIL(RR_ERR_BAD_ARG, "window_sample: null argument");
    const double nf = static_cast<double>(n);
    for (size_t i = 0; i < n; ++i) {
        const double x = 2.0 * (static_cast<double>(i) + 0.5) / nf - 1.0;
        switch (w->kind) {
            case RR_WIN_RECTANGULAR: out[i] = 1.0; break;
            case RR_WIN_KAISER: out[i] = kaiser_rel_with_beta(w->beta, x); break;
            default: RR_FAIL(RR_ERR_BAD_ARG, "window_sample: kind %d is not a built-in window", w->kind);
        }
    }
    return RR_OK;
}

// ---- host FFT (f64, any length) ---------------------------------------------
// Power-of-two lengths: iterative radix-2.  Other lengths: Bluestein's chirp-z
// on top of it.  Unnormalised in both directions (the rustfft contract).
static void fft_pow2(std::vector<cd> &a, bool inverse) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    std::vector<cd> root(n / 2 ? n / 2 : 1);
    for (size_t k = 0; k < n / 2; ++k) {
        const double ang = (inverse ? 2.0 : -2.0) * M_PI * static_cast<double>(k) / static_cast<double>(n);
        root[k] = cd(std::cos(ang), std::sin(ang));
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t stride = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const cd u = a[i + k], v = a[i + k + len / 2] * root[k * stride];
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

void fft_f64(std::vector<cd> &x, bool inverse) {
    const size_t n = x.size();
    if (n <= 1) return;
    if ((n & (n - 1)) == 0) {
        fft_pow2(x, inverse);
        return;
    }
    // Bluestein: X[k] = conj(c[k]) * sum_j (x[j] conj(c[j])) c[k-j],  c[m] = e^{+-j pi m^2 / n}
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    std::vector<cd> chirp(n);
    for (size_t k = 0; k < n; ++k) {
        const unsigned __int128 k2 = static_cast<unsigned __int128>(k) * k;
        const double r = static_cast<double>(static_cast<uint64_t>(k2 % (2 * static_cast<unsigned __int128>(n))));
        const double ang = (inverse ? 1.0 : -1.0) * M_PI * r / static_cast<double>(n);
        chirp[k] = cd(std::cos(ang), std::sin(ang));
    }
    std::vector<cd> a(m, cd(0, 0)), b(m, cd(0, 0));
    for (size_t k = 0; k < n; ++k) a[k] = x[k] * chirp[k];
    b[0] = std::conj(chirp[0]);
    for (size_t k = 1; k < n; ++k) b[k] = b[m - k] = std::conj(chirp[k]);
    fft_pow2(a, false);
    fft_pow2(b, false);
    for (size_t k = 0; k < m; ++k) a[k] *= b[k];
    fft_pow2(a, true);
    const double inv_m = 1.0 / static_cast<double>(m);
    for (size_t k = 0; k < n; ++k) x[k] = a[k] * inv_m * chirp[k];
}

// ---- FreqShifter ---------------------------------------------------------------
static int64_t saturating_i64(double v) {  // Rust `f64 as isize`
    if (std::isnan(v)) return 0;
    if (v >= 9223372036854775807.0) return std::numeric_limits<int64_t>::max();
    if (v <= -9223372036854775808.0) return std::numeric_limits<int64_t>::min();
    return static_cast<int64_t>(v);
}

int freq_to_ratio(double sample_rate, double precision, double shift, int64_t *numer, int64_t *denom) {
    int64_t d = saturating_i64(std::round(sample_rate / precision));
    int64_t n = saturating_i64(std::round(static_cast<double>(d) * shift / sample_rate));
    if (d == 0) RR_FAIL(RR_ERR_CONTRACT, "FreqShifter: sample_rate / precision rounds to 0 (Ratio::new panics)");
    // Euclid on magnitudes; Ratio::new keeps the denominator positive
    uint64_t a = n < 0 ? 0 - static_cast<uint64_t>(n) : static_cast<uint64_t>(n);
    uint64_t b = d < 0 ? 0 - static_cast<uint64_t>(d) : static_cast<uint64_t>(d);
    while (b) {
        const uint64_t r = a % b;
        a = b;
        b = r;
    }
    const int64_t g = static_cast<int64_t>(a);
    n /= g;
    d /= g;
    if (d < 0) {
        n = -n;
        d = -d;
    }
    *numer = n;
    *denom = d;
    return RR_OK;
}

template <class T> static inline T tsin(T);
template <class T> static inline T tcos(T);
template <> inline float tsin(float v) { return sinf(v); }
template <> inline float tcos(float v) { return cosf(v); }
template <> inline double tsin(double v) { return std::sin(v); }
template <> inline double tcos(double v) { return std::cos(v); }

// transform.rs:331-339.  The phase expression is evaluated in T on purpose: the
// reference computes `start_phase + flt!(i) / flt!(denom) * Flt::TAU()` in Flt.
template <class T> void nco_table(int64_t numer, int64_t denom, T start_phase, T *table) {
    const T tau = static_cast<T>(6.283185307179586476925286766559);
    const T den = static_cast<T>(denom);
    int64_t step = 0;
    for (int64_t k = 0; k < denom; ++k) {
        const T phase = start_phase + static_cast<T>(step) / den * tau;
        table[2 * k] = tcos<T>(phase);
        table[2 * k + 1] = tsin<T>(phase);
        step = (step + numer) % denom;  // truncating remainder, like Rust's `%`
    }
}
template void nco_table<float>(int64_t, int64_t, float, float *);
template void nco_table<double>(int64_t, int64_t, double, double *);

// ---- Filter ------------------------------------------------------------------------
int filter_design_taps(size_t n, const rr_c64 *resp, const double *window_rel, cd *taps) {
    if (n == 0) RR_FAIL(RR_ERR_CONTRACT, "Filter: empty chunk");
    if (!resp || !window_rel || !taps) RR_FAIL(RR_ERR_BAD_ARG, "filter_design_taps: null argument");
    const double nf = static_cast<double>(n);
    const double prescale = 2.0 * nf * nf;  // filters.rs:186
    std::vector<cd> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = cd(resp[i].re, resp[i].im) / prescale;
    fft_f64(h, true);
    const size_t half = n / 2;  // filters.rs:201-203 (odd n: last element stays)
    for (size_t i = 0; i < half; ++i) std::swap(h[i], h[i + half]);
    double before = 0.0, after = 0.0;
    for (size_t i = 0; i < n; ++i) {
        before += std::norm(h[i]);
        h[i] *= window_rel[i];
        after += std::norm(h[i]);
    }
    if (!(after > 0.0) || !std::isfinite(before / after))
        RR_FAIL(RR_ERR_CONTRACT, "Filter: window/response leave no energy (scale would be NaN/inf)");
    const double renorm = std::sqrt(before / after);
    // equivalent causal FIR: out[t] = sum_k g[k] x[t-k], g = 2n * h (two unnormalised 2n-point transforms)
    for (size_t i = 0; i < n; ++i) taps[i] = h[i] * renorm * (2.0 * nf);
    return RR_OK;
}

// ---- Downsampler ----------------------------------------------------------------------
int downsampler_design(double input_rate, double output_rate, double bandwidth, double quality,
                       std::vector<double> &ir) {
    if (!(input_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be positive");
    if (!(input_rate >= output_rate))
        RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be greater than or equal to output sample rate");
    const double margin = (output_rate - bandwidth) / 2.0;
    const int64_t len = saturating_i64(std::ceil(input_rate / margin * quality));
    if (len <= 0) RR_FAIL(RR_ERR_CONTRACT, "Downsampler: ir_len must be > 0");
    if (len > (int64_t(1) << 24)) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: impulse response of %lld taps is not supported", (long long)len);
    const size_t L = static_cast<size_t>(len);
    const double Lf = static_cast<double>(L);
    const double nulls = Lf * margin / input_rate;
    const double beta = std::sqrt(nulls * nulls - 1.0);  // Kaiser::with_null_at_bin
    ir.assign(L, 0.0);
    double energy = 0.0;
    for (size_t i = 0; i < L; ++i) {
        const double x = (static_cast<double>(i) + 0.5) - Lf / 2.0;
        const double y = sinc(x * output_rate / input_rate) * kaiser_rel_with_beta(beta, x * 2.0 / Lf);
        ir[i] = y;
        energy += y * y;
    }
    const double gain = 1.0 / std::sqrt(energy);
    for (double &y : ir) y *= gain;
    return RR_OK;
}

// ---- Upsampler (resampling.rs:203-236) --------------------------------------------------------
int upsampler_design(double input_rate, double output_rate, double bandwidth, double quality,
                     std::vector<double> &ir) {
    if (!(input_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be positive");
    if (!(input_rate <= output_rate))
        RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be smaller than or equal to output sample rate");
    if (!(bandwidth < input_rate)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be smaller than input sample rate");
    const double margin = (input_rate - bandwidth) / 2.0;
    const int64_t len = saturating_i64(std::ceil(output_rate / margin * quality));
    if (len <= 0) RR_FAIL(RR_ERR_CONTRACT, "Upsampler: ir_len must be > 0");
    if (len > (int64_t(1) << 24)) RR_FAIL(RR_ERR_BAD_ARG, "Upsampler: impulse response of %lld taps is not supported", (long long)len);
    const size_t L = static_cast<size_t>(len);
    const double Lf = static_cast<double>(L);
    const double nulls = Lf * margin / output_rate;
    const double beta = std::sqrt(nulls * nulls - 1.0);  // Kaiser::with_null_at_bin
    ir.assign(L, 0.0);
    double energy = 0.0;
    for (size_t i = 0; i < L; ++i) {
        const double x = (static_cast<double>(i) + 0.5) - Lf / 2.0;
        const double y = sinc(x * input_rate / output_rate) * kaiser_rel_with_beta(beta, x * 2.0 / Lf);
        ir[i] = y;
        energy += y * y;
    }
    const double gain = 1.0 / std::sqrt(energy);
    for (double &y : ir) y *= gain;
    return RR_OK;
}

// ---- Fourier ----------------------------------------------------------------------------
int fourier_design_window(size_t n, const double *window_rel, double *values) {
    if (n && (!window_rel || !values)) RR_FAIL(RR_ERR_BAD_ARG, "fourier_design_window: null argument");
    double energy = 0.0;
    for (size_t i = 0; i < n; ++i) energy += window_rel[i] * window_rel[i];
    const double gain = std::sqrt(static_cast<double>(n) / energy);
    for (size_t i = 0; i < n; ++i) values[i] = window_rel[i] * gain;
    return RR_OK;
}

// ---- decimation schedule ------------------------------------------------------------------
static bool is_integral(double v) { return std::isfinite(v) && v == std::floor(v) && std::fabs(v) < 4503599627370496.0; }

void Schedule::configure(double in_rate, double out_rate) {
    input_rate = in_rate;
    output_rate = out_rate;
    pos = 0.0;
    integer_ratio = false;
    D = 0;
    phase = 0;
    periodic = false;
    ra = rb = P = Q = 0;
    scale = 1.0;
    // the smallest s with both rates whole multiples of 2^-s; every sum and difference of resampling.rs:110-112 is then a multiple of
    // 2^-s below in + out, exact in f64 while (in + out) 2^s <= 2^53
    int sh = 0;
    while (sh <= 40 && std::isfinite(in_rate) && std::isfinite(out_rate) && out_rate > 0.0 &&
           !(is_integral(std::ldexp(in_rate, sh)) && is_integral(std::ldexp(out_rate, sh))))
        ++sh;
    const double in_s = std::ldexp(in_rate, sh), out_s = std::ldexp(out_rate, sh);
    if (sh <= 40 && is_integral(in_s) && is_integral(out_s) && out_s >= 1.0 && in_s >= out_s && in_s + out_s <= 9007199254740992.0) {
        scale = std::ldexp(1.0, sh);
        const uint64_t a = static_cast<uint64_t>(in_s), b = static_cast<uint64_t>(out_s);
        uint64_t x = a, y = b;
        while (y) {
            const uint64_t t = x % y;
            x = y;
            y = t;
        }
        periodic = true;
        ra = a;
        rb = b;
        P = a / x;
        Q = b / x;
        if (a % b == 0) {
            integer_ratio = true;
            D = a / b;
            phase = D - 1;  // pos = 0: the D-th input triggers the first output
        }
    }
}

// periodic schedules: pos is an integer in [0, ra); after n inputs pos + n rb has crossed ra that many times
static inline unsigned __int128 sched_total(uint64_t pos_units, uint64_t rb, size_t n) {
    return (unsigned __int128)pos_units + (unsigned __int128)rb * n;
}

void Schedule::first_emits(size_t count, int64_t *e) const {
    if (integer_ratio) {
        for (size_t m = 0; m < count; ++m) e[m] = (int64_t)(phase + m * D);
        return;
    }
    // output m is released by the first input t (0-based) with pos + (t + 1) rb >= (m + 1) ra
    const uint64_t p0 = pos_units();
    for (size_t m = 0; m < count; ++m) {
        const unsigned __int128 need = (unsigned __int128)(m + 1) * ra - p0;
        e[m] = (int64_t)((need + rb - 1) / rb) - 1;
    }
}

size_t Schedule::count(size_t n_in) const {
    if (integer_ratio) return n_in > phase ? (n_in - 1 - phase) / D + 1 : 0;
    if (periodic) return (size_t)(sched_total(pos_units(), rb, n_in) / ra);
    double p = pos;
    size_t c = 0;
    for (size_t t = 0; t < n_in; ++t) {
        p += output_rate;
        if (p >= input_rate) {
            p -= input_rate;
            ++c;
        }
    }
    return c;
}

size_t Schedule::advance(size_t n_in, std::vector<uint32_t> *emit) {
    if (integer_ratio) {
        const size_t c = count(n_in);
        if (emit) {
            emit->resize(c);
            for (size_t m = 0; m < c; ++m) (*emit)[m] = static_cast<uint32_t>(phase + m * D);
        }
        // k = inputs seen since the last emit; phase = D-1-k
        const uint64_t k = (D - 1 - phase + n_in) % D;
        phase = D - 1 - k;
        pos = static_cast<double>(k) * output_rate;
        return c;
    }
    if (periodic && !emit) {
        const unsigned __int128 tot = sched_total(pos_units(), rb, n_in);
        pos = static_cast<double>((uint64_t)(tot % ra)) / scale;
        return (size_t)(tot / ra);
    }
    if (emit) emit->clear();
    size_t c = 0;
    for (size_t t = 0; t < n_in; ++t) {  // resampling.rs:110-112, verbatim arithmetic
        pos += output_rate;
        if (pos >= input_rate) {
            pos -= input_rate;
            if (emit) emit->push_back(static_cast<uint32_t>(t));
            ++c;
        }
    }
    return c;
}

// ---- interpolation schedule -------------------------------------------------------------------
void UpSchedule::configure(double in_rate, double out_rate) {
    input_rate = in_rate;
    output_rate = out_rate;
    pos = 0.0;
    integer_ratio = false;
    U = 0;
    closed = false;
    ra = rb = 0;
    scale = 1.0;
    if (is_integral(in_rate) && is_integral(out_rate) && in_rate >= 1.0) {
        const uint64_t a = static_cast<uint64_t>(in_rate), b = static_cast<uint64_t>(out_rate);
        if (b % a == 0) {
            integer_ratio = true;  // pos returns to 0 after every input
            U = b / a;
        }
    }
    int sh = 0;
    while (sh <= 40 && std::isfinite(in_rate) && std::isfinite(out_rate) && in_rate > 0.0 && out_rate > 0.0 &&
           !(is_integral(std::ldexp(in_rate, sh)) && is_integral(std::ldexp(out_rate, sh))))
        ++sh;
    const double in_s = std::ldexp(in_rate, sh), out_s = std::ldexp(out_rate, sh);
    if (sh <= 40 && is_integral(in_s) && is_integral(out_s) && in_s >= 1.0 && out_s >= 1.0 && in_s + out_s <= 9007199254740992.0) {
        closed = true;
        scale = std::ldexp(1.0, sh);
        ra = static_cast<uint64_t>(in_s);
        rb = static_cast<uint64_t>(out_s);
    }
}

// closed form: the outputs released by the next n inputs = ceil((n rb - pos) / ra)
static inline size_t up_closed_count(uint64_t p0, uint64_t ra, uint64_t rb, size_t n) {
    const unsigned __int128 tot = (unsigned __int128)rb * n;
    if (tot <= p0) return 0;
    return (size_t)((tot - p0 + ra - 1) / ra);
}

size_t UpSchedule::count(size_t n_in) const {
    if (integer_ratio) return n_in * U;
    if (closed) return up_closed_count(pos_units(), ra, rb, n_in);
    double p = pos;
    size_t c = 0;
    for (size_t t = 0; t < n_in; ++t) {
        while (p < output_rate) {
            ++c;
            p += input_rate;
        }
        p -= output_rate;
    }
    return c;
}

size_t UpSchedule::advance(size_t n_in, std::vector<int32_t> *before) {
    if (before) before->resize(n_in);
    if (integer_ratio) {
        if (before)
            for (size_t t = 0; t < n_in; ++t) (*before)[t] = static_cast<int32_t>(t * U);
        return n_in * U;
    }
    if (closed && !before) {
        const uint64_t p0 = pos_units();
        const size_t c = up_closed_count(p0, ra, rb, n_in);
        // pos + c ra - n rb, in [0, ra)
        const unsigned __int128 np = (unsigned __int128)p0 + (unsigned __int128)ra * c - (unsigned __int128)rb * n_in;
        pos = static_cast<double>(static_cast<uint64_t>(np)) / scale;
        return c;
    }
    size_t c = 0;
    for (size_t t = 0; t < n_in; ++t) {  // resampling.rs:248-265, verbatim arithmetic
        if (before) (*before)[t] = static_cast<int32_t>(c);
        while (pos < output_rate) {
            ++c;
            pos += input_rate;
        }
        pos -= output_rate;
    }
    return c;
}

}  // namespace rr
