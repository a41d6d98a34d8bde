/*
 * rr_oracle.c — CPU restatement of the radiorust IQ hot path (see rr_oracle.h
 * for scope, citations and the parity-pinning statement).
 *
 * TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#include "rr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------- */
/* math.rs                                                                */
/* ---------------------------------------------------------------------- */

/* math.rs:7-20 — power series, stops when the sum no longer changes or is
 * no longer finite.  `i * i` is an i32 product in the reference (profile.dev
 * has overflow checks off, the series ends long before i reaches 46341). */
double rro_bessel_i0(double x)
{
    double base = x * x / 4.0;
    double addend = 1.0;
    double sum = 1.0;
    for (int i = 1;; i++) {
        addend *= base / (double)(i * i);
        double old = sum;
        sum += addend;
        if (sum == old || !isfinite(sum))
            break;
    }
    return sum;
}

/* math.rs:26-28 */
double rro_kaiser_rel_with_beta(double beta, double x)
{
    return rro_bessel_i0(beta * sqrt(1.0 - x * x));
}

/* math.rs:31-33 */
double rro_kaiser_alpha_to_beta(double alpha)
{
    return alpha * M_PI;
}

/* math.rs:37-39 — note: no factor pi */
double rro_kaiser_null_at_bin_to_beta(double n)
{
    return sqrt(n * n - 1.0);
}

/* math.rs:42-49 */
double rro_sinc(double x)
{
    if (x == 0.0)
        return 1.0;
    double t = x * M_PI;
    return sin(t) / t;
}

/* filters.rs:20-27: Complex { re: 1.0, im: tau * TAU * frequency }.finv(); num-complex 0.4 (Cargo.toml:18):
 * finv(self) = self.conj() / norm / norm with norm = re.hypot(im) */
void rro_deemphasis_factor(double tau, double frequency, double *out)
{
    const double re = 1.0, im = tau * (2.0 * M_PI) * frequency;
    const double norm = hypot(re, im);
    out[0] = re / norm / norm;
    out[1] = -im / norm / norm;
}

/* windowing.rs:14-20 (Rectangular), 24-51 (Kaiser), 58-67 (CustomWindow) */
double rro_window_value(const rro_window *w, double x)
{
    switch (w->kind) {
    case RRO_WIN_RECT:
        return 1.0;
    case RRO_WIN_KAISER:
        return rro_kaiser_rel_with_beta(w->beta, x);
    case RRO_WIN_CUSTOM:
        return w->fn(x, w->ud);
    }
    abort();
}

/* ---------------------------------------------------------------------- */
/* FreqShifter ratio                                                      */
/* ---------------------------------------------------------------------- */

/* `as isize` in Rust saturates and maps NaN to 0 */
static int64_t f64_as_isize(double v)
{
    if (isnan(v))
        return 0;
    if (v >= 9223372036854775807.0)
        return INT64_MAX;
    if (v <= -9223372036854775808.0)
        return INT64_MIN;
    return (int64_t)v;
}

static int64_t gcd_i64(int64_t a, int64_t b)
{
    uint64_t x = a < 0 ? (uint64_t)(-(a + 1)) + 1u : (uint64_t)a;
    uint64_t y = b < 0 ? (uint64_t)(-(b + 1)) + 1u : (uint64_t)b;
    while (y) {
        uint64_t t = x % y;
        x = y;
        y = t;
    }
    return (int64_t)x;
}

/* transform.rs:298-302; Ratio::new (num-rational 0.4): panics on denom == 0,
 * divides both by gcd, keeps the denominator positive. */
void rro_freq_to_ratio(double sample_rate, double precision, double frequency,
                       int64_t *numer, int64_t *denom)
{
    int64_t d = f64_as_isize(round(sample_rate / precision));
    int64_t n = f64_as_isize(round((double)d * frequency / sample_rate));
    if (d == 0)
        abort(); /* "denominator == 0" panic */
    int64_t g = gcd_i64(n, d);
    n /= g;
    d /= g;
    if (d < 0) {
        n = -n;
        d = -d;
    }
    *numer = n;
    *denom = d;
}

/* ---------------------------------------------------------------------- */
/* synthetic IQ                                            */
/* ---------------------------------------------------------------------- */

static uint64_t splitmix64_mix(uint64_t z)
{
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

/* sample t of channel `seed`:
 *   u      = mix(seed * 0x9E3779B97F4A7C15 + t)
 *   noise  = 0.5 * (int32(u>>32), int32(u)) / 2^31          (exact in f64)
 *   tones  = 0.25 e^{+j2π t/16} + 0.25 e^{-j2π 3t/32}        (period 32, f64)
 *   x[t]   = (f32)(noise + tones)           one f64 add, one f64→f32 rounding
 */
void rro_synth_iq_f32(uint64_t seed, uint64_t t0, size_t n, float *out)
{
    double tone[32][2];
    for (int k = 0; k < 32; k++) {
        double a1 = 2.0 * M_PI * (double)(k % 16) / 16.0;
        double a2 = -2.0 * M_PI * (double)((3 * k) % 32) / 32.0;
        tone[k][0] = 0.25 * cos(a1) + 0.25 * cos(a2);
        tone[k][1] = 0.25 * sin(a1) + 0.25 * sin(a2);
    }
    const double sc = 1.0 / 4294967296.0; /* 0.5 / 2^31 */
    for (size_t i = 0; i < n; i++) {
        uint64_t t = t0 + i;
        uint64_t u = splitmix64_mix(seed * 0x9E3779B97F4A7C15ull + t);
        int32_t a = (int32_t)(uint32_t)(u >> 32);
        int32_t b = (int32_t)(uint32_t)(u & 0xFFFFFFFFu);
        out[2 * i] = (float)((double)a * sc + tone[t & 31][0]);
        out[2 * i + 1] = (float)((double)b * sc + tone[t & 31][1]);
    }
}

/* ---------------------------------------------------------------------- */
/* per-precision instantiations                                           */
/* ---------------------------------------------------------------------- */

#define FLT float
#define SUF(name) name##_f32
#define FSIN sinf
#define FCOS cosf
#define FATAN2 atan2f
#include "rr_oracle_impl.inc"
#undef FLT
#undef SUF
#undef FSIN
#undef FCOS
#undef FATAN2

#define FLT double
#define SUF(name) name##_f64
#define FSIN sin
#define FCOS cos
#define FATAN2 atan2
#include "rr_oracle_impl.inc"
