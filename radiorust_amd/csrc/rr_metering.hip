// rr_metering.hip — consumers of the chain's output that the reference computes on the
// host per spectrum: metering::{level, bandwidth, rescale_energy} (src/metering.rs:21-109)
// and the GainControl block's multiply (src/blocks/transform.rs:62-72), as device
// functions over batches of frames so that spectra need not leave the GPU.
//
// The reference accumulates in f64 in index order (and `bandwidth` scans with an early
// exit), so one lane per frame does the sequential part over energies staged in LDS by
// the whole workgroup; no a*b+c contraction anywhere (the reference is Rust).
#include <cstdlib>
#include "rr_kernels.hpp"
#include "rr_meter_dev.hpp"

namespace rr {

template <class T> struct V2m;
template <> struct V2m<float> { using type = float2; };
template <> struct V2m<double> { using type = double2; };

__device__ __forceinline__ float norm_sqr_rn(float2 v) { return __fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y)); }
__device__ __forceinline__ double norm_sqr_rn(double2 v) { return __dadd_rn(__dmul_rn(v.x, v.x), __dmul_rn(v.y, v.y)); }

constexpr int kMeterMaxN = 8192;  // energies of one frame in LDS as f64

// mode 0: level (metering.rs:21-30); mode 1: bandwidth (metering.rs:41-80)
template <class T>
__global__ __launch_bounds__(256) void k_meter(const typename V2m<T>::type *__restrict__ frames, int n, int mode,
                                               double double_percentile, double sample_rate,
                                               double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *e = reinterpret_cast<double *>(smem);
    const typename V2m<T>::type *src = frames + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) e[i] = (double)norm_sqr_rn(src[i]);
    __syncthreads();
    if (threadIdx.x != 0) return;
    double total = 0.0;
    for (int i = 0; i < n; ++i) total = __dadd_rn(total, e[i]);
    if (mode == 0) {
        out[blockIdx.x] = total / (double)n;
        return;
    }
    if (mode == 2) {  // the sum itself (the energy the fused epilogues report beside the bandwidth)
        out[blockIdx.x] = total;
        return;
    }
    const double limit = __dmul_rn(total, double_percentile) / 2.0;
    const int wrap = (n + 1) / 2;
    double used_bins = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        double old_e = 0.0, used = 0.0;
        for (int k = 0; k < n; ++k) {
            const int pos = pass == 0 ? k : n - 1 - k;
            const int idx = pos < n - wrap ? wrap + pos : pos - (n - wrap);
            const double new_e = __dadd_rn(old_e, e[idx]);
            if (new_e > limit) {
                used = __dadd_rn(used, __dsub_rn(limit, old_e) / __dsub_rn(new_e, old_e));
                break;
            }
            used = __dadd_rn(used, 1.0);
            old_e = new_e;
        }
        used_bins = __dadd_rn(used_bins, used);
    }
    const double bw = __dmul_rn(__dsub_rn((double)n, used_bins), sample_rate) / (double)n;
    out[blockIdx.x] = bw > 0.0 ? bw : 0.0;
}

// metering::bandwidth for frames of ANY length as a workgroup-wide parallel scan (rr_meter_dev.hpp): what the metered entry
// points run behind a transform that has no fused epilogue (lengths other than 4096, Complex<f64>), and rr_bandwidth_fast_dev.
// The energies are recomputed from the bins where they are needed (each bin is read two or three times, from L1 / L2).
template <class T>
__global__ __launch_bounds__(256) void k_bandwidth_par(const typename V2m<T>::type *__restrict__ frames, int n, double double_percentile,
                                                       double sample_rate, double *__restrict__ bw_out, double *__restrict__ energy_out) {
    __shared__ double scratch[kBwScratch];
    const typename V2m<T>::type *src = frames + (size_t)blockIdx.x * n;
    const int wrap = (n + 1) / 2;
    auto e_at = [&](int s) -> double {
        const int idx = s < n - wrap ? wrap + s : s - (n - wrap);
        return (double)norm_sqr_rn(src[idx]);
    };
    double total;
    const double bw = bandwidth_block256<0>(n, (int)threadIdx.x, e_at, double_percentile, sample_rate, scratch, &total);
    if (threadIdx.x == 0) {
        bw_out[blockIdx.x] = bw;
        if (energy_out) energy_out[blockIdx.x] = total;
    }
}

int launch_bandwidth_par(int dtype, hipStream_t s, double double_percentile, double sample_rate, const void *frames, size_t n,
                         size_t count, double *bw_out, double *energy_out) {
    if (count == 0) return RR_OK;
    if (n == 0) RR_FAIL(RR_ERR_CONTRACT, "metering: empty chunk");
    if (n > 0x3fffffffull || count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "metering: size out of range");
    if (dtype == RR_F32)
        hipLaunchKernelGGL(k_bandwidth_par<float>, dim3((unsigned)count), dim3(256), 0, s, (const float2 *)frames, (int)n,
                           double_percentile, sample_rate, bw_out, energy_out);
    else
        hipLaunchKernelGGL(k_bandwidth_par<double>, dim3((unsigned)count), dim3(256), 0, s, (const double2 *)frames, (int)n,
                           double_percentile, sample_rate, bw_out, energy_out);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ double div_rn(double a, double b) { return __ddiv_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }

// metering.rs:89-109, one lane per output value, all arithmetic in Flt
template <class T>
__global__ __launch_bounds__(256) void k_rescale_energy(const typename V2m<T>::type *__restrict__ frames, int n,
                                                        int resolution, T *__restrict__ out) {
    const int oi = blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= resolution) return;
    const typename V2m<T>::type *src = frames + (size_t)blockIdx.y * n;
    const T left = mul_rn(div_rn((T)oi, (T)resolution), (T)n);
    const T right = mul_rn(div_rn(add_rn((T)oi, (T)1), (T)resolution), (T)n);
    long lf = (long)floor((double)left);
    if (lf > n - 1) lf = n - 1;
    long rc = (long)ceil((double)right);
    if (rc > n) rc = n;
    T acc = 0;
    for (long ii = lf; ii < rc; ++ii) {
        const T lb = (T)ii > left ? (T)ii : left;
        const T up = add_rn((T)ii, (T)1);
        const T rb = up < right ? up : right;
        acc = add_rn(acc, mul_rn(norm_sqr_rn(src[ii]), sub_rn(rb, lb)));
    }
    out[(size_t)blockIdx.y * resolution + oi] = acc;
}

template <class T>
__global__ __launch_bounds__(256) void k_gain(const T *__restrict__ in, T *__restrict__ out, size_t n2, T g) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) out[i] = mul_rn(in[i], g);
}

int launch_meter(int dtype, hipStream_t s, int mode, double double_percentile, double sample_rate, const void *frames,
                 size_t n, size_t count, double *out) {
    if (count == 0) return RR_OK;
    if (n == 0) RR_FAIL(RR_ERR_CONTRACT, "metering: empty chunk");
    if (n > (size_t)kMeterMaxN) RR_FAIL(RR_ERR_BAD_ARG, "metering: frames longer than %d bins are not supported yet", kMeterMaxN);
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "metering: too many frames");
    const size_t lds = n * sizeof(double);
    if (dtype == RR_F32) {
        auto fn = k_meter<float>;
        RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)count), dim3(256), lds, s, (const float2 *)frames, (int)n, mode,
                           double_percentile, sample_rate, out);
    } else {
        auto fn = k_meter<double>;
        RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)count), dim3(256), lds, s, (const double2 *)frames, (int)n, mode,
                           double_percentile, sample_rate, out);
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_rescale_energy(int dtype, hipStream_t s, const void *frames, size_t n, size_t count, size_t resolution,
                          void *out) {
    if (count == 0 || resolution == 0) return RR_OK;
    if (n == 0) RR_FAIL(RR_ERR_CONTRACT, "rescale_energy: assert!(n > 0)");
    if (count > 65535 || n > 0x7fffffffull || resolution > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "rescale_energy: size out of range");
    dim3 grid((unsigned)((resolution + 255) / 256), (unsigned)count);
    if (dtype == RR_F32)
        hipLaunchKernelGGL(k_rescale_energy<float>, grid, dim3(256), 0, s, (const float2 *)frames, (int)n, (int)resolution, (float *)out);
    else
        hipLaunchKernelGGL(k_rescale_energy<double>, grid, dim3(256), 0, s, (const double2 *)frames, (int)n, (int)resolution, (double *)out);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Upsampler (resampling.rs:237-267), gather form, one output per lane.  The sums run in the
// reference's order (ascending input index; product rounded, then added), so f32 results are
// bit-equal to the scatter-add ring buffer of the reference.
// ---------------------------------------------------------------------------
template <class T, class CT>
__global__ __launch_bounds__(256) void k_upsample(const CT *__restrict__ hist, long hn, const CT *__restrict__ in,
                                                  long n_in, const T *__restrict__ ir, int L, long U,
                                                  const int *__restrict__ before, CT *__restrict__ out,
                                                  long n_out) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_out) return;
    T ar = 0, ai = 0;
    if (U > 0) {
        const long t_hi = m / U;  // the input that releases output m
        const int off0 = (int)(m - t_hi * U);
        const int jmax = (L - 1 - off0) / (int)U;  // contributions t_hi - jmax .. t_hi
        for (int j = jmax; j >= 0; --j) {
            const long t = t_hi - j;
            CT x;
            x.x = 0;
            x.y = 0;
            if (t >= 0)
                x = in[t];
            else if (t >= -hn)
                x = hist[hn + t];
            const T c = ir[off0 + (int)U * j];
            ar = add_rn(ar, mul_rn(x.x, c));
            ai = add_rn(ai, mul_rn(x.y, c));
        }
    } else {
        // largest virtual index v (0 .. hn + n_in - 1) with before[v] <= m
        long lo = 0, hi = hn + n_in - 1;
        while (lo < hi) {
            const long mid = (lo + hi + 1) >> 1;
            if ((long)before[mid] <= m)
                lo = mid;
            else
                hi = mid - 1;
        }
        long v0 = lo;
        while (v0 > 0 && m - (long)before[v0 - 1] < L) --v0;
        for (long v = v0; v <= lo; ++v) {
            const long off = m - (long)before[v];
            if (off < 0 || off >= L) continue;
            const CT x = v >= hn ? in[v - hn] : hist[v];
            const T c = ir[off];
            ar = add_rn(ar, mul_rn(x.x, c));
            ai = add_rn(ai, mul_rn(x.y, c));
        }
    }
    CT o;
    o.x = ar;
    o.y = ai;
    out[m] = o;
}

// Any pair of rates on a 2^-s grid (44 100 -> 48 000: ra : rb = 147 : 160), closed form: input t of the call has released
// before[t] = ceil((t rb - pos) / ra) outputs when it is added (UpSchedule), so output m sums over the inputs v with
// 0 <= m - before[v] < L, i.e. from v0 = floor(((m - L) ra + pos) / rb) + 1 to t_hi = floor((m ra + pos) / rb), in ascending order
// (the reference's order of additions), with before[v + 1] = before[v] + rb div ra + carry by whole-number steps.  No list of
// the call's length from the host (the gather form above took one: a host loop over the samples, an upload and a stream
// synchronisation per call), no search.
template <class T, class CT>
__global__ __launch_bounds__(256) void k_upsample_closed(const CT *__restrict__ hist, long hn, const CT *__restrict__ in,
                                                         const T *__restrict__ ir, int L, long ra, long rb, long pos0,
                                                         CT *__restrict__ out, long n_out) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_out) return;
    auto floordiv = [](long a, long b) -> long {  // b > 0
        long q = a / b;
        if (a % b < 0) --q;
        return q;
    };
    const long t_hi = floordiv(m * ra + pos0, rb);
    long v = floordiv((m - L) * ra + pos0, rb) + 1;
    if (v < -hn) v = -hn;  // (older inputs are not kept: their responses have ended)
    T ar = 0, ai = 0;
    if (v <= t_hi) {
        const long q0 = v * rb - pos0;
        const long B = -floordiv(-q0, ra);  // ceil(q0 / ra) = before[v]
        // from here on in 32 bits: the tap index off = m - before[v] (0 <= off < L for every v of the range, falling), the remainder
        // e = ra before[v] - (v rb - pos) in [0, ra), and before[v + 1] = before[v] + rb div ra + (rb mod ra > e)
        int off = (int)(m - B);
        unsigned e = (unsigned)(B * ra - q0);
        const int kq = (int)(rb / ra);
        const unsigned kr = (unsigned)(rb % ra), ura = (unsigned)ra;
        auto run = [&](const CT *__restrict__ x, int n) {
            for (int i = 0; i < n; ++i) {
                const CT s = x[i];
                const T c = ir[off];
                ar = add_rn(ar, mul_rn(s.x, c));
                ai = add_rn(ai, mul_rn(s.y, c));
                const bool carry = kr > e;
                off -= kq + (carry ? 1 : 0);
                e = carry ? e + (ura - kr) : e - kr;
            }
        };
        // the kept inputs (v < 0) first, then the call's own: ascending, the reference's order of additions
        const long h_end = t_hi < -1 ? t_hi : -1;
        if (v <= h_end) {
            run(hist + (hn + v), (int)(h_end - v + 1));
            v = h_end + 1;
        }
        if (v <= t_hi) run(in + v, (int)(t_hi - v + 1));
    }
    CT o;
    o.x = ar;
    o.y = ai;
    out[m] = o;
}

// The same with the tile's inputs and the taps staged in LDS: a workgroup's 256 outputs read the inputs v_lo .. v_hi (the first
// output's oldest to the last output's newest: ~ (256 + L) ra / rb of them) and the L taps - two LDS reads per product where the
// form above asks L1 twice (44 100 -> 48 000: 0.82 -> 0.43 ms per 2^24 input samples).  The sums in the same order.
template <class T, class CT>
__global__ __launch_bounds__(256) void k_upsample_closed_lds(const CT *__restrict__ hist, long hn, const CT *__restrict__ in,
                                                             const T *__restrict__ ir, int L, long ra, long rb, long pos0,
                                                             CT *__restrict__ out, long n_out, int nx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char up_lds[];
    CT *const xs = reinterpret_cast<CT *>(up_lds);       // nx inputs from v_lo on
    T *const irs = reinterpret_cast<T *>(xs + nx);       // the L taps
    auto floordiv = [](long a, long b) -> long {  // b > 0
        long q = a / b;
        if (a % b < 0) --q;
        return q;
    };
    const long m0 = (long)blockIdx.x * 256, m1 = (m0 + 255 < n_out ? m0 + 255 : n_out - 1);
    long v_lo = floordiv((m0 - L) * ra + pos0, rb) + 1;
    if (v_lo < -hn) v_lo = -hn;
    const long v_hi = floordiv(m1 * ra + pos0, rb);
    for (long i = threadIdx.x; v_lo + i <= v_hi && i < nx; i += 256) {
        const long v = v_lo + i;
        xs[i] = v >= 0 ? in[v] : hist[hn + v];
    }
    for (int i = threadIdx.x; i < L; i += 256) irs[i] = ir[i];
    __syncthreads();
    const long m = m0 + threadIdx.x;
    if (m >= n_out) return;
    const long t_hi = floordiv(m * ra + pos0, rb);
    long v = floordiv((m - L) * ra + pos0, rb) + 1;
    if (v < -hn) v = -hn;
    T ar = 0, ai = 0;
    if (v <= t_hi) {
        const long q0 = v * rb - pos0;
        const long B = -floordiv(-q0, ra);
        int off = (int)(m - B);
        unsigned e = (unsigned)(B * ra - q0);
        const int kq = (int)(rb / ra);
        const unsigned kr = (unsigned)(rb % ra), ura = (unsigned)ra;
        const CT *x = xs + (v - v_lo);
        const int n = (int)(t_hi - v + 1);
        for (int i = 0; i < n; ++i) {
            const CT sx = x[i];
            const T c = irs[off];
            ar = add_rn(ar, mul_rn(sx.x, c));
            ai = add_rn(ai, mul_rn(sx.y, c));
            const bool carry = kr > e;
            off -= kq + (carry ? 1 : 0);
            e = carry ? e + (ura - kr) : e - kr;
        }
    }
    CT o;
    o.x = ar;
    o.y = ai;
    out[m] = o;
}

int launch_upsample_closed(int dtype, hipStream_t s, const void *hist, size_t hn, const void *in, const void *ir, size_t L,
                           uint64_t ra, uint64_t rb, uint64_t pos0, void *out, size_t n_out) {
    if (n_out == 0) return RR_OK;
    if (ra == 0 || rb == 0 || ra >= (1ull << 31) || rb >= (1ull << 31) || n_out >= (1ull << 31) || L >= (1ull << 30))
        RR_FAIL(RR_ERR_BAD_ARG, "Upsampler: closed-form schedule out of range");
    const unsigned blocks = (unsigned)((n_out + 255) / 256);
    {
        // the staged form where the tile's inputs and the taps fit 48 KiB of LDS (RR_UPSAMPLER_LDS=0: the plain gather)
        const char *e = std::getenv("RR_UPSAMPLER_LDS");
        const size_t esz = dtype == RR_F32 ? 8 : 16, tsz = esz / 2;
        const size_t nx = ((256 + L) * ra + rb - 1) / rb + 4;
        const size_t lds = nx * esz + L * tsz;
        if (!(e && std::atoi(e) == 0) && lds <= 48 * 1024) {
            if (dtype == RR_F32)
                hipLaunchKernelGGL((k_upsample_closed_lds<float, float2>), dim3(blocks), dim3(256), lds, s, (const float2 *)hist, (long)hn,
                                   (const float2 *)in, (const float *)ir, (int)L, (long)ra, (long)rb, (long)pos0, (float2 *)out, (long)n_out,
                                   (int)nx);
            else
                hipLaunchKernelGGL((k_upsample_closed_lds<double, double2>), dim3(blocks), dim3(256), lds, s, (const double2 *)hist,
                                   (long)hn, (const double2 *)in, (const double *)ir, (int)L, (long)ra, (long)rb, (long)pos0, (double2 *)out,
                                   (long)n_out, (int)nx);
            RR_HIP(hipGetLastError());
            return RR_OK;
        }
    }
    if (dtype == RR_F32)
        hipLaunchKernelGGL((k_upsample_closed<float, float2>), dim3(blocks), dim3(256), 0, s, (const float2 *)hist, (long)hn,
                           (const float2 *)in, (const float *)ir, (int)L, (long)ra, (long)rb, (long)pos0, (float2 *)out, (long)n_out);
    else
        hipLaunchKernelGGL((k_upsample_closed<double, double2>), dim3(blocks), dim3(256), 0, s, (const double2 *)hist, (long)hn,
                           (const double2 *)in, (const double *)ir, (int)L, (long)ra, (long)rb, (long)pos0, (double2 *)out,
                           (long)n_out);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// Integer ratios U = 2 .. 16, f32: a lane produces the U outputs that one input releases (m = U t + p, p < U).
// They all sum over the same inputs t - jmax .. t, so a workgroup stages its 256 + J inputs
// in LDS once (the taps come as scalar loads) and every input read serves U outputs (the one-output-per-lane form above reads 30 inputs
// and 30 taps from L1/L2 per output: 1.38 ms for 2^24 inputs at U = 4).  Per output the same products in
// the same order, each product rounded and then added (-ffp-contract=off: packed multiply, packed add):
// still bit-equal to the reference's scatter-add.
constexpr int kUpJmax = 512;  // inputs in front of a tile that can reach into it: (L - 1) / U
template <int U>
__global__ __launch_bounds__(256) void k_upsample_int(const float2 *__restrict__ hist, long hn,
                                                      const float2 *__restrict__ in, long n_in,
                                                      const float *__restrict__ ir, int L, float2 *__restrict__ out,
                                                      long n_out) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char up_smem[];
    const int J = (L - 1) / U;  // the oldest input of output U t is t - J
    f2v *xs = reinterpret_cast<f2v *>(up_smem);  // inputs tile0 - J .. tile0 + 255
    const long tile0 = (long)blockIdx.x * 256;
    static_assert(256 + kUpJmax <= 3 * 256, "three loads per lane cover the tile");
    // (all of a lane's loads - up to 3: J <= kUpJmax - requested before the first is stored: one round trip per tile instead of two)
    {
        float2 x[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = threadIdx.x + 256 * u;
            const long t = tile0 - J + i;
            x[u].x = 0.f;
            x[u].y = 0.f;
            if (i < 256 + J) {
                if (t >= 0) {
                    if (t < n_in) x[u] = in[t];
                } else if (t >= -hn) {
                    x[u] = hist[hn + t];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < 256 + J) xs[i] = (f2v){x[u].x, x[u].y};
        }
    }
    __syncthreads();
    f2v acc[U];
#pragma unroll
    for (int p = 0; p < U; ++p) acc[p] = (f2v){0.f, 0.f};
    // The taps are read straight from the table: the index is the same in every lane, so they arrive as scalar
    // loads (as 120 broadcast reads from LDS per lane they were the bottleneck: 0.447 ms).
    // j = J: only the phases with p + U J < L have a tap there
    {
        const f2v x = xs[threadIdx.x];
#pragma unroll
        for (int p = 0; p < U; ++p)
            if (p + U * J < L) acc[p] = acc[p] + x * ir[p + U * J];
    }
#pragma unroll 8
    for (int j = J - 1; j >= 0; --j) {
        const f2v x = xs[threadIdx.x + (J - j)];
#pragma unroll
        for (int p = 0; p < U; ++p) acc[p] = acc[p] + x * ir[p + U * j];
    }
    // the tile's 256 U outputs leave in order: through LDS, so that every store instruction writes 512
    // contiguous bytes (a lane's own U outputs are 8 U bytes apart from its neighbour's)
    __syncthreads();  // the inputs have been read
    f2v *ys = xs;     // (256 U <= 256 + J is not guaranteed: the launcher sizes the buffer for both uses)
#pragma unroll
    for (int p = 0; p < U; ++p) ys[U * threadIdx.x + p] = acc[p];
    __syncthreads();
    const long mt = tile0 * U;
#pragma unroll
    for (int p = 0; p < U; ++p) {
        const long m = mt + 256 * p + threadIdx.x;
        if (m < n_out) __builtin_nontemporal_store(ys[256 * p + threadIdx.x], reinterpret_cast<f2v *>(out) + m);
    }
}

template <int U>
static void launch_upsample_int(hipStream_t s, const void *hist, size_t hn, const void *in, size_t n_in, const void *ir,
                                size_t L, void *out, size_t n_out) {
    const size_t J = (L - 1) / U;
    const size_t lds = ((256 + J) > 256 * (size_t)U ? (256 + J) : 256 * (size_t)U) * 8;
    const unsigned blocks = (unsigned)((n_out + 256 * U - 1) / (256 * U));
    hipLaunchKernelGGL(k_upsample_int<U>, dim3(blocks), dim3(256), lds, s, (const float2 *)hist, (long)hn,
                       (const float2 *)in, (long)n_in, (const float *)ir, (int)L, (float2 *)out, (long)n_out);
}

int launch_upsample(int dtype, hipStream_t s, const void *hist, size_t hn, const void *in, size_t n_in,
                    const void *ir, size_t L, uint64_t U, const int32_t *before, void *out, size_t n_out) {
    if (n_out == 0) return RR_OK;
    if (L > 0x7fffffffull || n_out > 0x7fffffffull * 256) RR_FAIL(RR_ERR_BAD_ARG, "Upsampler: size out of range");
    if (U == 0 && !before) RR_FAIL(RR_ERR_BAD_ARG, "Upsampler: schedule missing");
    if (dtype == RR_F32 && U >= 2 && U <= 16 && (L - 1) / U <= (size_t)kUpJmax && n_out >= 4096 && !std::getenv("RR_UPSAMPLER_GENERIC")) {
        switch (U) {
            case 9: launch_upsample_int<9>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 10: launch_upsample_int<10>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 11: launch_upsample_int<11>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 12: launch_upsample_int<12>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 13: launch_upsample_int<13>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 14: launch_upsample_int<14>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 15: launch_upsample_int<15>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 16: launch_upsample_int<16>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 2: launch_upsample_int<2>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 3: launch_upsample_int<3>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 4: launch_upsample_int<4>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 5: launch_upsample_int<5>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 6: launch_upsample_int<6>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 7: launch_upsample_int<7>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            case 8: launch_upsample_int<8>(s, hist, hn, in, n_in, ir, L, out, n_out); break;
            default: RR_FAIL(RR_ERR_BAD_ARG, "Upsampler: ratio %llu", (unsigned long long)U);
        }
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    const unsigned blocks = (unsigned)((n_out + 255) / 256);
    if (dtype == RR_F32)
        hipLaunchKernelGGL((k_upsample<float, float2>), dim3(blocks), dim3(256), 0, s, (const float2 *)hist, (long)hn,
                           (const float2 *)in, (long)n_in, (const float *)ir, (int)L, (long)U, before, (float2 *)out,
                           (long)n_out);
    else
        hipLaunchKernelGGL((k_upsample<double, double2>), dim3(blocks), dim3(256), 0, s, (const double2 *)hist, (long)hn,
                           (const double2 *)in, (long)n_in, (const double *)ir, (int)L, (long)U, before, (double2 *)out,
                           (long)n_out);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// FmDemod (modulation.rs:121-130)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float atan2_t(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double atan2_t(double y, double x) { return atan2(y, x); }

template <class T, class CT>
__global__ __launch_bounds__(256) void k_fmdemod(const CT *__restrict__ in, long n, CT *__restrict__ out,
                                                 const CT *__restrict__ st_in, CT *__restrict__ st_out,
                                                 int have_prev, T factor, T gain) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const CT cur = in[t];
    CT o;  // (the state keeps the demodulator's own output; `gain` - a GainControl behind it, transform.rs:62-72 - rides on the store)
    if (t == 0 && !have_prev) {
        o = st_in[1];  // output_sample keeps its value (zero before the first pair)
    } else {
        const CT pv = t ? in[t - 1] : st_in[0];
        // sample * previous.conj() as num-complex multiplies: (a.re b.re - a.im b.im, a.re b.im + a.im b.re)
        const T cr = pv.x, ci = -pv.y;
        const T re = sub_rn(mul_rn(cur.x, cr), mul_rn(cur.y, ci));
        const T im = add_rn(mul_rn(cur.x, ci), mul_rn(cur.y, cr));
        o.x = mul_rn(atan2_t(im, re), factor);
        o.y = 0;
    }
    CT og;
    og.x = mul_rn(o.x, gain);
    og.y = mul_rn(o.y, gain);
    out[t] = og;
    if (t == n - 1) {
        st_out[0] = cur;
        st_out[1] = o;
    }
}

// Complex<f32>, long calls: arg() as a polynomial instead of libm's atan2f (which made the kernel VALU-bound:
// 0.221 ms per 2^26 samples).  a = min(|re|, |im|) / max(|re|, |im|) (v_rcp_f32 + one Newton step), atan(a) =
// a P(a^2) with P of degree 8 (Chebyshev interpolation on [0, 1]: 1.0e-7 rad worst case in f32 arithmetic), then the
// octant by subtractions from pi/2 and pi and the sign of im: within 4e-7 rad of the correctly rounded result
// (the reference's Complex::arg is libm's atan2f, modulation.rs:117).  +-0 follow atan2's sign rules; infinities
// do not (NaN).  A lane takes two neighbouring samples per trip (16-byte accesses), four trips in flight, the
// workgroups of an XCD on a contiguous eighth of every grid stride (as k_freqshift).
__device__ __forceinline__ float arg_poly(float im, float re) {
    const float ax = __builtin_fabsf(re), ay = __builtin_fabsf(im);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float rc = __builtin_amdgcn_rcpf(mx);
    float a = mn * rc;
    a = __builtin_fmaf(__builtin_fmaf(-a, mx, mn), rc, a);
    a = mx == 0.f ? 0.f : a;
    const float q = a * a;
    float p = 0x1.6a9512p-9f;
    p = __builtin_fmaf(p, q, -0x1.01bda4p-6f);
    p = __builtin_fmaf(p, q, 0x1.5931p-5f);
    p = __builtin_fmaf(p, q, -0x1.316ecap-4f);
    p = __builtin_fmaf(p, q, 0x1.b2edbp-4f);
    p = __builtin_fmaf(p, q, -0x1.22c55ap-3f);
    p = __builtin_fmaf(p, q, 0x1.996efcp-3f);
    p = __builtin_fmaf(p, q, -0x1.55548ep-2f);
    p = __builtin_fmaf(p, q, 1.0f);
    float r = p * a;
    r = ay > ax ? 0x1.921fb6p+0f - r : r;
    r = __builtin_signbit(re) ? 0x1.921fb6p+1f - r : r;
    return __builtin_copysignf(r, im);
}

__device__ __forceinline__ float fm_one(float2 cur, float2 pv, float factor) {
    // sample * previous.conj() as num-complex multiplies (the products and sums rounded one by one, as above)
    const float cr = pv.x, ci = -pv.y;
    const float re = sub_rn(mul_rn(cur.x, cr), mul_rn(cur.y, ci));
    const float im = add_rn(mul_rn(cur.x, ci), mul_rn(cur.y, cr));
    return mul_rn(arg_poly(im, re), factor);
}

__global__ __launch_bounds__(256) void k_fmdemod_pairs(const float2 *__restrict__ in, long n, float2 *__restrict__ out,
                                                       const float2 *__restrict__ st_in, float2 *__restrict__ st_out,
                                                       int have_prev, float factor, float gain) {
    typedef float f4s __attribute__((ext_vector_type(4)));
    const long npair = n >> 1;  // whole pairs; an odd last sample is the last pair's owner's
    const long stride = (long)gridDim.x * 256;
    const long lb = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // grid: a multiple of 8
    long i = lb * 256 + threadIdx.x;
    auto finish = [&](long ip, f4s v, float2 pv) {
        float2 o0, o1;
        o0.y = 0.f;
        o1.y = 0.f;
        if (ip == 0 && !have_prev)
            o0 = st_in[1];  // output_sample keeps its value (zero before the first pair)
        else
            o0.x = fm_one(float2{v.x, v.y}, pv, factor);
        o1.x = fm_one(float2{v.z, v.w}, float2{v.x, v.y}, factor);
        __builtin_nontemporal_store((f4s){mul_rn(o0.x, gain), mul_rn(o0.y, gain), mul_rn(o1.x, gain), mul_rn(o1.y, gain)},
                                    reinterpret_cast<f4s *>(out + 2 * ip));
        if (2 * ip + 2 == n) {
            st_out[0] = float2{v.z, v.w};
            st_out[1] = o1;
        } else if (2 * ip + 3 == n) {  // the odd last sample
            const float2 cur = in[n - 1];
            float2 o;
            o.x = fm_one(cur, float2{v.z, v.w}, factor);
            o.y = 0.f;
            out[n - 1] = float2{mul_rn(o.x, gain), 0.f};
            st_out[0] = cur;
            st_out[1] = o;
        }
    };
    for (; i + 3 * stride < npair; i += 4 * stride) {
        f4s v[4];
        float2 pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4s *>(in + 2 * (i + u * stride)));
        // the sample in front of a lane's pair is its left neighbour's second one (a wave-wide shift by one lane);
        // lane 0 reads it from memory
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long ip = i + u * stride;
            pv[u].x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[u].z), 0x138, 0xf, 0xf, false));
            pv[u].y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[u].w), 0x138, 0xf, 0xf, false));
            if ((threadIdx.x & 63) == 0) pv[u] = ip ? in[2 * ip - 1] : st_in[0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(i + u * stride, v[u], pv[u]);
    }
    for (; i < npair; i += stride) {
        const f4s v = *reinterpret_cast<const f4s *>(in + 2 * i);
        finish(i, v, i ? in[2 * i - 1] : st_in[0]);
    }
}

int launch_fmdemod(int dtype, hipStream_t s, const void *in, size_t n, void *out, const void *st_in, void *st_out,
                   int have_prev, double factor, double gain) {
    if (n == 0) return RR_OK;
    static const bool generic = [] {
        const char *e = std::getenv("RR_FMDEMOD_GENERIC");
        return e && std::atoi(e) != 0;
    }();
    if (dtype == RR_F32 && n >= 4096 && !generic && reinterpret_cast<uintptr_t>(in) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(out) % 16 == 0) {
        size_t blocks = (n / 2 + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        blocks = (blocks + 7) / 8 * 8;
        hipLaunchKernelGGL(k_fmdemod_pairs, dim3((unsigned)blocks), dim3(256), 0, s, (const float2 *)in, (long)n, (float2 *)out,
                           (const float2 *)st_in, (float2 *)st_out, have_prev, (float)factor, (float)gain);
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (dtype == RR_F32)
        hipLaunchKernelGGL((k_fmdemod<float, float2>), dim3(blocks), dim3(256), 0, s, (const float2 *)in, (long)n,
                           (float2 *)out, (const float2 *)st_in, (float2 *)st_out, have_prev, (float)factor, (float)gain);
    else
        hipLaunchKernelGGL((k_fmdemod<double, double2>), dim3(blocks), dim3(256), 0, s, (const double2 *)in, (long)n,
                           (double2 *)out, (const double2 *)st_in, (double2 *)st_out, have_prev, factor, gain);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_gain(int dtype, hipStream_t s, double gain, const void *in, size_t n, void *out) {
    if (n == 0) return RR_OK;
    size_t blocks = (2 * n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == RR_F32)
        hipLaunchKernelGGL(k_gain<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float *)in, (float *)out, 2 * n, (float)gain);
    else
        hipLaunchKernelGGL(k_gain<double>, dim3((unsigned)blocks), dim3(256), 0, s, (const double *)in, (double *)out, 2 * n, gain);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using namespace rr;

static int dev_check(int dtype, int device) {
    if (dtype != RR_F32 && dtype != RR_F64) RR_FAIL(RR_ERR_BAD_ARG, "unknown dtype %d", dtype);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) RR_FAIL(RR_ERR_HIP, "no HIP device available; this backend has no CPU fallback");
    if (device < 0 || device >= count) RR_FAIL(RR_ERR_BAD_ARG, "device %d out of range", device);
    RR_HIP(hipSetDevice(device));
    return RR_OK;
}

// host-pointer convenience: copy in, run, copy out, blocking
template <class F>
static int host_roundtrip(int dtype, int device, const void *in, size_t in_bytes, void *out, size_t out_bytes, F &&run) {
    RR_TRY(dev_check(dtype, device));
    DevBuf din, dout;
    RR_TRY(din.reserve(in_bytes ? in_bytes : 16));
    RR_TRY(dout.reserve(out_bytes ? out_bytes : 16));
    if (in_bytes) RR_HIP(hipMemcpy(din.p, in, in_bytes, hipMemcpyHostToDevice));
    RR_TRY(run(din.p, dout.p));
    RR_HIP(hipStreamSynchronize(nullptr));
    if (out_bytes) RR_HIP(hipMemcpy(out, dout.p, out_bytes, hipMemcpyDeviceToHost));
    return RR_OK;
}

extern "C" {

int rr_level_dev(int dtype, int device, void *stream, const void *d_frames, size_t n, size_t count, double *d_out) {
    RR_TRY(dev_check(dtype, device));
    return launch_meter(dtype, (hipStream_t)stream, 0, 0.0, 0.0, d_frames, n, count, d_out);
}
int rr_bandwidth_dev(int dtype, int device, void *stream, double double_percentile, double sample_rate,
                     const void *d_frames, size_t n, size_t count, double *d_out) {
    RR_TRY(dev_check(dtype, device));
    return launch_meter(dtype, (hipStream_t)stream, 1, double_percentile, sample_rate, d_frames, n, count, d_out);
}
int rr_bandwidth_fast_dev(int dtype, int device, void *stream, double double_percentile, double sample_rate,
                          const void *d_frames, size_t n, size_t count, double *d_bandwidth, double *d_energy) {
    RR_TRY(dev_check(dtype, device));
    if (!d_bandwidth) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return launch_bandwidth_par(dtype, (hipStream_t)stream, double_percentile, sample_rate, d_frames, n, count, d_bandwidth, d_energy);
}
int rr_rescale_energy_dev(int dtype, int device, void *stream, const void *d_frames, size_t n, size_t count,
                          size_t resolution, void *d_out) {
    RR_TRY(dev_check(dtype, device));
    return launch_rescale_energy(dtype, (hipStream_t)stream, d_frames, n, count, resolution, d_out);
}
int rr_gain_dev(int dtype, int device, void *stream, double gain, const void *d_in, size_t n, void *d_out) {
    RR_TRY(dev_check(dtype, device));
    return launch_gain(dtype, (hipStream_t)stream, gain, d_in, n, d_out);
}

int rr_level(int dtype, int device, const void *chunk, size_t n, double *out) {
    if (!out || (n && !chunk)) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return host_roundtrip(dtype, device, chunk, n * elem_size(dtype), out, sizeof(double), [&](void *di, void *dout) {
        return launch_meter(dtype, nullptr, 0, 0.0, 0.0, di, n, 1, (double *)dout);
    });
}
int rr_bandwidth(int dtype, int device, double double_percentile, double sample_rate, const void *bins, size_t n,
                 double *out) {
    if (!out || (n && !bins)) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return host_roundtrip(dtype, device, bins, n * elem_size(dtype), out, sizeof(double), [&](void *di, void *dout) {
        return launch_meter(dtype, nullptr, 1, double_percentile, sample_rate, di, n, 1, (double *)dout);
    });
}
int rr_rescale_energy(int dtype, int device, const void *input, size_t n, size_t resolution, void *output) {
    if ((resolution && !output) || (n && !input)) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return host_roundtrip(dtype, device, input, n * elem_size(dtype), output, resolution * (elem_size(dtype) / 2),
                          [&](void *di, void *dout) { return launch_rescale_energy(dtype, nullptr, di, n, 1, resolution, dout); });
}
int rr_gain(int dtype, int device, double gain, const void *in, size_t n, void *out) {
    if (n && (!in || !out)) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return host_roundtrip(dtype, device, in, n * elem_size(dtype), out, n * elem_size(dtype),
                          [&](void *di, void *dout) { return launch_gain(dtype, nullptr, gain, di, n, dout); });
}

}  // extern "C"
