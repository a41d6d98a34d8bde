// rr_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the IQ hot path.
//
// This file holds the GENERIC kernels: they accept every parameter set the
// reference blocks accept (any tap count, any resampling schedule, any chunk
// length) and are used block-by-block.  The fused fast path for the
// FreqShifter->Filter->Downsampler->Fourier chain lives in rr_ols.hip and rr_fft_regs.hip.
//
// These are vector (VALU + LDS) kernels: FIR dot products and FFT butterflies
// over Complex<f32>/<f64>; no MFMA (not a dense contraction).
#include "rr_kernels.hpp"

#include <cstdlib>
#include <mutex>
#include <tuple>

namespace rr {

template <class T> struct V2;
template <> struct V2<float> { using type = float2; };
template <> struct V2<double> { using type = double2; };
template <class T> using v2 = typename V2<T>::type;

#ifndef RR_V_FFT_NT
#define RR_V_FFT_NT 1  // the two-pass transforms' results (runs of whole lines, written once) by non-temporal stores: 65536 points 0.165 -> 0.149 ms
#endif
template <class T> __device__ __forceinline__ void st_result(v2<T> *p, v2<T> v) {
#if RR_V_FFT_NT
    typedef T vt __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store((vt){v.x, v.y}, reinterpret_cast<vt *>(p));
#else
    *p = v;
#endif
}

template <class T> __device__ __forceinline__ v2<T> cmul(v2<T> a, v2<T> b) {
    v2<T> r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}

static bool is_pow2_n(size_t n) { return n && (n & (n - 1)) == 0; }

// More than 64 KiB of dynamic LDS needs an opt-in per kernel AND per device (a process may hold handles on several devices, every
// entry point selects its own): done once per (kernel, device, size), remembered under a lock that a launch holds for nanoseconds.
int dyn_lds_optin(const void *fn, size_t bytes) {
    if (bytes <= 64 * 1024) return RR_OK;
    int dev = 0;
    RR_HIP(hipGetDevice(&dev));
    static std::mutex mu;
    static std::vector<std::tuple<const void *, int, size_t>> done;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const auto &d : done)
            if (std::get<0>(d) == fn && std::get<1>(d) == dev && std::get<2>(d) >= bytes) return RR_OK;
    }
    RR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    std::lock_guard<std::mutex> lk(mu);
    done.emplace_back(fn, dev, bytes);
    return RR_OK;
}
static int set_dyn_lds(const void *fn, size_t bytes) { return dyn_lds_optin(fn, bytes); }

// ---------------------------------------------------------------------------
// FreqShifter: y[t] = x[t] * p[(idx0 + t) mod denom]          transform.rs:341-348
// Pure streaming: 8 B in + 8 B out per sample (16 B/lane vector accesses for f32),
// the phase table stays in L2/MALL (it is at most a few MB for sane precisions).
// ---------------------------------------------------------------------------
#ifndef RR_V_FSXCD
#define RR_V_FSXCD 1  // 0.232 -> 0.199 ms per 2^26 samples
#endif
#ifndef RR_V_FSUNROLL
#define RR_V_FSUNROLL 1  // with the XCD mapping: 0.199 -> 0.177 ms
#endif
template <class T, int VEC>
__global__ __launch_bounds__(256) void k_freqshift(const v2<T> *__restrict__ in, v2<T> *__restrict__ out, size_t n,
                                                   const v2<T> *__restrict__ table, uint32_t denom, uint32_t idx0) {
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
#if RR_V_FSXCD
    // workgroups b, b + 8, .. share an XCD: give each XCD a contiguous eighth of every grid stride (grid: multiple of 8)
    const size_t lb = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#else
    const size_t lb = blockIdx.x;
#endif
    size_t i = (lb * blockDim.x + threadIdx.x) * VEC;
    // phase index of this thread's first sample and its advance per grid stride
    uint32_t r = (uint32_t)(((uint64_t)idx0 + i % denom) % denom);
    const uint32_t step = (uint32_t)((nthreads * VEC) % denom);
#if RR_V_FSUNROLL
    if constexpr (VEC == 2 && sizeof(T) == 4) {
        typedef float f4s __attribute__((ext_vector_type(4)));
        if (step == 0) {
            const v2<T> p0 = table[r], p1 = table[r + 1 == denom ? 0 : r + 1];
            const size_t stride = nthreads * VEC;
            for (; i + 3 * stride + VEC <= n; i += 4 * stride) {
                f4s v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4s *>(in + i + u * stride));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const v2<T> y0 = cmul<T>(make_float2(v[u].x, v[u].y), p0);
                    const v2<T> y1 = cmul<T>(make_float2(v[u].z, v[u].w), p1);
                    __builtin_nontemporal_store((f4s){y0.x, y0.y, y1.x, y1.y}, reinterpret_cast<f4s *>(out + i + u * stride));
                }
            }
        }
    }
#endif
    for (; i + VEC <= n; i += nthreads * VEC) {
        v2<T> x[VEC], y[VEC];
        if constexpr (VEC == 2 && sizeof(T) == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(in + i);
            x[0] = make_float2(v.x, v.y);
            x[1] = make_float2(v.z, v.w);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) x[k] = in[i + k];
        }
        uint32_t rr_ = r;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            y[k] = cmul<T>(x[k], table[rr_]);
            rr_ = rr_ + 1 == denom ? 0 : rr_ + 1;
        }
        if constexpr (VEC == 2 && sizeof(T) == 4) {
            *reinterpret_cast<float4 *>(out + i) = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) out[i + k] = y[k];
        }
        r += step;
        if (r >= denom) r -= denom;
    }
    // ragged tail (< VEC samples): the thread whose slot covers it
    if (VEC > 1 && i < n) {
        uint32_t rr_ = r;
        for (; i < n; ++i) {
            out[i] = cmul<T>(in[i], table[rr_]);
            rr_ = rr_ + 1 == denom ? 0 : rr_ + 1;
        }
    }
}

int launch_freqshift(int dtype, hipStream_t s, const void *in, void *out, size_t n, const void *table,
                     uint32_t denom, uint32_t idx0) {
    if (n == 0) return RR_OK;
    const int block = 256;
    const bool f32 = dtype == RR_F32;
    const bool vec = f32 && (reinterpret_cast<uintptr_t>(in) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    const size_t per = vec ? 2 : 1;
    size_t blocks = (n + block * per - 1) / (block * per);
    if (blocks > 256 * 16) {
        blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
        // ... and a whole number of table periods per grid stride where some grid of 8 .. 16 blocks per CU
        // gives that: the kernel then keeps the lane's two phasors in registers (blocks must be a multiple of
        // denom / gcd(denom, samples per block), and of 8 for the XCD mapping)
        uint64_t g = denom, t = (uint64_t)block * per;
        while (t) {
            const uint64_t q = g % t;
            g = t;
            t = q;
        }
        uint64_t unit = denom / g;
        while (unit % 8) unit *= 2;
        if (unit <= blocks && blocks / unit * unit >= 256 * 8) blocks = blocks / unit * unit;
    }
#if RR_V_FSXCD
    blocks = (blocks + 7) / 8 * 8;  // (the kernel's block -> XCD mapping)
#endif
    if (f32) {
        if (vec)
            hipLaunchKernelGGL((k_freqshift<float, 2>), dim3(blocks), dim3(block), 0, s, (const float2 *)in,
                               (float2 *)out, n, (const float2 *)table, denom, idx0);
        else
            hipLaunchKernelGGL((k_freqshift<float, 1>), dim3(blocks), dim3(block), 0, s, (const float2 *)in,
                               (float2 *)out, n, (const float2 *)table, denom, idx0);
    } else {
        hipLaunchKernelGGL((k_freqshift<double, 1>), dim3(blocks), dim3(block), 0, s, (const double2 *)in,
                           (double2 *)out, n, (const double2 *)table, denom, idx0);
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Generic gather-FIR (Filter in direct form, Downsampler):
//   out[m] = sum_{j<K} w[j] * x[e_m - (K-1) + j]
// One workgroup stages the input span of its outputs (plus the K-1 halo) in LDS
// with coalesced loads; each lane then owns R outputs that share every tap load.
// Taps are wave-uniform, read through the scalar cache.
// ---------------------------------------------------------------------------
template <class T, bool CT> struct TapType { using type = T; };
template <class T> struct TapType<T, true> { using type = v2<T>; };

template <class T, bool CT, bool LIST, int R>
__global__ __launch_bounds__(256) void k_fir(const v2<T> *__restrict__ hist, long hist_len,
                                             const v2<T> *__restrict__ in, long n_in,
                                             const typename TapType<T, CT>::type *__restrict__ taps, int K,
                                             v2<T> *__restrict__ out, size_t n_out, unsigned long long e0, uint32_t D,
                                             const uint32_t *__restrict__ emit, uint32_t outs_per_block, int Kc,
                                             uint32_t period_p, uint32_t period_q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2<T> *xs = reinterpret_cast<v2<T> *>(smem);
    const size_t m0 = (size_t)blockIdx.x * outs_per_block;
    if (m0 >= n_out) return;
    const size_t m1 = (m0 + outs_per_block < n_out) ? m0 + outs_per_block : n_out;
    auto e_of = [&](size_t m) -> long {
        if (LIST && period_q) {  // a periodic schedule: its first period_q positions, repeated every period_p inputs
            const size_t q = m / period_q;
            return (long)(q * period_p + emit[m - q * period_q]);
        }
        return LIST ? (long)emit[m] : (long)(e0 + m * (unsigned long long)D);
    };
    const long hi = e_of(m1 - 1);
    // Long responses are taken Kc taps at a time (Kc = K when the whole span fits the LDS tile): per pass the
    // tile holds the samples those taps touch, the partial sums stay in registers.  With more than one pass a
    // workgroup has at most 256 R outputs (the launcher's duty), i.e. one round of the output loop.
    const bool one_pass = Kc >= K;
    v2<T> acc_keep[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc_keep[r].x = 0;
        acc_keep[r].y = 0;
    }
    for (int jc = 0; jc < K; jc += Kc) {
        const int kc = (jc + Kc < K) ? Kc : K - jc;  // taps of this pass: jc .. jc + kc - 1
        const long lo = e_of(m0) - (K - 1) + jc;      // position of tap jc of output m0
        if (jc) __syncthreads();                      // the previous pass has been read
        for (long i = lo + threadIdx.x; i <= hi - (K - 1) + jc + kc - 1; i += blockDim.x) {
            v2<T> v;
            v.x = 0;
            v.y = 0;
            if (i >= 0) {
                if (i < n_in) v = in[i];
            } else if (i >= -hist_len) {
                v = hist[hist_len + i];
            }
            xs[i - lo] = v;
        }
        __syncthreads();
        for (size_t mb = m0 + threadIdx.x; mb < m1; mb += (size_t)blockDim.x * R) {
            int base[R];
            v2<T> acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const size_t m = mb + (size_t)r * blockDim.x;
                base[r] = (m < m1) ? (int)(e_of(m) - (K - 1) + jc - lo) : 0;
                acc[r] = acc_keep[r];
            }
            // two-level summation: partial sums over 64 taps keep the rounding error
            // of long filters (n = 4096) at the level of the FFT-based reference
            for (int j0 = 0; j0 < kc; j0 += 64) {
                v2<T> part[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    part[r].x = 0;
                    part[r].y = 0;
                }
                const int j1 = (j0 + 64 < kc) ? j0 + 64 : kc;
                for (int j = j0; j < j1; ++j) {
                    const auto w = taps[jc + j];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const v2<T> x = xs[base[r] + j];
                        if constexpr (CT) {
                            part[r].x += w.x * x.x - w.y * x.y;
                            part[r].y += w.x * x.y + w.y * x.x;
                        } else {
                            part[r].x += x.x * w;
                            part[r].y += x.y * w;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r].x += part[r].x;
                    acc[r].y += part[r].y;
                }
            }
            if (one_pass || jc + kc >= K) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const size_t m = mb + (size_t)r * blockDim.x;
                    if (m < m1) out[m] = acc[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) acc_keep[r] = acc[r];
            }
        }
    }
}

template <class T, bool CT, bool LIST>
static int launch_fir_t(hipStream_t s, const FirArgs &a) {
    constexpr int R = 4;
    const size_t esz = sizeof(v2<T>);
    const size_t budget = kFirLdsBytes / esz;  // samples
    if (a.K == 0) RR_FAIL(RR_ERR_BAD_ARG, "fir: no taps");
    const uint64_t step = LIST ? a.max_step : a.D;
    uint64_t opb, Kc = a.K;
    if ((uint64_t)a.K + 63 * step + 1 > budget) {
        // the response alone does not fit the tile: up to 256 outputs per workgroup (fewer when the outputs lie
        // far apart: 512 : 1 leaves 8), the taps in passes
        opb = 256;
        while (opb > 1 && (opb - 1) * step + 64 > budget / 2) opb /= 2;
        if ((opb - 1) * step + 64 > budget)
            RR_FAIL(RR_ERR_BAD_ARG, "fir: an output step of %llu samples exceeds the LDS tile (%zu samples)",
                    (unsigned long long)step, budget);
        Kc = (budget - (opb - 1) * step) & ~uint64_t(63);
    } else {
        opb = (budget - a.K) / (step ? step : 1) + 1;
        if (opb > 2048) opb = 2048;
        // keep >= ~4 workgroups per CU when the problem is large enough
        while (opb > 256 && (a.n_out + opb - 1) / opb < 1024) opb /= 2;
        if (opb > 64) opb &= ~uint64_t(63);
    }
    const size_t span = (opb - 1) * step + (Kc < a.K ? Kc : a.K);
    const size_t lds = span * esz;
    auto fn = k_fir<T, CT, LIST, R>;
    RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
    const size_t blocks = (a.n_out + opb - 1) / opb;
    if (blocks > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fir: too many workgroups");
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), lds, s, (const v2<T> *)a.hist, (long)a.hist_len,
                       (const v2<T> *)a.in, (long)a.n_in, (const typename TapType<T, CT>::type *)a.taps, (int)a.K,
                       (v2<T> *)a.out, a.n_out, (unsigned long long)a.e0, a.D, a.emit, (uint32_t)opb, (int)Kc, a.period_p,
                       a.period_q);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_fir(int dtype, hipStream_t s, const FirArgs &a) {
    if (a.n_out == 0) return RR_OK;
    const bool list = a.emit != nullptr;
    if (dtype == RR_F32) {
        if (a.complex_taps) return list ? launch_fir_t<float, true, true>(s, a) : launch_fir_t<float, true, false>(s, a);
        return list ? launch_fir_t<float, false, true>(s, a) : launch_fir_t<float, false, false>(s, a);
    }
    if (a.complex_taps) return list ? launch_fir_t<double, true, true>(s, a) : launch_fir_t<double, true, false>(s, a);
    return list ? launch_fir_t<double, false, true>(s, a) : launch_fir_t<double, false, false>(s, a);
}

// ---------------------------------------------------------------------------
// Filter, overlap-save fast convolution (filters.rs:240-259) for long filters.
// One workgroup per output chunk: [prev | cur] (2n samples) -> Stockham radix-2 FFT in
// LDS -> * H -> inverse (conjugate twiddles) -> the first n results.  ~20 log2(2n)
// flop per sample instead of 8n for the direct form.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_filter_ols(const v2<T> *__restrict__ hist, const v2<T> *__restrict__ in, int n,
                                                    int first_chunk, const v2<T> *__restrict__ H,
                                                    const v2<T> *__restrict__ tw, v2<T> *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = 2 * n;
    v2<T> *a = reinterpret_cast<v2<T> *>(smem);
    v2<T> *b = a + N;
    const long c = (long)blockIdx.x + first_chunk;  // chunk whose filtered version this workgroup emits
    const v2<T> *prev = (c == 0) ? hist : in + (c - 1) * (long)n;
    const v2<T> *cur = in + c * (long)n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        a[i] = prev[i];
        a[n + i] = cur[i];
    }
    __syncthreads();
    const int half = n;  // N / 2
    for (int dir = 0; dir < 2; ++dir) {
        for (int ns = 1; ns < N; ns <<= 1) {
            const int tstride = half / ns;
            for (int j = threadIdx.x; j < half; j += blockDim.x) {
                const int k = j & (ns - 1);
                v2<T> w = tw[k * tstride];
                if (dir) w.y = -w.y;
                const v2<T> u = a[j];
                const v2<T> v = cmul<T>(a[j + half], w);
                const int j0 = ((j - k) << 1) + k;
                v2<T> s, d;
                s.x = u.x + v.x;
                s.y = u.y + v.y;
                d.x = u.x - v.x;
                d.y = u.y - v.y;
                b[j0] = s;
                b[j0 + ns] = d;
            }
            __syncthreads();
            v2<T> *t = a;
            a = b;
            b = t;
        }
        if (dir == 0) {
            for (int i = threadIdx.x; i < N; i += blockDim.x) a[i] = cmul<T>(a[i], H[i]);
            __syncthreads();
        }
    }
    v2<T> *dst = out + (long)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = a[i];
}

static size_t ols_max_n(int dtype) { return dtype == RR_F32 ? 4096 : 2048; }  // 2 LDS buffers of 2n <= 128 KiB

bool ols_supported(int dtype, size_t n) { return is_pow2_n(n) && n >= 128 && n <= ols_max_n(dtype); }

int launch_filter_ols(int dtype, hipStream_t s, const void *hist, const void *in, size_t n, size_t nchunks,
                      int first_chunk, const void *H, const void *tw, void *out) {
    if (nchunks == 0) return RR_OK;
    if (nchunks > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Filter: too many chunks in one call");
    const size_t lds = 4 * n * elem_size(dtype);
    if (dtype == RR_F32) {
        auto fn = k_filter_ols<float>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)nchunks), dim3(256), lds, s, (const float2 *)hist, (const float2 *)in,
                           (int)n, first_chunk, (const float2 *)H, (const float2 *)tw, (float2 *)out);
    } else {
        auto fn = k_filter_ols<double>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)nchunks), dim3(256), lds, s, (const double2 *)hist, (const double2 *)in,
                           (int)n, first_chunk, (const double2 *)H, (const double2 *)tw, (double2 *)out);
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// history carry: new_hist = last H samples of [ zeros | old_hist | in ]
// ---------------------------------------------------------------------------
template <class T>
__global__ void k_update_hist(const v2<T> *__restrict__ old_hist, v2<T> *__restrict__ new_hist, long H,
                              const v2<T> *__restrict__ in, long n_in) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    const long src = n_in - H + i;  // index into `in`; negative -> old history
    v2<T> v;
    if (src >= 0)
        v = in[src];
    else
        v = old_hist[H + src];  // src >= -H always
    new_hist[i] = v;
}

int launch_update_hist(int dtype, hipStream_t s, const void *old_hist, void *new_hist, size_t H, const void *in,
                       size_t n_in) {
    if (H == 0) return RR_OK;
    const unsigned blocks = (unsigned)((H + 255) / 256);
    if (dtype == RR_F32)
        hipLaunchKernelGGL(k_update_hist<float>, dim3(blocks), dim3(256), 0, s, (const float2 *)old_hist,
                           (float2 *)new_hist, (long)H, (const float2 *)in, (long)n_in);
    else
        hipLaunchKernelGGL(k_update_hist<double>, dim3(blocks), dim3(256), 0, s, (const double2 *)old_hist,
                           (double2 *)new_hist, (long)H, (const double2 *)in, (long)n_in);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Fourier, power-of-two length: window * x -> Stockham autosort radix-2 in LDS,
// one workgroup per chunk.  center_dc is an index rotation on the store.
// (Generic version; the radix-16 register kernel for n = 4096 is in rr_fft_regs.hip.)
// ---------------------------------------------------------------------------
// Stockham autosort passes of a power-of-two transform between two LDS images (radix 4 while it fits, then radix 2); the
// caller has filled `a` and synchronised; returns the image that holds the result (synchronised).
template <class T>
__device__ __forceinline__ v2<T> *fft_pow2_passes(v2<T> *a, v2<T> *b, int n, const v2<T> *__restrict__ tw) {
    const int half = n >> 1;
    int ns = 1;
    // radix-4 passes while they fit (half the exchanges and barriers of radix 2; the table holds all n twiddles)
    for (; ns * 4 <= n; ns <<= 2) {
        const int quarter = n >> 2, tstride4 = quarter / ns;  // e^{-j 2 pi k c / (4 ns)} = tw[k c n / (4 ns)]
        for (int j = threadIdx.x; j < quarter; j += blockDim.x) {
            const int k = j & (ns - 1);
            const v2<T> x0 = a[j];
            const v2<T> x1 = cmul<T>(a[j + quarter], tw[k * tstride4]);
            const v2<T> x2 = cmul<T>(a[j + 2 * quarter], tw[2 * k * tstride4]);
            const v2<T> x3 = cmul<T>(a[j + 3 * quarter], tw[3 * k * tstride4]);
            v2<T> t0, t1, t2, t3, y;
            t0.x = x0.x + x2.x; t0.y = x0.y + x2.y;
            t1.x = x0.x - x2.x; t1.y = x0.y - x2.y;
            t2.x = x1.x + x3.x; t2.y = x1.y + x3.y;
            t3.x = x1.y - x3.y; t3.y = x3.x - x1.x;  // -j (x1 - x3)
            const int j0 = ((j - k) << 2) + k;
            y.x = t0.x + t2.x; y.y = t0.y + t2.y; b[j0] = y;
            y.x = t1.x + t3.x; y.y = t1.y + t3.y; b[j0 + ns] = y;
            y.x = t0.x - t2.x; y.y = t0.y - t2.y; b[j0 + 2 * ns] = y;
            y.x = t1.x - t3.x; y.y = t1.y - t3.y; b[j0 + 3 * ns] = y;
        }
        __syncthreads();
        v2<T> *t = a;
        a = b;
        b = t;
    }
    for (; ns < n; ns <<= 1) {
        const int tstride = half / ns;  // tw index step: e^{-j 2 pi k / (2 ns)} = tw[k * n / (2 ns)]
        for (int j = threadIdx.x; j < half; j += blockDim.x) {
            const int k = j & (ns - 1);
            const v2<T> u = a[j];
            const v2<T> v = cmul<T>(a[j + half], tw[k * tstride]);
            const int j0 = ((j - k) << 1) + k;
            v2<T> s, d;
            s.x = u.x + v.x;
            s.y = u.y + v.y;
            d.x = u.x - v.x;
            d.y = u.y - v.y;
            b[j0] = s;
            b[j0 + ns] = d;
        }
        __syncthreads();
        v2<T> *t = a;
        a = b;
        b = t;
    }
    return a;
}

template <class T>
// Frames are cut from the stream [ head (n_head samples) | in ] every `hop` samples (hop = n: the
// plain chunk-by-chunk Fourier; hop < n: the Overlapper's overlapping chunks, chunks.rs:194-242).
__global__ __launch_bounds__(1024) void k_fft_pow2(const v2<T> *__restrict__ head, long n_head,
                                                  const v2<T> *__restrict__ in, v2<T> *__restrict__ out, int n,
                                                  long hop, const T *__restrict__ window,
                                                  const v2<T> *__restrict__ tw, int center_dc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2<T> *a = reinterpret_cast<v2<T> *>(smem);
    v2<T> *b = a + n;
    const size_t chunk = blockIdx.x;
    const long base = (long)chunk * hop - n_head;
    v2<T> *dst = out + chunk * (size_t)n;
    // (four loads in flight per lane: as one loop there is a single load between a wait and the LDS store)
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * blockDim.x) {
        v2<T> v[4];
        T w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < n) {
                const long pos = base + i;
                v[u] = pos >= 0 ? in[pos] : head[n_head + pos];
                w[u] = window[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < n) a[i] = v2<T>{v[u].x * w[u], v[u].y * w[u]};
        }
    }
    __syncthreads();
    a = fft_pow2_passes<T>(a, b, n, tw);
    const int rot = center_dc ? (n >> 1) : 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int o = i + rot;
        if (o >= n) o -= n;
        dst[o] = a[i];
    }
}

// Bluestein's algorithm in ONE kernel for the lengths whose padded power-of-two length M fits two LDS images (M <= 8192 in f32:
// 2049 .. 4096 points; M <= 4096 in f64: up to 2048 points) and that have no kernel of their own (k_bluestein1024 / 4096 serve
// f32 up to 2048 points, the mixed-radix kernels the lengths 2^a 3^b 5^c 7^d 11^e 13^f): a workgroup per chunk,
//   v = x c (c = window conj(chirp), zero beyond n) -> DFT_M -> conj(. B) (B = DFT_M(chirp, wrapped) / M) -> DFT_M ->
//   conj(. chirp) -> n bins,
// the two transforms by the Stockham passes of k_fft_pow2.  The five launches it replaces (k_bs_pre, the transform, k_bs_mul,
// the transform, k_bs_post) move ~10 M elements through HBM per chunk; here a chunk is read once and written once.
template <class T>
__global__ __launch_bounds__(1024) void k_bluestein_lds(const v2<T> *__restrict__ head, long n_head, const v2<T> *__restrict__ in,
                                                       long hop, int n, int M, const v2<T> *__restrict__ c,
                                                       const v2<T> *__restrict__ B, const v2<T> *__restrict__ w,
                                                       const v2<T> *__restrict__ tw, v2<T> *__restrict__ out, int center_dc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2<T> *a = reinterpret_cast<v2<T> *>(smem);
    v2<T> *b = a + M;
    const size_t chunk = blockIdx.x;
    const long base = (long)chunk * hop - n_head;
    for (int i0 = threadIdx.x; i0 < M; i0 += 4 * blockDim.x) {
        v2<T> x[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < n) {
                const long pos = base + i;
                x[u] = pos >= 0 ? in[pos] : head[n_head + pos];
                cc[u] = c[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < M) a[i] = i < n ? cmul<T>(x[u], cc[u]) : v2<T>{(T)0, (T)0};
        }
    }
    __syncthreads();
    v2<T> *r = fft_pow2_passes<T>(a, b, M, tw);
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const v2<T> v = cmul<T>(r[i], B[i]);
        r[i] = v2<T>{v.x, -v.y};
    }
    __syncthreads();
    r = fft_pow2_passes<T>(r, r == a ? b : a, M, tw);
    v2<T> *dst = out + chunk * (size_t)n;
    const int rot = center_dc ? n / 2 : 0;  // rotate_right(n / 2)
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const v2<T> v = cmul<T>(r[k], w[k]);
        int o = k + rot;
        if (o >= n) o -= n;
        dst[o] = v2<T>{v.x, -v.y};
    }
}

bool bluestein_lds_supported(int dtype, size_t n, size_t M) {
    const size_t lim = dtype == RR_F32 ? 8192 : 4096;  // two images within 128 KiB (as k_fft_pow2)
    return n >= 32 && M >= 64 && M <= lim && M >= 2 * n - 1;
}

// c, B, w: the tables of the five-launch form (B unpermuted, already divided by M); tw = e^{-j 2 pi k / M} (M entries)
int launch_bluestein_lds(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, size_t M,
                         const void *c, const void *B, const void *w, const void *tw, void *out, bool center_dc, size_t count) {
    if (count == 0) return RR_OK;
    if (!bluestein_lds_supported(dtype, n, M)) RR_FAIL(RR_ERR_BAD_ARG, "Bluestein in LDS: %zu points, M = %zu", n, M);
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many chunks in one call");
    unsigned nt = (unsigned)(M / 4);
    if (nt > 1024) nt = 1024;
    if (nt < 64) nt = 64;
    if (dtype == RR_F32) {
        auto fn = k_bluestein_lds<float>;
        const size_t lds = 2 * M * sizeof(float2);
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)count), dim3(nt), lds, s, (const float2 *)head, (long)n_head, (const float2 *)in,
                           (long)hop, (int)n, (int)M, (const float2 *)c, (const float2 *)B, (const float2 *)w, (const float2 *)tw,
                           (float2 *)out, (int)center_dc);
    } else {
        auto fn = k_bluestein_lds<double>;
        const size_t lds = 2 * M * sizeof(double2);
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)count), dim3(nt), lds, s, (const double2 *)head, (long)n_head, (const double2 *)in,
                           (long)hop, (int)n, (int)M, (const double2 *)c, (const double2 *)B, (const double2 *)w, (const double2 *)tw,
                           (double2 *)out, (int)center_dc);
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// Fourier, any other length: direct DFT  X[k] = sum_j (w[j] x[j]) tw[(j k) mod n].
// grid = (k tiles, chunks).  O(n^2) — small/odd chunk lengths only.
template <class T>
__global__ __launch_bounds__(256) void k_dft_direct(const v2<T> *__restrict__ in, v2<T> *__restrict__ out, int n,
                                                    const T *__restrict__ window, const v2<T> *__restrict__ tw,
                                                    int center_dc) {
    __shared__ v2<T> xs[1024];
    const size_t chunk = blockIdx.y;
    const v2<T> *src = in + chunk * (size_t)n;
    v2<T> *dst = out + chunk * (size_t)n;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    v2<T> acc;
    acc.x = 0;
    acc.y = 0;
    for (int j0 = 0; j0 < n; j0 += 1024) {
        const int cnt = (n - j0 < 1024) ? n - j0 : 1024;
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            v2<T> v = src[j0 + i];
            const T w = window[j0 + i];
            v.x *= w;
            v.y *= w;
            xs[i] = v;
        }
        __syncthreads();
        if (k < n) {
            int idx = (int)(((long long)j0 * k) % n);
            for (int i = 0; i < cnt; ++i) {
                const v2<T> p = cmul<T>(xs[i], tw[idx]);
                acc.x += p.x;
                acc.y += p.y;
                idx += k;
                if (idx >= n) idx -= n;
            }
        }
    }
    if (k < n) {
        int o = k + (center_dc ? n / 2 : 0);  // rotate_right(n / 2)
        if (o >= n) o -= n;
        dst[o] = acc;
    }
}

static bool is_pow2(size_t n) { return n && (n & (n - 1)) == 0; }
// one LDS tile: 2 buffers <= 128 KiB for k_fft_pow2; f32 16 384 points through k_fft16384's single image (RR_FOURIER_16K=0: the two
// passes of k_fft_tile as before - A/B runs, tests)
static size_t pow2_limit(int dtype) {
    const char *e16 = std::getenv("RR_FOURIER_16K");  // (read per design: tests switch it within one process)
    const bool no16k = e16 && std::atoi(e16) == 0;
    static const bool generic = [] { const char *e = std::getenv("RR_FOURIER_GENERIC"); return e && std::atoi(e) != 0; }();
    return dtype == RR_F32 ? ((no16k || generic) ? 8192 : 16384) : 4096;
}

bool fourier_pow2_path(int dtype, size_t n) { return is_pow2(n) && n >= 2 && n <= pow2_limit(dtype); }

// any length up to 2^23 (Bluestein over power-of-two transforms of up to 2^24 points), powers of two up to 2^24
int fourier_supported(int dtype, size_t n) {
    (void)dtype;
    if (n == 0) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: empty chunk");
    if (is_pow2(n) ? n <= ((size_t)1 << 24) : n <= ((size_t)1 << 23)) return RR_OK;
    RR_FAIL(RR_ERR_BAD_ARG, "Fourier: chunk length %zu is not supported (powers of two up to 2^24, any length up to 2^23)", n);
}

template <class T>
static int launch_fourier_t(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t n,
                            size_t hop, size_t count, const void *window, const void *twiddle, bool center_dc,
                            int dtype) {
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many chunks in one call");
    if (fourier_pow2_path(dtype, n)) {
        const size_t lds = 2 * n * sizeof(v2<T>);
        auto fn = k_fft_pow2<T>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        // a workgroup per chunk, a lane per radix-4 butterfly: large chunks fill the CU with ONE workgroup (the two
        // Stockham images of 4096 f64 points are 128 KiB), so it gets up to 16 waves
        int threads = (int)(n / 4);
        if (threads > 1024) threads = 1024;
        if (threads < 64) threads = 64;
        hipLaunchKernelGGL(fn, dim3((unsigned)count), dim3(threads), lds, s, (const v2<T> *)head, (long)n_head,
                           (const v2<T> *)in, (v2<T> *)out, (int)n, (long)hop, (const T *)window, (const v2<T> *)twiddle,
                           (int)center_dc);
    } else {
        if (hop != n || n_head) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: overlapping chunks need a power-of-two length");
        if (count > 65535) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many non-power-of-two chunks in one call");
        dim3 grid((unsigned)((n + 255) / 256), (unsigned)count);
        hipLaunchKernelGGL(k_dft_direct<T>, grid, dim3(256), 0, s, (const v2<T> *)in, (v2<T> *)out, (int)n,
                           (const T *)window, (const v2<T> *)twiddle, (int)center_dc);
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Bluestein's algorithm for chunk lengths that are not powers of two (Complex<f32>): with w_m = e^{+j pi m^2 / n},
//   X[k] = conj(w_k) * sum_i (x[i] win[i] conj(w_i)) w_{k-i}
// a linear convolution, done as two M-point transforms (M = power of two >= 2 n - 1) by the kernels above:
//   a = x * c (c = win * conj(w), zero-padded to M)  ->  A = F(a)  ->  conj(A * B), B = F(w arranged circularly) / M
//   ->  F again (= conj of the inverse transform)  ->  X[k] = conj(result[k] * w_k).
// Three elementwise kernels around the two transforms; rr_fourier::transform_dev drives them.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_bs_pre(const v2<T> *__restrict__ head, long n_head, const v2<T> *__restrict__ in,
                                                long hop, long n, long M, const v2<T> *__restrict__ c, v2<T> *__restrict__ ws) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    v2<T> v;
    v.x = 0;
    v.y = 0;
    if (m < n) {
        const long i = (long)blockIdx.y * hop - n_head + m;
        const v2<T> x = (i >= 0) ? in[i] : head[n_head + i];
        v = cmul<T>(x, c[m]);
    }
    ws[(size_t)blockIdx.y * M + m] = v;
}
template <class T>
__global__ __launch_bounds__(256) void k_bs_mul(v2<T> *__restrict__ ws, const v2<T> *__restrict__ B, long M) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    v2<T> *p = ws + (size_t)blockIdx.y * M + m;
    const v2<T> v = cmul<T>(*p, B[m]);
    v2<T> o;
    o.x = v.x;
    o.y = -v.y;
    *p = o;
}
template <class T>
__global__ __launch_bounds__(256) void k_bs_post(const v2<T> *__restrict__ ws, const v2<T> *__restrict__ w, long n, long M,
                                                 v2<T> *__restrict__ out, int center_dc) {
    const long kk = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (kk >= n) return;
    const v2<T> v = cmul<T>(ws[(size_t)blockIdx.y * M + kk], w[kk]);
    long o = kk + (center_dc ? n / 2 : 0);  // rotate_right(n / 2)
    if (o >= n) o -= n;
    v2<T> r;
    r.x = v.x;
    r.y = -v.y;
    out[(size_t)blockIdx.y * n + o] = r;
}
int launch_bs_pre(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, size_t M,
                  const void *c, void *ws, size_t frames) {
    if (frames == 0) return RR_OK;
    if (frames > 65535) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many frames in one Bluestein pass");
    const dim3 grid((unsigned)((M + 255) / 256), (unsigned)frames);
    if (dtype == RR_F32)
        hipLaunchKernelGGL(k_bs_pre<float>, grid, dim3(256), 0, s, (const float2 *)head, (long)n_head, (const float2 *)in,
                           (long)hop, (long)n, (long)M, (const float2 *)c, (float2 *)ws);
    else
        hipLaunchKernelGGL(k_bs_pre<double>, grid, dim3(256), 0, s, (const double2 *)head, (long)n_head, (const double2 *)in,
                           (long)hop, (long)n, (long)M, (const double2 *)c, (double2 *)ws);
    RR_HIP(hipGetLastError());
    return RR_OK;
}
int launch_bs_mul(int dtype, hipStream_t s, void *ws, const void *B, size_t M, size_t frames) {
    if (frames == 0) return RR_OK;
    const dim3 grid((unsigned)((M + 255) / 256), (unsigned)frames);
    if (dtype == RR_F32)
        hipLaunchKernelGGL(k_bs_mul<float>, grid, dim3(256), 0, s, (float2 *)ws, (const float2 *)B, (long)M);
    else
        hipLaunchKernelGGL(k_bs_mul<double>, grid, dim3(256), 0, s, (double2 *)ws, (const double2 *)B, (long)M);
    RR_HIP(hipGetLastError());
    return RR_OK;
}
int launch_bs_post(int dtype, hipStream_t s, const void *ws, const void *w, size_t n, size_t M, void *out, bool center_dc,
                   size_t frames) {
    if (frames == 0) return RR_OK;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)frames);
    if (dtype == RR_F32)
        hipLaunchKernelGGL(k_bs_post<float>, grid, dim3(256), 0, s, (const float2 *)ws, (const float2 *)w, (long)n, (long)M,
                           (float2 *)out, (int)center_dc);
    else
        hipLaunchKernelGGL(k_bs_post<double>, grid, dim3(256), 0, s, (const double2 *)ws, (const double2 *)w, (long)n, (long)M,
                           (double2 *)out, (int)center_dc);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Power-of-two lengths beyond one LDS tile (n = 2^14 .. 2^24; analysis.rs:82-115 accepts any length): the
// four-step decomposition n = N1 N2 through HBM, two launches of one batched, strided radix-2 kernel:
//   A: for every n2: Y[k1][n2] = W_n^(n2 k1) * DFT_N1 over n1 of (window x)[N2 n1 + n2]     (workspace, [k1][n2])
//   B: for every k1: X[k1 + N1 k2] = DFT_N2 over n2 of Y[k1][n2]
// A workgroup takes C neighbouring sequences, so that every access touches C (A) or N2 (B) contiguous elements.
// The twiddles W_n^(n2 k1) are evaluated in f64 (sincospi of the exactly reduced phase), whatever the data type.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_fft_bs(const v2<T> *__restrict__ in, v2<T> *__restrict__ out, int N, int C,
                                                long in_seq, long in_el, long out_seq, long out_el, long chunk_stride,
                                                const T *__restrict__ window, const v2<T> *__restrict__ tw, long tw_M,
                                                int rot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2<T> *a = reinterpret_cast<v2<T> *>(smem);
    v2<T> *b = a + (size_t)C * N;
    const long s0 = (long)blockIdx.x * C;
    const v2<T> *src = in + (size_t)blockIdx.y * chunk_stride;
    v2<T> *dst = out + (size_t)blockIdx.y * chunk_stride;
    const int total = C * N;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        int sq, i;
        if (in_seq == 1) {
            i = idx / C;
            sq = idx - i * C;
        } else {
            sq = idx / N;
            i = idx - sq * N;
        }
        const long off = (s0 + sq) * in_seq + (long)i * in_el;
        v2<T> v = src[off];
        if (window) {
            const T w = window[off];
            v.x *= w;
            v.y *= w;
        }
        a[sq * N + i] = v;
    }
    __syncthreads();
    const int half = N >> 1;
    for (int ns = 1; ns < N; ns <<= 1) {
        const int tstride = half / ns;
        for (int jj = threadIdx.x; jj < C * half; jj += blockDim.x) {
            const int sq = jj / half, j = jj - sq * half;
            const int k = j & (ns - 1);
            const v2<T> *as = a + sq * N;
            v2<T> *bs = b + sq * N;
            const v2<T> u = as[j];
            const v2<T> v = cmul<T>(as[j + half], tw[k * tstride]);
            const int j0 = ((j - k) << 1) + k;
            v2<T> sm, df;
            sm.x = u.x + v.x;
            sm.y = u.y + v.y;
            df.x = u.x - v.x;
            df.y = u.y - v.y;
            bs[j0] = sm;
            bs[j0 + ns] = df;
        }
        __syncthreads();
        v2<T> *t = a;
        a = b;
        b = t;
    }
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        int sq, k;
        if (out_seq == 1) {
            k = idx / C;
            sq = idx - k * C;
        } else {
            sq = idx / N;
            k = idx - sq * N;
        }
        v2<T> v = a[sq * N + k];
        if (tw_M) {
            const long ph = ((s0 + sq) * (long)k) % tw_M;  // < 2^48: exact
            double sn, cs;
            sincospi(-2.0 * (double)ph / (double)tw_M, &sn, &cs);
            v2<T> w;
            w.x = (T)cs;
            w.y = (T)sn;
            v = cmul<T>(v, w);
        }
        int ko = k + rot;
        if (ko >= N) ko -= N;
        dst[(s0 + sq) * out_seq + (long)ko * out_el] = v;
    }
}

bool fft_big_supported(size_t n) { return is_pow2_n(n) && n >= 4 && n <= ((size_t)1 << 24); }
void fft_big_split(size_t n, size_t *N1, size_t *N2) {
    int lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
    *N1 = (size_t)1 << (lg / 2);
    *N2 = n / *N1;
}

// in, out: `count` chunks of n = N1 N2 elements each; ws: workspace of count * n elements; tw1 / tw2: e^{-j 2 pi k / N1}
// (N1 / 2 entries) and e^{-j 2 pi k / N2} (N2 / 2 entries); window: n reals or null
template <class T>
static int launch_fft_big_t(hipStream_t s, const void *in, void *out, void *ws, size_t n, size_t count, const void *window,
                            const void *tw1, const void *tw2, bool center_dc) {
    size_t N1, N2;
    fft_big_split(n, &N1, &N2);
    auto fn = k_fft_bs<T>;
    const size_t esz = sizeof(v2<T>), budget = 64 * 1024;
    auto tile = [&](size_t N, size_t nseq) {
        size_t c = budget / (2 * N * esz);
        if (c < 1) c = 1;
        if (c > 32) c = 32;
        while (c > nseq) c /= 2;
        return c;
    };
    const size_t CA = tile(N1, N2), CB = tile(N2, N1);
    const size_t ldsA = 2 * CA * N1 * esz, ldsB = 2 * CB * N2 * esz;
    RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), ldsA > ldsB ? ldsA : ldsB));
    if (count > 65535) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many chunks in one call");
    hipLaunchKernelGGL(fn, dim3((unsigned)(N2 / CA), (unsigned)count), dim3(256), ldsA, s, (const v2<T> *)in, (v2<T> *)ws, (int)N1,
                       (int)CA, 1L, (long)N2, 1L, (long)N2, (long)n, (const T *)window, (const v2<T> *)tw1, (long)n, 0);
    hipLaunchKernelGGL(fn, dim3((unsigned)(N1 / CB), (unsigned)count), dim3(256), ldsB, s, (const v2<T> *)ws, (v2<T> *)out, (int)N2,
                       (int)CB, (long)N2, 1L, 1L, (long)N1, (long)n, (const T *)nullptr, (const v2<T> *)tw2, 0L,
                       center_dc ? (int)(N2 / 2) : 0);
    RR_HIP(hipGetLastError());
    return RR_OK;
}
int launch_fft_big(int dtype, hipStream_t s, const void *in, void *out, void *ws, size_t n, size_t count, const void *window,
                   const void *tw1, const void *tw2, bool center_dc) {
    if (count == 0) return RR_OK;
    if (dtype == RR_F32) return launch_fft_big_t<float>(s, in, out, ws, n, count, window, tw1, tw2, center_dc);
    return launch_fft_big_t<double>(s, in, out, ws, n, count, window, tw1, tw2, center_dc);
}

// ---------------------------------------------------------------------------
// Four-step transform, second form (the default): the column transforms become ROW transforms between three tiled
// transposes, so that both sets of sub-transforms run through the fast contiguous kernels (k_fft64 .. k_fft8192)
// instead of the strided radix-2 kernel above:
//   T1  y[n2][n1] = w[N2 n1 + n2] x[N2 n1 + n2]          rows of N1 -> DFT_N1 -> Y[n2][k1]
//   T2  z[k1][n2] = Y[n2][k1] W_N^(n2 k1)                 rows of N2 -> DFT_N2 -> Z[k1][k2]
//   T3  X[k1 + N1 k2] = Z[k1][k2]                          (rows rotated by N2 / 2 for center_dc)
// W_N^e = tA[e >> h] tB[e & (2^h - 1)], two tables of about sqrt(N) entries computed in f64.
// A 32 x 32 tile per workgroup through LDS (rows padded to 33), every global access 256 contiguous bytes.
// ---------------------------------------------------------------------------
template <class T, int MODE>
__global__ __launch_bounds__(256) void k_transpose_mul(const v2<T> *__restrict__ in, v2<T> *__restrict__ out, int R, int C,
                                                       const T *__restrict__ window, const v2<T> *__restrict__ tB,
                                                       const v2<T> *__restrict__ tA, int h, int rot_rows) {
    __shared__ v2<T> tile[32][33];
    const size_t chunk = (size_t)blockIdx.z * R * C;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        v2<T> v = in[chunk + (size_t)r * C + c];
        if (MODE == 1) {
            const T w = window[(size_t)r * C + c];
            v.x *= w;
            v.y *= w;
        } else if (MODE == 2) {
            const unsigned e = ((unsigned)r * (unsigned)c) & ((unsigned)R * (unsigned)C - 1u);
            const v2<T> w = cmul<T>(tA[e >> h], tB[e & ((1u << h) - 1u)]);
            v = cmul<T>(v, w);
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = c0 + ty + 8 * i;
        const int r = r0 + tx;
        const v2<T> v = tile[tx][ty + 8 * i];
        c += rot_rows;
        if (c >= C) c -= C;
        out[chunk + (size_t)c * R + r] = v;
    }
}

int launch_transpose_mul(int dtype, hipStream_t s, const void *in, void *out, size_t R, size_t C, size_t count, int mode,
                         const void *window, const void *tB, const void *tA, int h, size_t rot_rows) {
    if (count == 0) return RR_OK;
    if (R % 32 || C % 32 || count > 65535 || R * C > ((size_t)1 << 24))
        RR_FAIL(RR_ERR_BAD_ARG, "transpose: %zu x %zu x %zu is outside the tiled kernel's range", R, C, count);
    const dim3 grid((unsigned)(C / 32), (unsigned)(R / 32), (unsigned)count);
#define RR_TR(TT, VV, MM)                                                                                             \
    hipLaunchKernelGGL((k_transpose_mul<TT, MM>), grid, dim3(256), 0, s, (const VV *)in, (VV *)out, (int)R, (int)C, \
                       (const TT *)window, (const VV *)tB, (const VV *)tA, h, (int)rot_rows)
    if (dtype == RR_F32) {
        if (mode == 1) RR_TR(float, float2, 1);
        else if (mode == 2) RR_TR(float, float2, 2);
        else RR_TR(float, float2, 0);
    } else {
        if (mode == 1) RR_TR(double, double2, 1);
        else if (mode == 2) RR_TR(double, double2, 2);
        else RR_TR(double, double2, 0);
    }
#undef RR_TR
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Four-step transform, third form (the default for 2^13 / 2^14 .. 2^18 points): TWO passes over HBM instead of five.
//   pass A  a workgroup takes a bundle of C neighbouring columns n2 (C x 8 or 16 bytes = one 128-byte line per row n1),
//           multiplies by the window, transforms the C columns over n1 (N1 points) in LDS, multiplies by W_N^(n2 k1) and
//           writes Y[k1][n2] back in the same [row][column] shape
//   pass B  a workgroup takes C neighbouring rows k1 of Y (contiguous), transforms them over n2 (N2 points) in LDS and
//           writes X[k1 + N1 k2]: for every k2 its C results are neighbours in memory (a 128-byte line)
// Both passes are the same kernel: a tile of Np x C elements in LDS (element (n, c) at n SN + c SC), in-place radix-4
// decimation-in-frequency passes with a lane per butterfly (one radix-2 pass at the end when log2 Np is odd), the result
// in digit-reversed order, which the store undoes in its address.  Twiddles W_Np^k from an LDS copy of the sub-transform's
// table.  The tile is Np x 128 bytes; used for Np <= 512 (N <= 2^18), beyond that the five-launch form takes over.
// ---------------------------------------------------------------------------
// Bluestein's element-wise stages folded into the passes (BS = 1; everything null / 0 otherwise):
//   pass A load   x c with c = window conj(chirp), zero beyond nvalid, frames from [ head | in ] at a hop  (pre != null)
//   pass A load   plain                                                                                    (window == null)
//   pass B store  conj(X[k] post[k]) in natural order                                                      (post != null, nout == 0)
//   pass B store  the bins k < nout only, conj(X[k] post[k]) at (k + rot) mod nout of the chunk's nout bins  (post != null, nout > 0)
template <class T>
struct TileBs {
    const v2<T> *head;
    long n_head, hop;
    const v2<T> *pre;
    int nvalid;
    const v2<T> *post;
    int nout;
    // fast convolution (the long Filter, rr_filter's conv path): frames may reach beyond the input - samples at g >= in_limit
    // read as zero (0: no limit) - and the last frame's cut may be shorter: nothing is stored at or beyond out_limit (0: no limit)
    long in_limit, out_limit;
};

template <class T, int MODE, int BS = 0>
__global__ __launch_bounds__(1024) void k_fft_tile(const v2<T> *__restrict__ in, v2<T> *__restrict__ out, int Np, int lgNp, int No,
                                                   const T *__restrict__ window, const v2<T> *__restrict__ twNp,
                                                   const v2<T> *__restrict__ tB, const v2<T> *__restrict__ tA, int h, int rot,
                                                   TileBs<T> bs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
    constexpr int C = 128 / (int)sizeof(v2<T>);
    v2<T> *const tile = reinterpret_cast<v2<T> *>(tile_raw);
    // MODE 0: lanes run along the columns (element (n, c) at n C + c); MODE 1: along a row (element (n, c) at c (Np + 1) + n)
    const int SN = MODE == 0 ? C : 1, SC = MODE == 0 ? 1 : Np + 1;
    v2<T> *const tw = tile + (MODE == 0 ? Np * C : C * (Np + 1));
    const int nt = blockDim.x, t = threadIdx.x;
    const size_t chunk = (size_t)blockIdx.y * (size_t)Np * (size_t)No;
    const int g0 = blockIdx.x * C;
    for (int i = t; i < Np; i += nt) tw[i] = twNp[i];
    // (four loads in flight per lane: left as one loop the compiler keeps a single load between a wait and the LDS store)
    if (MODE == 0) {
        for (int idx0 = t; idx0 < Np * C; idx0 += 4 * nt) {
            v2<T> v[4];
            T w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * nt;
                if (idx < Np * C) {
                    const size_t e = (size_t)(idx / C) * No + g0 + idx % C;
                    if (BS && bs.pre) {
                        v[u] = v2<T>{(T)0, (T)0};
                        if (e < (size_t)bs.nvalid) {
                            const long g = (long)blockIdx.y * bs.hop - bs.n_head + (long)e;
                            if (bs.in_limit == 0 || g < bs.in_limit) v[u] = cmul<T>(g >= 0 ? in[g] : bs.head[bs.n_head + g], bs.pre[e]);
                        }
                        w[u] = (T)1;
                    } else {
                        v[u] = in[chunk + e];
                        w[u] = (BS && !window) ? (T)1 : window[e];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * nt;
                if (idx < Np * C) tile[idx] = v2<T>{v[u].x * w[u], v[u].y * w[u]};  // (n C + c = idx)
            }
        }
    } else {
        for (int idx0 = t; idx0 < Np * C; idx0 += 4 * nt) {
            v2<T> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * nt;
                if (idx < Np * C) v[u] = in[chunk + (size_t)(g0 + (idx >> lgNp)) * Np + (idx & (Np - 1))];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * nt;
                if (idx < Np * C) tile[(idx >> lgNp) * (Np + 1) + (idx & (Np - 1))] = v[u];
            }
        }
    }
    __syncthreads();
    int L = Np, lgL = lgNp;
    for (; L >= 4; L >>= 2, lgL -= 2) {
        const int q = L >> 2, lgq = lgL - 2, step = Np >> lgL;
        for (int b = t; b < (Np >> 2) * C; b += nt) {
            int jj, c;
            if (MODE == 0) {
                c = b % C;
                jj = b / C;
            } else {
                jj = b & ((Np >> 2) - 1);
                c = b >> (lgNp - 2);
            }
            const int blk = jj >> lgq, j = jj & (q - 1);
            v2<T> *p = tile + ((blk << lgL) + j) * SN + c * SC;
            const v2<T> a0 = p[0], a1 = p[q * SN], a2 = p[2 * q * SN], a3 = p[3 * q * SN];
            const v2<T> s02 = {a0.x + a2.x, a0.y + a2.y}, d02 = {a0.x - a2.x, a0.y - a2.y};
            const v2<T> s13 = {a1.x + a3.x, a1.y + a3.y}, d13 = {a1.x - a3.x, a1.y - a3.y};
            // W_4 = -j: y1 = d02 - j d13, y3 = d02 + j d13
            const v2<T> y0 = {s02.x + s13.x, s02.y + s13.y}, y2 = {s02.x - s13.x, s02.y - s13.y};
            const v2<T> y1 = {d02.x + d13.y, d02.y - d13.x}, y3 = {d02.x - d13.y, d02.y + d13.x};
            p[0] = y0;
            if (j == 0) {
                p[q * SN] = y1;
                p[2 * q * SN] = y2;
                p[3 * q * SN] = y3;
            } else {
                const int k1 = j * step;
                p[q * SN] = cmul<T>(y1, tw[k1]);
                p[2 * q * SN] = cmul<T>(y2, tw[2 * k1]);
                p[3 * q * SN] = cmul<T>(y3, tw[3 * k1]);
            }
        }
        __syncthreads();
    }
    if (L == 2) {
        for (int b = t; b < (Np >> 1) * C; b += nt) {
            int jj, c;
            if (MODE == 0) {
                c = b % C;
                jj = b / C;
            } else {
                jj = b & ((Np >> 1) - 1);
                c = b >> (lgNp - 1);
            }
            v2<T> *p = tile + (2 * jj) * SN + c * SC;
            const v2<T> a0 = p[0], a1 = p[SN];
            p[0] = v2<T>{a0.x + a1.x, a0.y + a1.y};
            p[SN] = v2<T>{a0.x - a1.x, a0.y - a1.y};
        }
        __syncthreads();
    }
    // position p holds frequency k = d1 + 4 d2 + 16 d3 + .. of p = d1 Np / 4 + d2 Np / 16 + .. (the radix-2 digit last)
    for (int idx = t; idx < Np * C; idx += nt) {
        const int p = idx / C, c = idx % C;
        int k = 0, sh = 0, rem = p, lgs = lgNp;
        while (lgs >= 2) {
            lgs -= 2;
            const int d = rem >> lgs;
            rem -= d << lgs;
            k += d << sh;
            sh += 2;
        }
        if (lgs == 1) k += rem << sh;
        v2<T> v = tile[p * SN + c * SC];
        if (MODE == 0) {
            const unsigned e = (unsigned)k * (unsigned)(g0 + c);  // < N1 N2 <= 2^20
            v = cmul<T>(v, cmul<T>(tA[e >> h], tB[e & ((1u << h) - 1u)]));
            out[chunk + (size_t)k * No + g0 + c] = v;
        } else if (BS && bs.post) {
            const size_t ko = (size_t)k * No + g0 + c;  // X[k1 + N1 k2]
            if (bs.nout == 0) {
                const v2<T> y = cmul<T>(v, bs.post[ko]);
                st_result<T>(out + (chunk + ko), v2<T>{y.x, -y.y});
            } else if (ko < (size_t)bs.nout) {
                const v2<T> y = cmul<T>(v, bs.post[ko]);
                size_t o = ko + (size_t)rot;  // rotate_right(n / 2)
                if (o >= (size_t)bs.nout) o -= (size_t)bs.nout;
                const size_t oo = (size_t)blockIdx.y * (size_t)bs.nout + o;
                if (bs.out_limit == 0 || oo < (size_t)bs.out_limit) st_result<T>(out + oo, v2<T>{y.x, -y.y});
            }
        } else {
            int kk = k + rot;
            if (kk >= Np) kk -= Np;
            st_result<T>(out + (chunk + (size_t)kk * No + g0 + c), v);
        }
    }
}

bool fft_tile_supported(int dtype, size_t N1, size_t N2) {
    const size_t C = dtype == RR_F32 ? 16 : 8;
    // (the kernel itself takes Np = 1024, a 136 KiB tile and one workgroup per CU: measured 0.225 / 0.252 ms per 2^24 samples at
    // 2^19 / 2^20 points against 0.225 / 0.223 for the five launches, so those stay with the transposes)
    auto ok = [&](size_t n) { return n >= 64 && n <= 512 && is_pow2_n(n); };
    return ok(N1) && ok(N2) && N1 >= C && N2 >= C;
}

// pass 0: in = `count` chunks of N1 x N2 (row n1, column n2), out = the same shape holding Y[k1][n2] W_N^(n2 k1);
// pass 1: in = Y, out = X[k1 + N1 k2] (rows k2 rotated by rot for center_dc)
template <class T>
static int launch_fft_tile_t(hipStream_t s, int pass, const void *in, void *out, size_t N1, size_t N2, size_t count,
                             const void *window, const void *twNp, const void *tB, const void *tA, int h, size_t rot) {
    constexpr size_t C = 128 / sizeof(v2<T>);
    const size_t Np = pass == 0 ? N1 : N2, No = pass == 0 ? N2 : N1;
    int lg = 0;
    while (((size_t)1 << lg) < Np) ++lg;
    const size_t lds = (pass == 0 ? Np * C : C * (Np + 1)) * sizeof(v2<T>) + Np * sizeof(v2<T>);
    size_t nt = Np * C / 4;  // a lane per radix-4 butterfly
    if (nt > 1024) nt = 1024;
    if (nt < 256) nt = 256;
    const dim3 grid((unsigned)(No / C), (unsigned)count);
    if (pass == 0) {
        auto fn = k_fft_tile<T, 0>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, grid, dim3((unsigned)nt), lds, s, (const v2<T> *)in, (v2<T> *)out, (int)Np, lg, (int)No,
                           (const T *)window, (const v2<T> *)twNp, (const v2<T> *)tB, (const v2<T> *)tA, h, 0, TileBs<T>{});
    } else {
        auto fn = k_fft_tile<T, 1>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, grid, dim3((unsigned)nt), lds, s, (const v2<T> *)in, (v2<T> *)out, (int)Np, lg, (int)No,
                           (const T *)nullptr, (const v2<T> *)twNp, (const v2<T> *)nullptr, (const v2<T> *)nullptr, 0, (int)rot,
                           TileBs<T>{});
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}
// The passes with Bluestein's element-wise stages folded in (TileBs): stage 0 = pass A with x c at the load, 1 = pass B with
// conj(. B) at the store, 2 = pass A plain, 3 = pass B with conj(. chirp), the first n bins only, rotated by rot ELEMENTS.
template <class T>
static int launch_fft_tile_bs_t(hipStream_t s, int stage, const void *head, size_t n_head, const void *in, size_t hop, void *out,
                                size_t N1, size_t N2, size_t count, size_t n, const void *table, const void *twNp, const void *tB,
                                const void *tA, int h, size_t rot, long in_limit, long out_limit) {
    constexpr size_t C = 128 / sizeof(v2<T>);
    const bool passA = stage == 0 || stage == 2;
    const size_t Np = passA ? N1 : N2, No = passA ? N2 : N1;
    int lg = 0;
    while (((size_t)1 << lg) < Np) ++lg;
    const size_t lds = (passA ? Np * C : C * (Np + 1)) * sizeof(v2<T>) + Np * sizeof(v2<T>);
    size_t nt = Np * C / 4;
    if (nt > 1024) nt = 1024;
    if (nt < 256) nt = 256;
    const dim3 grid((unsigned)(No / C), (unsigned)count);
    TileBs<T> bs{};
    bs.in_limit = in_limit;
    bs.out_limit = out_limit;
    if (stage == 0) {
        bs.head = (const v2<T> *)head;
        bs.n_head = (long)n_head;
        bs.hop = (long)hop;
        bs.pre = (const v2<T> *)table;
        bs.nvalid = (int)n;
    } else if (stage == 1) {
        bs.post = (const v2<T> *)table;
    } else if (stage == 3) {
        bs.post = (const v2<T> *)table;
        bs.nout = (int)n;
    }
    if (passA) {
        auto fn = k_fft_tile<T, 0, 1>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, grid, dim3((unsigned)nt), lds, s, (const v2<T> *)in, (v2<T> *)out, (int)Np, lg, (int)No,
                           (const T *)nullptr, (const v2<T> *)twNp, (const v2<T> *)tB, (const v2<T> *)tA, h, 0, bs);
    } else {
        auto fn = k_fft_tile<T, 1, 1>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, grid, dim3((unsigned)nt), lds, s, (const v2<T> *)in, (v2<T> *)out, (int)Np, lg, (int)No,
                           (const T *)nullptr, (const v2<T> *)twNp, (const v2<T> *)nullptr, (const v2<T> *)nullptr, 0, (int)rot, bs);
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}
int launch_fft_tile_bs(int dtype, hipStream_t s, int stage, const void *head, size_t n_head, const void *in, size_t hop, void *out,
                       size_t N1, size_t N2, size_t count, size_t n, const void *table, const void *twNp, const void *tB,
                       const void *tA, int h, size_t rot, long in_limit, long out_limit) {
    if (count == 0) return RR_OK;
    if (!fft_tile_supported(dtype, N1, N2) || count > 65535 || n > N1 * N2)
        RR_FAIL(RR_ERR_BAD_ARG, "tile transform: %zu x %zu x %zu is outside the kernel's range", N1, N2, count);
    if (dtype == RR_F32)
        return launch_fft_tile_bs_t<float>(s, stage, head, n_head, in, hop, out, N1, N2, count, n, table, twNp, tB, tA, h, rot, in_limit, out_limit);
    return launch_fft_tile_bs_t<double>(s, stage, head, n_head, in, hop, out, N1, N2, count, n, table, twNp, tB, tA, h, rot, in_limit, out_limit);
}

int launch_fft_tile(int dtype, hipStream_t s, int pass, const void *in, void *out, size_t N1, size_t N2, size_t count,
                    const void *window, const void *twNp, const void *tB, const void *tA, int h, size_t rot) {
    if (count == 0) return RR_OK;
    if (!fft_tile_supported(dtype, N1, N2) || count > 65535)
        RR_FAIL(RR_ERR_BAD_ARG, "tile transform: %zu x %zu x %zu is outside the kernel's range", N1, N2, count);
    if (dtype == RR_F32) return launch_fft_tile_t<float>(s, pass, in, out, N1, N2, count, window, twNp, tB, tA, h, rot);
    return launch_fft_tile_t<double>(s, pass, in, out, N1, N2, count, window, twNp, tB, tA, h, rot);
}

// ---------------------------------------------------------------------------
// Mixed-radix transform: chunk lengths n = 2^a 3^b 5^c 7^d 11^e 13^f (up to 8192 in f32, 4096 in f64) that are not powers of two (1000, 3000, 4800, 1001 ..; analysis.rs:82-115
// accepts any length), one workgroup per chunk, the whole transform in ONE LDS image of n elements: in-place decimation-in-
// frequency passes of radix 5, 4, 3, 2 with a lane per butterfly, the result in mixed-radix digit-reversed order which the store
// undoes in its address.  Bluestein's algorithm needs two power-of-two transforms of M >= 2 n - 1 points for the same chunk
// (n = 3000: M = 8192 and five launches; n = 1000: two 4096-point transforms in one kernel); here the work is n log n.
//   pass with block length L and radix r, q = L / r: inputs a_s = x[blk L + j + s q], outputs
//   x[blk L + j + m q] = W_L^(j m) sum_s a_s W_r^(s m); position p = d_1 n / r_1 + d_2 n / (r_1 r_2) + .. ends as frequency
//   k = d_1 + r_1 (d_2 + r_2 (..)).
// Frames come from [ head | in ] at any hop (Fourier: hop = n; Stft: overlapping spans).
// ---------------------------------------------------------------------------
struct MixedPlan {
    int nrad;
    unsigned char radix[12];
    unsigned short q[12];   // butterflies per block of the pass = L / r
    float rq[12];           // 1 / q: b / q = (int)((b + 0.5) rq) exactly for b, q <= 4096 (the product's error, (b / q) 1.2e-7, stays
                            // below the 0.5 / q the half moves it away from a whole number) - an integer division by a run-time
                            // value costs ~35 instructions; checked for every b <= 8192, q <= 4096, and the passes are index arithmetic more than anything else
};
__device__ __forceinline__ int div_small(int b, float rq) { return (int)(((float)b + 0.5f) * rq); }

// radix 7, 11, 13 (lengths with those prime factors, at most a pass or two each): the R x R sum itself, W_R^k = tw[k ws]
template <class T, int R>
__device__ __forceinline__ void odd_bfly(v2<T> *p, int q, const v2<T> *__restrict__ tw, int k1, int ws) {
    v2<T> a[R];
#pragma unroll
    for (int s = 0; s < R; ++s) a[s] = p[s * q];
    v2<T> y0 = a[0];
#pragma unroll
    for (int s = 1; s < R; ++s) {
        y0.x += a[s].x;
        y0.y += a[s].y;
    }
    p[0] = y0;
#pragma unroll
    for (int m = 1; m < R; ++m) {
        v2<T> acc = a[0];
#pragma unroll
        for (int s = 1; s < R; ++s) {
            const v2<T> w = tw[((s * m) % R) * ws];
            acc.x += a[s].x * w.x - a[s].y * w.y;
            acc.y += a[s].x * w.y + a[s].y * w.x;
        }
        p[m * q] = cmul<T>(acc, tw[m * k1]);
    }
}

// one radix-r butterfly (r = 2, 3, 4, 5; 7, 11, 13) of a decimation-in-frequency pass, in place: inputs p[s q], outputs p[m q] =
// W^(m k1) sum_s a_s W_r^(s m) with W^k = tw[k] (a table of ntab entries)
// ODD: the kernel instance that also carries the radix-7 / 11 / 13 butterflies (their 13 live inputs cost the other lengths
// a third of their residency when they shared one instance: 1000 points 0.139 -> 0.225 ms per 2^24 samples)
template <class T, bool ODD>
__device__ __forceinline__ void mixed_bfly(v2<T> *p, int q, int r, const v2<T> *__restrict__ tw, int k1, int ntab) {
    if (ODD && r > 5) {
        if constexpr (ODD) {
            if (r == 7) odd_bfly<T, 7>(p, q, tw, k1, ntab / 7);
            else if (r == 11) odd_bfly<T, 11>(p, q, tw, k1, ntab / 11);
            else odd_bfly<T, 13>(p, q, tw, k1, ntab / 13);
        }
    } else if (r == 4) {
        const v2<T> a0 = p[0], a1 = p[q], a2 = p[2 * q], a3 = p[3 * q];
        const v2<T> s02 = {a0.x + a2.x, a0.y + a2.y}, d02 = {a0.x - a2.x, a0.y - a2.y};
        const v2<T> s13 = {a1.x + a3.x, a1.y + a3.y}, d13 = {a1.x - a3.x, a1.y - a3.y};
        p[0] = v2<T>{s02.x + s13.x, s02.y + s13.y};
        const v2<T> y1 = {d02.x + d13.y, d02.y - d13.x}, y2 = {s02.x - s13.x, s02.y - s13.y};
        const v2<T> y3 = {d02.x - d13.y, d02.y + d13.x};
        p[q] = cmul<T>(y1, tw[k1]);
        p[2 * q] = cmul<T>(y2, tw[2 * k1]);
        p[3 * q] = cmul<T>(y3, tw[3 * k1]);
    } else if (r == 2) {
        const v2<T> a0 = p[0], a1 = p[q];
        p[0] = v2<T>{a0.x + a1.x, a0.y + a1.y};
        p[q] = cmul<T>(v2<T>{a0.x - a1.x, a0.y - a1.y}, tw[k1]);
    } else if (r == 3) {
        // W_3 = -1/2 - j sqrt(3)/2
        const T h = (T)0.86602540378443864676;
        const v2<T> a0 = p[0], a1 = p[q], a2 = p[2 * q];
        const v2<T> s = {a1.x + a2.x, a1.y + a2.y}, d = {a1.x - a2.x, a1.y - a2.y};
        p[0] = v2<T>{a0.x + s.x, a0.y + s.y};
        const v2<T> m = {a0.x - (T)0.5 * s.x, a0.y - (T)0.5 * s.y};
        // -j h d = (h d.y, -h d.x)
        const v2<T> y1 = {m.x + h * d.y, m.y - h * d.x}, y2 = {m.x - h * d.y, m.y + h * d.x};
        p[q] = cmul<T>(y1, tw[k1]);
        p[2 * q] = cmul<T>(y2, tw[2 * k1]);
    } else {  // r == 5
        const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;  // cos(2 pi / 5), cos(4 pi / 5)
        const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;   // sin(2 pi / 5), sin(4 pi / 5)
        const v2<T> a0 = p[0], a1 = p[q], a2 = p[2 * q], a3 = p[3 * q], a4 = p[4 * q];
        const v2<T> s14 = {a1.x + a4.x, a1.y + a4.y}, d14 = {a1.x - a4.x, a1.y - a4.y};
        const v2<T> s23 = {a2.x + a3.x, a2.y + a3.y}, d23 = {a2.x - a3.x, a2.y - a3.y};
        p[0] = v2<T>{a0.x + s14.x + s23.x, a0.y + s14.y + s23.y};
        const v2<T> m1 = {a0.x + c1 * s14.x + c2 * s23.x, a0.y + c1 * s14.y + c2 * s23.y};
        const v2<T> m2 = {a0.x + c2 * s14.x + c1 * s23.x, a0.y + c2 * s14.y + c1 * s23.y};
        // forward kernel e^{-j ..}: y_1 = m1 - j (s1 d14 + s2 d23), y_4 = m1 + j (..); y_2 = m2 - j (s2 d14 - s1 d23), y_3 = m2 + j (..)
        const v2<T> u1 = {s1 * d14.x + s2 * d23.x, s1 * d14.y + s2 * d23.y};
        const v2<T> u2 = {s2 * d14.x - s1 * d23.x, s2 * d14.y - s1 * d23.y};
        const v2<T> y1 = {m1.x + u1.y, m1.y - u1.x}, y4 = {m1.x - u1.y, m1.y + u1.x};
        const v2<T> y2 = {m2.x + u2.y, m2.y - u2.x}, y3 = {m2.x - u2.y, m2.y + u2.x};
        p[q] = cmul<T>(y1, tw[k1]);
        p[2 * q] = cmul<T>(y2, tw[2 * k1]);
        p[3 * q] = cmul<T>(y3, tw[3 * k1]);
        p[4 * q] = cmul<T>(y4, tw[4 * k1]);
    }
}

// TWLDS: the twiddle table W_n^k is copied into LDS behind the image (2 n elements per workgroup; lengths up to half the
// largest image) - a pass then waits for LDS reads instead of L2 hits between its butterflies' loads and stores
template <class T, bool TWLDS, bool ODD>
__global__ __launch_bounds__(1024) void k_fft_mixed(const v2<T> *__restrict__ head, long n_head, const v2<T> *__restrict__ in,
                                                   long base0, long hop, int n, int branches, MixedPlan plan,
                                                   const T *__restrict__ window, const v2<T> *__restrict__ tw_g,
                                                   v2<T> *__restrict__ out, int center_dc, unsigned count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mixed_raw[];
    v2<T> *const x = reinterpret_cast<v2<T> *>(mixed_raw);
    const int nt = blockDim.x, t = threadIdx.x;
    const unsigned fr = blockIdx.x;
    if (fr >= count) return;
    const v2<T> *tw = tw_g;
    if constexpr (TWLDS) {
        v2<T> *twl = x + n;
        for (int i = t; i < n; i += nt) twl[i] = tw_g[i];
        tw = twl;  // (visible after the barrier behind the load of the samples)
    }
    // frame fr starts base0 + fr hop samples into `in` (negative: inside head, which ends where `in` begins); with
    // branches > 1 the frame is the fold of that many windowed chunks of n samples (the polyphase channelizer's front end,
    // chunks.rs:194-242 + analysis.rs:105-112 with every branches-th bin kept: v[i] = sum_p w[i + n p] x[base + i + n p])
    const long base = base0 + (long)fr * hop;
    if (branches == 1) {
        // (four loads in flight per lane: left as one loop the compiler keeps a single load between a wait and the LDS store)
        for (int i0 = t; i0 < n; i0 += 4 * nt) {
            v2<T> v[4];
            T w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nt;
                if (i < n) {
                    const long g = base + i;
                    v[u] = g >= 0 ? in[g] : head[n_head + g];
                    w[u] = window[i];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nt;
                if (i < n) x[i] = v2<T>{v[u].x * w[u], v[u].y * w[u]};
            }
        }
    } else {
        // (two elements x four branches per trip: eight samples and eight window values requested before the first is used - one
        //  element and one branch at a time the loop waited for every load: 1000 bins x 2 taps 0.75 ms per 2^26 samples, half of it here;
        //  the sums in the same order, branch by branch)
        for (int i0 = t; i0 < n; i0 += 2 * nt) {
            v2<T> acc[2] = {{(T)0, (T)0}, {(T)0, (T)0}};
            for (int pb0 = 0; pb0 < branches; pb0 += 4) {
                v2<T> v[2][4];
                T w[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = i0 + u * nt, pb = pb0 + q;
                        v[u][q] = v2<T>{(T)0, (T)0};
                        w[u][q] = (T)0;
                        if (i < n && pb < branches) {
                            const long g = base + i + (long)pb * n;
                            v[u][q] = g >= 0 ? in[g] : head[n_head + g];
                            w[u][q] = window[i + pb * n];
                        }
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (pb0 + q < branches) {
                            acc[u].x += v[u][q].x * w[u][q];
                            acc[u].y += v[u][q].y * w[u][q];
                        }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (i0 + u * nt < n) x[i0 + u * nt] = acc[u];
        }
    }
    __syncthreads();
    int L = n;
    for (int ps = 0; ps < plan.nrad; ++ps) {
        const int r = plan.radix[ps], q = plan.q[ps], step = n / L, nb = n / r;
        const float rq = plan.rq[ps];
        for (int b = t; b < nb; b += nt) {
            const int blk = div_small(b, rq), j = b - blk * q;
            v2<T> *p = x + blk * L + j;
            const int k1 = j * step;  // W_L^j = tw[j n / L]
            mixed_bfly<T, ODD>(p, q, r, tw, k1, n);
        }
        __syncthreads();
        L = q;
    }
    // Frequency k = d_1 + r_1 (d_2 + r_2 (..)) sits at position p = d_1 q_1 + d_2 q_2 + .. (q_i = the span of pass i).  The lanes walk
    // over the OUTPUT index, so that the global stores are contiguous and the scattered side is the LDS read (walking over the
    // positions instead - one 8-byte store per 128-byte line and lane - was the first form).
    // Short chunks keep the first form: their few output lines merge in L2, and the walk over positions has less index
    // arithmetic (per 2^24 samples, positions / outputs: 96 points 0.186 / 0.204 ms, 300 0.137 / 0.163, 1000 0.142 / 0.147,
    // 3000 0.166 / 0.158, 4000 0.174 / 0.157, 8000 0.207 / 0.177).
    v2<T> *dst = out + (size_t)fr * n;
    const int rot = center_dc ? n / 2 : 0;  // rotate_right(n / 2): bin k goes to k + rot
    if (n < 2048) {
        for (int i = t; i < n; i += nt) {
            int k = 0, mul = 1, rem = i;
            for (int ps = 0; ps < plan.nrad; ++ps) {  // span of digit ps = the pass's q
                const int d = div_small(rem, plan.rq[ps]);
                rem -= d * plan.q[ps];
                k += d * mul;
                mul *= plan.radix[ps];
            }
            int o = k + rot;
            if (o >= n) o -= n;
            dst[o] = x[i];  // (scattered 8-byte stores: as non-temporal ones they no longer combine in L2 - 1000 points 0.126 -> 0.539 ms)
        }
        return;
    }
    for (int o = t; o < n; o += nt) {
        int rem = o - rot;
        if (rem < 0) rem += n;
        int p = 0;
        for (int ps = 0; ps < plan.nrad; ++ps) {
            const int r = plan.radix[ps];
            const int hi = div_small(rem, r == 2 ? 0.5f : r == 3 ? (1.0f / 3.0f) : r == 4 ? 0.25f : r == 5 ? 0.2f
                                          : r == 7 ? (1.0f / 7.0f) : r == 11 ? (1.0f / 11.0f) : (1.0f / 13.0f));
            p += (rem - hi * r) * plan.q[ps];
            rem = hi;
        }
        dst[o] = x[p];
    }
}

// Short chunks (a few hundred points): SEVERAL chunks per workgroup, side by side in LDS, their butterflies spread over the
// workgroup's lanes together - one chunk of 96 points keeps 24 of a wave's 64 lanes busy through four passes and barriers, and
// a workgroup per chunk is mostly launch and barrier.  fpw chunks per workgroup (fpw n <= 4096 elements), the table W_n in LDS
// behind them; everything else as k_fft_mixed (one lane per butterfly, in-place passes, the digit reversal at the store).
template <class T, bool ODD>
__global__ __launch_bounds__(1024) void k_fft_mixed_multi(const v2<T> *__restrict__ head, long n_head, const v2<T> *__restrict__ in,
                                                         long base0, long hop, int n, int branches, int fpw, MixedPlan plan,
                                                         const T *__restrict__ window, const v2<T> *__restrict__ tw_g,
                                                         v2<T> *__restrict__ out, int center_dc, unsigned count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mixedm_raw[];
    v2<T> *const x = reinterpret_cast<v2<T> *>(mixedm_raw);
    v2<T> *const tw = x + fpw * n;
    const int nt = blockDim.x, t = threadIdx.x;
    const unsigned fr0 = blockIdx.x * (unsigned)fpw;
    const int nfr = count - fr0 < (unsigned)fpw ? (int)(count - fr0) : fpw;  // (the grid covers count exactly: fr0 < count)
    for (int i = t; i < n; i += nt) tw[i] = tw_g[i];
    const float rn = 1.0f / (float)n;
    for (int idx = t; idx < nfr * n; idx += nt) {
        const int f = div_small(idx, rn), i = idx - f * n;
        const long base = base0 + (long)(fr0 + f) * hop;
        v2<T> acc = {(T)0, (T)0};
        for (int pb = 0; pb < branches; ++pb) {
            const long g = base + i + (long)pb * n;
            const v2<T> v = g >= 0 ? in[g] : head[n_head + g];
            const T w = window[i + pb * n];
            acc.x += v.x * w;
            acc.y += v.y * w;
        }
        x[idx] = acc;
    }
    __syncthreads();
    int L = n;
    for (int ps = 0; ps < plan.nrad; ++ps) {
        const int r = plan.radix[ps], q = plan.q[ps], step = n / L, nb = n / r;
        const float rq = plan.rq[ps], rnb = 1.0f / (float)nb;
        for (int b = t; b < nfr * nb; b += nt) {
            const int f = div_small(b, rnb), bb = b - f * nb;
            const int blk = div_small(bb, rq), j = bb - blk * q;
            mixed_bfly<T, ODD>(x + f * n + blk * L + j, q, r, tw, j * step, n);
        }
        __syncthreads();
        L = q;
    }
    const int rot = center_dc ? n / 2 : 0;  // rotate_right(n / 2)
    for (int idx = t; idx < nfr * n; idx += nt) {
        const int f = div_small(idx, rn), i = idx - f * n;
        int k = 0, mul = 1, rem = i;
        for (int ps = 0; ps < plan.nrad; ++ps) {
            const int d = div_small(rem, plan.rq[ps]);
            rem -= d * plan.q[ps];
            k += d * mul;
            mul *= plan.radix[ps];
        }
        int o = k + rot;
        if (o >= n) o -= n;
        out[(size_t)(fr0 + f) * n + o] = x[idx];
    }
}

// n = 2^a 3^b 5^c, not a power of two, at most nmax points: the radices, largest first (5s, 4s, 3s, at most one 2)
static bool mixed_plan_impl(size_t n, size_t nmax, MixedPlan *pl, bool pow2_too) {
    if (n < 2 || n > nmax || (!pow2_too && (n < 6 || is_pow2_n(n)))) return false;
    size_t m = n;
    int a = 0, b = 0, c = 0, k = 0;
    while (m % 2 == 0) m /= 2, ++a;
    while (m % 3 == 0) m /= 3, ++b;
    while (m % 5 == 0) m /= 5, ++c;
    for (size_t pr : {13, 11, 7})
        while (m % pr == 0) {
            m /= pr;
            if (k >= 12) return false;
            pl->radix[k++] = (unsigned char)pr;
        }
    if (m != 1) return false;
    if (k + c + a / 2 + b + (a & 1) > 12) return false;
    for (int i = 0; i < c; ++i) pl->radix[k++] = 5;
    for (int i = 0; i < a / 2; ++i) pl->radix[k++] = 4;
    for (int i = 0; i < b; ++i) pl->radix[k++] = 3;
    if (a & 1) pl->radix[k++] = 2;
    if (k > 12) return false;
    pl->nrad = k;
    size_t L = n;
    for (int i = 0; i < k; ++i) {
        L /= pl->radix[i];
        pl->q[i] = (unsigned short)L;
        pl->rq[i] = 1.0f / (float)L;
    }
    return true;
}
// one LDS image of at most 64 KiB: 8192 points in f32, 4096 in f64
static size_t mixed_max(int dtype) { return dtype == RR_F32 ? 8192 : 4096; }
static bool mixed_plan(size_t n, size_t nmax, MixedPlan *pl) { return mixed_plan_impl(n, nmax, pl, false); }
static bool mixed_plan_any(size_t n, size_t nmax, MixedPlan *pl) { return mixed_plan_impl(n, nmax, pl, true); }
// the radices of k_fft_mixed's passes for a length it serves (for rr_fourier_route)
int fft_mixed_radices(int dtype, size_t n, unsigned char *radices, int cap) {
    MixedPlan pl;
    if (!mixed_plan(n, mixed_max(dtype), &pl)) return 0;
    for (int i = 0; i < pl.nrad && i < cap; ++i) radices[i] = pl.radix[i];
    return pl.nrad;
}
bool fft_mixed_supported(int dtype, size_t n) {
    MixedPlan pl;
    return mixed_plan(n, mixed_max(dtype), &pl);
}
// ---------------------------------------------------------------------------
// Lengths 2^a 3^b 5^c beyond one LDS image (20000, 48000, 100000 ..; up to 512 x 512): the four-step split N = N1 N2 in two
// passes over HBM like k_fft_tile, with mixed-radix sub-transforms (k_fft_mixed's butterflies on a tile of Np x C elements).
//   pass A  bundles of C neighbouring columns n2: window, transforms over n1, times W_N^(n2 k1), Y[k1][n2] in the same shape
//   pass B  bundles of C neighbouring rows k1: transforms over n2, X[k1 + N1 k2] (C neighbours per k2)
// N1 and N2 need not be multiples of C: the last bundle is partly empty.  W_N^(k1 n2) with n2 = C bx + c is the product of
// T1[k1 bx] = W_N^(C k1 bx) and T2[k1 c] = W_N^(k1 c): two exact table indices, no division by a run-time length.
// Frames come from [ head | in ] at any hop.
// ---------------------------------------------------------------------------
template <class T, int MODE, bool ODD>
__global__ __launch_bounds__(1024) void k_fft_tilem(const v2<T> *__restrict__ head, long n_head, const v2<T> *__restrict__ in,
                                                    long hop, v2<T> *__restrict__ out, int Np, int No, MixedPlan plan,
                                                    const T *__restrict__ window, const v2<T> *__restrict__ twNp,
                                                    const v2<T> *__restrict__ T1, const v2<T> *__restrict__ T2, int rot) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tilem_raw[];
    constexpr int C = 128 / (int)sizeof(v2<T>);
    v2<T> *const tile = reinterpret_cast<v2<T> *>(tilem_raw);
    const int SN = MODE == 0 ? C : 1, SC = MODE == 0 ? 1 : Np + 1;
    v2<T> *const tw = tile + (MODE == 0 ? Np * C : C * (Np + 1));
    const int nt = blockDim.x, t = threadIdx.x;
    const size_t N = (size_t)Np * (size_t)No;
    const int g0 = blockIdx.x * C;
    const int cv = No - g0 < C ? No - g0 : C;  // columns (rows) of this bundle that exist
    for (int i = t; i < Np; i += nt) tw[i] = twNp[i];
    if (MODE == 0) {
        const long base = (long)blockIdx.y * hop - n_head;  // frames from [ head | in ]
        // (four loads in flight per lane, as k_fft_tile)
        for (int idx0 = t; idx0 < Np * C; idx0 += 4 * nt) {
            v2<T> v[4];
            T w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * nt;
                v[u] = v2<T>{(T)0, (T)0};
                w[u] = (T)0;
                if (idx < Np * C && idx % C < cv) {
                    const size_t e = (size_t)(idx / C) * No + g0 + idx % C;
                    const long g = base + (long)e;
                    v[u] = g >= 0 ? in[g] : head[n_head + g];
                    w[u] = window[e];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * nt;
                if (idx < Np * C) tile[idx] = v2<T>{v[u].x * w[u], v[u].y * w[u]};  // (n C + c = idx)
            }
        }
    } else {
        const float rNp = 1.0f / (float)Np;
        const size_t chunk = (size_t)blockIdx.y * N;
        for (int idx0 = t; idx0 < Np * C; idx0 += 4 * nt) {
            v2<T> v[4];
            int at[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * nt;
                v[u] = v2<T>{(T)0, (T)0};
                at[u] = -1;
                if (idx < Np * C) {
                    const int c = div_small(idx, rNp), n = idx - c * Np;
                    at[u] = c * (Np + 1) + n;
                    if (c < cv) v[u] = in[chunk + (size_t)(g0 + c) * Np + n];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (at[u] >= 0) tile[at[u]] = v[u];
        }
    }
    __syncthreads();
    int L = Np, step = 1;
    for (int ps = 0; ps < plan.nrad; ++ps) {
        const int r = plan.radix[ps], q = plan.q[ps], nb = step * q;  // nb = Np / r butterflies per column
        const float rq = plan.rq[ps], rnb = 1.0f / (float)nb;
        for (int b = t; b < nb * C; b += nt) {
            int jj, c;
            if (MODE == 0) {
                c = b % C;
                jj = b / C;
            } else {
                c = div_small(b, rnb);
                jj = b - c * nb;
            }
            const int blk = div_small(jj, rq), j = jj - blk * q;
            mixed_bfly<T, ODD>(tile + (blk * L + j) * SN + c * SC, q * SN, r, tw, j * step, Np);
        }
        __syncthreads();
        L = q;
        step *= r;
    }
    const size_t ochunk = (size_t)blockIdx.y * N;
    for (int idx = t; idx < Np * C; idx += nt) {
        const int p = idx / C, c = idx % C;
        if (c >= cv) continue;
        int k = 0, mul = 1, rem = p;
        for (int ps = 0; ps < plan.nrad; ++ps) {
            const int d = div_small(rem, plan.rq[ps]);
            rem -= d * plan.q[ps];
            k += d * mul;
            mul *= plan.radix[ps];
        }
        v2<T> v = tile[p * SN + c * SC];
        if (MODE == 0) {
            v = cmul<T>(v, cmul<T>(T1[(size_t)k * blockIdx.x], T2[k * c]));
            out[ochunk + (size_t)k * No + g0 + c] = v;
        } else {
            size_t o = (size_t)k * No + g0 + c + (size_t)rot;  // X[k1 + N1 k2], rotated right by rot = n / 2 ELEMENTS for center_dc
            if (o >= N) o -= N;
            st_result<T>(out + (ochunk + o), v);
        }
    }
}

static bool mixed_plan_any(size_t n, size_t nmax, MixedPlan *pl);  // (powers of two too)

// the split of a length 2^a 3^b 5^c into two factors of C .. 512 points each, as balanced as possible
bool fft_tilem_split(int dtype, size_t n, size_t *N1, size_t *N2) {
    const size_t C = dtype == RR_F32 ? 16 : 8;
    if (n > 512 * 512 || n < C * C) return false;
    {
        size_t m = n;
        for (size_t pr : {2, 3, 5, 7, 11, 13})
            while (m % pr == 0) m /= pr;
        if (m != 1) return false;
    }
    size_t best = 0;
    for (size_t d = C; d <= 512; ++d) {
        if (n % d) continue;
        const size_t e = n / d;
        if (e < C || e > 512) continue;
        const size_t lo = d < e ? d : e;
        // the more balanced pair; among equals the one whose row length N2 is a whole number of 128-byte lines
        if (lo > best || (lo == best && e % C == 0 && *N2 % C != 0)) {
            best = lo;
            *N1 = d;
            *N2 = e;
        }
    }
    return best != 0;
}

template <class T>
static int launch_fft_tilem_t(hipStream_t s, int pass, const void *head, size_t n_head, const void *in, size_t hop, void *out,
                              size_t N1, size_t N2, size_t count, const void *window, const void *twNp, const void *T1,
                              const void *T2, size_t rot) {
    constexpr size_t C = 128 / sizeof(v2<T>);
    const size_t Np = pass == 0 ? N1 : N2, No = pass == 0 ? N2 : N1;
    MixedPlan pl;
    if (!mixed_plan_any(Np, 512, &pl)) RR_FAIL(RR_ERR_BAD_ARG, "tile transform: %zu points", Np);
    const size_t lds = (pass == 0 ? Np * C : C * (Np + 1)) * sizeof(v2<T>) + Np * sizeof(v2<T>);
    size_t nt = (Np * C / 4 + 63) / 64 * 64;
    if (nt > 1024) nt = 1024;
    if (nt < 256) nt = 256;
    const dim3 grid((unsigned)((No + C - 1) / C), (unsigned)count);
    const bool odd = pl.radix[0] > 5;  // (the primes beyond 5 come first in the plan)
#define RR_TILEM(MM, OO)                                                                                                        \
    do {                                                                                                                        \
        auto fn = k_fft_tilem<T, MM, OO>;                                                                                       \
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));                                                           \
        if (MM == 0)                                                                                                            \
            hipLaunchKernelGGL(fn, grid, dim3((unsigned)nt), lds, s, (const v2<T> *)head, (long)n_head, (const v2<T> *)in,     \
                               (long)hop, (v2<T> *)out, (int)Np, (int)No, pl, (const T *)window, (const v2<T> *)twNp,          \
                               (const v2<T> *)T1, (const v2<T> *)T2, 0);                                                        \
        else                                                                                                                    \
            hipLaunchKernelGGL(fn, grid, dim3((unsigned)nt), lds, s, (const v2<T> *)nullptr, 0L, (const v2<T> *)in, 0L,        \
                               (v2<T> *)out, (int)Np, (int)No, pl, (const T *)nullptr, (const v2<T> *)twNp,                     \
                               (const v2<T> *)nullptr, (const v2<T> *)nullptr, (int)rot);                                       \
    } while (0)
    if (pass == 0) {
        if (odd) RR_TILEM(0, true);
        else RR_TILEM(0, false);
    } else {
        if (odd) RR_TILEM(1, true);
        else RR_TILEM(1, false);
    }
#undef RR_TILEM
    RR_HIP(hipGetLastError());
    return RR_OK;
}
int launch_fft_tilem(int dtype, hipStream_t s, int pass, const void *head, size_t n_head, const void *in, size_t hop, void *out,
                     size_t N1, size_t N2, size_t count, const void *window, const void *twNp, const void *T1, const void *T2,
                     size_t rot) {
    if (count == 0) return RR_OK;
    if (count > 65535) RR_FAIL(RR_ERR_BAD_ARG, "tile transform: too many chunks in one launch");
    if (dtype == RR_F32)
        return launch_fft_tilem_t<float>(s, pass, head, n_head, in, hop, out, N1, N2, count, window, twNp, T1, T2, rot);
    return launch_fft_tilem_t<double>(s, pass, head, n_head, in, hop, out, N1, N2, count, window, twNp, T1, T2, rot);
}

// Where k_fft_mixed is ahead of the Bluestein kernels (ms per 2^24 samples, one session: mixed 96 / 300 / 500 / 1000 / 1200 / 1536 /
// 2000 / 3000 / 4000 points 0.185 / 0.141 / 0.128 / 0.148 / 0.149 / 0.185 / 0.166 / 0.203 / 0.231; k_bluestein1024 41 / n, k_bluestein4096
// 222 / n - their cost per chunk does not depend on n -, the five launches beyond 2048 points 0.80 / 0.62 at 3000 / 4000).  Below 32
// points the direct kernel stays.  Complex<f64> has no one-kernel Bluestein: mixed wherever it applies.
bool fft_mixed_preferred(int dtype, size_t n) {
    if (n < 32) return false;
    if (dtype != RR_F32) return true;
    // lengths with a factor 7, 11 or 13 (the R x R butterflies): 1001 points 0.301 ms against k_bluestein4096's 0.219; beyond 2048
    // points, where Bluestein takes five launches, 4004 points 0.225 against 0.635
    if (n % 7 == 0 || n % 11 == 0 || n % 13 == 0) return n > 2048;
    return n < 320 || (n > 512 && n <= 1280) || n > 2048;
}
int launch_fft_mixed(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n,
                     const void *window, const void *tw, void *out, bool center_dc, size_t count) {
    return launch_fft_mixed_fold(dtype, s, head, n_head, in, -(long)n_head, hop, n, 1, window, tw, out, center_dc, count);
}
int launch_fft_mixed_fold(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, long base0, size_t hop,
                          size_t n, size_t branches, const void *window, const void *tw, void *out, bool center_dc, size_t count) {
    if (count == 0) return RR_OK;
    if (branches < 1 || base0 < -(long)n_head) RR_FAIL(RR_ERR_BAD_ARG, "mixed-radix transform: frame 0 starts in front of the history");
    MixedPlan pl;
    if (!mixed_plan_any(n, mixed_max(dtype), &pl)) RR_FAIL(RR_ERR_BAD_ARG, "mixed-radix transform: %zu points", n);
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many chunks in one call");
    // a lane per butterfly of the widest pass, whole waves
    // a lane per butterfly of the radix-4 passes, at most 512 lanes (1024 beyond 4096 points): measured per 2^24 samples with
    // at most 256 / 512 / 1024 lanes - 3000 points 0.205 / 0.166 / 0.195 ms, 4000 0.237 / 0.174 / 0.186, 4800 0.268 / 0.197 / 0.184,
    // 8000 0.347 / 0.248 / 0.207 (RR_FOURIER_MIXED_NT overrides the limit)
    static const unsigned nt_env = [] { const char *e = std::getenv("RR_FOURIER_MIXED_NT"); return e ? (unsigned)std::atoi(e) : 0u; }();
    const unsigned nt_max = nt_env ? nt_env : (n > 4096 ? 1024u : 512u);
    unsigned nt = (unsigned)((n / 4 + 63) / 64 * 64);
    if (nt > nt_max) nt = nt_max;
    if (nt < 64) nt = 64;
    // short chunks: several per workgroup (k_fft_mixed_multi) - as many as make ~2048 elements, at most 16 (RR_FOURIER_MIXED_FPW overrides; 1 = off)
    {
        static const int fpw_env = [] { const char *e = std::getenv("RR_FOURIER_MIXED_FPW"); return e ? std::atoi(e) : 0; }();
        // (per 2^24 samples, several chunks per workgroup / one: 96 points 0.126 / 0.178 ms, 300 0.125 / 0.134, 500 0.123 / 0.112,
        // 720 0.157 / 0.135, 1000 0.140 / 0.128; the 100-bin channelizer 0.745 / 0.926 ms per 2^26: up to 384 points)
        size_t fpw = fpw_env > 0 ? (size_t)fpw_env : (n <= 384 ? 2048 / n : 1);
        if (fpw > 16) fpw = 16;
        if (fpw * n > 4096) fpw = 4096 / n;
        if (fpw > count) fpw = count;
        if (fpw >= 2) {
            const size_t esz2 = dtype == RR_F32 ? sizeof(float2) : sizeof(double2);
            const size_t lds2 = (fpw + 1) * n * esz2;
            unsigned nt2 = (unsigned)((fpw * n / 4 + 63) / 64 * 64);
            if (nt2 > 512) nt2 = 512;
            if (nt2 < 64) nt2 = 64;
            const unsigned grid2 = (unsigned)((count + fpw - 1) / fpw);
            const bool odd2 = pl.radix[0] > 5;
#define RR_MIXEDM(TT, VV, OO)                                                                                                  \
    do {                                                                                                                       \
        auto fn = k_fft_mixed_multi<TT, OO>;                                                                                   \
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds2));                                                         \
        hipLaunchKernelGGL(fn, dim3(grid2), dim3(nt2), lds2, s, (const VV *)head, (long)n_head, (const VV *)in, base0,         \
                           (long)hop, (int)n, (int)branches, (int)fpw, pl, (const TT *)window, (const VV *)tw, (VV *)out,      \
                           (int)center_dc, (unsigned)count);                                                                   \
    } while (0)
            if (dtype == RR_F32) {
                if (odd2) RR_MIXEDM(float, float2, true);
                else RR_MIXEDM(float, float2, false);
            } else {
                if (odd2) RR_MIXEDM(double, double2, true);
                else RR_MIXEDM(double, double2, false);
            }
#undef RR_MIXEDM
            RR_HIP(hipGetLastError());
            return RR_OK;
        }
    }
    // twiddles in LDS while image + table stay within 32 KiB per workgroup (at least four workgroups per CU)
    static const int twlds_env = [] { const char *e = std::getenv("RR_FOURIER_MIXED_TWLDS"); return e ? std::atoi(e) : 1; }();
    const size_t esz = dtype == RR_F32 ? sizeof(float2) : sizeof(double2);
    const size_t twcap = twlds_env > 1 ? (size_t)twlds_env : 32768;
    const bool twlds = twlds_env != 0 && 2 * n * esz <= twcap;
    const size_t lds = (twlds ? 2 : 1) * n * esz;
    const bool odd = pl.radix[0] > 5;  // (the primes beyond 5 come first in the plan)
#define RR_MIXED(TT, VV, TL)                                                                                              \
    do {                                                                                                                  \
        if (odd) RR_MIXED2(TT, VV, TL, true);                                                                             \
        else RR_MIXED2(TT, VV, TL, false);                                                                                \
    } while (0)
#define RR_MIXED2(TT, VV, TL, OO)                                                                                         \
    do {                                                                                                                  \
        auto fn = k_fft_mixed<TT, TL, OO>;                                                                                \
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));                                                     \
        hipLaunchKernelGGL(fn, dim3((unsigned)count), dim3(nt), lds, s, (const VV *)head, (long)n_head, (const VV *)in,  \
                           base0, (long)hop, (int)n, (int)branches, pl, (const TT *)window, (const VV *)tw, (VV *)out,    \
                           (int)center_dc, (unsigned)count);                                                              \
    } while (0)
    if (dtype == RR_F32) {
        if (twlds) RR_MIXED(float, float2, true);
        else RR_MIXED(float, float2, false);
    } else {
        if (twlds) RR_MIXED(double, double2, true);
        else RR_MIXED(double, double2, false);
    }
#undef RR_MIXED
#undef RR_MIXED2
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_fourier_overlapped(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, void *out,
                              size_t n, size_t hop, size_t count, const void *window, const void *twiddle, bool center_dc) {
    if (count == 0) return RR_OK;
    RR_TRY(fourier_supported(dtype, n));
    static const bool generic = [] { const char *e = std::getenv("RR_FOURIER_GENERIC"); return e && std::atoi(e) != 0; }();
    // Complex<f64>, 4096 points: the register kernel of rr_f64.hip (RR_FOURIER_GENERIC=1 keeps the Stockham passes of k_fft_pow2)
    if (dtype == RR_F64 && n == 4096 && !generic)
        return launch_fft4096_f64(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
    if (dtype == RR_F32 && n == 4096 && stft4096_supported(hop) && count >= 64 && !generic)
        return launch_stft4096(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
    if (dtype == RR_F32 && n == 4096)
        return launch_fft4096(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
    if (dtype == RR_F32 && (n == 64 || n == 128) && hop == n && n_head == 0 && !generic)
        return launch_fft_small(s, in, out, n, count, window, twiddle, center_dc);
    if (dtype == RR_F32 && n == 8192 && !generic) {
        // 512 lanes with 16 values each (rr_fft_big.hpp) instead of k_fft8192's 256 lanes with 32: 0.194 against 0.230 ms per 2^26
        // samples in one session (RR_FOURIER_8K=regs keeps k_fft8192 - A/B runs, tests)
        const char *e8 = std::getenv("RR_FOURIER_8K");
        if (e8 && !std::strcmp(e8, "regs")) return launch_fft8192(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
        return launch_fft8192_big(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
    }
    if (dtype == RR_F32 && n == 16384)  // (pow2_limit: only without RR_FOURIER_GENERIC / RR_FOURIER_16K=0)
        return launch_fft16384(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
    if (dtype == RR_F32 && n == 512 && !generic)
        return launch_fft512(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
    if (dtype == RR_F32 && n == 2048 && !generic)
        return launch_fft2048(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
    if (dtype == RR_F32 && n == 1024 && !generic)
        return launch_fft1024(s, head, n_head, in, out, count, window, twiddle, center_dc, hop);
    // 256-point chunks side by side: the channelizer's one-branch case (fold of one chunk = window * x, every bin kept)
    if (dtype == RR_F32 && n == 256 && hop == 256 && !center_dc && !generic)
        return launch_channelizer256(s, head, n_head, in, -(long)n_head, 1, count, window, twiddle, out);
    if (dtype == RR_F32)
        return launch_fourier_t<float>(s, head, n_head, in, out, n, hop, count, window, twiddle, center_dc, dtype);
    return launch_fourier_t<double>(s, head, n_head, in, out, n, hop, count, window, twiddle, center_dc, dtype);
}

int launch_fourier(int dtype, hipStream_t s, const void *in, void *out, size_t n, size_t count, const void *window,
                   const void *twiddle, bool center_dc) {
    return launch_fourier_overlapped(dtype, s, nullptr, 0, in, out, n, n, count, window, twiddle, center_dc);
}

// ---------------------------------------------------------------------------
// Channelizer, general form (any number of bins, any hop that divides P M): the fold alone,
//   y[f][r] = sum_{p < P} w[r + M p] x[base0 + hop f + r + M p],  r < M,
// written frame by frame to a workspace; the M-point transforms follow through rr_fourier (any M).
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_chan_fold(const v2<T> *__restrict__ hist, long hist_len, const v2<T> *__restrict__ in,
                                                   long base0, long hop, int M, int P, const T *__restrict__ window,
                                                   v2<T> *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const long base = base0 + (long)blockIdx.y * hop + r;
    v2<T> acc;
    acc.x = 0;
    acc.y = 0;
    for (int p = 0; p < P; ++p) {
        const long i = base + (long)M * p;
        const v2<T> x = (i >= 0) ? in[i] : hist[hist_len + i];
        const T w = window[r + M * p];
        acc.x += x.x * w;
        acc.y += x.y * w;
    }
    out[(size_t)blockIdx.y * M + r] = acc;
}
int launch_chan_fold(int dtype, hipStream_t s, const void *hist, size_t hist_len, const void *in, long base0, size_t hop,
                     size_t M, size_t P, size_t frames, const void *window, void *out) {
    if (frames == 0) return RR_OK;
    if (frames > 65535) RR_FAIL(RR_ERR_BAD_ARG, "Channelizer: too many frames in one pass");
    const dim3 grid((unsigned)((M + 255) / 256), (unsigned)frames);
    if (dtype == RR_F32)
        hipLaunchKernelGGL(k_chan_fold<float>, grid, dim3(256), 0, s, (const float2 *)hist, (long)hist_len, (const float2 *)in,
                           base0, (long)hop, (int)M, (int)P, (const float *)window, (float2 *)out);
    else
        hipLaunchKernelGGL(k_chan_fold<double>, grid, dim3(256), 0, s, (const double2 *)hist, (long)hist_len, (const double2 *)in,
                           base0, (long)hop, (int)M, (int)P, (const double *)window, (double2 *)out);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Polyphase FFT channelizer: fold P branches of the windowed P*M-sample span into M
// bins, then an M-point Stockham radix-2 FFT in LDS; one workgroup per hop.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_channelizer(const v2<T> *__restrict__ hist, long hist_len,
                                                     const v2<T> *__restrict__ in, long base0, int M, int P,
                                                     const T *__restrict__ window, const v2<T> *__restrict__ tw,
                                                     v2<T> *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2<T> *a = reinterpret_cast<v2<T> *>(smem);
    v2<T> *b = a + M;
    const long base = base0 + (long)blockIdx.x * M;
    for (int r = threadIdx.x; r < M; r += blockDim.x) {
        v2<T> acc;
        acc.x = 0;
        acc.y = 0;
        for (int p = 0; p < P; ++p) {
            const long i = base + r + (long)M * p;
            const v2<T> x = (i >= 0) ? in[i] : hist[hist_len + i];
            const T w = window[r + M * p];
            acc.x += x.x * w;
            acc.y += x.y * w;
        }
        a[r] = acc;
    }
    __syncthreads();
    const int half = M >> 1;
    for (int ns = 1; ns < M; ns <<= 1) {
        const int tstride = half / ns;
        for (int j = threadIdx.x; j < half; j += blockDim.x) {
            const int k = j & (ns - 1);
            const v2<T> u = a[j];
            const v2<T> v = cmul<T>(a[j + half], tw[k * tstride]);
            const int j0 = ((j - k) << 1) + k;
            v2<T> s, d;
            s.x = u.x + v.x;
            s.y = u.y + v.y;
            d.x = u.x - v.x;
            d.y = u.y - v.y;
            b[j0] = s;
            b[j0 + ns] = d;
        }
        __syncthreads();
        v2<T> *t = a;
        a = b;
        b = t;
    }
    v2<T> *dst = out + (long)blockIdx.x * M;
    for (int i = threadIdx.x; i < M; i += blockDim.x) dst[i] = a[i];
}

// one fused fold + FFT kernel: hop = M for powers of two up to one LDS tile; hop < M through k_channelizer256
bool channelizer_fused_supported(int dtype, size_t M, size_t P, size_t hop) {
    const bool pow2 = (M & (M - 1)) == 0;
    if (hop == M) return pow2 && M <= (dtype == RR_F32 ? 8192u : 4096u);
    return channelizer256_supported(dtype, M, P, hop) && !std::getenv("RR_CHANNELIZER_GENERIC");
}

int launch_channelizer(int dtype, hipStream_t s, const void *hist, size_t hist_len, const void *in, long base0,
                       size_t M, size_t P, size_t nframes, const void *window, const void *tw, void *out, size_t hop) {
    if (nframes == 0) return RR_OK;
    if (nframes > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "channelizer: too many frames in one call");
    if (hop == 0) hop = M;
    if (channelizer256_supported(dtype, M, P, hop) && !std::getenv("RR_CHANNELIZER_GENERIC"))
        return launch_channelizer256(s, hist, hist_len, in, base0, P, nframes, window, tw, out, hop);
    if (hop != M) RR_FAIL(RR_ERR_BAD_ARG, "channelizer: hop %zu of %zu bins has no fused kernel", hop, M);
    if (dtype == RR_F32 && (M == 512 || M == 1024 || M == 2048 || M == 4096) && !std::getenv("RR_CHANNELIZER_GENERIC")) {
        // frame 0 starts at base0 (<= 0) relative to in[0]: the last -base0 samples of the history are its head
        const size_t nh = base0 < 0 ? (size_t)(-base0) : 0;
        if (base0 <= 0 && nh <= hist_len) {
            const char *hd = static_cast<const char *>(hist) + (hist_len - nh) * sizeof(float2);
            if (M == 512) return launch_chan512(s, hd, nh, in, out, nframes, window, tw, hop, P);
            if (M == 1024) return launch_chan1024(s, hd, nh, in, out, nframes, window, tw, hop, P);
            if (M == 2048) return launch_chan2048(s, hd, nh, in, out, nframes, window, tw, hop, P);
            return launch_chan4096(s, hd, nh, in, out, nframes, window, tw, hop, P);
        }
    }
    const size_t lds = 2 * M * elem_size(dtype);
    int threads = (int)(M / 2);
    threads = threads > 256 ? 256 : (threads < 64 ? 64 : threads);
    if (dtype == RR_F32) {
        auto fn = k_channelizer<float>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)nframes), dim3(threads), lds, s, (const float2 *)hist, (long)hist_len,
                           (const float2 *)in, base0, (int)M, (int)P, (const float *)window, (const float2 *)tw,
                           (float2 *)out);
    } else {
        auto fn = k_channelizer<double>;
        RR_TRY(set_dyn_lds(reinterpret_cast<const void *>(fn), lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)nframes), dim3(threads), lds, s, (const double2 *)hist, (long)hist_len,
                           (const double2 *)in, base0, (int)M, (int)P, (const double *)window, (const double2 *)tw,
                           (double2 *)out);
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Synthetic IQ source: counter based, any sample independently.
// ---------------------------------------------------------------------------
struct ToneTable {
    double re[32], im[32];
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__global__ __launch_bounds__(256) void k_synth(unsigned long long seed, unsigned long long t0, size_t n,
                                               float2 *__restrict__ out, ToneTable tones) {
    __shared__ double tr[32], ti[32];
    if (threadIdx.x < 32) {
        tr[threadIdx.x] = tones.re[threadIdx.x];
        ti[threadIdx.x] = tones.im[threadIdx.x];
    }
    __syncthreads();
    const double sc = 1.0 / 4294967296.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long t = t0 + i;
        const unsigned long long u = mix64(seed * 0x9E3779B97F4A7C15ull + t);
        const int a = (int)(unsigned)(u >> 32);
        const int b = (int)(unsigned)(u & 0xFFFFFFFFull);
        float2 v;
        v.x = (float)((double)a * sc + tr[t & 31]);
        v.y = (float)((double)b * sc + ti[t & 31]);
        out[i] = v;
    }
}

int launch_synth(hipStream_t s, uint64_t seed, uint64_t t0, size_t n, void *out) {
    if (n == 0) return RR_OK;
    ToneTable tones;
    for (int k = 0; k < 32; ++k) {
        const double a1 = 2.0 * M_PI * (double)(k % 16) / 16.0;
        const double a2 = -2.0 * M_PI * (double)((3 * k) % 32) / 32.0;
        tones.re[k] = 0.25 * cos(a1) + 0.25 * cos(a2);
        tones.im[k] = 0.25 * sin(a1) + 0.25 * sin(a2);
    }
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned long long)seed,
                       (unsigned long long)t0, n, (float2 *)out, tones);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
