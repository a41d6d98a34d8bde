// rr_fused.hip — the fast path of the chain FreqShifter -> Filter -> Downsampler
// -> Fourier for Complex<f32> on gfx950 (CDNA4).
//
// Kernel 1  k_mix_fir_decim<D,R,T>
//   v[m] = sum_{i<Lc} c[i] * xs[e_m - i],   xs[t] = x[t] * p[(idx0 + t) mod denom]
//   c = reverse(ir) (*) g : the Downsampler's real impulse response (resampling.rs:
//   82-99) convolved with the Filter's equivalent causal taps (filters.rs:184-259),
//   evaluated only at the emitted positions e_m = e0 + D*m (resampling.rs:110-112).
//   HBM traffic: 8 B read per input sample + 8/D B written: the algorithmic
//   minimum of the mix+filter+decimate stage.
//   Structure per workgroup (T lanes, R outputs per lane):
//     * the input span (+ halo) is loaded once with 16-B coalesced loads, mixed
//       with the NCO phasor on the way and written to LDS rows of R*D samples,
//       padded so that the per-lane ds_read_b128 of the compute phase are
//       bank-conflict free;
//     * each lane keeps R accumulators and a rotating window of R blocks of D
//       samples in registers; per tap group it reads ONE new block and D taps
//       (LDS broadcast) and issues R*D packed FMAs (v_pk_fma_f32, re/im in one
//       64-bit register pair) — measured on MI355X: packed FMAs with VGPR
//       operands sustain ~108 TFLOP/s at 2 waves/SIMD, scalar-operand FMAs ~55;
//     * results go through LDS so that the global stores are coalesced.
//   blockIdx -> tile mapping keeps neighbouring tiles (which share the halo)
//   on one XCD, i.e. one L2.
//
// Kernel 2  k_fft4096: window * v -> 4096-point forward DFT, radix-16 x 3 in
//   registers (Stockham autosort through a padded 32 KiB LDS image), one
//   workgroup per spectrum (analysis.rs:105-115); center_dc is an index
//   rotation on the store.
#include "rr_blocks.hpp"

#include <cmath>

namespace rr {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// geometry shared by host and device
// ---------------------------------------------------------------------------
template <int D, int R> struct FirGeom {
    static constexpr int RD = R * D;                       // samples per LDS row
    static constexpr int ROW_BYTES = RD * 8;
    // row stride: odd multiple of 16 B -> the 16 lanes of a ds_read_b128 group
    // hit 16 different 16-B slots of the 256-B bank row
    static constexpr int STRIDE = ROW_BYTES + (((ROW_BYTES / 16) & 1) ? 32 : 16);
    static constexpr int OUT_STRIDE = R * 8 + 16;          // staged outputs per lane
};

template <int D, int R, int T>
__global__ __launch_bounds__(T) void k_mix_fir_decim(const float2 *__restrict__ xh, int hx,
                                                     const float2 *__restrict__ in, long n_in, int in_aligned16,
                                                     const float2 *__restrict__ nco, unsigned denom, unsigned idx0,
                                                     const float *__restrict__ taps, int Gp,
                                                     float2 *__restrict__ out, long n_out, long e0,
                                                     unsigned ntiles) {
    using G = FirGeom<D, R>;
    constexpr int RD = G::RD, STRIDE = G::STRIDE;
    constexpr int OUTS = T * R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rows = T + Gp / R;
    float *tap_lds = reinterpret_cast<float *>(smem + (size_t)rows * STRIDE);

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round robin), give
    // them consecutive tiles so the halo re-read hits that XCD's L2
    unsigned tile;
    {
        const unsigned b = blockIdx.x, q = ntiles >> 3, rmd = ntiles & 7, xcd = b & 7;
        tile = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (b >> 3);
    }
    const long mt = (long)tile * OUTS;
    const long tile_lo = e0 + (long)D * mt - (long)D * Gp + 1;  // oldest sample of the tile
    const int NS = rows * RD;

    // ---- taps -> LDS -------------------------------------------------------
    for (int i = threadIdx.x; i < Gp * D; i += T) tap_lds[i] = taps[i];

    // ---- load + mix --------------------------------------------------------
    {
        const long lo_even = tile_lo - (tile_lo & 1);
        const int npairs = (int)((tile_lo + NS - lo_even + 1) >> 1);
        const unsigned step = (unsigned)((2 * T) % denom);
        long ph = ((long)idx0 + lo_even + 2 * (long)threadIdx.x) % (long)denom;
        if (ph < 0) ph += denom;
        unsigned rr_ = (unsigned)ph;
        auto put = [&](int s, f2 v) {
            if (s >= 0 && s < NS) *reinterpret_cast<f2 *>(smem + (s / RD) * STRIDE + (s % RD) * 8) = v;
        };
        // interior tiles (all but the first and last few): every pair is a 16-B
        // aligned load inside `in`.  U pairs per lane are requested before the
        // first one is used, so the HBM latency is paid once per batch.
        const bool interior = in_aligned16 && lo_even >= 0 && lo_even + 2L * npairs <= n_in;  // workgroup-uniform
        if (interior) {
            constexpr int U = 6;
            const f4 *src = reinterpret_cast<const f4 *>(in + lo_even);
            for (int pb = 0; pb < npairs; pb += U * T) {
                f4 x[U];
                float2 p0[U], p1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    int pi = pb + u * T + (int)threadIdx.x;
                    pi = pi < npairs ? pi : npairs - 1;  // clamp instead of branching around the load
                    x[u] = src[pi];
                    p0[u] = nco[rr_];
                    p1[u] = nco[(rr_ + 1 == denom) ? 0 : rr_ + 1];
                    rr_ += step;
                    if (rr_ >= denom) rr_ -= denom;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pi = pb + u * T + (int)threadIdx.x;
                    if (pi < npairs) {
                        const int s0 = (int)(lo_even - tile_lo) + 2 * pi;
                        put(s0, (f2){x[u].x * p0[u].x - x[u].y * p0[u].y, x[u].x * p0[u].y + x[u].y * p0[u].x});
                        put(s0 + 1, (f2){x[u].z * p1[u].x - x[u].w * p1[u].y, x[u].z * p1[u].y + x[u].w * p1[u].x});
                    }
                }
            }
        } else {
            // edge tiles: history (already mixed), end of input, unaligned input
            for (int pi = threadIdx.x; pi < npairs; pi += T) {
                const long pe = lo_even + 2 * (long)pi;
                const unsigned r1 = (rr_ + 1 == denom) ? 0 : rr_ + 1;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const long pos = pe + k;
                    f2 v = {0.f, 0.f};
                    if (pos < 0) {
                        if (pos >= -(long)hx) {
                            const float2 h = xh[hx + pos];
                            v = (f2){h.x, h.y};
                        }
                    } else if (pos < n_in) {
                        const float2 x = in[pos];
                        const float2 pp = nco[k ? r1 : rr_];
                        v = (f2){x.x * pp.x - x.y * pp.y, x.x * pp.y + x.y * pp.x};
                    }
                    put((int)(pos - tile_lo), v);
                }
                rr_ += step;
                if (rr_ >= denom) rr_ -= denom;
            }
        }
    }
    __syncthreads();

    // ---- FIR: rotating register window, packed FMAs ------------------------
    f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = (f2){0.f, 0.f};
    f2 W[R][D];
    const char *lane = smem + (size_t)threadIdx.x * STRIDE;
    // block b of this lane lives at row b / R, column block b % R
    auto load_block = [&](const char *row0, int b_static, f2(&dst)[D]) {
        const char *p = row0 + (b_static / R) * STRIDE + (b_static % R) * (D * 8);
        if constexpr (D % 2 == 0) {
#pragma unroll
            for (int q = 0; q < D / 2; ++q) {
                const f4 v = *reinterpret_cast<const f4 *>(p + 16 * q);
                dst[2 * q] = (f2){v.x, v.y};
                dst[2 * q + 1] = (f2){v.z, v.w};
            }
        } else {
#pragma unroll
            for (int q = 0; q < D; ++q) dst[q] = *reinterpret_cast<const f2 *>(p + 8 * q);
        }
    };
#pragma unroll
    for (int b = 0; b < R - 1; ++b) load_block(lane, b, W[b]);
    const int nouter = Gp / R;
    for (int to = 0; to < nouter; ++to) {
        const char *row0 = lane + (size_t)to * STRIDE;
        const float *tp = tap_lds + to * RD;
#pragma unroll
        for (int ti = 0; ti < R; ++ti) {
            load_block(row0, ti + R - 1, W[(ti + R - 1) % R]);
            float c[D];
            if constexpr (D % 4 == 0) {
#pragma unroll
                for (int q = 0; q < D / 4; ++q) {
                    const f4 t4 = *reinterpret_cast<const f4 *>(tp + ti * D + 4 * q);
                    c[4 * q] = t4.x;
                    c[4 * q + 1] = t4.y;
                    c[4 * q + 2] = t4.z;
                    c[4 * q + 3] = t4.w;
                }
            } else if constexpr (D % 2 == 0) {
#pragma unroll
                for (int q = 0; q < D / 2; ++q) {
                    const f2 t2 = *reinterpret_cast<const f2 *>(tp + ti * D + 2 * q);
                    c[2 * q] = t2.x;
                    c[2 * q + 1] = t2.y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < D; ++q) c[q] = tp[ti * D + q];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    const f2 cc = {c[q], c[q]};
                    acc[r] = __builtin_elementwise_fma(W[(ti + r) % R][q], cc, acc[r]);
                }
            }
        }
    }
    __syncthreads();

    // ---- stage the R outputs of each lane, then store coalesced --------------
    {
        char *o = smem + (size_t)threadIdx.x * G::OUT_STRIDE;
#pragma unroll
        for (int r = 0; r < R; ++r) *reinterpret_cast<f2 *>(o + 8 * r) = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int e = threadIdx.x + k * T;
        const long m = mt + e;
        if (m < n_out) {
            const f2 v = *reinterpret_cast<const f2 *>(smem + (e / R) * G::OUT_STRIDE + (e % R) * 8);
            float2 w;
            w.x = v.x;
            w.y = v.y;
            out[m] = w;
        }
    }
}

template <int D, int R, int T>
static int launch_mfd(hipStream_t s, const FusedFirArgs &a) {
    using G = FirGeom<D, R>;
    constexpr int OUTS = T * R;
    const int rows = T + a.Gp / R;
    const size_t lds = (size_t)rows * G::STRIDE + (size_t)a.Gp * D * sizeof(float);
    auto fn = k_mix_fir_decim<D, R, T>;
    if (lds > 64 * 1024)
        RR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t ntiles = (a.n_out + OUTS - 1) / OUTS;
    if (ntiles > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fused FIR: too many tiles");
    const int aligned = (reinterpret_cast<uintptr_t>(a.in) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(fn, dim3((unsigned)ntiles), dim3(T), lds, s, (const float2 *)a.xh, (int)a.hx,
                       (const float2 *)a.in, (long)a.n_in, aligned, (const float2 *)a.nco, a.denom, a.idx0,
                       (const float *)a.taps, a.Gp, (float2 *)a.out, (long)a.n_out, (long)a.e0, (unsigned)ntiles);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

bool fused_fir_supported(uint64_t D, size_t Lc) {
    if (Lc == 0 || Lc > 4096) return false;  // LDS rows grow with the tap count
    return D == 2 || D == 4 || D == 8 || D == 10;
}

int fused_fir_R(uint64_t D) {
    switch (D) {
        case 2: return 16;
        case 4: return 8;
        case 8: return 4;
        case 10: return 3;
    }
    return 0;
}

int launch_fused_fir(hipStream_t s, const FusedFirArgs &a) {
    if (a.n_out == 0) return RR_OK;
    switch (a.D) {
        case 2: return launch_mfd<2, 16, 128>(s, a);
        case 4: return launch_mfd<4, 8, 128>(s, a);
        case 8: return launch_mfd<8, 4, 128>(s, a);
        case 10: return launch_mfd<10, 3, 128>(s, a);
    }
    RR_FAIL(RR_ERR_BAD_ARG, "fused FIR: decimation %u not instantiated", a.D);
}

// ---------------------------------------------------------------------------
// 4096-point windowed FFT, radix 16 x 3
// ---------------------------------------------------------------------------
__device__ __forceinline__ f2 cmulf(f2 a, f2 b) { return (f2){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ f2 mul_mj(f2 a) { return (f2){a.y, -a.x}; }  // * (-j)

// forward 4-point DFT (kernel e^{-j 2 pi n k / 4})
__device__ __forceinline__ void dft4(f2 &a, f2 &b, f2 &c, f2 &d) {
    const f2 s0 = a + c, s1 = a - c, s2 = b + d, s3 = mul_mj(b - d);
    a = s0 + s2;
    b = s1 + s3;
    c = s0 - s2;
    d = s1 - s3;
}

// in-register forward 16-point DFT, natural order in and out
__device__ __forceinline__ void dft16(f2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, H = 0.70710678118654752440f;
    // t[a][b] = DFT4 over m of v[a + 4m]
#pragma unroll
    for (int a = 0; a < 4; ++a) dft4(v[a], v[a + 4], v[a + 8], v[a + 12]);  // v[a + 4b] now holds t[a][b]
    // twiddle W16^(a b)
    v[1 + 4] = cmulf(v[1 + 4], (f2){C1, -S1});   // a=1,b=1: W^1
    v[1 + 8] = cmulf(v[1 + 8], (f2){H, -H});     // a=1,b=2: W^2
    v[1 + 12] = cmulf(v[1 + 12], (f2){S1, -C1}); // a=1,b=3: W^3
    v[2 + 4] = cmulf(v[2 + 4], (f2){H, -H});     // a=2,b=1: W^2
    v[2 + 8] = mul_mj(v[2 + 8]);                 // a=2,b=2: W^4 = -j
    v[2 + 12] = cmulf(v[2 + 12], (f2){-H, -H});  // a=2,b=3: W^6
    v[3 + 4] = cmulf(v[3 + 4], (f2){S1, -C1});   // a=3,b=1: W^3
    v[3 + 8] = cmulf(v[3 + 8], (f2){-H, -H});    // a=3,b=2: W^6
    v[3 + 12] = cmulf(v[3 + 12], (f2){-C1, S1}); // a=3,b=3: W^9
    // X[b + 4c] = DFT4 over a of t[a][b]; t[a][b] sits in v[a + 4b]
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);  // v[4b + c] = X[b + 4c]
    // reorder to natural order: X[k], k = b + 4c  <-  v[4b + c]
    f2 t[16];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) t[b + 4 * c] = v[4 * b + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = t[k];
}

__device__ __forceinline__ int pad16(int i) { return i + (i >> 4); }

__global__ __launch_bounds__(256) void k_fft4096(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                 const float *__restrict__ window, const float2 *__restrict__ tw,
                                                 int center_dc) {
    __shared__ f2 lds[4096 + 256];
    const int j = threadIdx.x;
    const float2 *src = in + (size_t)blockIdx.x * 4096;
    float2 *dst = out + (size_t)blockIdx.x * 4096;
    f2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float2 x = src[j + 256 * k];
        const float w = window[j + 256 * k];
        v[k] = (f2){x.x * w, x.y * w};
    }
    // pass 0 (Ns = 1): no twiddles; out index 16 j + k
    dft16(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad16(16 * j + k)] = v[k];
    __syncthreads();
    // pass 1 (Ns = 16): twiddle e^{-j 2 pi k (j mod 16) / 256}; out (j/16)*256 + j%16 + 16 k
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds[pad16(j + 256 * k)];
    {
        const int jm = j & 15;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float2 t = tw[16 * k * jm];
            v[k] = cmulf(v[k], (f2){t.x, t.y});
        }
    }
    dft16(v);
    __syncthreads();
    {
        const int base = (j >> 4) * 256 + (j & 15);
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[pad16(base + 16 * k)] = v[k];
    }
    __syncthreads();
    // pass 2 (Ns = 256): twiddle e^{-j 2 pi k j / 4096}; out j + 256 k
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds[pad16(j + 256 * k)];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float2 t = tw[k * j];
        v[k] = cmulf(v[k], (f2){t.x, t.y});
    }
    dft16(v);
    const int rot = center_dc ? 2048 : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int o = (j + 256 * k + rot) & 4095;
        float2 w;
        w.x = v[k].x;
        w.y = v[k].y;
        dst[o] = w;
    }
}

int launch_fft4096(hipStream_t s, const void *in, void *out, size_t count, const void *window, const void *tw4096,
                   bool center_dc) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fft4096: too many frames");
    hipLaunchKernelGGL(k_fft4096, dim3((unsigned)count), dim3(256), 0, s, (const float2 *)in, (float2 *)out,
                       (const float *)window, (const float2 *)tw4096, (int)center_dc);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// tail drop on an interrupt: new[i] = old[i - drop] (zeros shifted in at the front)
__global__ void k_drop_tail(const float2 *__restrict__ oldh, float2 *__restrict__ newh, int H, int drop) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    float2 v;
    v.x = 0.f;
    v.y = 0.f;
    if (i >= drop) v = oldh[i - drop];
    newh[i] = v;
}

int launch_drop_tail(hipStream_t s, const void *oldh, void *newh, size_t H, size_t drop) {
    if (H == 0) return RR_OK;
    hipLaunchKernelGGL(k_drop_tail, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, s, (const float2 *)oldh,
                       (float2 *)newh, (int)H, (int)drop);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
