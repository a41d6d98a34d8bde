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

// pairs of samples each lane prefetches per tile (upper bound of the real count)
constexpr int kNPF = 18;

template <int D, int R, int T>
__global__ __launch_bounds__(T, 2) void k_mix_fir_decim(const float2 *__restrict__ xh, int hx,
                                                        const float2 *__restrict__ in, long n_in, int in_aligned16,
                                                        const float2 *__restrict__ nco, unsigned denom, unsigned idx0,
                                                        const float *__restrict__ taps, int Gp,
                                                        float2 *__restrict__ out, long n_out, int out_aligned16,
                                                        long e0, unsigned ntiles, unsigned tiles_per_wg) {
    using G = FirGeom<D, R>;
    constexpr int RD = G::RD, STRIDE = G::STRIDE;
    constexpr int OUTS = T * R;
    static_assert(RD == 32 && (2 * T) % RD == 0, "LDS write addresses advance by whole rows per prefetch slot");
    constexpr int ROWS_PER_SLOT = 2 * T / RD;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // 32 spare bytes in front: when the tile starts at an odd sample, the first lane's
    // pair straddles the tile start and its first half (sample -1 = row -1, column 31)
    // lands there; slack rows at the end take the pairs past the tile end
    char *smem = smem_raw + 32;
    const int rows = T + Gp / R;
    const int NS = rows * RD;
    const int lds_rows = rows + ROWS_PER_SLOT + 1;
    float *tap_lds = reinterpret_cast<float *>(smem + (size_t)lds_rows * STRIDE);
    for (int i = threadIdx.x; i < Gp * D; i += T) tap_lds[i] = taps[i];

    // XCD-aware work split: workgroups b, b+8, b+16.. share an XCD (round robin);
    // give each workgroup a contiguous run of tiles and neighbouring runs to one XCD,
    // so the halo a tile shares with its predecessor is an L2 hit.
    unsigned chunk;
    {
        const unsigned b = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, rmd = nwg & 7, xcd = b & 7;
        chunk = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (b >> 3);
    }
    const unsigned tile_begin = chunk * tiles_per_wg;
    unsigned tile_end = tile_begin + tiles_per_wg;
    if (tile_end > ntiles) tile_end = ntiles;
    if (tile_begin >= tile_end) return;

    // per-lane constants of the load phase -------------------------------------
    const long lo0 = e0 - (long)D * Gp + 1;            // tile_lo of tile 0
    const int odd = (int)(lo0 & 1);                    // same for every tile (D*OUTS is even)
    const int npairs = (NS + odd + 1) >> 1;
    const int sfirst = 2 * (int)threadIdx.x - odd;     // LDS sample index of this lane's first prefetched sample
    // byte address of sample s (floor division, so s = -1 is row -1, column 31 = -24)
    auto lds_addr = [&](int s) -> int { return (s >> 5) * STRIDE + (s & 31) * 8; };
    const int a0 = lds_addr(sfirst), a1 = lds_addr(sfirst + 1);
    const unsigned step = (unsigned)((2 * T) % denom);
    const unsigned tstep = (unsigned)(((long)D * OUTS) % denom);
    const bool nco_const = (step == 0 && tstep == 0);  // phasor of a lane never changes (e.g. denom = 8)

    auto tile_lo_of = [&](unsigned tile) -> long { return lo0 + (long)D * OUTS * tile; };
    auto interior_of = [&](long tile_lo) -> bool {
        const long le = tile_lo - odd;
        return in_aligned16 && le >= 0 && le + 2L * kNPF * T <= n_in;  // the whole prefetch window is inside `in`
    };
    long tile_lo = tile_lo_of(tile_begin);
    unsigned rbase;
    {
        long ph = ((long)idx0 + (tile_lo - odd) + 2 * (long)threadIdx.x) % (long)denom;
        if (ph < 0) ph += denom;
        rbase = (unsigned)ph;
    }
    float2 pc0 = nco[rbase], pc1 = nco[(rbase + 1 == denom) ? 0 : rbase + 1];

    f4 x[kNPF];
    auto prefetch = [&](long tlo) {
        const f4 *src = reinterpret_cast<const f4 *>(in + (tlo - odd)) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < kNPF; ++u) x[u] = src[u * T];
    };
    bool cur_interior = interior_of(tile_lo);
    if (cur_interior) prefetch(tile_lo);

    for (unsigned tile = tile_begin; tile < tile_end; ++tile) {
        // ---- stage: (prefetched) raw samples -> mix -> LDS ----------------------
        if (cur_interior) {
            if (nco_const) {
#pragma unroll
                for (int u = 0; u < kNPF; ++u) {
                    const bool ok = (int)threadIdx.x + u * T < npairs;
                    const f2 v0 = {x[u].x * pc0.x - x[u].y * pc0.y, x[u].x * pc0.y + x[u].y * pc0.x};
                    const f2 v1 = {x[u].z * pc1.x - x[u].w * pc1.y, x[u].z * pc1.y + x[u].w * pc1.x};
                    if (ok) {
                        *reinterpret_cast<f2 *>(smem + a0 + u * ROWS_PER_SLOT * STRIDE) = v0;
                        *reinterpret_cast<f2 *>(smem + a1 + u * ROWS_PER_SLOT * STRIDE) = v1;
                    }
                }
            } else {
                unsigned rr_ = rbase;
#pragma unroll
                for (int u = 0; u < kNPF; ++u) {
                    const bool ok = (int)threadIdx.x + u * T < npairs;
                    const float2 p0 = nco[rr_], p1 = nco[(rr_ + 1 == denom) ? 0 : rr_ + 1];
                    const f2 v0 = {x[u].x * p0.x - x[u].y * p0.y, x[u].x * p0.y + x[u].y * p0.x};
                    const f2 v1 = {x[u].z * p1.x - x[u].w * p1.y, x[u].z * p1.y + x[u].w * p1.x};
                    if (ok) {
                        *reinterpret_cast<f2 *>(smem + a0 + u * ROWS_PER_SLOT * STRIDE) = v0;
                        *reinterpret_cast<f2 *>(smem + a1 + u * ROWS_PER_SLOT * STRIDE) = v1;
                    }
                    rr_ += step;
                    if (rr_ >= denom) rr_ -= denom;
                }
            }
        } else {
            // edge tiles: history (already mixed), end of input, unaligned input
            unsigned rr_ = rbase;
            const long lo_even = tile_lo - odd;
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
                        const float2 xx = in[pos];
                        const float2 pp = nco[k ? r1 : rr_];
                        v = (f2){xx.x * pp.x - xx.y * pp.y, xx.x * pp.y + xx.y * pp.x};
                    }
                    const int sidx = (int)(pos - tile_lo);
                    if (sidx >= 0 && sidx < NS) *reinterpret_cast<f2 *>(smem + lds_addr(sidx)) = v;
                }
                rr_ += step;
                if (rr_ >= denom) rr_ -= denom;
            }
        }
        __syncthreads();

        // ---- prefetch the next tile while this one is filtered -------------------
        const long next_lo = tile_lo + (long)D * OUTS;
        const bool next_interior = (tile + 1 < tile_end) && interior_of(next_lo);
        if (next_interior) prefetch(next_lo);

        // ---- FIR: rotating register window, packed FMAs --------------------------
        f2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = (f2){0.f, 0.f};
        {
            f2 W[R][D];
            const char *lane = smem + (size_t)threadIdx.x * STRIDE;
            // block b of this lane lives at row b / R, column block b % R
            auto load_block = [&](const char *row0, int b_static, f2(&dst)[D]) {
                const char *p = row0 + (b_static / R) * STRIDE + (b_static % R) * (D * 8);
#pragma unroll
                for (int q = 0; q < D / 2; ++q) {
                    const f4 v = *reinterpret_cast<const f4 *>(p + 16 * q);
                    dst[2 * q] = (f2){v.x, v.y};
                    dst[2 * q + 1] = (f2){v.z, v.w};
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
                    } else {
#pragma unroll
                        for (int q = 0; q < D / 2; ++q) {
                            const f2 t2 = *reinterpret_cast<const f2 *>(tp + ti * D + 2 * q);
                            c[2 * q] = t2.x;
                            c[2 * q + 1] = t2.y;
                        }
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
        }
        __syncthreads();  // every wave is done reading this tile's samples

        // ---- store: R consecutive outputs per lane ---------------------------------
        {
            const long m0 = (long)tile * OUTS + (long)threadIdx.x * R;
            float2 *o = out + m0;
            if (out_aligned16 && m0 + R <= n_out) {
#pragma unroll
                for (int r = 0; r < R; r += 2)
                    *reinterpret_cast<f4 *>(o + r) = (f4){acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y};
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (m0 + r < n_out) {
                        float2 w;
                        w.x = acc[r].x;
                        w.y = acc[r].y;
                        o[r] = w;
                    }
            }
        }
        tile_lo = next_lo;
        cur_interior = next_interior;
        rbase += tstep;
        if (rbase >= denom) rbase -= denom;
    }
}

template <int D, int R, int T>
static int launch_mfd(hipStream_t s, const FusedFirArgs &a) {
    using G = FirGeom<D, R>;
    constexpr int OUTS = T * R;
    const int rows = T + a.Gp / R;
    const int lds_rows = rows + 2 * T / G::RD + 1;
    const size_t lds = 32 + (size_t)lds_rows * G::STRIDE + (size_t)a.Gp * D * sizeof(float);
    if ((rows * G::RD + 2) / 2 > kNPF * T) RR_FAIL(RR_ERR_BAD_ARG, "fused FIR: %d tap groups exceed the prefetch window", a.Gp);
    auto fn = k_mix_fir_decim<D, R, T>;
    if (lds > 64 * 1024)
        RR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t ntiles = (a.n_out + OUTS - 1) / OUTS;
    if (ntiles > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fused FIR: too many tiles");
    // persistent grid: 4 workgroups of 2 waves per CU (LDS-limited), 256 CUs
    size_t nwg = 256 * 4;
    if (nwg > ntiles) nwg = ntiles;
    const size_t tpw = (ntiles + nwg - 1) / nwg;
    nwg = (ntiles + tpw - 1) / tpw;
    const int in_al = (reinterpret_cast<uintptr_t>(a.in) % 16 == 0) ? 1 : 0;
    const int out_al = (reinterpret_cast<uintptr_t>(a.out) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(T), lds, s, (const float2 *)a.xh, (int)a.hx, (const float2 *)a.in,
                       (long)a.n_in, in_al, (const float2 *)a.nco, a.denom, a.idx0, (const float *)a.taps, a.Gp,
                       (float2 *)a.out, (long)a.n_out, out_al, (long)a.e0, (unsigned)ntiles, (unsigned)tpw);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

bool fused_fir_supported(uint64_t D, size_t Lc) {
    if (!(D == 2 || D == 4 || D == 8) || Lc == 0) return false;
    // tap groups (padded to a multiple of R) must fit the per-lane prefetch window
    const int R = fused_fir_R(D);
    const size_t groups = (Lc + D - 1) / D;
    const size_t gp = (groups + R - 1) / R * R;
    const size_t rows = 128 + gp / R;
    return (rows * 32 + 2) / 2 <= (size_t)kNPF * 128;
}

int fused_fir_R(uint64_t D) {
    switch (D) {
        case 2: return 16;
        case 4: return 8;
        case 8: return 4;
    }
    return 0;
}

int launch_fused_fir(hipStream_t s, const FusedFirArgs &a) {
    if (a.n_out == 0) return RR_OK;
    switch (a.D) {
        case 2: return launch_mfd<2, 16, 128>(s, a);
        case 4: return launch_mfd<4, 8, 128>(s, a);
        case 8: return launch_mfd<8, 4, 128>(s, a);
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
