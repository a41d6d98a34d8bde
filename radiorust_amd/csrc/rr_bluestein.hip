// rr_bluestein.hip — Bluestein's algorithm in ONE kernel for the Fourier block's chunk lengths that are not powers of two,
// Complex<f32> (analysis.rs:82-115 accepts any length): k_bluestein1024 (32 .. 512 points, a wave per chunk),
// k_bluestein4096 (513 .. 2048), k_bluestein_big<M> (2049 .. 8192 with a prime factor beyond 13; k_bluestein8192 on request).
// (split out of rr_fused.hip in round 3; derivations and dropped variants: DESIGN_HISTORY.md 4)
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"
#include "rr_meter_dev.hpp"
#include "rr_fft_regs.hpp"
#include "rr_fft_big.hpp"

#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdlib>
#include <utility>
#include <vector>

namespace rr {

// ---------------------------------------------------------------------------
// Kernel 2y  k_bluestein8192: the Fourier block for 2049 .. 4096 points whose length has a prime factor beyond 13 (the others
// run the mixed-radix passes): Bluestein's algorithm in ONE kernel around two 8192-point transforms in LDS, as
// k_bluestein4096 does around two 4096-point ones.  The result layout of fft8192_regs - lane t holds X[t + 256 m], m < 32 - is
// the input layout of the next transform (t + 256 h + 512 k = t + 256 (2 k + h)): between the two transforms the values stay in
// the lane, only their register names change.  The five launches it replaces took 0.56-0.75 ms per 2^24 samples.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bluestein8192(const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in,
                                                       long hop, int n, const float2 *__restrict__ c, const float2 *__restrict__ B,
                                                       const float2 *__restrict__ w, const float2 *__restrict__ tw,
                                                       float2 *__restrict__ out, int center_dc, unsigned count) {
    extern __shared__ __attribute__((aligned(16))) char bs8192_smem[];
    f2 *lds = reinterpret_cast<f2 *>(bs8192_smem);  // 8192 + 512 elements
    const int t = threadIdx.x;
    const unsigned fr = blockIdx.x;
    if (fr >= count) return;
    const long base = (long)fr * hop - n_head;
    f2 v[2][16];
    // (every load at a clamped index and selected afterwards, as k_bluestein4096)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float2 xs[16], cs[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = t + 256 * h + 512 * k;
            const int mc = m < n ? m : n - 1;
            const long i = base + mc;
            xs[k] = (i >= 0) ? in[i] : head[n_head + i];
            cs[k] = c[mc];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f2 p = cmul((f2){xs[k].x, xs[k].y}, (f2){cs[k].x, cs[k].y});
            v[h][k] = (t + 256 * h + 512 * k < n) ? p : (f2){0.f, 0.f};
        }
    }
    fft8192_regs(v, lds, tw, t);
    // * B, and into the next transform's input names: u[h][k] = Z[t + 256 (2 k + h)], Z[t + 256 m] = v[m >> 4][m & 15]
    f2 u[2][16];
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        const float2 b = B[t + 256 * m];
        u[m & 1][m >> 1] = cmul(v[m >> 4][m & 15], (f2){b.x, b.y});
    }
    __syncthreads();  // the first transform's last pass has been read
    fft8192_regs(u, lds, tw, t);
    // u[m >> 4][m & 15] = DFT(Z)[t + 256 m] = 8192 IDFT(Z)[tau], tau = (8192 - t - 256 m) mod 8192; bins tau < n leave, times conj(chirp[tau])
    float2 *dst = out + (size_t)fr * n;
    const int rot = center_dc ? n / 2 : 0;  // rotate_right(n / 2)
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        const int tau = (8192 - t - 256 * m) & 8191;
        const float2 wt = w[tau < n ? tau : 0];
        const f2 r = cmul_conj(u[m >> 4][m & 15], (f2){wt.x, wt.y});
        int o = tau + rot;
        if (o >= n) o -= n;
        if (tau < n) dst[o] = float2{r.x, r.y};
    }
}

bool bluestein8192_supported(int dtype, size_t n) { return dtype == RR_F32 && n > 2048 && n <= 4096 && (n & (n - 1)) != 0; }

int launch_bluestein8192(hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                         const void *B, const void *w, const void *tw8192, void *out, bool center_dc, size_t count) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many chunks in one call");
    const size_t lds = (8192 + 512) * sizeof(float2);
    RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(k_bluestein8192), lds));
    hipLaunchKernelGGL(k_bluestein8192, dim3((unsigned)count), dim3(256), lds, s, (const float2 *)head, (long)n_head,
                       (const float2 *)in, (long)hop, (int)n, (const float2 *)c, (const float2 *)B, (const float2 *)w,
                       (const float2 *)tw8192, (float2 *)out, (int)center_dc, (unsigned)count);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Kernel 2z  k_bluestein_big<N>: the same algorithm around two transforms of N = 8192 / 16 384 points by rr_fft_big.hpp's
// workgroup transform (N / 16 lanes with 16 values each): 4097 .. 8192 points with a prime factor beyond 13 in ONE kernel instead of
// four launches of the two-pass tile transform through HBM (which move 4 x 16 384 x 16 bytes per chunk for 16 n algorithmic ones),
// and 2049 .. 4096 points around 8192-point ones (k_bluestein8192's register transforms stay behind RR_FOURIER_BS8K=regs).  Lane j's results X[j + T k] of the first transform are the second
// one's inputs at the same index: between the two the values stay in their registers.  B is read in pairs:
// Bp[(kp T + j) 2 + h] = B[j + T (2 kp + h)] (as k_filter_blkbig's G).
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(N / 16) void k_bluestein_big(const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in,
                                                          long hop, int n, const float2 *__restrict__ c, const void *__restrict__ Bp,
                                                          const float2 *__restrict__ w, const float2 *__restrict__ tw,
                                                          float2 *__restrict__ out, int center_dc, unsigned count) {
    constexpr int T = N / 16;
    extern __shared__ __attribute__((aligned(16))) f2 bsbig_smem[];
    f2 *const img = bsbig_smem;
    f2 *const tab = bsbig_smem + (N + N / 16);
    const int j = threadIdx.x;
    const unsigned fr = blockIdx.x;
    if (fr >= count) return;
    const long base = (long)fr * hop - n_head;
    f2 v[16];
    {
        // n <= N / 2: the values k >= 8 are the zero padding; loads at a clamped index and selected afterwards (k_bluestein4096)
        float2 xs[8], cs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = j + T * k;
            const int mc = m < n ? m : n - 1;
            const long i = base + mc;
            xs[k] = (i >= 0) ? in[i] : head[n_head + i];
            cs[k] = c[mc];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f2 p = cmul((f2){xs[k].x, xs[k].y}, (f2){cs[k].x, cs[k].y});
            v[k] = (j + T * k < n) ? p : (f2){0.f, 0.f};
        }
#pragma unroll
        for (int k = 8; k < 16; ++k) v[k] = (f2){0.f, 0.f};
    }
    BigFftLane<N> ln;
    ln.init(tw, tab, j);
    const __amdgpu_buffer_rsrc_t rsB = rsrc_of(Bp, 8u * N);
    float4 g4[8];
    big_fft<N, true>(v, img, ln, j, false, [&] {
        if constexpr (N == 8192) {
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) g4[kp] = buf_ld_f4<0>(rsB, 16u * j, 16u * T * kp);
        }
    });
    if constexpr (N == 8192) {
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
            v[2 * kp] = cmul(v[2 * kp], (f2){g4[kp].x, g4[kp].y});
            v[2 * kp + 1] = cmul(v[2 * kp + 1], (f2){g4[kp].z, g4[kp].w});
        }
    } else {  // (128 registers per lane at 1024 lanes: B in two halves, as k_filter_blkbig)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) g4[kp] = buf_ld_f4<0>(rsB, 16u * j, 16u * T * (4 * h + kp));
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                v[8 * h + 2 * kp] = cmul(v[8 * h + 2 * kp], (f2){g4[kp].x, g4[kp].y});
                v[8 * h + 2 * kp + 1] = cmul(v[8 * h + 2 * kp + 1], (f2){g4[kp].z, g4[kp].w});
            }
        }
    }
    // the chirp values of the bins this lane stores: tau = (N - j - T k) mod N < n needs j + T k > N / 2 (k >= 8), or j = k = 0
    float2 wt[9];
    big_fft<N, true>(v, img, ln, j, true, [&] {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int k = q == 0 ? 0 : q + 7;
            const int tau = (N - j - T * k) & (N - 1);
            wt[q] = w[tau < n ? tau : 0];
        }
    });
    float2 *dst = out + (size_t)fr * n;
    const int rot = center_dc ? n / 2 : 0;  // rotate_right(n / 2)
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int k = q == 0 ? 0 : q + 7;
        const int tau = (N - j - T * k) & (N - 1);
        const f2 r = cmul_conj(v[k], (f2){wt[q].x, wt[q].y});
        int o = tau + rot;
        if (o >= n) o -= n;
        if (tau < n) dst[o] = float2{r.x, r.y};
    }
}

bool bluestein_big_supported(int dtype, size_t n, size_t *M) {
    if (dtype != RR_F32 || (n & (n - 1)) == 0) return false;
    if (n > 4096 && n <= 8192) {
        *M = 16384;
        return true;
    }
    if (n > 2048 && n <= 4096) {
        *M = 8192;
        return true;
    }
    return false;
}

template <int N>
static int launch_bluestein_big_n(hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                                  const void *Bp, const void *w, const void *twN, void *out, bool center_dc, size_t count) {
    constexpr size_t lds = (size_t)big_fft_lds_elems<N>() * sizeof(f2);
    RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(k_bluestein_big<N>), lds));
    hipLaunchKernelGGL(k_bluestein_big<N>, dim3((unsigned)count), dim3(N / 16), lds, s, (const float2 *)head, (long)n_head,
                       (const float2 *)in, (long)hop, (int)n, (const float2 *)c, Bp, (const float2 *)w, (const float2 *)twN,
                       (float2 *)out, (int)center_dc, (unsigned)count);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_bluestein_big(hipStream_t s, size_t M, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                         const void *Bp, const void *w, const void *twM, void *out, bool center_dc, size_t count) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many chunks in one call");
    if (n > M / 2 || (M != 8192 && M != 16384)) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: %zu points around transforms of %zu", n, M);
    return M == 8192 ? launch_bluestein_big_n<8192>(s, head, n_head, in, hop, n, c, Bp, w, twM, out, center_dc, count)
                     : launch_bluestein_big_n<16384>(s, head, n_head, in, hop, n, c, Bp, w, twM, out, center_dc, count);
}

// ---------------------------------------------------------------------------
// Kernel 2b  k_bluestein4096: the Fourier block for chunk lengths 513 .. 2048 that are not powers of two
// (analysis.rs:82-115 accepts any length), Bluestein's algorithm in ONE kernel, a workgroup per chunk:
//   v = x c (c = window conj(chirp), zero beyond n)  ->  DFT_4096  ->  * B (B = DFT_4096(chirp, wrapped) / 4096)
//   ->  the inverse as a second forward DFT_4096 read at the reversed index  ->  * conj(chirp)  ->  n bins.
// The five-launch form (k_bs_pre, k_fftM, k_bs_mul, k_fftM, k_bs_post through HBM) moves ~10 M 8 bytes per chunk
// for 16 n algorithmic ones; here a chunk is read once and written once.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bluestein4096(const float2 *__restrict__ head, long n_head,
                                                       const float2 *__restrict__ in, long hop, int n,
                                                       const float2 *__restrict__ c, const float2 *__restrict__ B,
                                                       const float2 *__restrict__ w, const float2 *__restrict__ tw,
                                                       float2 *__restrict__ out, int center_dc, unsigned count) {
    __shared__ f2 lds[4096 + 256];
    const int j = threadIdx.x;
    const unsigned fr = blockIdx.x;
    if (fr >= count) return;
    const long base = (long)fr * hop - n_head;
    f2 v[16];
    // every load is issued unconditionally at a clamped index and selected afterwards: under a condition each of
    // them would be a round trip of its own (measured: 41 us per chunk instead of 11)
    {
        float2 xs[16], cs[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = j + 256 * k;
            const int mc = m < n ? m : n - 1;
            const long i = base + mc;
            xs[k] = (i >= 0) ? in[i] : head[n_head + i];
            cs[k] = c[mc];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f2 p = cmul((f2){xs[k].x, xs[k].y}, (f2){cs[k].x, cs[k].y});
            v[k] = (j + 256 * k < n) ? p : (f2){0.f, 0.f};
        }
    }
    fft4096_regs(v, lds, tw, j);
    {
        float2 b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) b[k] = B[j + 256 * k];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = cmul(v[k], (f2){b[k].x, b[k].y});
    }
    __syncthreads();  // the first transform's last pass has been read
    // (the chirp values of the bins this lane will store, requested ahead of the second transform)
    float2 wt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int t = (4096 - j - 256 * k) & 4095;
        wt[k] = w[t < n ? t : 0];
    }
    fft4096_regs(v, lds, tw, j);
    // v[k] = DFT(Z)[j + 256 k] = 4096 IDFT(Z)[t], t = (4096 - j - 256 k) mod 4096; bins t < n leave, times conj(chirp[t])
    float2 *dst = out + (size_t)fr * n;
    const int rot = center_dc ? n / 2 : 0;  // rotate_right(n / 2)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int t = (4096 - j - 256 * k) & 4095;
        const f2 r = cmul_conj(v[k], (f2){wt[k].x, wt[k].y});
        int o = t + rot;
        if (o >= n) o -= n;
        if (t < n) dst[o] = float2{r.x, r.y};
    }
}

bool bluestein4096_supported(int dtype, size_t n) { return dtype == RR_F32 && n > 512 && n <= 2048 && (n & (n - 1)) != 0; }

int launch_bluestein4096(hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                         const void *B, const void *w, const void *tw4096, void *out, bool center_dc, size_t count) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many chunks in one call");
    hipLaunchKernelGGL(k_bluestein4096, dim3((unsigned)count), dim3(256), 0, s, (const float2 *)head, (long)n_head,
                       (const float2 *)in, (long)hop, (int)n, (const float2 *)c, (const float2 *)B, (const float2 *)w,
                       (const float2 *)tw4096, (float2 *)out, (int)center_dc, (unsigned)count);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Kernel 2c  k_bluestein1024: the Fourier block for chunk lengths 32 .. 512 that are not powers of two, Bluestein's
// algorithm with ONE WAVE per chunk - k_filter_wave's two transforms around a table product:
//   v = x c (c = window conj(chirp), zero beyond n) -> DFT_1024 -> * B (B = DFT_1024(chirp, wrapped) / 1024)
//   -> IDFT_1024 (the same network run backwards on the conjugate) -> * conj(chirp) -> n bins.
// c carries one more (zero) entry when n is odd, so that a lane's pair (2 l, 2 l + 1) is one 16-byte read; B is
// pair-interleaved like k_filter_wave's response.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RR_V_FLTWOCC, RR_V_FLTWOCC))) void k_bluestein1024(
    const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in, long hop, int n,
    const float2 *__restrict__ c, const float2 *__restrict__ B, const float2 *__restrict__ w,
    const float2 *__restrict__ tw, float2 *__restrict__ out, int center_dc, unsigned count) {
    __shared__ __attribute__((aligned(16))) f2 lds[kWaveLds];
    const int l = threadIdx.x;
    // chunks dealt to the XCDs in a moving window, 16 neighbouring chunks per XCD (as k_fft1024's frames)
    const unsigned fr = blockIdx.x / 128 * 128 + (blockIdx.x % 128 & 7) * 16 + (blockIdx.x % 128 >> 3);
    if (fr >= count) return;
    const long base = (long)fr * hop - n_head;
    // every load at a clamped index, selected afterwards (a load under a condition is a round trip of its own)
    f2 v[16];
    {
        float2 xs[16];
        float4 cs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = 2 * l + 128 * k;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int mc = m + j < n ? m + j : n - 1;
                const long i = base + mc;
                xs[2 * k + j] = (i >= 0) ? in[i] : head[n_head + i];
            }
            cs[k] = *reinterpret_cast<const float4 *>(c + (m < n ? m : 0));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = 2 * l + 128 * k;
            const f2 p0 = cmul((f2){xs[2 * k].x, xs[2 * k].y}, (f2){cs[k].x, cs[k].y});
            const f2 p1 = cmul((f2){xs[2 * k + 1].x, xs[2 * k + 1].y}, (f2){cs[k].z, cs[k].w});
            v[2 * k] = m < n ? p0 : (f2){0.f, 0.f};
            v[2 * k + 1] = m + 1 < n ? p1 : (f2){0.f, 0.f};
        }
    }
    f2 t_p1, t_p2[2];
    {
        const float4 *tl = reinterpret_cast<const float4 *>(tw + 1024) + l;
        const float4 s0 = tl[0], s1 = tl[64];
        t_p1 = (f2){s0.x, s0.y};
        t_p2[0] = (f2){s0.z, s0.w};
        t_p2[1] = (f2){s1.x, s1.y};
    }
    float4 h4[8];
    f2 X[16];
    wave_dft1024(v, X, lds, l, t_p1, t_p2, [&] {
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) h4[kp] = reinterpret_cast<const float4 *>(B)[l + 64 * kp];
    });
    // the chirp values of the bins this lane will store, requested ahead of the second transform
    float2 wt[16];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = 2 * l + j + 128 * k;
            wt[2 * k + j] = w[t < n ? t : 0];
        }
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
        const f2 p0 = cmul(X[2 * kp], (f2){h4[kp].x, h4[kp].y}), p1 = cmul(X[2 * kp + 1], (f2){h4[kp].z, h4[kp].w});
        X[2 * kp] = (f2){p0.x, -p0.y};
        X[2 * kp + 1] = (f2){p1.x, -p1.y};
    }
    wave_sync();  // the forward image has been read
    wave_dft1024_t(X, v, lds, l, t_p1, t_p2);
    // conj(v[2 k + j]) = IDFT(Z)[t], t = 2 l + j + 128 k; bins t < n leave, times conj(chirp[t])
    float2 *dst = out + (size_t)fr * n;
    const int rot = center_dc ? n / 2 : 0;  // rotate_right(n / 2)
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = 2 * l + j + 128 * k;
            const f2 y = {v[2 * k + j].x, -v[2 * k + j].y};
            const f2 r = cmul_conj(y, (f2){wt[2 * k + j].x, wt[2 * k + j].y});
            int o = t + rot;
            if (o >= n) o -= n;
            if (t < n) dst[o] = float2{r.x, r.y};
        }
}

bool bluestein1024_supported(int dtype, size_t n) { return dtype == RR_F32 && n >= 32 && n <= 512 && (n & (n - 1)) != 0; }

int launch_bluestein1024(hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                         const void *Bp, const void *w, const void *tw1024, void *out, bool center_dc, size_t count) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: too many chunks in one call");
    const unsigned grid = (unsigned)((count + 127) / 128 * 128);
    hipLaunchKernelGGL(k_bluestein1024, dim3(grid), dim3(64), 0, s, (const float2 *)head, (long)n_head, (const float2 *)in,
                       (long)hop, (int)n, (const float2 *)c, (const float2 *)Bp, (const float2 *)w, (const float2 *)tw1024,
                       (float2 *)out, (int)center_dc, (unsigned)count);
    RR_HIP(hipGetLastError());
    return RR_OK;
}


}  // namespace rr
