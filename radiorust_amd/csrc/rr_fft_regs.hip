// rr_fft_regs.hip — the Fourier block's register-resident transforms for Complex<f32> (analysis.rs:105-115), frames from
// [ head | in ] at any hop (Fourier, Stft, the chain's last stage), and the polyphase channelizers built on them:
//   k_fft64 / 128 / 512 / 1024 / 2048 / 4096 / 8192, k_fft4096_bank, k_stft4096 (overlapping frames, sliding window in registers),
//   k_chan512_quad / k_chan1024_multi / k_chan2048_pair / k_chan4096_pair (several neighbouring frames per wave / workgroup).
// METER instances compute metering::bandwidth behind the transform (rr_meter_dev.hpp).
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
// 4096-point windowed FFT, radix 16 x 3
#ifndef RR_V_FFT_LD_NT
#define RR_V_FFT_LD_NT 1  // streaming hint on the frame loads when frames do not overlap
#endif
// ---------------------------------------------------------------------------
// The input stream of frames is [ head (n_head samples) | in ]: the head is the
// Downsampler's partly filled output chunk left over by the previous call.
// (frames round robin over the XCDs: a contiguous eighth per XCD measured 0.181 against 0.176 ms per 2^26 samples, a
//  moving window no gain)
// FOLD: the polyphase channelizer with 4096 bins - the frame is the fold of `branches` windowed chunks,
// v[i] = sum_p w[i + 4096 p] x[base + i + 4096 p] (window: 4096 branches plain values), then the same transform.
// METER: metering::bandwidth (and the frame's energy) computed from the bins while they are still in registers
// (rr_meter_dev.hpp); fm.store = 0 drops the spectra altogether.
template <bool FOLD, bool METER = false>
__device__ __forceinline__ void fft4096_body(const float2 *__restrict__ head, long n_head,
                                             const float2 *__restrict__ in, float2 *__restrict__ out,
                                             const float *__restrict__ window, const float2 *__restrict__ tw,
                                             int center_dc, long hop, unsigned count, int branches, const FrameMeter &fm, const unsigned bx) {
    __shared__ f2 lds[4096 + 256];
    const int j = threadIdx.x;
    // (frames in reverse order - the most recently written first - measured no different in the chain)
    // FOLD: a frame shares branches - 1 of its chunks with each neighbour: neighbouring frames go to one XCD (workgroups b, b + 8, ..
    // share one), 8 at a time, so that a chunk is fetched into one L2 instead of into `branches` of them
    const unsigned fr = FOLD ? bx / 64 * 64 + (bx % 64 & 7) * 8 + (bx % 64 >> 3) : bx;
    if (FOLD && fr >= count) return;
    const long base = (long)fr * hop - n_head;  // index into `in` of this frame's first sample
    float2 *dst = out + (size_t)fr * 4096;
    f2 v[16];
    if constexpr (FOLD) {
        const float2 s1 = tw[16 * (j & 15)], s2 = tw[j];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (f2){0.f, 0.f};
        for (int p = 0; p < branches; ++p) {
            const long bp = base + 4096L * p + j;
            float2 x[16];
            float w[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long i = bp + 256 * k;
                x[k] = (i >= 0) ? in[i] : head[n_head + i];
                w[k] = window[4096 * p + j + 256 * k];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = __builtin_elementwise_fma((f2){x[k].x, x[k].y}, (f2){w[k], w[k]}, v[k]);
        }
        dft16(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(lds + pad16(16 * j + k), v[k]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = lds_ld(lds + pad16(j + 256 * k));
        apply_twiddle_powers(v, (f2){s1.x, s1.y});
        dft16(v);
        __syncthreads();
        {
            const int b2 = (j >> 4) * 256 + (j & 15);
#pragma unroll
            for (int k = 0; k < 16; ++k) lds_st(lds + (b2 + 16 * k), v[k]);  // (second exchange: no padding, see below)
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = lds_ld(lds + (j + 256 * k));
        apply_twiddle_powers(v, (f2){s2.x, s2.y});
        dft16(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<f2 *>(dst) + (j + 256 * k));
        return;
    }
    // the lane's 16 window values as 4 loads of 16 bytes (packed copy behind the table), its two twiddle
    // seeds up front, the frame's samples with the streaming hint when frames do not overlap
    float wv[16];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(window + 4096) + 4 * j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w4 = wp[q];
            wv[4 * q] = w4.x;
            wv[4 * q + 1] = w4.y;
            wv[4 * q + 2] = w4.z;
            wv[4 * q + 3] = w4.w;
        }
    }
    const float2 s1 = tw[16 * (j & 15)], s2 = tw[j];
    if (base >= 0 && hop >= 4096) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f2 x = RR_V_FFT_LD_NT ? __builtin_nontemporal_load(reinterpret_cast<const f2 *>(in + base + j) + 256 * k)
                                        : *(reinterpret_cast<const f2 *>(in + base + j) + 256 * k);
            v[k] = x * wv[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long i = base + j + 256 * k;
            const float2 x = (i >= 0) ? in[i] : head[n_head + i];
            v[k] = (f2){x.x * wv[k], x.y * wv[k]};
        }
    }
    // pass 0 (Ns = 1): no twiddles; out index 16 j + k
    dft16(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds_st(lds + pad16(16 * j + k), v[k]);
    __syncthreads();
    // pass 1 (Ns = 16): twiddle e^{-j 2 pi k (j mod 16) / 256}; out (j/16)*256 + j%16 + 16 k
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds_ld(lds + pad16(j + 256 * k));
    {
        // e^{-j 2 pi k (j mod 16) / 256} = w^k with w = tw[16 (j mod 16)]: one
        // table read, powers by a depth-4 product tree (error ~4 ulp, not 15)
        const float2 t = s1;
        apply_twiddle_powers(v, (f2){t.x, t.y});
    }
    dft16(v);
    __syncthreads();
    {
        // The second exchange uses the image WITHOUT padding: its stores go in groups of 16 lanes = 16 neighbouring elements
        // (conflict-free under any layout), and the reads - halves of 32 lanes over 64 banks - want their 32 neighbouring
        // elements in one piece; the padded rows of the first exchange (needed by ITS stores, 16 elements apart per lane)
        // cost every read of a half-wave a second cycle.
        const int base = (j >> 4) * 256 + (j & 15);
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(lds + (base + 16 * k), v[k]);
    }
    __syncthreads();
    // pass 2 (Ns = 256): twiddle e^{-j 2 pi k j / 4096}; out j + 256 k
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds_ld(lds + (j + 256 * k));
    {
        const float2 t = s2;
        apply_twiddle_powers(v, (f2){t.x, t.y});
    }
    dft16(v);
    const int rot = center_dc ? 2048 : 0;
    if (!METER || fm.store) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int o = (j + 256 * k + rot) & 4095;
            __builtin_nontemporal_store(v[k], reinterpret_cast<f2 *>(dst) + o);
        }
    }
    if constexpr (METER) {
        double total;
        const int tw_ = (int)(fr & 3u);  // (the finishing wave rotates with the frame: rr_meter_dev.hpp)
        const double bw = frame4096_bandwidth(v, lds, j, rot, fm.double_percentile, fm.sample_rate, &total, tw_);
        if (j == 64 * tw_) {
            fm.bw[fr] = bw;
            if (fm.energy) fm.energy[fr] = total;
        }
    }
}

template <bool FOLD, bool METER = false>
__global__ __launch_bounds__(256) void k_fft4096(const float2 *__restrict__ head, long n_head,
                                                 const float2 *__restrict__ in, float2 *__restrict__ out,
                                                 const float *__restrict__ window, const float2 *__restrict__ tw,
                                                 int center_dc, long hop, unsigned count, int branches, FrameMeter fm) {
    fft4096_body<FOLD, METER>(head, n_head, in, out, window, tw, center_dc, hop, count, branches, fm, blockIdx.x);
}

// the channels of a bank: frame blockIdx.x of channel blockIdx.y (see k_ols_wave_bank)
__global__ __launch_bounds__(256) void k_fft4096_bank(const BankTable chan, long n_head, const float *__restrict__ window,
                                                      const float2 *__restrict__ tw, int center_dc, unsigned count) {
    const BankPtrs c = chan.c[blockIdx.y];
    fft4096_body<false, false>((const float2 *)c.head, n_head, (const float2 *)c.dec, (float2 *)c.out, window, tw, center_dc, 4096L, count,
                               1, FrameMeter{}, blockIdx.x);
}

int launch_fft4096(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw4096, bool center_dc, size_t hop, hipEvent_t ev_start,
                   hipEvent_t ev_stop, const FrameMeter *fm) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fft4096: too many frames");
    const unsigned grid = (unsigned)count;
    if (fm) {
        hipLaunchKernelGGL((k_fft4096<false, true>), dim3(grid), dim3(256), 0, s, (const float2 *)head, (long)n_head,
                           (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw4096,
                           (int)center_dc, (long)hop, (unsigned)count, 1, *fm);
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    if (ev_start && ev_stop)
        hipExtLaunchKernelGGL(k_fft4096<false>, dim3(grid), dim3(256), 0, s, ev_start, ev_stop, 0, (const float2 *)head,
                              (long)n_head, (const float2 *)in, (float2 *)out, (const float *)window,
                              (const float2 *)tw4096, (int)center_dc, (long)hop, (unsigned)count, 1, FrameMeter{});
    else
        hipLaunchKernelGGL(k_fft4096<false>, dim3(grid), dim3(256), 0, s, (const float2 *)head, (long)n_head,
                           (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw4096,
                           (int)center_dc, (long)hop, (unsigned)count, 1, FrameMeter{});
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// the 4096-bin polyphase channelizer (see k_fft4096<true>)
// ---------------------------------------------------------------------------
// Kernel 2y  k_fft_big<16384> ("k_fft16384"): window * v -> 16 384-point forward DFT in ONE pass over HBM: a workgroup of 1024 lanes per frame, 16
// values per lane, the transform of rr_fft_big.hpp (radix 16 x 16 x 16 x 4 through one 136 KiB image in LDS, one workgroup per
// CU) - the two passes of k_fft_tile move every sample through HBM twice.
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(N / 16) void k_fft_big(const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in,
                                                    float2 *__restrict__ out, const float *__restrict__ window,
                                                    const float2 *__restrict__ tw, int center_dc, long hop) {
    constexpr int T = N / 16;
    extern __shared__ __attribute__((aligned(16))) f2 fft16k_smem[];
    f2 *const img = fft16k_smem;
    const int j = threadIdx.x;
    const long base = (long)blockIdx.x * hop - n_head;
    f2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long i = base + j + T * k;
        const float2 x = (i >= 0) ? in[i] : head[n_head + i];
        const float w = window[j + T * k];
        v[k] = (f2){x.x * w, x.y * w};
    }
    BigFftLane<N> ln;
    ln.init(tw, img + (N + N / 16), j);
    big_fft<N>(v, img, ln, j, false, [] {});
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)blockIdx.x * N;
    const int rot = center_dc ? N / 2 : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v[k], dst + ((j + T * k + rot) & (N - 1)));
}

template <int N>
static int launch_fft_big_n(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count, const void *window,
                            const void *twN, bool center_dc, size_t hop) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fft_big: too many frames");
    constexpr size_t lds = (size_t)big_fft_lds_elems<N>() * sizeof(f2);
    RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(k_fft_big<N>), lds));
    hipLaunchKernelGGL(k_fft_big<N>, dim3((unsigned)count), dim3(N / 16), lds, s, (const float2 *)head, (long)n_head, (const float2 *)in,
                       (float2 *)out, (const float *)window, (const float2 *)twN, (int)center_dc, (long)hop);
    RR_HIP(hipGetLastError());
    return RR_OK;
}
int launch_fft16384(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count, const void *window,
                    const void *tw16384, bool center_dc, size_t hop) {
    return launch_fft_big_n<16384>(s, head, n_head, in, out, count, window, tw16384, center_dc, hop);
}
// 8192 points through the same transform, 512 lanes with 16 values each: the default since it measured 69 % against the 58 % of
// k_fft8192's 256 lanes with 32 values (RR_FOURIER_8K=regs)
int launch_fft8192_big(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count, const void *window,
                       const void *tw8192, bool center_dc, size_t hop) {
    return launch_fft_big_n<8192>(s, head, n_head, in, out, count, window, tw8192, center_dc, hop);
}

__global__ __launch_bounds__(256) void k_chan4096_pair(const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in,
                                                       float2 *__restrict__ out, const float *__restrict__ window,
                                                       const float2 *__restrict__ tw, unsigned count, int branches);
bool chan_pair_enabled() {  // two frames per wave / workgroup where neighbouring frames share chunks (RR_CHAN_PAIR=0: one)
    static const bool pair = [] { const char *e = std::getenv("RR_CHAN_PAIR"); return !(e && std::atoi(e) == 0); }();
    return pair;
}

int launch_chan4096(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                    const void *window, const void *tw4096, size_t hop, size_t branches) {
    if (count == 0) return RR_OK;
    if (chan_pair_enabled() && hop == 4096 && branches >= 2 && count <= 0x7fffff00ull) {
        const size_t pairs = (count + 1) / 2;
        hipLaunchKernelGGL(k_chan4096_pair, dim3((unsigned)((pairs + 63) / 64 * 64)), dim3(256), 0, s, (const float2 *)head,
                           (long)n_head, (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw4096,
                           (unsigned)count, (int)branches);
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "channelizer: too many frames");
    hipLaunchKernelGGL(k_fft4096<true>, dim3((unsigned)((count + 63) / 64 * 64)), dim3(256), 0, s, (const float2 *)head, (long)n_head,
                       (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw4096, 0, (long)hop,
                       (unsigned)count, (int)branches, FrameMeter{});
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Kernel 2s  k_fft512: window * v -> 512-point forward DFT, radix 8 x 8 x 8, one wave per frame (8 values per
// lane, wave-local exchanges through a padded 4.5 KiB image).
// ---------------------------------------------------------------------------
// the wave-local 512-point transform: in a[k] = x[l + 64 k], out a[k] = X[l + 64 k] (s1 = tw[8 (l mod 8)], s2 = tw[l])
__device__ __forceinline__ void wave_dft512(f2 (&a)[8], f2 *lds, int l, f2 s1, f2 s2) {
    dft8(a);  // pass 0 (Ns = 1): out 8 l + k
#pragma unroll
    for (int k = 0; k < 8; ++k) lds_st(lds + pad8(8 * l + k), a[k]);
    wave_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = lds_ld(lds + pad8(l + 64 * k));
    twiddle8(a, s1);  // pass 1 (Ns = 8): e^{-j 2 pi (l mod 8) k / 64}; out (l / 8) 64 + l % 8 + 8 k
    dft8(a);
    wave_sync();
    {
        const int b = (l >> 3) * 64 + (l & 7);
#pragma unroll
        for (int k = 0; k < 8; ++k) lds_st(lds + pad8(b + 8 * k), a[k]);
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = lds_ld(lds + pad8(l + 64 * k));
    twiddle8(a, s2);  // pass 2 (Ns = 64): e^{-j 2 pi l k / 512}; out l + 64 k
    dft8(a);
}

// The 512-bin channelizer at hop = 512 with FOUR neighbouring frames per wave (8 values per lane and frame): branches + 3 chunk
// reads for four frames instead of 4 branches (4 taps per branch: 7 instead of 16); chunk p goes into frame r with the window's
// segment p - r.
// (WS: the sliding segments - 152 registers, three waves per SIMD; with 2 taps per branch the four waves of the plain form are ahead: 0.199 against 0.213 ms)
template <bool WS>
__global__ __launch_bounds__(64) void k_chan512_quad(const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in,
                                                     float2 *__restrict__ out, const float *__restrict__ window,
                                                     const float2 *__restrict__ tw, unsigned count, int branches) {
    __shared__ f2 lds[512 + 64];
    const int l = threadIdx.x;
    const unsigned q = blockIdx.x / 128 * 128 + (blockIdx.x % 128 & 7) * 16 + (blockIdx.x % 128 >> 3);
    const unsigned f0 = 4 * q;
    if (f0 >= count) return;
    const int nfr = count - f0 < 4u ? (int)(count - f0) : 4;
    const long base = (long)f0 * 512 - n_head;
    f2 v[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[r][k] = (f2){0.f, 0.f};
    const int chunks = branches + nfr - 1;
    // (the window's last four segments stay in registers, segment s in slot s mod 4 - the chunk loop is unrolled by 4 -: one new
    //  segment per chunk instead of four, as k_chan1024_multi)
    [[maybe_unused]] float wseg[4][8];
    for (int p0 = 0; p0 < chunks; p0 += 4) {
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int p = p0 + pi;
            if (p >= chunks) break;
            if (WS && p < branches) {
#pragma unroll
                for (int k = 0; k < 8; ++k) wseg[pi][k] = window[512 * p + l + 64 * k];
            }
            f2 x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long i = base + 512L * p + l + 64 * k;
                const float2 t = (i >= 0) ? in[i] : head[n_head + i];
                x[k] = (f2){t.x, t.y};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int seg = p - r;
                if (seg >= 0 && seg < branches && r < nfr) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float w = WS ? wseg[(pi - r + 4) % 4][k] : window[512 * seg + l + 64 * k];
                        v[r][k] = __builtin_elementwise_fma(x[k], (f2){w, w}, v[r][k]);
                    }
                }
            }
        }
    }
    const float2 s1 = tw[8 * (l & 7)], s2 = tw[l];
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)f0 * 512;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r < nfr) {
            if (r) wave_sync();  // the previous transform's last reads are done
            wave_dft512(v[r], lds, l, (f2){s1.x, s1.y}, (f2){s2.x, s2.y});
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[r][k], dst + 512 * r + (l + 64 * k));
        }
    }
}

// (FOLD: the 512-bin polyphase channelizer - the frame is the fold of `branches` windowed chunks)
template <bool FOLD>
__global__ __launch_bounds__(64) void k_fft512(const float2 *__restrict__ head, long n_head,
                                               const float2 *__restrict__ in, float2 *__restrict__ out,
                                               const float *__restrict__ window, const float2 *__restrict__ tw,
                                               int center_dc, long hop, unsigned count, int branches) {
    __shared__ f2 lds[512 + 64];
    const int l = threadIdx.x;
    // frames dealt to the XCDs in a moving window, 16 neighbouring frames per XCD
    const unsigned fr = blockIdx.x / 128 * 128 + (blockIdx.x % 128 & 7) * 16 + (blockIdx.x % 128 >> 3);
    if (fr >= count) return;
    const long base = (long)fr * hop - n_head;
    f2 a[8];
    if constexpr (FOLD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = (f2){0.f, 0.f};
        for (int p = 0; p < branches; ++p) {
            float2 x[8];
            float w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long i = base + 512L * p + l + 64 * k;
                x[k] = (i >= 0) ? in[i] : head[n_head + i];
                w[k] = window[512 * p + l + 64 * k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_elementwise_fma((f2){x[k].x, x[k].y}, (f2){w[k], w[k]}, a[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long i = base + l + 64 * k;
            const float2 x = (i >= 0) ? in[i] : head[n_head + i];
            const float w = window[l + 64 * k];
            a[k] = (f2){x.x * w, x.y * w};
        }
    }
    const float2 s1 = tw[8 * (l & 7)], s2 = tw[l];  // tw[k] = e^{-j 2 pi k / 512}
    wave_dft512(a, lds, l, (f2){s1.x, s1.y}, (f2){s2.x, s2.y});
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)fr * 512;
    const int rot = center_dc ? 256 : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(a[k], dst + ((l + 64 * k + rot) & 511));
}

int launch_fft512(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                  const void *window, const void *tw512, bool center_dc, size_t hop) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "fft512: too many frames");
    const unsigned grid = (unsigned)((count + 127) / 128 * 128);
    hipLaunchKernelGGL(k_fft512<false>, dim3(grid), dim3(64), 0, s, (const float2 *)head, (long)n_head, (const float2 *)in,
                       (float2 *)out, (const float *)window, (const float2 *)tw512, (int)center_dc, (long)hop,
                       (unsigned)count, 1);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_chan512(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw512, size_t hop, size_t branches) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "channelizer: too many frames");
    if (chan_pair_enabled() && hop == 512 && branches >= 2) {
        const size_t quads = (count + 3) / 4;
        if (branches >= 3)
            hipLaunchKernelGGL(k_chan512_quad<true>, dim3((unsigned)((quads + 127) / 128 * 128)), dim3(64), 0, s, (const float2 *)head,
                               (long)n_head, (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw512,
                               (unsigned)count, (int)branches);
        else
            hipLaunchKernelGGL(k_chan512_quad<false>, dim3((unsigned)((quads + 127) / 128 * 128)), dim3(64), 0, s, (const float2 *)head,
                               (long)n_head, (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw512,
                               (unsigned)count, (int)branches);
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    const unsigned grid = (unsigned)((count + 127) / 128 * 128);
    hipLaunchKernelGGL(k_fft512<true>, dim3(grid), dim3(64), 0, s, (const float2 *)head, (long)n_head, (const float2 *)in,
                       (float2 *)out, (const float *)window, (const float2 *)tw512, 0, (long)hop, (unsigned)count,
                       (int)branches);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Kernel 2t  k_fft64 / k_fft128: the small chunk lengths, several frames per wave (8 x 64 or 4 x 128 points):
// 64 = radix 8 x 8 with 8 lanes per frame, 128 = radix 8 x 16 with 16 lanes per frame (the radix-16 pass on the
// lower 8 lanes of each frame).  Frames side by side only (hop = n).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_fft64(const float2 *__restrict__ in, float2 *__restrict__ out,
                                              const float *__restrict__ window, const float2 *__restrict__ tw,
                                              int center_dc, unsigned count) {
    __shared__ f2 lds[8 * 72];
    const int l = threadIdx.x, f = l >> 3, q = l & 7;
    const unsigned fr = blockIdx.x * 8 + f;
    const bool live = fr < count;
    const f2 *src = reinterpret_cast<const f2 *>(in) + (size_t)(live ? fr : 0) * 64 + q;
    f2 a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = src[8 * k] * window[q + 8 * k];
    const float2 s1 = tw[q];  // tw[k] = e^{-j 2 pi k / 64}
    dft8(a);                  // pass 0 (Ns = 1): out 8 q + k
    f2 *img = lds + 72 * f;
#pragma unroll
    for (int k = 0; k < 8; ++k) lds_st(img + pad8(8 * q + k), a[k]);
    wave_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = lds_ld(img + pad8(q + 8 * k));
    twiddle8(a, (f2){s1.x, s1.y});  // pass 1 (Ns = 8): e^{-j 2 pi q k / 64}; out q + 8 k
    dft8(a);
    if (!live) return;
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)fr * 64;
    const int rot = center_dc ? 32 : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[(q + 8 * k + rot) & 63] = a[k];
}

__global__ __launch_bounds__(64) void k_fft128(const float2 *__restrict__ in, float2 *__restrict__ out,
                                               const float *__restrict__ window, const float2 *__restrict__ tw,
                                               int center_dc, unsigned count) {
    __shared__ f2 lds[4 * 144];
    const int l = threadIdx.x, f = l >> 4, q = l & 15;
    const unsigned fr = blockIdx.x * 4 + f;
    const bool live = fr < count;
    const f2 *src = reinterpret_cast<const f2 *>(in) + (size_t)(live ? fr : 0) * 128 + q;
    f2 a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = src[16 * k] * window[q + 16 * k];
    const float2 s1 = tw[q & 7];  // tw[k] = e^{-j 2 pi k / 128}
    dft8(a);                      // pass 0 (Ns = 1): butterflies q over x[q + 16 k]; out 8 q + k
    f2 *img = lds + 144 * f;
#pragma unroll
    for (int k = 0; k < 8; ++k) lds_st(img + pad8(8 * q + k), a[k]);
    wave_sync();
    // pass 1 (Ns = 8, radix 16): butterflies j < 8 over y[j + 8 k], k < 16; twiddle tw[j]^k; out j + 8 k
    f2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds_ld(img + pad8((q & 7) + 8 * k));
    apply_twiddle_powers(v, (f2){s1.x, s1.y});
    dft16(v);
    if (!live || q >= 8) return;
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)fr * 128;
    const int rot = center_dc ? 64 : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) dst[(q + 8 * k + rot) & 127] = v[k];
}

int launch_fft_small(hipStream_t s, const void *in, void *out, size_t n, size_t count, const void *window, const void *tw,
                     bool center_dc) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "fft: too many frames");
    if (n == 64)
        hipLaunchKernelGGL(k_fft64, dim3((unsigned)((count + 7) / 8)), dim3(64), 0, s, (const float2 *)in, (float2 *)out,
                           (const float *)window, (const float2 *)tw, (int)center_dc, (unsigned)count);
    else
        hipLaunchKernelGGL(k_fft128, dim3((unsigned)((count + 3) / 4)), dim3(64), 0, s, (const float2 *)in, (float2 *)out,
                           (const float *)window, (const float2 *)tw, (int)center_dc, (unsigned)count);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Kernel 2h  k_fft2048: window * v -> 2048-point forward DFT, radix 16 x 16 x 8 (Stockham autosort through one
// padded 17 KiB LDS image), a workgroup of 128 lanes per frame, 16 values per lane: the 2048-point sibling of
// k_fft4096 (analysis.rs:105-115; rr_stft with 2048-sample spans).  The window table carries a packed copy
// behind its 2048 entries (wp[16 t + k] = w[t + 128 k], rr_fourier::prepare).
// ---------------------------------------------------------------------------
// the 2048-point transform of a 128-lane workgroup (radix 16 x 16 x 8 through one padded image) with the store of its result:
// in v[k] = x[t + 128 k]; X[j + 256 k] goes to dst[(j + 256 k + rot) mod 2048].  The caller synchronises before the image is reused.
__device__ __forceinline__ void fft2048_store(f2 (&v)[16], f2 *lds, const float2 *__restrict__ tw, int t, f2 *dst, int rot) {
    // twiddle seeds: pass 1 e^{-j 2 pi (t mod 16) / 256} = tw[8 (t mod 16)]; pass 2 tw[t], tw[t + 128]  (tw[k] = e^{-j 2 pi k / 2048})
    const float2 s1 = tw[8 * (t & 15)], s2a = tw[t], s2b = tw[t + 128];
    // pass 0 (Ns = 1, radix 16): butterflies t over x[t + 128 k]; out 16 t + k
    dft16(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds_st(lds + pad16(16 * t + k), v[k]);
    __syncthreads();
    // pass 1 (Ns = 16, radix 16): in y[t + 128 k]; out (t / 16) 256 + t % 16 + 16 k
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds_ld(lds + pad16(t + 128 * k));
    apply_twiddle_powers(v, (f2){s1.x, s1.y});
    dft16(v);
    __syncthreads();
    {
        const int b = (t >> 4) * 256 + (t & 15);
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(lds + pad16(b + 16 * k), v[k]);
    }
    __syncthreads();
    // pass 2 (Ns = 256, radix 8): butterflies j = t and t + 128 over z[j + 256 k]; out X[j + 256 k]
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
        const int j = t + 128 * sidx;
        f2 a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = lds_ld(lds + pad16(j + 256 * k));
        const float2 sw = sidx ? s2b : s2a;
        const f2 w1 = {sw.x, sw.y};
        const f2 w2 = cmulf(w1, w1), w3 = cmulf(w2, w1), w4 = cmulf(w2, w2);
        a[1] = cmulf(a[1], w1);
        a[2] = cmulf(a[2], w2);
        a[3] = cmulf(a[3], w3);
        a[4] = cmulf(a[4], w4);
        a[5] = cmulf(a[5], cmulf(w4, w1));
        a[6] = cmulf(a[6], cmulf(w4, w2));
        a[7] = cmulf(a[7], cmulf(w4, w3));
        dft8(a);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(a[k], dst + ((j + 256 * k + rot) & 2047));
    }
}

// (FOLD: the 2048-bin polyphase channelizer - the frame is the fold of `branches` windowed chunks, window: plain values)
template <bool FOLD>
__global__ __launch_bounds__(128) void k_fft2048(const float2 *__restrict__ head, long n_head,
                                                 const float2 *__restrict__ in, float2 *__restrict__ out,
                                                 const float *__restrict__ window, const float2 *__restrict__ tw,
                                                 int center_dc, long hop, int branches, unsigned count) {
    __shared__ f2 lds[2048 + 128];
    const int t = threadIdx.x;
    // FOLD: neighbouring frames (which share branches - 1 chunks) go to one XCD, 16 at a time (as k_fft4096<true>)
    const unsigned fr = FOLD ? blockIdx.x / 128 * 128 + (blockIdx.x % 128 & 7) * 16 + (blockIdx.x % 128 >> 3) : blockIdx.x;
    if (fr >= count) return;
    const long base = (long)fr * hop - n_head;
    f2 v[16];
    if constexpr (FOLD) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (f2){0.f, 0.f};
        for (int p = 0; p < branches; ++p) {
            const long bp = base + 2048L * p + t;
            float2 x[16];
            float w[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long i = bp + 128 * k;
                x[k] = (i >= 0) ? in[i] : head[n_head + i];
                w[k] = window[2048 * p + t + 128 * k];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = __builtin_elementwise_fma((f2){x[k].x, x[k].y}, (f2){w[k], w[k]}, v[k]);
        }
    } else {
        float wv[16];
        const float4 *wp = reinterpret_cast<const float4 *>(window + 2048) + 4 * t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w4 = wp[q];
            wv[4 * q] = w4.x;
            wv[4 * q + 1] = w4.y;
            wv[4 * q + 2] = w4.z;
            wv[4 * q + 3] = w4.w;
        }
        if (base >= 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = reinterpret_cast<const f2 *>(in + base + t)[128 * k] * wv[k];
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long i = base + t + 128 * k;
                const float2 x = (i >= 0) ? in[i] : head[n_head + i];
                v[k] = (f2){x.x * wv[k], x.y * wv[k]};
            }
        }
    }
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)fr * 2048;
    fft2048_store(v, lds, tw, t, dst, center_dc ? 1024 : 0);
}

// The 2048-bin channelizer at hop = 2048 with TWO neighbouring frames per workgroup (as k_chan4096_pair)
__global__ __launch_bounds__(128) void k_chan2048_pair(const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in,
                                                       float2 *__restrict__ out, const float *__restrict__ window,
                                                       const float2 *__restrict__ tw, unsigned count, int branches) {
    __shared__ f2 lds[2048 + 128];
    const int t = threadIdx.x;
    const unsigned q = blockIdx.x / 128 * 128 + (blockIdx.x % 128 & 7) * 16 + (blockIdx.x % 128 >> 3);
    const unsigned fa = 2 * q;
    if (fa >= count) return;
    const bool has_b = fa + 1 < count;
    const long base = (long)fa * 2048 - n_head;
    f2 va[16], vb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) va[k] = vb[k] = (f2){0.f, 0.f};
    const int chunks = branches + (has_b ? 1 : 0);
    // (a segment of the window serves frame a at chunk p and frame b at chunk p + 1: it stays in registers in between - the loop
    //  is unrolled by 2, segment s in slot s mod 2 -, 32 instead of 48 vector-memory instructions per chunk)
    float wseg[2][16];
    for (int p0 = 0; p0 < chunks; p0 += 2) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int p = p0 + pi;
            if (p >= chunks) break;
            if (p < branches) {
#pragma unroll
                for (int k = 0; k < 16; ++k) wseg[pi][k] = window[2048 * p + t + 128 * k];
            }
            const long bp = base + 2048L * p + t;
            f2 x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long i = bp + 128 * k;
                const float2 xx = (i >= 0) ? in[i] : head[n_head + i];
                x[k] = (f2){xx.x, xx.y};
            }
            if (p < branches) {
#pragma unroll
                for (int k = 0; k < 16; ++k) va[k] = __builtin_elementwise_fma(x[k], (f2){wseg[pi][k], wseg[pi][k]}, va[k]);
            }
            if (p >= 1) {
#pragma unroll
                for (int k = 0; k < 16; ++k) vb[k] = __builtin_elementwise_fma(x[k], (f2){wseg[pi ^ 1][k], wseg[pi ^ 1][k]}, vb[k]);
            }
        }
    }
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)fa * 2048;
    fft2048_store(va, lds, tw, t, dst, 0);
    if (!has_b) return;  // (uniform over the workgroup)
    __syncthreads();  // the first transform's last pass has been read
    fft2048_store(vb, lds, tw, t, dst + 2048, 0);
}

int launch_fft2048(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw2048, bool center_dc, size_t hop) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fft2048: too many frames");
    hipLaunchKernelGGL(k_fft2048<false>, dim3((unsigned)count), dim3(128), 0, s, (const float2 *)head, (long)n_head,
                       (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw2048, (int)center_dc,
                       (long)hop, 1, (unsigned)count);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_chan2048(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                    const void *window, const void *tw2048, size_t hop, size_t branches) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "channelizer: too many frames");
    if (chan_pair_enabled() && hop == 2048 && branches >= 2) {
        const size_t pairs = (count + 1) / 2;
        hipLaunchKernelGGL(k_chan2048_pair, dim3((unsigned)((pairs + 127) / 128 * 128)), dim3(128), 0, s, (const float2 *)head,
                           (long)n_head, (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw2048,
                           (unsigned)count, (int)branches);
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    hipLaunchKernelGGL(k_fft2048<true>, dim3((unsigned)((count + 127) / 128 * 128)), dim3(128), 0, s, (const float2 *)head,
                       (long)n_head, (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw2048, 0, (long)hop,
                       (int)branches, (unsigned)count);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Kernel 2x  k_fft8192: window * v -> 8192-point forward DFT, radix 16 x 16 x 32 (Stockham autosort through one
// padded 68 KiB LDS image), a workgroup of 256 lanes per frame, 32 values per lane.  The radix-32 butterfly
// of the last pass is two 16-point DFTs (even / odd inputs) and 16 radix-2 butterflies with W_32^m.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fft8192(const float2 *__restrict__ head, long n_head,
                                                 const float2 *__restrict__ in, float2 *__restrict__ out,
                                                 const float *__restrict__ window, const float2 *__restrict__ tw,
                                                 int center_dc, long hop) {
    extern __shared__ __attribute__((aligned(16))) char fft8192_smem[];
    f2 *lds = reinterpret_cast<f2 *>(fft8192_smem);  // 8192 + 512 elements
    const int t = threadIdx.x;
    const long base = (long)blockIdx.x * hop - n_head;
    f2 v[2][16];
    // pass 0 (Ns = 1, radix 16): butterflies j = t, t + 256 over x[j + 512 k]; out 16 j + k
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = t + 256 * h;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long i = base + j + 512 * k;
            const float2 x = (i >= 0) ? in[i] : head[n_head + i];
            const float w = window[j + 512 * k];
            v[h][k] = (f2){x.x * w, x.y * w};
        }
        dft16(v[h]);
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(lds + pad16(16 * j + k), v[h][k]);
    }
    __syncthreads();
    // pass 1 (Ns = 16, radix 16): in y[j + 512 k]; twiddle e^{-j 2 pi (j mod 16) k / 256} = tw[32 (j mod 16)]^k;
    // out (j / 16) 256 + j % 16 + 16 k
    {
        const float2 s1 = tw[32 * (t & 15)];  // (t + 256) mod 16 = t mod 16: one seed for both butterflies
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = t + 256 * h;
#pragma unroll
            for (int k = 0; k < 16; ++k) v[h][k] = lds_ld(lds + pad16(j + 512 * k));
            apply_twiddle_powers(v[h], (f2){s1.x, s1.y});
            dft16(v[h]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = t + 256 * h;
        const int b = (j >> 4) * 256 + (j & 15);
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(lds + pad16(b + 16 * k), v[h][k]);
    }
    __syncthreads();
    // pass 2 (Ns = 256, radix 32): butterfly t over z[t + 256 k], k < 32; twiddle tw[t]^k; out X[t + 256 k]
    // even inputs k = 2 a -> v[0][a] * (w^2)^a; odd inputs k = 2 a + 1 -> v[1][a] * w * (w^2)^a
    {
        const float2 s2 = tw[t];
        const f2 w = {s2.x, s2.y};
        const f2 w2 = cmulf(w, w);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            v[0][a] = lds_ld(lds + pad16(t + 256 * (2 * a)));
            v[1][a] = cmulf(lds[pad16(t + 256 * (2 * a + 1))], w);
        }
        apply_twiddle_powers(v[0], w2);
        apply_twiddle_powers(v[1], w2);
        dft16(v[0]);  // E[m]
        dft16(v[1]);  // O[m]
        f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)blockIdx.x * 8192;
        const int rot = center_dc ? 4096 : 0;
        // X[m] = E[m] + W_32^m O[m], X[m + 16] = E[m] - W_32^m O[m];  W_32^m = e^{-j 2 pi m / 32} = tw[256 m] (a scalar read)
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 c = tw[256 * m];
            const f2 o = cmulf(v[1][m], (f2){c.x, c.y});
            __builtin_nontemporal_store(v[0][m] + o, dst + ((t + 256 * m + rot) & 8191));
            __builtin_nontemporal_store(v[0][m] - o, dst + ((t + 256 * (m + 16) + rot) & 8191));
        }
    }
}

int launch_fft8192(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw8192, bool center_dc, size_t hop) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fft8192: too many frames");
    const size_t lds = (8192 + 512) * sizeof(float2);
    RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(k_fft8192), lds));
    hipLaunchKernelGGL(k_fft8192, dim3((unsigned)count), dim3(256), lds, s, (const float2 *)head, (long)n_head,
                       (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw8192, (int)center_dc,
                       (long)hop);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

__global__ __launch_bounds__(256) void k_chan4096_pair(const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in,
                                                       float2 *__restrict__ out, const float *__restrict__ window,
                                                       const float2 *__restrict__ tw, unsigned count, int branches) {
    __shared__ f2 lds[4096 + 256];
    const int j = threadIdx.x;
    // pairs dealt to the XCDs 8 at a time (neighbouring pairs share chunks, too)
    const unsigned q = blockIdx.x / 64 * 64 + (blockIdx.x % 64 & 7) * 8 + (blockIdx.x % 64 >> 3);
    const unsigned fa = 2 * q;
    if (fa >= count) return;
    const bool has_b = fa + 1 < count;
    const long base = (long)fa * 4096 - n_head;
    f2 va[16], vb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) va[k] = vb[k] = (f2){0.f, 0.f};
    const int chunks = branches + (has_b ? 1 : 0);
    // (a segment of the window serves frame a at chunk p and frame b at chunk p + 1: it stays in registers in between - the loop
    //  is unrolled by 2, segment s in slot s mod 2 -, 32 instead of 48 vector-memory instructions per chunk)
    float wseg[2][16];
    for (int p0 = 0; p0 < chunks; p0 += 2) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int p = p0 + pi;
            if (p >= chunks) break;
            if (p < branches) {
#pragma unroll
                for (int k = 0; k < 16; ++k) wseg[pi][k] = window[4096 * p + j + 256 * k];
            }
            const long bp = base + 4096L * p + j;
            f2 x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long i = bp + 256 * k;
                const float2 xx = (i >= 0) ? in[i] : head[n_head + i];
                x[k] = (f2){xx.x, xx.y};
            }
            if (p < branches) {
#pragma unroll
                for (int k = 0; k < 16; ++k) va[k] = __builtin_elementwise_fma(x[k], (f2){wseg[pi][k], wseg[pi][k]}, va[k]);
            }
            if (p >= 1) {
#pragma unroll
                for (int k = 0; k < 16; ++k) vb[k] = __builtin_elementwise_fma(x[k], (f2){wseg[pi ^ 1][k], wseg[pi ^ 1][k]}, vb[k]);
            }
        }
    }
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)fa * 4096;
    fft4096_regs(va, lds, tw, j);
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(va[k], dst + (j + 256 * k));
    if (!has_b) return;  // (uniform over the workgroup)
    __syncthreads();  // the first transform's last pass has been read
    fft4096_regs(vb, lds, tw, j);
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(vb[k], dst + 4096 + (j + 256 * k));
}

// ---------------------------------------------------------------------------
// Kernel 2r  k_stft4096<SH>: overlapping 4096-point frames at a hop of 256 SH samples (the Overlapper in front of
// the Fourier block, chunks.rs:179-271, with P = 16 / SH chunks per span), a workgroup per RUN of neighbouring frames.
// Lane j holds x[j + 256 k]: the next frame's samples are the current ones moved down SH registers plus SH new
// loads, which are requested before the current frame's transform starts.  k_fft4096 at hop < 4096 reads every
// sample 16 / SH times (from L2) and pays a full load latency per frame: it makes 16384 frames in 0.165 ms whatever
// the hop, i.e. 51 % of the 8 + 32 B per input sample of the 1024 x 4 case.
// ---------------------------------------------------------------------------
constexpr unsigned kStftWin = 4;  // neighbouring runs per XCD (they share 4096 - hop samples)
template <int SH, bool METER = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_stft4096(const float2 *__restrict__ head, long n_head,
                                                  const float2 *__restrict__ in, float2 *__restrict__ out,
                                                  const float *__restrict__ window, const float2 *__restrict__ tw,
                                                  int center_dc, unsigned count, unsigned R, unsigned nruns, FrameMeter fm = FrameMeter{}) {
    __shared__ f2 lds[4096 + 256];
    const int j = threadIdx.x;
    constexpr unsigned G = kStftWin;
    const unsigned grp = blockIdx.x / (8 * G), rem = blockIdx.x % (8 * G);
    const unsigned run = grp * 8 * G + (rem & 7) * G + (rem >> 3);
    if (run >= nruns) return;
    const unsigned f0 = run * R;
    const unsigned nf = count - f0 < R ? count - f0 : R;
    constexpr long hop = 256L * SH;
    const long base0 = (long)f0 * hop - n_head;  // index into `in` of the run's first sample
    auto ld = [&](long i) -> f2 {
        const float2 *p = i >= 0 ? in + i : head + (n_head + i);
        return *reinterpret_cast<const f2 *>(p);
    };
    float wv[16];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(window + 4096) + 4 * j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w4 = wp[q];
            wv[4 * q] = w4.x;
            wv[4 * q + 1] = w4.y;
            wv[4 * q + 2] = w4.z;
            wv[4 * q + 3] = w4.w;
        }
    }
    f2 xr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xr[k] = ld(base0 + j + 256 * k);
    const int rot = center_dc ? 2048 : 0;
#pragma unroll 1
    for (unsigned i = 0; i < nf; ++i) {
        // (METER: the lane index is made opaque per frame - hoisted out of the loop, the transform's and the epilogue's lane
        //  addresses together no longer fit the 168 registers of three waves per SIMD and came back from scratch every frame)
        int jl = j;
        if constexpr (METER) asm volatile("" : "+v"(jl));
        f2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = xr[k] * wv[k];
        // the window moves on, and the next frame's new samples are requested before this frame's transform starts
        // (the last frame of the run asks for its own last ones again: always valid)
#pragma unroll
        for (int k = 0; k + SH < 16; ++k) xr[k] = xr[k + SH];
        const long nb = base0 + (long)(i + 1 < nf ? i + 1 : i) * hop;
#pragma unroll
        for (int q = 0; q < SH; ++q) xr[16 - SH + q] = ld(nb + j + 256 * (16 - SH + q));
        if (i) __syncthreads();  // the previous frame's last pass has been read
        fft4096_regs(v, lds, tw, jl);
        if (!METER || fm.store) {
            float2 *dst = out + (size_t)(f0 + i) * 4096;
#pragma unroll
            for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<f2 *>(dst) + ((jl + 256 * k + rot) & 4095));
        }
        if constexpr (METER) {  // (the next frame's first barrier separates this frame's last scratch reads from its image stores)
            double total;
            const int tw_ = (int)((f0 + i) & 3u);
            const double bw = frame4096_bandwidth(v, lds, jl, rot, fm.double_percentile, fm.sample_rate, &total, tw_);
            if (jl == 64 * tw_) {
                fm.bw[f0 + i] = bw;
                if (fm.energy) fm.energy[f0 + i] = total;
            }
        }
    }
}

// frames per run: as few rounds of the grid as possible at 3 workgroups per CU, and long runs within that
static unsigned stft_run_length(size_t count) {
    const size_t slots = 256 * 3;
    size_t best = 1, best_cost = ~size_t(0);
    for (size_t r = 1; r <= 16; ++r) {
        const size_t runs = (count + r - 1) / r, rounds = (runs + slots - 1) / slots;
        const size_t cost = rounds * (r + 2);  // a run costs its frames plus ~2 frames' worth of start-up
        if (cost < best_cost || (cost == best_cost && r > best)) {
            best = r;
            best_cost = cost;
        }
    }
    return (unsigned)best;
}

bool stft4096_supported(size_t hop) { return hop == 256 || hop == 512 || hop == 1024 || hop == 2048; }
// (hop 4096 = frames side by side, i.e. only the request one frame ahead at 3 workgroups per CU, measured slower than
//  k_fft4096: 0.220 against 0.175 ms per 2^26 samples, chain step 0.182 against 0.174)

int launch_stft4096(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                    const void *window, const void *tw4096, bool center_dc, size_t hop, const FrameMeter *fm) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "stft4096: too many frames");
    const unsigned R = stft_run_length(count);
    const unsigned nruns = (unsigned)((count + R - 1) / R);
    const unsigned grid = (nruns + 8 * kStftWin - 1) / (8 * kStftWin) * (8 * kStftWin);
#define RR_STFT_CASE(SH)                                                                                                   \
    case 256 * SH:                                                                                                         \
        if (fm)                                                                                                            \
            hipLaunchKernelGGL((k_stft4096<SH, true>), dim3(grid), dim3(256), 0, s, (const float2 *)head, (long)n_head,   \
                               (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw4096,           \
                               (int)center_dc, (unsigned)count, R, nruns, *fm);                                            \
        else                                                                                                               \
            hipLaunchKernelGGL((k_stft4096<SH, false>), dim3(grid), dim3(256), 0, s, (const float2 *)head, (long)n_head,  \
                               (const float2 *)in, (float2 *)out, (const float *)window, (const float2 *)tw4096,           \
                               (int)center_dc, (unsigned)count, R, nruns, FrameMeter{});                                    \
        break;
    switch (hop) {
        RR_STFT_CASE(1)
        RR_STFT_CASE(2)
        RR_STFT_CASE(4)
        RR_STFT_CASE(8)
    default: RR_FAIL(RR_ERR_BAD_ARG, "stft4096: hop %zu not instantiated", hop);
    }
#undef RR_STFT_CASE
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_fft4096_bank(hipStream_t s, const BankTable &d_chan, size_t channels, size_t n_head, size_t count, const void *window,
                        const void *tw4096, bool center_dc) {
    if (count == 0 || channels == 0) return RR_OK;
    if (count > 0x7fffffffull || channels > kBankGroup) RR_FAIL(RR_ERR_BAD_ARG, "fft4096 bank: too many frames or channels");
    hipLaunchKernelGGL(k_fft4096_bank, dim3((unsigned)count, (unsigned)channels), dim3(256), 0, s, d_chan, (long)n_head,
                       (const float *)window, (const float2 *)tw4096, (int)center_dc, (unsigned)count);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Kernel 2w  k_fft1024: window * v -> 1024-point forward DFT with one wave per frame (analysis.rs:105-115
// for chunks of 1024): the forward network of k_filter_wave on the windowed samples.  Frames come from
// [ head | in ] at distance `hop` (the overlapped analysis of rr_stft), the twiddle table carries the lane
// seeds behind its 1024 entries (rr_fourier::prepare).
// ---------------------------------------------------------------------------
// FOLD: the polyphase channelizer with 1024 bins (BASELINE configs[2] at another size): the frame is the fold of
// `branches` windowed chunks, v[i] = sum_p w[i + 1024 p] x[base + i + 1024 p], then the same transform.
template <bool FOLD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fft1024(
    const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in, float2 *__restrict__ out,
    const float *__restrict__ window, const float2 *__restrict__ tw, int center_dc, long hop, unsigned count,
    int branches) {
    __shared__ __attribute__((aligned(16))) f2 lds[kWaveLds];
    const int l = threadIdx.x;
    // frames dealt to the XCDs in a moving window, 16 neighbouring frames per XCD
    const unsigned fr = blockIdx.x / 128 * 128 + (blockIdx.x % 128 & 7) * 16 + (blockIdx.x % 128 >> 3);
    if (fr >= count) return;
    const long base = (long)fr * hop - n_head;
    f2 v[16];
    if constexpr (FOLD) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (f2){0.f, 0.f};
        for (int p = 0; p < branches; ++p) {
            const long bp = base + 1024L * p;
            if (bp >= 0) {
                const f4u *src = reinterpret_cast<const f4u *>(in + bp) + l;
                f4u x[8];
                float2 w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    x[k] = *(src + 64 * k);
                    w[k] = *reinterpret_cast<const float2 *>(window + 1024 * p + 2 * l + 128 * k);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[2 * k] = __builtin_elementwise_fma((f2){x[k].x, x[k].y}, (f2){w[k].x, w[k].x}, v[2 * k]);
                    v[2 * k + 1] = __builtin_elementwise_fma((f2){x[k].z, x[k].w}, (f2){w[k].y, w[k].y}, v[2 * k + 1]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const long i = bp + 2 * l + j + 128 * k;
                        const float2 x = (i >= 0) ? in[i] : head[n_head + i];
                        const float w = window[1024 * p + 2 * l + j + 128 * k];
                        v[2 * k + j] = __builtin_elementwise_fma((f2){x.x, x.y}, (f2){w, w}, v[2 * k + j]);
                    }
            }
        }
    } else if (base >= 0) {
        const f4u *src = reinterpret_cast<const f4u *>(in + base) + l;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f4u x = hop >= 1024 ? __builtin_nontemporal_load(src + 64 * k) : *(src + 64 * k);
            const float2 w = *reinterpret_cast<const float2 *>(window + 2 * l + 128 * k);
            v[2 * k] = (f2){x.x * w.x, x.y * w.x};
            v[2 * k + 1] = (f2){x.z * w.y, x.w * w.y};
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long i = base + 2 * l + j + 128 * k;
                const float2 x = (i >= 0) ? in[i] : head[n_head + i];
                const float w = window[2 * l + j + 128 * k];
                v[2 * k + j] = (f2){x.x * w, x.y * w};
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
    f2 X[16];
    wave_dft1024(v, X, lds, l, t_p1, t_p2, [] {});
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)fr * 1024;
    const int rot = center_dc ? 512 : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(X[k], dst + ((l + 64 * k + rot) & 1023));
}

// The 1024-bin channelizer at hop = 1024 with R neighbouring frames per wave: frames R q .. R q + R - 1 share most of their chunks,
// so the wave reads branches + R - 1 chunks for R frames instead of R branches (8 taps per branch, R = 2: 9 instead of 16) -
// k_fft1024<true> is bound by those reads (every frame re-reads its chunks from L2).  Chunk p goes into frame r with the window's
// segment p - r.
template <int R>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(R == 2 ? 4 : 2, R == 2 ? 4 : (R == 3 ? 3 : 2)))) void k_chan1024_multi(
    const float2 *__restrict__ head, long n_head, const float2 *__restrict__ in, float2 *__restrict__ out,
    const float *__restrict__ window, const float2 *__restrict__ tw, unsigned count, int branches) {
    __shared__ __attribute__((aligned(16))) f2 lds[kWaveLds];
    const int l = threadIdx.x;
    // runs dealt to the XCDs in a moving window, 16 neighbouring runs per XCD
    const unsigned q = blockIdx.x / 128 * 128 + (blockIdx.x % 128 & 7) * 16 + (blockIdx.x % 128 >> 3);
    const unsigned f0 = R * q;
    if (f0 >= count) return;
    const int nfr = count - f0 < (unsigned)R ? (int)(count - f0) : R;
    const long base = (long)f0 * 1024 - n_head;
    f2 v[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) v[r][k] = (f2){0.f, 0.f};
    const int chunks = branches + nfr - 1;
    // The window's segment s is used by chunk s + r of frame r: the lane keeps the last R segments in registers (segment s in slot
    // s mod R - the loop over the chunks is unrolled by R, so the slots are register names) and reads ONE new segment per chunk
    // instead of R (8 taps per branch, R = 4: 16 instead of 40 vector-memory instructions per chunk).
#ifndef RR_V_CHAN1024_WSLIDE
#define RR_V_CHAN1024_WSLIDE 1
#endif
    // (R = 2 at four waves per SIMD has no room for the 32 registers: 22 spilled dwords, 2 taps per branch 0.203 -> 0.249 ms; R = 4, 8 taps: 0.328 -> 0.305)
    constexpr bool WS = RR_V_CHAN1024_WSLIDE && R >= 3;
    [[maybe_unused]] float2 wseg[R][8];
    for (int p0 = 0; p0 < chunks; p0 += R) {
#pragma unroll
        for (int pi = 0; pi < R; ++pi) {
            const int p = p0 + pi;
            if (p >= chunks) break;
            const long bp = base + 1024L * p;
            if (WS && p < branches) {
#pragma unroll
                for (int k = 0; k < 8; ++k) wseg[pi][k] = *reinterpret_cast<const float2 *>(window + 1024 * p + 2 * l + 128 * k);
            }
            f2 x[16];
            if (bp >= 0) {
                const f4u *src = reinterpret_cast<const f4u *>(in + bp) + l;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f4u t = *(src + 64 * k);
                    x[2 * k] = (f2){t.x, t.y};
                    x[2 * k + 1] = (f2){t.z, t.w};
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const long i = bp + 2 * l + j + 128 * k;
                        const float2 t = (i >= 0) ? in[i] : head[n_head + i];
                        x[2 * k + j] = (f2){t.x, t.y};
                    }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int seg = p - r;
                if (seg >= 0 && seg < branches && r < nfr) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float2 w = WS ? wseg[(pi - r + R) % R][k]
                                                              : *reinterpret_cast<const float2 *>(window + 1024 * seg + 2 * l + 128 * k);
                        v[r][2 * k] = __builtin_elementwise_fma(x[2 * k], (f2){w.x, w.x}, v[r][2 * k]);
                        v[r][2 * k + 1] = __builtin_elementwise_fma(x[2 * k + 1], (f2){w.y, w.y}, v[r][2 * k + 1]);
                    }
                }
            }
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
    f2 *dst = reinterpret_cast<f2 *>(out) + (size_t)f0 * 1024;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r < nfr) {
            if (r) wave_sync();  // the previous transform's last reads are done
            f2 X[16];
            wave_dft1024(v[r], X, lds, l, t_p1, t_p2, [] {});
#pragma unroll
            for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(X[k], dst + 1024 * r + (l + 64 * k));
        }
    }
}

int launch_fft1024(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw1024, bool center_dc, size_t hop) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "fft1024: too many frames");
    const unsigned grid = (unsigned)((count + 127) / 128 * 128);
    hipLaunchKernelGGL(k_fft1024<false>, dim3(grid), dim3(64), 0, s, (const float2 *)head, (long)n_head, (const float2 *)in,
                       (float2 *)out, (const float *)window, (const float2 *)tw1024, (int)center_dc, (long)hop,
                       (unsigned)count, 1);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// the 1024-bin polyphase channelizer: frame f = DFT_1024 of the fold of `branches` windowed chunks starting hop f
// samples behind the start of [ head | in ]; window: 1024 branches values; tw1024 with the lane seeds
int launch_chan1024(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                    const void *window, const void *tw1024, size_t hop, size_t branches) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffff00ull) RR_FAIL(RR_ERR_BAD_ARG, "channelizer: too many frames");
    // critically sampled with at least two taps per branch: two frames per wave (RR_CHAN_PAIR=0 keeps one)
    if (chan_pair_enabled() && hop == 1024 && branches >= 2) {
        // frames per wave: 8 taps per branch, per 2^26 samples: 2 frames 0.368 ms, 3 frames 0.354, 4 frames 0.336 (one frame: 0.471);
        // with the window's segments kept in registers (R >= 3): 2 / 4 / 6 / 8 / 16 taps per branch 0.200 / 0.233 / 0.270 / 0.309 / 0.447 ms for three frames,
        // 0.224 / 0.246 / 0.274 / 0.304 / 0.421 for four, 0.210 / 0.266 for two: four from 8 taps per branch on, three below (RR_CHAN1024_RUN = 2 / 3 / 4 overrides)
        static const int runlen = [] { const char *e = std::getenv("RR_CHAN1024_RUN"); return e ? std::atoi(e) : 0; }();
        const size_t R = runlen == 3 ? 3 : runlen == 4 ? 4 : runlen == 2 ? 2 : (branches >= 8 ? 4 : 3);
        const size_t runs = (count + R - 1) / R;
        const unsigned g2 = (unsigned)((runs + 127) / 128 * 128);
#define RR_CH1024(RR_)                                                                                                      \
    hipLaunchKernelGGL(k_chan1024_multi<RR_>, dim3(g2), dim3(64), 0, s, (const float2 *)head, (long)n_head, (const float2 *)in, \
                       (float2 *)out, (const float *)window, (const float2 *)tw1024, (unsigned)count, (int)branches)
        if (R == 3) RR_CH1024(3);
        else if (R == 4) RR_CH1024(4);
        else RR_CH1024(2);
#undef RR_CH1024
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    const unsigned grid = (unsigned)((count + 127) / 128 * 128);
    hipLaunchKernelGGL(k_fft1024<true>, dim3(grid), dim3(64), 0, s, (const float2 *)head, (long)n_head, (const float2 *)in,
                       (float2 *)out, (const float *)window, (const float2 *)tw1024, 0, (long)hop, (unsigned)count,
                       (int)branches);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// the six twiddle seeds per lane of the wave-level 1024-point transforms, appended behind e^{-j 2 pi k / 1024}
// (3 entries of two twiddles per lane; k_ols_wave, k_filter_wave, k_fft1024)
void append_wave1024_seeds(std::vector<float> &twb) {
    const size_t N = 1024;
    auto twv = [&](size_t i, float *dst) {
        dst[0] = twb[2 * i];
        dst[1] = twb[2 * i + 1];
    };
    twb.resize(2 * (N + 2 * 11 * 64));
    for (size_t l = 0; l < 64; ++l) {
        float *e0 = &twb[2 * N + 4 * l], *e1 = e0 + 4 * 64, *e2 = e1 + 4 * 64, *e3 = e2 + 4 * 64, *e4 = e3 + 4 * 64,
              *e5 = e4 + 4 * 64, *e6 = e5 + 4 * 64, *e7 = e6 + 4 * 64, *e8 = e7 + 4 * 64, *e9 = e8 + 4 * 64, *e10 = e9 + 4 * 64;
        twv(8 * (l & 7), e0);         // pass 1
        twv(l, e0 + 2);               // pass 2, m = 0
        twv(l + 64, e1);              // pass 2, m = 1
        twv(64 * (l & 3), e1 + 2);    // inverse pass 1
        twv(16 * (l & 15), e2);       // inverse pass 2
        twv(4 * l, e2 + 2);           // inverse pass 3
        twv(128 * (l & 1), e3);       // k_ols_wave<8>: inverse DFT_128, passes 1 .. 3
        twv(32 * (l & 7), e3 + 2);
        twv(8 * (l & 31), e4);
        twv(16 * (l & 7), e4 + 2);    // k_ols_wave<2>: DFT_512, passes 1 and 2
        twv(2 * l, e5);
        e5[2] = e5[3] = 0.f;
        twv(4 * (l >> 1), e6);        // k_ols_wave<4, POLY>: W_256^(l >> 1), W_32^(l >> 3), then the inverse's three
        twv(32 * (l >> 3), e6 + 2);
        twv(64 * (l & 3), e7);
        twv(16 * (l & 15), e7 + 2);
        twv(4 * l, e8);
        e8[2] = e8[3] = 0.f;
        twv(8 * (l >> 2), e9);        // k_ols_wave<8, POLY>: W_128^(l >> 2), W_16^(l >> 3)
        twv(64 * (l >> 3), e9 + 2);
        twv(2 * l, e10);              // k_ols_wave<2, POLY>: W_512^l, W_64^(l >> 3)
        twv(16 * (l >> 3), e10 + 2);
    }
}


}  // namespace rr
