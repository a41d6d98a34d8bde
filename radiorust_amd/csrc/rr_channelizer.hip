// rr_channelizer.hip — k_channelizer256: the 256-bin polyphase channelizer (BASELINE configs[2]), Complex<f32>, a wave per run
// of frames, the sliding window of chunks and the lane's window values in registers (critically sampled and oversampled 2 x / 4 x).
// The channelizers with 512 .. 4096 bins live beside the transforms they are built on (rr_fft_regs.hip).
// (split out of rr_fused.hip in round 3; derivations and dropped variants: DESIGN_HISTORY.md 4)
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"
#include "rr_meter_dev.hpp"
#include "rr_fft_regs.hpp"

#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdlib>
#include <utility>
#include <vector>

namespace rr {

// ---------------------------------------------------------------------------
// Kernel 5  k_channelizer256<P>: the 256-bin polyphase channelizer (BASELINE configs[2]) with one
// wave per run of frames.  Frame f folds the P chunks 256 (f + p) .. + 255 of the windowed span into
// 256 values (4 per lane), then a forward DFT_256 as radix 4 x 4 x 4 x 4 with wave-local exchanges
// (no workgroup barrier).  Consecutive frames share P - 1 chunks, so a wave keeps a sliding window of
// chunks in registers and loads one new chunk (4 loads) per frame; the window values of the lane
// (4 P) and the three twiddle seeds stay in registers for the whole run.
// ---------------------------------------------------------------------------
#ifndef RR_V_CHAN_NT
#define RR_V_CHAN_NT 2  // bit 0: the pieces by non-temporal loads, bit 1: the bins by non-temporal stores (cfg3: 0.212 / 0.231 / 0.200 / 0.212 ms for 0 .. 3: the pieces neighbouring runs share want their L2 copies)
#endif
#define RR_V_CHANWIN 4  // cfg3: one contiguous eighth of the runs per XCD 0.250 ms; windows of 1 .. 6 and 64 runs per XCD 0.222-0.226; 8: 0.232, 16: 0.265, 32: 0.233
// H = hop / 64: 4 is the critically sampled filterbank (one new chunk of 256 per frame); 2 and 1 are the filterbanks
// oversampled 2 and 4 times (Rechunker(hop) -> Overlapper -> Fourier -> every P-th bin, hop 128 / 64): the window of
// samples a lane keeps moves on by H pieces of 64 per frame.
template <int P, int H>
__global__ __launch_bounds__(64) void k_channelizer256(const float2 *__restrict__ hist, long hist_len,
                                                       const float2 *__restrict__ in, long base0,
                                                       const float *__restrict__ window,
                                                       const float2 *__restrict__ tw, float2 *__restrict__ out,
                                                       unsigned nframes, unsigned run) {
    __shared__ __attribute__((aligned(16))) f2 lds[320];  // B(i) = i + 4 (i >> 4), i < 256
    const int l = threadIdx.x, g = l >> 4, q = l & 15;
    // workgroups b, b + 8, .. share an XCD: neighbouring runs on one XCD (grid: multiple of 8)
    // runs dealt to the XCDs in a moving window: RR_V_CHANWIN neighbouring runs per XCD (as k_ols_wave's blocks)
    const unsigned rb = blockIdx.x / (8 * RR_V_CHANWIN) * (8 * RR_V_CHANWIN) + (blockIdx.x % (8 * RR_V_CHANWIN) & 7) * RR_V_CHANWIN +
                        (blockIdx.x % (8 * RR_V_CHANWIN) >> 3);
    const unsigned f0 = rb * run;
    if (f0 >= nframes) return;
    const unsigned cnt = nframes - f0 < run ? nframes - f0 : run;

    float wv[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[p][c] = window[l + 64 * c + 256 * p];
    // twiddle seeds of passes 1..3: e^{-j 2 pi (l mod ns) / (4 ns)}, ns = 4, 16, 64 (tw[k] = e^{-j 2 pi k / 256})
    const float2 s1 = tw[(l & 3) * 16], s2 = tw[q * 4], s3 = tw[l];
    const f2 seed[3] = {(f2){s1.x, s1.y}, (f2){s2.x, s2.y}, (f2){s3.x, s3.y}};

    // piece qi of the stream (64 samples) starts at base0 + 64 qi; base0 and hist_len are multiples of the hop (64 H),
    // so a piece lies entirely in the history or entirely in the input.  Frame f covers the pieces H f .. H f + 4 P - 1.
    auto load_piece = [&](long qi) -> f2 {
        const long pos = base0 + 64 * qi;
        const float2 *src = pos >= 0 ? in + pos : hist + (hist_len + pos);
#if RR_V_CHAN_NT & 1
        return __builtin_nontemporal_load(reinterpret_cast<const f2 *>(src) + l);
#else
        const float2 v = src[l];
        return (f2){v.x, v.y};
#endif
    };
    constexpr int NQ = 4 * P;
    f2 xs[NQ];
#pragma unroll
    for (int i = 0; i + H < NQ; ++i) xs[i] = load_piece((long)H * f0 + i);

    f2 *const b_rd = lds + (l + 4 * g);  // B(l + 64 c) = b_rd + 80 c
    // the pieces a frame adds are requested one frame ahead (the last frame of the run asks for its own once more)
    // (cfg3: 0.217 -> 0.207 ms; two frames ahead no further gain; runs of 8 or 16 frames alike, 32 .. 128 slower: 0.226-0.236)
    f2 nx[H];
#pragma unroll
    for (int i = 0; i < H; ++i) nx[i] = load_piece((long)H * f0 + (NQ - H) + i);
    for (unsigned it = 0; it < cnt; ++it) {
#pragma unroll
        for (int i = 0; i < H; ++i) xs[NQ - H + i] = nx[i];
        {
            const long fn = (long)f0 + (it + 1 < cnt ? it + 1 : it);
#pragma unroll
            for (int i = 0; i < H; ++i) nx[i] = load_piece((long)H * fn + (NQ - H) + i);
        }
        // fold: y[c] = sum_p w[l + 64 c + 256 p] x[hop f + l + 64 c + 256 p]
        f2 y[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f2 acc = xs[c] * wv[0][c];
#pragma unroll
            for (int p = 1; p < P; ++p) acc = __builtin_elementwise_fma(xs[c + 4 * p], (f2){wv[p][c], wv[p][c]}, acc);
            y[c] = acc;
        }
#pragma unroll
        for (int i = 0; i + H < NQ; ++i) xs[i] = xs[i + H];
        // forward DFT_256, Stockham radix 4 x 4: butterfly l reads in[l + 64 c]
        dft4(y[0], y[1], y[2], y[3]);  // pass 0 (Ns = 1): out 4 l + c
        wave_sync();                   // the previous frame's reads are done
        {
            f2 *row = lds + (4 * l + 4 * (l >> 2));
            *reinterpret_cast<float4 *>(row) = (float4){y[0].x, y[0].y, y[1].x, y[1].y};
            *reinterpret_cast<float4 *>(row + 2) = (float4){y[2].x, y[2].y, y[3].x, y[3].y};
        }
        wave_sync();
#pragma unroll
        for (int pass = 1; pass < 4; ++pass) {
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = lds_ld(b_rd + (80 * c));
            const f2 w1 = seed[pass - 1];
            const f2 w2 = cmul(w1, w1);
            const f2 w3 = cmul(w2, w1);
            y[1] = cmul(y[1], w1);
            y[2] = cmul(y[2], w2);
            y[3] = cmul(y[3], w3);
            dft4(y[0], y[1], y[2], y[3]);
            if (pass == 3) break;  // natural order: y[c] = X[l + 64 c]
            wave_sync();
            if (pass == 1) {
                f2 *col = lds + (20 * (l >> 2) + (l & 3));
#pragma unroll
                for (int c = 0; c < 4; ++c) lds_st(col + (4 * c), y[c]);
            } else {
                f2 *col = lds + (80 * g + q);
#pragma unroll
                for (int c = 0; c < 4; ++c) lds_st(col + (20 * c), y[c]);
            }
            wave_sync();
        }
        float2 *dst = out + (size_t)(f0 + it) * 256 + l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#if RR_V_CHAN_NT & 2
            __builtin_nontemporal_store(y[c], reinterpret_cast<f2 *>(dst) + 64 * c);
#else
            float2 o;
            o.x = y[c].x;
            o.y = y[c].y;
            dst[64 * c] = o;
#endif
        }
    }
}

bool channelizer256_supported(int dtype, size_t M, size_t P, size_t hop) {
    if (dtype != RR_F32 || M != 256) return false;
    if (hop == 256) return P == 1 || P == 2 || P == 3 || P == 4 || P == 6 || P == 8;
    return (hop == 128 || hop == 64) && (P == 2 || P == 4 || P == 8);
}

int launch_channelizer256(hipStream_t s, const void *hist, size_t hist_len, const void *in, long base0, size_t P,
                          size_t nframes, const void *window, const void *tw, void *out, size_t hop) {
    if (nframes == 0) return RR_OK;
    if (nframes > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "channelizer: too many frames in one call");
    if (!channelizer256_supported(RR_F32, 256, P, hop))
        RR_FAIL(RR_ERR_BAD_ARG, "channelizer256: %zu taps per branch at hop %zu not instantiated", P, hop);
#define RR_V_CHANRUN 16
    const unsigned run = RR_V_CHANRUN;
    const unsigned grid = (unsigned)(((nframes + run - 1) / run + 8 * RR_V_CHANWIN - 1) / (8 * RR_V_CHANWIN) * (8 * RR_V_CHANWIN));
#define RR_CHAN_LAUNCH(PP, HH)                                                                                       \
    hipLaunchKernelGGL((k_channelizer256<PP, HH>), dim3(grid), dim3(64), 0, s, (const float2 *)hist, (long)hist_len, \
                       (const float2 *)in, base0, (const float *)window, (const float2 *)tw, (float2 *)out,         \
                       (unsigned)nframes, run)
    if (hop == 256) {
        switch (P) {
            case 1: RR_CHAN_LAUNCH(1, 4); break;
            case 2: RR_CHAN_LAUNCH(2, 4); break;
            case 3: RR_CHAN_LAUNCH(3, 4); break;
            case 4: RR_CHAN_LAUNCH(4, 4); break;
            case 6: RR_CHAN_LAUNCH(6, 4); break;
            case 8: RR_CHAN_LAUNCH(8, 4); break;
        }
    } else if (hop == 128) {
        switch (P) {
            case 2: RR_CHAN_LAUNCH(2, 2); break;
            case 4: RR_CHAN_LAUNCH(4, 2); break;
            case 8: RR_CHAN_LAUNCH(8, 2); break;
        }
    } else {
        switch (P) {
            case 2: RR_CHAN_LAUNCH(2, 1); break;
            case 4: RR_CHAN_LAUNCH(4, 1); break;
            case 8: RR_CHAN_LAUNCH(8, 1); break;
        }
    }
#undef RR_CHAN_LAUNCH
    RR_HIP(hipGetLastError());
    return RR_OK;
}


}  // namespace rr
