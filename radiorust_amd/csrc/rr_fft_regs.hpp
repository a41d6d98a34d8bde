// rr_fft_regs.hpp — the in-register / wave-local transform bodies that more than one translation unit runs:
//   fft4096_regs     DFT_4096 of a 256-lane workgroup, 16 values per lane, one padded LDS image  (k_stft4096, k_ols_frame, k_bluestein4096)
//   wave_dft1024(_t) DFT_1024 of ONE wave with wave-local exchanges, and the same network run backwards
//                    (k_filter_wave, k_fft1024, k_chan1024_multi, k_bluestein1024)
// plus the constants and the streaming load those kernels share.  Device code only (included by rr_ols.hip,
// rr_fft_regs.hip, rr_bluestein.hip).
#pragma once
#include "rr_wave_math.hpp"

namespace rr {

// ---------------------------------------------------------------------------
// LDS images of one wave (element = 8 bytes).  Image A holds the 1024-point intermediate of
// the forward transform at  A(i) = i + 2 (i >> 4) + 8 (i >> 8):  rows of 16 elements at a
// stride of 18 (16-byte aligned for ds_write_b128, and an odd multiple of 16 bytes so that the
// 16 rows a quarter-wave touches fall on distinct banks), plus 8 per 256 (the best of the
// paddings i + a (i >> 4) + b (i >> 7) + c (i >> 8) under a 64-bank model of the four access
// patterns of the radix 8 x 16 x 8 transform).  Image B (256-point inverse) uses B(i) = i + 4 (i >> 4).
// Every access pattern below is (lane term) + (compile-time offset), spelled out so that the
// offsets land in the instructions' immediate fields instead of per-access address arithmetic.
constexpr int kWaveLds = 1176;  // A(1023) + 1 = 1174, rounded up to a multiple of 8

// Measured and dropped (the bodies are in the history, commit 28a7624): persistent one-wave
// workgroups that request the next block's samples while the current one is transformed (0.143-0.150 ms against
// 0.135: 32 more registers, 3 instead of 4 waves per SIMD), CU-resident workgroups of 12-16 waves with H staged in
// LDS (0.149-0.19 ms), runs of neighbouring blocks per wave (0.155-0.160 ms), several waves per workgroup.
// Stamps showed a block at 14.0k cycles of which 7.3k are the wait for its own 8 KiB of samples (HBM latency under
// load ~3 us) and 5.6k the transforms; four such waves per SIMD hide each other's waits.
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));  // two complex samples, 8-byte aligned
// (the sample stream passes through once: with the streaming hint it does not displace the 16 KiB of H / twiddle
//  tables from the CU's 32 KiB L1)
__device__ __forceinline__ f4u ld_stream(const f4u *p) { return __builtin_nontemporal_load(p); }

// forward 4096-point DFT in registers + one padded LDS image (radix 16 x 3, Stockham)
__device__ __forceinline__ void fft4096_regs(f2 (&v)[16], f2 *lds, const float2 *__restrict__ tw, int j) {
    // in: v[k] = x[j + 256 k]; out: v[k] = X[j + 256 k]
    dft16(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds_st(lds + pad16(16 * j + k), v[k]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds_ld(lds + pad16(j + 256 * k));
    {
        const float2 t = tw[16 * (j & 15)];
        apply_twiddle_powers(v, (f2){t.x, t.y});
    }
    dft16(v);
    __syncthreads();
    {
        const int base = (j >> 4) * 256 + (j & 15);  // (second exchange: no padding, as k_fft4096)
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(lds + (base + 16 * k), v[k]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds_ld(lds + (j + 256 * k));
    {
        const float2 t = tw[j];
        apply_twiddle_powers(v, (f2){t.x, t.y});
    }
    dft16(v);
}

// The 4096-bin channelizer at hop = 4096 with TWO neighbouring frames per workgroup (as k_chan1024_pair): branches + 1 chunk reads
// for two frames instead of 2 branches; chunk p goes into frame A with the window's segment p and into frame B with segment p - 1.

// ---------------------------------------------------------------------------
// forward DFT_1024 of v (pair layout: v[2 k' + j] = x[2 l + j + 128 k']) -> X[k] = DFT[l + 64 k]
// (`mid` runs between the second exchange's writes and reads, where the fewest registers are live)
template <class Mid>
__device__ __forceinline__ void wave_dft1024(f2 (&v)[16], f2 (&X)[16], f2 *lds, int l, f2 t_p1, const f2 (&t_p2)[2], Mid &&mid) {
    const int g = l >> 4;
    f2 *const a_rd = lds + (l + 2 * g);  // A(l + 64 m + 256 c) = a_rd + 72 m + 296 c
    {
        f2 e0[8], e1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            e0[k] = v[2 * k];
            e1[k] = v[2 * k + 1];
        }
        dft8(e0);
        dft8(e1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = e0[k];
            v[8 + k] = e1[k];
        }
    }
    {
        f2 *row = lds + (18 * l + 8 * g);  // A(16 l + e) = 18 l + 8 g + e
#pragma unroll
        for (int k = 0; k < 16; k += 2)
            *reinterpret_cast<float4 *>(row + k) = (float4){v[k].x, v[k].y, v[k + 1].x, v[k + 1].y};
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds_ld(a_rd + (72 * (k & 3) + 296 * (k >> 2)));  // in[l + 64 k]
    twiddle16(v, t_p1);
    dft16(v);
    wave_sync();
    {
        f2 *col = lds + (144 * (l >> 3) + 8 * (l >> 4) + (l & 7));
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(col + (8 * k + 2 * (k >> 1)), v[k]);
    }
    mid();
    wave_sync();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f2 a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = lds_ld(a_rd + (72 * m + 144 * c + 8 * (c >> 1)));
        const f2 w1 = t_p2[m];
        const f2 w2 = cmul(w1, w1);
        const f2 w3 = cmul(w2, w1);
        const f2 w4 = cmul(w2, w2);
        a[1] = cmul(a[1], w1);
        a[2] = cmul(a[2], w2);
        a[3] = cmul(a[3], w3);
        a[4] = cmul(a[4], w4);
        a[5] = cmul(a[5], cmul(w4, w1));
        a[6] = cmul(a[6], cmul(w4, w2));
        a[7] = cmul(a[7], cmul(w4, w3));
        dft8(a);
#pragma unroll
        for (int c = 0; c < 8; ++c) X[m + 2 * c] = a[c];
    }
}

// The same transform run backwards (every stage transposed, in reverse order; the DFT matrix is symmetric,
// so this is again the forward DFT_1024): input Z[k] = z[l + 64 k] - the layout wave_dft1024 leaves its
// result in -, output in the pair layout v[2 k' + j] = DFT[2 l + j + 128 k'].  The LDS image is the same,
// reads and writes change places.
__device__ __forceinline__ void wave_dft1024_t(f2 (&Z)[16], f2 (&v)[16], f2 *lds, int l, f2 t_p1, const f2 (&t_p2)[2]) {
    const int g = l >> 4;
    f2 *const a_rd = lds + (l + 2 * g);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f2 a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = Z[m + 2 * c];
        dft8(a);
        const f2 w1 = t_p2[m];
        const f2 w2 = cmul(w1, w1);
        const f2 w3 = cmul(w2, w1);
        const f2 w4 = cmul(w2, w2);
        a[1] = cmul(a[1], w1);
        a[2] = cmul(a[2], w2);
        a[3] = cmul(a[3], w3);
        a[4] = cmul(a[4], w4);
        a[5] = cmul(a[5], cmul(w4, w1));
        a[6] = cmul(a[6], cmul(w4, w2));
        a[7] = cmul(a[7], cmul(w4, w3));
#pragma unroll
        for (int c = 0; c < 8; ++c) lds_st(a_rd + (72 * m + 144 * c + 8 * (c >> 1)), a[c]);
    }
    wave_sync();
    {
        const f2 *col = lds + (144 * (l >> 3) + 8 * (l >> 4) + (l & 7));
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = lds_ld(col + (8 * k + 2 * (k >> 1)));
    }
    dft16(v);
    twiddle16(v, t_p1);
    wave_sync();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds_st(a_rd + (72 * (k & 3) + 296 * (k >> 2)), v[k]);
    wave_sync();
    {
        const f2 *row = lds + (18 * l + 8 * g);
        f2 e0[8], e1[8];
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const float4 p = *reinterpret_cast<const float4 *>(row + k), q = *reinterpret_cast<const float4 *>(row + 8 + k);
            e0[k] = (f2){p.x, p.y};
            e0[k + 1] = (f2){p.z, p.w};
            e1[k] = (f2){q.x, q.y};
            e1[k + 1] = (f2){q.z, q.w};
        }
        dft8(e0);
        dft8(e1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[2 * k] = e0[k];
            v[2 * k + 1] = e1[k];
        }
    }
}

#define RR_V_FLTWWIN 64
#define RR_V_FLTWNT 3  // bit 0: streaming stores, bit 1: streaming loads (measured n = 64: 0.203 / 0.196 / 0.202 / 0.193 ms for 0 / 1 / 2 / 3)
#define RR_V_FLTWOCC 3  // waves per SIMD the register budget is cut for (140 registers; at 4 the kernel spills 12)

}  // namespace rr
