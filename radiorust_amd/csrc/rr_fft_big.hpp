// rr_fft_big.hpp — forward DFTs of 8192 / 16 384 points held by ONE workgroup: N / 16 lanes with 16 values each, Stockham radix
// 16 x 16 x 16 x (N / 4096) through one image in LDS (N + N / 16 elements + a 240-entry twiddle table).  Shared by the Filter's
// k_filter_blkbig<N> (rr_filter_ols.hip) and the Fourier block's k_fft16384 (rr_fft_regs.hip).
//
// Butterfly i of a pass of radix R behind Ns points reads in[i + k N / R], multiplies by W_(Ns R)^((i mod Ns) k) and writes
// (i div Ns) Ns R + (i mod Ns) + k Ns.  With T = N / 16 lanes, lane j:
//   pass 0  R = 16, Ns = 1     butterfly j                     out 16 j + k             (image padded 17 per 16)
//   pass 1  R = 16, Ns = 16    twiddles W_256^((j mod 16) k)   out (j div 16) 256 + (j mod 16) + 16 k
//   pass 2  R = 16, Ns = 256   W_4096^((j mod 256) k)          out (j div 256) 4096 + (j mod 256) + 256 k
//   pass 3  R = N / 4096, Ns = 4096: butterflies i = j + T c, c < 16 / R, over the lane's values v[c + (16 / R) k], twiddles
//           W_N^((j + T c) k) = (W_N^j W_16^c)^k; X[i + 4096 k] comes out in v[c + (16 / R) k]: v[kk] = X[j + T kk], natural order
#pragma once
#include "rr_wave_math.hpp"

namespace rr {

constexpr int kBigTab = 16 * 15;  // W_256^(r k), r < 16, k = 1 .. 15
template <int N>
constexpr int big_fft_lds_elems() { return N + N / 16 + kBigTab; }  // (pad16(N - 1) = N + N / 16 - 2)

// LDS-only workgroup barrier: the plain __syncthreads() also drains vmcnt, i.e. it would wait for the table loads in flight
__device__ __forceinline__ void big_lds_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Loads through a buffer descriptor: one lane offset in a VGPR, the per-load offset in an SGPR - no 64-bit
// per-lane address arithmetic.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ f2 buf_ld_f2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX);
    return (f2){__uint_as_float(r.x), __uint_as_float(r.y)};
}
template <int AUX>
__device__ __forceinline__ float4 buf_ld_f4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
    return float4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
}

// the lane's constants: seeds of passes 2 and 3 and its row of the pass-1 table; fills the table (read behind the first barrier)
template <int N>
struct BigFftLane {
    f2 s2, s3;
    const f2 *trow;
    __device__ __forceinline__ void init(const float2 *__restrict__ tw, f2 *tab, int j) {
        const float2 t2 = tw[(N / 4096) * (j & 255)], t3 = tw[j];
        s2 = (f2){t2.x, t2.y};
        s3 = (f2){t3.x, t3.y};
        if (j < kBigTab) {
            const int r = j / 15, k = j - 15 * r + 1;
            const float2 t = tw[((N / 256) * r * k) & (N - 1)];
            tab[j] = (f2){t.x, t.y};
        }
        trow = tab + 15 * (j & 15) - 1;  // W_256^((j mod 16) k) = trow[k]
    }
};

// in: v[k] = x[j + T k]; out: v[k] = X[j + T k].  `late` runs in front of the last butterflies, where few registers are live.
// HIBASE (16 384 points, a kernel that runs the transform twice): see below
template <int N, bool HIBASE = false, class Late>
__device__ __forceinline__ void big_fft(f2 (&v)[16], f2 *img, const BigFftLane<N> &ln, int j, bool pre_barrier, Late &&late) {
    constexpr int T = N / 16, R3 = N / 4096, NB = 16 / R3;
    static_assert(N == 8192 || N == 16384, "8192 or 16384 points");
    const f2 *const rd = img + (j + (j >> 4));                 // pad16(j + T k) = rd + (T + T / 16) k
    f2 *const w0 = img + 17 * j;                                // pad16(16 j + k) = w0 + k
    f2 *const w1 = img + ((j >> 4) * 256 + (j & 15));           // + 16 k
    f2 *const w2 = img + ((j >> 8) * 4096 + (j & 255));         // + 256 k
    const f2 *const rd1 = img + j;                              // j + T k = rd1 + T k
    // (16 384 points: the reads k >= 8 lie more than 64 KiB - the reach of an LDS instruction's offset field - behind their base;
    //  a second base each, opaque to the compiler, instead of eight addresses per pattern kept in registers across both transforms)
    const f2 *rdh = rd + (T + T / 16) * 8, *rd1h = rd1 + T * 8;
    if constexpr (N == 16384 && HIBASE) asm volatile("" : "+v"(rdh), "+v"(rd1h));
    dft16(v);
    if (pre_barrier) big_lds_bar();  // the previous transform's last reads are done
#pragma unroll
    for (int k = 0; k < 16; ++k) lds_stv(w0 + k, v[k]);
    big_lds_bar();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = k < 8 ? lds_ldv(rd + (T + T / 16) * k) : lds_ldv(rdh + (T + T / 16) * (k - 8));
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], lds_ldv(ln.trow + k));
    dft16(v);
    big_lds_bar();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds_stv(w1 + 16 * k, v[k]);
    big_lds_bar();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = k < 8 ? lds_ldv(rd1 + T * k) : lds_ldv(rd1h + T * (k - 8));
    twiddle16(v, ln.s2);
    dft16(v);
    big_lds_bar();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds_stv(w2 + 256 * k, v[k]);
    big_lds_bar();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = k < 8 ? lds_ldv(rd1 + T * k) : lds_ldv(rd1h + T * (k - 8));
    late();
    constexpr float WR[8] = {1.f, 0.92387953251128673848f, 0.70710678118654752440f, 0.38268343236508978178f,
                             0.f, -0.38268343236508978178f, -0.70710678118654752440f, -0.92387953251128673848f};
    constexpr float WI[8] = {0.f, -0.38268343236508978178f, -0.70710678118654752440f, -0.92387953251128673848f,
                             -1.f, -0.92387953251128673848f, -0.70710678118654752440f, -0.38268343236508978178f};
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const f2 t1 = c == 0 ? ln.s3 : cmulc(ln.s3, WR[c], WI[c]);
        if constexpr (R3 == 4) {
            const f2 t2 = cmul(t1, t1), t3 = cmul(t2, t1);
            v[c + 4] = cmul(v[c + 4], t1);
            v[c + 8] = cmul(v[c + 8], t2);
            v[c + 12] = cmul(v[c + 12], t3);
            dft4(v[c], v[c + 4], v[c + 8], v[c + 12]);
        } else {
            const f2 b = cmul(v[c + 8], t1), s = v[c] + b;
            v[c + 8] = v[c] - b;
            v[c] = s;
        }
    }
}

}  // namespace rr
