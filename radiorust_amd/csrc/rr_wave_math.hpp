// rr_wave_math.hpp — packed complex f32 arithmetic and in-register DFT butterflies shared by the
// gfx950 kernels (rr_ols.hip, rr_fft_regs.hip, rr_bluestein.hip, rr_channelizer.hip, rr_filter_ols.hip, rr_decim.hip): a complex lives in one 64-bit VGPR pair,
// every helper is one or two VOP3P instructions.  Device code only.
#pragma once
#include <hip/hip_runtime.h>

namespace rr {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ---- packed complex arithmetic -------------------------------------------------
// A complex f32 lives in one 64-bit VGPR pair (re = low, im = high).  Multiplying by
// +-j is a swap with one sign flip; VOP3P packed adds take that for free through
// their op_sel / neg modifiers, so butterflies need no moves.  hipcc does not find
// these forms on its own (it emitted ~340 v_mov per 4096-point transform), hence the
// four one-instruction helpers below.
__device__ __forceinline__ f2 add_mj(f2 a, f2 t) {  // a + (-j) t = (a.x + t.y, a.y - t.x)
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));
    return r;
}
__device__ __forceinline__ f2 add_pj(f2 a, f2 t) {  // a + (+j) t = (a.x - t.y, a.y + t.x)
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(t));
    return r;
}
__device__ __forceinline__ f2 mul_mj(f2 a) {  // (-j) a = (a.y, -a.x)
    f2 r;
    asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ f2 mul_pj(f2 a) {  // (+j) a = (-a.y, a.x)
    f2 r;
    asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
// a * w with the pre-rotated partner wj = (+j) w = (-w.y, w.x): two packed ops
__device__ __forceinline__ f2 cmul2(f2 a, f2 w, f2 wj) { return __builtin_elementwise_fma(a.yy, wj, a.xx * w); }
__device__ __forceinline__ f2 cmulf(f2 a, f2 b) { return cmul2(a, b, mul_pj(b)); }

// forward 4-point DFT (kernel e^{-j 2 pi n k / 4}): 8 packed adds
__device__ __forceinline__ void dft4(f2 &a, f2 &b, f2 &c, f2 &d) {
    const f2 s0 = a + c, s1 = a - c, s2 = b + d, t = b - d;
    a = s0 + s2;
    c = s0 - s2;
    b = add_mj(s1, t);
    d = add_pj(s1, t);
}
// inverse 4-point DFT (kernel e^{+j 2 pi n k / 4})
__device__ __forceinline__ void idft4(f2 &a, f2 &b, f2 &c, f2 &d) {
    const f2 s0 = a + c, s1 = a - c, s2 = b + d, t = b - d;
    a = s0 + s2;
    c = s0 - s2;
    b = add_pj(s1, t);
    d = add_mj(s1, t);
}

// multiply by a compile-time constant (wr, wi): two packed ops on constant pairs
__device__ __forceinline__ f2 cmulc(f2 v, float wr, float wi) {
    return __builtin_elementwise_fma(v.yy, (f2){-wi, wr}, v.xx * (f2){wr, wi});
}

// in-register forward 16-point DFT, natural order in and out
__device__ __forceinline__ void dft16(f2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, H = 0.70710678118654752440f;
    // t[a][b] = DFT4 over m of v[a + 4m]
#pragma unroll
    for (int a = 0; a < 4; ++a) dft4(v[a], v[a + 4], v[a + 8], v[a + 12]);  // v[a + 4b] now holds t[a][b]
    // twiddle W16^(a b)
    v[1 + 4] = cmulc(v[1 + 4], C1, -S1);   // a=1,b=1: W^1
    v[1 + 8] = cmulc(v[1 + 8], H, -H);     // a=1,b=2: W^2
    v[1 + 12] = cmulc(v[1 + 12], S1, -C1); // a=1,b=3: W^3
    v[2 + 4] = cmulc(v[2 + 4], H, -H);     // a=2,b=1: W^2
    v[2 + 8] = mul_mj(v[2 + 8]);           // a=2,b=2: W^4 = -j
    v[2 + 12] = cmulc(v[2 + 12], -H, -H);  // a=2,b=3: W^6
    v[3 + 4] = cmulc(v[3 + 4], S1, -C1);   // a=3,b=1: W^3
    v[3 + 8] = cmulc(v[3 + 8], -H, -H);    // a=3,b=2: W^6
    v[3 + 12] = cmulc(v[3 + 12], -C1, S1); // a=3,b=3: W^9
    // X[b + 4c] = DFT4 over a of t[a][b]; t[a][b] sits in v[a + 4b]
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);  // v[4b + c] = X[b + 4c]
    // natural order: X[k], k = b + 4c  <-  v[4b + c]  (register renaming, no instructions)
    f2 t[16];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) t[b + 4 * c] = v[4 * b + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = t[k];
}

// in-register forward 8-point DFT, natural order in and out (29 packed ops)
__device__ __forceinline__ void dft8(f2 (&v)[8]) {
    constexpr float H = 0.70710678118654752440f;
    f2 a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
    f2 b0 = v[0] - v[4], b1 = v[1] - v[5], b2 = v[2] - v[6], b3 = v[3] - v[7];
    b1 = cmulc(b1, H, -H);   // W8^1
    b2 = mul_mj(b2);         // W8^2 = -j
    b3 = cmulc(b3, -H, -H);  // W8^3
    dft4(a0, a1, a2, a3);    // X[0], X[2], X[4], X[6]
    dft4(b0, b1, b2, b3);    // X[1], X[3], X[5], X[7]
    v[0] = a0;
    v[2] = a1;
    v[4] = a2;
    v[6] = a3;
    v[1] = b0;
    v[3] = b1;
    v[5] = b2;
    v[7] = b3;
}

__device__ __forceinline__ int pad16(int i) { return i + (i >> 4); }

// v[k] *= w^k, k = 1..15.  Powers by a product tree at most 4 deep (error ~4 ulp, not 15);
// every power is kept with its rotated partner (+j) w^k so that each product is two packed ops.
__device__ __forceinline__ void apply_twiddle_powers(f2 (&v)[16], f2 w) {
    f2 p[16], q[16];  // p[k] = w^k, q[k] = (+j) w^k
    p[1] = w;
    q[1] = mul_pj(w);
#define RR_TWP(k, a, b)          \
    p[k] = cmul2(p[a], p[b], q[b]); \
    q[k] = mul_pj(p[k]);
    RR_TWP(2, 1, 1)
    RR_TWP(3, 2, 1)
    RR_TWP(4, 2, 2)
    RR_TWP(5, 4, 1)
    RR_TWP(6, 4, 2)
    RR_TWP(7, 4, 3)
    RR_TWP(8, 4, 4)
    RR_TWP(9, 8, 1)
    RR_TWP(10, 8, 2)
    RR_TWP(11, 8, 3)
    RR_TWP(12, 8, 4)
    RR_TWP(13, 8, 5)
    RR_TWP(14, 8, 6)
    RR_TWP(15, 8, 7)
#undef RR_TWP
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul2(v[k], p[k], q[k]);
}


__device__ __forceinline__ int pad8(int i) { return i + (i >> 3); }
__device__ __forceinline__ void twiddle8(f2 (&a)[8], f2 w1) {  // a[k] *= w1^k
    const f2 w2 = cmulf(w1, w1), w3 = cmulf(w2, w1), w4 = cmulf(w2, w2);
    a[1] = cmulf(a[1], w1);
    a[2] = cmulf(a[2], w2);
    a[3] = cmulf(a[3], w3);
    a[4] = cmulf(a[4], w4);
    a[5] = cmulf(a[5], cmulf(w4, w1));
    a[6] = cmulf(a[6], cmulf(w4, w2));
    a[7] = cmulf(a[7], cmulf(w4, w3));
}

__device__ __forceinline__ f2 cmul(f2 a, f2 w) {  // a * w, two packed ops, no rotated copy of w
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ f2 cmul_conj(f2 a, f2 w) {  // a * conj(w)
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// acc + a * w
__device__ __forceinline__ f2 cmac(f2 acc, f2 a, f2 w) {
    f2 t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(w), "v"(t));
    return r;
}

// v[k] *= w^k, k = 1..15, product tree at most 4 deep
__device__ __forceinline__ void twiddle16(f2 (&v)[16], f2 w) {
    f2 p[16];
    p[1] = w;
    p[2] = cmul(p[1], p[1]);
    p[3] = cmul(p[2], p[1]);
    p[4] = cmul(p[2], p[2]);
    p[5] = cmul(p[4], p[1]);
    p[6] = cmul(p[4], p[2]);
    p[7] = cmul(p[4], p[3]);
    p[8] = cmul(p[4], p[4]);
#pragma unroll
    for (int k = 9; k < 16; ++k) p[k] = cmul(p[8], p[k - 8]);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], p[k]);
}


__device__ __forceinline__ void wave_sync() {
    // all 64 lanes of the only wave: order LDS writes before the following reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8-byte LDS accesses that the load/store optimizer must not pair up: it merges neighbouring ds_read_b64 /
// ds_write_b64 into ds_read2_b64 / ds_write2_b64, which take twice the LDS cycles of two single operations on
// this part (MI355X_MICROARCH.md, LDS table).  Volatile accesses in the LDS address space stay single.
typedef __attribute__((address_space(3))) f2 lds_f2;
__device__ __forceinline__ f2 lds_ldv(const f2 *p) { return *(const volatile lds_f2 *)p; }
__device__ __forceinline__ void lds_stv(f2 *p, f2 v) { *(volatile lds_f2 *)p = v; }
// the same, switchable per build for A/B runs of the older kernels (RR_V_LDSVOL)
#ifndef RR_V_LDSVOL
#define RR_V_LDSVOL 1
#endif
__device__ __forceinline__ f2 lds_ld(const f2 *p) {
#if RR_V_LDSVOL
    return lds_ldv(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void lds_st(f2 *p, f2 v) {
#if RR_V_LDSVOL
    lds_stv(p, v);
#else
    *p = v;
#endif
}

// ---- the 8192-point transform of a 256-lane workgroup, 32 values per lane (radix 16 x 16 x 32 through one padded LDS image
// of 8192 + 512 elements): the body of k_fft8192 ----
// The same transform as a function for kernels that run it twice (k_bluestein8192): in v[h][k] = x[t + 256 h + 512 k];
// out v[0][m] = X[t + 256 m], v[1][m] = X[t + 256 (m + 16)], m < 16.  The caller synchronises before the image is reused.
__device__ __forceinline__ void fft8192_regs(f2 (&v)[2][16], f2 *lds, const float2 *__restrict__ tw, int t) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = t + 256 * h;
        dft16(v[h]);
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(lds + pad16(16 * j + k), v[h][k]);
    }
    __syncthreads();
    {
        const float2 s1 = tw[32 * (t & 15)];
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
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 c = tw[256 * m];  // W_32^m (a scalar read)
            const f2 o = cmulf(v[1][m], (f2){c.x, c.y});
            const f2 e = v[0][m];
            v[0][m] = e + o;
            v[1][m] = e - o;
        }
    }
}


}  // namespace rr
