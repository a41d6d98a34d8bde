// rr_f64.hip — the Complex<f64> fast paths (numbers.rs:23-42: every block of the reference is generic over f32 and f64;
// analysis.rs:139-209, the reference's own Fourier test, runs in f64).
//
//   k_fft4096_f64   the 4096-point windowed transform of the Fourier block (analysis.rs:105-115) with the lane's 16 values in
//                   REGISTERS (radix 16 x 16 x 16, one padded LDS image of 4096 + 256 elements of 16 bytes = 68 KiB, two
//                   workgroups per CU) instead of k_fft_pow2<double>'s radix-4 Stockham passes between two LDS images
//                   (128 KiB: one workgroup per CU, six LDS round trips: 28 % of the 32 B/sample roofline).
//
// No packed arithmetic here (gfx950 has no packed f64): plain complex products, which the compiler contracts into FMAs.
#include "rr_blocks.hpp"

#include <cstdlib>

namespace rr {

namespace {

struct cd2 {
    double x, y;
};
__device__ __forceinline__ cd2 operator+(cd2 a, cd2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd2 operator-(cd2 a, cd2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd2 mul_mj(cd2 a) { return {a.y, -a.x}; }  // (-j) a
__device__ __forceinline__ cd2 add_mj(cd2 a, cd2 t) { return {a.x + t.y, a.y - t.x}; }  // a + (-j) t
__device__ __forceinline__ cd2 add_pj(cd2 a, cd2 t) { return {a.x - t.y, a.y + t.x}; }  // a + (+j) t
__device__ __forceinline__ cd2 cmul(cd2 a, cd2 w) { return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ cd2 cmulc(cd2 v, double wr, double wi) { return {v.x * wr - v.y * wi, v.x * wi + v.y * wr}; }

__device__ __forceinline__ void dft4(cd2 &a, cd2 &b, cd2 &c, cd2 &d) {  // forward: e^{-j 2 pi n k / 4}
    const cd2 s0 = a + c, s1 = a - c, s2 = b + d, t = b - d;
    a = s0 + s2;
    c = s0 - s2;
    b = add_mj(s1, t);
    d = add_pj(s1, t);
}

// in-register forward 16-point DFT, natural order in and out (the structure of rr_wave_math.hpp's dft16)
__device__ __forceinline__ void dft16(cd2 (&v)[16]) {
    constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, H = 0.70710678118654752440;
#pragma unroll
    for (int a = 0; a < 4; ++a) dft4(v[a], v[a + 4], v[a + 8], v[a + 12]);
    v[1 + 4] = cmulc(v[1 + 4], C1, -S1);
    v[1 + 8] = cmulc(v[1 + 8], H, -H);
    v[1 + 12] = cmulc(v[1 + 12], S1, -C1);
    v[2 + 4] = cmulc(v[2 + 4], H, -H);
    v[2 + 8] = mul_mj(v[2 + 8]);
    v[2 + 12] = cmulc(v[2 + 12], -H, -H);
    v[3 + 4] = cmulc(v[3 + 4], S1, -C1);
    v[3 + 8] = cmulc(v[3 + 8], -H, -H);
    v[3 + 12] = cmulc(v[3 + 12], -C1, S1);
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
    cd2 t[16];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) t[b + 4 * c] = v[4 * b + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = t[k];
}

// v[k] *= w^k, k = 1 .. 15, the powers by a product tree at most 4 deep
__device__ __forceinline__ void twiddle16(cd2 (&v)[16], cd2 w) {
    cd2 p[16];
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

__device__ __forceinline__ int pad16(int i) { return i + (i >> 4); }

// frames from [ head (n_head samples) | in ] at distance hop; tw[k] = e^{-j 2 pi k / 4096}, 4096 entries
__global__ __launch_bounds__(256) void k_fft4096_f64(const double2 *__restrict__ head, long n_head, const double2 *__restrict__ in,
                                                     double2 *__restrict__ out, const double *__restrict__ window,
                                                     const double2 *__restrict__ tw, int center_dc, long hop) {
    extern __shared__ __attribute__((aligned(16))) char smem64[];
    cd2 *const lds = reinterpret_cast<cd2 *>(smem64);  // 4096 + 256 elements
    const int j = threadIdx.x;
    const unsigned fr = blockIdx.x;
    const long base = (long)fr * hop - n_head;
    cd2 v[16];
    const double2 s1 = tw[16 * (j & 15)], s2 = tw[j];
    if (base >= 0) {
        const double2 *src = in + base + j;
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            x[k] = hop >= 4096 ? __builtin_nontemporal_load(reinterpret_cast<const d2v *>(src + 256 * k)) : *reinterpret_cast<const d2v *>(src + 256 * k);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double w = window[j + 256 * k];
            v[k] = {x[k].x * w, x[k].y * w};
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long i = base + j + 256 * k;
            const double2 x = (i >= 0) ? in[i] : head[n_head + i];
            const double w = window[j + 256 * k];
            v[k] = {x.x * w, x.y * w};
        }
    }
    // pass 0 (no twiddles), out index 16 j + k; padded rows for these stores
    dft16(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad16(16 * j + k)] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds[pad16(j + 256 * k)];
    twiddle16(v, cd2{s1.x, s1.y});  // e^{-j 2 pi k (j mod 16) / 256}
    dft16(v);
    __syncthreads();
    {
        const int b2 = (j >> 4) * 256 + (j & 15);  // second exchange: no padding (as k_fft4096)
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[b2 + 16 * k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds[j + 256 * k];
    twiddle16(v, cd2{s2.x, s2.y});  // e^{-j 2 pi k j / 4096}
    dft16(v);
    const int rot = center_dc ? 2048 : 0;
    double2 *dst = out + (size_t)fr * 4096;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store((d2v){v[k].x, v[k].y}, reinterpret_cast<d2v *>(dst + ((j + 256 * k + rot) & 4095)));
    }
}

}  // namespace

int launch_fft4096_f64(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count, const void *window,
                       const void *tw4096, bool center_dc, size_t hop) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fft4096 (f64): too many frames");
    constexpr size_t lds = (4096 + 256) * 16;
    static bool attr_set = false;
    if (!attr_set) {
        RR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft4096_f64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_fft4096_f64, dim3((unsigned)count), dim3(256), lds, s, (const double2 *)head, (long)n_head, (const double2 *)in,
                       (double2 *)out, (const double *)window, (const double2 *)tw4096, (int)center_dc, (long)hop);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
