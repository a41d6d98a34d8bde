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


// ---------------------------------------------------------------------------
// k_ols4096_f64: overlap-save fast convolution for Complex<f64> - the Filter (D = 1), the Downsampler at integer ratios and the
// f64 chain's front end (a phase table riding along on the load) in one kernel:
//   y = IDFT_4096(DFT_4096(x_block) * G),  G = DFT_4096(c) / 4096,  overlap V >= Lc - 1 with (4096 - V) a multiple of D,
//   out[m] = y[e0 + D m]: of a block's 4096 - V valid results every D-th is stored.
// The transforms are k_fft4096_f64's (16 values per lane in registers, radix 16 x 16 x 16 through one 68 KiB image, two
// workgroups per CU); the inverse is the forward routine with the result index reversed.  The direct forms this replaces sit on
// the f64 FMA issue rate (rr_decim.hip: 368 FMAs per output at the chain's 184 taps); here a block costs 2 x 12 x 4096 butterfly
// levels whatever the response, and the kernel is bound by its 16 + 16 / D bytes per sample.
// ---------------------------------------------------------------------------
struct Ols64Args {
    const double2 *hist;   // the samples in front of the call (already mixed where a phase table rides along)
    int hist_len;
    const double2 *in;
    long n_in;
    const double2 *G;      // DFT_4096(c) / 4096, natural order
    const double2 *tw;     // e^{-j 2 pi k / 4096}
    int V;
    double2 *out;
    long n_out;
    long e0;               // out[m] = y[e0 + D m]
    unsigned D, per_block; // per_block = (4096 - V) / D outputs per block
    unsigned nblocks;
    double2 *hist_out;     // receives the last hist_out_len (mixed) samples of [ hist | in ] (may be null)
    int hist_out_len;
    const double2 *nco;    // denom entries + entry 0 once more + the rotations by 128 k samples (rr_freqshifter::prepare); denom = 0: no mixer
    unsigned denom, idx0;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ols4096_f64(Ols64Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem64[];
    cd2 *const lds = reinterpret_cast<cd2 *>(smem64);  // 4096 + 256 elements
    const int j = threadIdx.x;
    const int hop = 4096 - a.V;
    // neighbouring blocks (which share V samples) on one XCD, in a window of 8 x 8 blocks
    constexpr unsigned W = 8;
    const unsigned grp = blockIdx.x / (8 * W), rem = blockIdx.x % (8 * W);
    const unsigned blk = grp * 8 * W + (rem & 7) * W + (rem >> 3);
    if (blk >= a.nblocks) return;
    const long b0 = a.e0 - a.V + (long)blk * hop;
    auto mixed = [&](long pos) -> cd2 {  // sample `pos` of [ hist | in ], mixed
        cd2 h = {0.0, 0.0};
        if (pos >= 0) {
            if (pos < a.n_in) {
                const double2 x = a.in[pos];
                h = {x.x, x.y};
                if (a.denom) {
                    const double2 pp = a.nco[(unsigned)(((long)a.idx0 + pos) % (long)a.denom)];
                    h = cmul(h, cd2{pp.x, pp.y});
                }
            }
        } else if (pos >= -(long)a.hist_len) {
            const double2 x = a.hist[a.hist_len + pos];
            h = {x.x, x.y};
        }
        return h;
    };
    if (a.hist_out && blk == a.nblocks - 1)
        for (int i = j; i < a.hist_out_len; i += 256) {
            const cd2 h = mixed(a.n_in - a.hist_out_len + i);
            a.hist_out[i] = double2{h.x, h.y};
        }
    cd2 v[16];
    if (b0 >= 0 && b0 + 4096 <= a.n_in) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        const double2 *src = a.in + b0 + j;
        d2v x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(src + 256 * k));
        if (a.denom) {
            // the lane's phasor at the block start from the table, the 15 steps of 256 samples by the pure rotation behind it
            const double2 p0 = a.nco[(unsigned)(((long)a.idx0 + b0 + j) % (long)a.denom)];
            const double2 rt = a.nco[a.denom + 1 + 2];
            cd2 pp = {p0.x, p0.y};
            const cd2 rot = {rt.x, rt.y};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                v[k] = cmul(cd2{x[k].x, x[k].y}, pp);
                pp = cmul(pp, rot);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = {x[k].x, x[k].y};
        }
    } else {
        // edge blocks (history in front, nothing behind the input): one sample at a time through the image - unrolled beside the
        // interior path, the 16 index computations cost more registers than both transforms
#pragma unroll 1
        for (int k = 0; k < 16; ++k) lds[j + 256 * k] = mixed(b0 + j + 256 * k);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = lds[j + 256 * k];  // (the lane's own elements: no barrier)
        __syncthreads();                                       // ... but one in front of the first exchange's stores
    }
    const double2 s1 = a.tw[16 * (j & 15)], s2 = a.tw[j];
    auto transform = [&](bool pre_barrier) {
        dft16(v);
        if (pre_barrier) __syncthreads();  // the previous transform's last reads are done
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[pad16(16 * j + k)] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = lds[pad16(j + 256 * k)];
        twiddle16(v, cd2{s1.x, s1.y});
        dft16(v);
        __syncthreads();
        {
            const int b2 = (j >> 4) * 256 + (j & 15);
#pragma unroll
            for (int k = 0; k < 16; ++k) lds[b2 + 16 * k] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = lds[j + 256 * k];
        twiddle16(v, cd2{s2.x, s2.y});
        dft16(v);
    };
    transform(false);
    // (G in two halves behind the transform: requested in front of its last butterflies, 16 more values of 16 bytes beside the lane's
    //  16 and the twiddles' 16 went past the 256 registers of two waves per SIMD - and scratch is HBM traffic)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        asm volatile("" ::: "memory");
        double2 g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = a.G[j + 256 * (8 * h + k)];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[8 * h + k] = cmul(v[8 * h + k], cd2{g[k].x, g[k].y});
    }
    transform(true);
    // y[t] = v[k] with t = (4096 - (j + 256 k)) mod 4096; valid for t >= V; stored if D divides t - V: out[blk per_block + (t - V) / D]
    const long mbase = (long)blk * a.per_block;
    typedef double d2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int t = (4096 - j - 256 * k) & 4095;
        const int u = t - a.V;
        if (u < 0) continue;
        unsigned i = (unsigned)u;
        if (a.D > 1) {
            if (i % a.D) continue;
            i /= a.D;
        }
        const long m = mbase + i;
        if (m < a.n_out) __builtin_nontemporal_store((d2v){v[k].x, v[k].y}, reinterpret_cast<d2v *>(a.out + m));
    }
}

}  // namespace

int launch_fft4096_f64(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count, const void *window,
                       const void *tw4096, bool center_dc, size_t hop) {
    if (count == 0) return RR_OK;
    if (count > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fft4096 (f64): too many frames");
    constexpr size_t lds = (4096 + 256) * 16;
    RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(k_fft4096_f64), lds));
    hipLaunchKernelGGL(k_fft4096_f64, dim3((unsigned)count), dim3(256), lds, s, (const double2 *)head, (long)n_head, (const double2 *)in,
                       (double2 *)out, (const double *)window, (const double2 *)tw4096, (int)center_dc, (long)hop);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// V: the overlap for a response of Lc taps at decimation D - at least Lc - 1, 4096 - V a multiple of D (0: does not fit)
size_t ols4096_f64_overlap(size_t Lc, uint64_t D) {
    if (Lc < 1 || Lc - 1 > 2048 || D < 1 || D > 1024) return 0;
    size_t V = Lc - 1;
    while (V < 4096 && (4096 - V) % D) ++V;
    if (V < 1) V = (4096 % D == 0 && D > 1) ? D : 1;  // (a response of one tap: still a block with a hop the ratio divides)
    while (V < 4096 && (4096 - V) % D) ++V;
    return (V <= 3072) ? V : 0;
}

// out[m] = sum_i c[i] x[e0 + D m - i] over [ hist | in ], optionally x * nco on the load (hist / hist_out then hold mixed samples)
int launch_ols4096_f64(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G, const void *tw4096,
                       size_t V, uint64_t D, void *out, size_t n_out, long e0, void *hist_out, size_t hist_out_len, const void *nco,
                       uint32_t denom, uint32_t idx0) {
    if (n_out == 0) {
        if (hist_out) RR_FAIL(RR_ERR_BAD_ARG, "k_ols4096_f64 cannot leave a history without producing outputs");
        return RR_OK;
    }
    if (V < 1 || V > 3072 || D < 1 || (4096 - V) % D) RR_FAIL(RR_ERR_BAD_ARG, "k_ols4096_f64: overlap %zu at decimation %llu", V, (unsigned long long)D);
    Ols64Args a;
    a.hist = (const double2 *)hist;
    a.hist_len = (int)hist_len;
    a.in = (const double2 *)in;
    a.n_in = (long)n_in;
    a.G = (const double2 *)G;
    a.tw = (const double2 *)tw4096;
    a.V = (int)V;
    a.out = (double2 *)out;
    a.n_out = (long)n_out;
    a.e0 = e0;
    a.D = (unsigned)D;
    a.per_block = (unsigned)((4096 - V) / D);
    const size_t nblocks = (n_out + a.per_block - 1) / a.per_block;
    if (nblocks > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "k_ols4096_f64: too many blocks");
    a.nblocks = (unsigned)nblocks;
    a.hist_out = (double2 *)hist_out;
    a.hist_out_len = (int)hist_out_len;
    a.nco = (const double2 *)nco;
    a.denom = nco ? denom : 0;
    a.idx0 = idx0;
    constexpr size_t lds = (4096 + 256) * 16;
    RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(k_ols4096_f64), lds));
    const unsigned grid = (unsigned)((nblocks + 63) / 64 * 64);
    hipLaunchKernelGGL(k_ols4096_f64, dim3(grid), dim3(256), lds, s, a);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
