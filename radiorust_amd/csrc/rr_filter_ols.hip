// rr_filter_ols.hip — the Filter block's fast convolution (filters.rs:240-259) for long responses
// (n = 256 .. 2048 taps, BASELINE configs[4]: n = 1024 at 2 GS/s), Complex<f32>, on gfx950.
//
//   y = IDFT_4096(DFT_4096(x_block) * G),  G = DFT_4096(g) / 4096,  overlap V >= n - 1,
// the same linear convolution as the reference's 2n-point transforms (it differs by rounding only).
// A workgroup of 256 lanes per 4096-sample block, 16 values per lane, radix 16 x 16 x 16 (Stockham
// autosort through a padded LDS image).
//
// What this kernel does differently from the first version (k_filter_ols4096, rr_fused.hip; 1212 vector
// instructions per wave and block, 36-42 % of the 16 B/sample roofline, VALU-bound):
//   * the inverse transform is the FORWARD routine with the output index reversed,
//     IDFT(Z)[t] = DFT(Z)[(4096 - t) mod 4096]: no conjugations, the reversal is an address on the store;
//   * complex products are the two-instruction VOP3P forms of rr_wave_math.hpp (no rotated partners);
//   * interior blocks load through a uniform base + lane offset and store through a buffer descriptor
//     whose range check drops the lanes outside the block's valid part: no per-element index logic;
//   * LDS reads and writes stay single 8-byte operations (the two-address forms run at half rate);
//   * MODE 1: persistent workgroups that request the next block's samples before transforming the
//     current one; MODE 2: the same with two LDS images (4 instead of 8 barriers per block) and the
//     twiddle powers and the lane's G values kept in registers across blocks.
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"

#include <hip/hip_fp16.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace rr {

namespace {

constexpr int kImg = 4096 + 256;  // padded image: pad16(4095) = 4350

#ifndef RR_V_F4KVOL
#define RR_V_F4KVOL 1  // 0: plain LDS accesses, which the compiler pairs into ds_read2_b64 / ds_write2_b64 (A/B runs)
#endif
__device__ __forceinline__ f2 img_ld(const f2 *p) { return RR_V_F4KVOL ? lds_ldv(p) : *p; }
__device__ __forceinline__ void img_st(f2 *p, f2 v) {
    if (RR_V_F4KVOL) lds_stv(p, v);
    else *p = v;
}

// LDS-only workgroup barrier: the plain __syncthreads() also drains vmcnt, i.e. it would wait for the
// next block's prefetch and for the previous block's stores at every exchange.
__device__ __forceinline__ void lds_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Loads through a buffer descriptor: one lane offset in a VGPR, the per-load offset in an SGPR - no 64-bit
// per-lane address arithmetic (which the compiler hoists out of the block loop and then spills).
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

// p[k] = w^k, k = 1 .. 15 (product tree at most 4 deep)
__device__ __forceinline__ void powers16(f2 (&p)[16], f2 w) {
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
}

struct Blk4096Args {
    const float2 *hist;
    int hist_len;
    const float2 *in;
    long n_in;
    const void *G;       // pair-interleaved: Gp[kp][j] = {G[j + 512 kp], G[j + 512 kp + 256]}, f32 or f16
    const float2 *tw;    // e^{-j 2 pi k / 4096}
    int V;
    void *out;
    long n_out;
    long e0;
    unsigned nblocks;
    unsigned blk_lo, blk_hi;  // the blocks that lie entirely inside the input
    unsigned npersist;        // persistent forms: workgroups of the block loop (the rest: one edge block each)
};

// MODE 0: one block per workgroup.  MODE 1: persistent, next block's samples prefetched, one image.
// MODE 2: persistent + prefetch, two images, twiddle powers kept.
template <bool OUT16, bool G16, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? 4 : (MODE == 2 ? 2 : 3), MODE == 0 ? 4 : (MODE == 2 ? 2 : 3))))
void k_filter_blk4096(Blk4096Args a) {
    constexpr bool DB = MODE == 2, KEEP = MODE == 2, PF = MODE != 0;
    constexpr bool KEEPG = MODE >= 2;  // MODE 3: MODE 1 with the lane's G values kept in registers
    // where the next block's samples are requested: in the middle of the forward transform (MODE 2: 256 registers)
    // or behind the G product (MODE 1: 32 registers fewer through the forward transform)
    constexpr bool PFEARLY = MODE == 2;
    __shared__ __attribute__((aligned(16))) f2 lds_all[DB ? 2 * kImg : kImg];
    f2 *const imgA = lds_all, *const imgB = DB ? lds_all + kImg : lds_all;
    const int j = threadIdx.x;
    const int hop = 4096 - a.V;
    const int esz = OUT16 ? 4 : 8;

    // lane terms of the four access patterns; everything else is an immediate offset
    const int rd_off = j + (j >> 4);                 // pad16(j + 256 k) = rd_off + 272 k
    const int w0_off = 17 * j;                       // pad16(16 j + k)  = w0_off + k
    const int w1_off = (j >> 4) * 272 + (j & 15);    // pad16((j / 16) 256 + j % 16 + 16 k) = w1_off + 17 k

    // twiddle seeds: pass 1 e^{-j 2 pi (j mod 16) / 256} = tw[16 (j mod 16)], pass 2 tw[j]
    f2 s1, s2;
    {
        const float2 t1 = a.tw[16 * (j & 15)], t2 = a.tw[j];
        s1 = (f2){t1.x, t1.y};
        s2 = (f2){t2.x, t2.y};
    }
    f2 p1[16], p2[16];
    if constexpr (KEEP) {
        powers16(p1, s1);
        powers16(p2, s2);
    }

    // forward DFT_4096: in v[k] = x[j + 256 k], out v[k] = X[j + 256 k].  `mid` runs between the writes and the
    // reads of the second exchange, where the fewest registers are live: the place to request tables
    // and the next block's samples; `late` in front of the last butterflies (MODE 0: 128 registers).
    auto transform = [&](f2 (&v)[16], bool pre_barrier, auto &&mid, auto &&late) {
        dft16(v);
        if (!DB && pre_barrier) lds_bar();  // the previous transform's last reads are done
        {
            f2 *w = imgA + w0_off;
#pragma unroll
            for (int k = 0; k < 16; ++k) img_st(w + k, v[k]);
        }
        lds_bar();
        {
            const f2 *r = imgA + rd_off;
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = img_ld(r + 272 * k);
        }
        if constexpr (KEEP) {
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], p1[k]);
        } else {
            twiddle16(v, s1);
        }
        dft16(v);
        if (!DB) lds_bar();
        {
            f2 *w = imgB + w1_off;
#pragma unroll
            for (int k = 0; k < 16; ++k) img_st(w + 17 * k, v[k]);
        }
        mid();
        lds_bar();
        {
            const f2 *r = imgB + rd_off;
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = img_ld(r + 272 * k);
        }
        if constexpr (KEEP) {
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], p2[k]);
        } else {
            twiddle16(v, s2);
        }
        late();
        dft16(v);
    };

    const long n_clamp = a.n_in - 4096;  // PF: the launcher guarantees n_in >= 4096
    const __amdgpu_buffer_rsrc_t rsG = rsrc_of(a.G, G16 ? 16384 : 32768);
    f2 x[16];  // PF: the next block's samples
    float4 gkeep[8];  // KEEPG: the lane's 16 G values are the same for every block
    if constexpr (KEEPG) {
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
            if constexpr (G16) {
                const f2 raw = buf_ld_f2<0>(rsG, 8u * j, 2048u * kp);
                gkeep[kp] = float4{raw.x, raw.y, 0.f, 0.f};
            } else {
                gkeep[kp] = buf_ld_f4<0>(rsG, 16u * j, 4096u * kp);
            }
        }
    }
    auto request = [&](unsigned blk_, f2(&dst)[16]) {  // the 4096 samples of an interior block (clamped into the input)
        long b = a.e0 - a.V + (long)blk_ * hop;
        b = b < 0 ? 0 : (b > n_clamp ? n_clamp : b);
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.in + b, 32768);
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[k] = buf_ld_f2<2>(rs, 8u * j, 2048u * k);
    };
    auto load_edge = [&](unsigned blk_, f2(&v)[16]) {
        // edges: the previous chunk in front (none after a reset), nothing behind the input
        const long b0 = a.e0 - a.V + (long)blk_ * hop;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long pos = b0 + j + 256 * k;
            float2 xv;
            xv.x = 0.f;
            xv.y = 0.f;
            if (pos >= 0) {
                if (pos < a.n_in) xv = a.in[pos];
            } else if (pos >= -(long)a.hist_len) {
                xv = a.hist[a.hist_len + pos];
            }
            v[k] = (f2){xv.x, xv.y};
        }
    };
    // one block: forward transform, * G, forward transform again, reversed store.  PFN: request block `nblk`'s
    // samples on the way (into x).
    auto do_block = [&](auto PFN, f2(&v)[16], unsigned blk_, unsigned nblk, bool pre_barrier) {
        constexpr bool pfn = decltype(PFN)::value;
        // The block's 16 G values (8 reads of 16 or 8 bytes per lane) are requested in the middle of the forward
        // transform, the next block's samples AFTER them: loads complete in order (vmcnt), so a G value requested
        // behind the prefetch could not be waited for without waiting for the prefetch as well.  The prefetch is
        // unconditional, so that no later wait has to assume it might not have been issued.
        float4 g4[8];
        if constexpr (KEEPG) {
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) g4[kp] = gkeep[kp];
        }
        auto load_g = [&] {
            if constexpr (KEEPG) return;
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) {
                if constexpr (G16) {
                    const f2 raw = buf_ld_f2<0>(rsG, 8u * j, 2048u * kp);
                    g4[kp] = float4{raw.x, raw.y, 0.f, 0.f};
                } else {
                    g4[kp] = buf_ld_f4<0>(rsG, 16u * j, 4096u * kp);
                }
            }
        };
        transform(
            v, pre_barrier,
            [&] {
                if constexpr (PF) load_g();
                if constexpr (pfn && PFEARLY) request(nblk, x);
            },
            [&] {
                if constexpr (!PF) load_g();
            });
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
            f2 ga, gb;
            if constexpr (G16) {
                const unsigned ra = __float_as_uint(g4[kp].x), rb = __float_as_uint(g4[kp].y);
                const float2 fa = __half22float2(*reinterpret_cast<const __half2 *>(&ra));
                const float2 fb = __half22float2(*reinterpret_cast<const __half2 *>(&rb));
                ga = (f2){fa.x, fa.y};
                gb = (f2){fb.x, fb.y};
            } else {
                ga = (f2){g4[kp].x, g4[kp].y};
                gb = (f2){g4[kp].z, g4[kp].w};
            }
            v[2 * kp] = cmul(v[2 * kp], ga);
            v[2 * kp + 1] = cmul(v[2 * kp + 1], gb);
        }
        if constexpr (pfn && !PFEARLY) request(nblk, x);
        transform(v, true, [] {}, [] {});
        // y[t] = v[k] with t = (4096 - (j + 256 k)) mod 4096; valid for t >= V: output mbase + t - V.
        // Offsets in the block's output window: (hop - j - 256 k) elements; t = 0 lands on `hop` (past the
        // window), t < V wraps to a huge offset: the descriptor's range check drops both.
        const long mbase = (long)blk_ * hop;
        const long left = a.n_out - mbase;
        const unsigned recs = (unsigned)(left < hop ? left : hop) * (unsigned)esz;
        char *obase = reinterpret_cast<char *>(a.out) + mbase * esz;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase, 0, recs, 0x00020000);
        const unsigned lane_off = (unsigned)(hop - j) * (unsigned)esz;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned off = lane_off - (unsigned)(256 * k * esz);
            if constexpr (OUT16) {
                const __half2 h = __floats2half2_rn(v[k].x, v[k].y);
                __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const unsigned *>(&h), rs, off, 0, 2);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(v[k], rs, off, 0, 2);
            }
        }
    };

    if constexpr (!PF) {
        // one block per workgroup.  Workgroups b, b + 8, .. share an XCD; neighbouring blocks (which share V
        // samples) go to one XCD, the XCDs work side by side in a moving window of 8 x 16 blocks.
        constexpr unsigned W = 16;
        const unsigned grp = blockIdx.x / (8 * W), rem = blockIdx.x % (8 * W);
        const unsigned blk = grp * 8 * W + (rem & 7) * W + (rem >> 3);
        if (blk >= a.nblocks) return;
        f2 v[16];
        if (blk >= a.blk_lo && blk < a.blk_hi) request(blk, v);
        else load_edge(blk, v);
        do_block(std::false_type{}, v, blk, blk, false);
    } else {
        // Persistent workgroups over the interior blocks [blk_lo, blk_hi) - every sample inside the input - and one
        // extra workgroup per edge block (it reaches into the history or past the end of the input).  The edge
        // path has per-element conditions; kept out of the loop, the loop's waits stay counted (a join with
        // conditional loads makes the compiler wait for everything, the previous block's stores included).
        if (blockIdx.x >= a.npersist) {
            const unsigned e = blockIdx.x - a.npersist;
            const unsigned blk = e < a.blk_lo ? e : a.blk_hi + (e - a.blk_lo);
            if (blk >= a.nblocks) return;
            f2 v[16];
            load_edge(blk, v);
            do_block(std::false_type{}, v, blk, blk, false);
            return;
        }
        // an XCD takes a contiguous run of every round's blocks
        const unsigned per_xcd = a.npersist >> 3;  // npersist: a multiple of 8
        unsigned blk = a.blk_lo + (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (blk >= a.blk_hi) return;
        request(blk, x);
        // The first block's samples are waited for HERE (a use the compiler has to honour).  Entering the loop with
        // them in flight, the wait at the top of the loop would have to serve two queue shapes - nothing behind the
        // samples on this way in, sixteen stores behind them on the way round - and the compiler then waits for
        // everything: every block would wait for the previous block's stores.
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        asm volatile("" : "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
        for (;;) {
            // what stays in registers across blocks is decided here, not by invariant-code motion (which would
            // hoist the 30 twiddle powers and every address, and then spill)
            if constexpr (!KEEP) asm volatile("" : "+v"(s1), "+v"(s2));
            f2 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = x[k];
            const unsigned nblk = blk + a.npersist;
            const bool more = nblk < a.blk_hi;
            do_block(std::true_type{}, v, blk, more ? nblk : blk, true);
            if (!more) break;
            blk = nblk;
        }
    }
}

}  // namespace

bool filter_blk4096_supported(int dtype, size_t n) {
    return dtype == RR_F32 && (n == 256 || n == 512 || n == 1024 || n == 2048);
}

// variant: 0 = one block per workgroup, 1 = persistent + prefetch, 2 = persistent, two images, powers kept
int launch_filter_blk4096(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G,
                          const void *tw4096, size_t n, void *out, size_t n_out, long e0, bool out_f16, bool g_f16,
                          int variant, size_t persist_min_blocks) {
    if (n_out == 0) return RR_OK;
    Blk4096Args a;
    a.hist = (const float2 *)hist;
    a.hist_len = (int)hist_len;
    a.in = (const float2 *)in;
    a.n_in = (long)n_in;
    a.G = G;
    a.tw = (const float2 *)tw4096;
    a.V = (int)n;  // n is a multiple of 256 here, V >= n - 1
    a.out = out;
    a.n_out = (long)n_out;
    a.e0 = e0;
    const size_t hop = 4096 - a.V;
    const size_t nblocks = (n_out + hop - 1) / hop;
    if (nblocks > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Filter: too many blocks");
    a.nblocks = (unsigned)nblocks;
    // blocks entirely inside the input: b0 = e0 - V + blk * hop >= 0 and b0 + 4096 <= n_in
    {
        const long first = e0 - a.V;
        long lo = first >= 0 ? 0 : (-first + (long)hop - 1) / (long)hop;
        long hi = ((long)n_in - 4096 - first) >= 0 ? ((long)n_in - 4096 - first) / (long)hop + 1 : 0;
        if (hi > (long)nblocks) hi = (long)nblocks;
        if (lo > hi) lo = hi;
        a.blk_lo = (unsigned)lo;
        a.blk_hi = (unsigned)hi;
    }
    int mode = variant;
    if (mode < 0 || mode > 3) mode = 1;
    if (n_in < 4096 || nblocks < persist_min_blocks || a.blk_hi == a.blk_lo) mode = 0;  // the persistent forms prefetch whole blocks from inside the input
    unsigned grid;
    a.npersist = 0;
    if (mode == 0) {
        grid = (unsigned)((nblocks + 127) / 128 * 128);
    } else {
        const unsigned per_cu = mode == 2 ? 2 : 3;
        const size_t interior = a.blk_hi - a.blk_lo;
        a.npersist = (unsigned)(interior < 256 * per_cu ? (interior + 7) / 8 * 8 : 256 * per_cu);
        grid = a.npersist + (unsigned)(nblocks - interior);
    }
#define RR_BLK_LAUNCH(O, GG, M) hipLaunchKernelGGL((k_filter_blk4096<O, GG, M>), dim3(grid), dim3(256), 0, s, a)
#define RR_BLK_MODE(O, GG)                 \
    do {                                   \
        if (mode == 0) RR_BLK_LAUNCH(O, GG, 0); \
        else if (mode == 1) RR_BLK_LAUNCH(O, GG, 1); \
        else if (mode == 2) RR_BLK_LAUNCH(O, GG, 2); \
        else RR_BLK_LAUNCH(O, GG, 3);      \
    } while (0)
    if (out_f16) {
        if (g_f16) RR_BLK_MODE(true, true);
        else RR_BLK_MODE(true, false);
    } else {
        if (g_f16) RR_BLK_MODE(false, true);
        else RR_BLK_MODE(false, false);
    }
#undef RR_BLK_MODE
#undef RR_BLK_LAUNCH
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
