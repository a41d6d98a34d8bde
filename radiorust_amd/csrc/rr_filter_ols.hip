// rr_filter_ols.hip — the Filter block's fast convolution (filters.rs:240-259) for long responses
// (n = 256 .. 2048 taps, BASELINE configs[4]: n = 1024 at 2 GS/s), Complex<f32>, on gfx950.
//
//   y = IDFT_4096(DFT_4096(x_block) * G),  G = DFT_4096(g) / 4096,  overlap V >= n - 1,
// the same linear convolution as the reference's 2n-point transforms (it differs by rounding only).
// A workgroup of 256 lanes per 4096-sample block, 16 values per lane, radix 16 x 16 x 16 (Stockham
// autosort through a padded LDS image).
//
// What this kernel does differently from the first version (k_filter_ols4096: 1212 vector instructions per wave
// and block, 0.32 ms per 2^26 samples = 41 % of the 16 B/sample roofline, VALU-bound at 71 % busy):
//   * the inverse transform is the FORWARD routine with the output index reversed,
//     IDFT(Z)[t] = DFT(Z)[(4096 - t) mod 4096]: no conjugations, the reversal is an address on the store;
//   * complex products are the two-instruction VOP3P forms of rr_wave_math.hpp (no rotated partners), and the
//     powers of the second pass's twiddles (which depend on the lane only through j mod 16) come from a
//     240-entry table in LDS instead of a product tree per lane;
//   * interior blocks load through a buffer descriptor (one lane offset, scalar per-load offsets) and every
//     block stores through one whose range check drops the lanes outside the block's valid part: no
//     per-element index logic, no 64-bit per-lane address arithmetic;
//   * LDS reads and writes stay single 8-byte operations: paired by the compiler into ds_read2_b64 /
//     ds_write2_b64 the same kernel is 5 % slower (0.295 against 0.280 ms per call);
//   * the last block's workgroup also writes the next call's history (no second launch).
// Measured: persistent workgroups with the next block's samples prefetched into registers
// (3 per CU), the same with two LDS images and twiddle powers + G kept in registers (2 per CU), and with G
// kept (3 per CU) were all 1-6 % SLOWER than one block per workgroup at 4 workgroups per CU, and are not kept
// (git tag cfg5-variants-kept holds them).
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"
#include "rr_fft_big.hpp"

#include <hip/hip_fp16.h>

#include <cstdlib>
#include <cstring>

namespace rr {

namespace {

constexpr int kImg = 4096 + 256;  // padded image: pad16(4095) = 4350
constexpr int kTab = 16 * 15;     // W_256^(r k), r < 16, k = 1 .. 15

#ifndef RR_V_F4KVOL
#define RR_V_F4KVOL 1  // 0: plain LDS accesses, which the compiler pairs into ds_read2_b64 / ds_write2_b64 (A/B runs)
#endif
#ifndef RR_V_F4K_TAIL
#define RR_V_F4K_TAIL 0  // 1: the block's last V samples (the next block's first) WITHOUT the streaming hint.  Measured (round 3,
                         // one session): FETCH_SIZE x 2 = 610.5 MB per 2^26-sample call against 598.3 MB with the hint on every load,
                         // 0.2704 against 0.2698 ms - two thirds of the overlap hit in L2 either way; the hint stays on
#endif
#ifndef RR_V_F4K_DEAD
#define RR_V_F4K_DEAD 1  // 0: also issue the stores whose 256 lanes all fall outside the block's valid part (A/B runs)
#endif
#ifndef RR_V_F4KTAB
#define RR_V_F4KTAB 1  // 0: the second pass's twiddle powers by a product tree per lane (A/B runs)
#endif
__device__ __forceinline__ f2 img_ld(const f2 *p) { return RR_V_F4KVOL ? lds_ldv(p) : *p; }
__device__ __forceinline__ void img_st(f2 *p, f2 v) {
    if (RR_V_F4KVOL) lds_stv(p, v);
    else *p = v;
}

// LDS-only workgroup barrier: the plain __syncthreads() also drains vmcnt, i.e. it would wait for the
// table loads in flight at every exchange.
__device__ __forceinline__ void lds_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
    float2 *hist_out;         // receives the last hist_out_len samples of [ hist | in ] (may be null)
    int hist_out_len;
    int nparts;               // PARTS: partitions of 2048 taps, G holds their tables one behind the other
    SelectArgs sel;           // SEL: the Downsampler's periodic schedule (launch_decim_select_blk)
};

// ACC: the block's results are added to what `out` holds (responses longer than 2048 taps run as partitions of 2048,
// one launch each, the later ones delayed by 2048 p samples and accumulating).
// PARTS: responses beyond 2048 taps in ONE launch.  g = sum_p delay(g_p, 2048 p) with partitions g_p of 2048 taps, so
// y_block = IDFT( sum_p DFT(x_block delayed by 2048 p) G_p ): the workgroup transforms its block of the stream at the
// nparts delays, sums the products in registers and runs ONE inverse - nparts + 1 transforms per block where a launch per
// partition (the first form, kept as ACC) takes 2 nparts and reads and rewrites the output nparts - 1 times.
// SEL: the Downsampler for ANY periodic schedule with responses of up to 2048 taps (k_filter_wave<true> serves up to 385): the
// response at every input position, the results of the releasing positions stored - see k_filter_wave<true> in rr_ols.hip for
// the schedule's arithmetic; here every one of a lane's 16 results reduces its own F(t) = Rb + t rb (exact in f64).
template <bool OUT16, bool G16, bool ACC, bool PARTS = false, bool SEL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PARTS ? 3 : 4, PARTS ? 3 : 4))) void k_filter_blk4096(Blk4096Args a) {
    __shared__ __attribute__((aligned(16))) f2 img[kImg];
    __shared__ __attribute__((aligned(16))) f2 tab[kTab];
    const int j = threadIdx.x;
    const int hop = 4096 - a.V;
    constexpr int esz = OUT16 ? 4 : 8;

    // One block per workgroup.  Workgroups b, b + 8, .. share an XCD; neighbouring blocks (which share V
    // samples) go to one XCD, the XCDs work side by side in a moving window of 8 x 16 blocks.
    constexpr unsigned W = 16;
    const unsigned grp = blockIdx.x / (8 * W), rem = blockIdx.x % (8 * W);
    const unsigned blk = grp * 8 * W + (rem & 7) * W + (rem >> 3);
    if (blk >= a.nblocks) return;
    const long b0 = a.e0 - a.V + (long)blk * hop;

    // the block's samples: v[k] = x[b0 - delay + j + 256 k]
    f2 v[16];
    auto load_block = [&](long delay) {
        const long bs = b0 - delay;
        if (blk >= a.blk_lo && blk < a.blk_hi && bs >= 0) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.in + bs, 32768);
            // the streaming hint on every load (RR_V_F4K_TAIL: without it on the block's last V samples, which the next
            // block reads again - no fewer bytes fetched, see above)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (RR_V_F4K_TAIL && 256 * (k + 1) > hop) v[k] = buf_ld_f2<0>(rs, 8u * j, 2048u * k);
                else v[k] = buf_ld_f2<2>(rs, 8u * j, 2048u * k);
            }
        } else {
            // edges: the previous chunk in front (none after a reset), nothing behind the input
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long pos = bs + j + 256 * k;
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
        }
    };
    load_block(0);
    // twiddles: pass 1 e^{-j 2 pi (j mod 16) k / 256} = tw[16 (j mod 16) k] from the LDS table (filled below, read
    // after the first exchange's barrier); pass 2 tw[j]^k by a product tree
    f2 s2;
    {
        const float2 t2 = a.tw[j];
        s2 = (f2){t2.x, t2.y};
    }
#if RR_V_F4KTAB
    if (j < kTab) {
        const int r = j / 15, k = j - 15 * r + 1;
        const float2 t = a.tw[(16 * r * k) & 4095];
        tab[j] = (f2){t.x, t.y};
    }
#else
    f2 s1;
    {
        const float2 t1 = a.tw[16 * (j & 15)];
        s1 = (f2){t1.x, t1.y};
    }
#endif

    // The next call's history = the last samples of [ hist | in ]: written by the workgroup of the last block
    // (its own loads are on the way, the copy rides along).
    if (a.hist_out && blk == a.nblocks - 1) {
        for (int i = j; i < a.hist_out_len; i += 256) {
            const long pos = a.n_in - a.hist_out_len + i;
            float2 h;
            h.x = 0.f;
            h.y = 0.f;
            if (pos >= 0) h = a.in[pos];
            else if (pos >= -(long)a.hist_len) h = a.hist[a.hist_len + pos];
            a.hist_out[i] = h;
        }
    }

    // lane terms of the four access patterns; everything else is an immediate offset
    // First exchange: rows of 16 padded to 17 (pad16) - its stores, 16 elements apart from lane to lane, need that; its reads pay
    // a second cycle per half-wave for the pad inside their 32 elements.  Second exchange: NO padding - its stores go in groups of
    // 16 lanes = 16 neighbouring elements whatever the layout, and its reads then find their 32 elements in one piece.
    const f2 *const rd = img + (j + (j >> 4));               // pad16(j + 256 k) = rd + 272 k
    f2 *const w0 = img + 17 * j;                              // pad16(16 j + k)  = w0 + k
    f2 *const w1 = img + ((j >> 4) * 256 + (j & 15));         // (j / 16) 256 + j % 16 + 16 k = w1 + 16 k
    const f2 *const rd1 = img + j;                            // j + 256 k = rd1 + 256 k
    [[maybe_unused]] const f2 *const trow = tab + 15 * (j & 15) - 1;  // W_256^((j mod 16) k) = trow[k]
    const __amdgpu_buffer_rsrc_t rsG = rsrc_of(a.G, PARTS ? 32768u * (unsigned)a.nparts : (G16 ? 16384u : 32768u));

    // forward DFT_4096: in v[k] = x[j + 256 k], out v[k] = X[j + 256 k]; `late` runs in front of the last
    // butterflies, where few registers are live (the place to request G)
    auto transform = [&](bool pre_barrier, auto &&late) {
        dft16(v);
        if (pre_barrier) lds_bar();  // the previous transform's last reads are done
#pragma unroll
        for (int k = 0; k < 16; ++k) img_st(w0 + k, v[k]);
        lds_bar();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = img_ld(rd + 272 * k);
#if RR_V_F4KTAB
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], lds_ldv(trow + k));
#else
        twiddle16(v, s1);
#endif
        dft16(v);
        lds_bar();
#pragma unroll
        for (int k = 0; k < 16; ++k) img_st(w1 + 16 * k, v[k]);
        lds_bar();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = img_ld(rd1 + 256 * k);
        twiddle16(v, s2);
        late();
        dft16(v);
    };

    // the lane's 16 G values: 8 reads of 16 (8) bytes, requested in front of the forward transform's last butterflies
    float4 g4[8];
    auto forward_times_g = [&](bool pre_barrier, unsigned gofs) {
        transform(pre_barrier, [&] {
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) {
                if constexpr (G16) {
                    const f2 raw = buf_ld_f2<0>(rsG, 8u * j, 2048u * kp);
                    g4[kp] = float4{raw.x, raw.y, 0.f, 0.f};
                } else {
                    g4[kp] = buf_ld_f4<0>(rsG, 16u * j + gofs, 4096u * kp);
                }
            }
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
    };
    if constexpr (PARTS) {
        f2 acc[16];
        forward_times_g(false, 0u);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = v[k];
        for (int pt = 1; pt < a.nparts; ++pt) {
            load_block(2048L * pt);
            forward_times_g(true, 32768u * (unsigned)pt);
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] += v[k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = acc[k];
    } else {
        forward_times_g(false, 0u);
    }
    transform(true, [] {});

    if constexpr (SEL) {
        // F(t) = base0 + blk hop rb + t rb for the block's element t (position blk hop + t - V), base0 = pos + V (ra - rb): element t
        // releases iff (F mod ra) + rb >= ra and then carries m = floor(F / ra) - V
        const SelectArgs &sel = a.sel;
        const double dra = (double)sel.ra;
        auto reduce = [&](double x, unsigned &r) -> unsigned {  // x = q ra + r, exact
            double q = __builtin_floor(x * sel.inv_ra);
            double rr = __builtin_fma(-q, dra, x);
            if (rr < 0.0) { rr += dra; q -= 1.0; }
            if (rr >= dra) { rr -= dra; q += 1.0; }
            r = (unsigned)rr;
            return (unsigned)q;
        };
        unsigned Rb;
        const unsigned qb = reduce((double)sel.base_r + (double)blk * (double)sel.hr, Rb);
        const long mb = (long)sel.base_q + (long)blk * (long)sel.hq + (long)qb - a.V;  // m of a release at the block's F = Rb
        const long lft = a.n_out - mb;
        const unsigned rec = (unsigned)(lft < 8192 ? (lft > 0 ? lft : 0) : 8192) * 8u;
        const __amdgpu_buffer_rsrc_t rsel = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float2 *>(a.out) + mb, 0, rec, 0x00020000);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (256 * k > hop) continue;  // (no lane's index reaches V)
            const int t = (4096 - j - 256 * k) & 4095;
            unsigned R;
            const unsigned C = reduce((double)Rb + (double)t * (double)sel.rb, R);
            const bool em = t >= a.V && R + sel.rb >= sel.ra;
            __builtin_amdgcn_raw_buffer_store_b64(v[k], rsel, em ? C * 8u : 0xffffffffu, 0, 0);
        }
        return;
    }
    // y[t] = v[k] with t = (4096 - (j + 256 k)) mod 4096; valid for t >= V: output mbase + t - V.
    // Offsets in the block's output window: (hop - j - 256 k) elements; t = 0 lands on `hop` (past the
    // window), t < V wraps to a huge offset: the descriptor's range check drops both.
    const long mbase = (long)blk * hop;
    const long left = a.n_out - mbase;
    const unsigned recs = (unsigned)(left < hop ? left : hop) * (unsigned)esz;
    char *obase = reinterpret_cast<char *>(a.out) + mbase * esz;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase, 0, recs, 0x00020000);
    const unsigned lane_off = (unsigned)(hop - j) * (unsigned)esz;
    if constexpr (ACC && !OUT16) {
        f2 old[16];  // (lanes outside the block's valid part read zeros and store nothing)
#pragma unroll
        for (int k = 0; k < 16; ++k) old[k] = buf_ld_f2<0>(rs, lane_off - (unsigned)(256 * k * esz), 0);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += old[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned off = lane_off - (unsigned)(256 * k * esz);
        // lane j's value k is y[4096 - j - 256 k]: with 256 k > hop no lane's index reaches V (n = 1024: k = 13, 14, 15) - the
        // descriptor's range check would drop all 256 lanes; a uniform branch drops the instruction
        if (RR_V_F4K_DEAD && 256 * k > hop) continue;
        if constexpr (OUT16) {
            const __half2 h = __floats2half2_rn(v[k].x, v[k].y);
            __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const unsigned *>(&h), rs, off, 0, 2);
        } else {
            __builtin_amdgcn_raw_buffer_store_b64(v[k], rs, off, 0, 2);
        }
    }
}


// ---------------------------------------------------------------------------
// k_filter_blkbig<N>: the same fast convolution with blocks of N = 8192 / 16 384 points for responses of 2049 .. 8192 taps: ONE
// forward and one inverse transform per block, N - V results valid (n = 4096 with 16 384 points: 75 %) - the partitions of 2048
// taps above run n / 2048 + 1 transforms of 4096 points for 2049 results (n = 4096: 6 transformed samples per result against
// 2.7 here).  The transform: rr_fft_big.hpp (N / 16 lanes, 16 values per lane, radix 16 x 16 x 16 x N / 4096 through ONE image
// in LDS).  The inverse is the forward routine with the result index reversed, as above.
// ---------------------------------------------------------------------------
struct BlkBigArgs {
    const float2 *hist;
    int hist_len;
    const float2 *in;
    long n_in;
    const void *G;       // pair-interleaved: Gp[kp][j] = {G[j + 2 T kp], G[j + 2 T kp + T]}, j < T = N / 16
    const float2 *tw;    // e^{-j 2 pi k / N}
    int V;
    float2 *out;
    long n_out;
    long e0;
    unsigned nblocks;
    unsigned blk_lo, blk_hi;  // the blocks that lie entirely inside the input
    float2 *hist_out;
    int hist_out_len;
};

// N = 16384: 1024 lanes, last pass radix 4, one workgroup per CU (136 KiB of LDS); N = 8192: 512 lanes, last pass radix 2 (its
// butterflies i = j + 512 c, c < 8, over v[c + 8 k]), two workgroups per CU
template <int N>
__global__ __launch_bounds__(N / 16) void k_filter_blkbig(BlkBigArgs a) {
    constexpr int T = N / 16;  // lanes
    static_assert(N == 8192 || N == 16384, "blocks of 8192 or 16384 points");
    extern __shared__ __attribute__((aligned(16))) f2 dynbig[];
    f2 *const img = dynbig;
    f2 *const tab = dynbig + (N + N / 16);
    const int j = threadIdx.x;
    const int hop = N - a.V;
    // neighbouring blocks (which share V samples) on one XCD, in a window of 8 x 4 blocks
    constexpr unsigned W = 4;
    const unsigned grp = blockIdx.x / (8 * W), rem = blockIdx.x % (8 * W);
    const unsigned blk = grp * 8 * W + (rem & 7) * W + (rem >> 3);
    if (blk >= a.nblocks) return;
    const long b0 = a.e0 - a.V + (long)blk * hop;

    f2 v[16];  // v[k] = x[b0 + j + T k]
    if (blk >= a.blk_lo && blk < a.blk_hi) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.in + b0, 8u * N);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = buf_ld_f2<2>(rs, 8u * j, 8u * T * k);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long pos = b0 + j + T * k;
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
    }
    BigFftLane<N> ln;
    ln.init(a.tw, tab, j);
    if (a.hist_out && blk == a.nblocks - 1) {  // the next call's history = the last samples of [ hist | in ]
        for (int i = j; i < a.hist_out_len; i += T) {
            const long pos = a.n_in - a.hist_out_len + i;
            float2 h;
            h.x = 0.f;
            h.y = 0.f;
            if (pos >= 0) h = a.in[pos];
            else if (pos >= -(long)a.hist_len) h = a.hist[a.hist_len + pos];
            a.hist_out[i] = h;
        }
    }

    const __amdgpu_buffer_rsrc_t rsG = rsrc_of(a.G, 8u * N);
    // the lane's 16 G values: 8 reads of 16 bytes; 8192 points: requested in front of the forward transform's last butterflies; 16 384
    // points (128 registers per lane at 1024 lanes): in two halves behind it - kept through the last pass they spilled 10 registers,
    // and scratch is HBM traffic
    float4 g4[8];
    big_fft<N, true>(v, img, ln, j, false, [&] {
        if constexpr (N == 8192) {
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) g4[kp] = buf_ld_f4<0>(rsG, 16u * j, 16u * T * kp);
        }
    });
    if constexpr (N == 8192) {
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
            v[2 * kp] = cmul(v[2 * kp], (f2){g4[kp].x, g4[kp].y});
            v[2 * kp + 1] = cmul(v[2 * kp + 1], (f2){g4[kp].z, g4[kp].w});
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) g4[kp] = buf_ld_f4<0>(rsG, 16u * j, 16u * T * (4 * h + kp));
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                v[8 * h + 2 * kp] = cmul(v[8 * h + 2 * kp], (f2){g4[kp].x, g4[kp].y});
                v[8 * h + 2 * kp + 1] = cmul(v[8 * h + 2 * kp + 1], (f2){g4[kp].z, g4[kp].w});
            }
        }
    }
    big_fft<N, true>(v, img, ln, j, true, [] {});

    // y[t] = v[k] with t = (N - (j + T k)) mod N; valid for t >= V: output mbase + t - V = mbase + hop - j - T k
    // (t = 0 lands on `hop`, t < V wraps to a huge offset: the descriptor's range check drops both)
    const long mbase = (long)blk * hop;
    const long left = a.n_out - mbase;
    const unsigned recs = (unsigned)(left < hop ? left : hop) * 8u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out + mbase, 0, recs, 0x00020000);
    const unsigned lane_off = (unsigned)(hop - j) * 8u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (T * k > hop) continue;  // (no lane's index reaches V: a uniform branch drops the instruction)
        __builtin_amdgcn_raw_buffer_store_b64(v[k], rs, lane_off - (unsigned)(T * k * 8), 0, 2);
    }
}

}  // namespace

int launch_filter_blk4096(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G,
                          const void *tw4096, size_t n, void *out, size_t n_out, long e0, bool out_f16, bool g_f16,
                          void *hist_out, size_t hist_out_len, bool accumulate, size_t nparts) {
    if (n_out == 0) return RR_OK;
    if (nparts > 1 && (out_f16 || g_f16 || accumulate)) RR_FAIL(RR_ERR_BAD_ARG, "Filter: the partitioned form is f32 only");
    if (accumulate && (out_f16 || g_f16)) RR_FAIL(RR_ERR_BAD_ARG, "Filter: partitions accumulate in f32");
    Blk4096Args a;
    a.hist = (const float2 *)hist;
    a.hist_len = (int)hist_len;
    a.in = (const float2 *)in;
    a.n_in = (long)n_in;
    a.G = G;
    a.tw = (const float2 *)tw4096;
    a.V = (int)n;  // V >= n - 1; the tables are laid out for any V <= 2048
    a.out = out;
    a.n_out = (long)n_out;
    a.e0 = e0;
    a.hist_out = (float2 *)hist_out;
    a.hist_out_len = (int)hist_out_len;
    a.nparts = (int)(nparts ? nparts : 1);
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
    const unsigned grid = (unsigned)((nblocks + 127) / 128 * 128);
    if (nparts > 1) {
        hipLaunchKernelGGL((k_filter_blk4096<false, false, false, true>), dim3(grid), dim3(256), 0, s, a);
    } else if (accumulate) {
        hipLaunchKernelGGL((k_filter_blk4096<false, false, true>), dim3(grid), dim3(256), 0, s, a);
    } else if (out_f16) {
        if (g_f16) hipLaunchKernelGGL((k_filter_blk4096<true, true, false>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_filter_blk4096<true, false, false>), dim3(grid), dim3(256), 0, s, a);
    } else {
        if (g_f16) hipLaunchKernelGGL((k_filter_blk4096<false, true, false>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_filter_blk4096<false, false, false>), dim3(grid), dim3(256), 0, s, a);
    }
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// The Downsampler through k_filter_blk4096<.., SEL>: out[m] = sum_i c[i] x[e_m - i] for the releases e_m of the periodic schedule
// (ra, rb, pos) among the n_in positions of the call; G / tw4096 = the tables of c = reverse(ir) as the Filter's, L <= 2048 taps.
bool decim_select_blk_supported(int dtype, uint64_t ra, uint64_t rb, size_t L) {
    return dtype == RR_F32 && L >= 2 && L <= 2048 && rb >= 1 && rb < ra && ra < (1ull << 31);
}
int launch_decim_select_blk(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G, const void *tw4096,
                            size_t L, void *out, size_t n_out, uint64_t ra, uint64_t rb, uint64_t pos) {
    if (n_out == 0 || n_in == 0) return RR_OK;
    if (!decim_select_blk_supported(RR_F32, ra, rb, L) || pos >= ra)
        RR_FAIL(RR_ERR_BAD_ARG, "Downsampler (select, 4096-point blocks): rates %llu : %llu, pos %llu, %zu taps", (unsigned long long)ra,
                (unsigned long long)rb, (unsigned long long)pos, L);
    Blk4096Args a;
    a.hist = (const float2 *)hist;
    a.hist_len = (int)hist_len;
    a.in = (const float2 *)in;
    a.n_in = (long)n_in;
    a.G = G;
    a.tw = (const float2 *)tw4096;
    a.V = (int)L;  // V >= L - 1
    a.out = out;
    a.n_out = (long)n_out;
    a.e0 = 0;
    a.hist_out = nullptr;
    a.hist_out_len = 0;
    a.nparts = 1;
    const size_t hop = 4096 - a.V;
    const size_t nblocks = (n_in + hop - 1) / hop;  // the response at every position of the call
    if (nblocks > 0x3fffffull) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler (select): too many blocks");
    a.nblocks = (unsigned)nblocks;
    {
        const long first = -(long)a.V;
        long lo = (-first + (long)hop - 1) / (long)hop;
        long hi = ((long)n_in - 4096 - first) >= 0 ? ((long)n_in - 4096 - first) / (long)hop + 1 : 0;
        if (hi > (long)nblocks) hi = (long)nblocks;
        if (lo > hi) lo = hi;
        a.blk_lo = (unsigned)lo;
        a.blk_hi = (unsigned)hi;
    }
    SelectArgs &q = a.sel;
    q.ra = (uint32_t)ra;
    q.rb = (uint32_t)rb;
    q.inv_ra = 1.0 / (double)ra;
    q.kr = q.kq = 0;  // (k_filter_wave<true>'s steps of 128 positions: not used here)
    q.hr = (uint32_t)((hop * rb) % ra);
    q.hq = (uint32_t)((hop * rb) / ra);
    const uint64_t base0 = pos + (uint64_t)a.V * (ra - rb);
    q.base_r = (uint32_t)(base0 % ra);
    q.base_q = (uint32_t)(base0 / ra);
    const unsigned grid = (unsigned)((nblocks + 127) / 128 * 128);
    hipLaunchKernelGGL((k_filter_blk4096<false, false, false, false, true>), dim3(grid), dim3(256), 0, s, a);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

bool filter_blkbig_supported(int dtype, size_t n) { return dtype == RR_F32 && n >= 2 && n - 1 <= 8192; }

template <int N>
static int launch_filter_blkbig_n(hipStream_t s, BlkBigArgs &a, size_t n_in, size_t n_out, long e0) {
    const size_t hop = N - a.V;
    const size_t nblocks = (n_out + hop - 1) / hop;
    if (nblocks > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Filter: too many blocks");
    a.nblocks = (unsigned)nblocks;
    {
        const long first = e0 - a.V;
        long lo = first >= 0 ? 0 : (-first + (long)hop - 1) / (long)hop;
        long hi = ((long)n_in - N - first) >= 0 ? ((long)n_in - N - first) / (long)hop + 1 : 0;
        if (hi > (long)nblocks) hi = (long)nblocks;
        if (lo > hi) lo = hi;
        a.blk_lo = (unsigned)lo;
        a.blk_hi = (unsigned)hi;
    }
    constexpr size_t lds = (size_t)big_fft_lds_elems<N>() * sizeof(f2);
    RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(k_filter_blkbig<N>), lds));
    const unsigned grid = (unsigned)((nblocks + 31) / 32 * 32);
    hipLaunchKernelGGL(k_filter_blkbig<N>, dim3(grid), dim3(N / 16), lds, s, a);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// N = 8192 or 16384 points per block, overlap V >= n - 1 (V < N); G = DFT_N(g) / N pair-interleaved, twN = e^{-j 2 pi k / N}
int launch_filter_blkbig(hipStream_t s, size_t N, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G,
                         const void *twN, size_t V, void *out, size_t n_out, long e0, void *hist_out, size_t hist_out_len) {
    if (n_out == 0) return RR_OK;
    if ((N != 8192 && N != 16384) || V < 1 || V > N / 2) RR_FAIL(RR_ERR_BAD_ARG, "Filter: overlap %zu / blocks of %zu points", V, N);
    BlkBigArgs a;
    a.hist = (const float2 *)hist;
    a.hist_len = (int)hist_len;
    a.in = (const float2 *)in;
    a.n_in = (long)n_in;
    a.G = G;
    a.tw = (const float2 *)twN;
    a.V = (int)V;
    a.out = (float2 *)out;
    a.n_out = (long)n_out;
    a.e0 = e0;
    a.hist_out = (float2 *)hist_out;
    a.hist_out_len = (int)hist_out_len;
    return N == 8192 ? launch_filter_blkbig_n<8192>(s, a, n_in, n_out, e0) : launch_filter_blkbig_n<16384>(s, a, n_in, n_out, e0);
}

}  // namespace rr
