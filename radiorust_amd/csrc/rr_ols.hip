// rr_ols.hip — the overlap-save (and direct-form) kernels of the chain's FIR stage and of the Filter block, Complex<f32>, gfx950:
//   k_mix_fir_decim   mix + combined FIR + decimate, direct form (round 1's kernel, on request)
//   k_ols_decim4      the same by overlap-save, a workgroup per 4096-block (long combined responses)
//   k_ols_wave(_bank) ... a WAVE per 1024-block, decimation 2 / 4 / 8, polyphase forward transform: the FIR stage of every chain
//                     shape and of the stand-alone Downsampler; _bank: the channels of an rr_chainbank in one launch
//   k_ols_frame       k_ols_wave's blocks + the 4096-point Fourier stage (+ metering::bandwidth) in ONE kernel: the benchmark's
//   k_filter_wave     the Filter alone (n <= 385), a wave per 1024-block
// (split out of rr_fused.hip in round 3; the kernels' derivations and the variants measured and dropped: DESIGN_HISTORY.md 4)
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"
#include "rr_meter_dev.hpp"
#include "rr_fft_regs.hpp"
#include "rr_ols_dev.hpp"

#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdlib>
#include <utility>
#include <vector>

namespace rr {

// ---------------------------------------------------------------------------
// geometry shared by host and device
// ---------------------------------------------------------------------------
template <int D, int R> struct FirGeom {
    static constexpr int RD = R * D;                       // samples per LDS row
    static constexpr int ROW_BYTES = RD * 8;
    // row stride: odd multiple of 16 B -> the 16 lanes of a ds_read_b128 group
    // hit 16 different 16-B slots of the 256-B bank row
    static constexpr int STRIDE = ROW_BYTES + (((ROW_BYTES / 16) & 1) ? 32 : 16);
    static constexpr int OUT_STRIDE = R * 8 + 16;          // staged outputs per lane
};

// pairs of samples each lane prefetches per tile (upper bound of the real count)
constexpr int kNPF = 18;  // R*D = 32 samples per lane row
// generic: enough 16-B pairs per lane for a tile of T rows of RD samples plus up to 16 halo rows
constexpr int npf_for(int RD) { return RD * 9 / 16; }


__device__ __forceinline__ void lds_barrier() {
    // Measured on MI355X: the plain barrier, which also drains vmcnt, is 8 % FASTER here than
    // a raw s_barrier + lgkmcnt(0) that lets the prefetch and the stores stay in flight.
    __syncthreads();
}

template <int D, int R, int T>
__global__ __launch_bounds__(T, 2) void k_mix_fir_decim(const float2 *__restrict__ xh, int hx,
                                                        const float2 *__restrict__ in, long n_in, int in_aligned16,
                                                        const float2 *__restrict__ nco, unsigned denom, unsigned idx0,
                                                        const float *__restrict__ taps, int Gp,
                                                        float2 *__restrict__ out, long n_out, int out_aligned16,
                                                        long e0, unsigned ntiles, unsigned tiles_per_wg,
                                                        float2 *__restrict__ xh_out, int hx_out) {
    using G = FirGeom<D, R>;
    constexpr int RD = G::RD, STRIDE = G::STRIDE;
    constexpr int OUTS = T * R;
    static_assert((RD == 32 || RD == 16) && (2 * T) % RD == 0, "LDS write addresses advance by whole rows per prefetch slot");
    constexpr int ROWS_PER_SLOT = 2 * T / RD;
    constexpr int NPF = npf_for(RD);        // 18 for RD = 32, 9 for RD = 16
    constexpr int LOG2_RD = RD == 32 ? 5 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // 32 spare bytes in front: when the tile starts at an odd sample, the first lane's
    // pair straddles the tile start and its first half (sample -1 = row -1, column 31)
    // lands there; slack rows at the end take the pairs past the tile end
    char *smem = smem_raw + 32;
    const int rows = T + (Gp + R - 2) / R;  // the last lane reads blocks up to R*(T-1) + Gp + R - 2
    const int NS = rows * RD;
    const int lds_rows = rows + ROWS_PER_SLOT + 1;
    float *tap_lds = reinterpret_cast<float *>(smem + (size_t)lds_rows * STRIDE);
    for (int i = threadIdx.x; i < Gp * D; i += T) tap_lds[i] = taps[i];

    // XCD-aware work split: workgroups b, b+8, b+16.. share an XCD (round robin);
    // give each workgroup a contiguous run of tiles and neighbouring runs to one XCD,
    // so the halo a tile shares with its predecessor is an L2 hit.
    unsigned chunk;
    {
        const unsigned b = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, rmd = nwg & 7, xcd = b & 7;
        chunk = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (b >> 3);
    }
    // an XCD owns a contiguous range of tiles and its workgroups take them round robin, so that the
    // tiles in flight at any moment are neighbours in memory (grid: multiple of 8); contiguous runs
    // per workgroup (RR_V_CONTIG) measured 3 % slower
    const unsigned tstride = gridDim.x >> 3;
    const unsigned per_xcd_t = (ntiles + 7) >> 3;
    const unsigned tile_begin = (blockIdx.x & 7) * per_xcd_t + (blockIdx.x >> 3);
    unsigned tile_end = ((blockIdx.x & 7) + 1) * per_xcd_t;
    if (tile_end > ntiles) tile_end = ntiles;
    (void)tiles_per_wg;
    // The workgroup that owns the last run of tiles also leaves the mixed-sample
    // history for the next call: xh_out = the last hx_out mixed samples of this call.
    if (xh_out && chunk == gridDim.x - 1) {
        for (int i = threadIdx.x; i < hx_out; i += T) {
            const long pos = n_in - hx_out + i;
            float2 v;
            if (pos >= 0) {
                const float2 xx = in[pos];
                const float2 pp = nco[(unsigned)(((long)idx0 + pos) % (long)denom)];
                v.x = xx.x * pp.x - xx.y * pp.y;
                v.y = xx.x * pp.y + xx.y * pp.x;
            } else {
                v = (pos >= -(long)hx) ? xh[hx + pos] : float2{0.f, 0.f};
            }
            xh_out[i] = v;
        }
    }
    if (tile_begin >= tile_end) return;

    // per-lane constants of the load phase -------------------------------------
    const long lo0 = e0 - (long)D * Gp + 1;            // tile_lo of tile 0
    const int odd = (int)(lo0 & 1);                    // same for every tile (D*OUTS is even)
    const int npairs = (NS + odd + 1) >> 1;
    const int sfirst = 2 * (int)threadIdx.x - odd;     // LDS sample index of this lane's first prefetched sample
    // byte address of sample s (floor division, so s = -1 is row -1, column 31 = -24)
    auto lds_addr = [&](int s) -> int { return (s >> LOG2_RD) * STRIDE + (s & (RD - 1)) * 8; };
    const int a0 = lds_addr(sfirst), a1 = lds_addr(sfirst + 1);
    const unsigned step = (unsigned)((2 * T) % denom);
    const unsigned tstep = (unsigned)(((long)D * OUTS * tstride) % denom);
    const bool nco_const = (step == 0 && tstep == 0);  // phasor of a lane never changes (e.g. denom = 8)

    auto tile_lo_of = [&](unsigned tile) -> long { return lo0 + (long)D * OUTS * tile; };
    auto interior_of = [&](long tile_lo) -> bool {
        const long le = tile_lo - odd;
        return in_aligned16 && le >= 0 && le + 2L * NPF * T <= n_in;  // the whole prefetch window is inside `in`
    };
    long tile_lo = tile_lo_of(tile_begin);
    unsigned rbase;
    {
        long ph = ((long)idx0 + (tile_lo - odd) + 2 * (long)threadIdx.x) % (long)denom;
        if (ph < 0) ph += denom;
        rbase = (unsigned)ph;
    }
    float2 pc0 = nco[rbase], pc1 = nco[(rbase + 1 == denom) ? 0 : rbase + 1];

    f4 x[NPF];
    auto prefetch = [&](long tlo) {
        const f4 *src = reinterpret_cast<const f4 *>(in + (tlo - odd)) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < NPF; ++u) x[u] = src[u * T];
    };
    bool cur_interior = interior_of(tile_lo);
    if (cur_interior) prefetch(tile_lo);

    for (unsigned tile = tile_begin; tile < tile_end; tile += tstride) {
        // ---- stage: (prefetched) raw samples -> mix -> LDS; each prefetch slot is
        //      re-issued for the next tile as soon as it has been consumed ----------
        const long next_lo = tile_lo + (long)D * OUTS * tstride;
        const bool next_interior = (tile + tstride < tile_end) && interior_of(next_lo);
        // x * p = x.re * (p.re, p.im) + x.im * (-p.im, p.re): one packed mul + one packed fma
        auto mix = [](f2 xv, f2 p, f2 pj) -> f2 { return __builtin_elementwise_fma(xv.yy, pj, xv.xx * p); };
        if (cur_interior) {
            if (nco_const) {
                const f2 q0 = {pc0.x, pc0.y}, q0j = {-pc0.y, pc0.x}, q1 = {pc1.x, pc1.y}, q1j = {-pc1.y, pc1.x};
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const bool ok = (int)threadIdx.x + u * T < npairs;
                    const f2 v0 = mix(x[u].xy, q0, q0j), v1 = mix(x[u].zw, q1, q1j);
                    if (ok) {
                        *reinterpret_cast<f2 *>(smem + a0 + u * ROWS_PER_SLOT * STRIDE) = v0;
                        *reinterpret_cast<f2 *>(smem + a1 + u * ROWS_PER_SLOT * STRIDE) = v1;
                    }
                }
            } else {
                unsigned rr_ = rbase;
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const bool ok = (int)threadIdx.x + u * T < npairs;
                    const float2 p0 = nco[rr_], p1 = nco[(rr_ + 1 == denom) ? 0 : rr_ + 1];
                    const f2 v0 = mix(x[u].xy, (f2){p0.x, p0.y}, (f2){-p0.y, p0.x});
                    const f2 v1 = mix(x[u].zw, (f2){p1.x, p1.y}, (f2){-p1.y, p1.x});
                    if (ok) {
                        *reinterpret_cast<f2 *>(smem + a0 + u * ROWS_PER_SLOT * STRIDE) = v0;
                        *reinterpret_cast<f2 *>(smem + a1 + u * ROWS_PER_SLOT * STRIDE) = v1;
                    }
                    rr_ += step;
                    if (rr_ >= denom) rr_ -= denom;
                }
            }
        } else {
            // edge tiles: history (already mixed), end of input, unaligned input
            unsigned rr_ = rbase;
            const long lo_even = tile_lo - odd;
            for (int pi = threadIdx.x; pi < npairs; pi += T) {
                const long pe = lo_even + 2 * (long)pi;
                const unsigned r1 = (rr_ + 1 == denom) ? 0 : rr_ + 1;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const long pos = pe + k;
                    f2 v = {0.f, 0.f};
                    if (pos < 0) {
                        if (pos >= -(long)hx) {
                            const float2 h = xh[hx + pos];
                            v = (f2){h.x, h.y};
                        }
                    } else if (pos < n_in) {
                        const float2 xx = in[pos];
                        const float2 pp = nco[k ? r1 : rr_];
                        v = (f2){xx.x * pp.x - xx.y * pp.y, xx.x * pp.y + xx.y * pp.x};
                    }
                    const int sidx = (int)(pos - tile_lo);
                    if (sidx >= 0 && sidx < NS) *reinterpret_cast<f2 *>(smem + lds_addr(sidx)) = v;
                }
                rr_ += step;
                if (rr_ >= denom) rr_ -= denom;
            }
            if (next_interior) prefetch(next_lo);
        }
        lds_barrier();
        // prefetch the next tile once the stage barrier is passed (re-issuing each slot inside
        // the stage loop, or in slices between FIR rounds, measured 4-14 % slower)
        if (cur_interior && next_interior) prefetch(next_lo);

        // ---- FIR: rotating register window, packed FMAs --------------------------
        f2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = (f2){0.f, 0.f};
        {
            f2 W[R][D];
            const char *lane = smem + (size_t)threadIdx.x * STRIDE;
            // block b of this lane lives at row b / R, column block b % R
            auto load_block = [&](const char *row0, int b_static, f2(&dst)[D]) {
                const char *p = row0 + (b_static / R) * STRIDE + (b_static % R) * (D * 8);
#pragma unroll
                for (int q = 0; q < D / 2; ++q) {
                    const f4 v = *reinterpret_cast<const f4 *>(p + 16 * q);
                    dst[2 * q] = (f2){v.x, v.y};
                    dst[2 * q + 1] = (f2){v.z, v.w};
                }
            };
#pragma unroll
            for (int b = 0; b < R; ++b) load_block(lane, b, W[b]);
            // one tap group: read block (t + R - 1) and D taps, R*D packed FMAs
            // taps are read half a round (R/2 groups of D) ahead into registers
            constexpr int HT = (R / 2) * D;  // floats per half round
            auto load_taps = [&](const float *tp, float(&ct)[HT]) {
#pragma unroll
                for (int q = 0; q < HT / 4; ++q) {
                    const f4 t4 = *reinterpret_cast<const f4 *>(tp + 4 * q);
                    ct[4 * q] = t4.x;
                    ct[4 * q + 1] = t4.y;
                    ct[4 * q + 2] = t4.z;
                    ct[4 * q + 3] = t4.w;
                }
            };
            // one tap group: D taps, R*D packed FMAs; output 0 is the last user of block t
            // (slot ti): once it is done the slot takes block t + R, which is not needed before
            // output R-1 of the NEXT step -- two steps of FMAs cover the LDS latency
            auto fir_step = [&](auto TI, const char *row0, const float(&ct)[HT]) {
                constexpr int ti = decltype(TI)::value;
                constexpr int tl = ti % (R / 2);
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    const f2 cc = {ct[tl * D + q], ct[tl * D + q]};
                    acc[0] = __builtin_elementwise_fma(W[ti][q], cc, acc[0]);
                }
                load_block(row0, ti + R, W[ti]);
#pragma unroll
                for (int r = 1; r < R; ++r) {
#pragma unroll
                    for (int q = 0; q < D; ++q) {
                        const f2 cc = {ct[tl * D + q], ct[tl * D + q]};
                        acc[r] = __builtin_elementwise_fma(W[(ti + r) % R][q], cc, acc[r]);
                    }
                }
            };
            const int nouter = Gp / R, rem = Gp % R;
            float ca[HT], cb[HT];
            load_taps(tap_lds, ca);
            for (int to = 0; to < nouter; ++to) {
                const char *row0 = lane + (size_t)to * STRIDE;
                const float *tp = tap_lds + to * RD;
                load_taps(tp + HT, cb);
                [&]<int... I>(std::integer_sequence<int, I...>) {
                    (fir_step(std::integral_constant<int, I>{}, row0, ca), ...);
                }(std::make_integer_sequence<int, R / 2>{});
                load_taps(tp + 2 * HT, ca);  // first half of the next round (a spare round of taps is allocated)
                [&]<int... I>(std::integer_sequence<int, I...>) {
                    (fir_step(std::integral_constant<int, I + R / 2>{}, row0, cb), ...);
                }(std::make_integer_sequence<int, R / 2>{});
            }
            if (rem) {  // the last, partial round (workgroup-uniform)
                const char *row0 = lane + (size_t)nouter * STRIDE;
                const float *tp = tap_lds + nouter * RD;
                load_taps(tp + HT, cb);
                [&]<int... I>(std::integer_sequence<int, I...>) {
                    ((I < rem ? fir_step(std::integral_constant<int, I>{}, row0, ca) : (void)0), ...);
                }(std::make_integer_sequence<int, R / 2>{});
                [&]<int... I>(std::integer_sequence<int, I...>) {
                    ((I + R / 2 < rem ? fir_step(std::integral_constant<int, I + R / 2>{}, row0, cb) : (void)0), ...);
                }(std::make_integer_sequence<int, R / 2 - 1>{});
            }
        }
        lds_barrier();  // every wave is done reading this tile's samples

        // ---- store: R consecutive outputs per lane ---------------------------------
        {
            const long m0 = (long)tile * OUTS + (long)threadIdx.x * R;
            float2 *o = out + m0;
            if (out_aligned16 && m0 + R <= n_out) {
#pragma unroll
                for (int r = 0; r < R; r += 2)
                    *reinterpret_cast<f4 *>(o + r) = (f4){acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y};
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (m0 + r < n_out) {
                        float2 w;
                        w.x = acc[r].x;
                        w.y = acc[r].y;
                        o[r] = w;
                    }
            }
        }
        tile_lo = next_lo;
        cur_interior = next_interior;
        rbase += tstep;
        if (rbase >= denom) rbase -= denom;
    }
}

template <int D, int R, int T>
static int launch_mfd(hipStream_t s, const FusedFirArgs &a) {
    using G = FirGeom<D, R>;
    constexpr int OUTS = T * R;
    const int rows = T + (a.Gp + R - 2) / R;
    const int lds_rows = rows + 2 * T / G::RD + 1;
    const size_t lds = 32 + (size_t)lds_rows * G::STRIDE + ((size_t)a.Gp * D + 3 * G::RD) * sizeof(float);
    if ((rows * G::RD + 2) / 2 > npf_for(G::RD) * T) RR_FAIL(RR_ERR_BAD_ARG, "fused FIR: %d tap groups exceed the prefetch window", a.Gp);
    auto fn = k_mix_fir_decim<D, R, T>;
    RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(fn), lds));
    const size_t ntiles = (a.n_out + OUTS - 1) / OUTS;
    if (ntiles > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fused FIR: too many tiles");
    // persistent grid: 8 waves per CU (LDS-limited), 256 CUs
    size_t nwg = 256 * (512 / T) * (32 / G::RD);  // LDS-limited: 8 waves/CU at 32 samples per lane, 12-16 at 16
    if (nwg > ntiles) nwg = ntiles;
    nwg = (nwg + 7) / 8 * 8;
    const size_t tpw = 0;
    const int in_al = (reinterpret_cast<uintptr_t>(a.in) % 16 == 0) ? 1 : 0;
    const int out_al = (reinterpret_cast<uintptr_t>(a.out) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(T), lds, s, (const float2 *)a.xh, (int)a.hx, (const float2 *)a.in,
                       (long)a.n_in, in_al, (const float2 *)a.nco, a.denom, a.idx0, (const float *)a.taps, a.Gp,
                       (float2 *)a.out, (long)a.n_out, out_al, (long)a.e0, (unsigned)ntiles, (unsigned)tpw,
                       (float2 *)a.xh_out, (int)a.hx);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

bool fused_fir_supported(uint64_t D, size_t Lc) {
    if (!(D == 2 || D == 4 || D == 8) || Lc == 0) return false;
    // tap groups (padded to a multiple of R) must fit the per-lane prefetch window
    const int R = fused_fir_R(D);
    const size_t gp = (Lc + D - 1) / D;
    const size_t rows = 128 + (gp + R - 2) / R;
    return (rows * 32 + 2) / 2 <= (size_t)kNPF * 128;  // (T = 128 geometry; T = 256 has more slack)
}

int fused_fir_R(uint64_t D) {
    switch (D) {
        case 2: return 16;
        case 4: return 8;
        case 8: return 4;
    }
    return 0;
}

int launch_fused_fir(hipStream_t s, const FusedFirArgs &a) {
    if (a.n_out == 0) return RR_OK;
    switch (a.D) {
        case 2: return launch_mfd<2, 16, 128>(s, a);
        case 4: return launch_mfd<4, 8, 128>(s, a);
        case 8: return launch_mfd<8, 4, 128>(s, a);
    }
    RR_FAIL(RR_ERR_BAD_ARG, "fused FIR: decimation %u not instantiated", a.D);
}

// ---------------------------------------------------------------------------
// Kernel 3  k_ols_decim4: the same mix + combined FIR + 4x decimation as
// k_mix_fir_decim, computed by overlap-save fast convolution:
//   block of 4096 mixed samples -> forward DFT (radix 16 x 3, as k_fft4096)
//   -> * H  (H = DFT_4096(c) / 4096, c = reverse(ir) (*) g, real or complex)
//   -> fold the four 1024-bin quarters (decimation by 4 in time = aliasing in frequency)
//   -> inverse DFT_1024 (radix 4 x 5) -> the last (4096 - V)/4 results are valid.
// ~30 packed VALU ops per input sample instead of ~49 for the direct form, and the
// occupancy/LDS profile of k_fft4096 (34.8 KiB, 16 waves/CU).  Blocks start at
// e0 - V + b * (4096 - V), so output m of the call is sample (V/4 + i) of block b.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_ols_decim4(const float2 *__restrict__ xh, int hx,
                                                    const float2 *__restrict__ in, long n_in,
                                                    const float2 *__restrict__ nco, unsigned denom, unsigned idx0,
                                                    const float2 *__restrict__ H, const float2 *__restrict__ tw,
                                                    int V, float2 *__restrict__ out, long n_out, long e0,
                                                    float2 *__restrict__ xh_out, int hx_out) {
    __shared__ f2 lds[4096 + 256];
    const int j = threadIdx.x;
    const int hop = 4096 - V, per_block = hop >> 2;
    const long b0 = e0 - V + (long)blockIdx.x * hop;  // position of the block's first sample

    // the last workgroup also leaves the mixed-sample history for the next call
    if (xh_out && blockIdx.x == gridDim.x - 1) {
        for (int i = j; i < hx_out; i += 256) {
            const long pos = n_in - hx_out + i;
            float2 v;
            if (pos >= 0) {
                const float2 xx = in[pos];
                const float2 pp = nco[(unsigned)(((long)idx0 + pos) % (long)denom)];
                v.x = xx.x * pp.x - xx.y * pp.y;
                v.y = xx.x * pp.y + xx.y * pp.x;
            } else {
                v = (pos >= -(long)hx) ? xh[hx + pos] : float2{0.f, 0.f};
            }
            xh_out[i] = v;
        }
    }

    // ---- load + mix: v[k] = xs[b0 + j + 256 k] ----------------------------------
    f2 v[16];
    {
        const unsigned kstep = 256u % denom;
        long ph = ((long)idx0 + b0 + j) % (long)denom;
        if (ph < 0) ph += denom;
        unsigned r = (unsigned)ph;
        const bool interior = b0 >= 0 && b0 + 4096 <= n_in;
        if (interior) {
            const float2 *src = in + b0 + j;
            float2 x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = src[256 * k];
            if (kstep == 0) {  // the phasor of a lane does not change (e.g. denom = 8)
                const float2 p = nco[r];
                const f2 pp = {p.x, p.y}, pj = {-p.y, p.x};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const f2 xv = {x[k].x, x[k].y};
                    v[k] = __builtin_elementwise_fma(xv.yy, pj, xv.xx * pp);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float2 p = nco[r];
                    const f2 xv = {x[k].x, x[k].y};
                    v[k] = __builtin_elementwise_fma(xv.yy, (f2){-p.y, p.x}, xv.xx * (f2){p.x, p.y});
                    r += kstep;
                    if (r >= denom) r -= denom;
                }
            }
        } else {  // edges: history (already mixed) in front, nothing behind the input
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long pos = b0 + j + 256 * k;
                f2 t = {0.f, 0.f};
                if (pos < 0) {
                    if (pos >= -(long)hx) {
                        const float2 h = xh[hx + pos];
                        t = (f2){h.x, h.y};
                    }
                } else if (pos < n_in) {
                    const float2 xx = in[pos];
                    const float2 p = nco[r];
                    t = (f2){xx.x * p.x - xx.y * p.y, xx.x * p.y + xx.y * p.x};
                }
                v[k] = t;
                r += kstep;
                if (r >= denom) r -= denom;
            }
        }
    }
    // ---- forward DFT_4096: v[k] = X[j + 256 k] -------------------------------------
    fft4096_regs(v, lds, tw, j);
    // ---- * H and fold: Y[j + 256 c] = sum_q X[j + 256 (c + 4 q)] H[...] -----------
    f2 y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f2 acc = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 h = H[j + 256 * (c + 4 * q)];
            const f2 xv = v[c + 4 * q];
            acc = __builtin_elementwise_fma(xv.yy, (f2){-h.y, h.x}, __builtin_elementwise_fma(xv.xx, (f2){h.x, h.y}, acc));
        }
        y[c] = acc;
    }
    // ---- inverse DFT_1024, Stockham radix 4 x 5 ---------------------------------------
#pragma unroll
    for (int pass = 0; pass < 5; ++pass) {
        const int ns = 1 << (2 * pass);
        if (pass > 0) {
            __syncthreads();  // previous use of the LDS image is over
            // (for pass 0 the forward transform's last reads were followed by a barrier-free
            //  register phase; the barrier below orders them before the first writes)
        }
        if (pass > 0) {
            // twiddles e^{+j 2 pi c (j mod ns) / (4 ns)} = conj(tw[(j mod ns) * 1024 / ns])^c
            const float2 t = tw[(j & (ns - 1)) * (1024 / ns)];
            const f2 w1 = {t.x, -t.y}, q1 = mul_pj(w1);
            const f2 w2 = cmul2(w1, w1, q1), q2 = mul_pj(w2);
            const f2 w3 = cmul2(w2, w1, q1), q3 = mul_pj(w3);
            y[1] = cmul2(y[1], w1, q1);
            y[2] = cmul2(y[2], w2, q2);
            y[3] = cmul2(y[3], w3, q3);
        }
        idft4(y[0], y[1], y[2], y[3]);
        if (pass == 4) break;  // natural order: y[c] = result[j + 256 c]
        const int k = j & (ns - 1);
        const int o = ((j - k) << 2) + k;  // (j / ns) * 4 ns + j mod ns
        if (pass == 0) __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) lds_st(lds + pad16(o + c * ns), y[c]);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = lds_ld(lds + pad16(j + 256 * c));
    }
    // ---- store the valid part --------------------------------------------------------
    const long mbase = (long)blockIdx.x * per_block;
    const int first = V >> 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int tau = j + 256 * c;
        const long m = mbase + (tau - first);
        if (tau >= first && m < n_out) {
            float2 w;
            w.x = y[c].x;
            w.y = y[c].y;
            out[m] = w;
        }
    }
}

// ---------------------------------------------------------------------------
// Kernel 3w  k_ols_wave: overlap-save with 1024-sample blocks, ONE WAVE per block.
// k_ols_decim4 spends most of its time at the ~11 workgroup barriers of a block (its
// transform phases were measured purely additive to the load/store floor); with one
// wave per block every exchange is wave-local: LDS ordering inside a wave needs no
// s_barrier, the 16 waves of a CU run independent blocks, and H / twiddle tables are
// 8 KiB each (L1-resident).  Forward DFT_1024 = radix 16 x 16 x 4 (16 values per lane),
// * H, fold 4 -> 1 in the lane, inverse DFT_256 = radix 4 x 4 x 4 x 4 (one butterfly per
// lane and pass).  V = overlap (multiple of 64, >= Lc - 1), hop = 1024 - V; the overlap
// re-reads come from L2 because neighbouring blocks run on the same XCD.
#ifndef RR_V_OLSW_ST_AUX
#define RR_V_OLSW_ST_AUX 0  // cache policy of the decimated samples' stores: default.  With the streaming hint (2) the kernel
                            // itself is as fast, but the Fourier stage behind it in the chain then reads its 134 MB from HBM
                            // instead of (mostly) the memory-side cache: k_fft4096 0.048 -> 0.043 ms, chain step 0.1764 -> 0.173 ms
#endif
// The last V samples of a block are the first V of the next one: the 128-sample pieces of a block that reach into them are loaded
// WITHOUT the streaming hint, so that the neighbour finds them in L2 (k_ols_frame: RR_V_FRAME_LD_TAILK).  8 = by the block's
// overlap (a wave-uniform test per piece); 0 .. 7: the pieces from that one on, whatever the overlap (A/B runs).  One session, ms
// per 2^26 samples, hint on every piece / by the overlap: Downsampler 4 : 1 with 299 taps (V = 304) 0.153 - 0.157 / 0.146 - 0.147;
// with 120 taps 0.125 / 0.121; 2 : 1 and the Filter of 64 .. 256 taps inside the noise (scripts/tailk_probe.py).
#ifndef RR_V_OLSW_TAILK
#define RR_V_OLSW_TAILK 8
#endif
#ifndef RR_V_FLTW_TAILK
#define RR_V_FLTW_TAILK 9  // k_filter_wave: the hint on every piece (9) - by the overlap (8) the Filter of 64 taps measured 5 - 9 % slower
#endif                     // (0.225 - 0.230 against 0.241 - 0.245 ms in the A/B session, 0.191 against 0.208 between two profile sessions)
constexpr unsigned kWaveWin = 64;  // blocks dealt to the XCDs in a moving window, that many neighbouring blocks per XCD
                                   // (one contiguous eighth of the stream per XCD: 0.1375 -> 0.135 ms; 16 .. 1024 alike)
// D = 4 is the benchmark's form.  D = 2 and D = 8 fold the spectrum into 2 resp. 8 parts instead of 4 (decimation =
// aliasing in frequency: Y[i] = sum_q X[i + (1024 / D) q] H[..]) and differ in the inverse only: 512 points as the
// forward radix 8 x 8 x 8 routine of k_fft512 with the result index reversed (IDFT(Z)[t] = DFT(Z)[(N - t) mod N]),
// 128 points as radix 2 x 4 x 4 x 4 on the lower half of the wave.
// POLY (D = 4 only): the forward transform in polyphase form.  With x_p[m] = xs[4 m + p], X[k + 256 q] =
// sum_p W_1024^((k + 256 q) p) X_p[k] (X_p = DFT_256 x_p), so the folded spectrum is
//   Y[k] = sum_p X_p[k] G_p[k],   G_p[k] = sum_q H[k + 256 q] W_1024^((k + 256 q) p)   (host, f64),
// four 256-point transforms (radix 8 x 8 x 4) and the same 16 products instead of one 1024-point transform: the last
// radix-4 stage of the long transform and a third of the twiddles are gone (-90 of 520 vector instructions per block,
// k_ols_wave 0.1337 -> 0.129 ms by an instruction-count ablation before it was written).  A lane's 16 samples are two
// phases p = 2 (l & 1) + j at m = (l >> 1) + 32 k', so the passes are: radix 8 over k' in the lane (as before),
// twiddle W_256^(mu kappa1), exchange, radix 8 over mu2 (mu = mu1 + 4 mu2), exchange, twiddle W_32^(mu1 kappa2a),
// radix 4 over mu1 for all four phases in lane kappa1 + 8 kappa2a - which leaves X_p[l + 64 c], the very layout the
// product with G and the inverse DFT_256 want.  Both exchanges move (j = 0, 1) pairs as 16-byte accesses.
#ifndef RR_V_OLSW_OCC
#define RR_V_OLSW_OCC 4
#endif


// MF: no mixer in the kernel - the stand-alone Downsampler (its table is all ones), or the chain with the mixer folded into the
// response tables (NCO periods that divide 8, rr_chain::ensure_mixfold; SW: the spectrum taken 128 bins further on, D = 4)
// (the kernel's body as a function of the workgroup index bx: k_ols_wave runs it for one stream, k_ols_wave_bank for the
//  channels of a bank - the same stream parameters, per-channel pointers, channel = blockIdx.y)
// GP (with MF, POLY): ANY NCO period with the mixer moved behind the filter - see k_ols_frame<.., GP>: H holds the tables of the
// response c[i] w^-i, the results are multiplied by the phase table's entries at their positions b0 + D tau: the block's own
// phasor (a scalar read) x the lane's constant (behind the table: w^(4 l), w^(1024 - 2 l), w^(8 (l mod 32)) for D = 4, 2, 8)
// x the rotations by 128 samples.  1024 / D products per block instead of 1024.
template <int D, bool POLY, bool MF = false, bool SW = false, bool GP = false>
__device__ __forceinline__ void ols_wave_body(
    const float2 *__restrict__ xh, int hx, const float2 *__restrict__ in, long n_in, const float2 *__restrict__ nco,
    unsigned denom, unsigned idx0, const float2 *__restrict__ H, const float2 *__restrict__ tw, int V,
    float2 *__restrict__ out, long n_out, long e0, float2 *__restrict__ xh_out, int hx_out, unsigned nblocks,
    unsigned ph0, unsigned hopm, unsigned kstep, double inv_denom, const unsigned bx, const unsigned G) {
    __shared__ __attribute__((aligned(16))) f2 lds[POLY ? 1136 : kWaveLds];  // (POLY: 2 (63 + 72 * 7) + 2 elements)
    const int l = threadIdx.x;
    // Workgroups b, b+8, .. share an XCD.  Blocks are dealt so that neighbouring blocks run on one XCD - the V
    // samples two neighbours share come from HBM once -, and the XCDs work side by side in a moving window of
    // 8 G blocks (instead of one far-apart eighth of the stream per XCD).
    // (G = kWaveWin, a constant, in k_ols_wave; the banks' launches - many short streams - take a smaller window so that a
    //  channel's grid is not rounded up to 512 workgroups: 64 channels x 2^16 samples were 5056 blocks in 32768 workgroups)
    const unsigned grp = bx / (8 * G), rem = bx % (8 * G);
    const unsigned blk = grp * 8 * G + (rem & 7) * G + (rem >> 3);
    if (blk >= nblocks) return;
    static_assert(D == 2 || D == 4 || D == 8, "fold 2, 4 or 8");
    constexpr int ND = 16 / D;  // bins per lane behind the fold
    const int hop = 1024 - V, per_block = hop / D;
    const long b0 = e0 - V + (long)blk * hop;

    if (xh_out && blk == nblocks - 1) {  // mixed-sample history for the next call
        for (int i = l; i < hx_out; i += 64) {
            const long pos = n_in - hx_out + i;
            float2 v;
            if (pos >= 0) {
                const float2 xx = in[pos];
                const float2 pp = nco[(unsigned)(((long)idx0 + pos) % (long)denom)];
                v.x = xx.x * pp.x - xx.y * pp.y;
                v.y = xx.x * pp.y + xx.y * pp.x;
            } else {
                v = (pos >= -(long)hx) ? xh[hx + pos] : float2{0.f, 0.f};
            }
            xh_out[i] = v;
        }
    }

    // ---- NCO phase of the block's first sample: (idx0 + b0) mod denom -------------------------
    // ph0 = (idx0 + e0 - V) mod denom and hopm = hop mod denom come from the host; the block's
    // term blk * hopm < 2^53 is reduced in f64 (exact) instead of a 64-bit integer division.
    unsigned base = ph0;
    if (hopm != 0) {
        const double dn = (double)denom;
        const double prod = __builtin_fma((double)blk, (double)hopm, (double)ph0);
        const double qd = __builtin_floor(prod * inv_denom);
        double rd = __builtin_fma(-qd, dn, prod);
        if (rd < 0.0) rd += dn;
        if (rd >= dn) rd -= dn;
        base = (unsigned)rd;
    }

    // The block's samples first, then the lane constants of the transforms: all requested before
    // anything waits (the wave-level fences below would otherwise pin each of these L2-latency
    // loads right in front of its use).  Vector memory costs per instruction here, not per byte
    // (measured), so everything comes in 16-byte pieces: lane l takes the sample pairs
    // x[2 l + 128 k' .. + 1], k' < 8, and its 6 twiddle seeds as 3 packed entries.
    f4u x[8];
    const bool interior = b0 >= 0 && b0 + 1024 <= n_in;
    if (interior) {
        const f4u *src = reinterpret_cast<const f4u *>(in + b0) + l;
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (RR_V_OLSW_TAILK == 8 ? 128 * (k + 1) <= hop : k < RR_V_OLSW_TAILK) ? ld_stream(src + 64 * k) : *(src + 64 * k);
    }
    [[maybe_unused]] f2 gph = {1.f, 0.f};  // GP: the phasor of the lane's result c = 0
    [[maybe_unused]] float2 gpb, gpl;
    if constexpr (GP) {
        static_assert(MF && POLY && !SW, "GP: the blocks transform the samples as they are, polyphase tables");
        gpb = ld_uniform(nco + __builtin_amdgcn_readfirstlane(base));
        gpl = nco[denom + 9 + (D == 4 ? 0 : D == 2 ? 64 : 128) + l];
    }
    const int g = l >> 4, q = l & 15;
    // seeds: pass 1 tw[8 (l mod 8)]; pass 2 tw[l], tw[l + 64]; inverse (D = 4) tw[64 (l mod 4)], tw[16 (l mod 16)], tw[4 l];
    // D = 8: tw[128 (l mod 2)], tw[32 (l mod 8)], tw[8 (l mod 32)]; D = 2: tw[16 (l mod 8)], tw[2 l]  (append_wave1024_seeds)
    f2 t_p1, t_p2[2], t_inv[3];
    {
        const float4 *tl = reinterpret_cast<const float4 *>(tw + 1024) + l;
        const float4 s0 = tl[0], s1 = tl[64];
        t_p1 = (f2){s0.x, s0.y};
        t_p2[0] = (f2){s0.z, s0.w};
        t_p2[1] = (f2){s1.x, s1.y};
        if (POLY && D == 4) {  // tw[4 (l >> 1)], tw[32 (l >> 3)] and the three seeds of the inverse
            const float4 s6 = tl[384], s7 = tl[448], s8 = tl[512];
            t_p1 = (f2){s6.x, s6.y};
            t_p2[0] = (f2){s6.z, s6.w};
            t_inv[0] = (f2){s7.x, s7.y};
            t_inv[1] = (f2){s7.z, s7.w};
            t_inv[2] = (f2){s8.x, s8.y};
        } else if (POLY && D == 8) {  // tw[8 (l >> 2)], tw[64 (l >> 3)]; the inverse's as below
            const float4 s9 = tl[576], s3 = tl[192], s4 = tl[256];
            t_p1 = (f2){s9.x, s9.y};
            t_p2[0] = (f2){s9.z, s9.w};
            t_inv[0] = (f2){s3.x, s3.y};
            t_inv[1] = (f2){s3.z, s3.w};
            t_inv[2] = (f2){s4.x, s4.y};
        } else if (POLY) {  // D = 2: tw[2 l], tw[16 (l >> 3)]; the inverse's tw[16 (l mod 8)], tw[2 l]
            const float4 s9 = tl[640], s4 = tl[256];
            t_p1 = (f2){s9.x, s9.y};
            t_p2[0] = (f2){s9.z, s9.w};
            t_inv[0] = (f2){s4.z, s4.w};
            t_inv[1] = t_p1;
            t_inv[2] = t_p1;
        } else if (D == 4) {
            const float4 s2 = tl[128];
            t_inv[0] = (f2){s1.z, s1.w};
            t_inv[1] = (f2){s2.x, s2.y};
            t_inv[2] = (f2){s2.z, s2.w};
        } else if (D == 8) {
            const float4 s3 = tl[192], s4 = tl[256];
            t_inv[0] = (f2){s3.x, s3.y};
            t_inv[1] = (f2){s3.z, s3.w};
            t_inv[2] = (f2){s4.x, s4.y};
        } else {
            const float4 s4 = tl[256], s5 = tl[320];
            t_inv[0] = (f2){s4.z, s4.w};
            t_inv[1] = (f2){s5.x, s5.y};
            t_inv[2] = t_inv[1];
        }
    }
    [[maybe_unused]] f2 *const a_rd = lds + (l + 2 * g);  // A(l + 64 m + 256 c) = a_rd + 72 m + 296 c
    // ---- phase of the lane's first sample (2 l into the block): (base + 2 l) mod denom -----------
    unsigned r = base + 2u * (unsigned)l;
    if (denom >= 128u) {
        if (r >= denom) r -= denom;
    } else if ((denom & (denom - 1u)) == 0u) {
        r &= denom - 1u;
    } else {
        r %= denom;
    }
    // ---- mix: v[2 k' + j] = xs[b0 + 2 l + j + 128 k'] ------------------------------------------
    // (the NCO table carries entry 0 once more behind entry denom - 1, so the pair r, r + 1 is one 16-byte read)
    f2 v[16];
    if (interior && MF) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[2 * k] = (f2){x[k].x, x[k].y};
            v[2 * k + 1] = (f2){x[k].z, x[k].w};
        }
    } else if (interior) {
        const f4u pp = *reinterpret_cast<const f4u *>(nco + r);
        if (kstep == 0) {  // the period divides 128: one pair of phasors per lane
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[2 * k] = cmul((f2){x[k].x, x[k].y}, (f2){pp.x, pp.y});
                v[2 * k + 1] = cmul((f2){x[k].z, x[k].w}, (f2){pp.z, pp.w});
            }
        } else {
            // general period: the lane's pair at the block start from the table, the seven
            // 128-sample steps by the rotations kept behind the table (one product each; a
            // last-bit difference from the table's own entries, far inside the chain's 1e-5)
            const f2 p0 = {pp.x, pp.y}, p1 = {pp.z, pp.w};
            v[0] = cmul((f2){x[0].x, x[0].y}, p0);
            v[1] = cmul((f2){x[0].z, x[0].w}, p1);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float2 rt = ld_uniform(nco + (denom + 1 + k));  // uniform address: a scalar read
                const f2 rot = {rt.x, rt.y};
                v[2 * k] = cmul((f2){x[k].x, x[k].y}, cmul(p0, rot));
                v[2 * k + 1] = cmul((f2){x[k].z, x[k].w}, cmul(p1, rot));
            }
        }
    } else {
        // edges: history (already mixed) in front, nothing behind the input.  Every lane reads
        // some valid address and selects afterwards.
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f4u pp = *reinterpret_cast<const f4u *>(nco + r);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long pos = b0 + 2 * l + j + 128 * k;
                const bool inr = pos >= 0 && pos < n_in;
                const bool hst = pos < 0 && pos >= -(long)hx;
                const float2 *ptr = inr ? in + pos : xh + (hst ? hx + pos : 0);
                const float2 xx = *ptr;
                const f2 p = j ? (f2){pp.z, pp.w} : (f2){pp.x, pp.y};
                // (MF: the block wants the samples UNMIXED - the history, which holds mixed ones, times conj(p))
                const f2 pk = MF ? (f2){inr ? 1.f : (hst ? p.x : 0.f), hst ? -p.y : 0.f}
                                 : (f2){inr ? p.x : (hst ? 1.f : 0.f), inr ? p.y : 0.f};
                const f2 xv = {(inr || hst) ? xx.x : 0.f, (inr || hst) ? xx.y : 0.f};
                v[2 * k + j] = cmul(xv, pk);
            }
            r += kstep;
            if (r >= denom) r -= denom;
        }
    }
    // the 16 H values of the lane, used in pass 2 (H arrives pair-interleaved from the host,
    // Hp[kp][l] = {H[l + 128 kp], H[l + 128 kp + 64]}: 8 loads of 16 bytes per lane)
    float2 hv[16];
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
        const float4 h4 = reinterpret_cast<const float4 *>(H)[l + 64 * kp];
        hv[2 * kp] = float2{h4.x, h4.y};
        hv[2 * kp + 1] = float2{h4.z, h4.w};
    }
    f2 y[ND];
    if constexpr (POLY) {
        // ---- the D transforms of 1024 / D points of the phases x_p[m] = xs[D m + p]: radix 8 (k') x 8 (mu2) x RC (mu1) ----
        // lane l = low3 + 8 mu2, low3 = a + (D / 2) mu1: phases p = 2 a + j, mu = mu1 + RC mu2, m = mu + (128 / D) k'
        constexpr int RC = 16 / D;  // radix of the last pass = bins per lane (ND)
        f2 e0[8], e1[8];            // phase j = 0 / 1 of this lane, over k'
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            e0[k] = v[2 * k];
            e1[k] = v[2 * k + 1];
        }
        dft8(e0);
        dft8(e1);
        {   // * W_(1024/D)^(mu kappa1): powers of one seed
            const f2 w1 = t_p1, w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
            const f2 w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
            e0[1] = cmul(e0[1], w1); e1[1] = cmul(e1[1], w1);
            e0[2] = cmul(e0[2], w2); e1[2] = cmul(e1[2], w2);
            e0[3] = cmul(e0[3], w3); e1[3] = cmul(e1[3], w3);
            e0[4] = cmul(e0[4], w4); e1[4] = cmul(e1[4], w4);
            e0[5] = cmul(e0[5], w5); e1[5] = cmul(e1[5], w5);
            e0[6] = cmul(e0[6], w6); e1[6] = cmul(e1[6], w6);
            e0[7] = cmul(e0[7], w7); e1[7] = cmul(e1[7], w7);
        }
        // exchange 1: element (low3, mu2, j, kappa1) at 2 (l + 72 kappa1) + j; the reader - lane low3 + 8 kappa1 - takes
        // mu2 = 0 .. 7: 2 ((l & 7) + 72 (l >> 3) + 8 mu2) + j
        {
            f2 *row = lds + 2 * l;
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<float4 *>(row + 144 * k) = (float4){e0[k].x, e0[k].y, e1[k].x, e1[k].y};
        }
        wave_sync();
        {
            const f2 *col = lds + 2 * ((l & 7) + 72 * (l >> 3));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 r = *reinterpret_cast<const float4 *>(col + 16 * k);
                e0[k] = (f2){r.x, r.y};
                e1[k] = (f2){r.z, r.w};
            }
        }
        dft8(e0);  // over mu2: out kappa2a
        dft8(e1);
        wave_sync();  // the first image has been read
        // exchange 2: element (low3, kappa1, j, kappa2a) at 2 (kappa1 + 8 kappa2a + 65 low3) + j; the reader is lane kappa1 + 8 kappa2a.
        // (A 16-byte store is served in groups of 8 neighbouring lanes over 32 banks: the planes of the eight low3 values must
        // start 4 banks apart - 130 elements = 260 dwords = 4 mod 32; a pitch of 132 elements put lanes t and t + 4 on the same
        // banks, 2-way conflicts on every store of this exchange.)
        {
            f2 *row = lds + 2 * ((l >> 3) + 65 * (l & 7));
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<float4 *>(row + 16 * k) = (float4){e0[k].x, e0[k].y, e1[k].x, e1[k].y};
        }
        wave_sync();
        f2 d[D][RC];  // [phase p][mu1]
#pragma unroll
        for (int a = 0; a < D / 2; ++a)
#pragma unroll
            for (int m1 = 0; m1 < RC; ++m1) {
                const float4 r = *reinterpret_cast<const float4 *>(lds + 2 * l + 130 * (a + (D / 2) * m1));
                d[2 * a][m1] = (f2){r.x, r.y};
                d[2 * a + 1][m1] = (f2){r.z, r.w};
            }
        {   // * W_(8 RC)^(mu1 kappa2a), kappa2a = l >> 3, then radix RC over mu1: X_p[l + 64 c]
            f2 w[RC];
            w[1] = t_p2[0];
            if constexpr (RC >= 4) {
                w[2] = cmul(w[1], w[1]);
                w[3] = cmul(w[2], w[1]);
            }
            if constexpr (RC == 8) {
                w[4] = cmul(w[2], w[2]);
                w[5] = cmul(w[4], w[1]);
                w[6] = cmul(w[4], w[2]);
                w[7] = cmul(w[4], w[3]);
            }
#pragma unroll
            for (int pp = 0; pp < D; ++pp) {
#pragma unroll
                for (int m1 = 1; m1 < RC; ++m1) d[pp][m1] = cmul(d[pp][m1], w[m1]);
                if constexpr (RC == 2) {
                    const f2 s0 = d[pp][0] + d[pp][1], s1 = d[pp][0] - d[pp][1];
                    d[pp][0] = s0;
                    d[pp][1] = s1;
                } else if constexpr (RC == 4) {
                    dft4(d[pp][0], d[pp][1], d[pp][2], d[pp][3]);
                } else {
                    dft8(d[pp]);
                }
            }
        }
        // Y[l + 64 c] = sum_p X_p[l + 64 c] G_p[l + 64 c]; the table holds entry i = ND p + c as half of the 16-byte piece [i >> 1][l]
#pragma unroll
        for (int c = 0; c < ND; ++c) {
            f2 acc = cmul(d[0][c], (f2){hv[c].x, hv[c].y});
#pragma unroll
            for (int pp = 1; pp < D; ++pp) acc = cmac(acc, d[pp][c], (f2){hv[ND * pp + c].x, hv[ND * pp + c].y});
            y[c] = acc;
        }
    } else {
    // ---- forward DFT_1024 = radix 8 x 16 x 8 (Stockham) -------------------------------------------
    // pass 0 (Ns = 1): butterflies 2 l + j over x[2 l + j + 128 k']; out 8 (2 l + j) + r = 16 l + 8 j + r
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
    // pass 1 (Ns = 8, radix 16): twiddle e^{-j 2 pi k (l mod 8) / 128}; out 128 (l / 8) + l % 8 + 8 k
    twiddle16(v, t_p1);
    dft16(v);
    wave_sync();
    {
        // A(128 h + p + 8 k), h = l / 8, p = l % 8: 144 h + 8 (h / 2) + p + 8 k + 2 (k / 2)
        f2 *col = lds + (144 * (l >> 3) + 8 * (l >> 4) + (l & 7));
#pragma unroll
        for (int k = 0; k < 16; ++k) lds_st(col + (8 * k + 2 * (k >> 1)), v[k]);
    }
    wave_sync();
    // pass 2 (Ns = 128, radix 8): butterflies t = l + 64 m over in[t + 128 c]; out X[t + 128 r],
    // i.e. X[l + 64 k] with k = m + 2 r
    f2 X[16];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f2 a[8];
        // A(l + 64 m + 128 c) = a_rd + 72 m + 144 c + 8 (c / 2)
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
    // * H and fold the D parts of 1024 / D bins: Y[l + 64 m] = sum_q X[l + 64 (m + ND q)] H[l + 64 (m + ND q)]
#pragma unroll
    for (int m = 0; m < ND; ++m) {
        f2 acc = cmul(X[m], (f2){hv[m].x, hv[m].y});
#pragma unroll
        for (int q2 = 1; q2 < D; ++q2) acc = cmac(acc, X[m + ND * q2], (f2){hv[m + ND * q2].x, hv[m + ND * q2].y});
        y[m] = acc;
    }
    }
    const int first = V / D;
    const long mb = (long)blk * per_block;
    const long left = n_out - mb;
    const unsigned recs = (unsigned)(left < per_block ? left : per_block) * 8u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + mb, 0, recs, 0x00020000);
    if constexpr (D == 4) {
        if (SW) {  // results with alternating signs = the spectrum 128 bins further on
            const f2 t0 = y[0], t1 = y[1];
            y[0] = y[2];
            y[1] = y[3];
            y[2] = t0;
            y[3] = t1;
        }
        // ---- inverse DFT_256, Stockham radix 4 x 4: butterfly l reads in[l + 64 c] ------------------
        // Each of the three exchanges has its own image layout (inv256_rd): the LDS serves 8-byte reads in halves of 32 lanes
        // over 64 banks, 8-byte stores in groups of 16 lanes and 16-byte stores in groups of 8 lanes over 32 banks, and one
        // padding for all three (4 elements per 16, the first form) left 2-way conflicts on the first exchange's stores and on
        // the reads of the first and the last.
        // pass 0 (Ns = 1): no twiddle; out 4 l + c, element i at 2 (i >> 2) + (i & 1) + 144 ((i >> 1) & 1): the lane's pairs
        // {y0, y1} at 2 l and {y2, y3} at 144 + 2 l - neighbouring lanes store neighbouring 16-byte pieces, and the 32
        // elements a half-wave reads are two runs of 16, 32 banks apart
        idft4(y[0], y[1], y[2], y[3]);
        wave_sync();  // the forward image has been read
        {
            f2 *row = lds + 2 * l;
            *reinterpret_cast<float4 *>(row) = (float4){y[0].x, y[0].y, y[1].x, y[1].y};
            *reinterpret_cast<float4 *>(row + 144) = (float4){y[2].x, y[2].y, y[3].x, y[3].y};
        }
        wave_sync();
#pragma unroll
        for (int pass = 1; pass < 4; ++pass) {
            const f2 *const rd = inv256_rd(lds, l, pass);
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = lds_ld(rd + ((pass == 1 ? 32 : 80) * c));
            // twiddles e^{+j 2 pi c (l mod ns) / (4 ns)} = conj(tw[(l mod ns) 256 / ns])^c, ns = 4^pass
            const f2 w1 = t_inv[pass - 1];
            const f2 w2 = cmul(w1, w1);
            const f2 w3 = cmul(w2, w1);
            y[1] = cmul_conj(y[1], w1);
            y[2] = cmul_conj(y[2], w2);
            y[3] = cmul_conj(y[3], w3);
            idft4(y[0], y[1], y[2], y[3]);
            if (pass == 3) break;  // natural order: y[c] = result[l + 64 c]
            wave_sync();
            if (pass == 1) {  // out 16 (l >> 2) + (l & 3) + 4 c, element i at i + 4 (i >> 4): 20 (l >> 2) + (l & 3) + 4 c
                f2 *col = lds + (20 * (l >> 2) + (l & 3));
#pragma unroll
                for (int c = 0; c < 4; ++c) lds_st(col + (4 * c), y[c]);
            } else {  // out 64 g + q + 16 c, element i at i + 16 (i >> 6): 80 g + q + 16 c
                f2 *col = lds + (80 * g + q);
#pragma unroll
                for (int c = 0; c < 4; ++c) lds_st(col + (16 * c), y[c]);
            }
            wave_sync();
        }
        // ---- store the valid part -----------------------------------------------------------------
        // Buffer stores: the lanes outside the block's valid part (and behind the end of the output)
        // carry an out-of-range offset and are dropped by the address check - four stores in
        // straight-line code, with the streaming hint.
        if constexpr (GP) {  // result tau = l + 64 c at b0 + 4 tau
            gph = cmul((f2){gpb.x, gpb.y}, (f2){gpl.x, gpl.y});
            y[0] = cmul(y[0], gph);
#pragma unroll
            for (int c = 1; c < 4; ++c) {
                const float2 rt = ld_uniform(nco + (denom + 1 + 2 * c));
                y[c] = cmul(y[c], cmul(gph, (f2){rt.x, rt.y}));
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int tau = l + 64 * c;
            const unsigned off = tau >= first ? (unsigned)(tau - first) * 8u : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b64(y[c], rs, off, 0, RR_V_OLSW_ST_AUX);
        }
    } else if constexpr (D == 2) {
        // ---- inverse DFT_512 as the forward radix 8 x 8 x 8 (k_fft512's passes) with the result index reversed ----
        dft8(y);  // pass 0 (Ns = 1): butterfly l over Y[l + 64 c]; out 8 l + c
        wave_sync();  // the forward image has been read
#pragma unroll
        for (int k = 0; k < 8; ++k) lds_st(lds + pad8(8 * l) + k, y[k]);
        wave_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = lds_ld(lds + (l + (l >> 3)) + 72 * k);  // pad8(l + 64 k)
        twiddle8(y, t_inv[0]);  // pass 1 (Ns = 8): e^{-j 2 pi (l mod 8) k / 64}; out (l / 8) 64 + l % 8 + 8 k
        dft8(y);
        wave_sync();
        {
            f2 *col = lds + (72 * (l >> 3) + (l & 7));  // pad8(64 h + p + 8 k) = 72 h + p + 9 k
#pragma unroll
            for (int k = 0; k < 8; ++k) lds_st(col + 9 * k, y[k]);
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = lds_ld(lds + (l + (l >> 3)) + 72 * k);
        twiddle8(y, t_inv[1]);  // pass 2 (Ns = 64): e^{-j 2 pi l k / 512}; out l + 64 k
        dft8(y);
        // y[k] = DFT(Y)[l + 64 k] = result[(512 - l - 64 k) mod 512]
        if constexpr (GP) {  // result tau at b0 + 2 tau: w^(2 tau) = w^(1024 - 2 l) conj(w^(128 k)), and 1 for tau = 0 (lane 0, k = 0)
            gph = cmul((f2){gpb.x, gpb.y}, (f2){gpl.x, gpl.y});
            y[0] = cmul(y[0], l == 0 ? (f2){gpb.x, gpb.y} : gph);
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float2 rt = ld_uniform(nco + (denom + 1 + k));
                y[k] = cmul(y[k], cmul_conj(gph, (f2){rt.x, rt.y}));
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int tau = (512 - l - 64 * k) & 511;
            const unsigned off = tau >= first ? (unsigned)(tau - first) * 8u : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b64(y[k], rs, off, 0, 2);
        }
    } else {
        // ---- inverse DFT_128, Stockham radix 2 x 4 x 4 x 4; the radix-4 passes on lanes 0 .. 31 ------------------
        // pass 0 (radix 2, Ns = 1): butterfly l over Y[l], Y[l + 64]; out 2 l + c
        {
            const f2 s0 = y[0] + y[1], s1 = y[0] - y[1];
            wave_sync();  // the forward image has been read
            *reinterpret_cast<float4 *>(lds + 2 * l) = (float4){s0.x, s0.y, s1.x, s1.y};
        }
        wave_sync();
        const int j = l & 31;
        f2 z[4];
#pragma unroll
        for (int pass = 1; pass < 4; ++pass) {
#pragma unroll
            for (int c = 0; c < 4; ++c) z[c] = lds_ld(lds + j + 32 * c);
            // twiddles e^{+j 2 pi c (j mod ns) / (4 ns)}, ns = 2, 8, 32
            const f2 w1 = t_inv[pass - 1];
            const f2 w2 = cmul(w1, w1);
            const f2 w3 = cmul(w2, w1);
            z[1] = cmul_conj(z[1], w1);
            z[2] = cmul_conj(z[2], w2);
            z[3] = cmul_conj(z[3], w3);
            idft4(z[0], z[1], z[2], z[3]);
            if (pass == 3) break;  // natural order: z[c] = result[j + 32 c]
            wave_sync();
            if (l < 32) {
                // pass 1: out 8 (j >> 1) + (j & 1) + 2 c;  pass 2: out 32 (j >> 3) + (j & 7) + 8 c
                f2 *col = pass == 1 ? lds + (8 * (j >> 1) + (j & 1)) : lds + (32 * (j >> 3) + (j & 7));
                const int st = pass == 1 ? 2 : 8;
#pragma unroll
                for (int c = 0; c < 4; ++c) lds_st(col + st * c, z[c]);
            }
            wave_sync();
        }
        if constexpr (GP) {  // result tau = j + 32 c at b0 + 8 tau
            gph = cmul((f2){gpb.x, gpb.y}, (f2){gpl.x, gpl.y});
            z[0] = cmul(z[0], gph);
#pragma unroll
            for (int c = 1; c < 4; ++c) {
                const float2 rt = ld_uniform(nco + (denom + 1 + 2 * c));
                z[c] = cmul(z[c], cmul(gph, (f2){rt.x, rt.y}));
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int tau = j + 32 * c;
            const unsigned off = (l < 32 && tau >= first) ? (unsigned)(tau - first) * 8u : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b64(z[c], rs, off, 0, 2);
        }
    }
}

template <int D, bool POLY, bool MF = false, bool SW = false, bool GP = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RR_V_OLSW_OCC, RR_V_OLSW_OCC))) void k_ols_wave(
    const float2 *__restrict__ xh, int hx, const float2 *__restrict__ in, long n_in, const float2 *__restrict__ nco,
    unsigned denom, unsigned idx0, const float2 *__restrict__ H, const float2 *__restrict__ tw, int V,
    float2 *__restrict__ out, long n_out, long e0, float2 *__restrict__ xh_out, int hx_out, unsigned nblocks,
    unsigned ph0, unsigned hopm, unsigned kstep, double inv_denom) {
    ols_wave_body<D, POLY, MF, SW, GP>(xh, hx, in, n_in, nco, denom, idx0, H, tw, V, out, n_out, e0, xh_out, hx_out, nblocks, ph0, hopm,
                                   kstep, inv_denom, blockIdx.x, kWaveWin);
}

// The channels of a bank (rr_chainbank: K independent streams with the same parameters that advance in lockstep): the same
// launch parameters for all of them, the streams' own buffers from a table, channel = blockIdx.y.
template <int D, bool POLY, bool MF = false, bool SW = false, bool GP = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RR_V_OLSW_OCC, RR_V_OLSW_OCC))) void k_ols_wave_bank(
    const BankTable chan, int hx, long n_in, const float2 *__restrict__ nco, unsigned denom, unsigned idx0,
    const float2 *__restrict__ H, const float2 *__restrict__ tw, int V, long n_out, long e0, int hx_out, unsigned nblocks,
    unsigned ph0, unsigned hopm, unsigned kstep, double inv_denom, unsigned gwin) {
    const BankPtrs c = chan.c[blockIdx.y];
    ols_wave_body<D, POLY, MF, SW, GP>((const float2 *)c.xh, hx, (const float2 *)c.in, n_in, nco, denom, idx0, H, tw, V, (float2 *)c.dec,
                                   n_out, e0, (float2 *)c.xh_out, hx_out, nblocks, ph0, hopm, kstep, inv_denom, blockIdx.x, gwin);
}

// V: the Lc - 1 wrapped samples rounded up to a multiple of `granule` - 16 for k_ols_wave (V / D whole for D = 2, 4, 8,
// block starts on 128-byte lines), 64 for k_ols_frame (208 outputs per block compiled in)
int ols_wave_overlap(size_t Lc, size_t granule) {
    const size_t v = (Lc - 1 + granule - 1) / granule * granule;
    return v == 0 ? (int)granule : (int)v;
}

bool ols_wave_supported(uint64_t D, size_t Lc) { return (D == 2 || D == 4 || D == 8) && Lc >= 1 && Lc - 1 <= 512; }

template <int D, bool POLY>
static int launch_ols_wave_d(hipStream_t s, const FusedFirArgs &a) {
    const int per_block = (1024 - a.V) / D;
    const size_t nblocks = (a.n_out + per_block - 1) / per_block;
    if (nblocks > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS: too many blocks");
    const int64_t den = (int64_t)a.denom;
    int64_t ph = ((int64_t)a.idx0 + a.e0 - a.V) % den;
    if (ph < 0) ph += den;
    const unsigned hopm = (unsigned)((int64_t)(1024 - a.V) % den), kstep = (unsigned)(128 % den);
    const unsigned grid = (unsigned)((nblocks + 8 * kWaveWin - 1) / (8 * kWaveWin) * (8 * kWaveWin));
#define RR_OLSW_LAUNCH(MF_, SW_, GP_)                                                                                                 \
    do {                                                                                                                              \
        if (a.ev_start && a.ev_stop)                                                                                                  \
            hipExtLaunchKernelGGL((k_ols_wave<D, POLY, MF_, SW_, GP_>), dim3(grid), dim3(64), 0, s, a.ev_start, a.ev_stop, 0,         \
                                  (const float2 *)a.xh, (int)a.hx, (const float2 *)a.in, (long)a.n_in, (const float2 *)a.nco, a.denom, \
                                  a.idx0, (const float2 *)a.H, (const float2 *)a.tw4096, a.V, (float2 *)a.out, (long)a.n_out,         \
                                  (long)a.e0, (float2 *)a.xh_out, (int)a.hx, (unsigned)nblocks, (unsigned)ph, hopm, kstep,            \
                                  1.0 / (double)den);                                                                                 \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_ols_wave<D, POLY, MF_, SW_, GP_>), dim3(grid), dim3(64), 0, s, (const float2 *)a.xh, (int)a.hx,     \
                               (const float2 *)a.in, (long)a.n_in, (const float2 *)a.nco, a.denom, a.idx0, (const float2 *)a.H,       \
                               (const float2 *)a.tw4096, a.V, (float2 *)a.out, (long)a.n_out, (long)a.e0, (float2 *)a.xh_out,         \
                               (int)a.hx, (unsigned)nblocks, (unsigned)ph, hopm, kstep, 1.0 / (double)den);                           \
    } while (0)
    // (a.mixfold: the caller's table is all ones - the stand-alone Downsampler - or a.H holds the tables with the mixer folded in)
    if constexpr (POLY) {
        if (a.genfold) RR_OLSW_LAUNCH(true, false, true);
        else if (a.mixfold && D == 4 && a.sigma < 0.f) RR_OLSW_LAUNCH(true, true, false);
        else if (a.mixfold) RR_OLSW_LAUNCH(true, false, false);
        else RR_OLSW_LAUNCH(false, false, false);
    } else {
        if (a.genfold) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS: the mixer behind the filter wants the polyphase tables");
        RR_OLSW_LAUNCH(false, false, false);
    }
#undef RR_OLSW_LAUNCH
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_ols_wave(hipStream_t s, const FusedFirArgs &a) {
    if (a.n_out == 0) return RR_OK;
    if (a.blk == 2048) return launch_ols_wave2k(s, a);
    if (a.blk != 1024) return launch_ols_wg(s, a);  // (256 D samples: 1536 at 6 : 1, 2560 .. 16 384 from 10 : 1 on)
    switch (a.D) {
    case 2: return a.poly ? launch_ols_wave_d<2, true>(s, a) : launch_ols_wave_d<2, false>(s, a);
    case 4: return a.poly ? launch_ols_wave_d<4, true>(s, a) : launch_ols_wave_d<4, false>(s, a);
    case 8: return a.poly ? launch_ols_wave_d<8, true>(s, a) : launch_ols_wave_d<8, false>(s, a);
    }
    RR_FAIL(RR_ERR_BAD_ARG, "fused OLS: decimation %u not instantiated", a.D);
}

template <int D>
static int launch_ols_wave_bank_d(hipStream_t s, const FusedFirArgs &a, const BankTable &d_chan, size_t channels) {
    const int per_block = (1024 - a.V) / D;
    const size_t nblocks = (a.n_out + per_block - 1) / per_block;
    if (nblocks > 0x7ffffff0ull || channels > kBankGroup) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS bank: too many blocks or channels");
    const int64_t den = (int64_t)a.denom;
    int64_t ph = ((int64_t)a.idx0 + a.e0 - a.V) % den;
    if (ph < 0) ph += den;
    const unsigned hopm = (unsigned)((int64_t)(1024 - a.V) % den), kstep = (unsigned)(128 % den);
    // window of neighbouring blocks per XCD: the stream kernel's 64 for long calls, 8 / 1 for short ones (the grid is rounded up to 8 windows)
    const unsigned gwin = nblocks >= 4096 ? kWaveWin : (nblocks >= 64 ? 8u : 1u);
    const unsigned grid = (unsigned)((nblocks + 8 * gwin - 1) / (8 * gwin) * (8 * gwin));
#define RR_OLSWB_LAUNCH(MF_, SW_, GP_)                                                                                                 \
    hipLaunchKernelGGL((k_ols_wave_bank<D, true, MF_, SW_, GP_>), dim3(grid, (unsigned)channels), dim3(64), 0, s, d_chan, (int)a.hx,     \
                       (long)a.n_in, (const float2 *)a.nco, a.denom, a.idx0, (const float2 *)a.H, (const float2 *)a.tw4096, a.V,    \
                       (long)a.n_out, (long)a.e0, (int)a.hx, (unsigned)nblocks, (unsigned)ph, hopm, kstep, 1.0 / (double)den, gwin)
    if (a.genfold) RR_OLSWB_LAUNCH(true, false, true);
    else if (a.mixfold && D == 4 && a.sigma < 0.f) RR_OLSWB_LAUNCH(true, true, false);
    else if (a.mixfold) RR_OLSWB_LAUNCH(true, false, false);
    else RR_OLSWB_LAUNCH(false, false, false);
#undef RR_OLSWB_LAUNCH
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// The channels' pending samples (BankPtrs::head, n of them) copied to BankPtrs::out in ONE launch: a bank step in which no frame
// completes appends to the pending chunk, which must then live in the chain's own buffer (64 copies of a few KiB as 64 calls cost
// more than the step's kernel)
__global__ __launch_bounds__(256) void k_bank_copy(const BankTable chan, long n) {
    const BankPtrs c = chan.c[blockIdx.y];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) reinterpret_cast<float2 *>(c.out)[i] = reinterpret_cast<const float2 *>(c.head)[i];
}
int launch_bank_copy(hipStream_t s, const BankTable &d_chan, size_t channels, size_t n) {
    if (n == 0 || channels == 0) return RR_OK;
    if (channels > kBankGroup || n > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "bank copy: too many channels or samples");
    hipLaunchKernelGGL(k_bank_copy, dim3((unsigned)((n + 255) / 256), (unsigned)channels), dim3(256), 0, s, d_chan, (long)n);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_ols_wave_bank(hipStream_t s, const FusedFirArgs &a, const BankTable &d_chan, size_t channels) {
    if (a.n_out == 0 || channels == 0) return RR_OK;
    if (!a.poly) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS bank: polyphase tables only");
    if (a.blk == 2048) return launch_ols_wave2k_bank(s, a, d_chan, channels);
    if (a.blk != 1024) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS bank: no bank form of the workgroup kernel (6 : 1, 10 .. 64 : 1)");
    switch (a.D) {
    case 2: return launch_ols_wave_bank_d<2>(s, a, d_chan, channels);
    case 4: return launch_ols_wave_bank_d<4>(s, a, d_chan, channels);
    case 8: return launch_ols_wave_bank_d<8>(s, a, d_chan, channels);
    }
    RR_FAIL(RR_ERR_BAD_ARG, "fused OLS bank: decimation %u not instantiated", a.D);
}

// ---------------------------------------------------------------------------
// Kernel 4w  k_filter_wave: the Filter block alone (filters.rs:240-259) for short responses (n <= 385
// taps), Complex<f32>, with ONE WAVE per 1024-sample block - k_ols_wave's structure without the
// mixer and the decimation:
//   y = IDFT_1024(DFT_1024(x_block) * H),  H = DFT_1024(g) / 1024,  V = ceil((n - 1) / 64) * 64,
// the last 1024 - V results of a block are valid (n = 64: 94 %).  The forward transform is k_ols_wave's
// (radix 8 x 16 x 8 on sample pairs); the inverse is the same routine on conj(Y) - Y leaves the forward
// transform as Y[l + 64 k] and goes through LDS once more to come back in the pair layout.  All
// exchanges are wave-local (no workgroup barrier); 16 waves per CU.
//
// SEL: the Downsampler for ANY periodic schedule (integer rates ra > rb, resampling.rs:103-112: input t releases an output when
// pos + (t + 1) rb crosses a multiple of ra) - the response is applied at every input position as above and only the results at
// the releasing positions are stored, out[m] for the m-th release.  Which of a lane's 16 positions release, and which m they
// carry, follows from F(t) = pos + t rb (mod ra, and its quotient) by whole-number steps: per block and per lane one reduction
// in f64 (exact below 2^53), then additions with a wrap - no table, no period length in the kernel.  HBM sees 8 B in and
// 8 rb / ra B out per sample; the arithmetic is the Filter's, whatever the ratio (48 000 -> 44 100: P : Q = 160 : 147).
template <bool SEL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RR_V_FLTWOCC, RR_V_FLTWOCC))) void k_filter_wave(
    const float2 *__restrict__ hist, int hist_len, const float2 *__restrict__ in, long n_in,
    const float2 *__restrict__ H, const float2 *__restrict__ tw, int V, float2 *__restrict__ out, long n_out, long e0,
    unsigned nblocks, const SelectArgs sel) {
    __shared__ __attribute__((aligned(16))) f2 lds[kWaveLds];
    const int l = threadIdx.x;
    // workgroups b, b + 8, .. share an XCD: neighbouring blocks (which share V samples) on one XCD
    // (in a moving window of 8 G blocks, G neighbouring blocks per XCD, as k_ols_wave)
    constexpr unsigned G = RR_V_FLTWWIN;
    const unsigned blk = blockIdx.x / (8 * G) * (8 * G) + (blockIdx.x % (8 * G) & 7) * G + (blockIdx.x % (8 * G) >> 3);
    if (blk >= nblocks) return;
    const int hop = 1024 - V;
    const long b0 = e0 - V + (long)blk * hop;
    f2 v[16];
    if (b0 >= 0 && b0 + 1024 <= n_in) {
        const f4u *src = reinterpret_cast<const f4u *>(in + b0) + l;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f4u x = ((RR_V_FLTWNT & 2) && (RR_V_FLTW_TAILK == 8 ? 128 * (k + 1) <= hop : k < RR_V_FLTW_TAILK)) ? __builtin_nontemporal_load(src + 64 * k)
                                                                                                                  : *(src + 64 * k);
            v[2 * k] = (f2){x.x, x.y};
            v[2 * k + 1] = (f2){x.z, x.w};
        }
    } else {
        // edges: the previous chunk in front (none after a reset), nothing behind the input
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long pos = b0 + 2 * l + j + 128 * k;
                const bool inr = pos >= 0 && pos < n_in;
                const bool hst = pos < 0 && pos >= -(long)hist_len;
                const float2 *ptr = inr ? in + pos : hist + (hst ? hist_len + pos : 0);
                f2 xv = {0.f, 0.f};
                if (inr || hst) {
                    const float2 xx = *ptr;
                    xv = (f2){xx.x, xx.y};
                }
                v[2 * k + j] = xv;
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
        for (int kp = 0; kp < 8; ++kp) h4[kp] = reinterpret_cast<const float4 *>(H)[l + 64 * kp];
    });
    // Z = conj(X H), then the transform run backwards: conj(y) in the pair layout
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
        const f2 p0 = cmul(X[2 * kp], (f2){h4[kp].x, h4[kp].y}), p1 = cmul(X[2 * kp + 1], (f2){h4[kp].z, h4[kp].w});
        X[2 * kp] = (f2){p0.x, -p0.y};
        X[2 * kp + 1] = (f2){p1.x, -p1.y};
    }
    wave_sync();  // the forward image has been read
    wave_dft1024_t(X, v, lds, l, t_p1, t_p2);
    if constexpr (SEL) {
        // F(t) = base0 + blk hop rb + t rb for the block's element t (position e0 - V + blk hop + t), base0 = pos + V (ra - rb) >= 0:
        // the V positions in front of the block's first result count as V releases more (C below), which keeps F non-negative.
        // Element t releases iff (F mod ra) + rb >= ra, and then carries m = floor(F / ra) - V.
        const double dra = (double)sel.ra;
        auto reduce = [&](double x, unsigned &r) -> unsigned {  // x = q ra + r, exact
            double q = __builtin_floor(x * sel.inv_ra);
            double rr = __builtin_fma(-q, dra, x);
            if (rr < 0.0) { rr += dra; q -= 1.0; }
            if (rr >= dra) { rr -= dra; q += 1.0; }
            r = (unsigned)rr;
            return (unsigned)q;
        };
        unsigned Rb, R;
        const unsigned qb = reduce((double)sel.base_r + (double)blk * (double)sel.hr, Rb);
        const long mbase = (long)sel.base_q + (long)blk * (long)sel.hq + (long)qb - V;  // m of a release at the block's F = Rb
        unsigned C = reduce((double)Rb + (double)(2 * l) * (double)sel.rb, R);         // releases since the block's F = Rb
        const long left = n_out - mbase;  // (> 0: a block holds a position of the call, its release lies within n_out or behind it)
        const unsigned recs = (unsigned)(left < 2048 ? left : 2048) * 8u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + mbase, 0, recs, 0x00020000);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = 2 * l + 128 * k;
            const f2 y0 = {v[2 * k].x, -v[2 * k].y}, y1 = {v[2 * k + 1].x, -v[2 * k + 1].y};
            const unsigned a0 = R + sel.rb;  // (< 2^32: ra < 2^31)
            const bool em0 = a0 >= sel.ra;
            const unsigned R1 = em0 ? a0 - sel.ra : a0, C1 = C + (em0 ? 1u : 0u);
            const bool em1 = R1 + sel.rb >= sel.ra;
            const unsigned off0 = (t >= V && em0) ? C * 8u : 0xffffffffu, off1 = (t >= V && em1) ? C1 * 8u : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b64(y0, rs, off0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(y1, rs, off1, 0, 0);
            const unsigned a2 = R + sel.kr;  // 128 positions on
            const bool w2 = a2 >= sel.ra;
            R = w2 ? a2 - sel.ra : a2;
            C += sel.kq + (w2 ? 1u : 0u);
        }
        return;
    }
    const long mb = (long)blk * hop;
    const long left = n_out - mb;
    const unsigned recs = (unsigned)(left < hop ? left : hop) * 8u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + mb, 0, recs, 0x00020000);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int t = 2 * l + 128 * k;  // V is even: a pair is valid or not as a whole
        const unsigned off = t >= V ? (unsigned)(t - V) * 8u : 0xffffffffu;
        const f2 y0 = {v[2 * k].x, -v[2 * k].y}, y1 = {v[2 * k + 1].x, -v[2 * k + 1].y};
        // (a pair that straddles the end of the output: only its first half is stored)
        if (off != 0xffffffffu && off + 16u > recs) {
            if (off + 8u <= recs) __builtin_amdgcn_raw_buffer_store_b64(y0, rs, off, 0, RR_V_FLTWNT & 1 ? 2 : 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128((f4){y0.x, y0.y, y1.x, y1.y}, rs, off, 0, RR_V_FLTWNT & 1 ? 2 : 0);
        }
    }
}

bool filter_wave_supported(int dtype, size_t n) { return dtype == RR_F32 && n >= 2 && n - 1 <= 384; }

int launch_filter_wave(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *H,
                       const void *tw, int V, void *out, size_t n_out, long e0) {
    if (n_out == 0) return RR_OK;
    const size_t hop = 1024 - V;
    const size_t nblocks = (n_out + hop - 1) / hop;
    if (nblocks > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Filter: too many blocks");
    const size_t grid = (nblocks + 8 * RR_V_FLTWWIN - 1) / (8 * RR_V_FLTWWIN) * (8 * RR_V_FLTWWIN);
    hipLaunchKernelGGL(k_filter_wave<false>, dim3((unsigned)grid), dim3(64), 0, s, (const float2 *)hist, (int)hist_len,
                       (const float2 *)in, (long)n_in, (const float2 *)H, (const float2 *)tw, V, (float2 *)out, (long)n_out,
                       e0, (unsigned)nblocks, SelectArgs{});
    RR_HIP(hipGetLastError());
    return RR_OK;
}

bool decim_select_supported(int dtype, uint64_t ra, uint64_t rb, size_t L) {
    return filter_wave_supported(dtype, L) && rb >= 1 && rb < ra && ra < (1ull << 31);
}

// The Downsampler through k_filter_wave<true>: out[m] = sum_i c[i] x[e_m - i] for the releases e_m of the periodic schedule
// (ra, rb, pos) among the n_in positions of the call; H = the tables of c = reverse(ir) as the Filter's.
int launch_decim_select(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *H, const void *tw,
                        int V, void *out, size_t n_out, uint64_t ra, uint64_t rb, uint64_t pos) {
    if (n_out == 0 || n_in == 0) return RR_OK;
    if (!(rb >= 1 && rb < ra && ra < (1ull << 31) && pos < ra) || V < 2 || V > 384 || (V & 1))
        RR_FAIL(RR_ERR_BAD_ARG, "Downsampler (select): rates %llu : %llu, pos %llu, overlap %d", (unsigned long long)ra,
                (unsigned long long)rb, (unsigned long long)pos, V);
    const size_t hop = 1024 - V;
    const size_t nblocks = (n_in + hop - 1) / hop;  // the response at every position of the call
    if (nblocks > 0x3fffffull) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler (select): too many blocks");  // (blk hr stays below 2^53)
    SelectArgs a;
    a.ra = (uint32_t)ra;
    a.rb = (uint32_t)rb;
    a.inv_ra = 1.0 / (double)ra;
    a.kr = (uint32_t)((128 * rb) % ra);
    a.kq = (uint32_t)((128 * rb) / ra);
    a.hr = (uint32_t)((hop * rb) % ra);
    a.hq = (uint32_t)((hop * rb) / ra);
    const uint64_t base0 = pos + (uint64_t)V * (ra - rb);
    a.base_r = (uint32_t)(base0 % ra);
    a.base_q = (uint32_t)(base0 / ra);
    const size_t grid = (nblocks + 8 * RR_V_FLTWWIN - 1) / (8 * RR_V_FLTWWIN) * (8 * RR_V_FLTWWIN);
    hipLaunchKernelGGL(k_filter_wave<true>, dim3((unsigned)grid), dim3(64), 0, s, (const float2 *)hist, (int)hist_len,
                       (const float2 *)in, (long)n_in, (const float2 *)H, (const float2 *)tw, V, (float2 *)out, (long)n_out,
                       0l, (unsigned)nblocks, a);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int ols_decim_overlap(size_t Lc) {  // V: multiple of 256 covering the Lc - 1 wrapped samples
    const size_t v = (Lc - 1 + 255) / 256 * 256;
    return v == 0 ? 256 : (int)v;
}

bool ols_decim_supported(uint64_t D, size_t Lc) { return D == 4 && Lc >= 1 && Lc - 1 <= 2048; }

int launch_ols_decim(hipStream_t s, const FusedFirArgs &a) {
    if (a.n_out == 0) return RR_OK;
    const int per_block = (4096 - a.V) / 4;
    const size_t nblocks = (a.n_out + per_block - 1) / per_block;
    if (nblocks > 0x7fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS: too many blocks");
    hipLaunchKernelGGL(k_ols_decim4, dim3((unsigned)nblocks), dim3(256), 0, s, (const float2 *)a.xh, (int)a.hx,
                       (const float2 *)a.in, (long)a.n_in, (const float2 *)a.nco, a.denom, a.idx0,
                       (const float2 *)a.H, (const float2 *)a.tw4096, a.V, (float2 *)a.out, (long)a.n_out, (long)a.e0,
                       (float2 *)a.xh_out, (int)a.hx);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

bool filter_ols4096_supported(int dtype, size_t n) { return dtype == RR_F32 && n >= 129 && n <= 2048; }

// tail drop on an interrupt: new[i] = old[i - drop] (zeros shifted in at the front)
__global__ void k_drop_tail(const float2 *__restrict__ oldh, float2 *__restrict__ newh, int H, int drop) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    float2 v;
    v.x = 0.f;
    v.y = 0.f;
    if (i >= drop) v = oldh[i - drop];
    newh[i] = v;
}


int launch_drop_tail(hipStream_t s, const void *oldh, void *newh, size_t H, size_t drop) {
    if (H == 0) return RR_OK;
    hipLaunchKernelGGL(k_drop_tail, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, s, (const float2 *)oldh,
                       (float2 *)newh, (int)H, (int)drop);
    RR_HIP(hipGetLastError());
    return RR_OK;
}


}  // namespace rr
