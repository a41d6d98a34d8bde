// rr_ols_wave2k.hip — k_ols_wave2k: the fused mixer + FIR + decimation by 8 (rr_chain's front end, the stand-alone Downsampler at
// 8 : 1; transform.rs:171-260, filters.rs:240-259, resampling.rs:20-134 in one pass) with ONE WAVE per block of 2048 samples.
//
// k_ols_wave<8> (rr_ols.hip) transforms 1024 samples per wave and keeps (1024 - V) / 8 results; the 8 : 1 chains carry long
// combined responses (the bench shape: 351 taps, V = 352: 66 % of a block kept, 1.52 samples transformed per sample), and its
// inverse - 128 points - keeps half of the wave busy.  Here a block is 2048 samples (V = 352: 83 % kept, 1.21) and the inverse
// has 256 points, a full wave.  In polyphase form (as k_ols_wave<D, POLY>), with x_p[m] = xs[8 m + p], p < 8, m < 256:
//   Y[k] = sum_p X_p[k] G_p[k],  X_p = DFT_256 x_p,  G_p[k] = sum_q H[k + 256 q] W_2048^((k + 256 q) p),  H = DFT_2048(c) / 2048,
//   result[tau] = IDFT_256(Y)[tau] = sum_i c[i] xs[8 tau - i]   (tau >= V / 8)
// - eight 256-point transforms, which are exactly TWO runs of the forward part of k_ols_wave<4, POLY> (four 256-point
// transforms of the phases 2 (l & 1) + j, radix 8 x 8 x 4 through two wave-local exchanges): half h takes the phases
// p = 4 h + 2 (l & 1) + j, i.e. the samples xs[4 l - 2 (l & 1) + 4 h + j + 256 k'], j < 2, k' < 8 - per k' and half one 16-byte
// load per lane, the two halves' loads interleaved on the same lines.  The 16 values of a half live in registers only while it
// runs; the products with G_p accumulate in the four bins Y[l + 64 c] a lane keeps; the inverse is k_ols_wave<4>'s.
//
// Mixer: MF = the samples as they are (the Downsampler: every phasor 1), GP = the mixer BEHIND the filter for any NCO period
// (tables of c[i] w^-i, rr_chain::ensure_genfold; a result at b0 + 8 tau is multiplied by the phase table's entry there: the
// block's phasor x the lane's w^(8 l) x the rotations by 512 samples), otherwise the mixer in front of the transform by a walk
// through the phase table (the call behind a retune only: slow and plain).
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"
#include "rr_fft_regs.hpp"
#include "rr_ols_dev.hpp"

#include <hip/hip_ext.h>

#include <cstdlib>

namespace rr {

#ifndef RR_V_OLSW2K_WIN
#define RR_V_OLSW2K_WIN 32
#endif
#ifndef RR_V_OLSW2K_NT
#define RR_V_OLSW2K_NT 0  // the streaming hint on the block's loads: 0 none, 1 both halves, 2 the second half's (A/B runs, below)
#endif
// Measured per 2^26 samples (scripts/wave2k_probe.py, one session; k_ols_wave<8> beside it):
//                              L = 128   L = 288   L = 461   chain 8 : 1 / FFT 1024   / FFT 4096
//   k_ols_wave<8>              0.110     0.126     0.162     0.169                     0.166 ms
//   k_ols_wave2k               0.109     0.113     0.123     0.151                     0.151
// - with the streaming hint on the loads (as k_ols_wave) it was SLOWER than k_ols_wave<8> (0.142 / 0.154 / 0.170 / 0.183): every
//   line is touched by two load instructions (the two halves), and the hint drops it behind the first;
// - 4 waves per SIMD (128 registers): 58 .. 66 spilled, 0.22 .. 0.26 ms; 2 waves per SIMD: 3 % behind 3;
// - windows of 16 / 64 blocks per XCD: the same within 1 %;
// - the stand-alone Downsampler's stores with the streaming hint: 0.118 -> 0.113 (L = 288); the chain's without (the Fourier
//   kernel reads them next).
constexpr unsigned kWave2kWin = RR_V_OLSW2K_WIN;  // blocks dealt to the XCDs in a moving window, that many neighbouring blocks per XCD (k_ols_wave: 64 of half the size)
#ifndef RR_V_OLSW2K_OCC
#define RR_V_OLSW2K_OCC 3  // 156 registers per lane; at 4 waves per SIMD (128) the kernel spills 58 .. 66
#endif

template <bool MF, bool GP>
__device__ __forceinline__ void ols_wave2k_body(const float2 *__restrict__ xh, int hx, const float2 *__restrict__ in, long n_in,
                                                const float2 *__restrict__ nco, unsigned denom, unsigned idx0,
                                                const float2 *__restrict__ G, const float2 *__restrict__ tw, int V,
                                                float2 *__restrict__ out, long n_out, long e0, float2 *__restrict__ xh_out, int hx_out,
                                                unsigned nblocks, unsigned ph0, unsigned hopm, unsigned kstep, double inv_denom,
                                                const unsigned bx, const unsigned Gw) {
    static_assert(!GP || MF, "the mixer behind the filter: the blocks transform the samples as they are");
    __shared__ __attribute__((aligned(16))) f2 lds[1136];  // (k_ols_wave<4, POLY>'s image: 2 (63 + 72 * 7) + 2 elements)
    const int l = threadIdx.x;
    const unsigned grp = bx / (8 * Gw), rem = bx % (8 * Gw);
    const unsigned blk = grp * 8 * Gw + (rem & 7) * Gw + (rem >> 3);
    if (blk >= nblocks) return;
    const int hop = 2048 - V, per_block = hop >> 3;
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
    // NCO phase of the block's first sample: (idx0 + b0) mod denom = (ph0 + blk hopm) mod denom, reduced in f64 (exact below 2^53)
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
    const int off = 4 * l - 2 * (l & 1);  // the lane's first sample of half 0; half 1: + 4; k': + 256
    const bool interior = b0 >= 0 && b0 + 2048 <= n_in;
    f4u x[2][8];
    if (MF && interior) {
        const f4u *src = reinterpret_cast<const f4u *>(in + b0 + off);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x[0][k] = RR_V_OLSW2K_NT == 1 ? ld_stream(src + 128 * k) : *(src + 128 * k);
            x[1][k] = RR_V_OLSW2K_NT >= 1 ? ld_stream(src + 128 * k + 2) : *(src + 128 * k + 2);
        }
    }
    [[maybe_unused]] float2 gpb, gpl, rot256, rot512;
    if constexpr (GP) {
        gpb = ld_uniform(nco + __builtin_amdgcn_readfirstlane(base));
        gpl = nco[denom + 9 + 128 + l];                 // w^(8 (l mod 32))
        rot256 = ld_uniform(nco + (denom + 1 + 2));     // w^256
        rot512 = ld_uniform(nco + (denom + 1 + 4));     // w^512
    }
    // lane constants: tw[4 (l >> 1)], tw[32 (l >> 3)] and the three seeds of the inverse (append_wave1024_seeds, as k_ols_wave<4, POLY>)
    f2 t_p1, t_p2, t_inv[3];
    {
        const float4 *tl = reinterpret_cast<const float4 *>(tw + 1024) + l;
        const float4 s6 = tl[384], s7 = tl[448], s8 = tl[512];
        t_p1 = (f2){s6.x, s6.y};
        t_p2 = (f2){s6.z, s6.w};
        t_inv[0] = (f2){s7.x, s7.y};
        t_inv[1] = (f2){s7.z, s7.w};
        t_inv[2] = (f2){s8.x, s8.y};
    }
    f2 v[2][16];  // v[h][2 k' + j]
    if (MF && interior) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[h][2 * k] = (f2){x[h][k].x, x[h][k].y};
                v[h][2 * k + 1] = (f2){x[h][k].z, x[h][k].w};
            }
    } else {
        // edges (history - already mixed - in front, nothing behind the input) and the mixer in front: element by element, every
        // lane reads some valid address and selects afterwards; the phase index walks the table in steps of 256 samples
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned r = (base + (unsigned)(off + 4 * h)) % denom;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f4u pp = *reinterpret_cast<const f4u *>(nco + r);  // (entry 0 once more behind entry denom - 1)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const long pos = b0 + off + 4 * h + j + 256 * k;
                    const bool inr = pos >= 0 && pos < n_in;
                    const bool hst = pos < 0 && pos >= -(long)hx;
                    const float2 *ptr = inr ? in + pos : xh + (hst ? hx + pos : 0);
                    const float2 xx = *ptr;
                    const f2 p = j ? (f2){pp.z, pp.w} : (f2){pp.x, pp.y};
                    // (MF: the block wants the samples UNMIXED - the history, which holds mixed ones, times conj(p))
                    const f2 pk = MF ? (f2){inr ? 1.f : (hst ? p.x : 0.f), hst ? -p.y : 0.f}
                                     : (f2){inr ? p.x : (hst ? 1.f : 0.f), inr ? p.y : 0.f};
                    const f2 xv = {(inr || hst) ? xx.x : 0.f, (inr || hst) ? xx.y : 0.f};
                    v[h][2 * k + j] = cmul(xv, pk);
                }
                r += kstep;
                if (r >= denom) r -= denom;
            }
        }
    }

    f2 y[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // the lane's 16 entries of G_p for the half's four phases: piece [h][i >> 1][l], entry i = 4 pp + c in half (i & 1)
        float4 ga[8];
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) ga[kp] = reinterpret_cast<const float4 *>(G)[512 * h + l + 64 * kp];
        f2 e0_[8], e1_[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            e0_[k] = v[h][2 * k];
            e1_[k] = v[h][2 * k + 1];
        }
        dft8(e0_);
        dft8(e1_);
        {   // * W_256^(mu kappa1): powers of one seed
            const f2 w1 = t_p1, w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
            const f2 w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
            e0_[1] = cmul(e0_[1], w1); e1_[1] = cmul(e1_[1], w1);
            e0_[2] = cmul(e0_[2], w2); e1_[2] = cmul(e1_[2], w2);
            e0_[3] = cmul(e0_[3], w3); e1_[3] = cmul(e1_[3], w3);
            e0_[4] = cmul(e0_[4], w4); e1_[4] = cmul(e1_[4], w4);
            e0_[5] = cmul(e0_[5], w5); e1_[5] = cmul(e1_[5], w5);
            e0_[6] = cmul(e0_[6], w6); e1_[6] = cmul(e1_[6], w6);
            e0_[7] = cmul(e0_[7], w7); e1_[7] = cmul(e1_[7], w7);
        }
        if (h) wave_sync();  // the first half's last reads are done
        {
            f2 *row = lds + 2 * l;
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<float4 *>(row + 144 * k) = (float4){e0_[k].x, e0_[k].y, e1_[k].x, e1_[k].y};
        }
        wave_sync();
        {
            const f2 *col = lds + 2 * ((l & 7) + 72 * (l >> 3));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 r = *reinterpret_cast<const float4 *>(col + 16 * k);
                e0_[k] = (f2){r.x, r.y};
                e1_[k] = (f2){r.z, r.w};
            }
        }
        dft8(e0_);
        dft8(e1_);
        wave_sync();
        {
            f2 *row = lds + 2 * ((l >> 3) + 65 * (l & 7));  // (planes 130 elements apart: k_ols_wave's exchange 2)
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<float4 *>(row + 16 * k) = (float4){e0_[k].x, e0_[k].y, e1_[k].x, e1_[k].y};
        }
        wave_sync();
        const f2 w1 = t_p2, w2 = cmul(w1, w1), w3 = cmul(w2, w1);
        // the half's phases two at a time (as poly4_block): 8 values of the image in registers instead of 16
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f2 d[2][4];
#pragma unroll
            for (int m1 = 0; m1 < 4; ++m1) {
                const float4 r = *reinterpret_cast<const float4 *>(lds + 2 * l + 130 * (a + 2 * m1));
                d[0][m1] = (f2){r.x, r.y};
                d[1][m1] = (f2){r.z, r.w};
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                d[pp][1] = cmul(d[pp][1], w1);
                d[pp][2] = cmul(d[pp][2], w2);
                d[pp][3] = cmul(d[pp][3], w3);
                dft4(d[pp][0], d[pp][1], d[pp][2], d[pp][3]);
            }
            // phase 2 a + pp of the half = pieces 2 (2 a + pp), 2 (2 a + pp) + 1
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const float4 g0 = ga[4 * a + 2 * pp], g1 = ga[4 * a + 2 * pp + 1];
                if (h == 0 && a == 0 && pp == 0) {
                    y[0] = cmul(d[pp][0], (f2){g0.x, g0.y});
                    y[1] = cmul(d[pp][1], (f2){g0.z, g0.w});
                    y[2] = cmul(d[pp][2], (f2){g1.x, g1.y});
                    y[3] = cmul(d[pp][3], (f2){g1.z, g1.w});
                } else {
                    y[0] = cmac(y[0], d[pp][0], (f2){g0.x, g0.y});
                    y[1] = cmac(y[1], d[pp][1], (f2){g0.z, g0.w});
                    y[2] = cmac(y[2], d[pp][2], (f2){g1.x, g1.y});
                    y[3] = cmac(y[3], d[pp][3], (f2){g1.z, g1.w});
                }
            }
        }
    }
    // ---- inverse DFT_256 (radix 4 x 4 x 4 x 4, as k_ols_wave<4>: one image layout per exchange, inv256_rd) ----
    const int g = l >> 4, q = l & 15;
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
        const f2 w1 = t_inv[pass - 1];
        const f2 w2 = cmul(w1, w1);
        const f2 w3 = cmul(w2, w1);
        y[1] = cmul_conj(y[1], w1);
        y[2] = cmul_conj(y[2], w2);
        y[3] = cmul_conj(y[3], w3);
        idft4(y[0], y[1], y[2], y[3]);
        if (pass == 3) break;  // natural order: y[c] = result[l + 64 c]
        wave_sync();
        if (pass == 1) {
            f2 *col = lds + (20 * (l >> 2) + (l & 3));
#pragma unroll
            for (int c = 0; c < 4; ++c) lds_st(col + (4 * c), y[c]);
        } else {
            f2 *col = lds + (80 * g + q);
#pragma unroll
            for (int c = 0; c < 4; ++c) lds_st(col + (16 * c), y[c]);
        }
        wave_sync();
    }
    if constexpr (GP) {  // result tau = l + 64 c at b0 + 8 tau: the block's phasor x w^(8 l) x w^(512 c)
        f2 gph = cmul((f2){gpb.x, gpb.y}, (f2){gpl.x, gpl.y});
        if (l >= 32) gph = cmul(gph, (f2){rot256.x, rot256.y});
        const f2 r1 = {rot512.x, rot512.y}, r2 = cmul(r1, r1), r3 = cmul(r2, r1);
        y[0] = cmul(y[0], gph);
        y[1] = cmul(y[1], cmul(gph, r1));
        y[2] = cmul(y[2], cmul(gph, r2));
        y[3] = cmul(y[3], cmul(gph, r3));
    }
    // the valid part by buffer stores: lanes outside it (and behind the end of the output) carry an out-of-range offset
    const int first = V >> 3;
    const long mb = (long)blk * per_block;
    const long left = n_out - mb;
    const unsigned recs = (unsigned)(left < per_block ? left : per_block) * 8u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + mb, 0, recs, 0x00020000);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int tau = l + 64 * c;
        const unsigned o = tau >= first ? (unsigned)(tau - first) * 8u : 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b64(y[c], rs, o, 0, (MF && !GP) ? 2 : 0);
    }
}

template <bool MF, bool GP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RR_V_OLSW2K_OCC, RR_V_OLSW2K_OCC))) void k_ols_wave2k(
    const float2 *__restrict__ xh, int hx, const float2 *__restrict__ in, long n_in, const float2 *__restrict__ nco, unsigned denom,
    unsigned idx0, const float2 *__restrict__ G, const float2 *__restrict__ tw, int V, float2 *__restrict__ out, long n_out, long e0,
    float2 *__restrict__ xh_out, int hx_out, unsigned nblocks, unsigned ph0, unsigned hopm, unsigned kstep, double inv_denom) {
    ols_wave2k_body<MF, GP>(xh, hx, in, n_in, nco, denom, idx0, G, tw, V, out, n_out, e0, xh_out, hx_out, nblocks, ph0, hopm, kstep,
                            inv_denom, blockIdx.x, kWave2kWin);
}

// the channels of a bank (rr_chainbank), as k_ols_wave_bank: channel = blockIdx.y
template <bool MF, bool GP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RR_V_OLSW2K_OCC, RR_V_OLSW2K_OCC))) void k_ols_wave2k_bank(
    const BankTable chan, int hx, long n_in, const float2 *__restrict__ nco, unsigned denom, unsigned idx0,
    const float2 *__restrict__ G, const float2 *__restrict__ tw, int V, long n_out, long e0, int hx_out, unsigned nblocks,
    unsigned ph0, unsigned hopm, unsigned kstep, double inv_denom, unsigned gwin) {
    const BankPtrs c = chan.c[blockIdx.y];
    ols_wave2k_body<MF, GP>((const float2 *)c.xh, hx, (const float2 *)c.in, n_in, nco, denom, idx0, G, tw, V, (float2 *)c.dec, n_out, e0,
                            (float2 *)c.xh_out, hx_out, nblocks, ph0, hopm, kstep, inv_denom, blockIdx.x, gwin);
}

// 8 : 1 with a combined response of up to 1025 taps (an overlap of at most half a block); RR_OLSW_2K=0 keeps k_ols_wave<8>
bool ols_wave2k_supported(uint64_t D, size_t Lc) {
    const char *e = std::getenv("RR_OLSW_2K");  // (read per design: tests switch it within one process)
    const bool off = e && std::atoi(e) == 0;
    return !off && D == 8 && Lc >= 1 && Lc - 1 <= 1024;
}

struct Wave2kGeom {
    size_t nblocks;
    unsigned ph, hopm, kstep;
    double inv_den;
};
static int wave2k_geom(const FusedFirArgs &a, Wave2kGeom &g) {
    if (a.D != 8 || a.V < 16 || a.V > 1024 || a.V % 16) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS (2048-sample blocks): D %u, overlap %d", a.D, a.V);
    const int per_block = (2048 - a.V) / 8;
    g.nblocks = (a.n_out + per_block - 1) / per_block;
    if (g.nblocks > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS: too many blocks");
    const int64_t den = (int64_t)a.denom;
    int64_t ph = ((int64_t)a.idx0 + a.e0 - a.V) % den;
    if (ph < 0) ph += den;
    g.ph = (unsigned)ph;
    g.hopm = (unsigned)((int64_t)(2048 - a.V) % den);
    g.kstep = (unsigned)(256 % den);
    g.inv_den = 1.0 / (double)den;
    return RR_OK;
}

int launch_ols_wave2k(hipStream_t s, const FusedFirArgs &a) {
    if (a.n_out == 0) return RR_OK;
    Wave2kGeom g;
    RR_TRY(wave2k_geom(a, g));
    const unsigned grid = (unsigned)((g.nblocks + 8 * kWave2kWin - 1) / (8 * kWave2kWin) * (8 * kWave2kWin));
#define RR_OLSW2K_LAUNCH(MF_, GP_)                                                                                                    \
    do {                                                                                                                              \
        if (a.ev_start && a.ev_stop)                                                                                                  \
            hipExtLaunchKernelGGL((k_ols_wave2k<MF_, GP_>), dim3(grid), dim3(64), 0, s, a.ev_start, a.ev_stop, 0, (const float2 *)a.xh, \
                                  (int)a.hx, (const float2 *)a.in, (long)a.n_in, (const float2 *)a.nco, a.denom, a.idx0,             \
                                  (const float2 *)a.H, (const float2 *)a.tw4096, a.V, (float2 *)a.out, (long)a.n_out, (long)a.e0,    \
                                  (float2 *)a.xh_out, (int)a.hx, (unsigned)g.nblocks, g.ph, g.hopm, g.kstep, g.inv_den);             \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_ols_wave2k<MF_, GP_>), dim3(grid), dim3(64), 0, s, (const float2 *)a.xh, (int)a.hx,                 \
                               (const float2 *)a.in, (long)a.n_in, (const float2 *)a.nco, a.denom, a.idx0, (const float2 *)a.H,      \
                               (const float2 *)a.tw4096, a.V, (float2 *)a.out, (long)a.n_out, (long)a.e0, (float2 *)a.xh_out,        \
                               (int)a.hx, (unsigned)g.nblocks, g.ph, g.hopm, g.kstep, g.inv_den);                                    \
    } while (0)
    if (a.genfold) RR_OLSW2K_LAUNCH(true, true);
    else if (a.mixfold) RR_OLSW2K_LAUNCH(true, false);
    else RR_OLSW2K_LAUNCH(false, false);
#undef RR_OLSW2K_LAUNCH
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int launch_ols_wave2k_bank(hipStream_t s, const FusedFirArgs &a, const BankTable &d_chan, size_t channels) {
    if (a.n_out == 0 || channels == 0) return RR_OK;
    if (channels > kBankGroup) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS bank: too many channels");
    Wave2kGeom g;
    RR_TRY(wave2k_geom(a, g));
    const unsigned gwin = g.nblocks >= 2048 ? kWave2kWin : (g.nblocks >= 64 ? 8u : 1u);
    const unsigned grid = (unsigned)((g.nblocks + 8 * gwin - 1) / (8 * gwin) * (8 * gwin));
#define RR_OLSW2KB_LAUNCH(MF_, GP_)                                                                                                  \
    hipLaunchKernelGGL((k_ols_wave2k_bank<MF_, GP_>), dim3(grid, (unsigned)channels), dim3(64), 0, s, d_chan, (int)a.hx, (long)a.n_in, \
                       (const float2 *)a.nco, a.denom, a.idx0, (const float2 *)a.H, (const float2 *)a.tw4096, a.V, (long)a.n_out,   \
                       (long)a.e0, (int)a.hx, (unsigned)g.nblocks, g.ph, g.hopm, g.kstep, g.inv_den, gwin)
    if (a.genfold) RR_OLSW2KB_LAUNCH(true, true);
    else if (a.mixfold) RR_OLSW2KB_LAUNCH(true, false);
    else RR_OLSW2KB_LAUNCH(false, false);
#undef RR_OLSW2KB_LAUNCH
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
