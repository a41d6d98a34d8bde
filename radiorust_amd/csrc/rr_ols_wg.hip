// rr_ols_wg.hip — k_ols_wg<NW>: the fused mixer + FIR + decimation by D = 5 .. 64 (first 16, 32, 64; rr_chain's front end, the stand-alone
// Downsampler; transform.rs:171-260, filters.rs:240-259, resampling.rs:20-134 in one pass) by overlap-save with ONE WORKGROUP of
// NW = D / 4 waves per block of N = 256 D samples.
//
// At these ratios the polyphase decimator (k_decim_poly, direct form) pays L / D products per input sample, and L grows with D
// (16 : 1 with 960 taps 0.41 ms per 2^26 samples, 64 : 1 with 3840 taps 2.2 ms); overlap-save costs the same whatever L.  In
// polyphase form (as k_ols_wave2k, rr_ols_wave2k.hip), with x_p[m] = xs[D m + p], p < D, m < 256:
//   Y[k] = sum_p X_p[k] G_p[k],  X_p = DFT_256 x_p,  G_p[k] = sum_q H[k + 256 q] W_N^((k + 256 q) p),  H = DFT_N(c) / N,
//   result[tau] = IDFT_256(Y)[tau] = sum_i c[i] xs[D tau - i]   (tau >= V / D)
// - D transforms of 256 points = D / 4 runs of the four-phase forward transform of k_ols_wave<4, POLY> (radix 8 x 8 x 4 through two
// wave-local exchanges), ONE RUN PER WAVE: wave w takes the phases 4 w .. 4 w + 3.  The block's samples come in by loads that are
// contiguous across the workgroup and go through LDS to the wave and lane that transform them (a run reads 4 of every D samples:
// 32 bytes of every line - loaded by the waves themselves, every line was requested NW times).  Each wave multiplies its four
// transforms by its G_p into the four bins Y[l + 64 c] a lane keeps; the waves 1 .. NW - 1 then leave their bins in their
// exchange images, and wave 0 adds them up, runs the 256-point inverse (k_ols_wave<4>'s) and stores the (N - V) / D results.
// LDS: NW images of 9 KiB - 4 / 2 / 1 workgroups = 16 waves per CU.
//
// Measured, the stand-alone Downsampler per 2^26 samples (scripts/decim_pow2_probe.py, one session; k_decim_poly -> this kernel):
//   16 : 1 with 240 / 480 / 960 taps      0.146 / 0.230 / 0.411  ->  0.116 / 0.123 / 0.138 ms
//   32 : 1 with 480 / 960 / 1920 taps     0.191 / 0.363 / 0.728  ->  0.140 / 0.143 / 0.161
//   64 : 1 with 960 / 1920 / 3840 taps    0.323 / 0.587 / 2.247  ->  0.163 / 0.177 / 0.196
// First forms, one session each: four waves with D / 16 runs each, every wave loading its own phases: 16 : 1 0.135, 32 : 1 0.162,
// 64 : 1 0.167 ms - TCP -> L2 requests 2 - 3 x those of k_ols_wave2k (PMC); the same with the workgroup-contiguous loads per step:
// 16 : 1 0.115 (one step), but 32 : 1 0.179 and 64 : 1 0.176 (two barriers per step, the loads' latency in front of every step).
//
// Mixer: MF = the samples as they are (the Downsampler), GP = the mixer BEHIND the filter for any NCO period (tables of
// c[i] w^-i, rr_chain::ensure_genfold; a result at b0 + D tau is multiplied by the phase table's own entry at that position),
// otherwise the mixer in front of the transform by a walk through the phase table (the call behind a retune: slow and plain).
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"
#include "rr_fft_regs.hpp"
#include "rr_ols_dev.hpp"

#include <hip/hip_ext.h>

#include <cstdlib>

namespace rr {

#ifndef RR_V_OLSWG_COOP
#define RR_V_OLSWG_COOP 1  // the block's samples by workgroup-contiguous loads and through LDS to their lanes (0: every wave loads its own phases)
#endif
#ifndef RR_V_OLSWG_WIN
#define RR_V_OLSWG_WIN 8
#endif
#ifndef RR_V_OLSWG_NT
#define RR_V_OLSWG_NT 1  // the streaming hint on the staging loads (windows of 2 / 8 / 32 blocks per XCD, the hint on / off: all within 2 %)
#endif
constexpr unsigned kWgWin = RR_V_OLSWG_WIN;  // blocks dealt to the XCDs in a moving window, that many neighbouring blocks per XCD
constexpr int kWgImg = 1140;    // a wave's exchange image (k_ols_wave<4, POLY>'s 1136 elements: 2 (63 + 72 * 7) + 2) + 4: the images 8 banks apart

template <int NW, bool MF, bool GP>
__global__ __launch_bounds__(64 * NW) void k_ols_wg(const float2 *__restrict__ xh, int hx, const float2 *__restrict__ in, long n_in,
                                                   const float2 *__restrict__ nco, unsigned denom, unsigned idx0,
                                                   const float2 *__restrict__ G, const float2 *__restrict__ tw, int V,
                                                   float2 *__restrict__ out, long n_out, long e0, float2 *__restrict__ xh_out, int hx_out,
                                                   unsigned nblocks, unsigned ph0, unsigned hopm, unsigned kstep, double inv_denom,
                                                   const int D) {
    static_assert(!GP || MF, "the mixer behind the filter: the blocks transform the samples as they are");
    // D: the decimation, even, 4 (NW - 1) < D <= 4 NW as a rule (more waves than runs: the last ones find no phase and add zeros)
    constexpr int NT = 64 * NW;
    const int N = 256 * D, Dh = D >> 1;
    extern __shared__ __attribute__((aligned(16))) f2 wg_smem[];  // NW images
    f2 *const smem = wg_smem;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    f2 *const lds = smem + w * kWgImg;
    const unsigned bx = blockIdx.x;
    const unsigned grp = bx / (8 * kWgWin), rem = bx % (8 * kWgWin);
    const unsigned blk = grp * 8 * kWgWin + (rem & 7) * kWgWin + (rem >> 3);
    if (blk >= nblocks) return;
    const int hop = N - V, per_block = hop / D;
    const long b0 = e0 - V + (long)blk * hop;

    if (xh_out && blk == nblocks - 1) {  // mixed-sample history for the next call
        for (int i = tid; i < hx_out; i += NT) {
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
    const bool interior = b0 >= 0 && b0 + N <= n_in;
    const bool fast = MF && interior;
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
    f2 y[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    {
        const int h = w;                                           // this wave's run: the phases 4 w .. 4 w + 3
        const int off = D * (l >> 1) + 4 * h + 2 * (l & 1);        // the lane's first sample (j = 0, k' = 0); k': + 32 D
        // (phases beyond D - the last wave's, where D is no multiple of 4 - are zeros: live0 / live1 = the lane's phases j = 0 / 1 exist)
        const bool live0 = 4 * h + 2 * (l & 1) < D, live1 = 4 * h + 2 * (l & 1) + 1 < D, live = live0;
        f2 e0_[8], e1_[8];
        if (RR_V_OLSWG_COOP && fast) {
            // the block's N samples by loads that are contiguous across the workgroup (thread t takes the 16-byte chunks t, t + NT,
            // ..: every line is requested once, by one instruction - a wave that loads its own 4 of every D samples asks for 32 bytes
            // of each line, and the NW requests for a line reach L2 one by one: 2 - 3 x the requests, PMC), then through LDS to the
            // wave and lane that transform them: chunk c of period m is the pair a = c & 1 of wave c >> 1, lane 2 (m & 31) + a, value
            // k' = m >> 5 - the slot that lane's first exchange writes anyway
            if (D & 1) {
                // odd D: a period is no whole number of 16-byte chunks - sample by sample (8-byte loads, still contiguous across the
                // workgroup): sample n = D m + p goes to wave p >> 2, lane 2 (m & 31) + ((p >> 1) & 1), value k' = m >> 5, half p & 1
                const f2 *src1 = reinterpret_cast<const f2 *>(in + b0);
                const int ns = 256 * D;  // at most 16 per thread (4 D / NW)
                f2 s1[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int n = tid + NT * u;
                    s1[u] = __builtin_nontemporal_load(src1 + (n < ns ? n : ns - 1));
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int n = tid + NT * u, m = n / D, pq = n - m * D;
                    f2 *dst = smem + (pq >> 2) * kWgImg + 2 * (2 * (m & 31) + ((pq >> 1) & 1)) + (pq & 1) + 144 * (m >> 5);
                    if (n < ns) lds_st(dst, s1[u]);
                }
            } else {
            const f4u *src = reinterpret_cast<const f4u *>(in + b0);
            const int nch = 128 * D;  // 16-byte chunks of the block: at most 8 per thread (2 D / NW)
            f4u ch[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + NT * u;
                ch[u] = RR_V_OLSWG_NT ? ld_stream(src + (q < nch ? q : nch - 1)) : *(src + (q < nch ? q : nch - 1));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + NT * u, m = q / Dh, c = q - m * Dh;
                f2 *dst = smem + (c >> 1) * kWgImg + 2 * (2 * (m & 31) + (c & 1)) + 144 * (m >> 5);
                if (q < nch) *reinterpret_cast<float4 *>(dst) = (float4){ch[u].x, ch[u].y, ch[u].z, ch[u].w};
            }
            }
            __syncthreads();
            const f2 *row = lds + 2 * l;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 q4 = *reinterpret_cast<const float4 *>(row + 144 * k);
                e0_[k] = live0 ? (f2){q4.x, q4.y} : (f2){0.f, 0.f};
                e1_[k] = live1 ? (f2){q4.z, q4.w} : (f2){0.f, 0.f};
            }
        } else if (fast) {
            const f4u *src = reinterpret_cast<const f4u *>(in + b0 + (live ? off : 0));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f4u x = *(src + 16 * D * k);
                e0_[k] = live0 ? (f2){x.x, x.y} : (f2){0.f, 0.f};
                e1_[k] = live1 ? (f2){x.z, x.w} : (f2){0.f, 0.f};
            }
        } else {
            // edges (history - already mixed - in front, nothing behind the input) and the mixer in front: element by element, every
            // lane reads some valid address and selects afterwards; the phase index walks the table in steps of 32 D samples
            unsigned rr_ = (unsigned)(((unsigned long long)base + (unsigned)off) % denom);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f4u pp = *reinterpret_cast<const f4u *>(nco + rr_);  // (entry 0 once more behind entry denom - 1)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const long pos = b0 + off + j + 32 * D * k;
                    const bool inr = pos >= 0 && pos < n_in;
                    const bool hst = pos < 0 && pos >= -(long)hx;
                    const float2 *ptr = inr ? in + pos : xh + (hst ? hx + pos : 0);
                    const float2 xx = *ptr;
                    const f2 p = j ? (f2){pp.z, pp.w} : (f2){pp.x, pp.y};
                    // (MF: the block wants the samples UNMIXED - the history, which holds mixed ones, times conj(p))
                    const f2 pk = MF ? (f2){inr ? 1.f : (hst ? p.x : 0.f), hst ? -p.y : 0.f}
                                     : (f2){inr ? p.x : (hst ? 1.f : 0.f), inr ? p.y : 0.f};
                    const bool lv = j ? live1 : live0;
                    const f2 xv = {((inr || hst) && lv) ? xx.x : 0.f, ((inr || hst) && lv) ? xx.y : 0.f};
                    (j ? e1_[k] : e0_[k]) = cmul(xv, pk);
                }
                rr_ += kstep;
                if (rr_ >= denom) rr_ -= denom;
            }
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
                const float4 q4 = *reinterpret_cast<const float4 *>(col + 16 * k);
                e0_[k] = (f2){q4.x, q4.y};
                e1_[k] = (f2){q4.z, q4.w};
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
        // the lane's 16 entries of G_p for the run's four phases: piece [h][i >> 1][l], entry i = 4 pp + c in half (i & 1) - requested
        // here, behind the second exchange's stores (in front of the run they were 32 more registers through both exchanges, beside the
        // 32 of the next step's samples)
        float4 ga[8];
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) ga[kp] = reinterpret_cast<const float4 *>(G)[512 * h + l + 64 * kp];
        wave_sync();
        const f2 w1 = t_p2, w2 = cmul(w1, w1), w3 = cmul(w2, w1);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f2 d[2][4];
#pragma unroll
            for (int m1 = 0; m1 < 4; ++m1) {
                const float4 q4 = *reinterpret_cast<const float4 *>(lds + 2 * l + 130 * (a + 2 * m1));
                d[0][m1] = (f2){q4.x, q4.y};
                d[1][m1] = (f2){q4.z, q4.w};
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                d[pp][1] = cmul(d[pp][1], w1);
                d[pp][2] = cmul(d[pp][2], w2);
                d[pp][3] = cmul(d[pp][3], w3);
                dft4(d[pp][0], d[pp][1], d[pp][2], d[pp][3]);
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const float4 g0 = ga[4 * a + 2 * pp], g1 = ga[4 * a + 2 * pp + 1];
                y[0] = cmac(y[0], d[pp][0], (f2){g0.x, g0.y});
                y[1] = cmac(y[1], d[pp][1], (f2){g0.z, g0.w});
                y[2] = cmac(y[2], d[pp][2], (f2){g1.x, g1.y});
                y[3] = cmac(y[3], d[pp][3], (f2){g1.z, g1.w});
            }
        }
    }
    // ---- the waves' sums into wave 0: the waves 1 .. 3 leave theirs in their own images ----
    wave_sync();  // the wave's last reads of its image are done
    if (w) {
        *reinterpret_cast<float4 *>(lds + 2 * l) = (float4){y[0].x, y[0].y, y[1].x, y[1].y};
        *reinterpret_cast<float4 *>(lds + 128 + 2 * l) = (float4){y[2].x, y[2].y, y[3].x, y[3].y};
    }
    __syncthreads();
    if (w) return;
    // (what this tail of wave 0 costs while the other waves have left: the kernel without it - results dropped behind the barrier -
    //  measured 0.131 against 0.152 ms at 16 : 1, 0.133 against 0.155 at 32 : 1, 0.183 against 0.203 at 64 : 1, one session: 10 - 14 %)
#pragma unroll
    for (int o = 1; o < NW; ++o) {
        const float4 a4 = *reinterpret_cast<const float4 *>(smem + o * kWgImg + 2 * l);
        const float4 b4 = *reinterpret_cast<const float4 *>(smem + o * kWgImg + 128 + 2 * l);
        y[0] += (f2){a4.x, a4.y};
        y[1] += (f2){a4.z, a4.w};
        y[2] += (f2){b4.x, b4.y};
        y[3] += (f2){b4.z, b4.w};
    }
    // ---- inverse DFT_256 (radix 4 x 4 x 4 x 4, as k_ols_wave<4>: one image layout per exchange, inv256_rd) ----
    const int g = l >> 4, q = l & 15;
    idft4(y[0], y[1], y[2], y[3]);
    wave_sync();
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
    if constexpr (GP) {  // result tau = l + 64 c at b0 + D tau: the phase table's own entry there
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // (base < denom, the step below 2^14: one conditional subtraction where the table is longer than that, a 32-bit
            //  remainder otherwise - a 64-bit remainder per result cost as much as the inverse transform)
            const unsigned add = (unsigned)(D * (l + 64 * c));
            unsigned idx;
            if (denom > 16320u) {
                const unsigned long long r64 = (unsigned long long)base + add;
                idx = (unsigned)(r64 >= denom ? r64 - denom : r64);
            } else {
                idx = (base + add) % denom;
            }
            const float2 p = nco[idx];
            y[c] = cmul(y[c], (f2){p.x, p.y});
        }
    }
    // the valid part by buffer stores: lanes outside it (and behind the end of the output) carry an out-of-range offset
    const int first = V / D;
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

// Even ratios 6 and 10 .. 64 with a combined response of up to 128 D + 1 taps (N = 256 D); RR_OLS_WG=0 keeps k_decim_poly.  Powers of
// two always (16 : 1 with 240 taps: 0.146 -> 0.116 ms per 2^26 samples); the others - which the decimator serves well while the
// response is short - from 12 .. 24 taps per period on (10 : 1 with 145 taps, and the chain's 208 taps at 10 : 1, stay with
// k_decim_poly).
bool ols_wg_supported(uint64_t D, size_t Lc) {
    const char *e = std::getenv("RR_OLS_WG");  // (read per design: tests switch it within one process)
    if (e && std::atoi(e) == 0) return false;
    if (D < 5 || D == 8 || D > 64 || Lc < 1 || Lc - 1 > 128 * D) return false;  // (8 : 1: k_ols_wave2k)
    if (D == 16 || D == 32 || D == 64) return true;
    if (D & 1) {
        // odd ratios (the block staged sample by sample; 1 or 3 of the last wave's four phases empty): from 24 resp. 32 taps per period on
        // (5 : 1 - two waves, three of eight phase slots empty - only for very long responses: the chain's 183 taps at 5 : 1 measured
        //  0.304 ms through this kernel against the decimator's 0.251)
        const size_t per_odd = D == 5 ? 56 : ((D & 3) == 3 ? 24 : 32);
        return (e && std::atoi(e) == 2) || Lc >= per_odd * D;
    }
    // taps per period from which this kernel is ahead of the decimator (scripts/decim_pow2_probe.py, ms per 2^26 samples, decimator /
    // this kernel: 10 : 1 with 15 / 30 taps per period 0.132 / 0.199 against 0.157 / 0.167; 12 : 1 0.132 / 0.217 against 0.137 / 0.144;
    // 20 : 1 0.162 / 0.223 against 0.149 / 0.158; 48 : 1 0.233 / 0.455 against 0.184 / 0.195): a last wave with half of its lanes
    // empty (D = 4 NW - 2) moves the border up
    const size_t per = (D < 20 ? 16 : 12) + ((D & 3) == 2 ? 8 : 0);
    return (e && std::atoi(e) == 2) || Lc >= per * D;  // (=2: every even ratio whatever the length - tests, A/B runs)
}
// the overlap: Lc - 1 rounded up to a multiple of D (whole periods)
int ols_wg_overlap(uint64_t D, size_t Lc) {
    const size_t v = (Lc - 1 + D - 1) / D * D;
    return v == 0 ? (int)D : (int)v;
}
// waves per workgroup: the runs (D + 3) / 4, rounded up to an instantiated count
static int ols_wg_waves(uint64_t D) {
    const int need = (int)((D + 3) / 4);
    for (int nw : {2, 3, 4, 5, 6, 8, 12, 16})
        if (need <= nw) return nw;
    return 0;
}

template <int NW>
static int launch_ols_wg_n(hipStream_t s, const FusedFirArgs &a) {
    const int D = (int)a.D, N = 256 * D;
    if (a.V < D || a.V > N / 2 || a.V % D) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS (%d-sample blocks): overlap %d", N, a.V);
    const int per_block = (N - a.V) / D;
    const size_t nblocks = (a.n_out + per_block - 1) / per_block;
    if (nblocks > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS: too many blocks");
    const int64_t den = (int64_t)a.denom;
    int64_t ph = ((int64_t)a.idx0 + a.e0 - a.V) % den;
    if (ph < 0) ph += den;
    const unsigned hopm = (unsigned)((int64_t)(N - a.V) % den), kstep = (unsigned)((32 * D) % den);
    const unsigned grid = (unsigned)((nblocks + 8 * kWgWin - 1) / (8 * kWgWin) * (8 * kWgWin));
    constexpr size_t lds = (size_t)NW * kWgImg * sizeof(f2);
#define RR_OLSWG_LAUNCH(MF_, GP_)                                                                                                    \
    do {                                                                                                                             \
        RR_TRY(dyn_lds_optin(reinterpret_cast<const void *>(k_ols_wg<NW, MF_, GP_>), lds));                                          \
        if (a.ev_start && a.ev_stop)                                                                                                 \
            hipExtLaunchKernelGGL((k_ols_wg<NW, MF_, GP_>), dim3(grid), dim3(64 * NW), lds, s, a.ev_start, a.ev_stop, 0,             \
                                  (const float2 *)a.xh, (int)a.hx, (const float2 *)a.in, (long)a.n_in, (const float2 *)a.nco,        \
                                  a.denom, a.idx0, (const float2 *)a.H, (const float2 *)a.tw4096, a.V, (float2 *)a.out,              \
                                  (long)a.n_out, (long)a.e0, (float2 *)a.xh_out, (int)a.hx, (unsigned)nblocks, (unsigned)ph, hopm,   \
                                  kstep, 1.0 / (double)den, D);                                                                      \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_ols_wg<NW, MF_, GP_>), dim3(grid), dim3(64 * NW), lds, s, (const float2 *)a.xh, (int)a.hx,         \
                               (const float2 *)a.in, (long)a.n_in, (const float2 *)a.nco, a.denom, a.idx0, (const float2 *)a.H,     \
                               (const float2 *)a.tw4096, a.V, (float2 *)a.out, (long)a.n_out, (long)a.e0, (float2 *)a.xh_out,       \
                               (int)a.hx, (unsigned)nblocks, (unsigned)ph, hopm, kstep, 1.0 / (double)den, D);                      \
    } while (0)
    if (a.genfold) RR_OLSWG_LAUNCH(true, true);
    else if (a.mixfold) RR_OLSWG_LAUNCH(true, false);
    else RR_OLSWG_LAUNCH(false, false);
#undef RR_OLSWG_LAUNCH
    RR_HIP(hipGetLastError());
    return RR_OK;
}

int ols_wg_runs(uint64_t D) { return ols_wg_waves(D); }

int launch_ols_wg(hipStream_t s, const FusedFirArgs &a) {
    if (a.n_out == 0) return RR_OK;
    if (a.D < 5 || a.D > 64) RR_FAIL(RR_ERR_BAD_ARG, "fused OLS: decimation %u has no workgroup kernel", a.D);
    switch (ols_wg_waves(a.D)) {
    case 2: return launch_ols_wg_n<2>(s, a);
    case 3: return launch_ols_wg_n<3>(s, a);
    case 4: return launch_ols_wg_n<4>(s, a);
    case 5: return launch_ols_wg_n<5>(s, a);
    case 6: return launch_ols_wg_n<6>(s, a);
    case 8: return launch_ols_wg_n<8>(s, a);
    case 12: return launch_ols_wg_n<12>(s, a);
    case 16: return launch_ols_wg_n<16>(s, a);
    }
    RR_FAIL(RR_ERR_BAD_ARG, "fused OLS: decimation %u has no workgroup kernel", a.D);
}

}  // namespace rr
