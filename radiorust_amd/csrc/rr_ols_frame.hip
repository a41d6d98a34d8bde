// rr_ols_frame.hip — k_ols_frame: the chain's overlap-save blocks AND its Fourier stage in one kernel (the benchmark's kernel).
//   poly4_block        one 1024-sample block of k_ols_wave<4, POLY> as a function (a wave runs five of them in a row)
//   k_ols_frame<..>    4096-point frames (a workgroup per frame, optional metering epilogue) and 1024-point frames (a wave per frame)
//   launch_ols_frame   host side; ols_frame_supported: the shapes it is compiled for
// The wave-per-block kernels it grew out of are in rr_ols.hip.
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"
#include "rr_meter_dev.hpp"
#include "rr_fft_regs.hpp"
#include "rr_ols_dev.hpp"

#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>

namespace rr {

// ---------------------------------------------------------------------------
// Kernel 3f  k_ols_frame: k_ols_wave's blocks + the Fourier stage in one kernel, one workgroup
// (4 waves) per 4096-sample frame of the decimated stream.  MEASURED SLOWER than the two separate
// kernels (0.217 ms against 0.143 + 0.052 + gaps = 0.207 ms per 2^26 samples) and therefore only
// run on request (RR_FUSED_KERNEL=olsf): the 34 KiB frame buffer next to the wave images leaves room
// for 12 waves per CU with half-size images (0.217 ms; 8 waves with full-size ones: 0.233 ms) instead
// of 16, and a block is a 14 k-cycle dependent chain.  Kept as the documented experiment and as a
// parity case.  The decimated samples never
// touch HBM: the waves drop their blocks' outputs into a 32 KiB frame buffer in LDS, then four
// of them run the windowed DFT_4096 of k_fft4096 on it.  Per input sample that removes the
// 2 B written + 2 B read in between (and one launch); HBM sees 8 B in, 2 B out.
//
// Frame f of a call covers the decimated samples [4096 f - pl, 4096 (f + 1) - pl), pl = samples
// pending from the previous call (the first pl entries of frame 0 come from `pend_in`).  It is
// covered by 20 blocks of 208 outputs (4160 >= 4096; the 64 surplus outputs are the price, 1.6 %),
// wave w takes blocks w, w + 4, ..  Workgroup `nfr` (the frame that does not
// fill) writes its samples to `pend_out` instead of transforming them, and leaves the mixed-sample
// history for the next call.
// ---------------------------------------------------------------------------
int ols_wave_overlap(size_t Lc, size_t granule);
#ifndef RR_V_FRAMEWIN
#define RR_V_FRAMEWIN 1  // frames round robin over the XCDs: 0.1915 ms; a contiguous eighth per XCD (0): 0.1965
#endif
// (waves per frame, first form with full-size images: 4 waves 0.233 ms, 5 waves 0.30 (one workgroup per CU), 8 waves 0.26, 10 waves 0.28)
constexpr int kFrameBlocks = 20;

// One 1024-sample block of k_ols_wave<4, POLY> as a function (the fused frame kernel's waves run five of them in a
// row): v = mixed samples in the pair layout, hv = the lane's 16 entries of the polyphase tables G_p; y[c] =
// result[l + 64 c] of the 256-point inverse.  See k_ols_wave for the passes and the two exchange images.
#ifndef RR_V_FRAME_PAIRS
#define RR_V_FRAME_PAIRS 1  // poly4_block: the phases' values two phases at a time (0: all four at once, which spills - A/B)
#endif
#ifndef RR_V_FRAME_GLDS
#define RR_V_FRAME_GLDS 1  // k_ols_frame: the first half of the tables G_p in LDS (0: all of it from L2, for A/B)
#endif
constexpr int kPolyLds = 1136;  // 2 (63 + 72 * 7) + 2 elements
// SW: the spectrum is taken 128 bins further on (y[0] <-> y[2], y[1] <-> y[3] in front of the inverse): the results' signs
// alternate - k_ols_frame<true>'s fold of a mixer with s = 128 (rr_chain::ensure_mixfold)
// NG: how many of the four 16-byte pieces of the first half of G_p a lane finds in LDS (Glds; the rest comes from L2)
template <bool SW, int NG = 4>
__device__ __forceinline__ void poly4_block(f2 (&v)[16], f2 (&y)[4], f2 *lds, int l, f2 t_p1, f2 t_p2, const f2 (&t_inv)[3],
                                            const float2 *__restrict__ G, const float4 *Glds) {
    const int g = l >> 4, q = l & 15;
    f2 e0[8], e1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        e0[k] = v[2 * k];
        e1[k] = v[2 * k + 1];
    }
    dft8(e0);
    dft8(e1);
    {
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
    wave_sync();  // the previous block's last reads are done
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
    dft8(e0);
    dft8(e1);
    wave_sync();
    {
        f2 *row = lds + 2 * ((l >> 3) + 65 * (l & 7));  // (planes 130 elements apart: k_ols_wave's exchange 2)
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<float4 *>(row + 16 * k) = (float4){e0[k].x, e0[k].y, e1[k].x, e1[k].y};
    }
    wave_sync();
    // the lane's 16 entries of G_p are requested HERE, in two halves (phases 0, 1 / 2, 3), not in front of the transform as
    // k_ols_wave does: kept through the passes they are 32 registers the frame kernel does not have
#if RR_V_FRAME_PAIRS
    // ... and the phases are taken two at a time (a = 0: phases 0, 1 with the half of G_p in LDS; a = 1: phases 2, 3 with the half
    // from L2, requested in front of the first pair's arithmetic): 16 values of the exchange image in registers instead of 32 -
    // with all four phases' values live beside the five blocks' results the kernel spilled two results of its first block, and
    // scratch is HBM traffic (4 KiB written and read back per workgroup: 17 MB of the launch's 151 MB of writes)
    float4 ga[4], gb[4];
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) ga[kp] = (RR_V_FRAME_GLDS && kp < NG) ? Glds[l + 64 * kp] : reinterpret_cast<const float4 *>(G)[l + 64 * kp];
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) gb[kp] = reinterpret_cast<const float4 *>(G)[l + 64 * (4 + kp)];
    const f2 w1 = t_p2, w2 = cmul(w1, w1), w3 = cmul(w2, w1);
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
        // entry i = 4 p + c is half (i & 1) of piece i >> 1: phase 0 = pieces 0, 1; phase 1 = 2, 3; ..
        const float4 *gp = a == 0 ? ga : gb;
        if (a == 0) {
            y[0] = cmul(d[0][0], (f2){gp[0].x, gp[0].y});
            y[1] = cmul(d[0][1], (f2){gp[0].z, gp[0].w});
            y[2] = cmul(d[0][2], (f2){gp[1].x, gp[1].y});
            y[3] = cmul(d[0][3], (f2){gp[1].z, gp[1].w});
        } else {
            y[0] = cmac(y[0], d[0][0], (f2){gp[0].x, gp[0].y});
            y[1] = cmac(y[1], d[0][1], (f2){gp[0].z, gp[0].w});
            y[2] = cmac(y[2], d[0][2], (f2){gp[1].x, gp[1].y});
            y[3] = cmac(y[3], d[0][3], (f2){gp[1].z, gp[1].w});
        }
        y[0] = cmac(y[0], d[1][0], (f2){gp[2].x, gp[2].y});
        y[1] = cmac(y[1], d[1][1], (f2){gp[2].z, gp[2].w});
        y[2] = cmac(y[2], d[1][2], (f2){gp[3].x, gp[3].y});
        y[3] = cmac(y[3], d[1][3], (f2){gp[3].z, gp[3].w});
        if (a == 0) __builtin_amdgcn_sched_barrier(0);  // (the second pair's values are not read in front of the first pair's sums)
    }
#else
    float4 ga[4], gb[4];
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) ga[kp] = (RR_V_FRAME_GLDS && kp < NG) ? Glds[l + 64 * kp] : reinterpret_cast<const float4 *>(G)[l + 64 * kp];
    f2 d[4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m1 = 0; m1 < 4; ++m1) {
            const float4 r = *reinterpret_cast<const float4 *>(lds + 2 * l + 130 * (a + 2 * m1));
            d[2 * a][m1] = (f2){r.x, r.y};
            d[2 * a + 1][m1] = (f2){r.z, r.w};
        }
    {
        const f2 w1 = t_p2, w2 = cmul(w1, w1), w3 = cmul(w2, w1);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            d[pp][1] = cmul(d[pp][1], w1);
            d[pp][2] = cmul(d[pp][2], w2);
            d[pp][3] = cmul(d[pp][3], w3);
            dft4(d[pp][0], d[pp][1], d[pp][2], d[pp][3]);
        }
    }
    // entry i = 4 p + c is half (i & 1) of piece i >> 1: phase 0 = pieces 0, 1; phase 1 = 2, 3; ..
    y[0] = cmul(d[0][0], (f2){ga[0].x, ga[0].y});
    y[1] = cmul(d[0][1], (f2){ga[0].z, ga[0].w});
    y[2] = cmul(d[0][2], (f2){ga[1].x, ga[1].y});
    y[3] = cmul(d[0][3], (f2){ga[1].z, ga[1].w});
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) gb[kp] = reinterpret_cast<const float4 *>(G)[l + 64 * (4 + kp)];
    y[0] = cmac(y[0], d[1][0], (f2){ga[2].x, ga[2].y});
    y[1] = cmac(y[1], d[1][1], (f2){ga[2].z, ga[2].w});
    y[2] = cmac(y[2], d[1][2], (f2){ga[3].x, ga[3].y});
    y[3] = cmac(y[3], d[1][3], (f2){ga[3].z, ga[3].w});
    y[0] = cmac(y[0], d[2][0], (f2){gb[0].x, gb[0].y});
    y[1] = cmac(y[1], d[2][1], (f2){gb[0].z, gb[0].w});
    y[2] = cmac(y[2], d[2][2], (f2){gb[1].x, gb[1].y});
    y[3] = cmac(y[3], d[2][3], (f2){gb[1].z, gb[1].w});
    y[0] = cmac(y[0], d[3][0], (f2){gb[2].x, gb[2].y});
    y[1] = cmac(y[1], d[3][1], (f2){gb[2].z, gb[2].w});
    y[2] = cmac(y[2], d[3][2], (f2){gb[3].x, gb[3].y});
    y[3] = cmac(y[3], d[3][3], (f2){gb[3].z, gb[3].w});
#endif
    if (SW) {
        const f2 t0 = y[0], t1 = y[1];
        y[0] = y[2];
        y[1] = y[3];
        y[2] = t0;
        y[3] = t1;
    }
    // inverse DFT_256 (radix 4 x 4 x 4 x 4, as k_ols_wave<4>: one image layout per exchange, inv256_rd)
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
        if (pass == 3) break;
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
}

struct FrameArgs {
    const float2 *xh;       // mixed-sample history (hx samples, ends right before in[0])
    int hx;
    const float2 *in;
    long n_in;
    const float2 *nco;      // denom + 1 entries
    unsigned denom, idx0;
    const float2 *H;        // pair-interleaved, as for k_ols_wave
    const float2 *tw;       // 1024 twiddles + packed lane seeds
    int V;                  // 192 (per_block = 208 is compiled in through kFrameBlocks)
    long e0;                // input position of decimated sample 0 of this call
    long n_dec;             // decimated samples this call produces
    const float2 *pend_in;  // pl samples pending from the previous call
    int pl;
    float2 *pend_out;       // receives the (pl + n_dec) mod 4096 samples left over
    float2 *spectra;        // (pl + n_dec) / 4096 frames of 4096 bins
    const float *window;    // Fourier window, 4096
    const float2 *tw4096;   // e^{-j 2 pi k / 4096}
    int center_dc;
    float2 *xh_out;
    unsigned ph0;           // (idx0 + e0 - V - 4 pl) mod denom
    unsigned kstep;         // 128 mod denom
    double inv_denom;
    unsigned nfr;           // full frames
    int mixfold;            // the NCO's period divides 8 and H holds the tables with the mixer folded in (rr_chain::ensure_mixfold)
    float sigma;            // results at odd indices of a block times sigma (+-1)
    int nb;                 // blocks per frame: ceil(4096 / ((1024 - V) / 4)) = 18 / 19 / 20 for V = 64 / 128 / 192 (<= kFrameBlocks)
    FrameMeter fm;          // METER instances: metering::bandwidth per spectrum, computed behind the transform
};

// The fused frame kernel, second form (round 2).  What made the first one slower than the two kernels it replaces was
// its LDS: the 34 KiB frame buffer NEXT TO the wave images left room for 12 waves per CU (with half-size images).  Here
// the two share the same LDS in time: the four waves keep the results of their five blocks in registers (20 values per
// lane), and only when all of them are done do they drop them into what were their exchange images - now the frame, and then
// the exchange image of the DFT_4096.  37 KiB per workgroup = 4 workgroups = 16 waves per CU, as k_ols_wave has; the
// polyphase block transform leaves the registers for it (92 + 32 kept while the fifth block runs).
#ifndef RR_V_FRAME_LD_NT
#define RR_V_FRAME_LD_NT 1
#endif
// The last V samples of a block are the first V of the next one.  Loaded with the streaming hint they are gone from L2 when the
// next block asks for them (PMC round 2: 152.7 KB fetched per frame of 133 KB, i.e. 73 % of the overlap came from HBM twice);
// the pieces k' >= RR_V_FRAME_LD_TAILK of a block (V = 192: part of piece 6 and piece 7) are therefore loaded WITHOUT the hint.
#ifndef RR_V_FRAME_LD_TAILK
#define RR_V_FRAME_LD_TAILK 6
#endif
#ifndef RR_V_FRAME_CONSEC
#define RR_V_FRAME_CONSEC 0  // 1: a wave takes five NEIGHBOURING blocks (jb = 5 w + kb) instead of every fourth (A/B runs)
#endif
// MF: the mixer folded into the tables (rr_chain::ensure_mixfold) - instances of their own without the mixer's code; SW: poly4_block<SW>
// FULL: 20 blocks per frame (V = 192, cfg2) - no guard around a wave's blocks; !FULL: 18 / 19 blocks (V = 64 / 128)
// GP (with MF): ANY NCO period - the mixer moved behind the filter.  The phase table is a geometric sequence p[t] = p0 w^t, so
//   sum_i c[i] x[t - i] p[t - i] = p[t] sum_i (c[i] w^-i) x[t - i]:
// the blocks transform the samples as they are with the tables of the response c[i] w^-i (host, rr_chain::ensure_genfold), and the
// 4 results a lane keeps per block are multiplied by p at their own positions (one table read per lane and block + the three
// rotations by 256 samples kept behind the table): 14 packed instructions per block instead of the 60 of the mixer in front.
// LF = 1024: 1024-point spectra - ONE WAVE per frame: its five blocks one after the other (five blocks of 208 .. 240 results
// cover the 1024), the results dropped into the wave's own image, then the wave-level 1024-point transform of k_fft1024 on it;
// no workgroup barrier at all, the four waves of a workgroup are four neighbouring frames.  (No room for half of G_p in LDS beside
// four images of the 1024-point transform's size: all of it from L2.)
template <bool MF, bool SW, bool FULL = true, bool METER = false, bool GP = false, int LF = 4096>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ols_frame(FrameArgs a_) {
    static_assert(LF == 4096 || (LF == 1024 && FULL && !METER), "4096-point frames, or 1024-point frames without the metering epilogue");
    constexpr bool WF = LF == 1024;                       // a wave per frame
    constexpr int kImg = WF ? kWaveLds : kPolyLds;        // a wave's image
    constexpr int NL = WF ? 64 : 256;                     // lanes that work on one frame
    const FrameArgs &a = a_;
    const FrameArgs *ka = (const FrameArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    __shared__ __attribute__((aligned(16))) f2 smem[4 * kImg];  // LF = 4096: 4544 elements >= the 4352 of the padded frame image
#if RR_V_FRAME_GLDS
    // the first half of the response tables G_p (phases 0 and 1: 4 KiB) in LDS, in the 4.5 KiB per workgroup that four workgroups per
    // CU leave: every block otherwise pulls all 8 KiB from L2 - as many bytes as its samples, in 8 of its 16 vector-memory instructions
    __shared__ __attribute__((aligned(16))) float4 gl[WF ? 192 : 256];  // (LF = 1024: three of the four pieces fit beside the four images)
#endif
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    f2 *const fr = WF ? smem + w * kImg : smem;
    const int fl_ = WF ? l : tid;                         // the lane's index among the frame's lanes
    // The frame that does not fill (its samples go to pend_out) is workgroup 0: dispatched first it runs beside the first
    // round of frames; as the LAST workgroup it ran alone behind the four full rounds of a 2^26-sample call.
    // The others: frames dealt to the XCDs in a moving window, RR_V_FRAMEWIN neighbouring frames per XCD.
    // (LF = 1024: wave 0 of workgroup 0 is that frame, wave g takes frame g - 1.)
    unsigned f;
    if constexpr (WF) {
        const unsigned g = 4u * blockIdx.x + (unsigned)w;
        f = g == 0 ? a.nfr : g - 1;  // (waves without a frame leave behind the workgroup's one barrier, below)
    } else if (blockIdx.x == 0) {
        f = a.nfr;
    } else {
        const unsigned bx = blockIdx.x - 1;
        f = bx / (8 * RR_V_FRAMEWIN) * (8 * RR_V_FRAMEWIN) + (bx % (8 * RR_V_FRAMEWIN) & 7) * RR_V_FRAMEWIN + (bx % (8 * RR_V_FRAMEWIN) >> 3);
        if (f >= a.nfr) return;
    }
    const bool tail = f == a.nfr;  // the frame that does not fill: goes to pend_out
    const long F0 = (long)LF * f - a.pl;  // decimated index (of this call) of the frame's first sample
    const int hop = 1024 - a.V, per_block = hop >> 2, first = a.V >> 2;

    if (tail && ka->xh_out) {  // mixed-sample history for the next call
        const int hxe = ka->hx;
        for (int i = fl_; i < hxe; i += NL) {
            const long pos = a.n_in - hxe + i;
            float2 v;
            if (pos >= 0) {
                const float2 xx = a.in[pos];
                const float2 pp = a.nco[(unsigned)(((long)ka->idx0 + pos) % (long)a.denom)];
                v.x = xx.x * pp.x - xx.y * pp.y;
                v.y = xx.x * pp.y + xx.y * pp.x;
            } else {
                v = (pos >= -(long)hxe) ? ka->xh[hxe + pos] : float2{0.f, 0.f};
            }
            ka->xh_out[i] = v;
        }
    }

    // lane constants of the polyphase block transform: tw[4 (l >> 1)], tw[32 (l >> 3)], the inverse's three
    f2 t_p1, t_p2, t_inv[3];
    {
        const float4 *tl = reinterpret_cast<const float4 *>(a.tw + 1024) + l;
        const float4 s6 = tl[384], s7 = tl[448], s8 = tl[512];
        t_p1 = (f2){s6.x, s6.y};
        t_p2 = (f2){s6.z, s6.w};
        t_inv[0] = (f2){s7.x, s7.y};
        t_inv[1] = (f2){s7.z, s7.w};
        t_inv[2] = (f2){s8.x, s8.y};
    }
    f2 *const lds = smem + w * kImg;
    const float4 *glp = nullptr;
#if RR_V_FRAME_GLDS
    if (!WF || tid < 192) gl[tid] = reinterpret_cast<const float4 *>(a.H)[tid];
    __syncthreads();
    glp = gl;
#endif
    if constexpr (WF) {
        if (f > a.nfr) return;
    }
    const long n_clamp = a.n_in - 1024;  // the launcher guarantees n_in >= 1024
    // mixer folded into the tables (MF): the blocks transform the samples as they are; the phasor of a block's first sample (the same
    // for every block of the call) is in the table the host picked for this call, the alternating sign in poly4_block<SW>
    f2 keep[kFrameBlocks / 4][4];
    // phase of a block's first sample: (idx0 + b0) mod denom, b0 = const + 4 (4096 f + per_block jb)
    auto block_phase = [&](int jb) -> unsigned {
        const double dn = (double)a.denom;
        const double prod = (double)a.ph0 + (4.0 * LF) * (double)f + (double)(4 * per_block * jb);
        const double qd = __builtin_floor(prod * a.inv_denom);
        double rd = __builtin_fma(-qd, dn, prod);
        if (rd < 0.0) rd += dn;
        if (rd >= dn) rd -= dn;
        return (unsigned)rd;
    };
    // GP: the phasors of the wave's five blocks read up front (scalar reads: no SMEM result is outstanding inside the blocks, whose
    // LDS exchanges count on lgkmcnt) + the lane's constant e^{j 2 pi 4 l numer / denom} and the three rotations by 256 results
    [[maybe_unused]] float2 rot128[7];  // (mixer in front, general period: the seven steps by 128 samples - scalar reads, up front)
    if constexpr (!MF) {
#pragma unroll
        for (int k = 1; k < 8; ++k) rot128[k - 1] = ld_uniform(a.nco + (a.denom + 1 + k));
    }
    [[maybe_unused]] f2 glane = {1.f, 0.f};
    [[maybe_unused]] unsigned rdv[kFrameBlocks / 4];
    [[maybe_unused]] float2 pgv[kFrameBlocks / 4], rtv[3];
    if constexpr (GP) {
        const float2 gl_ = a.nco[a.denom + 9 + l];
        glane = (f2){gl_.x, gl_.y};
#pragma unroll
        for (int kb = 0; kb < kFrameBlocks / 4; ++kb) {
            rdv[kb] = __builtin_amdgcn_readfirstlane(block_phase(WF ? kb : RR_V_FRAME_CONSEC ? 5 * w + kb : w + 4 * kb));
            pgv[kb] = ld_uniform(a.nco + rdv[kb]);
        }
#pragma unroll
        for (int c = 1; c < 4; ++c) rtv[c - 1] = ld_uniform(a.nco + (a.denom + 1 + 2 * c));  // e^{j 2 pi (256 c numer mod denom) / denom}
    }
#pragma unroll
    for (int kb = 0; kb < kFrameBlocks / 4; ++kb) {
        const int jb = WF ? kb : RR_V_FRAME_CONSEC ? 5 * w + kb : w + 4 * kb;  // (five neighbouring blocks per wave instead: measured 0.171 against 0.159 ms)
        // (shorter responses: V = 64 / 128, 240 / 224 results per block - 18 / 19 blocks cover the frame, the last round's other waves idle)
        if (!FULL && kb >= 3 && __builtin_amdgcn_readfirstlane(jb) >= a.nb) continue;  // (a wave-uniform branch; rounds 0 .. 2 are always full)
        const long b0 = a.e0 - a.V + 4 * (F0 + (long)per_block * jb);
        // phase of the lane's first sample: (idx0 + b0 + 2 l) mod denom
        unsigned r = 0;
        if (!MF || !(b0 >= 0 && b0 <= n_clamp)) {  // (MF: only the edge blocks look at the table)
            unsigned rd0;
            if constexpr (GP) rd0 = rdv[kb];
            else rd0 = block_phase(jb);
            r = rd0 + 2u * (unsigned)l;
            if (a.denom >= 128u) {
                if (r >= a.denom) r -= a.denom;
            } else if ((a.denom & (a.denom - 1u)) == 0u) {
                r &= a.denom - 1u;
            } else {
                r %= a.denom;
            }
        }
        f2 v[16];
        if (b0 >= 0 && b0 <= n_clamp) {
            f4u x[8];
            {
                const f4u *src = reinterpret_cast<const f4u *>(a.in + b0) + l;
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = (RR_V_FRAME_LD_NT && k < RR_V_FRAME_LD_TAILK) ? ld_stream(src + 64 * k) : *(src + 64 * k);
            }
            if (MF) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[2 * k] = (f2){x[k].x, x[k].y};
                    v[2 * k + 1] = (f2){x[k].z, x[k].w};
                }
            } else if (a.kstep == 0) {
                const f4u pp = *reinterpret_cast<const f4u *>(a.nco + r);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[2 * k] = cmul((f2){x[k].x, x[k].y}, (f2){pp.x, pp.y});
                    v[2 * k + 1] = cmul((f2){x[k].z, x[k].w}, (f2){pp.z, pp.w});
                }
            } else {
                // general period: the lane's pair at the block start from the table, the seven 128-sample steps by the
                // rotations kept behind the table (as k_ols_wave; a table walk - 8 pair reads per block - cost 0.19 / 0.25 ms
                // per step for the 40 000- and the 10^8-entry tables)
                const f4u pp = *reinterpret_cast<const f4u *>(a.nco + r);
                const f2 p0 = {pp.x, pp.y}, p1 = {pp.z, pp.w};
                v[0] = cmul((f2){x[0].x, x[0].y}, p0);
                v[1] = cmul((f2){x[0].z, x[0].w}, p1);
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    const f2 rot = {rot128[k - 1].x, rot128[k - 1].y};
                    v[2 * k] = cmul((f2){x[k].x, x[k].y}, cmul(p0, rot));
                    v[2 * k + 1] = cmul((f2){x[k].z, x[k].w}, cmul(p1, rot));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f4u pp = *reinterpret_cast<const f4u *>(a.nco + r);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const long pos = b0 + 2 * l + j + 128 * k;
                    const bool inr = pos >= 0 && pos < a.n_in;
                    const int hxe = ka->hx;
                    const bool hst = pos < 0 && pos >= -(long)hxe;
                    const float2 *ptr = inr ? a.in + pos : ka->xh + (hst ? hxe + pos : 0);
                    const float2 xx = *ptr;
                    const f2 p = j ? (f2){pp.z, pp.w} : (f2){pp.x, pp.y};
                    // (mixer folded in: the block wants the samples UNMIXED - the history, which holds mixed ones, times conj(p))
                    const f2 pk = MF ? (f2){inr ? 1.f : (hst ? p.x : 0.f), hst ? -p.y : 0.f}
                                            : (f2){inr ? p.x : (hst ? 1.f : 0.f), inr ? p.y : 0.f};
                    const f2 xv = {(inr || hst) ? xx.x : 0.f, (inr || hst) ? xx.y : 0.f};
                    v[2 * k + j] = cmul(xv, pk);
                }
                r += a.kstep;
                if (r >= a.denom) r -= a.denom;
            }
        }
        poly4_block<SW, WF ? 3 : 4>(v, keep[kb], lds, l, t_p1, t_p2, t_inv, a.H, glp);
        if constexpr (GP) {
            // result tau = l + 64 c of the block sits at input position b0 + 4 tau
            const f2 gph = cmul((f2){pgv[kb].x, pgv[kb].y}, glane);
            keep[kb][0] = cmul(keep[kb][0], gph);
#pragma unroll
            for (int c = 1; c < 4; ++c) keep[kb][c] = cmul(keep[kb][c], cmul(gph, (f2){rtv[c - 1].x, rtv[c - 1].y}));
        }
        // (one block at a time: without this the five unrolled blocks' loads are all hoisted to the front)
        asm volatile("" : "+v"(keep[kb][0]), "+v"(keep[kb][1]), "+v"(keep[kb][2]), "+v"(keep[kb][3]));
    }
    if constexpr (WF) wave_sync();  // the wave is done with its exchange image: it becomes the frame
    else __syncthreads();            // every wave is done with its exchange images: they become the frame
    // the part of frame 0 that was pending
    if (f == 0)
        for (int i = fl_; i < a.pl; i += NL) {
            const float2 p = ka->pend_in[i];
            fr[i] = (f2){p.x, p.y};
        }
    // valid results tau = l + 64 c >= first; frame-relative index i = per_block * jb + tau - first
#pragma unroll
    for (int kb = 0; kb < kFrameBlocks / 4; ++kb) {
        const int jb = WF ? kb : RR_V_FRAME_CONSEC ? 5 * w + kb : w + 4 * kb;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int tau = l + 64 * c;
            const int i = per_block * jb + tau - first;
            const long m = F0 + i;
            // (LF = 1024: the five blocks' 1040 .. 1200 results end behind the frame - the next frame's wave computes those too)
            if (tau >= first && m >= 0 && m < a.n_dec && (FULL || jb < a.nb) && (!WF || i < LF)) fr[i] = keep[kb][c];
        }
    }
    if constexpr (WF) wave_sync();
    else __syncthreads();
    if (tail) {
        const long have = a.pl + a.n_dec - (long)LF * a.nfr;  // samples of the unfinished frame
        for (int i = fl_; i < have; i += NL) {
            float2 o;
            o.x = fr[i].x;
            o.y = fr[i].y;
            ka->pend_out[i] = o;
        }
        return;
    }
    if constexpr (WF) {
        // ---- Fourier: window, DFT_1024 by this wave (k_fft1024's network: sample pairs x[2 l + 128 k], + 1), optional DC centring ----
        f2 v[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 x = *reinterpret_cast<const float4 *>(fr + 2 * l + 128 * k);
            const float2 wn = *reinterpret_cast<const float2 *>(ka->window + 2 * l + 128 * k);
            v[2 * k] = (f2){x.x * wn.x, x.y * wn.x};
            v[2 * k + 1] = (f2){x.z * wn.y, x.w * wn.y};
        }
        f2 f_p1, f_p2[2];
        {
            const float4 *tl = reinterpret_cast<const float4 *>(ka->tw4096 + 1024) + l;  // (the Fourier block's 1024-point table + lane seeds)
            const float4 s0 = tl[0], s1 = tl[64];
            f_p1 = (f2){s0.x, s0.y};
            f_p2[0] = (f2){s0.z, s0.w};
            f_p2[1] = (f2){s1.x, s1.y};
        }
        wave_sync();  // the frame has been read: it becomes the transform's exchange image
        f2 X[16];
        wave_dft1024(v, X, fr, l, f_p1, f_p2, [] {});
        f2 *dst = reinterpret_cast<f2 *>(ka->spectra) + (size_t)f * 1024;
        const int rot = ka->center_dc ? 512 : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(X[k], dst + ((l + 64 * k + rot) & 1023));
        return;
    }
    // ---- Fourier: window, DFT_4096 (radix 16 x 3 as k_fft4096), optional DC centring ---------------
    f2 v[16];
    {
        // the lane's 16 window values as 4 loads of 16 bytes (the packed copy behind the table, as k_fft4096)
        const float4 *wp = reinterpret_cast<const float4 *>(ka->window + 4096) + 4 * tid;
        float wv[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 w4 = wp[q4];
            wv[4 * q4] = w4.x;
            wv[4 * q4 + 1] = w4.y;
            wv[4 * q4 + 2] = w4.z;
            wv[4 * q4 + 3] = w4.w;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fr[tid + 256 * k] * wv[k];
    }
    __syncthreads();  // the frame has been read: it becomes the padded exchange image
    fft4096_regs(v, fr, ka->tw4096, tid);
    const int rot = ka->center_dc ? 2048 : 0;
    if (!METER || ka->fm.store) {
        float2 *dst = ka->spectra + (size_t)f * 4096;
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<f2 *>(dst) + ((tid + 256 * k + rot) & 4095));
    }
    if constexpr (METER) {
        double total;
        const int tw_ = (int)(f & 3u);
        const double bw = frame4096_bandwidth(v, fr, tid, rot, ka->fm.double_percentile, ka->fm.sample_rate, &total, tw_);
        if (tid == 64 * tw_) {
            ka->fm.bw[f] = bw;
            if (ka->fm.energy) ka->fm.energy[f] = total;
        }
    }
}

bool ols_frame_supported(uint64_t D, size_t Lc, size_t fft_len) {
    // overlaps of 64 / 128 / 192 samples = 240 / 224 / 208 results per block, 18 / 19 / 20 blocks per frame (at most kFrameBlocks)
    // (1024-point spectra: a wave per frame, five blocks each - RR_FRAME_1K=0 keeps the two kernels)
    static const bool no1k = [] { const char *e = std::getenv("RR_FRAME_1K"); return e && std::atoi(e) == 0; }();
    return D == 4 && (fft_len == 4096 || (fft_len == 1024 && !no1k)) && Lc >= 1 && ols_wave_overlap(Lc, 64) <= 192;
}

int launch_ols_frame(hipStream_t s, const FusedFirArgs &a, const void *pend_in, size_t pl, void *pend_out, void *spectra,
                     const void *window, const void *tw4096, bool center_dc, const FrameMeter *fm, size_t fft_len) {
    if (a.V != 64 && a.V != 128 && a.V != 192) RR_FAIL(RR_ERR_BAD_ARG, "fused frame kernel: overlap %d not supported", a.V);
    if (fft_len != 4096 && fft_len != 1024) RR_FAIL(RR_ERR_BAD_ARG, "fused frame kernel: %zu-point spectra not instantiated", fft_len);
    if (fft_len == 1024 && fm) RR_FAIL(RR_ERR_BAD_ARG, "fused frame kernel: the metering epilogue rides on 4096-point spectra only");
    const size_t total = pl + a.n_out, nfr = total / fft_len;
    if (total == 0) return RR_OK;
    if (a.n_in < 1024) RR_FAIL(RR_ERR_BAD_ARG, "fused frame kernel: needs at least 1024 input samples per call");
    if (nfr > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "fused frame kernel: too many frames");
    FrameArgs f;
    f.xh = (const float2 *)a.xh;
    f.hx = (int)a.hx;
    f.in = (const float2 *)a.in;
    f.n_in = (long)a.n_in;
    f.nco = (const float2 *)a.nco;
    f.denom = a.denom;
    f.idx0 = a.idx0;
    f.H = (const float2 *)a.H;
    f.tw = (const float2 *)a.tw4096;  // the 1024-entry table + lane seeds (FusedFirArgs field name)
    f.V = a.V;
    f.e0 = (long)a.e0;
    f.n_dec = (long)a.n_out;
    f.pend_in = (const float2 *)pend_in;
    f.pl = (int)pl;
    f.pend_out = (float2 *)pend_out;
    f.spectra = (float2 *)spectra;
    f.window = (const float *)window;
    f.tw4096 = (const float2 *)tw4096;
    f.center_dc = center_dc ? 1 : 0;
    f.xh_out = (float2 *)a.xh_out;
    const int64_t den = (int64_t)a.denom;
    int64_t ph = ((int64_t)a.idx0 + a.e0 - a.V - 4 * (int64_t)pl) % den;
    if (ph < 0) ph += den;
    f.ph0 = (unsigned)ph;
    f.kstep = (unsigned)(128 % den);
    f.inv_denom = 1.0 / (double)den;
    f.nfr = (unsigned)nfr;
    f.mixfold = a.mixfold ? 1 : 0;
    f.sigma = a.sigma;
    {
        const int per_block = (1024 - a.V) / 4;
        f.nb = (int)((fft_len + per_block - 1) / per_block);
        if (f.nb > kFrameBlocks || (fft_len == 1024 && f.nb > kFrameBlocks / 4))
            RR_FAIL(RR_ERR_BAD_ARG, "fused frame kernel: %d blocks per frame", f.nb);
    }
    f.fm = fm ? *fm : FrameMeter{};
    if (fft_len == 1024) {
        // a wave per frame, wave 0 of the grid the frame that does not fill: nfr + 1 waves in workgroups of four
        const unsigned grid1k = (unsigned)((nfr + 1 + 3) / 4);
#define RR_FRAME1K_LAUNCH(MF_, SW_, GP_)                                                                                                     \
    do {                                                                                                                                      \
        if (a.ev_start && a.ev_stop)                                                                                                          \
            hipExtLaunchKernelGGL((k_ols_frame<MF_, SW_, true, false, GP_, 1024>), dim3(grid1k), dim3(256), 0, s, a.ev_start, a.ev_stop, 0, f); \
        else                                                                                                                                  \
            hipLaunchKernelGGL((k_ols_frame<MF_, SW_, true, false, GP_, 1024>), dim3(grid1k), dim3(256), 0, s, f);                           \
    } while (0)
        if (a.genfold) RR_FRAME1K_LAUNCH(true, false, true);
        else if (a.mixfold && a.sigma < 0.f) RR_FRAME1K_LAUNCH(true, true, false);
        else if (a.mixfold) RR_FRAME1K_LAUNCH(true, false, false);
        else RR_FRAME1K_LAUNCH(false, false, false);
#undef RR_FRAME1K_LAUNCH
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    const unsigned grid = 1u + (unsigned)((nfr + 8 * RR_V_FRAMEWIN - 1) / (8 * RR_V_FRAMEWIN) * (8 * RR_V_FRAMEWIN));
#define RR_FRAME_LAUNCH(MF_, SW_, FU_, GP_)                                                                                            \
    do {                                                                                                                                \
        if (fm)                                                                                                                         \
            hipLaunchKernelGGL((k_ols_frame<MF_, SW_, FU_, true, GP_>), dim3(grid), dim3(256), 0, s, f);                                \
        else if (a.ev_start && a.ev_stop)                                                                                               \
            hipExtLaunchKernelGGL((k_ols_frame<MF_, SW_, FU_, false, GP_>), dim3(grid), dim3(256), 0, s, a.ev_start, a.ev_stop, 0, f); \
        else                                                                                                                            \
            hipLaunchKernelGGL((k_ols_frame<MF_, SW_, FU_, false, GP_>), dim3(grid), dim3(256), 0, s, f);                               \
    } while (0)
    if (f.nb == kFrameBlocks) {
        if (a.genfold) RR_FRAME_LAUNCH(true, false, true, true);
        else if (a.mixfold && a.sigma < 0.f) RR_FRAME_LAUNCH(true, true, true, false);
        else if (a.mixfold) RR_FRAME_LAUNCH(true, false, true, false);
        else RR_FRAME_LAUNCH(false, false, true, false);
    } else {
        if (a.genfold) RR_FRAME_LAUNCH(true, false, false, true);
        else if (a.mixfold && a.sigma < 0.f) RR_FRAME_LAUNCH(true, true, false, false);
        else if (a.mixfold) RR_FRAME_LAUNCH(true, false, false, false);
        else RR_FRAME_LAUNCH(false, false, false, false);
    }
#undef RR_FRAME_LAUNCH
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
