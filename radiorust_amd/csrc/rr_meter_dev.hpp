// rr_meter_dev.hpp — metering::bandwidth (src/metering.rs:41-80) as a workgroup-wide parallel scan, for use as the
// EPILOGUE of a kernel that has just produced a frame of bins (k_fft4096 / k_stft4096 / k_ols_frame) and by the stand-alone
// kernel k_bandwidth_par (rr_metering.hip).  Device code only.
//
// The reference walks the bins in the order idcs = (wrap .. n).chain(0 .. wrap), wrap = (n + 1) / 2, adding energies in f64
// until the running sum exceeds limit = total * double_percentile / 2, once from the front and once from the back:
//   used = (whole bins before the crossing) + (limit - old) / (new - old).
// Here: a workgroup of 256 lanes per frame.  Lane j owns the scan positions [j C, (j + 1) C) of each direction and sums them
// in f64 (sequentially, in scan order); the partial sums are grouped once more, and ONE lane per direction walks groups,
// lanes and finally the bins of the lane where the crossing lies, exactly as the reference's loop does from there on.  The
// sums therefore associate differently from the reference's single sequential loop (last-bit differences of f64 sums: 1e-16
// relative; the reference's KATs hold to 1e-10); the serial kernel k_meter stays as the bit-exact checker
// (RR_METER_SERIAL=1 routes the metered entry points through it).
#pragma once
#include <hip/hip_runtime.h>

#include "rr_kernels.hpp"

namespace rr {

// Workgroup barrier that waits for the wave's LDS operations only.  __syncthreads() also drains the vector-memory counter: behind
// a frame's 16 stores it made every wave wait until its spectrum had left for HBM - the Stft kernel, bound by exactly those
// stores, took 1.17 x with the epilogue in place of 1.04 x without the stores.
__device__ __forceinline__ void meter_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// scratch: kBwScratch doubles of LDS that nobody else touches between the first barrier in here and the caller's next one.
// e_at(s): energy (as f64) of scan position s in FORWARD order, 0 <= s < n; any lane may call it for any s.
// All 256 lanes of the workgroup must call this (it contains workgroup barriers).  The result - the bandwidth, and in
// *total_out the sum of all energies - is valid in ONE LANE ONLY (lane 64 * tail_wave; return value 0 elsewhere): the other
// three waves leave after the second barrier and go on with the caller's next piece of work while that wave finishes.
//
// Three levels, so that almost nothing runs on more than a few lanes (a full prefix scan of f64 values over 256 lanes -
// twelve 64-bit lane shuffles per direction - cost as many instructions as half the 4096-point transform in front of it):
//   A  every lane: the sum of its own C energies                                                   -> P[256]
//      (the reverse walk uses the same sums from the other end: a sum of 16 values does not depend on the direction beyond
//       its last bits)
//   B  16 lanes of wave 0: the sums of 16 neighbouring P                                           -> Q[16]
//   C  lane 0 (forward) and lane 32 (reverse) of wave 0 walk Q, then the 16 P of the group where the running sum crosses
//      the limit, then the C energies of that lane's range - the reference's loop, with whole groups added at once in front
//      of the crossing.  Every level's 16 values are fetched together and walked in registers (fetched one by one inside the
//      loop the walk was 48 dependent LDS round trips: 3 us per spectrum).
constexpr int kBwScratch = 256 + 16 + 4;
// tail_wave (0 .. 3): the wave that finishes (B, C) - callers rotate it from spectrum to spectrum: a workgroup's wave i sits on
// SIMD i, so with always the same wave one SIMD of the CU carried every workgroup's serial tail (+ 16 % on the Stft kernel,
// + 4 % rotated).  The result is valid in lane 64 * tail_wave.
template <int C, class EnergyAt>
__device__ __forceinline__ double bandwidth_block256(int n, int j_, EnergyAt &&e_at, double double_percentile, double sample_rate,
                                                     double *scratch, double *total_out, int tail_wave = 0) {
    int j = j_;
    const int c = C > 0 ? C : (n + 255) / 256;
    double *P = scratch, *Q = scratch + 256, *R = scratch + 272;
    {
        const int lo = j * c < n ? j * c : n, hi = (j + 1) * c < n ? (j + 1) * c : n;
        double pf = 0.0;
        if (C > 0) {
#pragma unroll
            for (int i = 0; i < (C > 0 ? C : 1); ++i)
                if (lo + i < hi) pf += e_at(lo + i);
        } else {
            for (int s = lo; s < hi; ++s) pf += e_at(s);
        }
        P[j] = pf;
    }
    meter_lds_barrier();
    if ((j >> 6) != tail_wave) return 0.0;  // (wave-uniform: the other three waves are done)
    j &= 63;
    if (j < 16) {
        double x[16], q = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = P[16 * j + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) q += x[i];
        Q[j] = q;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double used = 0.0, total = 0.0;
    if (j == 0 || j == 32) {
        const bool rev = j == 32;
        // (every level's values are fetched in WALK order - the addresses follow the direction, the register indices do not)
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = Q[i];
        // the reference's total is ONE sum, used for both directions: both walking lanes add the same 16 values in the same order
#pragma unroll
        for (int i = 0; i < 16; ++i) total += x[i];
        if (rev) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = Q[15 - i];
        }
        const double limit = total * double_percentile / 2.0;
        used = (double)n;  // no crossing (total <= limit): every bin is "used" (the reference's loop runs to its end)
        double cum = 0.0;
        int g = 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double nw = cum + x[i];
            if (g == 16) {
                if (nw > limit) g = i;
                else cum = nw;
            }
        }
        if (g < 16) {
            const int gb = rev ? 15 - g : g;  // the group in forward numbering
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = P[16 * gb + (rev ? 15 - i : i)];
            int b = 15;  // (the last of the group takes the crossing if rounding hides it from the others)
            {
                bool found = false;
#pragma unroll
                for (int i = 0; i < 15; ++i) {
                    const double nw = cum + x[i];
                    if (!found) {
                        if (nw > limit) {
                            found = true;
                            b = i;
                        } else {
                            cum = nw;
                        }
                    }
                }
            }
            const int bf = 16 * gb + (rev ? 15 - b : b);  // the lane (forward numbering) whose range holds the crossing
            const int lo = bf * c < n ? bf * c : n, hi = (bf + 1) * c < n ? (bf + 1) * c : n;
            // walk positions: forward lo .. hi - 1 (walk index = s), reverse hi - 1 .. lo (walk index = n - 1 - s)
            const int w0 = rev ? n - hi : lo;  // walk index of the range's first bin in this direction
            used = (double)(w0 + (hi - lo));   // (crossing lost to rounding at the range's end: all of the range's bins)
            if (C > 0) {
                double ev[C > 0 ? C : 1];
#pragma unroll
                for (int i = 0; i < (C > 0 ? C : 1); ++i) ev[i] = (lo + i < hi) ? e_at(rev ? hi - 1 - i : lo + i) : 0.0;
                bool found = false;
#pragma unroll
                for (int i = 0; i < (C > 0 ? C : 1); ++i) {
                    const double nw = cum + ev[i];
                    if (!found && lo + i < hi) {
                        if (nw > limit) {
                            found = true;
                            used = (double)(w0 + i) + (limit - cum) / (nw - cum);
                        } else {
                            cum = nw;
                        }
                    }
                }
            } else {
                for (int i = 0; lo + i < hi; ++i) {
                    const double nw = cum + e_at(rev ? hi - 1 - i : lo + i);
                    if (nw > limit) {
                        used = (double)(w0 + i) + (limit - cum) / (nw - cum);
                        break;
                    }
                    cum = nw;
                }
            }
        }
        if (rev) R[0] = used;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (j != 0) return 0.0;
    double used_bins = 0.0;
    used_bins += used;
    used_bins += R[0];
    if (total_out) *total_out = total;
    const double bw = ((double)n - used_bins) * sample_rate / (double)n;
    return bw > 0.0 ? bw : 0.0;
}

// The epilogue behind a 4096-point transform of a 256-lane workgroup: lane j holds v[k] = X[j + 256 k], the frame's OUTPUT
// index of that bin is o = (j + 256 k + rot) & 4095 (rot = 2048 with center_dc).  `lds` is the transform's exchange image
// (>= 4096 + 64 complex f32 elements), free once every lane has left the transform: the energies (norm_sqr in f32 as the
// reference computes it, num-complex: re * re + im * im, no contraction) go there in scan order, 16 KiB, the scratch behind them.
typedef float rr_f2m __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double frame4096_bandwidth(const rr_f2m (&v)[16], void *lds, int j, int rot, double double_percentile,
                                                      double sample_rate, double *total_out, int tail_wave = 0) {
    float *e = reinterpret_cast<float *>(lds);
    double *scratch = reinterpret_cast<double *>(e + 4096);  // kBwScratch doubles: 2.2 KiB behind the 16 KiB of energies
    meter_lds_barrier();  // the transform's last reads of the image are done
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int o = (j + 256 * k + rot) & 4095;
            const float re2 = v[k].x * v[k].x, im2 = v[k].y * v[k].y;
            e[(o + 2048) & 4095] = re2 + im2;  // scan position of output index o: wrap = 2048
        }
    }
    meter_lds_barrier();
    return bandwidth_block256<16>(4096, j, [&](int s) { return (double)e[s]; }, double_percentile, sample_rate, scratch, total_out,
                                  tail_wave);
}

}  // namespace rr
