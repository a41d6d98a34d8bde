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

// scratch: kBwScratch doubles of LDS that nobody else touches between the first and the last barrier in here.
// e_at(s): energy (as f64) of scan position s in FORWARD order, 0 <= s < n; any lane may call it for any s.
// All 256 lanes of the workgroup must call this (it contains workgroup barriers).  Returns the bandwidth in every lane;
// *total_out (if non-null, in every lane) = the sum of all energies.
//
// Three levels, so that almost nothing runs on more than a few lanes (a full prefix scan of f64 values over 256 lanes -
// twelve 64-bit lane shuffles per direction - cost as many instructions as half the 4096-point transform in front of it):
//   A  every lane: the sum of its own C energies, for both directions             -> P[dir][256]
//   B  16 lanes per direction: the sums of 16 neighbouring P                        -> Q[dir][16]
//   C  ONE lane per direction walks Q, then the 16 P of the group where the running sum crosses the limit, then the C
//      energies of that lane's range - the reference's loop, with whole groups added at once in front of the crossing.
constexpr int kBwScratch = 2 * 256 + 2 * 16 + 4;
template <int C, class EnergyAt>
__device__ __forceinline__ double bandwidth_block256(int n, int j, EnergyAt &&e_at, double double_percentile, double sample_rate,
                                                     double *scratch, double *total_out) {
    const int c = C > 0 ? C : (n + 255) / 256;
    double *P = scratch, *Q = scratch + 512, *R = scratch + 544;
    {
        const int lo = j * c < n ? j * c : n, hi = (j + 1) * c < n ? (j + 1) * c : n;
        double pf = 0.0, pr = 0.0;  // (reverse position r is forward position n - 1 - r)
        if (C > 0) {
#pragma unroll
            for (int i = 0; i < (C > 0 ? C : 1); ++i)
                if (lo + i < hi) pf += e_at(lo + i);
#pragma unroll
            for (int i = 0; i < (C > 0 ? C : 1); ++i)
                if (lo + i < hi) pr += e_at(n - 1 - (lo + i));
        } else {
            for (int s = lo; s < hi; ++s) {
                pf += e_at(s);
                pr += e_at(n - 1 - s);
            }
        }
        P[j] = pf;
        P[256 + j] = pr;
    }
    __syncthreads();
    const int dir = j >> 6, t = j & 63;  // waves 0 and 1 do the rest: forward and reverse
    if (dir < 2 && t < 16) {
        const double *p = P + 256 * dir + 16 * t;
        double x[16], q = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = p[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) q += x[i];
        Q[16 * dir + t] = q;
    }
    __syncthreads();
    if (dir < 2 && t == 0) {
        // the reference's total is ONE sum, used for both directions: the forward groups' (both walking lanes add the same 16)
        double total = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) total += Q[i];
        const double limit = total * double_percentile / 2.0;
        double used = (double)n;  // no crossing (total <= limit): every bin is "used" (the reference's loop runs to its end)
        double cum = 0.0;
        int g = 0;
        for (; g < 16; ++g) {
            const double nw = cum + Q[16 * dir + g];
            if (nw > limit) break;
            cum = nw;
        }
        if (g < 16) {
            int b = 16 * g;
            for (; b < 16 * g + 15; ++b) {  // (the last of the group takes the crossing if rounding hides it from the others)
                const double nw = cum + P[256 * dir + b];
                if (nw > limit) break;
                cum = nw;
            }
            const int lo = b * c < n ? b * c : n, hi = (b + 1) * c < n ? (b + 1) * c : n;
            used = (double)hi;  // (crossing lost to rounding at the range's end: all of the range's bins)
            for (int s = lo; s < hi; ++s) {
                const double nw = cum + e_at(dir ? n - 1 - s : s);
                if (nw > limit) {
                    used = (double)s + (limit - cum) / (nw - cum);
                    break;
                }
                cum = nw;
            }
        }
        R[dir] = used;
        if (dir == 0) R[2] = total;
    }
    __syncthreads();
    double used_bins = 0.0;
    used_bins += R[0];
    used_bins += R[1];
    if (total_out) *total_out = R[2];
    const double bw = ((double)n - used_bins) * sample_rate / (double)n;
    return bw > 0.0 ? bw : 0.0;
}

// The epilogue behind a 4096-point transform of a 256-lane workgroup: lane j holds v[k] = X[j + 256 k], the frame's OUTPUT
// index of that bin is o = (j + 256 k + rot) & 4095 (rot = 2048 with center_dc).  `lds` is the transform's exchange image
// (>= 4096 + 64 complex f32 elements), free once every lane has left the transform: the energies (norm_sqr in f32 as the
// reference computes it, num-complex: re * re + im * im, no contraction) go there in scan order, 16 KiB, the scratch behind them.
typedef float rr_f2m __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double frame4096_bandwidth(const rr_f2m (&v)[16], void *lds, int j, int rot, double double_percentile,
                                                      double sample_rate, double *total_out) {
    float *e = reinterpret_cast<float *>(lds);
    double *scratch = reinterpret_cast<double *>(e + 4096);  // kBwScratch doubles: 4.3 KiB behind the 16 KiB of energies
    __syncthreads();  // the transform's last reads of the image are done
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int o = (j + 256 * k + rot) & 4095;
            const float re2 = v[k].x * v[k].x, im2 = v[k].y * v[k].y;
            e[(o + 2048) & 4095] = re2 + im2;  // scan position of output index o: wrap = 2048
        }
    }
    __syncthreads();
    return bandwidth_block256<16>(4096, j, [&](int s) { return (double)e[s]; }, double_percentile, sample_rate, scratch, total_out);
}

}  // namespace rr
