// rr_decim.hip — the Downsampler (resampling.rs:103-133) for ANY integer ratio and for rational ratios
// P : Q with a short period, Complex<f32>, real impulse response, on gfx950.
//
// The fused overlap-save kernels of rr_fused.hip serve the ratios 2, 4 and 8.  Everything else used to run
// the generic gather kernel k_fir at 12 % of its roofline — among it the reference's own pipelines
// (examples/bandwidth_meter/main.rs:56: 1024 k -> 102.4 k = 10 : 1, L = 145; simple_receiver.rs:28: 1024 k ->
// 384 k = 8 : 3, L = 34).  With a long impulse response and a large ratio the direct form is cheap in
// arithmetic (L / D real x complex MACs per input sample: 14.5 at 10 : 1) — what k_fir lost was occupancy
// (one 96 KiB tile per CU), per-element index logic, and LDS bank conflicts (lane stride D samples).
//
// Schedule.  Both rates integral => the reference's f64 schedule (pos += out; if pos >= in { pos -= in; emit })
// is exact and periodic: every P = in / g inputs release Q = out / g outputs.  Output m = Q a + b (b < Q) is
// released by input e_b + P a, where e_b are the first Q emission indices of the call (host, closed form:
// Schedule::first_emits).  out[m] = sum_j ir[j] x[e_m - (L - 1) + j]  (resampling.rs:112-120).
//
// Kernel.  A workgroup of 256 lanes takes a tile of TA periods (TA Q outputs).  The tile's input span is staged
// in LDS in POLYPHASE layout: the sample at position p_ref + P c + r sits at row r, column c (row stride S).
// Output (a, b) then reads, for the taps of row r, the columns a + c, c = 0 .. NC - 1: the 64 lanes of a wave
// (consecutive a, one b) read 64 consecutive LDS elements — conflict-free for every P — at addresses
// (lane term) + immediate.  The host lays the taps out to match, T[b][r][c] (zero where row r has no tap in
// column c), so the tap stream of a wave is one contiguous array read through the scalar cache.
// Per tap and lane: one 8-byte LDS read and one packed FMA.
//
// LDS per workgroup: P (TA + NC) samples (10 : 1, L = 145: 21.8 KiB, 7 workgroups per CU).  HBM traffic:
// 8 B read per input sample + 8 Q / P written = the algorithmic minimum; neighbouring tiles (which share
// P NC samples) are dealt to one XCD.  The last tile's workgroup also writes the next call's history.
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace rr {

namespace {

struct DecimArgs {
    const float2 *hist;  // positions [-hist_len, 0)
    int hist_len;
    const float2 *in;
    long n_in;
    int P, Q, NC;
    long p_ref;  // position (relative to in[0]) of tap 0 of output 0
    float2 *out;
    long n_out;
    int TA;  // periods per tile (a multiple of 64)
    int S;   // LDS row stride in samples, >= TA + NC
    unsigned ntiles;
    float2 *hist_out;  // receives the last hist_out_len samples of [ hist | in ] (may be null)
    int hist_out_len;
};

// (T is a parameter of its own, const and restrict: only then does the compiler read the wave-uniform taps through
//  the scalar cache; as a member of the argument struct they came as per-lane vector loads, waited for in every trip)
__global__ __launch_bounds__(256) void k_decim_poly(DecimArgs a, const float *__restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) char decim_smem[];
    f2 *const xs = reinterpret_cast<f2 *>(decim_smem);  // P rows of S samples
    f2 *const ost = xs + (size_t)a.P * a.S;             // Q > 1: TA Q staged outputs
    const int t = threadIdx.x;
    // tiles dealt to the XCDs in a moving window of 8 x 8: workgroups b, b + 8, .. share an XCD
    constexpr unsigned G = 8;
    const unsigned grp = blockIdx.x / (8 * G), rem = blockIdx.x % (8 * G);
    const unsigned tile = grp * 8 * G + (rem & 7) * G + (rem >> 3);
    if (tile >= a.ntiles) return;
    const int P = a.P, Q = a.Q, NC = a.NC, TA = a.TA, S = a.S;
    const long a0 = (long)tile * TA;
    const long p_lo = a.p_ref + (long)P * a0;  // position of row 0, column 0
    const int nld = P * (TA + NC);             // samples of the tile, row-major in time: q = P c + r

    if (a.hist_out && tile == a.ntiles - 1) {
        for (int i = t; i < a.hist_out_len; i += 256) {
            const long pos = a.n_in - a.hist_out_len + i;
            float2 h;
            h.x = 0.f;
            h.y = 0.f;
            if (pos >= 0) h = a.in[pos];
            else if (pos >= -(long)a.hist_len) h = a.hist[a.hist_len + pos];
            a.hist_out[i] = h;
        }
    }

    // ---- stage: coalesced 8-byte loads, polyphase scatter into LDS ---------------------------------------
    {
        int row = t % P, col = t / P;
        const int dr = 256 % P, dc = 256 / P;
        const bool interior = p_lo >= 0 && p_lo + nld <= a.n_in;
        if (interior) {
            const f2 *src = reinterpret_cast<const f2 *>(a.in + p_lo);
#pragma unroll 4
            for (int q = t; q < nld; q += 256) {
                const f2 v = __builtin_nontemporal_load(src + q);
                xs[row * S + col] = v;
                row += dr;
                col += dc;
                if (row >= P) {
                    row -= P;
                    ++col;
                }
            }
        } else {
            // edges: the history in front (zeros before it), nothing behind the input
            for (int q = t; q < nld; q += 256) {
                const long pos = p_lo + q;
                float2 xv;
                xv.x = 0.f;
                xv.y = 0.f;
                if (pos >= 0) {
                    if (pos < a.n_in) xv = a.in[pos];
                } else if (pos >= -(long)a.hist_len) {
                    xv = a.hist[a.hist_len + pos];
                }
                xs[row * S + col] = (f2){xv.x, xv.y};
                row += dr;
                col += dc;
                if (row >= P) {
                    row -= P;
                    ++col;
                }
            }
        }
    }
    __syncthreads();

    // ---- filter: a wave per (phase b, run of 64 periods); lane = period ----------------------------------
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int segs = TA >> 6, ntask = Q * segs;
    for (int task = w; task < ntask; task += 4) {
        const int b = task / segs, seg = task - b * segs;
        const int al = seg * 64 + lane;
        const f2 *base = xs + al;
        const float *tb = T + (size_t)b * P * NC;
        f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
        for (int r = 0; r < P; ++r) {
            const f2 *row = base + r * S;
            const float *tr = tb + r * NC;
#pragma unroll 4
            for (int c = 0; c < NC; c += 4) {
                const float4 t4 = *reinterpret_cast<const float4 *>(tr + c);  // uniform address: a scalar read
                const f2 x0 = lds_ldv(row + c), x1 = lds_ldv(row + c + 1), x2 = lds_ldv(row + c + 2), x3 = lds_ldv(row + c + 3);
                acc0 = __builtin_elementwise_fma(x0, (f2){t4.x, t4.x}, acc0);
                acc1 = __builtin_elementwise_fma(x1, (f2){t4.y, t4.y}, acc1);
                acc0 = __builtin_elementwise_fma(x2, (f2){t4.z, t4.z}, acc0);
                acc1 = __builtin_elementwise_fma(x3, (f2){t4.w, t4.w}, acc1);
            }
        }
        const f2 acc = acc0 + acc1;
        if (Q == 1) {
            const long m = a0 + al;
            if (m < a.n_out) __builtin_nontemporal_store(acc, reinterpret_cast<f2 *>(a.out) + m);
        } else {
            ost[Q * al + b] = acc;
        }
    }
    if (Q > 1) {
        __syncthreads();
        const long m0 = (long)Q * a0;
        for (int i = t; i < TA * Q; i += 256) {
            const long m = m0 + i;
            if (m < a.n_out) __builtin_nontemporal_store(ost[i], reinterpret_cast<f2 *>(a.out) + m);
        }
    }
}

}  // namespace

// LDS budget per workgroup: 4 or more workgroups per CU
static constexpr size_t kDecimLds = 40 * 1024;

static int decim_geometry(size_t P, size_t Q, size_t NC, int *TA, int *S) {
    for (int ta : {256, 128, 64}) {
        int s = ta + (int)NC;
        if (!(s & 1)) ++s;  // odd row stride: the staging writes of neighbouring rows fall on different banks
        const size_t bytes = (P * (size_t)s + (Q > 1 ? (size_t)ta * Q : 0)) * 8;
        if (bytes <= kDecimLds) {
            *TA = ta;
            *S = s;
            return (int)bytes;
        }
    }
    return 0;
}

bool decim_poly_supported(int dtype, uint64_t P, uint64_t Q, size_t L) {
    if (dtype != RR_F32 || P < 2 || P > 512 || Q < 1 || Q > 8 || Q >= P || L < 1) return false;
    const size_t NC = ((P - 1 + L + P - 1) / P + 3) / 4 * 4;
    int ta, s;
    return decim_geometry(P, Q, NC, &ta, &s) != 0;
}

// taps in the kernel's order: T[b][r][c] = ir[j] for (delta_b + j) = P c + r, zero elsewhere; delta_b = e[b] - e[0]
void build_decim_poly_taps(const std::vector<double> &ir, uint64_t P, uint64_t Q, const int64_t *e_first, std::vector<float> &T,
                           int *NC_out) {
    const size_t L = ir.size();
    size_t maxd = 0;
    for (uint64_t b = 0; b < Q; ++b) maxd = std::max(maxd, (size_t)(e_first[b] - e_first[0]));
    const size_t NC = ((maxd + L + P - 1) / P + 3) / 4 * 4;
    T.assign((size_t)Q * P * NC, 0.f);
    for (uint64_t b = 0; b < Q; ++b) {
        const size_t d = (size_t)(e_first[b] - e_first[0]);
        for (size_t j = 0; j < L; ++j) {
            const size_t idx = d + j, r = idx % P, c = idx / P;
            T[((size_t)b * P + r) * NC + c] = (float)ir[j];
        }
    }
    *NC_out = (int)NC;
}

int launch_decim_poly(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *T,
                      uint64_t P, uint64_t Q, int NC, size_t L, int64_t e_first0, void *out, size_t n_out, void *hist_out,
                      size_t hist_out_len) {
    if (n_out == 0) return RR_OK;
    DecimArgs a;
    a.hist = (const float2 *)hist;
    a.hist_len = (int)hist_len;
    a.in = (const float2 *)in;
    a.n_in = (long)n_in;
    a.P = (int)P;
    a.Q = (int)Q;
    a.NC = NC;
    a.p_ref = (long)e_first0 - (long)(L - 1);
    a.out = (float2 *)out;
    a.n_out = (long)n_out;
    a.hist_out = (float2 *)hist_out;
    a.hist_out_len = (int)hist_out_len;
    const int lds = decim_geometry(P, Q, (size_t)NC, &a.TA, &a.S);
    if (!lds) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: %llu : %llu with %d tap columns does not fit the LDS tile",
                      (unsigned long long)P, (unsigned long long)Q, NC);
    const size_t per_tile = (size_t)a.TA * Q;
    const size_t ntiles = (n_out + per_tile - 1) / per_tile;
    if (ntiles > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: too many tiles");
    a.ntiles = (unsigned)ntiles;
    const unsigned grid = (unsigned)((ntiles + 63) / 64 * 64);
    hipLaunchKernelGGL(k_decim_poly, dim3(grid), dim3(256), (size_t)lds, s, a, (const float *)T);  // T: [Q][P][NC] taps
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
