// rr_decim.hip — the Downsampler (resampling.rs:103-133) for ANY integer ratio and for rational ratios
// P : Q with a short period, Complex<f32>, real impulse response, on gfx950.
//
// The fused overlap-save kernels of rr_ols.hip serve the ratios 2, 4 and 8.  Everything else used to run
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
// (lane term) + a wave-uniform offset.  The host hands the taps over as a flat list per phase b,
// T[b][j] = { ir[j], byte offset of (row, column) of tap j }, read through the scalar cache (8 taps per
// s_load_dwordx16): exactly L taps per output whatever P is.  Per tap and lane: one 8-byte LDS read, one
// address add and one packed FMA (k_decim_poly<false>); with two neighbouring periods per lane one 16-byte read per
// four products (k_decim_poly<true>, the default where its tile fits: see the kernel).
//
// LDS per workgroup: P (TA + NC) samples (10 : 1, L = 145: 21.8 KiB, 7 workgroups per CU).  HBM traffic:
// 8 B read per input sample + 8 Q / P written = the algorithmic minimum; neighbouring tiles (which share
// P NC samples) are dealt to one XCD.  The last tile's workgroup also writes the next call's history.
#include "rr_blocks.hpp"
#include "rr_wave_math.hpp"

#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>

namespace rr {

namespace {

struct DecimArgs {
    const float2 *hist;  // positions [-hist_len, 0)
    int hist_len;
    const float2 *in;
    long n_in;
    int P, Q, NC, Lp;  // NC: tap columns (geometry only); Lp: taps per phase, padded to a multiple of 8
    int Lp2;           // PAIR: entries per phase of T2, a multiple of 4
    long p_ref;        // position (relative to in[0]) of tap 0 of output 0
    float2 *out;
    long n_out;
    int TA;  // periods per tile (a multiple of 64)
    int S;   // LDS row stride in samples, >= TA + NC
    unsigned ntiles;
    float2 *hist_out;  // receives the last hist_out_len samples of [ hist | in ] (may be null)
    int hist_out_len;
    // optional mixer in front (FreqShifter fused, transform.rs:341-348): samples of `in` are multiplied by
    // nco[(idx0 + pos) mod denom] as they are staged; `hist` and `hist_out` then hold MIXED samples.  denom = 0: no mixer
    const float2 *nco;
    unsigned denom, idx0;
    // phase of the first sample of tile 0, (idx0 + p_ref) mod denom, the advance per tile (P TA) mod denom, 1 / denom:
    // a tile's phase comes from three f64 operations (exact below 2^53) instead of a 64-bit integer division per lane
    unsigned ph_ref, ph_tile_step;
    double inv_denom;
};

// (T is a parameter of its own, const and restrict: only then does the compiler read the wave-uniform taps through
//  the scalar cache; as a member of the argument struct they came as per-lane vector loads, waited for in every trip)
// PAIR: a lane takes TWO neighbouring periods.  Period a + 1 reads, for the taps of a row, the columns one further on than period
// a - so one 16-byte LDS read of the columns (a + k, a + k + 1) serves four products,
//   acc(a) += t[k] x0 + t[k + 1] x1,   acc(a + 1) += t[k - 1] x0 + t[k] x1     (t = the row's taps by column, zero outside),
// where the one-period form reads 8 bytes per product: the inner loop is bound by the LDS (ds_read_b64: 2 LDS cycles per wave and
// tap against the 1 cycle per wave the four SIMDs need for its packed FMA), and ds_read_b128 moves 16 bytes per lane in 4.  The
// host's table T2 holds per phase the entries { t[k - 1], t[k], t[k + 1], byte offset of (row, k) } with (row S + k) even (16-byte
// aligned reads), k from c_lo - 1 to c_hi + 1: two more staged columns than the one-period form, all of the image initialised
// (what lies beyond a row's taps is multiplied by zero, so it must be finite).
template <bool PAIR>
__global__ __launch_bounds__(256) void k_decim_poly(DecimArgs a, const uint2 *__restrict__ T, const uint4 *__restrict__ T2) {
    extern __shared__ __attribute__((aligned(16))) char decim_smem[];
    f2 *const xs = reinterpret_cast<f2 *>(decim_smem);  // P rows of S samples (PAIR: + 2 zeros behind them)
    f2 *const ost = xs + (size_t)a.P * a.S + (PAIR ? 2 : 0);  // Q > 1: TA Q staged outputs
    const int t = threadIdx.x;
    // tiles dealt to the XCDs in a moving window of 8 x 8: workgroups b, b + 8, .. share an XCD
    constexpr unsigned G = 8;
    const unsigned grp = blockIdx.x / (8 * G), rem = blockIdx.x % (8 * G);
    const unsigned tile = grp * 8 * G + (rem & 7) * G + (rem >> 3);
    if (tile >= a.ntiles) return;
    const int P = a.P, Q = a.Q, TA = a.TA, S = a.S;
    const long a0 = (long)tile * TA;
    const long p_lo = a.p_ref + (long)P * a0;  // position of row 0, column 0
    const int nld = P * (TA + a.NC);           // samples of the tile, row-major in time: q = P c + r

    if (a.hist_out && tile == a.ntiles - 1) {
        for (int i = t; i < a.hist_out_len; i += 256) {
            const long pos = a.n_in - a.hist_out_len + i;
            float2 h;
            h.x = 0.f;
            h.y = 0.f;
            if (pos >= 0) {
                h = a.in[pos];
                if (a.denom) {
                    const float2 pp = a.nco[(unsigned)(((long)a.idx0 + pos) % (long)a.denom)];
                    const float2 x = h;
                    h.x = x.x * pp.x - x.y * pp.y;
                    h.y = x.x * pp.y + x.y * pp.x;
                }
            } else if (pos >= -(long)a.hist_len) {
                h = a.hist[a.hist_len + pos];
            }
            a.hist_out[i] = h;
        }
    }

    // ---- stage: coalesced 8-byte loads (8 in flight per lane), polyphase scatter into LDS ------------------
    {
        int row = t % P, col = t / P;
        const int dr = 256 % P, dc = 256 / P;
        auto step = [&] {
            row += dr;
            col += dc;
            if (row >= P) {
                row -= P;
                ++col;
            }
        };
        const bool interior = p_lo >= 0 && p_lo + nld <= a.n_in;
        // NCO phase of this lane's first sample and its step per 256 samples (interior tiles: every sample is in `in`)
        unsigned ph = 0, dph = 0, dph8 = 0;
        f2 rot = {1.f, 0.f};
        if (a.denom) {
            const double dn = (double)a.denom;
            const double prod = __builtin_fma((double)tile, (double)a.ph_tile_step, (double)a.ph_ref);
            const double qd = __builtin_floor(prod * a.inv_denom);
            double rd = __builtin_fma(-qd, dn, prod);
            if (rd < 0.0) rd += dn;
            if (rd >= dn) rd -= dn;
            unsigned r0 = (unsigned)rd + (unsigned)t;  // < 2 denom where denom >= 256
            if (a.denom >= 256u) {
                if (r0 >= a.denom) r0 -= a.denom;
            } else {
                r0 %= a.denom;
            }
            ph = r0;
            dph = 256u % a.denom;
            dph8 = 2048u % a.denom;
            // the PURE rotation by 256 samples, e^{j 2 pi (256 numer mod denom) / denom}, kept behind the table
            // (rr_freqshifter::prepare: entry denom + 1 + k steps 128 k samples).  The table's own entry dph is
            // that rotation times e^{j start_phase} (transform.rs:322-337: after a retune or a rate change the
            // table starts at the phase of the current phasor), so it must not serve as a step
            const float2 rt = a.nco[a.denom + 1 + 2];
            rot = (f2){rt.x, rt.y};
        }
        if (interior) {
            const f2 *src = reinterpret_cast<const f2 *>(a.in + p_lo);
            int q = t;
            for (; q + 7 * 256 < nld; q += 8 * 256) {
                f2 v[8];
                // (the phasor first: loads complete in order, so asked for last it would make the first use wait for all
                //  eight samples)
                float2 p0 = {1.f, 0.f};
                if (a.denom) p0 = a.nco[ph];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + q + 256 * u);
                if (a.denom) {
                    // the eight phasors are requested together with the samples (asked for one by one behind the
                    // samples they were eight more round trips per batch: 0.212 ms for the meter's front end)
                    // .. one table entry per batch, requested with the samples; the other seven by the rotation of 256
                    // samples (the table is a geometric sequence: a last-bit difference from its own entries, as in
                    // k_ols_wave's general-period path)
                    ph += dph8;
                    if (ph >= a.denom) ph -= a.denom;
                    f2 pp = {p0.x, p0.y};
                    v[0] = cmul(v[0], pp);
#pragma unroll
                    for (int u = 1; u < 8; ++u) {
                        pp = cmul(pp, rot);
                        v[u] = cmul(v[u], pp);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    xs[row * S + col] = v[u];
                    step();
                }
            }
            // the rest (fewer than 8 x 256 samples - with a short period, all of the tile) in batches of 4 and 2, so that a lane
            // still has several loads in flight (3 : 1: tiles of 1584 samples, which took six round trips one behind the other;
            // one batch of 8 with every load and store predicated was slower everywhere: 10 : 1 0.146 -> 0.160 ms)
            auto batch = [&](auto nb) {
                constexpr int NB = decltype(nb)::value;
                for (; q + (NB - 1) * 256 < nld; q += NB * 256) {
                    f2 v[NB];
                    float2 pp[NB];
                    // (the phasors first, all of them: asked for one by one behind the samples they are a round trip each)
                    if (a.denom) {
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            pp[u] = a.nco[ph];
                            ph += dph;
                            if (ph >= a.denom) ph -= a.denom;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NB; ++u) v[u] = __builtin_nontemporal_load(src + q + 256 * u);
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        xs[row * S + col] = a.denom ? cmul(v[u], (f2){pp[u].x, pp[u].y}) : v[u];
                        step();
                    }
                }
            };
            batch(std::integral_constant<int, 4>{});
            batch(std::integral_constant<int, 2>{});
            batch(std::integral_constant<int, 1>{});
        } else {
            // edges: the history in front (zeros before it), nothing behind the input
            for (int q = t; q < nld; q += 256) {
                const long pos = p_lo + q;
                float2 xv;
                xv.x = 0.f;
                xv.y = 0.f;
                if (pos >= 0) {
                    if (pos < a.n_in) {
                        xv = a.in[pos];
                        if (a.denom) {
                            const float2 pp = a.nco[(unsigned)(((long)a.idx0 + pos) % (long)a.denom)];
                            const float2 x = xv;
                            xv.x = x.x * pp.x - x.y * pp.y;
                            xv.y = x.x * pp.y + x.y * pp.x;
                        }
                    }
                } else if (pos >= -(long)a.hist_len) {
                    xv = a.hist[a.hist_len + pos];  // (already mixed)
                }
                xs[row * S + col] = (f2){xv.x, xv.y};
                step();
            }
        }
    }
    if constexpr (PAIR) {  // the columns no sample was staged into, and the two elements behind the last row
        for (int i = t; i < P * (S - (TA + a.NC)); i += 256) {
            const int r = i % P, c = TA + a.NC + i / P;
            xs[r * S + c] = (f2){0.f, 0.f};
        }
        if (t < 2) xs[P * S + t] = (f2){0.f, 0.f};
    }
    __syncthreads();

    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    if constexpr (PAIR) {
        // ---- filter: a wave per (phase b, run of 128 periods); lane = the periods 2 lane, 2 lane + 1 ----------------
        const int segs = TA >> 7, ntask = Q * segs;
        for (int task = w; task < ntask; task += 4) {
            const int b = task / segs, seg = task - b * segs;
            const int al = seg * 128 + 2 * lane;
            const char *base = reinterpret_cast<const char *>(xs + al);
            const uint4 *tl = T2 + (size_t)b * a.Lp2;
            f2 a0x = {0.f, 0.f}, a0y = {0.f, 0.f}, a1x = {0.f, 0.f}, a1y = {0.f, 0.f};
            for (int i = 0; i < a.Lp2; i += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint4 e = tl[i + u];  // uniform address: a scalar read
                    const float4 x = *reinterpret_cast<const float4 *>(base + e.w);
                    const float tm = __uint_as_float(e.x), t0 = __uint_as_float(e.y), tp = __uint_as_float(e.z);
                    const f2 x0 = {x.x, x.y}, x1 = {x.z, x.w};
                    a0x = __builtin_elementwise_fma(x0, (f2){t0, t0}, a0x);
                    a0y = __builtin_elementwise_fma(x1, (f2){tp, tp}, a0y);
                    a1x = __builtin_elementwise_fma(x0, (f2){tm, tm}, a1x);
                    a1y = __builtin_elementwise_fma(x1, (f2){t0, t0}, a1y);
                }
            }
            const f2 r0 = a0x + a0y, r1 = a1x + a1y;
            if (Q == 1) {
                const long m = a0 + al;
                typedef float f4s __attribute__((ext_vector_type(4), aligned(8)));
                if (m + 1 < a.n_out) __builtin_nontemporal_store((f4s){r0.x, r0.y, r1.x, r1.y}, reinterpret_cast<f4s *>(a.out + m));
                else if (m < a.n_out) __builtin_nontemporal_store(r0, reinterpret_cast<f2 *>(a.out) + m);
            } else {
                ost[Q * al + b] = r0;
                ost[Q * (al + 1) + b] = r1;
            }
        }
    } else {
    // ---- filter: a wave per (phase b, run of 64 periods); lane = period ----------------------------------
    const int segs = TA >> 6, ntask = Q * segs;
    for (int task = w; task < ntask; task += 4) {
        const int b = task / segs, seg = task - b * segs;
        const int al = seg * 64 + lane;
        const char *base = reinterpret_cast<const char *>(xs + al);
        const uint2 *tl = T + (size_t)b * a.Lp;
        f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
        for (int i = 0; i < a.Lp; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const uint2 e0 = tl[i + u], e1 = tl[i + u + 1];  // uniform addresses: scalar reads
                const f2 x0 = lds_ldv(reinterpret_cast<const f2 *>(base + e0.y));
                const f2 x1 = lds_ldv(reinterpret_cast<const f2 *>(base + e1.y));
                const float t0 = __uint_as_float(e0.x), t1 = __uint_as_float(e1.x);
                acc0 = __builtin_elementwise_fma(x0, (f2){t0, t0}, acc0);
                acc1 = __builtin_elementwise_fma(x1, (f2){t1, t1}, acc1);
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

// Complex<f64>: the same polyphase staging and tap list, without the mixer (the f64 FreqShifter stays its own block)
// and without the f32 kernel's tuning - 16-byte LDS elements, taps as { f64 value, byte offset } entries of 16 bytes.
typedef double d2 __attribute__((ext_vector_type(2)));
struct DecimArgs64 {
    const double2 *hist;
    int hist_len;
    const double2 *in;
    long n_in;
    int P, Q, NC, Lp;
    long p_ref;
    double2 *out;
    long n_out;
    int TA, S;
    unsigned ntiles;
    double2 *hist_out;
    int hist_out_len;
    // optional mixer in front (the f64 chain's fused step): samples of `in` times nco[(idx0 + pos) mod denom] as they are staged;
    // `hist` and `hist_out` hold MIXED samples.  denom = 0: no mixer
    const double2 *nco;
    unsigned denom, idx0;
};
struct Tap64 {
    double tap;
    unsigned off, pad;
};
__global__ __launch_bounds__(256) void k_decim_poly_f64(DecimArgs64 a, const Tap64 *__restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) char decim_smem[];
    d2 *const xs = reinterpret_cast<d2 *>(decim_smem);
    d2 *const ost = xs + (size_t)a.P * a.S;
    const int t = threadIdx.x;
    constexpr unsigned G = 8;
    const unsigned grp = blockIdx.x / (8 * G), rem = blockIdx.x % (8 * G);
    const unsigned tile = grp * 8 * G + (rem & 7) * G + (rem >> 3);
    if (tile >= a.ntiles) return;
    const int P = a.P, Q = a.Q, TA = a.TA, S = a.S;
    const long a0 = (long)tile * TA;
    const long p_lo = a.p_ref + (long)P * a0;
    const int nld = P * (TA + a.NC);
    auto mix = [&](double2 x, long pos) -> double2 {  // x * nco[(idx0 + pos) mod denom], pos >= 0
        const double2 pp = a.nco[(unsigned)(((long)a.idx0 + pos) % (long)a.denom)];
        double2 r;
        r.x = x.x * pp.x - x.y * pp.y;
        r.y = x.x * pp.y + x.y * pp.x;
        return r;
    };
    auto fetch = [&](long pos) -> double2 {
        double2 h;
        h.x = 0.0;
        h.y = 0.0;
        if (pos >= 0) {
            if (pos < a.n_in) {
                h = a.in[pos];
                if (a.denom) h = mix(h, pos);
            }
        } else if (pos >= -(long)a.hist_len) {
            h = a.hist[a.hist_len + pos];  // (already mixed)
        }
        return h;
    };
    if (a.hist_out && tile == a.ntiles - 1)
        for (int i = t; i < a.hist_out_len; i += 256) a.hist_out[i] = fetch(a.n_in - a.hist_out_len + i);
    {
        int row = t % P, col = t / P;
        const int dr = 256 % P, dc = 256 / P;
        const bool interior = p_lo >= 0 && p_lo + nld <= a.n_in;
        auto put = [&](double2 xv) {
            xs[row * S + col] = (d2){xv.x, xv.y};
            row += dr;
            col += dc;
            if (row >= P) {
                row -= P;
                ++col;
            }
        };
        int q = t;
        if (interior) {  // four loads of a lane in flight (as the f32 kernel's batches)
            const double2 *src = a.in + p_lo;
            // the lane's table index walks with its samples: + 256 per load, wrapped (one 64-bit division per lane and tile)
            unsigned ph = 0, dph = 0;
            if (a.denom) {
                ph = (unsigned)(((long)a.idx0 + p_lo + q) % (long)a.denom);
                dph = 256u % a.denom;
            }
            auto next_ph = [&] {
                const unsigned r = ph;
                ph += dph;
                if (ph >= a.denom) ph -= a.denom;
                return r;
            };
            for (; q + 3 * 256 < nld; q += 4 * 256) {
                double2 v[4], pp[4];
                if (a.denom) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) pp[u] = a.nco[next_ph()];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = src[q + 256 * u];
                if (a.denom) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double2 x = v[u];
                        v[u].x = x.x * pp[u].x - x.y * pp[u].y;
                        v[u].y = x.x * pp[u].y + x.y * pp[u].x;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) put(v[u]);
            }
            for (; q < nld; q += 256) {
                double2 x = src[q];
                if (a.denom) {
                    const double2 pp = a.nco[next_ph()];
                    const double2 y = x;
                    x.x = y.x * pp.x - y.y * pp.y;
                    x.y = y.x * pp.y + y.y * pp.x;
                }
                put(x);
            }
        } else {
            for (; q < nld; q += 256) put(fetch(p_lo + q));
        }
    }
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int segs = TA >> 6, ntask = Q * segs;
    for (int task = w; task < ntask; task += 4) {
        const int b = task / segs, seg = task - b * segs;
        const int al = seg * 64 + lane;
        const char *base = reinterpret_cast<const char *>(xs + al);
        const Tap64 *tl = T + (size_t)b * a.Lp;
        d2 acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
        for (int i = 0; i < a.Lp; i += 2) {
            const Tap64 e0 = tl[i], e1 = tl[i + 1];  // uniform addresses: scalar reads
            const d2 x0 = *reinterpret_cast<const d2 *>(base + e0.off);
            const d2 x1 = *reinterpret_cast<const d2 *>(base + e1.off);
            acc0 = __builtin_elementwise_fma(x0, (d2){e0.tap, e0.tap}, acc0);
            acc1 = __builtin_elementwise_fma(x1, (d2){e1.tap, e1.tap}, acc1);
        }
        const d2 acc = acc0 + acc1;
        if (Q == 1) {
            const long m = a0 + al;
            if (m < a.n_out) reinterpret_cast<d2 *>(a.out)[m] = acc;
        } else {
            ost[Q * al + b] = acc;
        }
    }
    if (Q > 1) {
        __syncthreads();
        const long m0 = (long)Q * a0;
        for (int i = t; i < TA * Q; i += 256) {
            const long m = m0 + i;
            if (m < a.n_out) reinterpret_cast<d2 *>(a.out)[m] = ost[i];
        }
    }
}


// ---------------------------------------------------------------------------
// k_decim_poly_f64r<R>: the same sums with R = 4 or 8 neighbouring periods per lane.  k_decim_poly_f64 reads 16 bytes of LDS for
// every pair of f64 FMAs - 1 KiB per wave and tap against the CU's 128 bytes per clock: LDS-bound at a fifth of the f64 rate
// (measured: 0.18 ms per 2^24 samples at 4 : 1 with 184 taps, exactly 8 clocks per wave and tap).  Outputs a and a + 1 of one phase
// b use the same row of the staged samples one column apart, x[r][a + c] and x[r][a + 1 + c]: a lane that owns the periods
// R l .. R l + R - 1 walks a row once - sample X_i = x[r][R l + c_lo + i] feeds acc_u += tap[i - u] X_i, u < R - and reads each
// sample ONCE for up to R taps: 1 / R of the LDS traffic, 2 R FMAs per read.  The columns are staged in R interleaved sets
// (column = R cc + s: set s, place cc), so that the lanes of a wave read neighbouring 16-byte elements whatever i is.
// Taps: per (phase b, row r) the first column c_lo and a run of `steps` f64 taps (zeros behind the row's own), read through the
// scalar cache (from LDS they cost the same 1 KiB per wave as a sample: uniform or not, a read returns 16 bytes to every lane).
// P >= 1 (P = Q = 1: a plain FIR).
// (The same form for Complex<f32> - measured in round 3, with the LDS offsets precomputed per step as k_decim_poly has them - was
//  SLOWER than k_decim_poly: 10 : 1 / 145 taps 0.187 against 0.136 ms per 2^26 samples, 8 : 3 0.178 against 0.149, 3 : 2 0.269
//  against 0.209; only 5 : 1 with 183 taps came out level.  k_decim_poly's staging and tile sizes carry it; not kept.)
// ---------------------------------------------------------------------------
struct DecimArgsR4 {
    const double2 *hist;
    int hist_len;
    const double2 *in;
    long n_in;
    int P, Q;
    long p_ref;
    double2 *out;
    long n_out;
    int TA, SR, steps, NCX;  // periods per tile (a multiple of 64 R); places per row of a set; tap steps per row (a multiple of 4); staged columns beyond TA
    unsigned ntiles;
    double2 *hist_out;
    int hist_out_len;
    const double2 *nco;
    unsigned denom, idx0;
    const int *CL;      // c_lo[b P + r]
    const double *TE;   // `steps` taps per (b, r)
};

__device__ __forceinline__ double ld_uniform_f64(const double *p) {  // (an s_load: see ld_uniform in rr_ols.hip)
    return *(const double __attribute__((address_space(4))) *)(unsigned long long)p;
}

template <int R>
__global__ __launch_bounds__(256) void k_decim_poly_f64r(DecimArgsR4 a) {
    extern __shared__ __attribute__((aligned(16))) char decim_smem[];
    const int P = a.P, Q = a.Q, TA = a.TA, SR = a.SR, steps = a.steps;
    d2 *const xs = reinterpret_cast<d2 *>(decim_smem);                 // R sets x P rows x SR places
    d2 *const ost = xs + (size_t)R * P * SR;                            // Q > 1: TA Q outputs in their order
    const int t = threadIdx.x;
    constexpr unsigned G = 8;
    const unsigned grp = blockIdx.x / (8 * G), rem = blockIdx.x % (8 * G);
    const unsigned tile = grp * 8 * G + (rem & 7) * G + (rem >> 3);
    if (tile >= a.ntiles) return;
    const long a0 = (long)tile * TA;
    const long p_lo = a.p_ref + (long)P * a0;
    const int nld = P * (TA + a.NCX);
    auto mix = [&](double2 x, long pos) -> double2 {  // x * nco[(idx0 + pos) mod denom], pos >= 0
        const double2 pp = a.nco[(unsigned)(((long)a.idx0 + pos) % (long)a.denom)];
        double2 r;
        r.x = x.x * pp.x - x.y * pp.y;
        r.y = x.x * pp.y + x.y * pp.x;
        return r;
    };
    auto fetch = [&](long pos) -> double2 {
        double2 h;
        h.x = 0.0;
        h.y = 0.0;
        if (pos >= 0) {
            if (pos < a.n_in) {
                h = a.in[pos];
                if (a.denom) h = mix(h, pos);
            }
        } else if (pos >= -(long)a.hist_len) {
            h = a.hist[a.hist_len + pos];  // (already mixed)
        }
        return h;
    };
    if (a.hist_out && tile == a.ntiles - 1)
        for (int i = t; i < a.hist_out_len; i += 256) a.hist_out[i] = fetch(a.n_in - a.hist_out_len + i);
    {
        int row = t % P, col = t / P;
        const int dr = 256 % P, dc = 256 / P;
        const bool interior = p_lo >= 0 && p_lo + nld <= a.n_in;
        auto put = [&](double2 xv) {
            xs[((col & (R - 1)) * P + row) * SR + (col / R)] = (d2){xv.x, xv.y};
            row += dr;
            col += dc;
            if (row >= P) {
                row -= P;
                ++col;
            }
        };
        int q = t;
        if (interior) {
            const double2 *src = a.in + p_lo;
            unsigned ph = 0, dph = 0;
            if (a.denom) {
                ph = (unsigned)(((long)a.idx0 + p_lo + q) % (long)a.denom);
                dph = 256u % a.denom;
            }
            auto next_ph = [&] {
                const unsigned r = ph;
                ph += dph;
                if (ph >= a.denom) ph -= a.denom;
                return r;
            };
            for (; q + 3 * 256 < nld; q += 4 * 256) {
                double2 v[4], pp[4];
                if (a.denom) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) pp[u] = a.nco[next_ph()];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = src[q + 256 * u];
                if (a.denom) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double2 x = v[u];
                        v[u].x = x.x * pp[u].x - x.y * pp[u].y;
                        v[u].y = x.x * pp[u].y + x.y * pp[u].x;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) put(v[u]);
            }
            for (; q < nld; q += 256) {
                double2 x = src[q];
                if (a.denom) {
                    const double2 pp = a.nco[next_ph()];
                    const double2 y = x;
                    x.x = y.x * pp.x - y.y * pp.y;
                    x.y = y.x * pp.y + y.y * pp.x;
                }
                put(x);
            }
        } else {
            for (; q < nld; q += 256) put(fetch(p_lo + q));
        }
    }
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int segs = TA / (64 * R), ntask = Q * segs;  // a wave takes 64 R periods of one phase
    for (int task = w; task < ntask; task += 4) {
        const int b = task / segs, seg = task - b * segs;
        const int lb = seg * 64 + lane;  // the lane's periods: R lb .. R lb + R - 1
        d2 acc[R];
#pragma unroll
        for (int u = 0; u < R; ++u) acc[u] = (d2){0.0, 0.0};
        for (int r = 0; r < P; ++r) {
            const int c_lo = a.CL[b * P + r];  // (uniform: a scalar read)
            const double *te = a.TE + (size_t)(b * P + r) * steps;
            double tp[R];  // te[i0 - R + k], k < R: the taps in front of the chunk (zeros in front of the row's first)
#pragma unroll
            for (int k = 0; k < R; ++k) tp[k] = 0.0;
            for (int i0 = 0; i0 < steps; i0 += 4) {
                double tc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) tc[k] = ld_uniform_f64(te + i0 + k);
                d2 X[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ci = c_lo + i0 + k;
                    X[k] = xs[((ci & (R - 1)) * P + r) * SR + (ci / R) + lb];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const double tv = k >= u ? tc[k - u] : tp[R + k - u];  // te[i0 + k - u]
                        acc[u] = __builtin_elementwise_fma(X[k], (d2){tv, tv}, acc[u]);
                    }
                }
#pragma unroll
                for (int k = 0; k + 4 < R; ++k) tp[k] = tp[k + 4];
#pragma unroll
                for (int k = 0; k < 4; ++k) tp[R - 4 + k] = tc[k];
            }
        }
        if (Q == 1) {
            const long m = a0 + R * lb;
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (m + u < a.n_out) reinterpret_cast<d2 *>(a.out)[m + u] = acc[u];
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) ost[Q * (R * lb + u) + b] = acc[u];
        }
    }
    if (Q > 1) {
        __syncthreads();
        const long m0 = (long)Q * a0;
        for (int i = t; i < TA * Q; i += 256) {
            const long m = m0 + i;
            if (m < a.n_out) reinterpret_cast<d2 *>(a.out)[m] = ost[i];
        }
    }
}

}  // namespace

// LDS per workgroup: tiles of about 24 KiB (6 workgroups per CU) where the period allows, never more than 64 KiB
#ifndef RR_V_DECIM_LDS_KB
#define RR_V_DECIM_LDS_KB 26  // (26: 3 : 1 takes tiles of 1024 periods = 24.4 KiB instead of 512)
#endif
static constexpr size_t kDecimLdsTarget = RR_V_DECIM_LDS_KB * 1024, kDecimLdsMax = 64 * 1024;

static size_t decim_nc(uint64_t P, size_t L) { return (P - 1 + L + P - 1) / P; }  // tap columns for any phase offset

// k_decim_poly_f64r4's tile: steps per row, staged columns beyond TA, places per row of a set, LDS bytes (0: does not fit)
struct DecimR4Plan {
    int R = 0, TA = 0, SR = 0, steps = 0, NCX = 0;
    size_t lds = 0;
};
static DecimR4Plan decim_r4_plan(size_t P, size_t Q, size_t L) {
    DecimR4Plan pl;
    // (=0: k_decim_poly_f64 as before; =8: R = 8 where its tile of 512 periods fits - A/B runs, tests.  Measured, f64 chain + f64
    //  Downsampler per 2^24 samples: R = 4 in tiles of 256 periods (8 workgroups per CU) 145 us, R = 8 in tiles of 512 (4 per CU) 196,
    //  one period per lane 158: a wave issues a v_fma_f64 every ~13 clocks on its own (scripts/ubench/valu_rate_f64: 23 TFLOP/s at one
    //  wave per SIMD, 51 at four) - the waves per SIMD count for more than the LDS traffic saved.  Of R = 8's 196 us, 78 are staging
    //  and stores (the tap loop cut to one step), 12 the stores alone)
    const char *e = std::getenv("RR_DECIM_F64_R4");
    const int mode = e ? std::atoi(e) : 4;
    if (mode == 0 || P < 1 || P > 64 || Q < 1 || Q > 8 || L < 1) return pl;
    const size_t CM = (L + P - 1) / P;  // taps of a row (the integers = r mod P in a run of L)
    for (int R : {8, 4}) {
        if (R > mode) continue;
        const size_t steps = (CM + (R - 1) + 3) / 4 * 4;  // + the R - 1 steps that only the later periods' taps use
        const size_t NCX = 2 + steps;                      // c_lo <= 1 (the phases' offsets stay below P): columns up to TA - R + 1 + steps - 1
        const int ta = 64 * R;
        size_t sr = (ta + NCX + R - 1) / R + 1;
        if (!(sr & 1)) ++sr;
        const size_t bytes = (R * P * sr + (Q > 1 ? (size_t)ta * Q : 0)) * 16;
        if (bytes <= 48 * 1024) {
            pl.R = R;
            pl.TA = ta;
            pl.SR = (int)sr;
            pl.steps = (int)steps;
            pl.NCX = (int)NCX;
            pl.lds = bytes;
            break;
        }
    }
    return pl;
}
bool decim_poly_f64r4_supported(uint64_t P, uint64_t Q, size_t L) { return decim_r4_plan(P, Q, L).lds != 0; }

static int decim_geometry(size_t P, size_t Q, size_t NC, int *TA, int *S, size_t esz = 8) {
    int best = 0;
    for (int ta : {1024, 512, 256, 128, 64}) {
        int s = ta + (int)NC;
        if (!(s & 1)) ++s;  // odd row stride: the staging writes of neighbouring rows fall on different banks
        const size_t bytes = (P * (size_t)s + (Q > 1 ? (size_t)ta * Q : 0)) * esz;
        if (bytes <= kDecimLdsTarget || (ta == 64 && bytes <= kDecimLdsMax)) {
            *TA = ta;
            *S = s;
            best = (int)bytes;
            break;
        }
    }
    return best;
}

// k_decim_poly<true> (two periods per lane): two more staged columns, tiles of at least 128 periods, two zeros behind the image
static int decim_geometry_pair(size_t P, size_t Q, size_t NC, int *TA, int *S) {
    for (int ta : {1024, 512, 256, 128}) {
        int s = ta + (int)NC + 2;
        if (!(s & 1)) ++s;
        const size_t bytes = (P * (size_t)s + 2 + (Q > 1 ? (size_t)ta * Q : 0)) * 8;
        if (bytes <= kDecimLdsTarget) {
            *TA = ta;
            *S = s;
            return (int)bytes;
        }
    }
    return 0;
}
static bool decim_pair_enabled() {
    const char *e = std::getenv("RR_DECIM_PAIR");  // (=0: one period per lane, A/B runs and tests; read per call)
    return !(e && std::atoi(e) == 0);
}

bool decim_poly_supported(int dtype, uint64_t P, uint64_t Q, size_t L) {
    if ((dtype != RR_F32 && dtype != RR_F64) || P < 2 || P > 512 || Q < 1 || Q > 8 || Q >= P || L < 1) return false;
    int ta, s;
    return decim_geometry(P, Q, decim_nc(P, L), &ta, &s, dtype == RR_F64 ? 16 : 8) != 0;
}

// taps in the kernel's order: T[b][j] = { ir[j], byte offset of tap j's (row, column) }, (delta_b + j) = P column + row,
// delta_b = e[b] - e[0]; padded to a multiple of 8 taps per phase with { 0, 0 }
void build_decim_poly_taps(const std::vector<double> &ir, uint64_t P, uint64_t Q, const int64_t *e_first, std::vector<uint32_t> &T,
                           int *Lp_out, int dtype) {
    const size_t L = ir.size(), Lp = (L + 7) / 8 * 8;
    int ta = 0, S = 0;
    const bool f64 = dtype == RR_F64;
    if (f64) {
        const DecimR4Plan pl = decim_r4_plan(P, Q, L);
        if (pl.lds) {
            // k_decim_poly_f64r<R>: c_lo per (b, r) (Q P ints, padded to an even count), then `steps` f64 taps per (b, r)
            const size_t nh = (Q * P + 1) / 2 * 2, per = (size_t)pl.steps;
            T.assign(nh + 2 * Q * P * per, 0u);
            std::vector<double> te(Q * P * per, 0.0);
            for (uint64_t b = 0; b < Q; ++b) {
                const size_t d = (size_t)(e_first[b] - e_first[0]);
                for (size_t r = 0; r < P; ++r) {
                    // the row's taps: idx = d + j = P c + r, j < L
                    const size_t c_lo = d > r ? (d - r + P - 1) / P : 0;
                    T[b * P + r] = (uint32_t)c_lo;
                    for (size_t c = c_lo;; ++c) {
                        const size_t idx = P * c + r;
                        if (idx < d) continue;
                        const size_t j = idx - d;
                        if (j >= L) break;
                        te[(b * P + r) * per + (c - c_lo)] = ir[j];
                    }
                }
            }
            std::memcpy(T.data() + nh, te.data(), te.size() * 8);
            *Lp_out = -1;  // (the table is k_decim_poly_f64r4's)
            return;
        }
    }
    decim_geometry(P, Q, decim_nc(P, L), &ta, &S, f64 ? 16 : 8);
    T.assign((size_t)Q * Lp * (f64 ? 4 : 2), 0u);
    for (uint64_t b = 0; b < Q; ++b) {
        const size_t d = (size_t)(e_first[b] - e_first[0]);
        for (size_t j = 0; j < L; ++j) {
            const size_t idx = d + j, r = idx % P, c = idx / P;
            if (f64) {  // { f64 tap, byte offset, pad }
                uint32_t *e = &T[((size_t)b * Lp + j) * 4];
                std::memcpy(e, &ir[j], 8);
                e[2] = (uint32_t)((r * (size_t)S + c) * 16);
            } else {
                const float tap = (float)ir[j];
                uint32_t bits;
                std::memcpy(&bits, &tap, 4);
                T[((size_t)b * Lp + j) * 2] = bits;
                T[((size_t)b * Lp + j) * 2 + 1] = (uint32_t)((r * (size_t)S + c) * 8);
            }
        }
    }
    *Lp_out = (int)Lp;
    if (f64) return;
    // k_decim_poly<true>'s table behind it: per phase the entries { t[k - 1], t[k], t[k + 1], byte offset of (row, k) }, rows one
    // behind the other, k in steps of 2 with (row S + k) even from c_lo - 1 (or c_lo) to c_hi + 1; Lp2 entries per phase (a multiple
    // of 4, padded with zeros) travel in the upper half of *Lp_out
    int ta2 = 0, S2 = 0;
    if (!decim_geometry_pair(P, Q, decim_nc(P, L), &ta2, &S2) || Lp >= 65536) return;
    std::vector<std::vector<uint32_t>> ent(Q);
    size_t most = 0;
    for (uint64_t b = 0; b < Q; ++b) {
        const long d = (long)(e_first[b] - e_first[0]);
        for (long r = 0; r < (long)P; ++r) {
            // the row's taps: idx = d + j = P c + r, 0 <= j < L
            const long c_lo = d > r ? (d - r + (long)P - 1) / (long)P : 0;
            if (d + (long)L - 1 < r) continue;
            const long c_hi = (d + (long)L - 1 - r) / (long)P;
            if (c_lo > c_hi) continue;
            auto tap = [&](long c) -> uint32_t {
                float v = 0.f;
                if (c >= c_lo && c <= c_hi) v = (float)ir[(size_t)((long)P * c + r - d)];
                uint32_t bits;
                std::memcpy(&bits, &v, 4);
                return bits;
            };
            long k = c_lo - 1;
            if ((r * (long)S2 + k) & 1) ++k;
            for (; k <= c_hi + 1; k += 2) {
                ent[b].push_back(tap(k - 1));
                ent[b].push_back(tap(k));
                ent[b].push_back(tap(k + 1));
                ent[b].push_back((uint32_t)((r * (long)S2 + k) * 8));
            }
        }
        most = std::max(most, ent[b].size() / 4);
    }
    const size_t Lp2 = (most + 3) / 4 * 4;
    if (Lp2 == 0 || Lp2 >= 32768) return;
    const size_t base = T.size();  // (Q Lp 2 words: a multiple of 16 words)
    T.resize(base + Q * Lp2 * 4, 0u);
    for (uint64_t b = 0; b < Q; ++b) std::memcpy(T.data() + base + b * Lp2 * 4, ent[b].data(), ent[b].size() * 4);
    *Lp_out = (int)(Lp | (Lp2 << 16));
}

int launch_decim_poly(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *T,
                      uint64_t P, uint64_t Q, int Lp, size_t L, int64_t e_first0, void *out, size_t n_out, void *hist_out,
                      size_t hist_out_len, const void *nco, uint32_t denom, uint32_t idx0, int dtype) {
    // the history is written by the LAST tile's workgroup: a call without outputs has no tile, so a caller that
    // would flip to hist_out afterwards must not come here (rr_downsampler::process_dev: `produce &&`)
    if (n_out == 0) {
        if (hist_out) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: k_decim_poly cannot leave a history without producing outputs");
        return RR_OK;
    }
    if (dtype == RR_F64 && Lp == -1) {
        const DecimR4Plan pl = decim_r4_plan(P, Q, L);
        if (!pl.lds) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: the tap table is k_decim_poly_f64r4's, the tile is not");
        DecimArgsR4 a;
        a.nco = (const double2 *)nco;
        a.denom = nco ? denom : 0;
        a.idx0 = idx0;
        a.hist = (const double2 *)hist;
        a.hist_len = (int)hist_len;
        a.in = (const double2 *)in;
        a.n_in = (long)n_in;
        a.P = (int)P;
        a.Q = (int)Q;
        a.p_ref = (long)e_first0 - (long)(L - 1);
        a.out = (double2 *)out;
        a.n_out = (long)n_out;
        a.hist_out = (double2 *)hist_out;
        a.hist_out_len = (int)hist_out_len;
        a.TA = pl.TA;
        a.SR = pl.SR;
        a.steps = pl.steps;
        a.NCX = pl.NCX;
        a.CL = (const int *)T;
        a.TE = (const double *)((const uint32_t *)T + (Q * P + 1) / 2 * 2);
        const size_t per_tile = (size_t)a.TA * Q;
        const size_t ntiles = (n_out + per_tile - 1) / per_tile;
        if (ntiles > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: too many tiles");
        a.ntiles = (unsigned)ntiles;
        const unsigned grid = (unsigned)((ntiles + 63) / 64 * 64);
        if (pl.R == 8) hipLaunchKernelGGL(k_decim_poly_f64r<8>, dim3(grid), dim3(256), pl.lds, s, a);
        else hipLaunchKernelGGL(k_decim_poly_f64r<4>, dim3(grid), dim3(256), pl.lds, s, a);
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    if (dtype == RR_F64) {
        DecimArgs64 a;
        a.nco = (const double2 *)nco;
        a.denom = nco ? denom : 0;
        a.idx0 = idx0;
        a.hist = (const double2 *)hist;
        a.hist_len = (int)hist_len;
        a.in = (const double2 *)in;
        a.n_in = (long)n_in;
        a.P = (int)P;
        a.Q = (int)Q;
        a.NC = (int)decim_nc(P, L);
        a.Lp = Lp;
        a.p_ref = (long)e_first0 - (long)(L - 1);
        a.out = (double2 *)out;
        a.n_out = (long)n_out;
        a.hist_out = (double2 *)hist_out;
        a.hist_out_len = (int)hist_out_len;
        const int lds = decim_geometry(P, Q, (size_t)a.NC, &a.TA, &a.S, 16);
        if (!lds) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: %llu : %llu with %d tap columns does not fit the LDS tile",
                          (unsigned long long)P, (unsigned long long)Q, a.NC);
        const size_t per_tile = (size_t)a.TA * Q;
        const size_t ntiles = (n_out + per_tile - 1) / per_tile;
        if (ntiles > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: too many tiles");
        a.ntiles = (unsigned)ntiles;
        const unsigned grid = (unsigned)((ntiles + 63) / 64 * 64);
        hipLaunchKernelGGL(k_decim_poly_f64, dim3(grid), dim3(256), (size_t)lds, s, a, (const Tap64 *)T);
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    DecimArgs a;
    a.hist = (const float2 *)hist;
    a.hist_len = (int)hist_len;
    a.in = (const float2 *)in;
    a.n_in = (long)n_in;
    a.P = (int)P;
    a.Q = (int)Q;
    a.NC = (int)decim_nc(P, L);
    const int Lp2 = Lp >> 16;  // (build_decim_poly_taps: the two-periods-per-lane table behind the first one, if its tile fits)
    Lp &= 0xffff;
    a.Lp = Lp;
    a.Lp2 = Lp2;
    a.p_ref = (long)e_first0 - (long)(L - 1);
    a.out = (float2 *)out;
    a.n_out = (long)n_out;
    a.hist_out = (float2 *)hist_out;
    a.hist_out_len = (int)hist_out_len;
    a.nco = (const float2 *)nco;
    a.denom = nco ? denom : 0;
    a.idx0 = idx0;
    a.ph_ref = a.ph_tile_step = 0;
    a.inv_denom = 0.0;
    int lds = 0;
    const bool pair = Lp2 > 0 && decim_pair_enabled() && (lds = decim_geometry_pair(P, Q, (size_t)a.NC, &a.TA, &a.S)) != 0;
    if (pair) a.NC += 2;
    else lds = decim_geometry(P, Q, (size_t)a.NC, &a.TA, &a.S);
    if (!lds) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: %llu : %llu with %d tap columns does not fit the LDS tile",
                      (unsigned long long)P, (unsigned long long)Q, a.NC);
    const size_t per_tile = (size_t)a.TA * Q;
    const size_t ntiles = (n_out + per_tile - 1) / per_tile;
    if (ntiles > 0x7ffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: too many tiles");
    a.ntiles = (unsigned)ntiles;
    if (a.denom) {
        const int64_t den = (int64_t)a.denom;
        int64_t r = ((int64_t)a.idx0 + a.p_ref) % den;
        if (r < 0) r += den;
        a.ph_ref = (unsigned)r;
        a.ph_tile_step = (unsigned)(((int64_t)P * a.TA) % den);
        a.inv_denom = 1.0 / (double)den;
    }
    const unsigned grid = (unsigned)((ntiles + 63) / 64 * 64);
    const uint4 *T2 = reinterpret_cast<const uint4 *>(static_cast<const uint32_t *>(T) + (size_t)Q * Lp * 2);
    if (pair) hipLaunchKernelGGL(k_decim_poly<true>, dim3(grid), dim3(256), (size_t)lds, s, a, (const uint2 *)T, T2);
    else hipLaunchKernelGGL(k_decim_poly<false>, dim3(grid), dim3(256), (size_t)lds, s, a, (const uint2 *)T, T2);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

}  // namespace rr
