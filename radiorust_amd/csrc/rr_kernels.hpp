// rr_kernels.hpp — launchers of the gfx950 kernels (rr_kernels.hip).
// All launchers are asynchronous on `stream` and return an rr_status.
#pragma once
#include <vector>
#include "rr_internal.hpp"

namespace rr {

// LDS bytes a generic FIR tile may use for samples
constexpr size_t kFirLdsBytes = 96 * 1024;

// out[i] = in[i] * table[(idx0 + i) % denom]   (transform.rs:341-348)
int launch_freqshift(int dtype, hipStream_t s, const void *in, void *out, size_t n, const void *table,
                     uint32_t denom, uint32_t idx0);

// Generic gather-FIR over the virtual stream  [ zeros | hist (hist_len) | in (n_in) ]:
//   out[m] = sum_{j<K} w[j] * x[e_m - (K-1) + j],
//   e_m = e0 + m*D (emit == nullptr) or emit[m]   (indices relative to in[0])
// complex_taps: w is K complex values (Filter, direct form), else K real values
// (Downsampler, resampling.rs:112-120, oldest sample times ir[0]).
// opt-in for more than 64 KiB of dynamic LDS, once per (kernel, device) (rr_kernels.hip)
int dyn_lds_optin(const void *fn, size_t bytes);

struct FirArgs {
    const void *hist = nullptr;
    size_t hist_len = 0;
    const void *in = nullptr;
    size_t n_in = 0;
    const void *taps = nullptr;
    uint32_t K = 0;
    bool complex_taps = false;
    void *out = nullptr;
    size_t n_out = 0;
    uint64_t e0 = 0;
    uint32_t D = 1;
    const uint32_t *emit = nullptr;
    uint32_t max_step = 1;  // upper bound of e_{m+1} - e_m (list mode)
    // period_q != 0: `emit` holds the first period_q positions of a periodic schedule, e_m = (m / period_q) period_p + emit[m mod period_q]
    // (no list of the call's length, no host loop over its samples)
    uint32_t period_p = 0, period_q = 0;
};
int launch_fir(int dtype, hipStream_t s, const FirArgs &a);

// Filter by overlap-save fast convolution, exactly the reference's per-chunk recipe
// (filters.rs:240-259): out_c = IFFT_2n(FFT_2n([chunk c-1 | chunk c]) * H)[0..n), n a power
// of two.  H: 2n complex (transform of [0_n | h], h = g / 2n); tw: n entries e^{-j 2 pi k / 2n}.
// first_chunk = 0: chunk -1 is `hist` (n samples); first_chunk = 1: chunk 0 only provides history.
bool ols_supported(int dtype, size_t n);
int launch_filter_ols(int dtype, hipStream_t s, const void *hist, const void *in, size_t n, size_t nchunks,
                      int first_chunk, const void *H, const void *tw, void *out);

// Filter fast convolution with 4096-point blocks (rr_filter_ols.hip), Complex<f32>, any n <= 2048 (V = n; longer
// responses: one launch per partition of 2048 taps, e0 moved back by 2048 p, accumulate = true from the second on): out[m] = sum_k g[k] x[e0 + m - k] over [ hist | in ]; G = DFT_4096(g)/4096 pair-interleaved
// (f32, or f16 with g_f16), tw4096[k] = e^{-j 2 pi k / 4096}; out_f16: outputs as {half re, half im}.
// hist_out (may be null) receives the last hist_out_len samples of [ hist | in ] - the next call's history -
// written by the kernel itself (only when n_out > 0, i.e. when a kernel is launched).
bool filter_ols4096_supported(int dtype, size_t n);
// k_filter_blkbig<N>: blocks of N = 8192 / 16384 points in LDS (responses of up to N / 2 + 1 taps in one forward and one inverse
// transform per block); G = DFT_N(g) / N pair-interleaved {G[j + 2 T kp], G[j + 2 T kp + T]}, T = N / 16, twN = e^{-j 2 pi k / N}
bool filter_blkbig_supported(int dtype, size_t n);
int launch_filter_blkbig(hipStream_t s, size_t N, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G,
                         const void *twN, size_t V, void *out, size_t n_out, long e0, void *hist_out, size_t hist_out_len);
int launch_filter_blk4096(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G,
                          const void *tw4096, size_t n, void *out, size_t n_out, long e0, bool out_f16, bool g_f16,
                          void *hist_out, size_t hist_out_len, bool accumulate = false, size_t nparts = 1);

// Downsampler for any integer ratio P : 1 and rational ratios P : Q with Q <= 8 (rr_decim.hip), Complex<f32>:
// out[Q a + b] = sum_j ir[j] x[e_first[b] + P a - (L - 1) + j] over [ hist | in ]; T = build_decim_poly_taps' table.
// hist_out (may be null) receives the last hist_out_len samples of [ hist | in ].  nco != null: the FreqShifter fused in
// front - in[pos] * nco[(idx0 + pos) mod denom] -, hist / hist_out then hold mixed samples.
bool decim_poly_supported(int dtype, uint64_t P, uint64_t Q, size_t L);
void build_decim_poly_taps(const std::vector<double> &ir, uint64_t P, uint64_t Q, const int64_t *e_first, std::vector<uint32_t> &T,
                           int *Lp_out, int dtype = RR_F32);
int launch_decim_poly(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *T,
                      uint64_t P, uint64_t Q, int Lp, size_t L, int64_t e_first0, void *out, size_t n_out, void *hist_out,
                      size_t hist_out_len, const void *nco = nullptr, uint32_t denom = 0, uint32_t idx0 = 0, int dtype = RR_F32);

// new_hist (H samples) = last H samples of [ zeros | old_hist (H) | in (n_in) ]
int launch_update_hist(int dtype, hipStream_t s, const void *old_hist, void *new_hist, size_t H, const void *in,
                       size_t n_in);

// Windowed forward DFT of `count` consecutive chunks of n samples each
// (analysis.rs:105-115).  twiddle = e^{-j 2 pi k / n}: n/2 entries for power-of-two
// n (LDS Stockham kernel), n entries otherwise (direct DFT kernel).
int fourier_supported(int dtype, size_t n);
// true: LDS Stockham kernel (twiddle table of n/2 entries); false: direct DFT (n entries)
bool fourier_pow2_path(int dtype, size_t n);
int launch_fourier(int dtype, hipStream_t s, const void *in, void *out, size_t n, size_t count, const void *window,
                   const void *twiddle, bool center_dc);
// the same over frames cut every `hop` samples from the stream [ head (n_head samples) | in ]
// (hop < n: overlapping chunks; power-of-two n only)
int launch_fourier_overlapped(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, void *out,
                              size_t n, size_t hop, size_t count, const void *window, const void *twiddle, bool center_dc);

// ---- fused fast path (rr_ols.hip, rr_fft_regs.hip), Complex<f32> only ------------------------
// v[m] = sum_i c[i] xs[e0 + D m - i]; xs = NCO-mixed input.  Virtual stream:
// positions [-hx, 0) come from `xh` (already mixed), [0, n_in) from `in` (raw,
// mixed on load with nco[(idx0 + pos) mod denom]).  `taps`: Gp*D floats in step
// order (see build_combined_taps).
struct FusedFirArgs {
    const void *xh = nullptr;
    size_t hx = 0;
    const void *in = nullptr;
    size_t n_in = 0;
    const void *nco = nullptr;
    uint32_t denom = 1, idx0 = 0;
    const void *taps = nullptr;
    int Gp = 0;
    void *out = nullptr;
    size_t n_out = 0;
    int64_t e0 = 0;
    uint32_t D = 1;
    void *xh_out = nullptr;  // receives the last hx mixed samples of this call (may be null)
    // overlap-save variant (k_ols_decim4)
    const void *H = nullptr;       // DFT_4096(c) / 4096, complex f32
    const void *tw4096 = nullptr;  // e^{-j 2 pi k / 4096}
    int V = 0;                     // overlap (samples), multiple of 256
    bool poly = false;             // k_ols_wave<4>: H holds the polyphase tables G_p (build_fused_fir_tables)
    int blk = 1024;                // k_ols_wave: samples per block; 2048 = k_ols_wave2k (8 : 1, rr_ols_wave2k.hip); 256 D = k_ols_wg (16 / 32 / 64 : 1)
    // k_ols_frame only: the NCO's period divides 8 and H holds the tables with the mixer folded in - the kernel transforms the
    // samples as they are and multiplies its results by nco[ph0] sigma^(index): see rr_chain::ensure_mixfold
    bool mixfold = false;
    float sigma = 1.f;
    // k_ols_frame only: ANY NCO period with the mixer moved behind the filter - H holds the tables of the response c[i] w^-i
    // (rr_chain::ensure_genfold), the kernel transforms the samples as they are and multiplies its results by the phase table's
    // entries at their positions
    bool genfold = false;
    // optional: the launch itself records its start / end in these events (hipExtLaunchKernel):
    // kernel-only timing without marker packets on the stream
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};
bool ols_decim_supported(uint64_t D, size_t Lc);
int ols_decim_overlap(size_t Lc);
int launch_ols_decim(hipStream_t s, const FusedFirArgs &a);
// one-wave-per-block variant (k_ols_wave): H = DFT_1024(c) / 1024 and tw = e^{-j 2 pi k / 1024}
// go in the H / tw4096 fields, V = ols_wave_overlap(Lc)
// k_filter_wave: the Filter alone, one wave per 1024-sample block (n - 1 <= 384, f32); H / tw as for k_ols_wave
bool filter_wave_supported(int dtype, size_t n);
// k_filter_wave<true> (the Downsampler for any periodic schedule): the schedule's constants, host side of launch_decim_select
struct SelectArgs {
    uint32_t ra = 0, rb = 0;       // input / output rate (integers, rb < ra < 2^31)
    uint32_t kr = 0, kq = 0;       // 128 rb = kq ra + kr
    uint32_t hr = 0, hq = 0;       // hop rb = hq ra + hr
    uint32_t base_r = 0, base_q = 0;  // pos + V (ra - rb) = base_q ra + base_r
    double inv_ra = 0.0;
};
bool decim_select_supported(int dtype, uint64_t ra, uint64_t rb, size_t L);
int launch_decim_select(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *H, const void *tw,
                        int V, void *out, size_t n_out, uint64_t ra, uint64_t rb, uint64_t pos);
// the same through k_filter_blk4096<.., SEL> for responses of 386 .. 2048 taps (G: the Filter's 4096-point table of c = reverse(ir))
bool decim_select_blk_supported(int dtype, uint64_t ra, uint64_t rb, size_t L);
int launch_decim_select_blk(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G, const void *tw4096,
                            size_t L, void *out, size_t n_out, uint64_t ra, uint64_t rb, uint64_t pos);
int launch_filter_wave(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *H,
                       const void *tw, int V, void *out, size_t n_out, long e0);
bool ols_wave_supported(uint64_t D, size_t Lc);
int ols_wave_overlap(size_t Lc, size_t granule = 16);
int launch_ols_wave(hipStream_t s, const FusedFirArgs &a);
// k_ols_frame: the same stage + the 4096-point Fourier stage in one kernel (a workgroup per frame of the
// decimated stream; H / tw4096 fields as for launch_ols_wave).  pend_in: pl decimated samples pending from
// the previous call; pend_out receives the (pl + n_out) mod 4096 left over; spectra: (pl + n_out) / 4096 frames.
// One channel of a bank (rr_chainbank) for the two kernels of its lockstep step: the channel's own buffers.
struct BankPtrs {
    const void *xh;    // mixed-sample history in front of this call
    const void *in;    // the call's samples
    void *dec;         // decimated samples of this call (k_ols_wave's output, k_fft4096's input)
    void *xh_out;      // history for the next call
    const void *head;  // decimated samples pending from the previous call (n_head of them)
    void *out;         // spectra
};
// The channels' buffers travel as a KERNEL ARGUMENT (64 channels x 48 bytes = 3 KiB of the 4 KiB an argument block may have):
// no table in device memory, no copy in front of the launch; banks of more than 64 channels take a launch per 64.
constexpr size_t kBankGroup = 64;
struct BankTable {
    BankPtrs c[kBankGroup];
};
// k_ols_wave / k_fft4096 for `channels` <= kBankGroup streams in ONE launch each: the shared launch parameters in `a` (its
// xh / in / out / xh_out are ignored), the per-channel buffers in `tab`
int launch_ols_wave_bank(hipStream_t s, const FusedFirArgs &a, const BankTable &tab, size_t channels);
// BankPtrs::head[0 .. n) -> BankPtrs::out for every channel in one launch (the pending chunks in front of a step without a frame)
int launch_bank_copy(hipStream_t s, const BankTable &tab, size_t channels, size_t n);
int launch_fft4096_bank(hipStream_t s, const BankTable &tab, size_t channels, size_t n_head, size_t count, const void *window,
                        const void *tw4096, bool center_dc);
// k_ols_wave2k: 8 : 1 with a wave per 2048-sample block (combined responses up to 1025 taps); tables: build_fused_fir_tables (blk = 2048)
bool ols_wave2k_supported(uint64_t D, size_t Lc);
int launch_ols_wave2k(hipStream_t s, const FusedFirArgs &a);
int launch_ols_wave2k_bank(hipStream_t s, const FusedFirArgs &a, const BankTable &tab, size_t channels);
// k_ols_wg: even ratios 10 .. 64 with a workgroup of ~D / 4 waves per block of 256 D samples (combined responses up to 128 D + 1 taps);
// tables: build_fused_fir_tables (blk = 256 D), V = ols_wg_overlap
bool ols_wg_supported(uint64_t D, size_t Lc);
int ols_wg_overlap(uint64_t D, size_t Lc);
int ols_wg_runs(uint64_t D);  // the runs of four phases the tables hold (= the kernel's waves)
int launch_ols_wg(hipStream_t s, const FusedFirArgs &a);
bool ols_frame_supported(uint64_t D, size_t Lc, size_t fft_len);
// What a metered kernel needs beside its transform's arguments: metering::bandwidth (metering.rs:41-80) per spectrum, computed
// behind the transform while the bins are in registers (rr_meter_dev.hpp), and the spectrum's energy sum |X|^2.
struct FrameMeter {
    double double_percentile = 0.0, sample_rate = 0.0;
    double *bw = nullptr;      // one value per frame
    double *energy = nullptr;  // sum |X|^2 per frame (may be null)
    int store = 1;             // 0: the spectra are not written at all (the caller only wants the figures)
};
int launch_ols_frame(hipStream_t s, const FusedFirArgs &a, const void *pend_in, size_t pl, void *pend_out, void *spectra,
                     const void *window, const void *tw4096, bool center_dc, const FrameMeter *fm = nullptr, size_t fft_len = 4096);
bool fused_fir_supported(uint64_t D, size_t Lc);
int fused_fir_R(uint64_t D);  // outputs per lane of the instantiation for D
int launch_fused_fir(hipStream_t s, const FusedFirArgs &a);
// 4096-point windowed forward DFT (radix 16 x 3); tw4096[k] = e^{-j 2 pi k / 4096}, 4096 entries
// frames are cut from the stream [ head (n_head samples) | in ]
// Bluestein's elementwise stages (rr_kernels.hip), f32: frames <= 65535 per launch
int launch_bs_pre(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, size_t M,
                  const void *c, void *ws, size_t frames);
int launch_bs_mul(int dtype, hipStream_t s, void *ws, const void *B, size_t M, size_t frames);
int launch_bs_post(int dtype, hipStream_t s, const void *ws, const void *w, size_t n, size_t M, void *out, bool center_dc,
                   size_t frames);
// power-of-two transforms beyond one LDS tile (up to 2^24 points): four-step through a workspace of count * n elements;
// tw1 / tw2 = e^{-j 2 pi k / N1}, e^{-j 2 pi k / N2} (half tables) for the split of fft_big_split
// tiled transpose of `count` R x C matrices with a factor on the way: mode 0 none (rows of the result rotated by rot_rows),
// 1 window[r C + c] (real), 2 the four-step twiddle W_(R C)^(r c) = tA[e >> h] tB[e & (2^h - 1)]
int launch_transpose_mul(int dtype, hipStream_t s, const void *in, void *out, size_t R, size_t C, size_t count, int mode,
                         const void *window, const void *tB, const void *tA, int h, size_t rot_rows);
// chunk lengths 2^a 3^b 5^c (<= 8192 in f32, <= 4096 in f64) that are not powers of two: mixed-radix passes in one LDS image (k_fft_mixed); window = n reals,
// tw = e^{-j 2 pi k / n} (n entries); frames from [ head | in ] at any hop
bool fft_mixed_supported(int dtype, size_t n);
int fft_mixed_radices(int dtype, size_t n, unsigned char *radices, int cap);
// the same with the frames starting base0 + f hop samples into `in` and each frame the fold of `branches` windowed chunks
// (window: branches * n values) - the polyphase channelizer with a bin count that is not a power of two, in one kernel
int launch_fft_mixed_fold(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, long base0, size_t hop,
                          size_t n, size_t branches, const void *window, const void *tw, void *out, bool center_dc, size_t count);
bool fft_mixed_preferred(int dtype, size_t n);  // measured crossover against the Bluestein kernels
int launch_fft_mixed(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n,
                     const void *window, const void *tw, void *out, bool center_dc, size_t count);
// lengths 2^a 3^b 5^c beyond one LDS image, up to 512 x 512: two passes (k_fft_tilem).  twNp = the pass's sub-transform table;
// T1[i] = W_N^(C i) (N1 ceil(N2 / C) entries), T2[i] = W_N^i (N1 C entries), C = 16 (f32) / 8 (f64) columns per bundle
bool fft_tilem_split(int dtype, size_t n, size_t *N1, size_t *N2);
int launch_fft_tilem(int dtype, hipStream_t s, int pass, const void *head, size_t n_head, const void *in, size_t hop, void *out,
                     size_t N1, size_t N2, size_t count, const void *window, const void *twNp, const void *T1, const void *T2,
                     size_t rot);
// two-pass four-step (k_fft_tile): pass 0 = window, column transforms over n1, twiddle; pass 1 = row transforms over n2 with
// the transposed store; twNp = e^{-j 2 pi k / Np} (Np entries) of the pass's sub-transform
bool fft_tile_supported(int dtype, size_t N1, size_t N2);
int launch_fft_tile(int dtype, hipStream_t s, int pass, const void *in, void *out, size_t N1, size_t N2, size_t count,
                    const void *window, const void *twNp, const void *tB, const void *tA, int h, size_t rot);
// the same passes with Bluestein's element-wise stages folded in: stage 0 = pass A with x * table (c; zero beyond n; frames from
// [head | in] at a hop), 1 = pass B storing conj(X * table) (B), 2 = pass A plain, 3 = pass B storing the first n bins of
// conj(X * table) (the chirp), rotated right by rot elements
int launch_fft_tile_bs(int dtype, hipStream_t s, int stage, const void *head, size_t n_head, const void *in, size_t hop, void *out,
                       size_t N1, size_t N2, size_t count, size_t n, const void *table, const void *twNp, const void *tB,
                       const void *tA, int h, size_t rot, long in_limit = 0, long out_limit = 0);
bool fft_big_supported(size_t n);
void fft_big_split(size_t n, size_t *N1, size_t *N2);
int launch_fft_big(int dtype, hipStream_t s, const void *in, void *out, void *ws, size_t n, size_t count, const void *window,
                   const void *tw1, const void *tw2, bool center_dc);
// k_ols4096_f64 (rr_f64.hip): Complex<f64> overlap-save in blocks of 4096 points - the Filter (D = 1), integer-ratio Downsamplers
// and the f64 chain's front end; G = DFT_4096(c) / 4096 in natural order, V from ols4096_f64_overlap (0: the response does not fit)
size_t ols4096_f64_overlap(size_t Lc, uint64_t D);
int launch_ols4096_f64(hipStream_t s, const void *hist, size_t hist_len, const void *in, size_t n_in, const void *G, const void *tw4096,
                       size_t V, uint64_t D, void *out, size_t n_out, long e0, void *hist_out, size_t hist_out_len, const void *nco,
                       uint32_t denom, uint32_t idx0);
// k_fft16384: 1024 lanes per 16384-sample frame (rr_fft_big.hpp), plain window and twiddle tables
int launch_fft16384(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count, const void *window,
                    const void *tw16384, bool center_dc, size_t hop);
int launch_fft8192_big(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count, const void *window,
                       const void *tw8192, bool center_dc, size_t hop);
// k_fft8192: 256 lanes per 8192-sample frame, 32 values per lane (plain window and twiddle tables)
int launch_fft8192(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw8192, bool center_dc, size_t hop);
// k_fft64 / k_fft128: 8 / 4 frames per wave, frames side by side (n = 64 or 128)
int launch_fft_small(hipStream_t s, const void *in, void *out, size_t n, size_t count, const void *window, const void *tw,
                     bool center_dc);
// k_fft512: a wave per 512-sample frame (plain window and twiddle tables)
int launch_fft512(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                  const void *window, const void *tw512, bool center_dc, size_t hop);
// k_fft2048: 128 lanes per 2048-sample frame; the window table carries the packed copy wp[16 t + k] = w[t + 128 k] behind its 2048 entries
int launch_fft2048(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw2048, bool center_dc, size_t hop);
// k_fft1024: a wave per 1024-sample frame; tw1024 = e^{-j 2 pi k / 1024} followed by the lane seeds (append_wave1024_seeds)
int launch_fft1024(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw1024, bool center_dc, size_t hop);
void append_wave1024_seeds(std::vector<float> &twb);
// k_bluestein4096: chunk lengths 513 .. 2048 that are not powers of two, one kernel per call (tables of rr_fourier::prepare
// for M = 4096: c[n] = window conj(chirp), B[4096] = DFT(chirp) / 4096, w[n] = chirp; tw4096 as for k_fft4096)
// Bluestein in one kernel with both transforms as Stockham passes between two LDS images (M <= 8192 in f32, <= 4096 in f64)
bool bluestein_lds_supported(int dtype, size_t n, size_t M);
int launch_bluestein_lds(int dtype, hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, size_t M,
                         const void *c, const void *B, const void *w, const void *tw, void *out, bool center_dc, size_t count);
// k_bluestein_big<M>: 4097 .. 8192 points in f32 (M = 16 384) and 2049 .. 4096 (M = 8192): one kernel around two
// transforms of rr_fft_big.hpp; B in k_filter_blkbig's pair layout, twM = W_M^i, i < M
bool bluestein_big_supported(int dtype, size_t n, size_t *M);
int launch_bluestein_big(hipStream_t s, size_t M, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                         const void *Bp, const void *w, const void *twM, void *out, bool center_dc, size_t count);
// k_bluestein8192: 2049 .. 4096 points in f32, one kernel around two 8192-point register transforms
bool bluestein8192_supported(int dtype, size_t n);
int launch_bluestein8192(hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                         const void *B, const void *w, const void *tw8192, void *out, bool center_dc, size_t count);
bool bluestein4096_supported(int dtype, size_t n);
int launch_bluestein4096(hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                         const void *B, const void *w, const void *tw4096, void *out, bool center_dc, size_t count);
// k_bluestein1024: chunk lengths 32 .. 512 that are not powers of two, a wave per chunk (tables for M = 1024: c[n (+1)],
// Bp[1024] pair-interleaved {B[l + 128 kp], B[l + 128 kp + 64]} at [kp][l], w[n]; tw1024 with the lane seeds)
bool bluestein1024_supported(int dtype, size_t n);
int launch_bluestein1024(hipStream_t s, const void *head, size_t n_head, const void *in, size_t hop, size_t n, const void *c,
                         const void *Bp, const void *w, const void *tw1024, void *out, bool center_dc, size_t count);
// polyphase channelizers with 512 / 1024 / 2048 / 4096 bins: the Fourier kernels with the fold of `branches` windowed chunks at the load
// (window: bins * branches plain values; tw1024 with the lane seeds / tw4096: at least 256 entries)
int launch_chan1024(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                    const void *window, const void *tw1024, size_t hop, size_t branches);
int launch_chan512(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw512, size_t hop, size_t branches);
int launch_chan2048(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                    const void *window, const void *tw2048, size_t hop, size_t branches);
int launch_chan4096(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                    const void *window, const void *tw4096, size_t hop, size_t branches);
// k_stft4096: runs of overlapping 4096-point frames (hop 256, 512, 1024 or 2048), the sliding window in registers
bool stft4096_supported(size_t hop);
int launch_stft4096(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                    const void *window, const void *tw4096, bool center_dc, size_t hop, const FrameMeter *fm = nullptr);
int launch_fft4096(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count,
                   const void *window, const void *tw4096, bool center_dc, size_t hop = 4096,
                   hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const FrameMeter *fm = nullptr);
int launch_drop_tail(hipStream_t s, const void *oldh, void *newh, size_t H, size_t drop);
// Complex<f64>, 4096 points, the lane's values in registers (rr_f64.hip); tw4096[k] = e^{-j 2 pi k / 4096} as double2
int launch_fft4096_f64(hipStream_t s, const void *head, size_t n_head, const void *in, void *out, size_t count, const void *window,
                       const void *tw4096, bool center_dc, size_t hop);
// metering (rr_metering.hip): the serial kernel (the reference's summation order, bit-equal to the oracle; mode 0 level,
// 1 bandwidth, 2 sum of energies) and the parallel scan for frames of any length
int launch_meter(int dtype, hipStream_t s, int mode, double double_percentile, double sample_rate, const void *frames,
                 size_t n, size_t count, double *out);
int launch_bandwidth_par(int dtype, hipStream_t s, double double_percentile, double sample_rate, const void *frames, size_t n,
                         size_t count, double *bw_out, double *energy_out);

// Polyphase channelizer: frame f = FFT_M( sum_{p<P} w[r + M p] x[M (f0 + f) + r + M p] ), r < M,
// over the virtual stream [ hist (hist_len samples, ends right before in[0]) | in ].
// base0 = index (relative to in[0], may be negative) of the first sample of frame 0.
int launch_channelizer(int dtype, hipStream_t s, const void *hist, size_t hist_len, const void *in, long base0,
                       size_t M, size_t P, size_t nframes, const void *window, const void *tw, void *out, size_t hop = 0);
// (hop = 0: M; hop < M only where channelizer_fused_supported says so)
bool channelizer_fused_supported(int dtype, size_t M, size_t P, size_t hop);
// the fold alone for any M and any hop (frames written to `out`, M values each); the transforms follow through rr_fourier
int launch_chan_fold(int dtype, hipStream_t s, const void *hist, size_t hist_len, const void *in, long base0, size_t hop,
                     size_t M, size_t P, size_t frames, const void *window, void *out);
// f32, M = 256, P in {1, 2, 3, 4, 6, 8} at hop 256 or P in {2, 4, 8} at hop 128 / 64 (the oversampled filterbanks): one wave
// per run of frames, sliding window of samples in registers, radix-4 DFT_256 with wave-local exchanges (rr_channelizer.hip)
bool channelizer256_supported(int dtype, size_t M, size_t P, size_t hop = 256);
int launch_channelizer256(hipStream_t s, const void *hist, size_t hist_len, const void *in, long base0, size_t P,
                          size_t nframes, const void *window, const void *tw, void *out, size_t hop = 256);

// Upsampler (resampling.rs:237-267) as a gather: out[m] = sum over the inputs t with
// 0 <= m - before[t] < L, ascending t, of x[t] * ir[m - before[t]], rounded like the reference's
// `ringbuf[i] += sample * ir` (product, then sum).  The virtual input stream is [hist (hn) | in];
// integer ratio: before[t] = U * t, `before` may be null; otherwise before[] has hn + n_in entries.
// (rr_metering.hip: built without a*b+c contraction)
// k_upsample_closed: any pair of rates on a 2^-s grid below 2^31 (UpSchedule::closed), no list of the call's length
int launch_upsample_closed(int dtype, hipStream_t s, const void *hist, size_t hn, const void *in, const void *ir, size_t L,
                           uint64_t ra, uint64_t rb, uint64_t pos0, void *out, size_t n_out);
int launch_upsample(int dtype, hipStream_t s, const void *hist, size_t hn, const void *in, size_t n_in,
                    const void *ir, size_t L, uint64_t U, const int32_t *before, void *out, size_t n_out);
// FmDemod (modulation.rs:121-130): out[t] = (arg(x[t] * conj(x[t-1])) * factor, 0); st_in / st_out:
// {previous sample, last output}; without a previous sample the first output repeats the last one.
int launch_fmdemod(int dtype, hipStream_t s, const void *in, size_t n, void *out, const void *st_in, void *st_out,
                   int have_prev, double factor, double gain = 1.0);

// SURVEY §8(d) synthetic IQ, f32
int launch_synth(hipStream_t s, uint64_t seed, uint64_t t0, size_t n, void *out);

}  // namespace rr
