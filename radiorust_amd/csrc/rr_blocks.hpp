// rr_blocks.hpp — handle structs = the per-task state of one reference block.
#pragma once
#include "rr_internal.hpp"
#include "rr_kernels.hpp"

// FreqShifter — transform.rs:306-311 (phase_vec, phase_idx, prev_sample_rate)
struct rr_freqshifter : rr_block {
    double precision = 1.0;
    double shift = 0.0;
    bool shift_changed = false;  // watch::Receiver::has_changed()
    bool have_rate = false;
    double prev_rate = 0.0;
    int64_t numer = 0, denom = 0;
    uint64_t phase_idx = 0;
    uint64_t table_version = 0;             // counts the recalculations of the table
    std::vector<unsigned char> host_table;  // denom complex<T>
    rr::DevBuf d_table;
    int prepare(double sample_rate);  // the `if recalculate {..}` body
    int process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
};

// Filter — filters.rs:161-170 (previous_chunk, extended_response, ...)
struct rr_filter : rr_block {
    // A GainControl behind the Filter folded into the response: the taps of every kernel's tables are gain_flt * g
    // (results differ from `sample * gain` by rounding only); set_gain() rebuilds the tables and keeps the history.
    double gain = 1.0;
    std::vector<rr::cd> taps_base;  // g[k] of the design, without the gain
    double design_rate = 0.0;
    int build_tables(bool reset_history);
    int set_gain(double g);
    // Long responses (f32: n > 2048, f64: n >= 4096, up to 2^17 taps) at n log n cost: overlap-save with blocks of conv_N = 2^14 .. 2^18
    // points (>= 4 n where that fits), each block's transform pair as the four passes of the two-pass tile transform (k_fft_tile, as
    // Bluestein beyond one LDS image): load + pass A, pass B x G, pass A, pass B + cut to the block's valid part.  The response's
    // spectrum carries the shift that moves the valid part to the front of the block.
    bool use_conv = false;
    size_t conv_N = 0;
    struct rr_fourier *conv_fft = nullptr;  // rectangular-window transform of conv_N points: the tile passes' tables
    rr::DevBuf d_convG, d_ones, conv_ws[2];
    int process_conv(const void *d_in, size_t n_in, void *d_out, size_t produce);
    ~rr_filter() override;
    bool designed = false;
    bool params_changed = false;
    double rate = 0.0;
    size_t n = 0;
    bool real_taps = false;
    std::vector<rr::cd> taps_f64;  // g[k], causal order
    rr::DevBuf d_taps;             // w[j] = g[n-1-j] as T or complex<T>
    bool use_ols = false;          // long power-of-two filters: overlap-save fast convolution
    rr::DevBuf d_H, d_olstw;       // H = FFT_2n([0 | g / 2n]) and e^{-j 2 pi k / 2n}, k < n
    bool use_ols4096 = false;      // f32, n = 129 .. 2048: 4096-point blocks, radix-16 kernel
    size_t npart = 0;              // f32, n > 2048: that kernel once per partition of 2048 taps, accumulating
    bool big_ols4096 = false;      // f32, n in {64, 128}: the same for calls of >= 16384 outputs
    bool use_ols64 = false;        // f64, n = 2 .. 2049: k_ols4096_f64 (rr_f64.hip), for calls of >= 4096 outputs
    rr::DevBuf d_G64, d_tw64;
    size_t V64 = 0;
    bool use_ols16k = false;       // f32, n = 2049 .. 8192: k_filter_blkbig<N> (blocks of 8192 / 16 384 points in LDS)
    rr::DevBuf d_G16k, d_tw16k;
    size_t V16k = 0, N16k = 0;
    bool use_wave = false;         // f32, n <= 385: k_filter_wave (a wave per 1024-block) for calls of >= 16384 outputs
    rr::DevBuf d_Hw, d_tww;        // its tables (DFT_1024(g) / 1024 pair-interleaved; twiddles + lane seeds)
    int wave_V = 0;
    int last_kernel = 0;           // 0 k_fir, 1 k_filter_ols (2n-point), 2 k_filter_blk4096, 3 k_filter_wave, 4 tile-transform blocks, 5 k_filter_blk16k
    rr::DevBuf d_G4096, d_tw4096;
    rr::DevBuf d_G4096h;           // the same table rounded to IEEE half (rr_filter_process_dev_f16's option)
    rr::DevBuf hist[2];            // previous_chunk (n samples), ping-pong
    int cur = 0;
    bool hist_valid = false;  // previous_chunk.is_some()
    uint64_t design_version = 0;
    bool needs_design(double sample_rate, size_t len) const {
        return !designed || params_changed || sample_rate != rate || len != n;
    }
    int design(double sample_rate, size_t len, const rr_c64 *resp, const double *window_rel);
    size_t peek(size_t n_in) const { return hist_valid ? n_in : (n_in >= n ? n_in - n : 0); }
    int process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out,
                    bool out_f16 = false, bool g_f16 = false);
};

// Downsampler — resampling.rs:62-67 (ir, ringbuf, ringbuf_pos, pos)
// tables of the fused decimating-FIR kernels (rr_api.hip: build_fused_fir_tables)
struct FusedFirTables {
    int kind = 0;
    std::vector<float> ctaps, H, tw;
    int Gp = 0, V = 0, N = 0;
    int blk = 1024;     // the wave kernels' block: 2048 = k_ols_wave2k's tables (8 : 1)
    bool poly = false;  // H = the polyphase tables of k_ols_wave<4, POLY>
    std::vector<rr::cd> G64;  // D = 4, polyphase: G_p[k] at 256 p + k in f64 (for the variant with the mixer folded in)
};
void build_fused_fir_tables(int kind, uint64_t D, const std::vector<double> &c, const std::vector<rr::cd> &cc, FusedFirTables &t);

struct rr_downsampler : rr_block {
    double output_rate = 0, bandwidth = 0, quality = 3.0;
    // A GainControl behind the Downsampler (examples/relm_app/simple_receiver.rs:52-56) folded into the impulse response:
    // ir_f64 = gain_flt * ir_base, so every kernel's tables carry it and the device does nothing for it (results differ from
    // `sample * gain` by rounding only).  gain_flt = the gain cast to Flt, as GainControl casts it (transform.rs:55,64).
    double gain = 1.0;
    std::vector<double> ir_base;
    int set_gain(double g);
    bool have_rate = false;
    double prev_rate = 0.0;
    rr::Schedule sched;
    size_t L = 0;
    std::vector<double> ir_f64;
    rr::DevBuf d_ir;
    rr::DevBuf hist[2];  // the ring buffer's content in time order (L samples)
    int cur = 0;
    std::vector<uint32_t> emit;
    rr::DevBuf d_emit;
    uint64_t design_version = 0;
    // integer ratios 2, 4, 8 (f32): the chain's fused kernels with an all-ones NCO table instead of k_fir
    int fast_kind = 0;  // rr_chain::FK_*; FK_NONE: k_fir
    uint64_t fast_version = ~0ull;
    rr::DevBuf f_ctaps, f_H, f_tw, f_one;
    int f_Gp = 0, f_V = 0;
    bool f_poly = false;
    int f_blk = 1024;  // (k_ols_wave2k: 2048)
    // k_decim_poly (any integer ratio, short-period rational ratios): taps in f_ctaps, laid out for the schedule phase
    int f_NC = 0;
    size_t f_V64 = 0;  // FK_OLS64 (Complex<f64>, integer ratio): k_ols4096_f64's overlap; its tables in f_H / f_tw
    uint64_t poly_version = ~0ull;
    std::vector<int64_t> poly_delta;
    int ensure_poly_taps(const int64_t *e_first);
    int last_kernel = 0;  // what the last call ran (rr_chain::FK_*, 0 = k_fir)
    int ensure_fast();
    int prepare(double input_rate);
    int peek(double input_rate, size_t n_in, size_t *n_out);
    // nco != null (only after can_fuse_mixer() said yes): in[pos] * nco[(nco_idx0 + pos) mod nco_denom] is what gets
    // filtered - a FreqShifter in front fused into k_decim_poly; the history then holds mixed samples
    int process_dev(double input_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out,
                    const void *nco = nullptr, uint32_t nco_denom = 0, uint32_t nco_idx0 = 0);
    bool can_fuse_mixer(double input_rate, size_t n_in);
    // set by rr_meter: a FreqShifter in front is to ride along (k_decim_poly takes it in its staging) - at the ratios where both
    // k_ols_wg and the decimator apply the decimator is kept, the mixed stream is never written
    bool mixer_rides = false;
};

// metering::bandwidth as the last step of a pipeline (examples/bandwidth_meter/main.rs:78): while `on`, every spectrum a call
// produces also gets one f64 in bw[] (and its energy in energy[], if non-null), computed behind the transform
struct MeterSink {
    bool on = false;
    double dp = 0.0, rate = 0.0;
    double *bw = nullptr, *energy = nullptr;
    size_t cap = 0;  // frames bw[] / energy[] hold
    int store = 1;   // 0: the spectra themselves are not written
    rr::FrameMeter frame_meter() const {
        rr::FrameMeter fm;
        fm.double_percentile = dp;
        fm.sample_rate = rate;
        fm.bw = bw;
        fm.energy = energy;
        fm.store = store;
        return fm;
    }
};

// Fourier — analysis.rs:67-73 (previous_chunk_len, fft, window_values)
// Rechunker(chunk_len) -> Overlapper(chunk_count) -> Fourier (chunks.rs:42-242, analysis.rs:26-133;
// the wiring of examples/bandwidth_meter/main.rs:66-69): every chunk_len new samples one windowed
// transform over the last chunk_count * chunk_len samples.
struct rr_stft : rr_block {
    size_t M = 0, P = 0;
    size_t have_chunks = 0;  // chunks in the Overlapper's history, < P
    rr::DevBuf hist[2];      // the last (P-1)*M samples
    int cur = 0;
    rr_fourier *fo = nullptr;  // window + twiddles of the P*M-point transform
    // the Rechunker's patchwork (chunks.rs:62-64): < M samples waiting for the rest of their chunk
    rr::DevBuf carry, work;
    size_t carry_len = 0;
    MeterSink sink;  // rr_stft_set_metering
    ~rr_stft() override;
    size_t peek(size_t n_in) const {
        const size_t chunks = (carry_len + n_in) / M, total = have_chunks + chunks;
        return total >= P ? (total - (P - 1)) * M * P : 0;
    }
    int process_dev(const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
};

// The reference's own hot-path caller in its own order (examples/bandwidth_meter/main.rs:53-69):
//   FreqShifter -> Downsampler(chunk_len, ..) -> Filter (chunks of chunk_len at the output rate) -> Overlapper(overlap)
//   -> Fourier, on one device without host hops: four block handles, intermediates in device buffers.
struct rr_meter : rr_block {
    rr_freqshifter *fs = nullptr;
    rr_downsampler *ds = nullptr;
    rr_filter *fl = nullptr;
    rr_stft *st = nullptr;
    size_t chunk_len = 0, overlap = 0;
    double output_rate = 0;
    rr::DevBuf mixed, dec, filt;
    rr::DevBuf bwbuf;  // rr_meter_process_bandwidth: the call's bandwidths on the device
    bool last_front_fused = false;  // the last call ran FreqShifter + Downsampler as one kernel
    size_t dec_len = 0;  // decimated samples waiting for the rest of their chunk: the Downsampler's partly filled output
                         // chunk (resampling.rs:121-131); it survives events and rate changes like the reference's
    ~rr_meter() override;
    void set_streams();
    int peek(double sample_rate, size_t n_in, size_t *n_frames);
    int process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
};

struct rr_upsampler : rr_block {
    double output_rate = 0, bandwidth = 0, quality = 3.0;
    bool have_rate = false;
    double prev_rate = 0.0;
    rr::UpSchedule sched;
    size_t L = 0, Hn = 0;  // taps; inputs kept from call to call
    std::vector<double> ir_f64;
    rr::DevBuf d_ir;
    rr::DevBuf hist[2];  // the last Hn inputs in time order
    int cur = 0;
    std::vector<int32_t> before_hist, before;  // generic ratio: outputs released before each kept / new input
    bool before_hist_stale = false;            // (closed-form calls do not keep it)
    rr::DevBuf d_before;
    int prepare(double input_rate);
    int peek(double input_rate, size_t n_in, size_t *n_out);
    int process_dev(double input_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
};

struct rr_fmdemod : rr_block {
    double deviation = 0;
    double gain = 1.0;  // a GainControl behind the demodulator (transform.rs:62-72), applied on the store: exact
    bool have_prev = false;
    rr::DevBuf state[2];  // {previous sample, last output}, ping-pong
    int cur = 0;
    bool state_init = false;
    int process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
};

struct rr_fourier : rr_block {
    rr_window window{RR_WIN_RECTANGULAR, 0.0};
    bool center_dc = false;
    size_t n = 0;  // designed chunk length (0 = none)
    std::vector<double> sampled;  // RR_WIN_SAMPLED: relative values for sampled_n
    size_t sampled_n = 0;
    std::vector<double> window_f64;  // scaled window of the current design
    rr::DevBuf d_window, d_tw;
    // powers of two beyond the LDS kernels (up to 2^24): four-step through a workspace in HBM
    bool big = false;
    rr::DevBuf big_ws;
    // .. by default as row transforms between tiled transposes (launch_transpose_mul): nested rectangular-window
    // transforms of N1 and N2 points, twiddle tables tB (2^big_h entries) | tA behind each other in d_tw
    bool big_t = false;
    int big_h = 0;
    // two passes (k_fft_tile) instead of five launches: e^{-j 2 pi k / N1} and e^{-j 2 pi k / N2} follow tB | tA in d_tw
    bool big_tile = false;
    size_t big_tw1_off = 0, big_tw2_off = 0;  // element offsets into d_tw
    rr_fourier *bigA = nullptr, *bigB = nullptr;
    rr::DevBuf big_ws2;
    // Bluestein for lengths that are not powers of two (n >= 32, either dtype): two transforms of bs_M points
    // by a nested rectangular-window Fourier, tables c = window * conj(chirp), B = F(chirp) / M, w = chirp
    bool force_mixed = false;  // set by the Channelizer: k_fft_mixed wherever it applies (the fold rides on its load)
    bool prefer_tile = false;  // set on Bluestein's inner transform: 16 384 points through k_fft_tile too (its passes carry the chirp products)
    bool mixed = false;     // 2^a 3^b 5^c points (<= 8192 in f32, <= 4096 in f64), not a power of two: k_fft_mixed (one launch, n log n work)
    // 2^a 3^b 5^c points beyond one LDS image (up to 512 x 512): two passes, k_fft_tilem; d_tw = twN1 | twN2 | T1 | T2
    bool tilem = false;
    size_t tm_N1 = 0, tm_N2 = 0, tm_T1 = 0, tm_T2 = 0;  // (T1 / T2: element offsets into d_tw)
    size_t bs_M = 0;
    bool bs_fused = false;  // f32, 513 .. 2048 points: k_bluestein4096 (one launch per call)
    bool bs_wave = false;   // f32, 32 .. 512 points: k_bluestein1024 (a wave per chunk)
    bool bs_fused8k = false;  // f32, 2049 .. 4096 points: k_bluestein8192 (RR_FOURIER_BS8K=regs)
    bool bs_big = false;      // f32, 2049 .. 8192 points: k_bluestein_big<8192 / 16384>
    bool bs_lds = false;    // M <= 8192 (f32) / 4096 (f64): k_bluestein_lds (one kernel, Stockham passes between two LDS images)
    rr_fourier *bs_fft = nullptr;
    rr::DevBuf d_bs_c, d_bs_B, d_bs_w, bs_ws[2];
    ~rr_fourier() override;
    int prepare(size_t len);
    // `count` windowed transforms of n points over [head | in] at distance hop
    int transform_dev(const void *head, size_t n_head, const void *in, void *out, size_t hop, size_t count);
    // the same + metering::bandwidth per frame: in the transform's own kernel where one with the epilogue exists (Complex<f32>,
    // 4096 points), else the transform (into meter_ws when the caller does not want the spectra) and the parallel scan behind it
    rr::DevBuf meter_ws;
    int transform_metered_dev(const void *head, size_t n_head, const void *in, void *out, size_t hop, size_t count,
                              const rr::FrameMeter &fm);
    int process_dev(size_t chunk_len, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
};

// Channelizer — Overlapper's history (chunks.rs:200-223) + Fourier's window (analysis.rs:67-73)
struct rr_channelizer : rr_block {
    size_t M = 0, P = 0;
    size_t hop = 0;          // samples between frames: M (critically sampled) or a divisor of P M (oversampled)
    size_t have_chunks = 0;  // chunks of `hop` samples in the Overlapper's history, < P M / hop
    rr::DevBuf hist[2];      // the last P M - hop samples
    int cur = 0;
    rr::DevBuf d_window, d_tw;
    // general form (hop != M, or M not a power of two): fold to a workspace, then M-point transforms by a
    // rectangular-window Fourier (any M)
    rr_fourier *fo = nullptr;
    rr::DevBuf fold_ws;
    ~rr_channelizer() override;
    size_t span_chunks() const { return P * M / hop; }  // the Overlapper's chunk count
    size_t peek(size_t n_in) const {
        const size_t chunks = n_in / hop, total = have_chunks + chunks, K = span_chunks();
        return total >= K ? (total - (K - 1)) * M : 0;
    }
    int process_dev(const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
};

// hipEvent brackets around the chain's kernels (measurement aid)
enum Stage { ST_FREQSHIFT = 0, ST_FILTER, ST_DECIM, ST_FOURIER, ST_FUSED_FIR, ST_FUSED_FFT, ST_COUNT };
struct StageTimers {
    bool on = false;
    int only_stage = -1;  // >= 0: only this stage is timed (each timed launch costs ~5 us of stream time)
    unsigned every = 1, seen = 0;  // begin_ext: one launch in `every` records its start / end
    struct Pair { hipEvent_t a, b; int stage; bool a_shared; };  // a_shared: `a` is the previous pair's `b`
    std::vector<Pair> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[ST_COUNT] = {};
    uint64_t launches[ST_COUNT] = {};
    int begin(int stage, hipStream_t s);  // returns index into pending or -1
    void end(int idx, hipStream_t s);
    // ends stage `idx` and begins `stage` with ONE event (a record costs ~4 us of stream time)
    int next(int idx, int stage, hipStream_t s);
    // a pair whose events the launch itself fills in (hipExtLaunchKernel): no marker packets
    bool begin_ext(int stage, hipEvent_t *a, hipEvent_t *b);
    int drain();
    void reset();
    ~StageTimers();
};

// Chain — four blocks wired on one device/stream (bandwidth_meter/main.rs:51-72)
struct rr_chain : rr_block {
    rr_chain_params p{};
    rr_freqshifter *fs = nullptr;
    rr_filter *fl = nullptr;
    rr_downsampler *ds = nullptr;
    rr_fourier *fo = nullptr;
    // Rechunker(filter_len) in front of the Filter: < filter_len mixed samples
    rr::DevBuf carry;
    size_t carry_len = 0;
    rr::DevBuf mixed, filtered, decim;
    // Downsampler's partly filled output chunk (resampling.rs:121-131)
    rr::DevBuf pending;
    size_t pending_len = 0;
    MeterSink sink;  // rr_chain_set_metering
    uint64_t mutations = 0;  // counts the entry points that change the chain's state (a bank's cheap "has anyone driven this lane?")
    int last_fused = 0;
    StageTimers timers;
    // ---- fused fast path state (rr_api.hip, "fused") ----
    rr::DevBuf xh[2];  // last HX mixed samples of the stream entering the Filter (+ carry at the tail)
    int xh_cur = 0;
    size_t HX = 0;
    size_t xh_count = 0;  // mixed samples tracked in xh since it was (re)allocated, saturating at HX
    bool blocks_stale = false;  // Filter/Downsampler histories not updated by the fused kernels
    uint64_t zrun = 0;          // Filter outputs since the last discontinuity
    rr::DevBuf d_ctaps;
    // fused path with fft_len 4096: the Downsampler's partly filled chunk stays where the
    // kernel wrote it (tail of one of two output buffers) instead of being copied around
    rr::DevBuf dec2[2];
    int dec_cur = 0;
    const void *pend_ptr = nullptr;  // non-null: pending samples live here, not in `pending`
    int Gp = 0;
    size_t Lc = 0;
    uint64_t ctaps_fl = ~0ull, ctaps_ds = ~0ull;
    // overlap-save variant of the fused FIR (k_ols_decim4)
    bool use_ols = false;
    int ols_V = 0;
    bool ols_poly = false;
    int ols_N = 4096;  // 4096: k_ols_decim4 (workgroup per block), 1024: k_ols_wave (wave per block)
    int ols_blk = 1024;  // the wave kernels' block: 2048 = k_ols_wave2k (8 : 1)
    rr::DevBuf d_olsH, d_tw4096;
    bool fused_candidate(double sample_rate) const;
    enum { FK_NONE = 0, FK_DIRECT, FK_OLS, FK_OLSW, FK_OLSF, FK_POLY, FK_SELECT = 10, FK_OLS64 = 11 };  // FK_POLY: k_decim_poly, FK_SELECT: k_filter_wave<true> (Downsampler only)
    static int pick_fused_kernel(uint64_t D, size_t lc, bool real_taps, size_t fft_len);
    bool use_frame = false;      // FK_OLSF: k_ols_frame (FIR stage + Fourier in one kernel)
    // Complex<f64>: mixer + combined FIR + decimation as ONE pass of k_decim_poly_f64 (the polyphase kernel with the phase table
    // riding along: 16 B read + 16 / D written per sample instead of the four blocks' 84), then the Fourier block
    bool use_ols64 = false;   // Complex<f64>: the front end through k_ols4096_f64 (tables in d_olsH / d_tw4096, overlap ols_V)
    bool use_poly64 = false;  // the front end through k_decim_poly(_f64): Complex<f64>, and f32 at integer ratios without an overlap-save kernel
    int poly64_Lp = 0;
    // k_ols_frame with the mixer folded into the tables (NCO periods that divide 8): G'_p[k] = G_p[(k + s) mod 256] e^{j 2 pi p numer / R}
    std::vector<rr::cd> olsG64;
    rr::DevBuf d_olsHmix;
    rr::DevBuf d_olsHgen;                 // k_ols_frame<.., GP>: tables of the response c[i] w^-i (ensure_genfold)
    std::vector<rr::cd> ctaps_cc;         // the combined taps in f64 (frame kernel only)
    int64_t gen_numer = -1, gen_denom = -1;
    uint64_t gen_ctaps_fl = ~0ull, gen_ctaps_ds = ~0ull;
    int ensure_genfold();
    int64_t mix_numer = 0, mix_denom = 0;
    uint64_t mix_ctaps_fl = ~0ull, mix_ctaps_ds = ~0ull, mix_table_version = ~0ull;
    uint64_t frame_table_version = ~0ull;  // the NCO table the mixed-sample history was last written with by a k_ols_frame / k_ols_wave call
    float mix_sigma = 1.f;
    int ensure_mixfold();
    int fold_mixer(rr::FusedFirArgs &a, int64_t back, bool frame = false);  // (frame: k_ols_frame - the one with the GP instance)
    rr::DevBuf pendbuf[2];       // its pending decimated samples, ping-pong
    int pb_cur = 0;
    int ensure_xh();
    int ensure_ctaps();
    int materialize();  // bring the per-block histories up to date after fused calls
    int materialize_pending_append(const void *newv, size_t dec);
    int process_fused(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
    int process_generic(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
    // ---- lockstep banks (rr_chainbank): the two-kernel fused step split into plan / pointers / commit, so that K chains
    // with the same parameters and the same stream position run their step as ONE launch per kernel ----
    struct BankStep {
        rr::FusedFirArgs a;  // the launch parameters every channel shares (no stream pointers)
        size_t whole = 0, dec = 0, nfr = 0, rest = 0, n_head = 0;
    };
    struct BankSig {  // what has to agree between the lanes of a bank for the lockstep step (stream position and table state)
        uint64_t phase_idx, zrun, sched_phase, fs_version, frame_version, ctaps_fl, ctaps_ds;
        size_t carry_len, pending_len, HX, xh_count, Lc;
        double sched_pos, rate;
        int xh_cur, dec_cur, hist_valid, use_frame, ols_N, ols_poly, pend_in_dec;
    };
    BankSig bank_signature() const;
    int bank_plan(double sample_rate, size_t n_in, size_t cap, BankStep &st, bool *ok);
    int bank_pointers(const BankStep &st, const void *d_in, void *d_out, rr::BankPtrs &p);
    void bank_commit(const BankStep &st, size_t n_in);
    ~rr_chain() override;
    int peek(double sample_rate, size_t n_in, size_t *n_frames);
    int process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out);
};

// K chains with the same parameters whose streams advance in lockstep (the K antennas of an array, the K sub-bands of a
// channelized receiver, ...): every lane is a full rr_chain (any lane can be driven alone at any time); when all of them are
// in the steady fused state at the same stream position, a call runs ONE k_ols_wave_bank and ONE k_fft4096_bank launch for all
// channels (channel = blockIdx.y) instead of 2 K launches - at the reference's chunk sizes (10^3 .. 10^5 samples,
// examples/bandwidth_meter/main.rs:56) a single channel's call is launch-bound.
struct rr_chainbank : rr_block {
    std::vector<rr_chain *> lanes;
    std::vector<uint64_t> seen;  // the lanes' mutation counters when the bank last found (or left) them in lockstep
    bool verified = false;       // the lanes' signatures agreed and nothing has touched a lane since
    int last_path = 0;  // 1: the last call ran in lockstep (two launches per 64 channels), 0: lane by lane
    ~rr_chainbank() override;
    int process_dev(double rate, const void *d_in, size_t in_stride, size_t n_in, void *d_out, size_t out_stride, size_t cap,
                    size_t *n_out);
};
