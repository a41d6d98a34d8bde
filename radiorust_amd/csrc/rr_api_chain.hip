// rr_api_chain.hip — Chain (the four blocks on one device: block by block or through the fused kernels), ChainBank (K chains in
// lockstep), Meter (the example's own order), the stage timers: host logic and extern "C" entry points.
#include "rr_api_common.hpp"

// ---------------------------------------------------------------------------
// Meter: FreqShifter -> Downsampler -> Filter -> Overlapper -> Fourier (examples/bandwidth_meter/main.rs:53-69)
// ---------------------------------------------------------------------------
rr_meter::~rr_meter() {
    delete fs;
    delete ds;
    delete fl;
    delete st;
}

void rr_meter::set_streams() {
    fs->stream = ds->stream = fl->stream = st->stream = stream;
    if (st->fo) st->fo->stream = stream;
}

int rr_meter::peek(double sample_rate, size_t n_in, size_t *n_frames) {
    size_t m = 0;
    RR_TRY(ds->peek(sample_rate, n_in, &m));
    const size_t whole = (dec_len + m) / chunk_len * chunk_len;
    const size_t k = fl->designed ? fl->peek(whole) : 0;
    *n_frames = st->peek(k) / (chunk_len * overlap);
    return RR_OK;
}

int rr_meter::process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    if (n_out) *n_out = 0;
    if (!fl->designed) RR_FAIL(RR_ERR_NEED_DESIGN, "Meter: the Filter has no design yet (rr_meter_filter_design)");
    size_t frames = 0;
    RR_TRY(peek(sample_rate, n_in, &frames));
    const size_t N = chunk_len * overlap;
    if ((!st->sink.on || st->sink.store) && frames * N > cap) RR_FAIL(RR_ERR_CAPACITY, "Meter: out_cap %zu < %zu", cap, frames * N);
    if (st->sink.on && frames > st->sink.cap)
        RR_FAIL(RR_ERR_CAPACITY, "Meter: room for %zu bandwidths, the call makes %zu spectra", st->sink.cap, frames);
    if (n_in == 0) return RR_OK;
    RR_TRY(select());
    set_streams();
    const size_t esz = elem_size(dtype);
    size_t m = 0;
    RR_TRY(ds->peek(sample_rate, n_in, &m));
    // room first (nothing has changed state yet); `dec` keeps the pending samples when it has to grow
    if ((dec_len + m) * esz > dec.cap) {
        rr::DevBuf bigger;
        RR_TRY(bigger.reserve((dec_len + m) * esz + chunk_len * esz));
        if (dec_len) RR_HIP(hipMemcpyAsync(bigger.p, dec.p, dec_len * esz, hipMemcpyDeviceToDevice, stream));
        RR_HIP(hipStreamSynchronize(stream));  // the old buffer is freed below
        std::swap(dec.p, bigger.p);
        std::swap(dec.cap, bigger.cap);
    }
    RR_TRY(filt.reserve((dec_len + m + 1) * esz));
    size_t got = 0;
    if (ds->can_fuse_mixer(sample_rate, n_in)) {
        // FreqShifter and Downsampler in ONE pass over the input (k_decim_poly with the phase table riding along): the
        // mixed stream is never written; the Downsampler's history holds mixed samples either way
        RR_TRY(fs->prepare(sample_rate));  // table for this rate and shift, phase kept (transform.rs:318-340)
        RR_TRY(ds->process_dev(sample_rate, d_in, n_in, dec.as<char>() + dec_len * esz, m, &got, fs->d_table.p, (uint32_t)fs->denom,
                               (uint32_t)fs->phase_idx));
        fs->phase_idx = (fs->phase_idx + n_in % (uint64_t)fs->denom) % (uint64_t)fs->denom;
        last_front_fused = true;
    } else {
        RR_TRY(mixed.reserve(n_in * esz));
        RR_TRY(fs->process_dev(sample_rate, d_in, n_in, mixed.p, n_in, &got));
        RR_TRY(ds->process_dev(sample_rate, mixed.p, n_in, dec.as<char>() + dec_len * esz, m, &got));
        last_front_fused = false;
    }
    const size_t total = dec_len + got, whole = total / chunk_len * chunk_len, left = total - whole;
    size_t wrote = 0;
    if (whole) {
        size_t k = 0;
        RR_TRY(fl->process_dev(output_rate, dec.p, whole, filt.p, whole, &k));
        if (k) RR_TRY(st->process_dev(filt.p, k, d_out, cap, &wrote));
        // the samples of the chunk that is still filling move to the front (left < chunk_len <= whole: no overlap)
        if (left) RR_HIP(hipMemcpyAsync(dec.p, dec.as<char>() + whole * esz, left * esz, hipMemcpyDeviceToDevice, stream));
    }
    dec_len = left;
    if (n_out) *n_out = wrote;
    return RR_OK;
}

// ---------------------------------------------------------------------------
// stage timers
// ---------------------------------------------------------------------------
int StageTimers::begin(int stage, hipStream_t s) {
    if (!on || (only_stage >= 0 && stage != only_stage)) return -1;
    if (pending.size() >= 8192 && drain() != RR_OK) return -1;
    auto get = [&]() -> hipEvent_t {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    };
    Pair p{get(), get(), stage, false};
    if (!p.a || !p.b) return -1;
    (void)hipEventRecord(p.a, s);
    pending.push_back(p);
    return (int)pending.size() - 1;
}
void StageTimers::end(int idx, hipStream_t s) {
    if (idx >= 0) (void)hipEventRecord(pending[idx].b, s);
}
bool StageTimers::begin_ext(int stage, hipEvent_t *a, hipEvent_t *b) {
    *a = *b = nullptr;
    if (!on || (only_stage >= 0 && stage != only_stage)) return false;
    if (every > 1 && (seen++ % every) != 0) return false;
    if (pending.size() >= 8192 && drain() != RR_OK) return false;
    hipEvent_t e[2] = {nullptr, nullptr};
    for (hipEvent_t &x : e) {
        if (!pool.empty()) {
            x = pool.back();
            pool.pop_back();
        } else if (hipEventCreate(&x) != hipSuccess) {
            return false;
        }
    }
    pending.push_back(Pair{e[0], e[1], stage, false});
    *a = e[0];
    *b = e[1];
    return true;
}
int StageTimers::next(int idx, int stage, hipStream_t s) {
    if (idx < 0) return begin(stage, s);
    end(idx, s);
    if (only_stage >= 0 && stage != only_stage) return -1;
    hipEvent_t e = nullptr;
    if (!pool.empty()) {
        e = pool.back();
        pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        return -1;
    }
    pending.push_back(Pair{pending[idx].b, e, stage, true});
    return (int)pending.size() - 1;
}
int StageTimers::drain() {
    for (Pair &p : pending) {
        RR_HIP(hipEventSynchronize(p.b));
        float ms = 0.f;
        RR_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        total_ms[p.stage] += ms;
        launches[p.stage] += 1;
        if (!p.a_shared) pool.push_back(p.a);
        pool.push_back(p.b);
    }
    pending.clear();
    return RR_OK;
}
void StageTimers::reset() {
    (void)drain();
    for (int i = 0; i < ST_COUNT; ++i) {
        total_ms[i] = 0;
        launches[i] = 0;
    }
}
StageTimers::~StageTimers() {
    for (Pair &p : pending) {
        if (!p.a_shared) (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
}

// ---------------------------------------------------------------------------
// Chain
//   process_generic : the four blocks one after the other (any parameters)
//   process_fused   : k_mix_fir_decim + k_fft4096/k_fft (Complex<f32>, integer
//                     decimation, real taps), selected per call when the whole
//                     call is in steady state; the two paths hand their state to
//                     each other exactly (materialize / xh history).
// ---------------------------------------------------------------------------
rr_chain::~rr_chain() {
    delete fs;
    delete fl;
    delete ds;
    delete fo;
}

int rr_chain::peek(double sample_rate, size_t n_in, size_t *n_frames) {
    const size_t nf = p.filter_len;
    const size_t chunks = (carry_len + n_in) / nf;
    const size_t filt = fl->peek(chunks * nf);
    size_t dec = 0;
    RR_TRY(ds->peek(sample_rate, filt, &dec));
    *n_frames = (pending_len + dec) / p.fft_len;
    return RR_OK;
}

// parameters for which the fused kernels exist at all (independent of stream state)
// Complex<f64>: overlap-save in blocks of 4096 points (any taps, Lc <= 2049; RR_CHAIN_F64_FUSED=poly keeps the decimator), else
// the polyphase decimator (real taps)
static bool chain_ols64_ok(const rr_chain *c, size_t lc) {
    const char *e = std::getenv("RR_CHAIN_F64_FUSED");  // ("0": the four blocks; "poly": the decimator; read per design)
    if (e && (!std::strcmp(e, "0") || !std::strcmp(e, "poly"))) return false;
    return c->ds->sched.D >= 1 && ols4096_f64_overlap(lc, c->ds->sched.D) != 0;
}
static bool chain_poly64_ok(const rr_chain *c, size_t lc) {
    const char *e = std::getenv("RR_CHAIN_F64_FUSED");
    if (e && !std::strcmp(e, "0")) return false;
    if (chain_ols64_ok(c, lc)) return true;
    return c->fl->real_taps && c->ds->sched.D >= 2 && decim_poly_supported(RR_F64, c->ds->sched.D, 1, lc);
}

bool rr_chain::fused_candidate(double sample_rate) const {
    if (!p.allow_fused) return false;
    if (!fl->designed) return false;
    if (!ds->have_rate || ds->prev_rate != sample_rate || !ds->sched.integer_ratio) return false;
    const size_t lc = ds->L + fl->n - 1;
    if (dtype == RR_F64) return chain_poly64_ok(this, lc);  // (RR_CHAIN_F64_FUSED=0 keeps the four blocks)
    return pick_fused_kernel(ds->sched.D, lc, fl->real_taps, p.fft_len) != FK_NONE;
}

int rr_chain::ensure_xh() {
    const size_t want = ds->L + 2 * fl->n + 8;
    if (want == HX) return RR_OK;
    const size_t bytes = want * elem_size(dtype);
    RR_TRY(xh[0].reserve(bytes));
    RR_TRY(xh[1].reserve(bytes));
    RR_HIP(hipMemsetAsync(xh[0].p, 0, bytes, stream));
    RR_HIP(hipMemsetAsync(xh[1].p, 0, bytes, stream));
    xh_cur = 0;
    HX = want;
    xh_count = 0;  // stay on the block-by-block path until the history has filled
    return RR_OK;
}

// Three fused mix + FIR + decimate implementations (measured on cfg2, 2^26 samples, Lc = 183):
//   direct  k_mix_fir_decim  direct form, real taps, D in {2, 4, 8}; cost ~ Lc           0.222 ms
//   ols     k_ols_decim4     overlap-save, workgroup per 4096-block, D = 4, any taps       0.21 ms
//   olsw    k_ols_wave<D>    overlap-save, wave per 1024-block, D in {2, 4, 8}, any taps, Lc <= 513   0.132 ms
//   olsf    k_ols_frame      olsw's blocks + the 4096-point Fourier stage in one kernel (a workgroup
//                            per frame), D = 4, 129 <= Lc <= 193, fft_len = 4096; 0.160 ms for BOTH
//                            stages against olsw + k_fft4096 = 0.170: the default where it applies
// Unforced: olsw wherever it applies (D in {2, 4, 8}, Lc <= 513), ols beyond an overlap of 384 at D = 4, the direct
// form for what is left.  RR_FUSED_KERNEL = direct | ols | olsw | olsf forces one of them where it applies (A/B
// runs and tests).
int rr_chain::pick_fused_kernel(uint64_t D, size_t lc, bool real_taps, size_t fft_len) {
    const bool can_direct = real_taps && fused_fir_supported(D, lc);
    const bool can_ols = ols_decim_supported(D, lc), can_wave = ols_wave_supported(D, lc) || ols_wave2k_supported(D, lc) || ols_wg_supported(D, lc);
    const bool can_frame = can_wave && ols_frame_supported(D, lc, fft_len);
    const char *e = std::getenv("RR_FUSED_KERNEL");
    if (e) {
        if (!std::strcmp(e, "direct") && can_direct) return FK_DIRECT;
        if (!std::strcmp(e, "ols") && can_ols) return FK_OLS;
        if (!std::strcmp(e, "olsw") && can_wave) return FK_OLSW;
        if (!std::strcmp(e, "olsf") && can_frame) return FK_OLSF;
    }
    // Overlap-save with a wave per 1024-block for every ratio it folds (2, 4, 8) and every length it reaches: since
    // round 2 it is ahead of the direct form for short responses too (scripts/bench_decim_ab.py, ms per 2^26 samples:
    // 2 : 1 L = 32 0.149 against 0.173, 4 : 1 L = 60 0.121 against 0.144, 8 : 1 L = 83 0.117 against 0.129).  Beyond an
    // overlap of 384 of the 1024 samples (Lc > 385) the 4096-blocks are ahead at 4 : 1 (measured: Lc = 455: 0.252
    // against 0.259 ms per step; Lc = 375: 0.250 against 0.222).
    // the whole chain in one kernel where its shape is compiled in (4 : 1, 4096-point Fourier stage, 129 <= Lc <= 193 - cfg2):
    // since its second form (time-shared LDS, polyphase blocks) 0.160 ms per 2^26 samples against 0.126 + 0.044
    if (can_frame) return FK_OLSF;
    if (can_wave && (ols_wave_overlap(lc) <= 384 || !can_ols)) return FK_OLSW;
    if (can_ols) return FK_OLS;
    if (can_direct) return FK_DIRECT;
    if (can_wave) return FK_OLSW;
    if (can_ols) return FK_OLS;
    // every other integer ratio (the example's 10 : 1, examples/bandwidth_meter/main.rs:56): mixer + combined response + decimation
    // as one pass of the polyphase decimator, as the f64 chain (real taps; RR_CHAIN_POLY=0 keeps the four blocks)
    static const bool poly_off = [] { const char *pe = std::getenv("RR_CHAIN_POLY"); return pe && std::atoi(pe) == 0; }();
    if (!poly_off && real_taps && D >= 2 && decim_poly_supported(RR_F32, D, 1, lc)) return FK_POLY;
    return FK_NONE;
}

// Host side of the fused decimating-FIR kernels' tables, for combined taps c (c[i] multiplies x[t - i];
// `c` real parts, `cc` complex).  kind = rr_chain::FK_*:
//   FK_DIRECT           ctaps in the step order of k_mix_fir_decim: tb[t*D + p] = c[D*(Gp-1-t) + (D-1-p)], zero beyond Lc
//   FK_OLS              H = DFT_4096(c) / 4096 and e^{-j 2 pi k / 4096}
//   FK_OLSW / FK_OLSF   H = DFT_1024(c) / 1024 pair-interleaved, e^{-j 2 pi k / 1024} + the lane seeds
void build_fused_fir_tables(int kind, uint64_t D_, const std::vector<double> &c, const std::vector<cd> &cc, FusedFirTables &t) {
    const size_t lc = c.size();
    t.kind = kind;
    t.poly = false;
    const bool wave = kind == rr_chain::FK_OLSW || kind == rr_chain::FK_OLSF;
    if (wave || kind == rr_chain::FK_OLS) {
        // H = DFT_N(c) / N (the inverse transform in the kernel is unnormalised)
        const size_t N = wave ? 1024 : 4096;
        std::vector<cd> h(N, cd(0, 0));
        for (size_t i = 0; i < lc && i < N; ++i) h[i] = cc[i];  // (k_ols_wave2k / k_ols_wg take longer responses: their own transforms below)
        fft_f64(h, false);
        std::vector<float> hb(2 * N), twb(2 * N);
        for (size_t i = 0; i < N; ++i) {
            hb[2 * i] = (float)(h[i].real() / (double)N);
            hb[2 * i + 1] = (float)(h[i].imag() / (double)N);
            const double ang = -2.0 * M_PI * (double)i / (double)N;
            twb[2 * i] = (float)std::cos(ang);
            twb[2 * i + 1] = (float)std::sin(ang);
        }
        static const bool no_poly = [] { const char *e = std::getenv("RR_OLSW_POLY"); return e && std::atoi(e) == 0; }();
        t.blk = 1024;
        if (kind == rr_chain::FK_OLSW && !no_poly && ols_wg_supported(D_, lc)) {
            // k_ols_wg (even ratios 10 .. 64, a workgroup per block of N = 256 D samples): Y[k] = sum_p X_p[k] G_p[k] over the D phases
            // x_p[m] = xs[D m + p] (X_p = DFT_256 x_p), G_p[k] = sum_q H[k + 256 q] W_N^((k + 256 q) p), H = DFT_N(c) / N, k < 256.
            // N need not be a power of two: H[k] = sum_(n2 < D) W_N^(n2 k) F_n2[k mod 256], F_n2 = DFT_256 of c[D n1 + n2].
            // Run h = p / 4 (four phases; the runs beyond D hold zeros): its 1024 entries one behind the other in
            // k_ols_wave<4, POLY>'s layout, as k_ols_wave2k's
            const size_t D = (size_t)D_, NN = 256 * D, runs = (size_t)ols_wg_runs(D_);
            std::vector<cd> wN(NN);
            for (size_t i = 0; i < NN; ++i) {
                const double ang = -2.0 * M_PI * (double)i / (double)NN;
                wN[i] = cd(std::cos(ang), std::sin(ang));
            }
            std::vector<cd> hN(NN, cd(0, 0)), F(256);
            for (size_t n2 = 0; n2 < D; ++n2) {
                for (size_t n1 = 0; n1 < 256; ++n1) F[n1] = D * n1 + n2 < lc ? cc[D * n1 + n2] : cd(0, 0);
                fft_f64(F, false);
                for (size_t k = 0; k < NN; ++k) hN[k] += wN[(n2 * k) % NN] * F[k & 255];
            }
            std::vector<float> gp(2 * 1024 * runs, 0.f);
            for (size_t pD = 0; pD < D; ++pD)
                for (size_t k = 0; k < 256; ++k) {
                    cd g(0, 0);
                    for (size_t qq = 0; qq < D; ++qq) {
                        const size_t kk = k + 256 * qq;
                        g += hN[kk] / (double)NN * wN[(kk * pD) % NN];
                    }
                    const size_t hh = pD / 4, pp = pD % 4, l = k % 64, c = k / 64, i = 4 * pp + c;
                    const size_t dst = 1024 * hh + ((i >> 1) * 64 + l) * 2 + (i & 1);
                    gp[2 * dst] = (float)g.real();
                    gp[2 * dst + 1] = (float)g.imag();
                }
            append_wave1024_seeds(twb);
            t.H.swap(gp);
            t.tw.swap(twb);
            t.poly = true;
            t.blk = (int)NN;
            t.V = ols_wg_overlap(D_, lc);
            t.N = 1024;  // (the wave kernels' mark: rr_chain::ols_N)
            return;
        }
        if (kind == rr_chain::FK_OLSW && !no_poly && ols_wave2k_supported(D_, lc)) {
            // k_ols_wave2k (8 : 1, a wave per 2048-sample block): Y[k] = sum_p X_p[k] G_p[k] over the 8 phases x_p[m] = xs[8 m + p]
            // (X_p = DFT_256 x_p), G_p[k] = sum_q H[k + 256 q] W_2048^((k + 256 q) p), H = DFT_2048(c) / 2048, k < 256.  The kernel
            // runs the phases in two halves of four (p = 4 hh + pp): half hh's 1024 entries one behind the other, lane l = k mod 64
            // reads entry i = 4 pp + k / 64 as one half of the 16-byte piece [hh][i >> 1][l] - the layout of k_ols_wave<4, POLY> twice
            std::vector<cd> h2(2048, cd(0, 0));
            for (size_t i = 0; i < lc; ++i) h2[i] = cc[i];
            fft_f64(h2, false);
            std::vector<float> gp(2 * 2048);
            for (size_t p8 = 0; p8 < 8; ++p8)
                for (size_t k = 0; k < 256; ++k) {
                    cd g(0, 0);
                    for (size_t qq = 0; qq < 8; ++qq) {
                        const size_t kk = k + 256 * qq;
                        const double ang = -2.0 * M_PI * (double)((kk * p8) % 2048) / 2048.0;
                        g += h2[kk] / 2048.0 * cd(std::cos(ang), std::sin(ang));
                    }
                    const size_t hh = p8 / 4, pp = p8 % 4, l = k % 64, c = k / 64, i = 4 * pp + c;
                    const size_t dst = 1024 * hh + ((i >> 1) * 64 + l) * 2 + (i & 1);
                    gp[2 * dst] = (float)g.real();
                    gp[2 * dst + 1] = (float)g.imag();
                }
            append_wave1024_seeds(twb);
            t.H.swap(gp);
            t.tw.swap(twb);
            t.poly = true;
            t.blk = 2048;
            t.V = ols_wave_overlap(lc, 16);
            t.N = 1024;  // (the wave kernels' mark: rr_chain::ols_N)
            return;
        }
        // (the frame kernel exists in the polyphase form only)
        if ((kind == rr_chain::FK_OLSW && (D_ == 2 || D_ == 4 || D_ == 8) && !no_poly) || (kind == rr_chain::FK_OLSF && D_ == 4)) {
            // k_ols_wave<D, POLY>: Y[k] = sum_p X_p[k] G_p[k] over the D phases x_p[m] = xs[D m + p] (X_p = DFT_(1024/D) x_p),
            // G_p[k] = sum_q H[k + (1024 / D) q] W_1024^((k + (1024 / D) q) p), k < 1024 / D; lane l = k mod 64 reads entry
            // i = (16 / D) p + k / 64 as one half of the 16-byte piece [i >> 1][l]   (RR_OLSW_POLY=0 keeps the 1024-point
            // forward transform: A/B runs)
            const size_t D = (size_t)D_, NB = 1024 / D, ND = 16 / D;
            std::vector<float> gp(2 * N);
            if (D == 4) t.G64.assign(1024, cd(0, 0));
            for (size_t pp = 0; pp < D; ++pp)
                for (size_t k = 0; k < NB; ++k) {
                    cd g(0, 0);
                    for (size_t qq = 0; qq < D; ++qq) {
                        const size_t kk = k + NB * qq;
                        const double ang = -2.0 * M_PI * (double)((kk * pp) % 1024) / 1024.0;
                        g += h[kk] / (double)N * cd(std::cos(ang), std::sin(ang));
                    }
                    const size_t l = k % 64, c = k / 64, i = ND * pp + c, dst = ((i >> 1) * 64 + l) * 2 + (i & 1);
                    gp[2 * dst] = (float)g.real();
                    gp[2 * dst + 1] = (float)g.imag();
                    if (D == 4) t.G64[256 * pp + k] = g;
                }
            hb.swap(gp);
            append_wave1024_seeds(twb);
            t.poly = true;
        } else if (wave) {  // k_ols_wave reads H as Hp[kp][l] = {H[l + 128 kp], H[l + 128 kp + 64]}, kp < 8, l < 64
            std::vector<float> hp(2 * N);
            for (size_t kp = 0; kp < 8; ++kp)
                for (size_t l = 0; l < 64; ++l)
                    for (size_t j = 0; j < 2; ++j) {
                        const size_t src = l + 128 * kp + 64 * j, dst = (kp * 64 + l) * 2 + j;
                        hp[2 * dst] = hb[2 * src];
                        hp[2 * dst + 1] = hb[2 * src + 1];
                    }
            hb.swap(hp);
            append_wave1024_seeds(twb);
        }
        t.H.swap(hb);
        t.tw.swap(twb);
        // k_ols_wave: the overlap in steps of 16 samples - block starts stay on 128-byte lines (in steps of 8, cfg2's
        // V = 184 instead of 192 measured 0.5-3 % SLOWER: every other block then starts in the middle of a line)
        t.V = wave ? ols_wave_overlap(lc, kind == rr_chain::FK_OLSF ? 64 : 16) : ols_decim_overlap(lc);
        t.N = (int)N;
        return;
    }
    const int D = (int)D_;
    const int gp = (int)((lc + D - 1) / D);  // tap groups of D; the kernel runs gp/R full rounds + a partial one
    t.ctaps.assign((size_t)gp * D, 0.f);
    for (int g = 0; g < gp; ++g)
        for (int q = 0; q < D; ++q) {
            const size_t i = (size_t)D * (gp - 1 - g) + (D - 1 - q);
            if (i < lc) t.ctaps[(size_t)g * D + q] = (float)c[i];
        }
    t.Gp = gp;
}

// k_ols_frame with the mixer folded into the response tables.  With the NCO's period R a divisor of 8 the mixed block is
// xs[b0 + i] = x[b0 + i] C e^{j 2 pi i numer / R}, C = p[(idx0 + b0) mod R] (the same for every block of a call: a block is 832 =
// 8 x 104 samples, a frame 16384), so the phases' transforms are those of the UNMIXED samples moved by s = 1024 numer / R bins
// (a multiple of 128) and turned by e^{j 2 pi p numer / R}:  X_p[k] = C e^{j 2 pi p numer / R} Xu_p[k - s].  With
//   G'_p[k] = G_p[(k + s) mod 256] e^{j 2 pi p numer / R}
// the kernel's sum over the phases is the true spectrum moved by s bins, and its inverse the true result times
// C (-1)^((s / 128) m): one product per result instead of one per sample, no table read, no phase arithmetic per block.
int rr_chain::ensure_mixfold() {
    const int64_t R = fs->denom;
    int64_t nu = fs->numer % R;
    if (nu < 0) nu += R;
    if (mix_numer == nu && mix_denom == R && mix_ctaps_fl == ctaps_fl && mix_ctaps_ds == ctaps_ds && mix_table_version == fs->table_version)
        return RR_OK;
    if (olsG64.size() != 1024) RR_FAIL(RR_ERR_BAD_ARG, "Chain: no polyphase tables to fold the mixer into");
    const size_t s = (size_t)((1024 * nu / R) % 256);  // R divides 8: whole
    // one table per phasor C = p[ph] the first sample of a call's blocks can meet (R <= 8 of them, 8 KiB each); the table's own
    // entries (as the kernel's mixer multiplies by them) in f64
    std::vector<float> gp(2 * 1024 * (size_t)R);
    const float *tab = reinterpret_cast<const float *>(fs->host_table.data());
    for (size_t ph = 0; ph < (size_t)R; ++ph) {
        const cd C((double)tab[2 * ph], (double)tab[2 * ph + 1]);
        for (size_t pp = 0; pp < 4; ++pp) {
            const double ang = 2.0 * M_PI * (double)((pp * (size_t)nu) % (size_t)R) / (double)R;
            const cd rot = cd(std::cos(ang), std::sin(ang)) * C;
            for (size_t k = 0; k < 256; ++k) {
                const cd g = olsG64[256 * pp + (k + s) % 256] * rot;
                const size_t l = k % 64, c = k / 64, i = 4 * pp + c, dst = ((i >> 1) * 64 + l) * 2 + (i & 1);
                gp[2 * (1024 * ph + dst)] = (float)g.real();
                gp[2 * (1024 * ph + dst) + 1] = (float)g.imag();
            }
        }
    }
    RR_TRY(upload(d_olsHmix, gp.data(), gp.size() * sizeof(float), stream));
    mix_sigma = (s / 128) & 1 ? -1.f : 1.f;
    mix_numer = nu;
    mix_denom = R;
    mix_ctaps_fl = ctaps_fl;
    mix_ctaps_ds = ctaps_ds;
    mix_table_version = fs->table_version;
    return RR_OK;
}

// NCO periods that divide 8 (the benchmark's fs / 8): the mixer folded into the tables (ensure_mixfold) - once the mixed-sample
// history in front of this call has been written under the table in use (RR_FRAME_MIXFOLD=0: never).  `back` = how far in front
// of e0 - V the call's first block starts (the blocks' hop, 832 samples, is a multiple of every such period).
int rr_chain::fold_mixer(FusedFirArgs &a, int64_t back, bool frame) {
    const char *env = std::getenv("RR_FRAME_MIXFOLD");  // (read per call: tests switch it within one process)
    const bool off = env && std::atoi(env) == 0;
    if (!off && a.D == 4 && fs->denom >= 1 && 8 % fs->denom == 0 && frame_table_version == fs->table_version && olsG64.size() == 1024) {
        RR_TRY(ensure_mixfold());
        // the table for the phasor of the blocks' first samples: ph0 = (idx0 + e0 - V - back) mod R
        int64_t ph = ((int64_t)a.idx0 + a.e0 - a.V - back) % (int64_t)fs->denom;
        if (ph < 0) ph += fs->denom;
        a.H = d_olsHmix.as<char>() + (size_t)ph * 1024 * 2 * sizeof(float);
        a.mixfold = true;
        a.sigma = mix_sigma;
    }
    else if (!off && (frame ? use_frame : true) && ols_poly && ols_N == 1024 && frame_table_version == fs->table_version && !ctaps_cc.empty()) {
        const char *eg = std::getenv("RR_FRAME_GENFOLD");  // (=0: the mixer in front of the transform, A/B runs and tests)
        if (!(eg && std::atoi(eg) == 0)) {
            RR_TRY(ensure_genfold());
            a.H = d_olsHgen.p;
            a.genfold = true;
        }
    }
    frame_table_version = fs->table_version;
    return RR_OK;
}

// k_ols_frame<.., GP>: every other NCO period - the mixer moved BEHIND the filter.  The phase table is p[t] = p0 w^t,
// w = e^{j 2 pi numer / denom}, so  sum_i c[i] x[t - i] p[t - i] = p[t] sum_i (c[i] w^-i) x[t - i]: the tables of the response
// c[i] w^-i (complex also where c is real), applied to the samples as they are; the kernel multiplies each result by the table's
// entry at its position.  w^-i from the reduced index (i numer mod denom) in f64.
int rr_chain::ensure_genfold() {
    const int64_t R = fs->denom;
    int64_t nu = fs->numer % R;
    if (nu < 0) nu += R;
    if (gen_numer == nu && gen_denom == R && gen_ctaps_fl == ctaps_fl && gen_ctaps_ds == ctaps_ds) return RR_OK;
    const size_t lc = ctaps_cc.size();
    std::vector<double> c(lc, 0.0);
    std::vector<cd> cc(lc);
    for (size_t i = 0; i < lc; ++i) {
        const int64_t ri = (int64_t)(((__int128)i * (__int128)nu) % (__int128)R);
        const double ang = -2.0 * M_PI * (double)ri / (double)R;
        cc[i] = ctaps_cc[i] * cd(std::cos(ang), std::sin(ang));
    }
    FusedFirTables t;
    build_fused_fir_tables(use_frame ? FK_OLSF : FK_OLSW, ds->sched.D, c, cc, t);  // (the same tables at 4 : 1: the frame kernel's blocks are k_ols_wave<4>'s)
    if (!t.poly) RR_FAIL(RR_ERR_BAD_ARG, "Chain: no polyphase tables for the mixer behind the filter");
    RR_TRY(upload(d_olsHgen, t.H.data(), t.H.size() * sizeof(float), stream));
    gen_numer = nu;
    gen_denom = R;
    gen_ctaps_fl = ctaps_fl;
    gen_ctaps_ds = ctaps_ds;
    return RR_OK;
}

// c = reverse(ir) (*) g in f64, cast to f32; tables by build_fused_fir_tables
int rr_chain::ensure_ctaps() {
    if (ctaps_fl == fl->design_version && ctaps_ds == ds->design_version) return RR_OK;
    const size_t n = fl->n, L = ds->L;
    const size_t lc = L + n - 1;
    std::vector<double> c(lc, 0.0);
    std::vector<cd> cc(lc, cd(0, 0));
    for (size_t j = 0; j < L; ++j) {
        const double a = ds->ir_f64[L - 1 - j];
        for (size_t k = 0; k < n; ++k) {
            c[j + k] += a * fl->taps_f64[k].real();
            cc[j + k] += a * (fl->real_taps ? cd(fl->taps_f64[k].real(), 0.0) : fl->taps_f64[k]);
        }
    }
    use_ols64 = false;
    if (dtype == RR_F64 && chain_ols64_ok(this, lc)) {
        std::vector<double> G, tw;
        ols64_tables(cc, G, tw);
        RR_TRY(upload(d_olsH, G.data(), G.size() * sizeof(double), stream));
        RR_TRY(upload(d_tw4096, tw.data(), tw.size() * sizeof(double), stream));
        ols_V = (int)ols4096_f64_overlap(lc, ds->sched.D);
        use_ols64 = true;
        use_poly64 = use_frame = use_ols = false;
        Lc = lc;
        ctaps_fl = fl->design_version;
        ctaps_ds = ds->design_version;
        return RR_OK;
    }
    const int fk = dtype == RR_F64 ? (int)FK_POLY : pick_fused_kernel(ds->sched.D, lc, fl->real_taps, p.fft_len);
    if (fk == FK_POLY) {
        // k_decim_poly(_f64)'s tap list for ir = reverse(c): out[m] = sum_j ir[j] xs[e_m - (Lc - 1) + j] = sum_i c[i] xs[e_m - i]
        std::vector<double> ir(lc);
        for (size_t j = 0; j < lc; ++j) ir[j] = c[lc - 1 - j];
        std::vector<uint32_t> T;
        const int64_t e0 = 0;
        int lp = 0;
        build_decim_poly_taps(ir, ds->sched.D, 1, &e0, T, &lp, dtype);
        RR_TRY(upload(d_ctaps, T.data(), T.size() * sizeof(uint32_t), stream));
        poly64_Lp = lp;
        use_poly64 = true;
        use_frame = use_ols = false;
        Lc = lc;
        ctaps_fl = fl->design_version;
        ctaps_ds = ds->design_version;
        return RR_OK;
    }
    use_poly64 = false;
    FusedFirTables t;
    build_fused_fir_tables(fk, ds->sched.D, c, cc, t);
    use_frame = fk == FK_OLSF;
    use_ols = fk != FK_DIRECT;
    if (fk == FK_OLSF || fk == FK_OLSW) ctaps_cc = cc; else ctaps_cc.clear();
    if (use_ols) {
        RR_TRY(upload(d_olsH, t.H.data(), t.H.size() * sizeof(float), stream));
        RR_TRY(upload(d_tw4096, t.tw.data(), t.tw.size() * sizeof(float), stream));
        ols_V = t.V;
        ols_poly = t.poly;
        ols_N = t.N;
        ols_blk = t.blk;
        olsG64.swap(t.G64);
    } else {
        RR_TRY(upload(d_ctaps, t.ctaps.data(), t.ctaps.size() * sizeof(float), stream));
        Gp = t.Gp;
    }
    Lc = lc;
    ctaps_fl = fl->design_version;
    ctaps_ds = ds->design_version;
    return RR_OK;
}

// After fused calls the Filter's previous chunk and the Downsampler's ring are
// stale; rebuild both from the mixed-sample history before anything reads them.
int rr_chain::materialize() {
    if (pend_ptr) {
        RR_TRY(select());
        if (pending_len)
            RR_HIP(hipMemcpyAsync(pending.p, pend_ptr, pending_len * elem_size(dtype), hipMemcpyDeviceToDevice, stream));
        pend_ptr = nullptr;
    }
    if (!blocks_stale) return RR_OK;
    RR_TRY(select());
    const size_t esz = elem_size(dtype), n = fl->n, L = ds->L;
    const size_t fed = HX - carry_len;  // xh[0 .. fed) went through the Filter, the rest is the carry
    // previous_chunk = the last n samples the Filter saw
    RR_HIP(hipMemcpyAsync(fl->hist[fl->cur].p, xh[xh_cur].as<char>() + (fed - n) * esz, n * esz, hipMemcpyDeviceToDevice, stream));
    // ring buffer = the last L Filter outputs, recomputed from the same samples
    FirArgs a;
    a.in = xh[xh_cur].p;
    a.n_in = HX;
    a.taps = fl->d_taps.p;
    a.K = (uint32_t)n;
    a.complex_taps = !fl->real_taps;
    a.out = ds->hist[ds->cur].p;
    a.n_out = L;
    a.e0 = fed - L;
    a.D = 1;
    RR_TRY(launch_fir(dtype, stream, a));
    blocks_stale = false;
    return RR_OK;
}

int rr_chain::process_fused(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    const size_t esz = elem_size(dtype), nf = p.filter_len, LF = p.fft_len;
    RR_TRY(fs->prepare(sample_rate));  // picks up set_shift (transform.rs:318-340)
    RR_TRY(ensure_ctaps());
    const size_t total = carry_len + n_in, whole = total / nf * nf, left = total - whole;
    const size_t dec = ds->sched.count(whole);
    const size_t have = pending_len + dec;
    const size_t nfr = have / LF, rest = have - nfr * LF;
    const bool split = (LF == 4096) && dtype == RR_F32;  // k_fft4096 reads [pending | new] from two places: no copies
    const bool store = !sink.on || sink.store;
    if (!split && pend_ptr) RR_TRY(materialize());
    FusedFirArgs a;
    a.xh = xh[xh_cur].p;
    a.hx = HX;
    a.in = d_in;
    a.n_in = n_in;
    a.nco = fs->d_table.p;
    a.denom = (uint32_t)fs->denom;
    a.idx0 = (uint32_t)fs->phase_idx;
    a.taps = d_ctaps.p;
    a.Gp = Gp;
    a.n_out = dec;
    a.e0 = (int64_t)ds->sched.first_emit() - (int64_t)carry_len;
    a.D = (uint32_t)ds->sched.D;
    a.xh_out = xh[xh_cur ^ 1].p;  // written by the kernel's last workgroup
    // (calls below 2^23 samples: k_ols_wave + k_fft4096 below - a workgroup of the frame kernel runs five blocks per wave in a
    //  row, 25 us even for one frame, where the two kernels take 12; from 2^24 samples on the frame kernel is ahead)
    const char *fke = std::getenv("RR_FUSED_KERNEL");  // (tests force the frame kernel on short streams)
    const bool frame_forced = fke && !std::strcmp(fke, "olsf");
    if (use_frame && !(LF != 4096 && sink.on) && n_in >= (frame_forced ? (size_t)1024 : (size_t)1 << 23)) {
        // one kernel: FIR stage + Fourier; the decimated samples stay on chip, only the unfinished
        // frame goes to a small pending buffer
        if (store && nfr * LF > cap) RR_FAIL(RR_ERR_CAPACITY, "Chain: out_cap %zu < %zu", cap, nfr * LF);
        RR_TRY(fo->prepare(LF));
        RR_TRY(pendbuf[0].reserve(LF * esz));
        RR_TRY(pendbuf[1].reserve(LF * esz));
        const void *pin = pend_ptr ? pend_ptr : pending.p;
        const int po = (pin == pendbuf[pb_cur ^ 1].p) ? pb_cur : (pb_cur ^ 1);
        a.H = d_olsH.p;
        a.tw4096 = d_tw4096.p;
        a.V = ols_V;
        a.poly = ols_poly;
        a.blk = ols_blk;
        RR_TRY(fold_mixer(a, 4 * (int64_t)pending_len, true));  // (the frame's first block starts 4 pl samples earlier, as launch_ols_frame)
        // (the launch records its own start / end: marker packets would cost ~4 us of stream time each)
        if (timers.on && !sink.on) timers.begin_ext(ST_FUSED_FIR, &a.ev_start, &a.ev_stop);
        const rr::FrameMeter fmv = sink.frame_meter();
        RR_TRY(launch_ols_frame(stream, a, pin, pending_len, pendbuf[po].p, d_out, fo->d_window.p, fo->d_tw.p,
                                fo->center_dc, sink.on ? &fmv : nullptr, LF));
        xh_cur ^= 1;
        if (left) RR_HIP(hipMemcpyAsync(carry.p, xh[xh_cur].as<char>() + (HX - left) * esz, left * esz, hipMemcpyDeviceToDevice, stream));
        const uint64_t den0 = (uint64_t)fs->denom;
        fs->phase_idx = (fs->phase_idx + n_in % den0) % den0;
        carry_len = left;
        ds->sched.advance(whole, nullptr);
        zrun += whole;
        blocks_stale = true;
        pend_ptr = pendbuf[po].p;
        pb_cur = po;
        pending_len = rest;
        last_fused = a.genfold ? 8 : a.mixfold ? 6 : FK_OLSF;  // (6: k_ols_frame<true>, the mixer folded into the tables; 8: <.., GP>, behind the filter)
        if (n_out) *n_out = nfr * LF;
        return RR_OK;
    }
    char *newv = nullptr;
    char *dbase = nullptr;
    // a call that completes no frame, through the wave kernels (8-byte stores): the decimated samples go straight behind the pending
    // ones in the chain's own buffer instead of through a copy (chunks of a few thousand samples: most calls)
    const bool append = split && nfr == 0 && dec > 0 && use_ols && ols_N == 1024 && !use_ols64 && !use_poly64;
    if (split) {
        DevBuf &buf = dec2[dec_cur ^ 1];  // never the buffer the pending samples live in
        RR_TRY(buf.reserve((dec + 2) * esz));
        newv = buf.as<char>();
        if (append) {
            if (pend_ptr) {
                if (pending_len) RR_HIP(hipMemcpyAsync(pending.p, pend_ptr, pending_len * esz, hipMemcpyDeviceToDevice, stream));
                pend_ptr = nullptr;
            }
            newv = pending.as<char>() + pending_len * esz;
        }
    } else {
        // pending outputs in front of the new ones; shifted by one sample when needed so
        // that the kernel's 16-byte stores of the new outputs are aligned
        const size_t off = pending_len & 1;
        RR_TRY(decim.reserve((have + off + 1) * esz));
        dbase = decim.as<char>() + off * esz;
        if (pending_len) RR_HIP(hipMemcpyAsync(dbase, pending.p, pending_len * esz, hipMemcpyDeviceToDevice, stream));
        newv = dbase + pending_len * esz;
    }
    a.out = newv;
    a.H = d_olsH.p;
    a.tw4096 = d_tw4096.p;
    a.V = ols_V;
    a.poly = ols_poly;
    a.blk = ols_blk;
    // k_ols_wave + k_fft4096: the launches record their own start / end (no marker packets, which
    // cost ~4 us of stream time each); the other kernels are bracketed by recorded events
    const bool ext = timers.on && use_ols && ols_N == 1024 && split && dec > 0 && !sink.on;
    int tk = -1;
    if (ext)
        timers.begin_ext(ST_FUSED_FIR, &a.ev_start, &a.ev_stop);
    else
        tk = timers.begin(ST_FUSED_FIR, stream);
    if (use_ols64) {
        if (dec == 0) RR_FAIL(RR_ERR_BAD_ARG, "Chain: a call through k_ols4096_f64 must produce output");
        RR_TRY(launch_ols4096_f64(stream, a.xh, HX, d_in, n_in, d_olsH.p, d_tw4096.p, (size_t)ols_V, ds->sched.D, newv, dec, (long)a.e0,
                                  a.xh_out, HX, a.nco, a.denom, a.idx0));
    } else if (use_poly64) {
        RR_TRY(launch_decim_poly(stream, a.xh, HX, d_in, n_in, d_ctaps.p, ds->sched.D, 1, poly64_Lp, Lc, a.e0, newv, dec, a.xh_out, HX,
                                 a.nco, a.denom, a.idx0, dtype));
        if (dec == 0)  // (no output, no tile: the history by a launch of its own)
            RR_FAIL(RR_ERR_BAD_ARG, "Chain: a call through the polyphase decimator must produce output");
    } else if (use_ols && ols_N == 1024) {
        if (ols_poly) RR_TRY(fold_mixer(a, 0));
        RR_TRY(launch_ols_wave(stream, a));
    } else if (use_ols)
        RR_TRY(launch_ols_decim(stream, a));
    else
        RR_TRY(launch_fused_fir(stream, a));
    xh_cur ^= 1;
    if (left) RR_HIP(hipMemcpyAsync(carry.p, xh[xh_cur].as<char>() + (HX - left) * esz, left * esz, hipMemcpyDeviceToDevice, stream));
    const uint64_t den = (uint64_t)fs->denom;
    fs->phase_idx = (fs->phase_idx + n_in % den) % den;
    carry_len = left;
    ds->sched.advance(whole, nullptr);
    zrun += whole;
    blocks_stale = true;
    // Fourier on whole frames, the rest stays pending (resampling.rs:121-131)
    size_t wrote = 0;
    hipEvent_t fa = nullptr, fb = nullptr;
    if (ext) {
        if (nfr) timers.begin_ext(ST_FOURIER, &fa, &fb);
    } else {
        tk = timers.next(tk, ST_FOURIER, stream);  // (the carry copy above, if any, counts for the FIR stage)
    }
    if (split) {
        if (store && nfr * LF > cap) RR_FAIL(RR_ERR_CAPACITY, "Chain: out_cap %zu < %zu", cap, nfr * LF);
        RR_TRY(fo->prepare(LF));
        const void *head = pend_ptr ? pend_ptr : pending.p;
        if (sink.on) {
            fo->stream = stream;
            RR_TRY(fo->transform_metered_dev(head, pending_len, newv, d_out, 4096, nfr, sink.frame_meter()));
        } else
        RR_TRY(launch_fft4096(stream, head, pending_len, newv, d_out, nfr, fo->d_window.p, fo->d_tw.p, fo->center_dc, 4096,
                              fa, fb));
        wrote = nfr * LF;
        if (nfr) {  // the leftover is the tail of the new outputs
            pend_ptr = newv + (nfr * LF - pending_len) * esz;
            dec_cur ^= 1;
        } else if (dec && !append) {
            // no frame completed: append the new outputs to the pending chunk
            RR_TRY(materialize_pending_append(newv, dec));
        }
    } else {
        if (sink.on) {
            RR_TRY(fo->prepare(LF));
            RR_TRY(fo->transform_metered_dev(nullptr, 0, dbase, d_out, LF, nfr, sink.frame_meter()));
            wrote = nfr * LF;
        } else
        RR_TRY(fo->process_dev(LF, dbase, nfr * LF, d_out, cap, &wrote));
        if (rest) RR_HIP(hipMemcpyAsync(pending.p, dbase + nfr * LF * esz, rest * esz, hipMemcpyDeviceToDevice, stream));
    }
    if (!ext) timers.end(tk, stream);
    pending_len = rest;
    last_fused = use_ols64 ? (int)FK_OLS64 : use_poly64 ? FK_POLY : use_ols ? (ols_N == 1024 ? (a.genfold ? 9 : a.mixfold ? 7 : FK_OLSW) : FK_OLS) : FK_DIRECT;  // (7: k_ols_wave<4, true, true>)
    if (n_out) *n_out = wrote;
    return RR_OK;
}

// ---- lockstep banks ------------------------------------------------------------------------------------------------
rr_chain::BankSig rr_chain::bank_signature() const {
    BankSig g{};
    g.phase_idx = fs->phase_idx;
    g.zrun = zrun;
    g.sched_phase = ds->sched.phase;
    g.fs_version = fs->table_version;
    g.frame_version = frame_table_version;
    g.ctaps_fl = ctaps_fl;
    g.ctaps_ds = ctaps_ds;
    g.carry_len = carry_len;
    g.pending_len = pending_len;
    g.HX = HX;
    g.xh_count = xh_count;
    g.Lc = Lc;
    g.sched_pos = ds->sched.pos;
    g.rate = ds->prev_rate;
    g.xh_cur = xh_cur;
    g.dec_cur = dec_cur;
    g.hist_valid = fl->hist_valid ? 1 : 0;
    g.use_frame = use_frame ? 1 : 0;
    g.ols_N = ols_N;
    g.ols_poly = ols_poly ? 1 : 0;
    g.pend_in_dec = pend_ptr ? (pend_ptr == pendbuf[0].p || pend_ptr == pendbuf[1].p ? 2 : 1) : 0;
    return g;
}

// The host half of process_dev + process_fused for the two-kernel step of a whole-chunk call, without a launch: *ok = false
// means "this call is not such a step" (the caller then drives the lanes one by one).
int rr_chain::bank_plan(double sample_rate, size_t n_in, size_t cap, BankStep &st, bool *ok) {
    *ok = false;
    if (fl->needs_design(sample_rate, p.filter_len)) return RR_OK;
    if (sink.on || timers.on || dtype != RR_F32 || p.fft_len != 4096) return RR_OK;
    const bool fused = fused_candidate(sample_rate) && HX != 0 && xh_count >= HX && fl->hist_valid && zrun + 1 >= ds->L && n_in >= HX;
    if (!fused || carry_len != 0 || n_in % p.filter_len != 0 || n_in > 0xfffffff0ull) return RR_OK;
    RR_TRY(select());
    RR_TRY(fs->prepare(sample_rate));
    RR_TRY(ensure_ctaps());
    if (!(use_ols && ols_N == 1024 && ols_poly && (ols_blk == 1024 || ols_blk == 2048))) return RR_OK;  // (k_ols_wg has no bank form)
    const char *fke = std::getenv("RR_FUSED_KERNEL");
    const bool frame_forced = fke && !std::strcmp(fke, "olsf");
    if (use_frame && n_in >= (frame_forced ? (size_t)1024 : (size_t)1 << 23)) return RR_OK;  // (the frame kernel's calls: lane by lane)
    st.whole = n_in;
    st.dec = ds->sched.count(n_in);
    const size_t have = pending_len + st.dec;
    st.nfr = have / 4096;
    st.rest = have - st.nfr * 4096;
    st.n_head = pending_len;
    if (st.dec == 0) return RR_OK;
    // (no frame completes: the step's one kernel appends to the pending chunk in the chain's own buffer - rr_chainbank::process_dev)
    if (st.nfr * 4096 > cap) RR_FAIL(RR_ERR_CAPACITY, "Chain: out_cap %zu < %zu", cap, st.nfr * 4096);
    RR_TRY(fo->prepare(4096));
    rr::FusedFirArgs &a = st.a;
    a = rr::FusedFirArgs{};
    a.hx = HX;
    a.n_in = n_in;
    a.nco = fs->d_table.p;
    a.denom = (uint32_t)fs->denom;
    a.idx0 = (uint32_t)fs->phase_idx;
    a.n_out = st.dec;
    a.e0 = (int64_t)ds->sched.first_emit();
    a.D = (uint32_t)ds->sched.D;
    a.H = d_olsH.p;
    a.tw4096 = d_tw4096.p;
    a.V = ols_V;
    a.poly = ols_poly;
    a.blk = ols_blk;
    RR_TRY(fold_mixer(a, 0));
    *ok = true;
    return RR_OK;
}

int rr_chain::bank_pointers(const BankStep &st, const void *d_in, void *d_out, rr::BankPtrs &bp) {
    DevBuf &buf = dec2[dec_cur ^ 1];  // never the buffer the pending samples live in
    RR_TRY(buf.reserve((st.dec + 2) * elem_size(dtype)));
    bp.xh = xh[xh_cur].p;
    bp.in = d_in;
    bp.dec = buf.p;
    if (st.nfr == 0) bp.dec = pending.as<char>() + pending_len * elem_size(dtype);  // (behind the pending samples: pending_len + dec < 4096)
    bp.xh_out = xh[xh_cur ^ 1].p;
    bp.head = pend_ptr ? pend_ptr : pending.p;
    bp.out = d_out;
    return RR_OK;
}

// (what process_fused does behind its two launches)
void rr_chain::bank_commit(const BankStep &st, size_t n_in) {
    const size_t esz = elem_size(dtype);
    char *newv = dec2[dec_cur ^ 1].as<char>();
    xh_cur ^= 1;
    const uint64_t den = (uint64_t)fs->denom;
    fs->phase_idx = (fs->phase_idx + n_in % den) % den;
    carry_len = 0;
    ds->sched.advance(st.whole, nullptr);
    zrun += st.whole;
    blocks_stale = true;
    if (st.nfr == 0) {
        pend_ptr = nullptr;  // (the pending chunk - copied there by the bank if it was not - and the new samples behind it)
    } else {
        pend_ptr = newv + (st.nfr * 4096 - pending_len) * esz;
        dec_cur ^= 1;
    }
    pending_len = st.rest;
    frame_table_version = fs->table_version;
    last_fused = st.a.genfold ? 9 : st.a.mixfold ? 7 : FK_OLSW;
}

rr_chainbank::~rr_chainbank() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (rr_chain *c : lanes) delete c;
}

int rr_chainbank::process_dev(double rate, const void *d_in, size_t in_stride, size_t n_in, void *d_out, size_t out_stride,
                              size_t cap, size_t *n_out) {
    if (n_out) *n_out = 0;
    const size_t K = lanes.size();
    if (K == 0) return RR_OK;
    if (n_in > in_stride && K > 1) RR_FAIL(RR_ERR_BAD_ARG, "ChainBank: %zu samples per channel, channels %zu apart", n_in, in_stride);
    const size_t esz = elem_size(dtype);
    RR_TRY(select());
    last_path = 0;
    rr_chain::BankStep st;
    bool ok = false;
    // (the plan's one side effect on the lane - fold_mixer notes the table its history is written under - is taken back when
    //  the step does not run in lockstep: the lane's own call decides again, exactly as a stand-alone chain would)
    const uint64_t frame_version0 = lanes[0]->frame_table_version;
    RR_TRY(lanes[0]->bank_plan(rate, n_in, cap, st, &ok));
    if (ok && st.nfr * 4096 > out_stride && K > 1) RR_FAIL(RR_ERR_CAPACITY, "ChainBank: %zu bins per channel, channels %zu apart", st.nfr * 4096, out_stride);
    if (ok) {
        // every lane at the same stream position with the same tables: lane 0's launch parameters are everybody's.  The full
        // comparison runs when somebody has touched a lane since the bank last saw them agree (the lanes count their mutating
        // entry points); in a steady stream of bank calls it is one comparison per lane
        if (seen.size() != K) seen.assign(K, ~0ull), verified = false;
        bool touched = !verified;
        for (size_t k = 0; k < K && !touched; ++k) touched = lanes[k]->mutations != seen[k];
        if (touched) {
            verified = false;
            const rr_chain::BankSig g0 = lanes[0]->bank_signature();
            for (size_t k = 1; k < K && ok; ++k) {
                rr_chain *c = lanes[k];
                if (c->fs->shift_changed || !c->fs->have_rate || c->fs->prev_rate != rate || c->fl->needs_design(rate, c->p.filter_len) ||
                    c->sink.on || c->timers.on) {
                    ok = false;
                    break;
                }
                const rr_chain::BankSig g = c->bank_signature();
                ok = std::memcmp(&g, &g0, sizeof g) == 0;
            }
        }
    }
    if (!ok) {
        // lane by lane (stream start, after an interrupt or a retune, ragged calls): every lane is a chain of its own
        lanes[0]->frame_table_version = frame_version0;
        verified = false;
        size_t got = 0;
        for (size_t k = 0; k < K; ++k) {
            size_t w = 0;
            RR_TRY(lanes[k]->process_dev(rate, static_cast<const char *>(d_in) + k * in_stride * esz, n_in,
                                         static_cast<char *>(d_out) + k * out_stride * esz, cap, &w));
            if (k == 0) got = w;
            else if (w != got) RR_FAIL(RR_ERR_BAD_ARG, "ChainBank: the channels have left lockstep (%zu against %zu bins)", w, got);
        }
        if (n_out) *n_out = got;
        return RR_OK;
    }
    // the channels' buffers travel in the launches' argument blocks, 64 channels per launch
    rr_chain *c0 = lanes[0];
    for (size_t k0 = 0; k0 < K; k0 += rr::kBankGroup) {
        const size_t G = K - k0 < rr::kBankGroup ? K - k0 : rr::kBankGroup;
        rr::BankTable tab;
        for (size_t k = 0; k < G; ++k)
            RR_TRY(lanes[k0 + k]->bank_pointers(st, static_cast<const char *>(d_in) + (k0 + k) * in_stride * esz,
                                                static_cast<char *>(d_out) + (k0 + k) * out_stride * esz, tab.c[k]));
        if (st.nfr == 0) {
            // no frame completes: the new samples go behind the pending ones in the chains' own buffers - where those are moved
            // first if the last step left them in its output (one launch for the group's channels)
            if (c0->pend_ptr && st.n_head) {
                rr::BankTable cp = tab;
                for (size_t k = 0; k < G; ++k) cp.c[k].out = lanes[k0 + k]->pending.p;  // (head = pend_ptr: bank_pointers)
                RR_TRY(launch_bank_copy(stream, cp, G, st.n_head));
            }
            RR_TRY(launch_ols_wave_bank(stream, st.a, tab, G));
            continue;
        }
        RR_TRY(launch_ols_wave_bank(stream, st.a, tab, G));
        RR_TRY(launch_fft4096_bank(stream, tab, G, st.n_head, st.nfr, c0->fo->d_window.p, c0->fo->d_tw.p, c0->fo->center_dc));
    }
    for (size_t k = 0; k < K; ++k) {
        lanes[k]->bank_commit(st, n_in);
        seen[k] = lanes[k]->mutations;
    }
    verified = true;
    last_path = 1;
    if (n_out) *n_out = st.nfr * 4096;
    return RR_OK;
}

// fewer than fft_len outputs in total: gather [pending | new] into the `pending` buffer
int rr_chain::materialize_pending_append(const void *newv, size_t dec) {
    const size_t esz = elem_size(dtype);
    if (pend_ptr) {
        if (pending_len)
            RR_HIP(hipMemcpyAsync(pending.p, pend_ptr, pending_len * esz, hipMemcpyDeviceToDevice, stream));
        pend_ptr = nullptr;
    }
    RR_HIP(hipMemcpyAsync(pending.as<char>() + pending_len * esz, newv, dec * esz, hipMemcpyDeviceToDevice, stream));
    return RR_OK;
}

int rr_chain::process_generic(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    if (blocks_stale || pend_ptr) RR_TRY(materialize());
    const size_t esz = elem_size(dtype);
    // 1. FreqShifter -> mixed[carry_len ..)
    const size_t total = carry_len + n_in;
    RR_TRY(mixed.reserve((total ? total : 1) * esz));
    if (carry_len) RR_HIP(hipMemcpyAsync(mixed.p, carry.p, carry_len * esz, hipMemcpyDeviceToDevice, stream));
    size_t got = 0;
    int tk = timers.begin(ST_FREQSHIFT, stream);
    RR_TRY(fs->process_dev(sample_rate, d_in, n_in, mixed.as<char>() + carry_len * esz, n_in, &got));
    timers.end(tk, stream);
    // 2. Filter on whole chunks
    const size_t nf = p.filter_len;
    const size_t whole = total / nf * nf;
    size_t filt = fl->peek(whole);
    RR_TRY(filtered.reserve((filt ? filt : 1) * esz));
    tk = timers.begin(ST_FILTER, stream);
    RR_TRY(fl->process_dev(sample_rate, mixed.p, whole, filtered.p, filt, &filt));
    timers.end(tk, stream);
    zrun += filt;
    // 3. Downsampler -> decim[pending_len ..)
    size_t dec = 0;
    RR_TRY(ds->peek(sample_rate, filt, &dec));
    const size_t have = pending_len + dec;
    RR_TRY(decim.reserve((have ? have : 1) * esz));
    if (pending_len) RR_HIP(hipMemcpyAsync(decim.p, pending.p, pending_len * esz, hipMemcpyDeviceToDevice, stream));
    tk = timers.begin(ST_DECIM, stream);
    RR_TRY(ds->process_dev(sample_rate, filtered.p, filt, decim.as<char>() + pending_len * esz, dec, &dec));
    timers.end(tk, stream);
    // keep the mixed-sample history the fused kernels start from
    if (fused_candidate(sample_rate)) {
        RR_TRY(ensure_xh());
        RR_TRY(launch_update_hist(dtype, stream, xh[xh_cur].p, xh[xh_cur ^ 1].p, HX, mixed.as<char>() + carry_len * esz, n_in));
        xh_cur ^= 1;
        xh_count = (xh_count + n_in > HX) ? HX : xh_count + n_in;
    }
    const size_t left = total - whole;
    if (left) RR_HIP(hipMemcpyAsync(carry.p, mixed.as<char>() + whole * esz, left * esz, hipMemcpyDeviceToDevice, stream));
    carry_len = left;
    // 4. Fourier on whole frames
    const size_t L = p.fft_len;
    const size_t nfr = have / L;
    size_t wrote = 0;
    tk = timers.begin(ST_FOURIER, stream);
    if (sink.on) {
        RR_TRY(fo->prepare(L));
        RR_TRY(fo->transform_metered_dev(nullptr, 0, decim.p, d_out, L, nfr, sink.frame_meter()));
        wrote = nfr * L;
    } else
    RR_TRY(fo->process_dev(L, decim.p, nfr * L, d_out, cap, &wrote));
    timers.end(tk, stream);
    const size_t rest = have - nfr * L;
    if (rest) RR_HIP(hipMemcpyAsync(pending.p, decim.as<char>() + nfr * L * esz, rest * esz, hipMemcpyDeviceToDevice, stream));
    pending_len = rest;
    last_fused = 0;
    if (n_out) *n_out = wrote;
    return RR_OK;
}

int rr_chain::process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    ++mutations;
    if (n_out) *n_out = 0;
    if (fl->needs_design(sample_rate, p.filter_len))
        RR_FAIL(RR_ERR_NEED_DESIGN, "Chain: Filter has no design for sample rate %g", sample_rate);
    size_t frames = 0;
    RR_TRY(peek(sample_rate, n_in, &frames));
    if ((!sink.on || sink.store) && frames * p.fft_len > cap) RR_FAIL(RR_ERR_CAPACITY, "Chain: out_cap %zu < %zu", cap, frames * p.fft_len);
    if (sink.on && frames > sink.cap) RR_FAIL(RR_ERR_CAPACITY, "Chain: room for %zu bandwidths, the call makes %zu spectra", sink.cap, frames);
    if (n_in > 0xfffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Chain: more than 2^32 samples in one call");
    RR_TRY(select());
    // The fused kernels apply when every output of this call is in steady state:
    // the Filter has its previous chunk, the Downsampler's window holds only real
    // Filter outputs of the current contiguous run, and the mixed history is filled.
    const bool fused = fused_candidate(sample_rate) && HX != 0 && xh_count >= HX && fl->hist_valid &&
                       zrun + 1 >= ds->L && n_in >= HX;
    if (fused) return process_fused(sample_rate, d_in, n_in, d_out, cap, n_out);
    return process_generic(sample_rate, d_in, n_in, d_out, cap, n_out);
}

extern "C" {

// ---- Chain ------------------------------------------------------------------------------
int rr_chain_create(const rr_chain_params *p, int device, rr_chain **out) {
    RR_GUARD_BEGIN
    if (!out || !p) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    if (p->filter_len == 0 || p->fft_len == 0) RR_FAIL(RR_ERR_BAD_ARG, "Chain: filter_len and fft_len must be > 0");
    if (p->fft_window.kind != RR_WIN_RECTANGULAR && p->fft_window.kind != RR_WIN_KAISER)
        RR_FAIL(RR_ERR_BAD_ARG, "Chain: fft_window must be a built-in window");
    if (!(p->output_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "output sample rate must be positive");
    if (!(p->bandwidth >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be positive");
    if (!(p->bandwidth < p->output_rate)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be smaller than output sample rate");
    RR_TRY(fourier_supported(p->dtype, p->fft_len));
    auto *h = new rr_chain;
    int s = h->init_base(K_CHAIN, p->dtype, device);
    if (s != RR_OK) {
        delete h;
        return s;
    }
    h->p = *p;
    auto sub = [&](rr_block *b, int kind) {
        b->kind = kind;
        b->dtype = p->dtype;
        b->device = device;
        b->stream = h->stream;  // shares the chain's stream; owns none
    };
    h->fs = new rr_freqshifter;
    sub(h->fs, K_FREQSHIFTER);
    h->fs->precision = p->precision;
    h->fs->shift = p->shift;
    h->fl = new rr_filter;
    sub(h->fl, K_FILTER);
    h->ds = new rr_downsampler;
    sub(h->ds, K_DOWNSAMPLER);
    h->ds->output_rate = p->output_rate;
    h->ds->bandwidth = p->bandwidth;
    h->ds->quality = p->quality;
    h->fo = new rr_fourier;
    sub(h->fo, K_FOURIER);
    h->fo->window = p->fft_window;
    h->fo->center_dc = p->center_dc != 0;
    const size_t esz = elem_size(p->dtype);
    s = h->carry.reserve(p->filter_len * esz);
    if (s == RR_OK) s = h->pending.reserve(p->fft_len * esz);
    if (s != RR_OK) {
        delete h;
        return s;
    }
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_chain_set_shift(rr_chain *h, double shift) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    ++h->mutations;
    h->fs->shift = shift;
    h->fs->shift_changed = true;
    return RR_OK;
}
int rr_chain_filter_needs_design(const rr_chain *h, double sample_rate, int *needed) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    if (!needed) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *needed = h->fl->needs_design(sample_rate, h->p.filter_len) ? 1 : 0;
    return RR_OK;
}
int rr_chain_filter_mark_params_changed(rr_chain *h) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    ++h->mutations;
    h->fl->params_changed = true;
    return RR_OK;
}
int rr_chain_filter_design(rr_chain *h, double sample_rate, const rr_c64 *resp, const double *window_rel) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_CHAIN);
    ++h->mutations;
    // the Downsampler keeps running across a Filter redesign: give it its ring back first
    RR_TRY(h->materialize());
    if (h->fl->designed && sample_rate != h->fl->rate && h->carry_len) {
        // the Rechunker drops a patchwork of another sample rate (chunks.rs:72-79); those samples were mixed with the
        // old NCO table and must not be prepended to the new-rate stream
        if (h->HX) {
            RR_TRY(h->select());
            RR_TRY(launch_drop_tail(h->stream, h->xh[h->xh_cur].p, h->xh[h->xh_cur ^ 1].p, h->HX, h->carry_len));
            h->xh_cur ^= 1;
            h->xh_count = h->xh_count > h->carry_len ? h->xh_count - h->carry_len : 0;
        }
        h->carry_len = 0;
    }
    RR_TRY(h->fl->design(sample_rate, h->p.filter_len, resp, window_rel));
    h->zrun = 0;
    return RR_OK;
    RR_GUARD_END
}
int rr_chain_interrupt(rr_chain *h) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_CHAIN);
    ++h->mutations;
    // The Rechunker in front of the Filter drops its patchwork (chunks.rs:80-88)
    // and the Filter its previous chunk (filters.rs:262-265); the other blocks
    // only forward the event.
    RR_TRY(h->materialize());
    if (h->HX && h->carry_len) {
        RR_TRY(h->select());
        RR_TRY(launch_drop_tail(h->stream, h->xh[h->xh_cur].p, h->xh[h->xh_cur ^ 1].p, h->HX, h->carry_len));
        h->xh_cur ^= 1;
        h->xh_count = h->xh_count > h->carry_len ? h->xh_count - h->carry_len : 0;
    }
    h->carry_len = 0;
    h->fl->hist_valid = false;
    h->zrun = 0;
    return RR_OK;
    RR_GUARD_END
}
int rr_chain_pending(const rr_chain *h, size_t *n) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    if (!n) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *n = h->carry_len;
    return RR_OK;
}
int rr_chain_peek(rr_chain *h, double sample_rate, size_t n_in, size_t *n_frames) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    if (!n_frames) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return h->peek(sample_rate, n_in, n_frames);
}
int rr_chain_process_dev(rr_chain *h, double rate, const void *d_in, size_t n_in, void *d_out, size_t cap,
                         size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_CHAIN);
    return h->process_dev(rate, d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
static int chain_host(rr_chain *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out,
                      bool blocking) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_CHAIN);
    if (n_out) *n_out = 0;
    if (h->fl->needs_design(rate, h->p.filter_len))
        RR_FAIL(RR_ERR_NEED_DESIGN, "Chain: Filter has no design for sample rate %g", rate);
    size_t frames = 0;
    RR_TRY(h->peek(rate, n_in, &frames));
    const size_t need = frames * h->p.fft_len;
    if (need > cap) RR_FAIL(RR_ERR_CAPACITY, "Chain: out_cap %zu < %zu", cap, need);
    size_t got = 0;
    RR_TRY(host_io(h, in, n_in, out, need, blocking, [&](void *di, void *dout, size_t *p) {
        int s = h->process_dev(rate, di, n_in, dout, need, p);
        got = *p;
        return s;
    }));
    if (n_out) *n_out = got;
    return RR_OK;
    RR_GUARD_END
}
int rr_chain_process(rr_chain *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    return chain_host(h, rate, in, n_in, out, cap, n_out, true);
}
int rr_chain_enqueue(rr_chain *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    return chain_host(h, rate, in, n_in, out, cap, n_out, false);
}
// ---- rr_chainbank ----
int rr_chainbank_create(const rr_chain_params *p, size_t channels, int device, rr_chainbank **out) {
    RR_GUARD_BEGIN
    if (!out || !p) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    if (channels == 0 || channels > 65535) RR_FAIL(RR_ERR_BAD_ARG, "ChainBank: 1 .. 65535 channels");
    auto *b = new rr_chainbank;
    int s = b->init_base(K_CHAINBANK, p->dtype, device);
    for (size_t k = 0; s == RR_OK && k < channels; ++k) {
        rr_chain *c = nullptr;
        s = rr_chain_create(p, device, &c);
        if (s == RR_OK) {
            chain_use_stream(c, b->stream);
            b->lanes.push_back(c);
        }
    }
    if (s != RR_OK) {
        delete b;
        return s;
    }
    *out = b;
    return RR_OK;
    RR_GUARD_END
}
int rr_chainbank_channels(const rr_chainbank *h, size_t *channels) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    if (!channels) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *channels = h->lanes.size();
    return RR_OK;
}
int rr_chainbank_channel(rr_chainbank *h, size_t k, rr_chain **lane) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    if (!lane || k >= h->lanes.size()) RR_FAIL(RR_ERR_BAD_ARG, "ChainBank: no channel %zu", k);
    *lane = h->lanes[k];
    return RR_OK;
}
int rr_chainbank_set_shift(rr_chainbank *h, double shift) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    for (rr_chain *c : h->lanes) RR_TRY(rr_chain_set_shift(c, shift));
    return RR_OK;
}
int rr_chainbank_filter_needs_design(const rr_chainbank *h, double sample_rate, int *needed) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    return rr_chain_filter_needs_design(h->lanes[0], sample_rate, needed);
}
int rr_chainbank_filter_mark_params_changed(rr_chainbank *h) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    for (rr_chain *c : h->lanes) RR_TRY(rr_chain_filter_mark_params_changed(c));
    return RR_OK;
}
int rr_chainbank_filter_design(rr_chainbank *h, double sample_rate, const rr_c64 *resp, const double *window_rel) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    for (rr_chain *c : h->lanes) RR_TRY(rr_chain_filter_design(c, sample_rate, resp, window_rel));
    return RR_OK;
}
int rr_chainbank_interrupt(rr_chainbank *h) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    for (rr_chain *c : h->lanes) RR_TRY(rr_chain_interrupt(c));
    return RR_OK;
}
int rr_chainbank_peek(rr_chainbank *h, double sample_rate, size_t n_in, size_t *n_frames) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    return rr_chain_peek(h->lanes[0], sample_rate, n_in, n_frames);
}
int rr_chainbank_process_dev(rr_chainbank *h, double rate, const void *d_in, size_t in_stride, size_t n_in, void *d_out,
                             size_t out_stride, size_t cap, size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    return h->process_dev(rate, d_in, in_stride, n_in, d_out, out_stride, cap, n_out);
    RR_GUARD_END
}
int rr_chainbank_last_path(const rr_chainbank *h, int *lockstep) {
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    if (!lockstep) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *lockstep = h->last_path;
    return RR_OK;
}
int rr_chainbank_destroy(rr_chainbank *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_CHAINBANK);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

int rr_chain_set_metering(rr_chain *h, double double_percentile, double *d_bandwidth, double *d_energy, size_t cap_frames,
                          int store_spectra) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    ++h->mutations;
    return set_sink(h->sink, double_percentile, h->p.output_rate, d_bandwidth, d_energy, cap_frames, store_spectra);
}
int rr_chain_last_path(const rr_chain *h, int *fused) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    *fused = h->last_fused;
    return RR_OK;
}
int rr_chain_timing_enable(rr_chain *h, int on) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    ++h->mutations;
    h->timers.on = on != 0;
    h->timers.only_stage = on == 2 ? ST_FUSED_FIR : -1;
    return RR_OK;
}
int rr_chain_timing_every(rr_chain *h, unsigned every) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    h->timers.every = every ? every : 1;
    h->timers.seen = 0;
    return RR_OK;
}
int rr_chain_timing_reset(rr_chain *h) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    RR_TRY(h->select());
    h->timers.seen = 0;
    h->timers.reset();
    return RR_OK;
}
int rr_chain_timing_read(rr_chain *h, int stage, double *total_ms, uint64_t *launches) {
    RR_CHECK_HANDLE(h, K_CHAIN);
    if (stage < 0 || stage >= ST_COUNT || !total_ms || !launches) RR_FAIL(RR_ERR_BAD_ARG, "bad stage");
    RR_TRY(h->select());
    RR_TRY(h->timers.drain());
    *total_ms = h->timers.total_ms[stage];
    *launches = h->timers.launches[stage];
    return RR_OK;
}
const char *rr_chain_timing_stage_name(int stage) {
    static const char *names[ST_COUNT] = {"freqshift", "filter_fir", "decim_fir", "fourier", "fused_mix_fir_decim", "fused_window_fft"};
    return (stage >= 0 && stage < ST_COUNT) ? names[stage] : nullptr;
}
int rr_chain_destroy(rr_chain *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_CHAIN);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

int rr_meter_create(const rr_meter_params *p, int device, rr_meter **out) {
    RR_GUARD_BEGIN
    if (!out || !p) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    if (p->chunk_len == 0) RR_FAIL(RR_ERR_CONTRACT, "chunk length must be positive");
    if (p->overlap == 0) RR_FAIL(RR_ERR_CONTRACT, "chunk count must be positive");  // chunks.rs:195
    if (!(p->output_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "output sample rate must be positive");  // resampling.rs:51-56
    if (!(p->bandwidth >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be positive");
    if (!(p->bandwidth < p->output_rate)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be smaller than output sample rate");
    rr_stft *st = nullptr;
    RR_TRY(rr_stft_create(p->dtype, p->chunk_len, p->overlap, &p->fft_window, p->center_dc, device, &st));
    auto *h = new rr_meter;
    h->st = st;
    int s = h->init_base(K_METER, p->dtype, device);
    if (s == RR_OK) {
        h->fs = new rr_freqshifter;
        s = h->fs->init_base(K_FREQSHIFTER, p->dtype, device);
    }
    if (s == RR_OK) {
        h->ds = new rr_downsampler;
        s = h->ds->init_base(K_DOWNSAMPLER, p->dtype, device);
    }
    if (s == RR_OK) {
        h->fl = new rr_filter;
        s = h->fl->init_base(K_FILTER, p->dtype, device);
    }
    if (s != RR_OK) {
        delete h;
        return s;
    }
    h->fs->precision = p->precision;
    h->fs->shift = p->shift;
    h->ds->output_rate = p->output_rate;
    h->ds->bandwidth = p->bandwidth;
    h->ds->quality = p->quality;
    h->ds->mixer_rides = true;  // (the FreqShifter in front rides along with k_decim_poly: rr_meter::process)
    h->chunk_len = p->chunk_len;
    h->overlap = p->overlap;
    h->output_rate = p->output_rate;
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_meter_set_shift(rr_meter *h, double shift) {
    RR_CHECK_HANDLE(h, K_METER);
    h->fs->shift = shift;
    h->fs->shift_changed = true;
    return RR_OK;
}
int rr_meter_filter_design(rr_meter *h, const rr_c64 *resp, const double *window_rel) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_METER);
    h->set_streams();
    return h->fl->design(h->output_rate, h->chunk_len, resp, window_rel);
    RR_GUARD_END
}
int rr_meter_event(rr_meter *h, int is_interrupt) {
    RR_CHECK_HANDLE(h, K_METER);
    if (is_interrupt) h->fl->hist_valid = false;  // filters.rs:262-265
    h->st->have_chunks = 0;                       // the Overlapper drops its history at any event (chunks.rs:225-233)
    h->st->carry_len = 0;
    return RR_OK;
}
int rr_meter_peek(rr_meter *h, double sample_rate, size_t n_in, size_t *n_frames) {
    RR_CHECK_HANDLE(h, K_METER);
    if (!n_frames) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return h->peek(sample_rate, n_in, n_frames);
}
int rr_meter_process_dev(rr_meter *h, double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_METER);
    return h->process_dev(sample_rate, d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_meter_process(rr_meter *h, double sample_rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_METER);
    if (n_out) *n_out = 0;
    size_t frames = 0;
    RR_TRY(h->peek(sample_rate, n_in, &frames));
    const size_t produce = frames * h->chunk_len * h->overlap;
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Meter: out_cap %zu < %zu", cap, produce);
    size_t got = 0;
    RR_TRY(host_io(h, in, n_in, out, produce, true, [&](void *di, void *dout, size_t *p) {
        int s = h->process_dev(sample_rate, di, n_in, dout, produce, p);
        got = *p;
        return s;
    }));
    if (n_out) *n_out = got;
    return RR_OK;
    RR_GUARD_END
}
int rr_meter_set_metering(rr_meter *h, double double_percentile, double *d_bandwidth, double *d_energy, size_t cap_frames,
                          int store_spectra) {
    RR_CHECK_HANDLE(h, K_METER);
    return set_sink(h->st->sink, double_percentile, h->output_rate, d_bandwidth, d_energy, cap_frames, store_spectra);
}
// The example's loop body as one call (examples/bandwidth_meter/main.rs:75-78): samples in, one metering::bandwidth per
// spectrum out - the spectra themselves never leave the chip (they are not even written to device memory).
int rr_meter_process_bandwidth(rr_meter *h, double sample_rate, const void *in, size_t n_in, double double_percentile,
                               double *bandwidth_out, size_t cap_frames, size_t *n_frames) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_METER);
    if (n_frames) *n_frames = 0;
    if (n_in && !in) RR_FAIL(RR_ERR_BAD_ARG, "null input");
    size_t frames = 0;
    RR_TRY(h->peek(sample_rate, n_in, &frames));
    if (frames > cap_frames) RR_FAIL(RR_ERR_CAPACITY, "Meter: room for %zu bandwidths, the call makes %zu spectra", cap_frames, frames);
    if (frames && !bandwidth_out) RR_FAIL(RR_ERR_BAD_ARG, "null output");
    RR_TRY(h->select());
    const size_t esz = elem_size(h->dtype);
    RR_TRY(h->stage_in.reserve((n_in ? n_in : 1) * esz));
    RR_TRY(h->bwbuf.reserve((frames ? frames : 1) * sizeof(double)));
    if (n_in) RR_HIP(hipMemcpyAsync(h->stage_in.p, in, n_in * esz, hipMemcpyHostToDevice, h->stream));
    const MeterSink saved = h->st->sink;
    MeterSink k;
    k.on = true;
    k.dp = double_percentile;
    k.rate = h->output_rate;
    k.bw = h->bwbuf.as<double>();
    k.cap = frames;
    k.store = 0;
    h->st->sink = k;
    size_t got = 0;
    const int rc = h->process_dev(sample_rate, h->stage_in.p, n_in, nullptr, 0, &got);
    h->st->sink = saved;
    RR_TRY(rc);
    if (frames) RR_HIP(hipMemcpyAsync(bandwidth_out, h->bwbuf.p, frames * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RR_HIP(hipStreamSynchronize(h->stream));
    if (n_frames) *n_frames = frames;
    return RR_OK;
    RR_GUARD_END
}
int rr_meter_last_path(const rr_meter *h, int *front_fused) {
    RR_CHECK_HANDLE(h, K_METER);
    if (!front_fused) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *front_fused = h->last_front_fused ? 1 : 0;
    return RR_OK;
}
int rr_meter_destroy(rr_meter *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_METER);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

}  // extern "C"
