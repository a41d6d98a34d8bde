// rr_api_fourier.hip — Fourier, Stft (Rechunker -> Overlapper -> Fourier) and Channelizer: host logic (which kernels a chunk
// length takes, tables, histories) and extern "C" entry points.
#include "rr_api_common.hpp"

// ---------------------------------------------------------------------------
// Overlapped Fourier analysis: Rechunker -> Overlapper -> Fourier
// ---------------------------------------------------------------------------
rr_stft::~rr_stft() { delete fo; }

int rr_stft::process_dev(const void *d_in_, size_t n_in_, void *d_out, size_t cap, size_t *n_out) {
    if (n_out) *n_out = 0;
    const size_t produce = peek(n_in_);
    const bool store = !sink.on || sink.store;
    if (store && produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Stft: out_cap %zu < %zu", cap, produce);
    if (sink.on && produce / (M * P) > sink.cap)
        RR_FAIL(RR_ERR_CAPACITY, "Stft: room for %zu bandwidths, the call makes %zu spectra", sink.cap, produce / (M * P));
    if (n_in_ == 0) return RR_OK;
    RR_TRY(select());
    const size_t N = M * P, H = (P - 1) * M, esz = elem_size(dtype);
    // Rechunker (chunks.rs:42-177): whole chunks of M out of [patchwork | input]; the rest waits
    const size_t total = carry_len + n_in_, n_in = total / M * M, left = total - n_in;
    RR_TRY(carry.reserve(M * esz));
    const char *d_in = static_cast<const char *>(d_in_);
    if (n_in == 0) {  // not even one chunk yet
        RR_HIP(hipMemcpyAsync(carry.as<char>() + carry_len * esz, d_in, n_in_ * esz, hipMemcpyDeviceToDevice, stream));
        carry_len = total;
        return RR_OK;
    }
    if (carry_len) {  // ragged input: the chunks are assembled once (aligned input takes the zero-copy path)
        RR_TRY(work.reserve(n_in * esz));
        RR_HIP(hipMemcpyAsync(work.p, carry.p, carry_len * esz, hipMemcpyDeviceToDevice, stream));
        RR_HIP(hipMemcpyAsync(work.as<char>() + carry_len * esz, d_in, (n_in - carry_len) * esz, hipMemcpyDeviceToDevice, stream));
        if (left) RR_HIP(hipMemcpyAsync(carry.p, d_in + (n_in_ - left) * esz, left * esz, hipMemcpyDeviceToDevice, stream));
        d_in = work.as<char>();
    } else if (left) {
        RR_HIP(hipMemcpyAsync(carry.p, d_in + n_in * esz, left * esz, hipMemcpyDeviceToDevice, stream));
    }
    carry_len = left;
    const size_t chunks = n_in / M;
    const size_t frames = produce / N;
    if (frames) {
        fo->stream = stream;
        RR_TRY(fo->prepare(N));
        // frame 0 ends with the chunk that completes the history (see rr_channelizer::process_dev)
        const size_t first_complete = (have_chunks >= P - 1) ? 0 : (P - 1 - have_chunks);
        const long base0 = ((long)first_complete - (long)(P - 1)) * (long)M;  // <= 0 only if the history holds it
        const size_t n_head = base0 < 0 ? (size_t)(-base0) : 0;
        const char *head = hist[cur].as<char>() + (H - n_head) * esz;
        const char *in0 = d_in + (base0 > 0 ? (size_t)base0 * esz : 0);
        if (sink.on)
            RR_TRY(fo->transform_metered_dev(head, n_head, in0, d_out, M, frames, sink.frame_meter()));
        else
            RR_TRY(fo->transform_dev(head, n_head, in0, d_out, M, frames));
    }
    if (H) {
        RR_TRY(launch_update_hist(dtype, stream, hist[cur].p, hist[cur ^ 1].p, H, d_in, n_in));
        cur ^= 1;
    }
    have_chunks = (have_chunks + chunks > P - 1) ? P - 1 : have_chunks + chunks;
    if (n_out) *n_out = produce;
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Fourier
// ---------------------------------------------------------------------------
// Which kernel family transforms a chunk of `len` points:
//   fast     the radix-8/16 register kernels of rr_fft_regs.hip (f32: 64 .. 8192) and the LDS radix-2 kernel (<= 8192 f32,
//            <= 4096 f64) for powers of two
//   big      powers of two beyond that, up to 2^24: four-step through HBM (launch_fft_big)
//   bluestein any other length >= 32 (either dtype): two power-of-two transforms of M >= 2 len - 1 points by a nested
//            rectangular-window Fourier (which is `fast` or `big` itself)
//   direct   other lengths below 32: the O(n^2) kernel
static bool is_pow2_sz(size_t n) { return n && (n & (n - 1)) == 0; }

// Which kernels transform a chunk of `len` points - ONE decision, used by prepare() and by rr_fourier_route() (host only).
struct FourierRoute {
    enum Kind { DIRECT, POW2, BIG_TILE, BIG_TRANSPOSE, BIG_GENERIC, MIXED, TILEM, BS_WAVE, BS_FUSED, BS_FUSED8K, BS_BIG, BS_LDS, BS_LAUNCHES } kind = DIRECT;
    size_t N1 = 0, N2 = 0;  // the four-step / two-pass split
    size_t M = 0;           // Bluestein's power-of-two length
};
static FourierRoute fourier_route(int dtype, size_t len, bool force_mixed, bool prefer_tile = false) {
    FourierRoute r;
    const bool pow2 = is_pow2_sz(len);
    const bool generic = [] { const char *e = std::getenv("RR_FOURIER_GENERIC"); return e && std::atoi(e) != 0; }();
    const int mixed_env = [] { const char *e = std::getenv("RR_FOURIER_MIXED"); return e ? std::atoi(e) : 1; }();  // 0 never, 2 wherever it applies
    if (pow2) {
        // (Bluestein's inner transform of 16 384 points stays on the two passes of k_fft_tile: four launches with the chirp products in
        //  their loads and stores against five around k_fft16384)
        if (len < 4 || (fourier_pow2_path(dtype, len) && !(prefer_tile && len > 8192))) {  // (a chunk of 1 sample is a power of two, too)
            r.kind = FourierRoute::POW2;
            return r;
        }
        fft_big_split(len, &r.N1, &r.N2);
        if (generic) {
            r.kind = FourierRoute::BIG_GENERIC;
            return r;
        }
        // 2^13 / 2^14 .. 2^18 points: two passes over HBM (k_fft_tile); RR_FOURIER_BIG=transpose keeps the five launches
        const bool force_tr = [] { const char *e = std::getenv("RR_FOURIER_BIG"); return e && std::string(e) == "transpose"; }();
        r.kind = (!force_tr && fft_tile_supported(dtype, r.N1, r.N2)) ? FourierRoute::BIG_TILE : FourierRoute::BIG_TRANSPOSE;
        return r;
    }
    // lengths 2^a 3^b 5^c (7^d 11^e 13^f): mixed-radix passes in one LDS image instead of Bluestein's two padded power-of-two
    // transforms, where measured faster (fft_mixed_preferred; RR_FOURIER_MIXED=0 keeps Bluestein, 2 takes it wherever it applies);
    // beyond one image the two passes of k_fft_tilem.  (Complex<f64> powers of two through the same in-place kernel were measured
    // SLOWER than k_fft_pow2's Stockham passes: 4096 points 0.355 against 0.241 ms per 2^24 samples, 256 points 0.237 against 0.133.)
    if (!generic && mixed_env != 0) {
        if (fft_mixed_supported(dtype, len)) {
            // 2049 .. 8192 points in f32: against k_bluestein_big<M>, one kernel around two workgroup transforms whose cost per chunk
            // does not depend on n (450 / n ms per 2^24 samples around M = 8192, 1240 / n around 16 384), the mixed passes cost by
            // their number: ~0.0275 ms each, a tenth more from seven passes on, + 0.1 with a radix 7 / 11 / 13 (2100 points 0.246 / 0.220, 2800 0.222 / 0.164, 7000 0.270 / 0.168).  One session
            // (scripts/mixed_vs_bluestein_probe.py), mixed / Bluestein: 2500 points 0.150 / 0.181, 3000 0.164 / 0.152, 4000 0.163 / 0.120,
            // 4004 0.224 / 0.118, 5000 0.173 / 0.218, 6000 0.164 / 0.182, 6144 (seven passes) 0.215 / 0.181, 7200 0.205 / 0.166, 8000 0.182 / 0.157
            bool prefer = fft_mixed_preferred(dtype, len);
            size_t Mb = 0;
            if (dtype == RR_F32 && len > 2048 && bluestein_big_supported(dtype, len, &Mb) && fourier_pow2_path(dtype, Mb) &&
                ![] { const char *e = std::getenv("RR_FOURIER_BS_BIG"); return e && std::atoi(e) == 0; }() &&
                !(Mb == 8192 && [] { const char *e = std::getenv("RR_FOURIER_BS8K"); return e && std::strcmp(e, "regs") == 0; }())) {
                const bool slow_radix = len % 7 == 0 || len % 11 == 0 || len % 13 == 0;
                unsigned char rad[16];
                const int passes = fft_mixed_radices(dtype, len, rad, 16);
                const double t_mixed = 0.0275 * passes * (passes >= 7 ? 1.1 : 1.0) + (slow_radix ? 0.1 : 0.0);
                const double t_bs = (Mb == 8192 ? 450.0 : 1240.0) / (double)len;
                prefer = passes > 0 && t_mixed < t_bs;
            }
            if (mixed_env == 2 || force_mixed || prefer) {
                r.kind = FourierRoute::MIXED;
                return r;
            }
        } else if (fft_tilem_split(dtype, len, &r.N1, &r.N2)) {
            r.kind = FourierRoute::TILEM;
            return r;
        }
    }
    if (len >= 32 && !(generic && len <= 16384)) {
        size_t M = 64;
        while (M < 2 * len - 1) M *= 2;
        r.kind = FourierRoute::BS_LAUNCHES;
        // 513 .. 2048 points in f32: the whole algorithm in one kernel around two 4096-point transforms in LDS; 32 .. 512 points
        // in f32: a wave per chunk around two 1024-point transforms (RR_FOURIER_GENERIC=1 keeps the five launches)
        if (!generic && bluestein4096_supported(dtype, len)) {
            r.kind = FourierRoute::BS_FUSED;
            M = 4096;
        } else if (!generic && bluestein1024_supported(dtype, len)) {
            r.kind = FourierRoute::BS_WAVE;
            M = 1024;
        } else if (!generic && bluestein8192_supported(dtype, len) &&
                   [] { const char *e = std::getenv("RR_FOURIER_BS8K"); return e && std::strcmp(e, "regs") == 0; }()) {
            // (on request: f32, 2049 .. 4096 points around two 8192-point register transforms at 256 lanes - 0.224 ms per 2^24 samples
            //  of 3001-point chunks against 0.161 by the workgroup transform below)
            r.kind = FourierRoute::BS_FUSED8K;
            M = 8192;
        } else if (size_t Mb = 0; !generic && bluestein_big_supported(dtype, len, &Mb) && fourier_pow2_path(dtype, Mb) &&
                                  ![] { const char *e = std::getenv("RR_FOURIER_BS_BIG"); return e && std::atoi(e) == 0; }()) {
            // f32, 2049 .. 8192 points: one kernel around two workgroup transforms of 8192 / 16 384 points (rr_fft_big.hpp)
            r.kind = FourierRoute::BS_BIG;
            M = Mb;
        } else if (!generic && bluestein_lds_supported(dtype, len, M) &&
                   ![] { const char *e = std::getenv("RR_FOURIER_BS_LDS"); return e && std::atoi(e) == 0; }()) {
            // f64 up to 2048 points: one kernel with the transforms as Stockham passes between two LDS images
            r.kind = FourierRoute::BS_LDS;
        }
        r.M = M;
        return r;
    }
    r.kind = FourierRoute::DIRECT;  // (also: RR_FOURIER_GENERIC=1 up to 16384 points)
    return r;
}

int rr_fourier::prepare(size_t len) {
    if (len == n) return RR_OK;
    RR_TRY(fourier_supported(dtype, len));
    std::vector<double> rel(len);
    if (window.kind == RR_WIN_SAMPLED) {
        if (sampled_n != len)
            RR_FAIL(RR_ERR_NEED_DESIGN, "Fourier: sampled window has %zu values, chunk has %zu", sampled_n, len);
        rel = sampled;
    } else {
        RR_TRY(window_sample(&window, len, rel.data()));
    }
    std::vector<double> vals(len);
    RR_TRY(fourier_design_window(len, rel.data(), vals.data()));
    const FourierRoute route = fourier_route(dtype, len, force_mixed, prefer_tile);
    using FR = FourierRoute;
    const bool use_big = route.kind == FR::BIG_TILE || route.kind == FR::BIG_TRANSPOSE || route.kind == FR::BIG_GENERIC;
    const bool generic = route.kind == FR::BIG_GENERIC;  // (only consulted on the `big` branches below)
    const bool use_mixed = route.kind == FR::MIXED, use_tilem = route.kind == FR::TILEM;
    const size_t tmN1 = route.N1, tmN2 = route.N2;
    const bool use_bs = route.kind == FR::BS_WAVE || route.kind == FR::BS_FUSED || route.kind == FR::BS_FUSED8K || route.kind == FR::BS_BIG ||
                        route.kind == FR::BS_LDS ||
                        route.kind == FR::BS_LAUNCHES;
    auto cast = [&](const std::vector<double> &src, std::vector<unsigned char> &dst) {
        if (dtype == RR_F32) cast_to<float>(src.data(), src.size(), dst);
        else cast_to<double>(src.data(), src.size(), dst);
    };
    std::vector<unsigned char> wb, tb;
    cast(vals, wb);
    big_t = false;
    if (use_big && !generic) {
        // four-step as row transforms between tiled transposes: nested transforms of N1 and N2 points and the
        // twiddles W_len^e = tA[e >> h] tB[e & (2^h - 1)]
        size_t N1, N2;
        fft_big_split(len, &N1, &N2);
        int lg = 0;
        while (((size_t)1 << lg) < len) ++lg;
        const int h = (lg + 1) / 2;
        const size_t nB = (size_t)1 << h, nA = len >> h;
        std::vector<double> tw(2 * (nA + nB));
        for (size_t i = 0; i < nB; ++i) {
            const double ang = -2.0 * M_PI * (double)i / (double)len;
            tw[2 * i] = std::cos(ang);
            tw[2 * i + 1] = std::sin(ang);
        }
        for (size_t i = 0; i < nA; ++i) {
            const double ang = -2.0 * M_PI * (double)(i << h) / (double)len;
            tw[2 * (nB + i)] = std::cos(ang);
            tw[2 * (nB + i) + 1] = std::sin(ang);
        }
        // two passes over HBM (k_fft_tile) with the sub-transforms' own tables behind tB | tA, or the five launches
        // (transposes around the fast row kernels): fourier_route
        big_tile = route.kind == FR::BIG_TILE;
        if (big_tile) {
            big_tw1_off = nA + nB;
            big_tw2_off = big_tw1_off + N1;
            for (size_t Nx : {N1, N2})
                for (size_t i = 0; i < Nx; ++i) {
                    const double ang = -2.0 * M_PI * (double)i / (double)Nx;
                    tw.push_back(std::cos(ang));
                    tw.push_back(std::sin(ang));
                }
        }
        cast(tw, tb);
        RR_TRY(upload(d_window, wb.data(), wb.size(), stream));
        RR_TRY(upload(d_tw, tb.data(), tb.size(), stream));
        if (!big_tile) {
            for (rr_fourier **sub : {&bigA, &bigB}) {
                if (!*sub) {
                    *sub = new rr_fourier;
                    RR_TRY((*sub)->init_base(K_FOURIER, dtype, device));
                }
                (*sub)->stream = stream;
            }
            RR_TRY(bigA->prepare(N1));  // rectangular windows: all ones
            RR_TRY(bigB->prepare(N2));
        }
        big_t = true;
        big_h = h;
    } else if (use_big) {
        // half tables e^{-j 2 pi k / N1}, e^{-j 2 pi k / N2} of the four-step split, one behind the other
        size_t N1, N2;
        fft_big_split(len, &N1, &N2);
        std::vector<double> tw(N1 + N2);  // (N1 / 2 + N2 / 2) complex
        for (size_t k = 0; k < N1 / 2; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)N1;
            tw[2 * k] = std::cos(ang);
            tw[2 * k + 1] = std::sin(ang);
        }
        for (size_t k = 0; k < N2 / 2; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)N2;
            tw[N1 + 2 * k] = std::cos(ang);
            tw[N1 + 2 * k + 1] = std::sin(ang);
        }
        cast(tw, tb);
        RR_TRY(upload(d_window, wb.data(), wb.size(), stream));
        RR_TRY(upload(d_tw, tb.data(), tb.size(), stream));
    } else if (use_tilem) {
        // e^{-j 2 pi k / N1} | e^{-j 2 pi k / N2} | T1[i] = W_N^(C i), i < N1 ceil(N2 / C) | T2[i] = W_N^i, i < N1 C
        const size_t Cc = dtype == RR_F32 ? 16 : 8, nbx = (tmN2 + Cc - 1) / Cc;
        std::vector<double> tw;
        tw.reserve(2 * (tmN1 + tmN2 + tmN1 * nbx + tmN1 * Cc));
        auto push = [&](size_t num, size_t den) {  // e^{-j 2 pi num / den}, the phase reduced exactly
            const double ang = -2.0 * M_PI * (double)(num % den) / (double)den;
            tw.push_back(std::cos(ang));
            tw.push_back(std::sin(ang));
        };
        for (size_t i = 0; i < tmN1; ++i) push(i, tmN1);
        for (size_t i = 0; i < tmN2; ++i) push(i, tmN2);
        for (size_t i = 0; i < tmN1 * nbx; ++i) push(Cc * i, len);
        for (size_t i = 0; i < tmN1 * Cc; ++i) push(i, len);
        cast(tw, tb);
        RR_TRY(upload(d_window, wb.data(), wb.size(), stream));
        RR_TRY(upload(d_tw, tb.data(), tb.size(), stream));
        tm_N1 = tmN1;
        tm_N2 = tmN2;
        tm_T1 = tmN1 + tmN2;
        tm_T2 = tm_T1 + tmN1 * nbx;
    } else if (use_bs) {
        RR_TRY(upload(d_window, wb.data(), wb.size(), stream));  // (kept for symmetry; Bluestein folds the window into c)
    } else {
        const size_t ntw = len;  // the radix-2 kernel uses the first half, radix-16 and direct all of it
        std::vector<double> tw(2 * ntw);
        for (size_t k = 0; k < ntw; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)len;
            tw[2 * k] = std::cos(ang);
            tw[2 * k + 1] = std::sin(ang);
        }
        cast(tw, tb);
        if (dtype == RR_F32 && (len == 4096 || len == 2048)) {
            // k_fft4096 / k_fft2048 read the 16 window values of a lane (w[j + T k], k < 16, T = len / 16 lanes) as four
            // 16-byte pieces from a second copy behind the table: wp[16 j + k] = w[j + T k]
            const size_t T = len / 16;
            std::vector<float> both(2 * len);
            std::memcpy(both.data(), wb.data(), len * sizeof(float));
            for (size_t j = 0; j < T; ++j)
                for (size_t k = 0; k < 16; ++k) both[len + 16 * j + k] = both[j + T * k];
            RR_TRY(upload(d_window, both.data(), both.size() * sizeof(float), stream));
        } else {
            RR_TRY(upload(d_window, wb.data(), wb.size(), stream));
        }
        if (dtype == RR_F32 && len == 1024) {  // k_fft1024 finds its lane seeds behind the table
            std::vector<float> twb(2 * 1024);
            std::memcpy(twb.data(), tb.data(), twb.size() * sizeof(float));
            append_wave1024_seeds(twb);
            RR_TRY(upload(d_tw, twb.data(), twb.size() * sizeof(float), stream));
        } else {
            RR_TRY(upload(d_tw, tb.data(), tb.size(), stream));
        }
    }
    window_f64.swap(vals);
    n = len;
    mixed = use_mixed;
    tilem = use_tilem;
    bs_M = 0;
    bs_fused = bs_wave = bs_lds = bs_fused8k = bs_big = false;
    big = use_big;
    if (use_bs) {
        const size_t M = route.M;
        bs_fused = route.kind == FR::BS_FUSED;  // k_bluestein4096
        bs_wave = route.kind == FR::BS_WAVE;    // k_bluestein1024
        bs_lds = route.kind == FR::BS_LDS;      // k_bluestein_lds
        bs_fused8k = route.kind == FR::BS_FUSED8K;  // k_bluestein8192
        bs_big = route.kind == FR::BS_BIG;          // k_bluestein_big<M>
        // chirp w_m = e^{+j pi m^2 / n}, the phase reduced exactly (m^2 mod 2n) before it is evaluated
        std::vector<cd> w(len);
        for (size_t m = 0; m < len; ++m) {
            const uint64_t r = (uint64_t)(((unsigned __int128)m * m) % (2 * len));
            const double ang = M_PI * (double)r / (double)len;
            w[m] = cd(std::cos(ang), std::sin(ang));
        }
        std::vector<cd> bb(M, cd(0, 0));
        bb[0] = w[0];
        for (size_t m = 1; m < len; ++m) bb[m] = bb[M - m] = w[m];
        fft_f64(bb, false);
        std::vector<double> cf(2 * (len + 1), 0.0), wf(2 * len), Bf(2 * M);  // (c: one zero entry behind an odd length)
        for (size_t m = 0; m < len; ++m) {
            const cd c = std::conj(w[m]) * window_f64[m];
            cf[2 * m] = c.real();
            cf[2 * m + 1] = c.imag();
            wf[2 * m] = w[m].real();
            wf[2 * m + 1] = w[m].imag();
        }
        for (size_t m = 0; m < M; ++m) {
            // k_bluestein1024 reads B pair-interleaved: [kp][l] = {B[l + 128 kp], B[l + 128 kp + 64]} (as k_filter_wave's H)
            size_t dst = m;
            if (bs_wave) {
                const size_t l = m % 64, j = (m / 64) % 2, kp = m / 128;
                dst = (kp * 64 + l) * 2 + j;
            }
            if (bs_big) {  // [kp][j][h] = B[j + T (2 kp + h)], T = M / 16 lanes
                const size_t T = M / 16, j = m % T, k = m / T;
                dst = ((k / 2) * T + j) * 2 + k % 2;
            }
            Bf[2 * dst] = bb[m].real() / (double)M;
            Bf[2 * dst + 1] = bb[m].imag() / (double)M;
        }
        std::vector<unsigned char> cb, wwb, Bb;
        cast(cf, cb);
        cast(wf, wwb);
        cast(Bf, Bb);
        RR_TRY(upload(d_bs_c, cb.data(), cb.size(), stream));
        RR_TRY(upload(d_bs_w, wwb.data(), wwb.size(), stream));
        RR_TRY(upload(d_bs_B, Bb.data(), Bb.size(), stream));
        if (!bs_fft) {
            bs_fft = new rr_fourier;
            RR_TRY(bs_fft->init_base(K_FOURIER, dtype, device));
        }
        bs_fft->stream = stream;
        bs_fft->prefer_tile = !bs_big;  // (k_bluestein_big reads the whole table W_M^i)
        RR_TRY(bs_fft->prepare(M));  // rectangular window: all ones
        bs_M = M;
    }
    return RR_OK;
}

rr_fourier::~rr_fourier() {
    delete bs_fft;
    delete bigA;
    delete bigB;
}

int rr_fourier::transform_dev(const void *head, size_t n_head, const void *in, void *out, size_t hop, size_t count) {
    const size_t esz = elem_size(dtype);
    if (big) {
        if (hop != n || n_head) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: overlapping chunks of more than 8192 points are not supported");
        // passes of at most 2^24 workspace elements
        size_t per_pass = ((size_t)1 << 24) / n;
        if (per_pass < 1) per_pass = 1;
        if (per_pass > 65535) per_pass = 65535;
        if (per_pass > count) per_pass = count;
        RR_TRY(big_ws.reserve(per_pass * n * esz));
        size_t N1, N2;
        fft_big_split(n, &N1, &N2);
        if (big_t && big_tile) {
            const char *tB = d_tw.as<char>(), *tA = tB + ((size_t)1 << big_h) * esz;
            const char *tw1 = tB + big_tw1_off * esz, *tw2 = tB + big_tw2_off * esz;
            for (size_t f0 = 0; f0 < count; f0 += per_pass) {
                const size_t F = count - f0 < per_pass ? count - f0 : per_pass;
                const char *src = static_cast<const char *>(in) + f0 * n * esz;
                char *dst = static_cast<char *>(out) + f0 * n * esz;
                RR_TRY(launch_fft_tile(dtype, stream, 0, src, big_ws.p, N1, N2, F, d_window.p, tw1, tB, tA, big_h, 0));
                RR_TRY(launch_fft_tile(dtype, stream, 1, big_ws.p, dst, N1, N2, F, nullptr, tw2, nullptr, nullptr, 0,
                                       center_dc ? N2 / 2 : 0));
            }
            return RR_OK;
        }
        if (big_t) {
            RR_TRY(big_ws2.reserve(per_pass * n * esz));
            bigA->stream = bigB->stream = stream;
            const char *tB = d_tw.as<char>(), *tA = tB + ((size_t)1 << big_h) * esz;
            for (size_t f0 = 0; f0 < count; f0 += per_pass) {
                const size_t F = count - f0 < per_pass ? count - f0 : per_pass;
                const char *src = static_cast<const char *>(in) + f0 * n * esz;
                char *dst = static_cast<char *>(out) + f0 * n * esz;
                RR_TRY(launch_transpose_mul(dtype, stream, src, big_ws.p, N1, N2, F, 1, d_window.p, nullptr, nullptr, 0, 0));
                RR_TRY(bigA->transform_dev(nullptr, 0, big_ws.p, big_ws2.p, N1, F * N2));
                RR_TRY(launch_transpose_mul(dtype, stream, big_ws2.p, big_ws.p, N2, N1, F, 2, nullptr, tB, tA, big_h, 0));
                RR_TRY(bigB->transform_dev(nullptr, 0, big_ws.p, big_ws2.p, N2, F * N1));
                RR_TRY(launch_transpose_mul(dtype, stream, big_ws2.p, dst, N1, N2, F, 0, nullptr, nullptr, nullptr, 0,
                                            center_dc ? N2 / 2 : 0));
            }
            return RR_OK;
        }
        const char *tw2 = d_tw.as<char>() + (N1 / 2) * esz;
        for (size_t f0 = 0; f0 < count; f0 += per_pass) {
            const size_t F = count - f0 < per_pass ? count - f0 : per_pass;
            RR_TRY(launch_fft_big(dtype, stream, static_cast<const char *>(in) + f0 * n * esz, static_cast<char *>(out) + f0 * n * esz,
                                  big_ws.p, n, F, d_window.p, d_tw.p, tw2, center_dc));
        }
        return RR_OK;
    }
    if (mixed) return launch_fft_mixed(dtype, stream, head, n_head, in, hop, n, d_window.p, d_tw.p, out, center_dc, count);
    if (tilem) {
        // passes of at most 2^24 workspace elements; frame f0's first sample sits f0 * hop behind the start of [head | in]
        size_t per_pass = ((size_t)1 << 24) / n;
        if (per_pass < 1) per_pass = 1;
        if (per_pass > 65535) per_pass = 65535;
        if (per_pass > count) per_pass = count;
        RR_TRY(big_ws.reserve(per_pass * n * esz));
        const char *tw1 = d_tw.as<char>(), *tw2 = tw1 + tm_N1 * esz, *T1 = tw1 + tm_T1 * esz, *T2 = tw1 + tm_T2 * esz;
        for (size_t f0 = 0; f0 < count; f0 += per_pass) {
            const size_t F = count - f0 < per_pass ? count - f0 : per_pass;
            const size_t skip = f0 * hop;
            const char *hd = static_cast<const char *>(head), *src = static_cast<const char *>(in);
            size_t nh = n_head;
            if (skip >= n_head) {
                src += (skip - n_head) * esz;
                nh = 0;
            } else {
                hd += skip * esz;
                nh = n_head - skip;
            }
            RR_TRY(launch_fft_tilem(dtype, stream, 0, hd, nh, src, hop, big_ws.p, tm_N1, tm_N2, F, d_window.p, tw1, T1, T2, 0));
            RR_TRY(launch_fft_tilem(dtype, stream, 1, nullptr, 0, big_ws.p, 0, static_cast<char *>(out) + f0 * n * esz, tm_N1,
                                    tm_N2, F, nullptr, tw2, nullptr, nullptr, center_dc ? n / 2 : 0));
        }
        return RR_OK;
    }
    if (!bs_M) return launch_fourier_overlapped(dtype, stream, head, n_head, in, out, n, hop, count, d_window.p, d_tw.p, center_dc);
    if (bs_wave)
        return launch_bluestein1024(stream, head, n_head, in, hop, n, d_bs_c.p, d_bs_B.p, d_bs_w.p, bs_fft->d_tw.p, out, center_dc, count);
    if (bs_big)
        return launch_bluestein_big(stream, bs_M, head, n_head, in, hop, n, d_bs_c.p, d_bs_B.p, d_bs_w.p, bs_fft->d_tw.p, out, center_dc,
                                    count);
    if (bs_fused8k)
        return launch_bluestein8192(stream, head, n_head, in, hop, n, d_bs_c.p, d_bs_B.p, d_bs_w.p, bs_fft->d_tw.p, out, center_dc, count);
    if (bs_lds)
        return launch_bluestein_lds(dtype, stream, head, n_head, in, hop, n, bs_M, d_bs_c.p, d_bs_B.p, d_bs_w.p, bs_fft->d_tw.p, out,
                                    center_dc, count);
    if (bs_fused)
        return launch_bluestein4096(stream, head, n_head, in, hop, n, d_bs_c.p, d_bs_B.p, d_bs_w.p, bs_fft->d_tw.p, out, center_dc, count);
    const size_t M = bs_M;
    // M = 2^13 / 2^14 .. 2^18 (the two-pass form of the nested transform): FOUR launches - the element-wise stages ride on the
    // loads and stores of k_fft_tile's passes (x c at the first load, conj(. B) at the second store, conj(. chirp) and the cut to
    // n bins at the last store): 8 passes over the padded length instead of 14.  RR_FOURIER_BS_FUSED=0 keeps the seven launches.
    if (bs_fft->big && bs_fft->big_tile && ![] { const char *e = std::getenv("RR_FOURIER_BS_FUSED"); return e && std::atoi(e) == 0; }()) {
        size_t N1, N2;
        fft_big_split(M, &N1, &N2);
        size_t per_pass = ((size_t)1 << 23) / M;
        if (per_pass < 1) per_pass = 1;
        if (per_pass > 65535) per_pass = 65535;
        if (per_pass > count) per_pass = count;
        RR_TRY(bs_ws[0].reserve(per_pass * M * esz));
        RR_TRY(bs_ws[1].reserve(per_pass * M * esz));
        const char *tB = bs_fft->d_tw.as<char>(), *tA = tB + ((size_t)1 << bs_fft->big_h) * esz;
        const char *tw1 = tB + bs_fft->big_tw1_off * esz, *tw2 = tB + bs_fft->big_tw2_off * esz;
        const int hh = bs_fft->big_h;
        for (size_t f0 = 0; f0 < count; f0 += per_pass) {
            const size_t F = count - f0 < per_pass ? count - f0 : per_pass;
            const size_t skip = f0 * hop;
            const char *hd = static_cast<const char *>(head), *src = static_cast<const char *>(in);
            size_t nh = n_head;
            if (skip >= n_head) {
                src += (skip - n_head) * esz;
                nh = 0;
            } else {
                hd += skip * esz;
                nh = n_head - skip;
            }
            RR_TRY(launch_fft_tile_bs(dtype, stream, 0, hd, nh, src, hop, bs_ws[0].p, N1, N2, F, n, d_bs_c.p, tw1, tB, tA, hh, 0));
            RR_TRY(launch_fft_tile_bs(dtype, stream, 1, nullptr, 0, bs_ws[0].p, 0, bs_ws[1].p, N1, N2, F, n, d_bs_B.p, tw2, nullptr,
                                      nullptr, 0, 0));
            RR_TRY(launch_fft_tile_bs(dtype, stream, 2, nullptr, 0, bs_ws[1].p, 0, bs_ws[0].p, N1, N2, F, n, nullptr, tw1, tB, tA, hh, 0));
            RR_TRY(launch_fft_tile_bs(dtype, stream, 3, nullptr, 0, bs_ws[0].p, 0, static_cast<char *>(out) + f0 * n * esz, N1, N2, F, n,
                                      d_bs_w.p, tw2, nullptr, nullptr, 0, center_dc ? n / 2 : 0));
        }
        return RR_OK;
    }
    // passes of at most 2^22 workspace elements per buffer (32 MiB each in f32)
    size_t per_pass = ((size_t)1 << 22) / M;
    if (per_pass < 1) per_pass = 1;
    if (per_pass > 65535) per_pass = 65535;
    if (per_pass > count) per_pass = count;
    RR_TRY(bs_ws[0].reserve(per_pass * M * esz));
    RR_TRY(bs_ws[1].reserve(per_pass * M * esz));
    bs_fft->stream = stream;
    for (size_t f0 = 0; f0 < count; f0 += per_pass) {
        const size_t F = count - f0 < per_pass ? count - f0 : per_pass;
        // frame f0's first sample sits f0 * hop behind the start of [head | in]
        const size_t skip = f0 * hop;
        const char *hd = static_cast<const char *>(head);
        const char *src = static_cast<const char *>(in);
        size_t nh = n_head;
        if (skip >= n_head) {
            src += (skip - n_head) * esz;
            nh = 0;
        } else {
            hd += skip * esz;
            nh = n_head - skip;
        }
        RR_TRY(launch_bs_pre(dtype, stream, hd, nh, src, hop, n, M, d_bs_c.p, bs_ws[0].p, F));
        RR_TRY(bs_fft->transform_dev(nullptr, 0, bs_ws[0].p, bs_ws[1].p, M, F));
        RR_TRY(launch_bs_mul(dtype, stream, bs_ws[1].p, d_bs_B.p, M, F));
        RR_TRY(bs_fft->transform_dev(nullptr, 0, bs_ws[1].p, bs_ws[0].p, M, F));
        RR_TRY(launch_bs_post(dtype, stream, bs_ws[0].p, d_bs_w.p, n, M, static_cast<char *>(out) + f0 * n * esz, center_dc, F));
    }
    return RR_OK;
}

int rr_fourier::transform_metered_dev(const void *head, size_t n_head, const void *in, void *out, size_t hop, size_t count,
                                      const rr::FrameMeter &fm) {
    if (count == 0) return RR_OK;
    if (!fm.bw) RR_FAIL(RR_ERR_BAD_ARG, "metering: no place for the bandwidths");
    if (fm.store && !out) RR_FAIL(RR_ERR_BAD_ARG, "null output");
    const char *se = std::getenv("RR_METER_SERIAL");  // (read per call: tests switch it within one process)
    const bool serial = se && std::atoi(se) != 0;
    static const bool generic = [] { const char *e = std::getenv("RR_FOURIER_GENERIC"); return e && std::atoi(e) != 0; }();
    if (!serial && !generic && dtype == RR_F32 && n == 4096 && !big && !mixed && !tilem && !bs_M) {
        // the epilogue rides on the transform's kernel: the bins never come back from memory
        if (stft4096_supported(hop) && count >= 64)
            return launch_stft4096(stream, head, n_head, in, out, count, d_window.p, d_tw.p, center_dc, hop, &fm);
        return launch_fft4096(stream, head, n_head, in, out, count, d_window.p, d_tw.p, center_dc, hop, nullptr, nullptr, &fm);
    }
    void *o = out;
    if (!fm.store || !out) {
        RR_TRY(meter_ws.reserve(count * n * elem_size(dtype)));
        o = meter_ws.p;
    }
    RR_TRY(transform_dev(head, n_head, in, o, hop, count));
    if (serial) {  // the reference's own summation order (bit-equal to the oracle): the checker path
        RR_TRY(launch_meter(dtype, stream, 1, fm.double_percentile, fm.sample_rate, o, n, count, fm.bw));
        if (fm.energy) RR_TRY(launch_meter(dtype, stream, 2, 0.0, 0.0, o, n, count, fm.energy));
        return RR_OK;
    }
    return launch_bandwidth_par(dtype, stream, fm.double_percentile, fm.sample_rate, o, n, count, fm.bw, fm.energy);
}

int rr_fourier::process_dev(size_t chunk_len, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    if (n_out) *n_out = 0;
    if (chunk_len == 0) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: chunk_len == 0");
    if (n_in % chunk_len) RR_FAIL(RR_ERR_BAD_ARG, "Fourier: %zu samples is not a whole number of %zu-sample chunks", n_in, chunk_len);
    if (n_in > cap) RR_FAIL(RR_ERR_CAPACITY, "Fourier: out_cap %zu < %zu", cap, n_in);
    if (n_in == 0) return RR_OK;
    RR_TRY(select());
    RR_TRY(prepare(chunk_len));
    RR_TRY(transform_dev(nullptr, 0, d_in, d_out, chunk_len, n_in / chunk_len));
    if (n_out) *n_out = n_in;
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Channelizer
// ---------------------------------------------------------------------------
rr_channelizer::~rr_channelizer() { delete fo; }

int rr_channelizer::process_dev(const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    if (n_out) *n_out = 0;
    if (n_in % hop) RR_FAIL(RR_ERR_BAD_ARG, "Channelizer: %zu samples is not a whole number of %zu-sample chunks", n_in, hop);
    const size_t produce = peek(n_in);
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Channelizer: out_cap %zu < %zu", cap, produce);
    if (n_in == 0) return RR_OK;
    RR_TRY(select());
    const size_t chunks = n_in / hop, K = span_chunks(), H = P * M - hop;
    const size_t frames = produce / M;
    if (frames) {
        // frame 0 ends with the chunk that completes the history: it starts (K - 1) chunks before that chunk
        const size_t first_complete = (have_chunks >= K - 1) ? 0 : (K - 1 - have_chunks);  // index of the chunk that ends frame 0
        const long base0 = ((long)first_complete - (long)(K - 1)) * (long)hop;
        if (!fo) {
            RR_TRY(launch_channelizer(dtype, stream, hist[cur].p, H, d_in, base0, M, P, frames, d_window.p, d_tw.p, d_out, hop));
        } else {
            // general form: fold every frame into the workspace, then the M-point transforms (any M)
            const size_t esz = elem_size(dtype);
            size_t per_pass = ((size_t)1 << 24) / M;
            if (per_pass < 1) per_pass = 1;
            if (per_pass > 65535) per_pass = 65535;
            if (per_pass > frames) per_pass = frames;
            RR_TRY(fold_ws.reserve(per_pass * M * esz));
            fo->stream = stream;
            fo->force_mixed = true;
            RR_TRY(fo->prepare(M));
            if (fo->mixed) {
                // bin counts 2^a 3^b 5^c: fold and transform in one kernel (k_fft_mixed with the fold at its load), no workspace
                RR_TRY(launch_fft_mixed_fold(dtype, stream, hist[cur].p, H, d_in, base0, hop, M, P, d_window.p, fo->d_tw.p, d_out,
                                             false, frames));
                per_pass = 0;
            }
            for (size_t f0 = 0; per_pass && f0 < frames; f0 += per_pass) {
                const size_t F = frames - f0 < per_pass ? frames - f0 : per_pass;
                RR_TRY(launch_chan_fold(dtype, stream, hist[cur].p, H, d_in, base0 + (long)(f0 * hop), hop, M, P, F, d_window.p,
                                        fold_ws.p));
                RR_TRY(fo->transform_dev(nullptr, 0, fold_ws.p, static_cast<char *>(d_out) + f0 * M * esz, M, F));
            }
        }
    }
    if (H) {
        RR_TRY(launch_update_hist(dtype, stream, hist[cur].p, hist[cur ^ 1].p, H, d_in, n_in));
        cur ^= 1;
    }
    have_chunks = (have_chunks + chunks > K - 1) ? K - 1 : have_chunks + chunks;
    if (n_out) *n_out = produce;
    return RR_OK;
}

extern "C" {

// ---- Stft (Rechunker -> Overlapper -> Fourier) -----------------------------------------------
int rr_stft_create(int dtype, size_t chunk_len, size_t chunk_count, const rr_window *window, int center_dc, int device,
                   rr_stft **out) {
    RR_GUARD_BEGIN
    if (!out || !window) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    if (chunk_len == 0) RR_FAIL(RR_ERR_CONTRACT, "chunk length must be positive");    // chunks.rs:56
    if (chunk_count == 0) RR_FAIL(RR_ERR_CONTRACT, "chunk count must be positive");   // chunks.rs:195
    const size_t N = chunk_len * chunk_count;
    // overlapped frames: the power-of-two kernels of one LDS tile, or Bluestein over power-of-two transforms (>= 32 points)
    if (!fourier_pow2_path(dtype, N) && !(N >= 32 && (N & (N - 1)) != 0 && N <= ((size_t)1 << 23)))
        RR_FAIL(RR_ERR_BAD_ARG, "Stft: chunk_len * chunk_count = %zu: powers of two up to %u, or any other length of 32 .. 2^23", N,
                dtype == RR_F32 ? 16384u : 4096u);
    if (window->kind != RR_WIN_RECTANGULAR && window->kind != RR_WIN_KAISER)
        RR_FAIL(RR_ERR_BAD_ARG, "Stft: window must be a built-in window");
    auto *h = new rr_stft;
    int st = h->init_base(K_STFT, dtype, device);
    if (st == RR_OK) {
        h->fo = new rr_fourier;
        st = h->fo->init_base(K_FOURIER, dtype, device);
    }
    if (st == RR_OK) {
        h->fo->window = *window;
        h->fo->center_dc = center_dc != 0;
        h->M = chunk_len;
        h->P = chunk_count;
        const size_t hb = (chunk_count - 1) * chunk_len * elem_size(dtype);
        st = h->hist[0].reserve(hb ? hb : 16);
        if (st == RR_OK) st = h->hist[1].reserve(hb ? hb : 16);
    }
    if (st != RR_OK) {
        delete h;
        return st;
    }
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_stft_reset(rr_stft *h) {
    RR_CHECK_HANDLE(h, K_STFT);
    h->have_chunks = 0;  // chunks.rs:225-233
    h->carry_len = 0;    // chunks.rs:80-88
    return RR_OK;
}
int rr_stft_pending(const rr_stft *h, size_t *n) {
    RR_CHECK_HANDLE(h, K_STFT);
    if (!n) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *n = h->carry_len;
    return RR_OK;
}
int rr_stft_peek(const rr_stft *h, size_t n_in, size_t *n_out) {
    RR_CHECK_HANDLE(h, K_STFT);
    if (!n_out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *n_out = h->peek(n_in);
    return RR_OK;
}
int rr_stft_process_dev(rr_stft *h, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_STFT);
    return h->process_dev(d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_stft_process(rr_stft *h, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_STFT);
    if (n_out) *n_out = 0;
    const size_t produce = h->peek(n_in);
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Stft: out_cap %zu < %zu", cap, produce);
    size_t got = 0;
    RR_TRY(host_io(h, in, n_in, out, produce, true, [&](void *di, void *dout, size_t *p) {
        int s = h->process_dev(di, n_in, dout, produce, p);
        got = *p;
        return s;
    }));
    if (n_out) *n_out = got;
    return RR_OK;
    RR_GUARD_END
}
int rr_stft_set_metering(rr_stft *h, double double_percentile, double sample_rate, double *d_bandwidth, double *d_energy,
                         size_t cap_frames, int store_spectra) {
    RR_CHECK_HANDLE(h, K_STFT);
    return set_sink(h->sink, double_percentile, sample_rate, d_bandwidth, d_energy, cap_frames, store_spectra);
}
int rr_stft_destroy(rr_stft *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_STFT);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

// ---- Fourier --------------------------------------------------------------------------
int rr_fourier_route(int dtype, size_t n, char *buf, size_t cap) {
    if (!buf || cap == 0) RR_FAIL(RR_ERR_BAD_ARG, "rr_fourier_route: no buffer");
    if (dtype != RR_F32 && dtype != RR_F64) RR_FAIL(RR_ERR_BAD_ARG, "rr_fourier_route: dtype");
    buf[0] = 0;
    RR_TRY(fourier_supported(dtype, n));
    const FourierRoute r = fourier_route(dtype, n, false);
    using FR = FourierRoute;
    switch (r.kind) {
        case FR::DIRECT: std::snprintf(buf, cap, "direct"); break;
        case FR::POW2: std::snprintf(buf, cap, "pow2"); break;
        case FR::BIG_TILE: std::snprintf(buf, cap, "pow2 two passes %zu x %zu", r.N1, r.N2); break;
        case FR::BIG_TRANSPOSE: std::snprintf(buf, cap, "pow2 five launches %zu x %zu", r.N1, r.N2); break;
        case FR::BIG_GENERIC: std::snprintf(buf, cap, "pow2 strided %zu x %zu", r.N1, r.N2); break;
        case FR::MIXED: {
            unsigned char rad[16];
            const int k = fft_mixed_radices(dtype, n, rad, 16);
            int pos = std::snprintf(buf, cap, "mixed");
            for (int i = 0; i < k && pos > 0 && (size_t)pos < cap; ++i) pos += std::snprintf(buf + pos, cap - pos, " %d", (int)rad[i]);
            break;
        }
        case FR::TILEM: std::snprintf(buf, cap, "mixed two passes %zu x %zu", r.N1, r.N2); break;
        case FR::BS_WAVE: std::snprintf(buf, cap, "bluestein wave M=%zu", r.M); break;
        case FR::BS_FUSED: std::snprintf(buf, cap, "bluestein one kernel M=%zu", r.M); break;
        case FR::BS_FUSED8K:
        case FR::BS_BIG:
        case FR::BS_LDS: std::snprintf(buf, cap, "bluestein one kernel M=%zu", r.M); break;
        case FR::BS_LAUNCHES: {
            // around the nested power-of-two transform: one launch each (M <= 8192 / 4096: five in all), its two passes with the
            // element-wise stages folded in (four), or its five launches (seventeen)
            const FourierRoute nested = fourier_route(dtype, r.M, false, true);
            const bool fused4 = nested.kind == FR::BIG_TILE &&
                                ![] { const char *e = std::getenv("RR_FOURIER_BS_FUSED"); return e && std::atoi(e) == 0; }();
            std::snprintf(buf, cap, "bluestein %s launches M=%zu", fused4 ? "four" : nested.kind == FR::POW2 ? "five" : "many", r.M);
            break;
        }
    }
    return RR_OK;
}

int rr_fourier_create(int dtype, const rr_window *window, int center_dc, int device, rr_fourier **out) {
    RR_GUARD_BEGIN
    if (!out || !window) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    if (window->kind < RR_WIN_RECTANGULAR || window->kind > RR_WIN_SAMPLED) RR_FAIL(RR_ERR_BAD_ARG, "unknown window kind");
    auto *h = new rr_fourier;
    int s = h->init_base(K_FOURIER, dtype, device);
    if (s != RR_OK) {
        delete h;
        return s;
    }
    h->window = *window;
    h->center_dc = center_dc != 0;
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_fourier_set_sampled_window(rr_fourier *h, size_t n, const double *window_rel) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FOURIER);
    if (h->window.kind != RR_WIN_SAMPLED) RR_FAIL(RR_ERR_BAD_ARG, "Fourier was not created with RR_WIN_SAMPLED");
    if (n && !window_rel) RR_FAIL(RR_ERR_BAD_ARG, "null");
    h->sampled.assign(window_rel, window_rel + n);
    h->sampled_n = n;
    h->n = 0;  // force a redesign at the next chunk
    return RR_OK;
    RR_GUARD_END
}
static int fourier_host(rr_fourier *h, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out,
                        bool blocking) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FOURIER);
    if (n_out) *n_out = 0;
    if (n_in == 0) RR_FAIL(RR_ERR_CONTRACT, "Fourier: empty chunk");
    if (n_in > cap) RR_FAIL(RR_ERR_CAPACITY, "Fourier: out_cap %zu < %zu", cap, n_in);
    RR_TRY(fourier_supported(h->dtype, n_in));
    RR_TRY(host_io(h, in, n_in, out, n_in, blocking, [&](void *di, void *dout, size_t *p) {
        return h->process_dev(n_in, di, n_in, dout, n_in, p);
    }));
    if (n_out) *n_out = n_in;
    return RR_OK;
    RR_GUARD_END
}
int rr_fourier_process(rr_fourier *h, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    return fourier_host(h, in, n_in, out, cap, n_out, true);
}
int rr_fourier_enqueue(rr_fourier *h, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    return fourier_host(h, in, n_in, out, cap, n_out, false);
}
int rr_fourier_process_dev(rr_fourier *h, size_t chunk_len, const void *d_in, size_t n_in, void *d_out, size_t cap,
                           size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FOURIER);
    return h->process_dev(chunk_len, d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_fourier_destroy(rr_fourier *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_FOURIER);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

int rr_channelizer_create_ex(int dtype, size_t bins, size_t taps_per_branch, size_t hop, const rr_window *window, int device,
                             rr_channelizer **out) {
    RR_GUARD_BEGIN
    if (!out || !window) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    if (taps_per_branch == 0) RR_FAIL(RR_ERR_CONTRACT, "chunk count must be positive");  // chunks.rs:195
    if (hop == 0) hop = bins;
    if (bins < 2 || bins > ((size_t)1 << 20)) RR_FAIL(RR_ERR_BAD_ARG, "Channelizer: bins must be in [2, 2^20]");
    if (hop > bins || (bins * taps_per_branch) % hop)
        RR_FAIL(RR_ERR_BAD_ARG, "Channelizer: the hop (%zu) must divide bins * taps_per_branch (%zu) and not exceed bins", hop,
                bins * taps_per_branch);
    if (window->kind != RR_WIN_RECTANGULAR && window->kind != RR_WIN_KAISER)
        RR_FAIL(RR_ERR_BAD_ARG, "Channelizer: window must be a built-in window");
    const bool fast = channelizer_fused_supported(dtype, bins, taps_per_branch, hop);  // one fused fold + FFT kernel
    auto *h = new rr_channelizer;
    int st = h->init_base(K_CHANNELIZER, dtype, device);
    if (st != RR_OK) {
        delete h;
        return st;
    }
    h->M = bins;
    h->P = taps_per_branch;
    h->hop = hop;
    const size_t n = bins * taps_per_branch;
    std::vector<double> rel(n), vals(n), tw(bins);
    st = window_sample(window, n, rel.data());
    if (st == RR_OK) st = fourier_design_window(n, rel.data(), vals.data());  // analysis.rs:88-101 over the P*M span
    for (size_t k = 0; k < bins / 2; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)bins;
        tw[2 * k] = std::cos(ang);
        tw[2 * k + 1] = std::sin(ang);
    }
    std::vector<unsigned char> wb, tb;
    if (dtype == RR_F32 && bins == 1024) {
        // k_fft1024<FOLD>: all 1024 twiddles and the lane seeds of the wave-level transform behind them
        cast_to<float>(vals.data(), n, wb);
        std::vector<float> twb(2 * 1024);
        for (size_t k = 0; k < 1024; ++k) {
            const double ang = -2.0 * M_PI * (double)k / 1024.0;
            twb[2 * k] = (float)std::cos(ang);
            twb[2 * k + 1] = (float)std::sin(ang);
        }
        append_wave1024_seeds(twb);
        tb.resize(twb.size() * sizeof(float));
        std::memcpy(tb.data(), twb.data(), tb.size());
    } else if (dtype == RR_F32) {
        cast_to<float>(vals.data(), n, wb);
        cast_to<float>(tw.data(), bins, tb);
    } else {
        cast_to<double>(vals.data(), n, wb);
        cast_to<double>(tw.data(), bins, tb);
    }
    if (st == RR_OK) st = upload(h->d_window, wb.data(), wb.size(), h->stream);
    if (st == RR_OK) st = upload(h->d_tw, tb.data(), tb.size(), h->stream);
    const size_t hb = (n - hop) * elem_size(dtype);
    if (st == RR_OK) st = h->hist[0].reserve(hb ? hb : 16);
    if (st == RR_OK) st = h->hist[1].reserve(hb ? hb : 16);
    if (st == RR_OK && !fast) {
        h->fo = new rr_fourier;  // rectangular window (all ones), no DC centring: the bare M-point transform
        st = h->fo->init_base(K_FOURIER, dtype, device);
        if (st == RR_OK) st = fourier_supported(dtype, bins);
    }
    if (st != RR_OK) {
        delete h;
        return st;
    }
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_channelizer_create(int dtype, size_t bins, size_t taps_per_branch, const rr_window *window, int device,
                          rr_channelizer **out) {
    return rr_channelizer_create_ex(dtype, bins, taps_per_branch, 0, window, device, out);
}
int rr_channelizer_reset(rr_channelizer *h) {
    RR_CHECK_HANDLE(h, K_CHANNELIZER);
    h->have_chunks = 0;
    return RR_OK;
}
int rr_channelizer_peek(const rr_channelizer *h, size_t n_in, size_t *n_out) {
    RR_CHECK_HANDLE(h, K_CHANNELIZER);
    if (!n_out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    if (n_in % h->hop) RR_FAIL(RR_ERR_BAD_ARG, "Channelizer: input must be whole chunks of %zu samples", h->hop);
    *n_out = h->peek(n_in);
    return RR_OK;
}
int rr_channelizer_process_dev(rr_channelizer *h, const void *d_in, size_t n_in, void *d_out, size_t cap,
                               size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_CHANNELIZER);
    return h->process_dev(d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_channelizer_process(rr_channelizer *h, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_CHANNELIZER);
    if (n_out) *n_out = 0;
    if (n_in % h->hop) RR_FAIL(RR_ERR_BAD_ARG, "Channelizer: input must be whole chunks of %zu samples", h->hop);
    const size_t produce = h->peek(n_in);
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Channelizer: out_cap %zu < %zu", cap, produce);
    size_t got = 0;
    RR_TRY(host_io(h, in, n_in, out, produce, true, [&](void *di, void *dout, size_t *p) {
        int s = h->process_dev(di, n_in, dout, produce, p);
        got = *p;
        return s;
    }));
    if (n_out) *n_out = got;
    return RR_OK;
    RR_GUARD_END
}
int rr_channelizer_destroy(rr_channelizer *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_CHANNELIZER);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

}  // extern "C"
