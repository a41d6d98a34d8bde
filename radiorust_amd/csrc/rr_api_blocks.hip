// rr_api_blocks.hip — FreqShifter, Filter, Downsampler, Upsampler, FmDemod: each block's host logic (the state one reference
// task keeps in its closure, the per-call choice of kernels) and its extern "C" entry points.
#include "rr_api_common.hpp"

// ---------------------------------------------------------------------------
// FreqShifter
// ---------------------------------------------------------------------------
int rr_freqshifter::prepare(double sample_rate) {
    const bool recalculate = shift_changed || !have_rate || sample_rate != prev_rate;  // transform.rs:318-319
    have_rate = true;
    prev_rate = sample_rate;
    if (!recalculate) return RR_OK;
    int64_t nu = 0, de = 0;
    RR_TRY(freq_to_ratio(sample_rate, precision, shift, &nu, &de));
    if (de > (int64_t(1) << 28))
        RR_FAIL(RR_ERR_BAD_ARG, "FreqShifter: phase table of %lld entries is not supported (raise `precision`)",
                (long long)de);
    // phase continuity (transform.rs:322-325): arg() of the current phasor, in Flt
    double start = 0.0;
    const size_t esz = elem_size(dtype);
    if (!host_table.empty()) {
        if (dtype == RR_F32) {
            const float *t = reinterpret_cast<const float *>(host_table.data()) + 2 * phase_idx;
            start = atan2f(t[1], t[0]);
        } else {
            const double *t = reinterpret_cast<const double *>(host_table.data()) + 2 * phase_idx;
            start = std::atan2(t[1], t[0]);
        }
    }
    numer = nu;
    denom = de;
    phase_idx = 0;
    shift_changed = false;
    ++table_version;
    // denom entries + entry 0 once more behind them (k_ols_wave reads the pair (r, r + 1) in one piece)
    // + 8 rotations e^{j 2 pi (128 k numer mod denom) / denom}, k < 8: the fused kernel steps a lane's
    // phasor by 128 samples with one product instead of one more table read
    // + 3 x 64 lane constants w^(4 l), w^(1024 - 2 l), w^(8 (l mod 32)), w = e^{j 2 pi numer / denom}, l < 64: the kernels with the
    // mixer behind the filter (k_ols_frame / k_ols_wave<.., GP>) make the phasor of a lane's result from the block's own (one
    // scalar read) and the lane's constant - results l + 64 c at 4 : 1, (512 - l - 64 k) mod 512 at 2 : 1, (l mod 32) + 32 c at 8 : 1
    host_table.resize(((size_t)de + 1 + 8 + 192) * esz);
    if (dtype == RR_F32)
        nco_table<float>(nu, de, (float)start, reinterpret_cast<float *>(host_table.data()));
    else
        nco_table<double>(nu, de, start, reinterpret_cast<double *>(host_table.data()));
    std::memcpy(host_table.data() + (size_t)de * esz, host_table.data(), esz);
    for (int k = 0; k < 8 + 192; ++k) {
        const int l = (k - 8) & 63;
        const int64_t step = k < 8 ? 128 * k : k < 72 ? 4 * l : k < 136 ? 1024 - 2 * l : 8 * (l & 31);
        const int64_t i = (int64_t)(((__int128)step * (__int128)nu) % (__int128)de);
        const double ang = 2.0 * M_PI * (double)i / (double)de;
        unsigned char *dst = host_table.data() + ((size_t)de + 1 + k) * esz;
        if (dtype == RR_F32) {
            const float v[2] = {(float)std::cos(ang), (float)std::sin(ang)};
            std::memcpy(dst, v, sizeof v);
        } else {
            const double v[2] = {std::cos(ang), std::sin(ang)};
            std::memcpy(dst, v, sizeof v);
        }
    }
    return upload(d_table, host_table.data(), host_table.size(), stream);
}

int rr_freqshifter::process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap,
                                size_t *n_out) {
    if (n_out) *n_out = 0;
    if (n_in > cap) RR_FAIL(RR_ERR_CAPACITY, "FreqShifter: out_cap %zu < %zu", cap, n_in);
    RR_TRY(select());
    RR_TRY(prepare(sample_rate));
    RR_TRY(launch_freqshift(dtype, stream, d_in, d_out, n_in, d_table.p, (uint32_t)denom, (uint32_t)phase_idx));
    phase_idx = (phase_idx + n_in % (uint64_t)denom) % (uint64_t)denom;
    if (n_out) *n_out = n_in;
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Filter
// ---------------------------------------------------------------------------
// round-to-nearest-even conversion of a finite float to IEEE binary16 bits (host side)
static uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));  // overflow / nan
    if (x < 0x38800000u) {  // subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 126 - (int)(x >> 23);  // 14 .. 24
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}


int rr_filter::design(double sample_rate, size_t len, const rr_c64 *resp, const double *window_rel) {
    RR_TRY(select());
    if (len > (size_t(1) << 24)) RR_FAIL(RR_ERR_BAD_ARG, "Filter: chunk length %zu is not supported", len);
    std::vector<cd> g(len);
    RR_TRY(filter_design_taps(len, resp, window_rel, g.data()));
    taps_base.swap(g);
    design_rate = sample_rate;
    return build_tables(true);
}

int rr_filter::set_gain(double g) {
    if (g == gain) return RR_OK;
    gain = g;
    if (!designed) return RR_OK;
    RR_TRY(select());
    return build_tables(false);  // (the previous chunk stays: GainControl has no state of its own)
}

// every kernel's tables from taps_base * gain
int rr_filter::build_tables(bool reset_history) {
    const size_t len = taps_base.size();
    const double sample_rate = design_rate;
    std::vector<cd> g(taps_base);
    {
        const double gf = gain_as_flt(dtype, gain);
        if (gf != 1.0)
            for (cd &v : g) v *= gf;
    }
    double max_re = 0.0, max_im = 0.0;
    for (const cd &v : g) {
        max_re = std::fmax(max_re, std::fabs(v.real()));
        max_im = std::fmax(max_im, std::fabs(v.imag()));
    }
    // A real-even response gives taps that are real up to ~1e-17 relative
    // rounding residue of the f64 inverse transform; carrying that residue costs
    // 2x the flops and changes results by < 1e-12 relative, far below f32 epsilon.
    real_taps = max_im <= 1e-12 * max_re;
    // device order: w[j] = g[n-1-j] so that out[m] = sum_j w[j] x[e_m-(n-1)+j]
    std::vector<double> w(real_taps ? len : 2 * len);
    for (size_t j = 0; j < len; ++j) {
        const cd v = g[len - 1 - j];
        if (real_taps)
            w[j] = v.real();
        else {
            w[2 * j] = v.real();
            w[2 * j + 1] = v.imag();
        }
    }
    std::vector<unsigned char> bytes;
    if (dtype == RR_F32)
        cast_to<float>(w.data(), w.size(), bytes);
    else
        cast_to<double>(w.data(), w.size(), bytes);
    RR_TRY(upload(d_taps, bytes.data(), bytes.size(), stream));
    use_ols4096 = filter_ols4096_supported(dtype, len);
    // shorter power-of-two filters: the same kernel for long calls (a 4096-block per 4096 - n outputs),
    // k_fir for short ones
    big_ols4096 = dtype == RR_F32 && (len == 64 || len == 128);
    // longer ones: partitions of 2048 taps, g = sum_p delay(g_p, 2048 p), one accumulating launch per partition
    npart = (dtype == RR_F32 && len > 2048) ? (len + 2047) / 2048 : 0;
    if (use_ols4096 || big_ols4096 || npart) {
        const size_t parts = npart ? npart : 1;
        std::vector<float> gb(parts * 2 * 4096), twb(2 * 4096);
        for (size_t pt = 0; pt < parts; ++pt) {
            std::vector<cd> gg(4096, cd(0, 0));
            for (size_t i = 0; i < 2048 && pt * 2048 + i < len; ++i) gg[i] = g[pt * 2048 + i];
            fft_f64(gg, false);
            float *dst0 = gb.data() + pt * 2 * 4096;
            for (size_t i = 0; i < 4096; ++i) {
                // pair-interleaved for 16-byte reads: Gp[kp][j] = {G[j + 512 kp], G[j + 512 kp + 256]}, j < 256
                const size_t kp = i / 512, r = i % 512, dst = (kp * 256 + r % 256) * 2 + r / 256;
                dst0[2 * dst] = (float)(gg[i].real() / 4096.0);
                dst0[2 * dst + 1] = (float)(gg[i].imag() / 4096.0);
            }
        }
        for (size_t i = 0; i < 4096; ++i) {
            const double ang = -2.0 * M_PI * (double)i / 4096.0;
            twb[2 * i] = (float)std::cos(ang);
            twb[2 * i + 1] = (float)std::sin(ang);
        }
        RR_TRY(upload(d_G4096, gb.data(), gb.size() * sizeof(float), stream));
        RR_TRY(upload(d_tw4096, twb.data(), twb.size() * sizeof(float), stream));
        if (!npart) {
            std::vector<uint16_t> gh(gb.size());
            for (size_t i = 0; i < gb.size(); ++i) gh[i] = f32_to_f16_bits(gb[i]);
            RR_TRY(upload(d_G4096h, gh.data(), gh.size() * sizeof(uint16_t), stream));
        }
    }
    // 2049 .. 8192 taps: blocks of 8192 / 16 384 points in LDS, one forward and one inverse transform per block
    // (RR_FILTER_KERNEL=parts keeps the partitions of 2048 taps: A/B runs, tests; RR_FILTER_BLOCK=8192|16384 picks the block)
    use_ols16k = false;
    {
        const char *e = std::getenv("RR_FILTER_KERNEL");
        const char *be = std::getenv("RR_FILTER_BLOCK");
        const size_t forced = be ? (size_t)std::atol(be) : 0;
        const bool want = npart || (forced && len > 1024);
        if (want && filter_blkbig_supported(dtype, len) && !(e && (!std::strcmp(e, "parts") || !std::strcmp(e, "fir") || !std::strcmp(e, "ols4096")))) {
            const size_t V = (len + 63) / 64 * 64;
            // measured (profiles/r03_extras.txt, ms per 2^26 samples): 3072 taps 0.376 with blocks of 8192 (two workgroups per CU)
            // against ~0.43 with 16 384; 4096 taps 0.469 against 0.450
            size_t N = V <= 3072 ? 8192 : 16384;
            if ((forced == 8192 || forced == 16384) && V <= forced / 2) N = forced;
            std::vector<cd> gg(N, cd(0, 0));
            for (size_t i = 0; i < len; ++i) gg[i] = g[i];
            fft_f64(gg, false);
            std::vector<float> gb(2 * N), twb(2 * N);
            const size_t T = N / 16;
            for (size_t i = 0; i < N; ++i) {
                // pair-interleaved for 16-byte reads: Gp[kp][j] = {G[j + 2 T kp], G[j + 2 T kp + T]}, j < T
                const size_t kp = i / (2 * T), r = i % (2 * T), dst = (kp * T + r % T) * 2 + r / T;
                gb[2 * dst] = (float)(gg[i].real() / (double)N);
                gb[2 * dst + 1] = (float)(gg[i].imag() / (double)N);
                const double ang = -2.0 * M_PI * (double)i / (double)N;
                twb[2 * i] = (float)std::cos(ang);
                twb[2 * i + 1] = (float)std::sin(ang);
            }
            RR_TRY(upload(d_G16k, gb.data(), gb.size() * sizeof(float), stream));
            RR_TRY(upload(d_tw16k, twb.data(), twb.size() * sizeof(float), stream));
            V16k = V;
            N16k = N;
            use_ols16k = true;
        }
    }
    // long responses: overlap-save with blocks of 2^15 (f64: 2^13) .. 2^18 points through the two-pass tile transform (RR_FILTER_CONV=0: the
    // partitions of 2048 taps / k_filter_ols / k_fir as before)
    use_conv = false;
    {
        static const bool conv_off = [] { const char *e = std::getenv("RR_FILTER_CONV"); return e && std::atoi(e) == 0; }();
        // f32: the partitions of 2048 taps (one launch, n / 2048 + 1 transforms of 4096 points per block) stay ahead of the four
        // passes over HBM up to 8192 taps (measured, profiles/r03_extras.txt); RR_FILTER_CONV_MIN moves the threshold (A/B runs)
        const char *me = std::getenv("RR_FILTER_CONV_MIN");  // (per design: tests move it within one process)
        const size_t min32 = me ? (size_t)std::atol(me) : (size_t)16384;
        const size_t minlen = dtype == RR_F32 ? min32 : 4096;
        if (!conv_off && len >= minlen && len <= ((size_t)1 << 17)) {
            size_t N = (size_t)1 << (dtype == RR_F32 ? 15 : 13);  // (the lengths of the two-pass tile transform: f32 beyond k_fft16384's one image)
            while (N < 4 * len && N < ((size_t)1 << 18)) N <<= 1;
            if (N > len) {
                if (!conv_fft) {
                    conv_fft = new rr_fourier;
                    RR_TRY(conv_fft->init_base(K_FOURIER, dtype, device));
                }
                conv_fft->stream = stream;
                RR_TRY(conv_fft->prepare(N));  // rectangular window
                if (conv_fft->big && conv_fft->big_tile) {
                    // G' = DFT_N(g) e^{+j 2 pi k V / N} / N with V = len: the block's valid results (circular indices V .. N - 1) come out first
                    std::vector<cd> gg(N, cd(0, 0));
                    for (size_t i = 0; i < len; ++i) gg[i] = g[i];
                    fft_f64(gg, false);
                    std::vector<double> gd(2 * N), ones(2 * N);
                    for (size_t k = 0; k < N; ++k) {
                        const double ang = 2.0 * M_PI * (double)((k * len) % N) / (double)N;
                        const cd v = gg[k] * cd(std::cos(ang), std::sin(ang)) / (double)N;
                        gd[2 * k] = v.real();
                        gd[2 * k + 1] = v.imag();
                        ones[2 * k] = 1.0;
                        ones[2 * k + 1] = 0.0;
                    }
                    std::vector<unsigned char> gb, ob;
                    if (dtype == RR_F32) {
                        cast_to<float>(gd.data(), gd.size(), gb);
                        cast_to<float>(ones.data(), ones.size(), ob);
                    } else {
                        cast_to<double>(gd.data(), gd.size(), gb);
                        cast_to<double>(ones.data(), ones.size(), ob);
                    }
                    RR_TRY(upload(d_convG, gb.data(), gb.size(), stream));
                    RR_TRY(upload(d_ones, ob.data(), ob.size(), stream));
                    conv_N = N;
                    use_conv = true;
                }
            }
        }
    }
    if (use_conv) use_ols4096 = big_ols4096 = use_ols16k = false, npart = 0;
    {
        const char *e = std::getenv("RR_FILTER_KERNEL");  // "ols4096" / "fir" keep the older kernels (A/B runs, tests)
        // (k_filter_wave up to 256 taps: beyond them its 1024-sample blocks keep less than 75 % and the 4096-point blocks are ahead -
        //  320 taps 0.218 against 0.207 ms per 2^26 samples, 384 taps 0.235 against 0.205, 256 taps 0.207 against 0.203: scripts/filter_ab_probe.py;
        //  RR_FILTER_KERNEL=wave keeps it up to its 385)
        const bool wave_all = e && !std::strcmp(e, "wave");
        use_wave = filter_wave_supported(dtype, len) && (len <= 256 || wave_all) && !(e && (!std::strcmp(e, "ols4096") || !std::strcmp(e, "fir")));
        if (e && !std::strcmp(e, "fir")) big_ols4096 = false;
    }
    if (use_wave) {
        std::vector<double> c(len);
        for (size_t i = 0; i < len; ++i) c[i] = g[i].real();
        FusedFirTables t;
        build_fused_fir_tables(rr_chain::FK_OLSW, 1, c, g, t);
        RR_TRY(upload(d_Hw, t.H.data(), t.H.size() * sizeof(float), stream));
        RR_TRY(upload(d_tww, t.tw.data(), t.tw.size() * sizeof(float), stream));
        wave_V = t.V;
    }
    // Complex<f64>, up to 2049 taps: blocks of 4096 points (k_ols4096_f64; RR_FILTER_KERNEL=fir|ols keeps k_fir / k_filter_ols<double>)
    use_ols64 = false;
    {
        const char *e = std::getenv("RR_FILTER_KERNEL");
        if (dtype == RR_F64 && len >= 2 && len - 1 <= 2048 && !use_conv && !(e && (!std::strcmp(e, "fir") || !std::strcmp(e, "ols")))) {
            std::vector<double> G, tw;
            ols64_tables(g, G, tw);
            RR_TRY(upload(d_G64, G.data(), G.size() * sizeof(double), stream));
            RR_TRY(upload(d_tw64, tw.data(), tw.size() * sizeof(double), stream));
            V64 = std::max<size_t>(64, (len - 1 + 63) / 64 * 64);
            use_ols64 = true;
        }
    }
    use_ols = !use_ols4096 && !big_ols4096 && !npart && ols_supported(dtype, len);
    if (use_ols) {
        // the reference's extended response (filters.rs:220-238), transformed in f64 here
        std::vector<cd> ext(2 * len, cd(0, 0));
        for (size_t i = 0; i < len; ++i) ext[len + i] = g[i] / (2.0 * (double)len);
        fft_f64(ext, false);
        std::vector<double> hh(4 * len), tw(2 * len);
        for (size_t i = 0; i < 2 * len; ++i) {
            hh[2 * i] = ext[i].real();
            hh[2 * i + 1] = ext[i].imag();
        }
        for (size_t k = 0; k < len; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)(2 * len);
            tw[2 * k] = std::cos(ang);
            tw[2 * k + 1] = std::sin(ang);
        }
        std::vector<unsigned char> hb2, tb2;
        if (dtype == RR_F32) {
            cast_to<float>(hh.data(), hh.size(), hb2);
            cast_to<float>(tw.data(), tw.size(), tb2);
        } else {
            cast_to<double>(hh.data(), hh.size(), hb2);
            cast_to<double>(tw.data(), tw.size(), tb2);
        }
        RR_TRY(upload(d_H, hb2.data(), hb2.size(), stream));
        RR_TRY(upload(d_olstw, tb2.data(), tb2.size(), stream));
    }
    taps_f64.swap(g);
    ++design_version;
    if (!reset_history) return RR_OK;
    const size_t hb = len * elem_size(dtype);
    RR_TRY(hist[0].reserve(hb));
    RR_TRY(hist[1].reserve(hb));
    n = len;
    rate = sample_rate;
    designed = true;
    params_changed = false;
    hist_valid = false;  // previous_chunk = None (filters.rs:187)
    cur = 0;
    return RR_OK;
}

rr_filter::~rr_filter() { delete conv_fft; }

// out[m] = sum_k g[k] x[e0 + m - k] for the call's `produce` outputs, e0 = 0 with a previous chunk in hist and n without:
// block f takes the stream's samples [f hop - V, f hop - V + N), V = n, hop = N - V, and yields outputs [f hop, (f + 1) hop)
int rr_filter::process_conv(const void *d_in, size_t n_in, void *d_out, size_t produce) {
    const size_t esz = elem_size(dtype), N = conv_N, V = n, hop = N - V;
    size_t N1, N2;
    fft_big_split(N, &N1, &N2);
    rr_fourier *ff = conv_fft;
    ff->stream = stream;
    const char *tB = ff->d_tw.as<char>(), *tA = tB + ((size_t)1 << ff->big_h) * esz;
    const char *tw1 = tB + ff->big_tw1_off * esz, *tw2 = tB + ff->big_tw2_off * esz;
    const int hh = ff->big_h;
    // the stream in front of output 0: the previous chunk (V samples), or - first chunk after a reset - the call's own first chunk
    const char *head = hist_valid ? hist[cur].as<char>() : static_cast<const char *>(d_in);
    const char *src = hist_valid ? static_cast<const char *>(d_in) : static_cast<const char *>(d_in) + V * esz;
    const size_t n_src = hist_valid ? n_in : n_in - V;
    const size_t frames = (produce + hop - 1) / hop;
    size_t per_pass = ((size_t)1 << 23) / N;
    if (per_pass < 1) per_pass = 1;
    if (per_pass > 65535) per_pass = 65535;
    if (per_pass > frames) per_pass = frames;
    RR_TRY(conv_ws[0].reserve(per_pass * N * esz));
    RR_TRY(conv_ws[1].reserve(per_pass * N * esz));
    for (size_t f0 = 0; f0 < frames; f0 += per_pass) {
        const size_t F = frames - f0 < per_pass ? frames - f0 : per_pass;
        const size_t skip = f0 * hop;  // samples of [head | src] in front of this pass's first block
        const char *hd = head, *sp = src;
        size_t nh = V;
        long lim = (long)n_src;
        if (skip >= V) {
            sp += (skip - V) * esz;
            lim -= (long)(skip - V);
            nh = 0;
        } else {
            hd += skip * esz;
            nh = V - skip;
        }
        RR_TRY(launch_fft_tile_bs(dtype, stream, 0, hd, nh, sp, hop, conv_ws[0].p, N1, N2, F, N, d_ones.p, tw1, tB, tA, hh, 0, lim, 0));
        RR_TRY(launch_fft_tile_bs(dtype, stream, 1, nullptr, 0, conv_ws[0].p, 0, conv_ws[1].p, N1, N2, F, N, d_convG.p, tw2, nullptr,
                                  nullptr, 0, 0));
        RR_TRY(launch_fft_tile_bs(dtype, stream, 2, nullptr, 0, conv_ws[1].p, 0, conv_ws[0].p, N1, N2, F, N, nullptr, tw1, tB, tA, hh, 0));
        RR_TRY(launch_fft_tile_bs(dtype, stream, 3, nullptr, 0, conv_ws[0].p, 0, static_cast<char *>(d_out) + f0 * hop * esz, N1, N2, F, hop,
                                  d_ones.p, tw2, nullptr, nullptr, 0, 0, 0, (long)(produce - f0 * hop)));
    }
    return RR_OK;
}

// n = 64, 128: calls that produce fewer samples than this stay on k_fir
static constexpr size_t kFilterBigCall = 16384;

int rr_filter::process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap,
                           size_t *n_out, bool out_f16, bool g_f16) {
    if (n_out) *n_out = 0;
    if ((out_f16 || g_f16) && !(designed && use_ols4096))
        RR_FAIL(RR_ERR_BAD_ARG, "Filter: half-precision output/response exists for Complex<f32>, n = 129 .. 2048");
    if (!designed || params_changed || sample_rate != rate)
        RR_FAIL(RR_ERR_NEED_DESIGN, "Filter: no design for sample rate %g (filters.rs:178-183)", sample_rate);
    if (n_in == 0) return RR_OK;
    if (n_in % n != 0)
        RR_FAIL(RR_ERR_NEED_DESIGN, "Filter: %zu samples is not a whole number of chunks of the designed length %zu", n_in, n);
    const size_t produce = peek(n_in);
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Filter: out_cap %zu < %zu", cap, produce);
    RR_TRY(select());
    last_kernel = 0;
    if (produce && use_conv && !out_f16 && !g_f16) {
        RR_TRY(process_conv(d_in, n_in, d_out, produce));
        last_kernel = 4;
        RR_TRY(launch_update_hist(dtype, stream, hist[cur].p, hist[cur ^ 1].p, n, d_in, n_in));
        cur ^= 1;
        hist_valid = true;
        if (n_out) *n_out = produce;
        return RR_OK;
    }
    if (produce && use_wave && produce >= kFilterBigCall && !out_f16 && !g_f16) {
        RR_TRY(launch_filter_wave(stream, hist[cur].p, hist_valid ? n : 0, d_in, n_in, d_Hw.p, d_tww.p, wave_V, d_out, produce,
                                  hist_valid ? 0 : (long)n));
        last_kernel = 3;
    } else if (produce && use_ols16k && !out_f16 && !g_f16) {
        last_kernel = 5;
        RR_TRY(launch_filter_blkbig(stream, N16k, hist[cur].p, hist_valid ? n : 0, d_in, n_in, d_G16k.p, d_tw16k.p, V16k, d_out, produce,
                                    hist_valid ? 0 : (long)n, hist[cur ^ 1].p, n));
        cur ^= 1;
        hist_valid = true;
        if (n_out) *n_out = produce;
        return RR_OK;
    } else if (produce && npart) {
        last_kernel = 2;
        // out[m] = sum_p sum_{k < 2048} g[2048 p + k] x[e0 + m - 2048 p - k]: partition p is the 2048-tap kernel run on
        // the stream delayed by 2048 p; the last launch also leaves the next call's history
        static const bool per_launch = [] { const char *e = std::getenv("RR_FILTER_PARTS"); return e && !std::strcmp(e, "acc"); }();
        if (!per_launch) {
            // one launch: the workgroup of a block transforms the stream at every partition's delay and sums the products
            // before ONE inverse (npart + 1 transforms per block; RR_FILTER_PARTS=acc keeps a launch per partition)
            RR_TRY(launch_filter_blk4096(stream, hist[cur].p, hist_valid ? n : 0, d_in, n_in, d_G4096.p, d_tw4096.p, 2048, d_out,
                                         produce, hist_valid ? 0 : (long)n, false, false, hist[cur ^ 1].p, n, false, npart));
        } else
        for (size_t pt = 0; pt < npart; ++pt)
            RR_TRY(launch_filter_blk4096(stream, hist[cur].p, hist_valid ? n : 0, d_in, n_in,
                                         static_cast<const char *>(d_G4096.p) + pt * 2 * 4096 * sizeof(float), d_tw4096.p, 2048,
                                         d_out, produce, (hist_valid ? 0 : (long)n) - (long)(2048 * pt), false, false,
                                         pt + 1 == npart ? hist[cur ^ 1].p : nullptr, n, pt > 0));
        cur ^= 1;
        hist_valid = true;
        if (n_out) *n_out = produce;
        return RR_OK;
    } else if (produce && (use_ols4096 || (big_ols4096 && produce >= kFilterBigCall))) {
        last_kernel = 2;
        // (the kernel's last workgroup also leaves the next call's history)
        RR_TRY(launch_filter_blk4096(stream, hist[cur].p, hist_valid ? n : 0, d_in, n_in, g_f16 ? d_G4096h.p : d_G4096.p,
                                     d_tw4096.p, n, d_out, produce, hist_valid ? 0 : (long)n, out_f16, g_f16, hist[cur ^ 1].p, n));
        cur ^= 1;
        hist_valid = true;
        if (n_out) *n_out = produce;
        return RR_OK;
    } else if (produce && use_ols64 && produce >= 4096) {
        last_kernel = 6;
        RR_TRY(launch_ols4096_f64(stream, hist[cur].p, hist_valid ? n : 0, d_in, n_in, d_G64.p, d_tw64.p, V64, 1, d_out, produce,
                                  hist_valid ? 0 : (long)n, hist[cur ^ 1].p, n, nullptr, 0, 0));
        cur ^= 1;
        hist_valid = true;
        if (n_out) *n_out = produce;
        return RR_OK;
    } else if (produce && use_ols) {
        last_kernel = 1;
        RR_TRY(launch_filter_ols(dtype, stream, hist[cur].p, d_in, n, produce / n, hist_valid ? 0 : 1, d_H.p, d_olstw.p, d_out));
    } else if (produce) {
        FirArgs a;
        a.hist = hist[cur].p;
        a.hist_len = hist_valid ? n : 0;
        a.in = d_in;
        a.n_in = n_in;
        a.taps = d_taps.p;
        a.K = (uint32_t)n;
        a.complex_taps = !real_taps;
        a.out = d_out;
        a.n_out = produce;
        a.e0 = hist_valid ? 0 : n;  // the first chunk after a reset is swallowed (filters.rs:240,260)
        a.D = 1;
        RR_TRY(launch_fir(dtype, stream, a));
    }
    // previous_chunk = Some(input_chunk): the last n samples
    RR_TRY(launch_update_hist(dtype, stream, hist[cur].p, hist[cur ^ 1].p, n, d_in, n_in));
    cur ^= 1;
    hist_valid = true;
    if (n_out) *n_out = produce;
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Downsampler
// ---------------------------------------------------------------------------
// Calls shorter than this stay on k_fir (a fused kernel's launch needs whole blocks to pay off)
static constexpr size_t kFastMinSamples = 4096;

int rr_downsampler::prepare(double input_rate) {
    if (have_rate && input_rate == prev_rate) return RR_OK;
    std::vector<double> ir;
    RR_TRY(downsampler_design(input_rate, output_rate, bandwidth, quality, ir));
    have_rate = true;
    prev_rate = input_rate;
    L = ir.size();
    ir_base.swap(ir);
    RR_TRY(set_gain(gain));  // ir_f64 = gain * ir_base, uploaded; ++design_version
    const size_t hb = L * elem_size(dtype);
    RR_TRY(hist[0].reserve(hb));
    RR_TRY(hist[1].reserve(hb));
    RR_HIP(hipMemsetAsync(hist[0].p, 0, hb, stream));  // ringbuf = vec![0; ir_len]
    cur = 0;
    sched.configure(input_rate, output_rate);  // pos = 0
    return RR_OK;
}

// (also the tail of prepare(): the tables every kernel reads, from ir_base and the gain; history and schedule stay)
int rr_downsampler::set_gain(double g) {
    gain = g;
    if (ir_base.empty()) return RR_OK;
    RR_TRY(select());
    const double gf = gain_as_flt(dtype, gain);
    ir_f64.resize(ir_base.size());
    for (size_t i = 0; i < ir_base.size(); ++i) ir_f64[i] = gf == 1.0 ? ir_base[i] : gf * ir_base[i];
    std::vector<unsigned char> bytes;
    if (dtype == RR_F32)
        cast_to<float>(ir_f64.data(), ir_f64.size(), bytes);
    else
        cast_to<double>(ir_f64.data(), ir_f64.size(), bytes);
    RR_TRY(upload(d_ir, bytes.data(), bytes.size(), stream));
    ++design_version;
    return RR_OK;
}

int rr_downsampler::peek(double input_rate, size_t n_in, size_t *n_out) {
    if (have_rate && input_rate == prev_rate) {
        *n_out = sched.count(n_in);
        return RR_OK;
    }
    if (!(input_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be positive");
    if (!(input_rate >= output_rate))
        RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be greater than or equal to output sample rate");
    Schedule tmp;
    tmp.configure(input_rate, output_rate);
    *n_out = tmp.count(n_in);
    return RR_OK;
}

// true when a call of n_in samples at this rate would run k_decim_poly, which can take a FreqShifter's table along
bool rr_downsampler::can_fuse_mixer(double input_rate, size_t n_in) {
    size_t produce = 0;
    if (dtype != RR_F32 || n_in < kFastMinSamples || peek(input_rate, n_in, &produce) != RR_OK || !produce) return false;
    if (select() != RR_OK || prepare(input_rate) != RR_OK || ensure_fast() != RR_OK) return false;
    return fast_kind == rr_chain::FK_POLY;
}

int rr_downsampler::process_dev(double input_rate, const void *d_in, size_t n_in, void *d_out, size_t cap,
                                size_t *n_out, const void *nco, uint32_t nco_denom, uint32_t nco_idx0) {
    if (n_out) *n_out = 0;
    size_t produce = 0;
    RR_TRY(peek(input_rate, n_in, &produce));
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Downsampler: out_cap %zu < %zu", cap, produce);
    if (n_in > 0xfffffff0ull) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: more than 2^32 samples in one call");
    RR_TRY(select());
    RR_TRY(prepare(input_rate));
    if (n_in == 0) return RR_OK;
    // The schedule (resampling.rs:110-112) is advanced on a COPY; `sched`, `cur` and the history are committed
    // only after the last launch of this call has been accepted, so that a failing call (a reserve, an upload,
    // a kernel's precondition) leaves the block where it was.
    rr::Schedule next = sched;
    last_kernel = 0;
    if (produce && n_in >= kFastMinSamples) {
        RR_TRY(ensure_fast());
        if (fast_kind == rr_chain::FK_POLY) {
            // any integer ratio, and rational ratios with a short period: k_decim_poly (rr_decim.hip)
            int64_t e_first[8];
            sched.first_emits((size_t)std::min<uint64_t>(sched.Q, produce), e_first);
            for (uint64_t b = produce; b < sched.Q; ++b) e_first[b] = e_first[0];  // (fewer outputs than one period)
            RR_TRY(ensure_poly_taps(e_first));
            next.advance(n_in, nullptr);
            RR_TRY(launch_decim_poly(stream, hist[cur].p, L, d_in, n_in, f_ctaps.p, sched.P, sched.Q, f_NC, L, e_first[0],
                                     d_out, produce, hist[cur ^ 1].p, L, nco, nco_denom, nco_idx0, dtype));
            sched = next;
            cur ^= 1;
            last_kernel = fast_kind;
            if (n_out) *n_out = produce;
            return RR_OK;
        }
        if (nco) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: a mixer can only ride along with k_decim_poly (can_fuse_mixer)");
        if (fast_kind == rr_chain::FK_OLS64) {
            // out[m] = sum_i c[i] x[e0 + D m - i], c = reverse(ir); the kernel's last workgroup leaves the last L samples as the history
            next.advance(n_in, nullptr);
            RR_TRY(launch_ols4096_f64(stream, hist[cur].p, L, d_in, n_in, f_H.p, f_tw.p, f_V64, sched.D, d_out, produce,
                                      (long)sched.first_emit(), hist[cur ^ 1].p, L, nullptr, 0, 0));
            sched = next;
            cur ^= 1;
            last_kernel = fast_kind;
            if (n_out) *n_out = produce;
            return RR_OK;
        }
        if (fast_kind == rr_chain::FK_SELECT) {
            next.advance(n_in, nullptr);
            if (f_V == 0)
                RR_TRY(launch_decim_select_blk(stream, hist[cur].p, L, d_in, n_in, f_H.p, f_tw.p, L, d_out, produce, sched.ra, sched.rb,
                                               sched.pos_units()));
            else
            RR_TRY(launch_decim_select(stream, hist[cur].p, L, d_in, n_in, f_H.p, f_tw.p, f_V, d_out, produce, sched.ra, sched.rb,
                                       sched.pos_units()));
            RR_TRY(launch_update_hist(dtype, stream, hist[cur].p, hist[cur ^ 1].p, L, d_in, n_in));
            sched = next;
            cur ^= 1;
            last_kernel = fast_kind;
            if (n_out) *n_out = produce;
            return RR_OK;
        }
        if (fast_kind != rr_chain::FK_NONE && sched.integer_ratio) {
            // the chain's kernels with every phasor = 1: out[m] = sum_i c[i] x[e0 + D m - i], c = reverse(ir);
            // the kernel's last workgroup leaves the last L samples as the next call's history
            FusedFirArgs f;
            f.xh = hist[cur].p;
            f.hx = L;
            f.in = d_in;
            f.n_in = n_in;
            f.nco = f_one.p;
            f.denom = 1;
            f.idx0 = 0;
            f.taps = f_ctaps.p;
            f.Gp = f_Gp;
            f.out = d_out;
            f.n_out = produce;
            f.e0 = (int64_t)sched.first_emit();
            f.D = (uint32_t)sched.D;
            f.xh_out = hist[cur ^ 1].p;
            f.H = f_H.p;
            f.tw4096 = f_tw.p;
            f.V = f_V;
            f.poly = f_poly;
            f.blk = f_blk;
            f.mixfold = true;  // every phasor is 1: the instances without a mixer (k_ols_wave<D, true, true>)
            next.advance(n_in, nullptr);
            if (fast_kind == rr_chain::FK_OLSW)
                RR_TRY(launch_ols_wave(stream, f));
            else if (fast_kind == rr_chain::FK_OLS)
                RR_TRY(launch_ols_decim(stream, f));
            else
                RR_TRY(launch_fused_fir(stream, f));
            sched = next;
            cur ^= 1;
            last_kernel = fast_kind;
            if (n_out) *n_out = produce;
            return RR_OK;
        }
    }
    if (nco) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: a mixer can only ride along with k_decim_poly (can_fuse_mixer)");
    FirArgs a;
    a.hist = hist[cur].p;
    a.hist_len = L;
    a.in = d_in;
    a.n_in = n_in;
    a.taps = d_ir.p;
    a.K = (uint32_t)L;
    a.complex_taps = false;
    a.out = d_out;
    a.n_out = produce;
    if (sched.integer_ratio) {
        a.e0 = sched.first_emit();
        a.D = (uint32_t)sched.D;
        if (sched.D > 0xffffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Downsampler: decimation factor too large");
        next.advance(n_in, nullptr);
    } else if (sched.periodic && sched.P < (1ull << 31) && sched.Q <= (1u << 22)) {
        // integer rates: the first Q releases (closed form), repeated every P inputs - no list of the call's length
        next.advance(n_in, nullptr);
        if (produce) {
            std::vector<int64_t> ef(sched.Q);
            sched.first_emits(sched.Q, ef.data());
            emit.resize(sched.Q);
            for (size_t b = 0; b < sched.Q; ++b) emit[b] = (uint32_t)ef[b];
            RR_TRY(d_emit.reserve(sched.Q * sizeof(uint32_t)));
            RR_HIP(hipMemcpyAsync(d_emit.p, emit.data(), sched.Q * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            RR_HIP(hipStreamSynchronize(stream));  // `emit` is reused by the next call
        }
        a.emit = d_emit.as<uint32_t>();
        a.period_p = (uint32_t)sched.P;
        a.period_q = (uint32_t)sched.Q;
        a.max_step = (uint32_t)std::ceil(input_rate / output_rate) + 1;
    } else {
        next.advance(n_in, &emit);
        if (produce) {
            RR_TRY(d_emit.reserve(produce * sizeof(uint32_t)));
            RR_HIP(hipMemcpyAsync(d_emit.p, emit.data(), produce * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            RR_HIP(hipStreamSynchronize(stream));  // `emit` is reused by the next call
        }
        a.emit = d_emit.as<uint32_t>();
        a.max_step = (uint32_t)std::ceil(input_rate / output_rate) + 1;
    }
    if (produce) RR_TRY(launch_fir(dtype, stream, a));
    RR_TRY(launch_update_hist(dtype, stream, hist[cur].p, hist[cur ^ 1].p, L, d_in, n_in));
    sched = next;
    cur ^= 1;
    if (n_out) *n_out = produce;
    return RR_OK;
}

// k_decim_poly's tap table depends on where in its period the schedule stands at the start of the call (the offsets
// of the first Q emissions); rebuilt when that changes (calls of a whole number of periods keep it).
int rr_downsampler::ensure_poly_taps(const int64_t *e_first) {
    std::vector<int64_t> delta(sched.Q);
    for (uint64_t b = 0; b < sched.Q; ++b) delta[b] = e_first[b] - e_first[0];
    if (poly_version == design_version && delta == poly_delta) return RR_OK;
    std::vector<uint32_t> T;
    int lp = 0;
    build_decim_poly_taps(ir_f64, sched.P, sched.Q, e_first, T, &lp, dtype);
    RR_TRY(upload(f_ctaps, T.data(), T.size() * sizeof(uint32_t), stream));
    f_NC = lp;
    poly_delta.swap(delta);
    poly_version = design_version;
    return RR_OK;
}

// Tables for the fast path (integer ratio, f32).  RR_DOWNSAMPLER_GENERIC=1 keeps k_fir (A/B runs, tests).
int rr_downsampler::ensure_fast() {
    if (fast_version == design_version) return RR_OK;
    fast_version = design_version;
    fast_kind = rr_chain::FK_NONE;
    const char *e = std::getenv("RR_DOWNSAMPLER_GENERIC");
    if (!sched.periodic || (e && std::atoi(e) != 0)) return RR_OK;
    // (f64: no fused overlap-save kernels, the polyphase kernel for every periodic ratio it fits)
    int kind = (dtype == RR_F32 && sched.integer_ratio) ? rr_chain::pick_fused_kernel(sched.D, L, true, 0) : rr_chain::FK_NONE;
    if (kind == rr_chain::FK_OLSF) kind = rr_chain::FK_OLSW;
    if (mixer_rides && kind == rr_chain::FK_OLSW && sched.D >= 5 && sched.D != 8 && decim_poly_supported(dtype, sched.P, sched.Q, L)) kind = rr_chain::FK_NONE;
    // RR_DOWNSAMPLER_SELECT=1: k_filter_wave<true> for every pair of integer rates it takes (A/B runs, tests)
    const char *se = std::getenv("RR_DOWNSAMPLER_SELECT");
    const bool force_select = se && std::atoi(se) != 0 &&
                              (decim_select_supported(dtype, sched.ra, sched.rb, L) || decim_select_blk_supported(dtype, sched.ra, sched.rb, L));
    if (force_select) kind = rr_chain::FK_NONE;
    {
        // RR_DOWNSAMPLER_POLY=1: k_decim_poly also where a fused kernel applies (A/B runs)
        const char *pe = std::getenv("RR_DOWNSAMPLER_POLY");
        if (pe && std::atoi(pe) != 0 && decim_poly_supported(dtype, sched.P, sched.Q, L)) kind = rr_chain::FK_NONE;
    }
    // Complex<f64> at integer ratios: overlap-save in blocks of 4096 points (k_ols4096_f64; RR_DOWNSAMPLER_POLY=1 keeps the decimator)
    if (dtype == RR_F64 && sched.integer_ratio && !force_select) {
        const char *pe = std::getenv("RR_DOWNSAMPLER_POLY");
        const size_t V = ols4096_f64_overlap(L, sched.D);
        if (V && !(pe && std::atoi(pe) != 0)) {
            std::vector<cd> cc(L);
            for (size_t i = 0; i < L; ++i) cc[i] = cd(ir_f64[L - 1 - i], 0.0);
            std::vector<double> G, tw;
            ols64_tables(cc, G, tw);
            RR_TRY(upload(f_H, G.data(), G.size() * sizeof(double), stream));
            RR_TRY(upload(f_tw, tw.data(), tw.size() * sizeof(double), stream));
            f_V64 = V;
            fast_kind = rr_chain::FK_OLS64;
            return RR_OK;
        }
    }
    if (kind == rr_chain::FK_NONE) {
        // every other integer ratio, and rational ratios with a short period (the tap table follows per call)
        if (decim_poly_supported(dtype, sched.P, sched.Q, L) && !force_select) {
            fast_kind = rr_chain::FK_POLY;
            poly_version = ~0ull;
        } else if (!decim_select_supported(dtype, sched.ra, sched.rb, L) && decim_select_blk_supported(dtype, sched.ra, sched.rb, L)) {
            // the same with responses of 386 .. 2048 taps: 4096-point blocks (k_filter_blk4096<.., SEL>); the Filter's table layout
            std::vector<cd> gg(4096, cd(0, 0));
            for (size_t i = 0; i < L; ++i) gg[i] = cd(ir_f64[L - 1 - i], 0.0);
            fft_f64(gg, false);
            std::vector<float> gb(2 * 4096), twb(2 * 4096);
            for (size_t i = 0; i < 4096; ++i) {
                const size_t kp = i / 512, r = i % 512, dst = (kp * 256 + r % 256) * 2 + r / 256;
                gb[2 * dst] = (float)(gg[i].real() / 4096.0);
                gb[2 * dst + 1] = (float)(gg[i].imag() / 4096.0);
                const double ang = -2.0 * M_PI * (double)i / 4096.0;
                twb[2 * i] = (float)std::cos(ang);
                twb[2 * i + 1] = (float)std::sin(ang);
            }
            RR_TRY(upload(f_H, gb.data(), gb.size() * sizeof(float), stream));
            RR_TRY(upload(f_tw, twb.data(), twb.size() * sizeof(float), stream));
            f_V = 0;  // (0: the 4096-point blocks)
            fast_kind = rr_chain::FK_SELECT;
        } else if (decim_select_supported(dtype, sched.ra, sched.rb, L)) {
            // every other pair of integer rates (48 000 -> 44 100: 160 : 147): the response at every position, as the Filter's
            // k_filter_wave, and the results of the releasing positions stored (k_filter_wave<true>)
            std::vector<double> c(L);
            std::vector<cd> cc(L);
            for (size_t i = 0; i < L; ++i) {
                c[i] = ir_f64[L - 1 - i];
                cc[i] = cd(c[i], 0.0);
            }
            FusedFirTables t;
            build_fused_fir_tables(rr_chain::FK_OLSW, 1, c, cc, t);
            RR_TRY(upload(f_H, t.H.data(), t.H.size() * sizeof(float), stream));
            RR_TRY(upload(f_tw, t.tw.data(), t.tw.size() * sizeof(float), stream));
            f_V = t.V;
            fast_kind = rr_chain::FK_SELECT;
        }
        return RR_OK;
    }
    std::vector<double> c(L);
    std::vector<cd> cc(L);
    for (size_t i = 0; i < L; ++i) {
        c[i] = ir_f64[L - 1 - i];
        cc[i] = cd(c[i], 0.0);
    }
    FusedFirTables t;
    build_fused_fir_tables(kind, sched.D, c, cc, t);
    if (kind == rr_chain::FK_DIRECT) {
        RR_TRY(upload(f_ctaps, t.ctaps.data(), t.ctaps.size() * sizeof(float), stream));
        f_Gp = t.Gp;
    } else {
        RR_TRY(upload(f_H, t.H.data(), t.H.size() * sizeof(float), stream));
        RR_TRY(upload(f_tw, t.tw.data(), t.tw.size() * sizeof(float), stream));
        f_V = t.V;
        f_poly = t.poly;
        f_blk = t.blk;
    }
    // NCO table of period 1: entry, wrap entry and the 8 rotations behind them (rr_freqshifter::prepare)
    float ones[2 * 10];
    for (int i = 0; i < 10; ++i) {
        ones[2 * i] = 1.f;
        ones[2 * i + 1] = 0.f;
    }
    RR_TRY(upload(f_one, ones, sizeof(ones), stream));
    fast_kind = kind;
    return RR_OK;
}

// ---------------------------------------------------------------------------
// Upsampler (resampling.rs:147-280)
// ---------------------------------------------------------------------------
int rr_upsampler::prepare(double input_rate) {
    if (have_rate && input_rate == prev_rate) return RR_OK;
    std::vector<double> ir;
    RR_TRY(upsampler_design(input_rate, output_rate, bandwidth, quality, ir));
    have_rate = true;
    prev_rate = input_rate;
    L = ir.size();
    std::vector<unsigned char> bytes;
    if (dtype == RR_F32)
        cast_to<float>(ir.data(), L, bytes);
    else
        cast_to<double>(ir.data(), L, bytes);
    RR_TRY(upload(d_ir, bytes.data(), bytes.size(), stream));
    ir_f64.swap(ir);
    sched.configure(input_rate, output_rate);  // pos = 0
    // an output gathers from at most ceil(L / U) inputs (integer ratio), or L (every input releases
    // at least one output)
    Hn = sched.integer_ratio ? (L + sched.U - 1) / sched.U : L;
    const size_t hb = Hn * elem_size(dtype);
    RR_TRY(hist[0].reserve(hb));
    RR_TRY(hist[1].reserve(hb));
    RR_HIP(hipMemsetAsync(hist[0].p, 0, hb, stream));  // ringbuf = vec![0; ir_len]: nothing before the first input
    cur = 0;
    before_hist.assign(Hn, -(int32_t(1) << 30));  // far outside every output's window
    before_hist_stale = false;
    return RR_OK;
}

int rr_upsampler::peek(double input_rate, size_t n_in, size_t *n_out) {
    if (have_rate && input_rate == prev_rate) {
        *n_out = sched.count(n_in);
        return RR_OK;
    }
    if (!(input_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be positive");
    if (!(input_rate <= output_rate))
        RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be smaller than or equal to output sample rate");
    if (!(bandwidth < input_rate)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be smaller than input sample rate");
    UpSchedule tmp;
    tmp.configure(input_rate, output_rate);
    *n_out = tmp.count(n_in);
    return RR_OK;
}

int rr_upsampler::process_dev(double input_rate, const void *d_in, size_t n_in, void *d_out, size_t cap,
                              size_t *n_out) {
    if (n_out) *n_out = 0;
    size_t produce = 0;
    RR_TRY(peek(input_rate, n_in, &produce));
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Upsampler: out_cap %zu < %zu", cap, produce);
    if (n_in > 0x3fffffffull || produce > 0x3fffffffull) RR_FAIL(RR_ERR_BAD_ARG, "Upsampler: more than 2^30 samples in one call");
    RR_TRY(select());
    RR_TRY(prepare(input_rate));
    if (n_in == 0) return RR_OK;
    // The schedule is advanced on a copy and the kept-input offsets are prepared aside: `sched`, `before_hist`, `cur`
    // change only after both launches have been accepted (a failing call leaves the block where it was).
    rr::UpSchedule next = sched;
    std::vector<int32_t> next_before_hist;
    const int32_t *d_bef = nullptr;
    // rates on a 2^-s grid (44 100 -> 48 000): the schedule in closed form on the device (RR_UPSAMPLER_GENERIC=1: the list, A/B runs and tests)
    // (integer ratios stay with k_upsample_int / the plain gather: routed here they measured slower - 10 x 0.62 against 0.43 ms per
    //  2^22 samples, Complex<f64> 4 x 0.25 against 0.20)
    const bool closed = !sched.integer_ratio && sched.closed && sched.ra < (1ull << 31) && sched.rb < (1ull << 31) &&
                        ![] { const char *e = std::getenv("RR_UPSAMPLER_GENERIC"); return e && std::atoi(e) != 0; }();
    if (closed) {
        const uint64_t p0 = sched.pos_units();
        next.advance(n_in, nullptr);
        RR_TRY(launch_upsample_closed(dtype, stream, hist[cur].p, Hn, d_in, d_ir.p, L, sched.ra, sched.rb, p0, d_out, produce));
        RR_TRY(launch_update_hist(dtype, stream, hist[cur].p, hist[cur ^ 1].p, Hn, d_in, n_in));
        sched = next;
        before_hist_stale = true;
        cur ^= 1;
        if (n_out) *n_out = produce;
        return RR_OK;
    }
    if (sched.integer_ratio) {
        next.advance(n_in, nullptr);
    } else {
        if (before_hist_stale) {
            // closed-form calls do not keep the list: kept input v = i - Hn (relative to this call) had released
            // ceil((v rb - pos) / ra) outputs (negative: before this call's first)
            const __int128 p0 = (__int128)sched.pos_units();
            for (size_t i = 0; i < Hn; ++i) {
                const __int128 q = ((__int128)i - (__int128)Hn) * (__int128)sched.rb - p0;  // < 0
                const __int128 c = -((-q) / (__int128)sched.ra);                             // ceil of a negative quotient
                before_hist[i] = (int32_t)std::max<__int128>(c, -((__int128)1 << 30));
            }
            before_hist_stale = false;
        }
        next.advance(n_in, &before);
        std::vector<int32_t> all(Hn + n_in);
        std::copy(before_hist.begin(), before_hist.end(), all.begin());
        std::copy(before.begin(), before.end(), all.begin() + Hn);
        RR_TRY(d_before.reserve(all.size() * sizeof(int32_t)));
        RR_HIP(hipMemcpyAsync(d_before.p, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        RR_HIP(hipStreamSynchronize(stream));  // `all` dies here
        d_bef = d_before.as<int32_t>();
        // the kept inputs of the next call, relative to its first output
        next_before_hist.resize(Hn);
        for (size_t i = 0; i < Hn; ++i) {
            const int64_t v = (int64_t)all[n_in + i] - (int64_t)produce;
            next_before_hist[i] = (int32_t)std::max<int64_t>(v, -(int64_t(1) << 30));
        }
    }
    RR_TRY(launch_upsample(dtype, stream, hist[cur].p, Hn, d_in, n_in, d_ir.p, L, sched.integer_ratio ? sched.U : 0, d_bef,
                           d_out, produce));
    RR_TRY(launch_update_hist(dtype, stream, hist[cur].p, hist[cur ^ 1].p, Hn, d_in, n_in));
    sched = next;
    if (!sched.integer_ratio) before_hist.swap(next_before_hist);
    cur ^= 1;
    if (n_out) *n_out = produce;
    return RR_OK;
}

// ---------------------------------------------------------------------------
// FmDemod (modulation.rs:83-158)
// ---------------------------------------------------------------------------
int rr_fmdemod::process_dev(double sample_rate, const void *d_in, size_t n_in, void *d_out, size_t cap, size_t *n_out) {
    if (n_out) *n_out = 0;
    if (n_in > cap) RR_FAIL(RR_ERR_CAPACITY, "FmDemod: out_cap %zu < %zu", cap, n_in);
    RR_TRY(select());
    if (!state_init) {
        const size_t sb = 2 * elem_size(dtype);
        RR_TRY(state[0].reserve(sb));
        RR_TRY(state[1].reserve(sb));
        RR_HIP(hipMemsetAsync(state[0].p, 0, sb, stream));  // output_sample = 0 (modulation.rs:107)
        cur = 0;
        state_init = true;
    }
    if (n_in == 0) return RR_OK;
    const double TAU = 6.283185307179586476925286766559;
    const double factor = sample_rate / deviation / TAU;  // modulation.rs:119, cast to Flt by the launcher
    RR_TRY(launch_fmdemod(dtype, stream, d_in, n_in, d_out, state[cur].p, state[cur ^ 1].p, have_prev ? 1 : 0, factor, gain));
    cur ^= 1;
    have_prev = true;
    if (n_out) *n_out = n_in;
    return RR_OK;
}

extern "C" {

// ---- FreqShifter --------------------------------------------------------------
int rr_freqshifter_create(int dtype, double precision, double shift, int device, rr_freqshifter **out) {
    RR_GUARD_BEGIN
    if (!out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    auto *h = new rr_freqshifter;
    int s = h->init_base(K_FREQSHIFTER, dtype, device);
    if (s != RR_OK) {
        delete h;
        return s;
    }
    h->precision = precision;
    h->shift = shift;
    *out = h;
    return RR_OK;
    RR_GUARD_END
}

int rr_freqshifter_set_shift(rr_freqshifter *h, double shift) {
    RR_CHECK_HANDLE(h, K_FREQSHIFTER);
    h->shift = shift;
    h->shift_changed = true;
    return RR_OK;
}
int rr_freqshifter_shift(const rr_freqshifter *h, double *shift) {
    RR_CHECK_HANDLE(h, K_FREQSHIFTER);
    *shift = h->shift;
    return RR_OK;
}
int rr_freqshifter_precision(const rr_freqshifter *h, double *precision) {
    RR_CHECK_HANDLE(h, K_FREQSHIFTER);
    *precision = h->precision;
    return RR_OK;
}

static int freqshifter_host(rr_freqshifter *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                            size_t *n_out, bool blocking) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FREQSHIFTER);
    if (n_out) *n_out = 0;
    if (n_in > cap) RR_FAIL(RR_ERR_CAPACITY, "FreqShifter: out_cap %zu < %zu", cap, n_in);
    RR_TRY(host_io(h, in, n_in, out, n_in, blocking, [&](void *di, void *dout, size_t *p) {
        return h->process_dev(rate, di, n_in, dout, n_in, p);
    }));
    if (n_out) *n_out = n_in;
    return RR_OK;
    RR_GUARD_END
}
int rr_freqshifter_process(rr_freqshifter *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                           size_t *n_out) {
    return freqshifter_host(h, rate, in, n_in, out, cap, n_out, true);
}
int rr_freqshifter_enqueue(rr_freqshifter *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                           size_t *n_out) {
    return freqshifter_host(h, rate, in, n_in, out, cap, n_out, false);
}
int rr_freqshifter_process_dev(rr_freqshifter *h, double rate, const void *d_in, size_t n_in, void *d_out,
                               size_t cap, size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FREQSHIFTER);
    return h->process_dev(rate, d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_freqshifter_destroy(rr_freqshifter *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_FREQSHIFTER);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

// ---- Filter -----------------------------------------------------------------------
int rr_filter_create(int dtype, int device, rr_filter **out) {
    RR_GUARD_BEGIN
    if (!out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    auto *h = new rr_filter;
    int s = h->init_base(K_FILTER, dtype, device);
    if (s != RR_OK) {
        delete h;
        return s;
    }
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_filter_needs_design(const rr_filter *h, double sample_rate, size_t n, int *needed) {
    RR_CHECK_HANDLE(h, K_FILTER);
    if (!needed) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *needed = h->needs_design(sample_rate, n) ? 1 : 0;
    return RR_OK;
}
int rr_filter_mark_params_changed(rr_filter *h) {
    RR_CHECK_HANDLE(h, K_FILTER);
    h->params_changed = true;
    return RR_OK;
}
int rr_filter_design(rr_filter *h, double sample_rate, size_t n, const rr_c64 *resp, const double *window_rel) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FILTER);
    return h->design(sample_rate, n, resp, window_rel);
    RR_GUARD_END
}
int rr_filter_set_gain(rr_filter *h, double gain) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FILTER);
    return h->set_gain(gain);
    RR_GUARD_END
}
int rr_filter_reset(rr_filter *h) {
    RR_CHECK_HANDLE(h, K_FILTER);
    h->hist_valid = false;
    return RR_OK;
}
static int filter_host(rr_filter *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out,
                       bool blocking) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FILTER);
    if (n_out) *n_out = 0;
    if (h->needs_design(rate, n_in))
        RR_FAIL(RR_ERR_NEED_DESIGN, "Filter: (rate %g, chunk %zu) needs a design (filters.rs:178-183)", rate, n_in);
    const size_t produce = h->peek(n_in);
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Filter: out_cap %zu < %zu", cap, produce);
    size_t got = 0;
    RR_TRY(host_io(h, in, n_in, out, produce, blocking, [&](void *di, void *dout, size_t *p) {
        int s = h->process_dev(rate, di, n_in, dout, produce, p);
        got = *p;
        return s;
    }));
    if (n_out) *n_out = got;
    return RR_OK;
    RR_GUARD_END
}
int rr_filter_process(rr_filter *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    return filter_host(h, rate, in, n_in, out, cap, n_out, true);
}
int rr_filter_enqueue(rr_filter *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    return filter_host(h, rate, in, n_in, out, cap, n_out, false);
}
int rr_filter_process_dev(rr_filter *h, double rate, const void *d_in, size_t n_in, void *d_out, size_t cap,
                          size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FILTER);
    return h->process_dev(rate, d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_filter_process_dev_f16(rr_filter *h, double sample_rate, const void *d_in, size_t n_in, void *d_out_f16,
                              size_t cap, size_t *n_out, int response_f16) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FILTER);
    return h->process_dev(sample_rate, d_in, n_in, d_out_f16, cap, n_out, true, response_f16 != 0);
    RR_GUARD_END
}
int rr_filter_last_kernel(const rr_filter *h, int *kernel) {
    RR_CHECK_HANDLE(h, K_FILTER);
    if (!kernel) RR_FAIL(RR_ERR_BAD_ARG, "null output");
    *kernel = h->last_kernel;
    return RR_OK;
}
int rr_filter_destroy(rr_filter *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_FILTER);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

// ---- Downsampler --------------------------------------------------------------------
int rr_downsampler_create(int dtype, double output_rate, double bandwidth, double quality, int device,
                          rr_downsampler **out) {
    RR_GUARD_BEGIN
    if (!out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    // resampling.rs:51-56
    if (!(output_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "output sample rate must be positive");
    if (!(bandwidth >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be positive");
    if (!(bandwidth < output_rate)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be smaller than output sample rate");
    auto *h = new rr_downsampler;
    int s = h->init_base(K_DOWNSAMPLER, dtype, device);
    if (s != RR_OK) {
        delete h;
        return s;
    }
    h->output_rate = output_rate;
    h->bandwidth = bandwidth;
    h->quality = quality;
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_downsampler_set_gain(rr_downsampler *h, double gain) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_DOWNSAMPLER);
    if (gain == h->gain) return RR_OK;
    return h->set_gain(gain);
    RR_GUARD_END
}
int rr_downsampler_peek(rr_downsampler *h, double input_rate, size_t n_in, size_t *n_out) {
    RR_CHECK_HANDLE(h, K_DOWNSAMPLER);
    if (!n_out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return h->peek(input_rate, n_in, n_out);
}
static int downsampler_host(rr_downsampler *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                            size_t *n_out, bool blocking) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_DOWNSAMPLER);
    if (n_out) *n_out = 0;
    size_t produce = 0;
    RR_TRY(h->peek(rate, n_in, &produce));
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Downsampler: out_cap %zu < %zu", cap, produce);
    size_t got = 0;
    RR_TRY(host_io(h, in, n_in, out, produce, blocking, [&](void *di, void *dout, size_t *p) {
        int s = h->process_dev(rate, di, n_in, dout, produce, p);
        got = *p;
        return s;
    }));
    if (n_out) *n_out = got;
    return RR_OK;
    RR_GUARD_END
}
int rr_downsampler_process(rr_downsampler *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                           size_t *n_out) {
    return downsampler_host(h, rate, in, n_in, out, cap, n_out, true);
}
int rr_downsampler_enqueue(rr_downsampler *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                           size_t *n_out) {
    return downsampler_host(h, rate, in, n_in, out, cap, n_out, false);
}
int rr_downsampler_process_dev(rr_downsampler *h, double rate, const void *d_in, size_t n_in, void *d_out,
                               size_t cap, size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_DOWNSAMPLER);
    return h->process_dev(rate, d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_downsampler_ir_len(const rr_downsampler *h, size_t *ir_len) {
    RR_CHECK_HANDLE(h, K_DOWNSAMPLER);
    *ir_len = h->L;
    return RR_OK;
}
int rr_downsampler_last_kernel(const rr_downsampler *h, int *kernel) {
    RR_CHECK_HANDLE(h, K_DOWNSAMPLER);
    if (!kernel) RR_FAIL(RR_ERR_BAD_ARG, "null output");
    *kernel = h->last_kernel;
    return RR_OK;
}
int rr_downsampler_destroy(rr_downsampler *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_DOWNSAMPLER);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

// ---- Upsampler ----------------------------------------------------------------------------
int rr_upsampler_create(int dtype, double output_rate, double bandwidth, double quality, int device,
                        rr_upsampler **out) {
    RR_GUARD_BEGIN
    if (!out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    // resampling.rs:185-186
    if (!(output_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "output sample rate must be positive");
    if (!(bandwidth >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be positive");
    auto *h = new rr_upsampler;
    int s = h->init_base(K_UPSAMPLER, dtype, device);
    if (s != RR_OK) {
        delete h;
        return s;
    }
    h->output_rate = output_rate;
    h->bandwidth = bandwidth;
    h->quality = quality;
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_upsampler_peek(rr_upsampler *h, double input_rate, size_t n_in, size_t *n_out) {
    RR_CHECK_HANDLE(h, K_UPSAMPLER);
    if (!n_out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return h->peek(input_rate, n_in, n_out);
}
static int upsampler_host(rr_upsampler *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                          size_t *n_out, bool blocking) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_UPSAMPLER);
    if (n_out) *n_out = 0;
    size_t produce = 0;
    RR_TRY(h->peek(rate, n_in, &produce));
    if (produce > cap) RR_FAIL(RR_ERR_CAPACITY, "Upsampler: out_cap %zu < %zu", cap, produce);
    size_t got = 0;
    RR_TRY(host_io(h, in, n_in, out, produce, blocking, [&](void *di, void *dout, size_t *p) {
        int s = h->process_dev(rate, di, n_in, dout, produce, p);
        got = *p;
        return s;
    }));
    if (n_out) *n_out = got;
    return RR_OK;
    RR_GUARD_END
}
int rr_upsampler_process(rr_upsampler *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                         size_t *n_out) {
    return upsampler_host(h, rate, in, n_in, out, cap, n_out, true);
}
int rr_upsampler_enqueue(rr_upsampler *h, double rate, const void *in, size_t n_in, void *out, size_t cap,
                         size_t *n_out) {
    return upsampler_host(h, rate, in, n_in, out, cap, n_out, false);
}
int rr_upsampler_process_dev(rr_upsampler *h, double rate, const void *d_in, size_t n_in, void *d_out, size_t cap,
                             size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_UPSAMPLER);
    return h->process_dev(rate, d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_upsampler_ir_len(const rr_upsampler *h, size_t *ir_len) {
    RR_CHECK_HANDLE(h, K_UPSAMPLER);
    *ir_len = h->L;
    return RR_OK;
}
int rr_upsampler_destroy(rr_upsampler *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_UPSAMPLER);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}
int rr_upsampler_design(double input_rate, double output_rate, double bandwidth, double quality, size_t *ir_len,
                        double *ir, size_t cap) {
    RR_GUARD_BEGIN
    if (!ir_len) RR_FAIL(RR_ERR_BAD_ARG, "null");
    std::vector<double> v;
    RR_TRY(upsampler_design(input_rate, output_rate, bandwidth, quality, v));
    *ir_len = v.size();
    if (ir) {
        if (cap < v.size()) RR_FAIL(RR_ERR_CAPACITY, "rr_upsampler_design: cap %zu < %zu", cap, v.size());
        std::memcpy(ir, v.data(), v.size() * sizeof(double));
    }
    return RR_OK;
    RR_GUARD_END
}

// ---- FmDemod ------------------------------------------------------------------------------
int rr_fmdemod_create(int dtype, double deviation, int device, rr_fmdemod **out) {
    RR_GUARD_BEGIN
    if (!out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *out = nullptr;
    auto *h = new rr_fmdemod;
    int s = h->init_base(K_FMDEMOD, dtype, device);
    if (s != RR_OK) {
        delete h;
        return s;
    }
    h->deviation = deviation;
    *out = h;
    return RR_OK;
    RR_GUARD_END
}
int rr_fmdemod_set_gain(rr_fmdemod *h, double gain) {
    RR_CHECK_HANDLE(h, K_FMDEMOD);
    h->gain = gain;
    return RR_OK;
}
int rr_fmdemod_set_deviation(rr_fmdemod *h, double deviation) {
    RR_CHECK_HANDLE(h, K_FMDEMOD);
    h->deviation = deviation;
    return RR_OK;
}
int rr_fmdemod_deviation(const rr_fmdemod *h, double *deviation) {
    RR_CHECK_HANDLE(h, K_FMDEMOD);
    *deviation = h->deviation;
    return RR_OK;
}
int rr_fmdemod_reset(rr_fmdemod *h) {
    RR_CHECK_HANDLE(h, K_FMDEMOD);
    h->have_prev = false;  // modulation.rs:145-149
    return RR_OK;
}
static int fmdemod_host(rr_fmdemod *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out,
                        bool blocking) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FMDEMOD);
    if (n_out) *n_out = 0;
    if (n_in > cap) RR_FAIL(RR_ERR_CAPACITY, "FmDemod: out_cap %zu < %zu", cap, n_in);
    RR_TRY(host_io(h, in, n_in, out, n_in, blocking, [&](void *di, void *dout, size_t *p) {
        return h->process_dev(rate, di, n_in, dout, n_in, p);
    }));
    if (n_out) *n_out = n_in;
    return RR_OK;
    RR_GUARD_END
}
int rr_fmdemod_process(rr_fmdemod *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    return fmdemod_host(h, rate, in, n_in, out, cap, n_out, true);
}
int rr_fmdemod_enqueue(rr_fmdemod *h, double rate, const void *in, size_t n_in, void *out, size_t cap, size_t *n_out) {
    return fmdemod_host(h, rate, in, n_in, out, cap, n_out, false);
}
int rr_fmdemod_process_dev(rr_fmdemod *h, double rate, const void *d_in, size_t n_in, void *d_out, size_t cap,
                           size_t *n_out) {
    RR_GUARD_BEGIN
    RR_CHECK_HANDLE(h, K_FMDEMOD);
    return h->process_dev(rate, d_in, n_in, d_out, cap, n_out);
    RR_GUARD_END
}
int rr_fmdemod_destroy(rr_fmdemod *h) {
    if (!h) return RR_OK;
    RR_CHECK_HANDLE(h, K_FMDEMOD);
    (void)hipSetDevice(h->device);
    delete h;
    return RR_OK;
}

}  // extern "C"
