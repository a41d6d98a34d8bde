// rr_api.hip — the extern "C" boundary (include/radiorust_amd.h) and the host
// logic of the four blocks + the chain.  No CPU fallback anywhere: without a
// HIP device every create returns RR_ERR_HIP.
#include "rr_blocks.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

using namespace rr;

// ---------------------------------------------------------------------------
// base
// ---------------------------------------------------------------------------
rr_block::~rr_block() {
    if (own_stream) {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(own_stream);
        (void)hipStreamDestroy(own_stream);
    }
}

int rr_block::select() const {
    RR_HIP(hipSetDevice(device));
    return RR_OK;
}

int rr_block::init_base(int kind_, int dtype_, int device_) {
    if (dtype_ != RR_F32 && dtype_ != RR_F64) RR_FAIL(RR_ERR_BAD_ARG, "unknown dtype %d", dtype_);
    kind = kind_;
    dtype = dtype_;
    device = device_;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        RR_FAIL(RR_ERR_HIP, "no HIP device available (%s); this backend has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_ < 0 || device_ >= count) RR_FAIL(RR_ERR_BAD_ARG, "device %d out of range (%d devices)", device_, count);
    RR_HIP(hipSetDevice(device_));
    RR_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    stream = own_stream;
    return RR_OK;
}

template <class T> static void cast_to(const double *src, size_t n, std::vector<unsigned char> &dst) {
    dst.resize(n * sizeof(T));
    T *d = reinterpret_cast<T *>(dst.data());
    for (size_t i = 0; i < n; ++i) d[i] = static_cast<T>(src[i]);
}

static int upload(DevBuf &buf, const void *src, size_t bytes, hipStream_t s) {
    RR_TRY(buf.reserve(bytes ? bytes : 16));
    if (bytes) {
        RR_HIP(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, s));
        // the source is pageable host memory owned by the handle and may be
        // rewritten by the next (re)design: make the copy complete here.
        RR_HIP(hipStreamSynchronize(s));
    }
    return RR_OK;
}

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
    host_table.resize(((size_t)de + 1 + 8) * esz);
    if (dtype == RR_F32)
        nco_table<float>(nu, de, (float)start, reinterpret_cast<float *>(host_table.data()));
    else
        nco_table<double>(nu, de, start, reinterpret_cast<double *>(host_table.data()));
    std::memcpy(host_table.data() + (size_t)de * esz, host_table.data(), esz);
    for (int k = 0; k < 8; ++k) {
        const int64_t i = (int64_t)(((__int128)128 * k * (__int128)nu) % (__int128)de);
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

static double gain_as_flt(int dtype, double g) { return dtype == RR_F32 ? (double)(float)g : g; }  // flt!(gain), transform.rs:55

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
    // long responses: overlap-save with blocks of 2^14 .. 2^18 points through the two-pass tile transform (RR_FILTER_CONV=0: the
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
            size_t N = (size_t)1 << (dtype == RR_F32 ? 14 : 13);
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
    if (use_conv) use_ols4096 = big_ols4096 = false, npart = 0;
    {
        const char *e = std::getenv("RR_FILTER_KERNEL");  // "ols4096" / "fir" keep the older kernels (A/B runs, tests)
        use_wave = filter_wave_supported(dtype, len) && !(e && (!std::strcmp(e, "ols4096") || !std::strcmp(e, "fir")));
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
    {
        // RR_DOWNSAMPLER_POLY=1: k_decim_poly also where a fused kernel applies (A/B runs)
        const char *pe = std::getenv("RR_DOWNSAMPLER_POLY");
        if (pe && std::atoi(pe) != 0 && decim_poly_supported(dtype, sched.P, sched.Q, L)) kind = rr_chain::FK_NONE;
    }
    if (kind == rr_chain::FK_NONE) {
        // every other integer ratio, and rational ratios with a short period (the tap table follows per call)
        if (decim_poly_supported(dtype, sched.P, sched.Q, L)) {
            fast_kind = rr_chain::FK_POLY;
            poly_version = ~0ull;
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
    if (sched.integer_ratio) {
        next.advance(n_in, nullptr);
    } else {
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

// ---------------------------------------------------------------------------
// Fourier
// ---------------------------------------------------------------------------
// Which kernel family transforms a chunk of `len` points:
//   fast     the radix-8/16 register kernels of rr_fused.hip (f32: 64 .. 8192) and the LDS radix-2 kernel (<= 8192 f32,
//            <= 4096 f64) for powers of two
//   big      powers of two beyond that, up to 2^24: four-step through HBM (launch_fft_big)
//   bluestein any other length >= 32 (either dtype): two power-of-two transforms of M >= 2 len - 1 points by a nested
//            rectangular-window Fourier (which is `fast` or `big` itself)
//   direct   other lengths below 32: the O(n^2) kernel
static bool is_pow2_sz(size_t n) { return n && (n & (n - 1)) == 0; }

// Which kernels transform a chunk of `len` points - ONE decision, used by prepare() and by rr_fourier_route() (host only).
struct FourierRoute {
    enum Kind { DIRECT, POW2, BIG_TILE, BIG_TRANSPOSE, BIG_GENERIC, MIXED, TILEM, BS_WAVE, BS_FUSED, BS_FUSED8K, BS_LDS, BS_LAUNCHES } kind = DIRECT;
    size_t N1 = 0, N2 = 0;  // the four-step / two-pass split
    size_t M = 0;           // Bluestein's power-of-two length
};
static FourierRoute fourier_route(int dtype, size_t len, bool force_mixed) {
    FourierRoute r;
    const bool pow2 = is_pow2_sz(len);
    const bool generic = [] { const char *e = std::getenv("RR_FOURIER_GENERIC"); return e && std::atoi(e) != 0; }();
    const int mixed_env = [] { const char *e = std::getenv("RR_FOURIER_MIXED"); return e ? std::atoi(e) : 1; }();  // 0 never, 2 wherever it applies
    if (pow2) {
        if (len < 4 || fourier_pow2_path(dtype, len)) {  // (a chunk of 1 sample is a power of two, too)
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
            if (mixed_env == 2 || force_mixed || fft_mixed_preferred(dtype, len)) {
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
        } else if (!generic && bluestein8192_supported(dtype, len)) {
            r.kind = FourierRoute::BS_FUSED8K;  // f32, 2049 .. 4096 points: one kernel around two 8192-point register transforms
            M = 8192;
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
    const FourierRoute route = fourier_route(dtype, len, force_mixed);
    using FR = FourierRoute;
    const bool use_big = route.kind == FR::BIG_TILE || route.kind == FR::BIG_TRANSPOSE || route.kind == FR::BIG_GENERIC;
    const bool generic = route.kind == FR::BIG_GENERIC;  // (only consulted on the `big` branches below)
    const bool use_mixed = route.kind == FR::MIXED, use_tilem = route.kind == FR::TILEM;
    const size_t tmN1 = route.N1, tmN2 = route.N2;
    const bool use_bs = route.kind == FR::BS_WAVE || route.kind == FR::BS_FUSED || route.kind == FR::BS_FUSED8K || route.kind == FR::BS_LDS ||
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
    bs_fused = bs_wave = bs_lds = bs_fused8k = false;
    big = use_big;
    if (use_bs) {
        const size_t M = route.M;
        bs_fused = route.kind == FR::BS_FUSED;  // k_bluestein4096
        bs_wave = route.kind == FR::BS_WAVE;    // k_bluestein1024
        bs_lds = route.kind == FR::BS_LDS;      // k_bluestein_lds
        bs_fused8k = route.kind == FR::BS_FUSED8K;  // k_bluestein8192
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
static bool chain_poly64_ok(const rr_chain *c, size_t lc) {
    static const bool off = [] { const char *e = std::getenv("RR_CHAIN_F64_FUSED"); return e && std::atoi(e) == 0; }();
    return !off && c->fl->real_taps && c->ds->sched.D >= 2 && decim_poly_supported(RR_F64, c->ds->sched.D, 1, lc);
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
    const bool can_ols = ols_decim_supported(D, lc), can_wave = ols_wave_supported(D, lc);
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
        for (size_t i = 0; i < lc; ++i) h[i] = cc[i];
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
int rr_chain::fold_mixer(FusedFirArgs &a, int64_t back) {
    const char *env = std::getenv("RR_FRAME_MIXFOLD");  // (read per call: tests switch it within one process)
    const bool off = env && std::atoi(env) == 0;
    if (!off && fs->denom >= 1 && 8 % fs->denom == 0 && frame_table_version == fs->table_version && olsG64.size() == 1024) {
        RR_TRY(ensure_mixfold());
        // the table for the phasor of the blocks' first samples: ph0 = (idx0 + e0 - V - back) mod R
        int64_t ph = ((int64_t)a.idx0 + a.e0 - a.V - back) % (int64_t)fs->denom;
        if (ph < 0) ph += fs->denom;
        a.H = d_olsHmix.as<char>() + (size_t)ph * 1024 * 2 * sizeof(float);
        a.mixfold = true;
        a.sigma = mix_sigma;
    }
    frame_table_version = fs->table_version;
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
    if (dtype == RR_F64) {
        // k_decim_poly_f64's tap list for ir = reverse(c): out[m] = sum_j ir[j] xs[e_m - (Lc - 1) + j] = sum_i c[i] xs[e_m - i]
        std::vector<double> ir(lc);
        for (size_t j = 0; j < lc; ++j) ir[j] = c[lc - 1 - j];
        std::vector<uint32_t> T;
        const int64_t e0 = 0;
        int lp = 0;
        build_decim_poly_taps(ir, ds->sched.D, 1, &e0, T, &lp, RR_F64);
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
    const int fk = pick_fused_kernel(ds->sched.D, lc, fl->real_taps, p.fft_len);
    FusedFirTables t;
    build_fused_fir_tables(fk, ds->sched.D, c, cc, t);
    use_frame = fk == FK_OLSF;
    use_ols = fk != FK_DIRECT;
    if (use_ols) {
        RR_TRY(upload(d_olsH, t.H.data(), t.H.size() * sizeof(float), stream));
        RR_TRY(upload(d_tw4096, t.tw.data(), t.tw.size() * sizeof(float), stream));
        ols_V = t.V;
        ols_poly = t.poly;
        ols_N = t.N;
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
    if (use_frame && n_in >= (frame_forced ? (size_t)1024 : (size_t)1 << 23)) {
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
        RR_TRY(fold_mixer(a, 4 * (int64_t)pending_len));  // (the frame's first block starts 4 pl samples earlier, as launch_ols_frame)
        // (the launch records its own start / end: marker packets would cost ~4 us of stream time each)
        if (timers.on && !sink.on) timers.begin_ext(ST_FUSED_FIR, &a.ev_start, &a.ev_stop);
        const rr::FrameMeter fmv = sink.frame_meter();
        RR_TRY(launch_ols_frame(stream, a, pin, pending_len, pendbuf[po].p, d_out, fo->d_window.p, fo->d_tw.p,
                                fo->center_dc, sink.on ? &fmv : nullptr));
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
        last_fused = a.mixfold ? 6 : FK_OLSF;  // (6: k_ols_frame<true>, the mixer folded into the tables)
        if (n_out) *n_out = nfr * LF;
        return RR_OK;
    }
    char *newv = nullptr;
    char *dbase = nullptr;
    if (split) {
        DevBuf &buf = dec2[dec_cur ^ 1];  // never the buffer the pending samples live in
        RR_TRY(buf.reserve((dec + 2) * esz));
        newv = buf.as<char>();
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
    // k_ols_wave + k_fft4096: the launches record their own start / end (no marker packets, which
    // cost ~4 us of stream time each); the other kernels are bracketed by recorded events
    const bool ext = timers.on && use_ols && ols_N == 1024 && split && dec > 0 && !sink.on;
    int tk = -1;
    if (ext)
        timers.begin_ext(ST_FUSED_FIR, &a.ev_start, &a.ev_stop);
    else
        tk = timers.begin(ST_FUSED_FIR, stream);
    if (use_poly64) {
        RR_TRY(launch_decim_poly(stream, a.xh, HX, d_in, n_in, d_ctaps.p, ds->sched.D, 1, poly64_Lp, Lc, a.e0, newv, dec, a.xh_out, HX,
                                 a.nco, a.denom, a.idx0, RR_F64));
        if (dec == 0)  // (no output, no tile: the history by a launch of its own)
            RR_FAIL(RR_ERR_BAD_ARG, "Chain: a fused f64 call must produce output");
    } else if (use_ols && ols_N == 1024) {
        if (ols_poly && a.D == 4) RR_TRY(fold_mixer(a, 0));
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
        } else if (dec) {
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
    last_fused = use_poly64 ? FK_POLY : use_ols ? (ols_N == 1024 ? (a.mixfold ? 7 : FK_OLSW) : FK_OLS) : FK_DIRECT;  // (7: k_ols_wave<4, true, true>)
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
    if (!(use_ols && ols_N == 1024 && ols_poly)) return RR_OK;
    const char *fke = std::getenv("RR_FUSED_KERNEL");
    const bool frame_forced = fke && !std::strcmp(fke, "olsf");
    if (use_frame && n_in >= (frame_forced ? (size_t)1024 : (size_t)1 << 23)) return RR_OK;  // (the frame kernel's calls: lane by lane)
    st.whole = n_in;
    st.dec = ds->sched.count(n_in);
    const size_t have = pending_len + st.dec;
    st.nfr = have / 4096;
    st.rest = have - st.nfr * 4096;
    st.n_head = pending_len;
    if (st.nfr == 0 || st.dec == 0) return RR_OK;  // (no frame completes: the pending chunk is appended to, lane by lane)
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
    if (a.D == 4) RR_TRY(fold_mixer(a, 0));
    *ok = true;
    return RR_OK;
}

int rr_chain::bank_pointers(const BankStep &st, const void *d_in, void *d_out, rr::BankPtrs &bp) {
    DevBuf &buf = dec2[dec_cur ^ 1];  // never the buffer the pending samples live in
    RR_TRY(buf.reserve((st.dec + 2) * elem_size(dtype)));
    bp.xh = xh[xh_cur].p;
    bp.in = d_in;
    bp.dec = buf.p;
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
    pend_ptr = newv + (st.nfr * 4096 - pending_len) * esz;
    dec_cur ^= 1;
    pending_len = st.rest;
    frame_table_version = fs->table_version;
    last_fused = st.a.mixfold ? 7 : FK_OLSW;
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

// ---------------------------------------------------------------------------
// host-pointer entry points: H2D -> process_dev -> D2H on the handle's stream
// ---------------------------------------------------------------------------
template <class F>
static int host_io(rr_block *h, const void *in, size_t n_in, void *out, size_t need_out, bool blocking, F &&run) {
    if (n_in && !in) RR_FAIL(RR_ERR_BAD_ARG, "null input");
    if (need_out && !out) RR_FAIL(RR_ERR_BAD_ARG, "null output");
    RR_TRY(h->select());
    const size_t esz = elem_size(h->dtype);
    RR_TRY(h->stage_in.reserve((n_in ? n_in : 1) * esz));
    RR_TRY(h->stage_out.reserve((need_out ? need_out : 1) * esz));
    if (n_in) RR_HIP(hipMemcpyAsync(h->stage_in.p, in, n_in * esz, hipMemcpyHostToDevice, h->stream));
    size_t produced = 0;
    RR_TRY(run(h->stage_in.p, h->stage_out.p, &produced));
    if (produced) RR_HIP(hipMemcpyAsync(out, h->stage_out.p, produced * esz, hipMemcpyDeviceToHost, h->stream));
    if (blocking) RR_HIP(hipStreamSynchronize(h->stream));
    return RR_OK;
}

#define RR_CHECK_HANDLE(h, k)                                          \
    do {                                                               \
        if (!(h) || (h)->kind != (k)) RR_FAIL(RR_ERR_BAD_ARG, "bad handle"); \
    } while (0)

#define RR_GUARD_BEGIN try {
#define RR_GUARD_END                                              \
    }                                                             \
    catch (const std::bad_alloc &) {                              \
        RR_FAIL(RR_ERR_BAD_ARG, "out of host memory");            \
    }                                                             \
    catch (...) {                                                 \
        RR_FAIL(RR_ERR_BAD_ARG, "unexpected C++ exception");      \
    }

extern "C" {

int rr_version(void) { return 100; }
const char *rr_last_error_string(void) { return rr::last_error(); }

int rr_device_count(int *count) {
    if (!count) RR_FAIL(RR_ERR_BAD_ARG, "null");
    *count = 0;
    RR_HIP(hipGetDeviceCount(count));
    return RR_OK;
}

int rr_device_pci_bus_id(int device, char *out, size_t out_cap) {
    if (!out || out_cap < 16) RR_FAIL(RR_ERR_BAD_ARG, "rr_device_pci_bus_id: needs a buffer of at least 16 bytes");
    out[0] = 0;
    int count = 0;
    RR_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) RR_FAIL(RR_ERR_BAD_ARG, "device %d out of range", device);
    RR_HIP(hipDeviceGetPCIBusId(out, (int)out_cap, device));
    return RR_OK;
}

static void chain_use_stream(rr_chain *c, hipStream_t st) {
    c->stream = st;
    c->fs->stream = c->fl->stream = c->ds->stream = c->fo->stream = st;
}
int rr_set_stream(rr_block *h, void *hip_stream) {
    if (!h) RR_FAIL(RR_ERR_BAD_ARG, "null handle");
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    if (h->kind == K_CHAIN) chain_use_stream(static_cast<rr_chain *>(h), h->stream);
    if (h->kind == K_CHAINBANK)
        for (rr_chain *c : static_cast<rr_chainbank *>(h)->lanes) chain_use_stream(c, h->stream);
    return RR_OK;
}

int rr_wait(rr_block *h) {
    if (!h) RR_FAIL(RR_ERR_BAD_ARG, "null handle");
    RR_TRY(h->select());
    RR_HIP(hipStreamSynchronize(h->stream));
    return RR_OK;
}

int rr_query(rr_block *h) {
    if (!h) RR_FAIL(RR_ERR_BAD_ARG, "null handle");
    RR_TRY(h->select());
    hipError_t e = hipStreamQuery(h->stream);
    if (e == hipSuccess) return RR_OK;
    if (e == hipErrorNotReady) return RR_ERR_NOT_READY;
    RR_FAIL(RR_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
}

int rr_host_alloc(size_t bytes, void **out) {
    if (!out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    RR_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return RR_OK;
}
int rr_host_free(void *p) {
    if (p) RR_HIP(hipHostFree(p));
    return RR_OK;
}
int rr_host_register(void *p, size_t bytes) {
    if (!p) RR_FAIL(RR_ERR_BAD_ARG, "null");
    RR_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return RR_OK;
}
int rr_host_unregister(void *p) {
    if (!p) RR_FAIL(RR_ERR_BAD_ARG, "null");
    RR_HIP(hipHostUnregister(p));
    return RR_OK;
}

// ---- design math ------------------------------------------------------------
double rr_bessel_i0(double x) { return rr::bessel_i0(x); }
double rr_kaiser_rel_with_beta(double beta, double x) { return rr::kaiser_rel_with_beta(beta, x); }
double rr_kaiser_alpha_to_beta(double alpha) { return alpha * M_PI; }
double rr_kaiser_null_at_bin_to_beta(double n) { return std::sqrt(n * n - 1.0); }
double rr_sinc(double x) { return rr::sinc(x); }
int rr_deemphasis_factor(double tau, double frequency, rr_c64 *out) {
    if (!out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    // Complex { re: 1.0, im: tau * TAU * frequency }.finv(); num-complex 0.4: finv = conj / norm / norm, norm = hypot(re, im)
    const double re = 1.0, im = tau * (2.0 * M_PI) * frequency;
    const double norm = std::hypot(re, im);
    out->re = re / norm / norm;
    out->im = -im / norm / norm;
    return RR_OK;
}
int rr_window_sample(const rr_window *w, size_t n, double *out) { return rr::window_sample(w, n, out); }

int rr_freqshifter_ratio(double sample_rate, double precision, double shift, int64_t *numer, int64_t *denom) {
    if (!numer || !denom) RR_FAIL(RR_ERR_BAD_ARG, "null");
    return rr::freq_to_ratio(sample_rate, precision, shift, numer, denom);
}

int rr_freqshifter_table(int dtype, int64_t numer, int64_t denom, double start_phase, void *table) {
    if (!table || denom <= 0) RR_FAIL(RR_ERR_BAD_ARG, "bad table arguments");
    if (dtype == RR_F32)
        nco_table<float>(numer, denom, (float)start_phase, static_cast<float *>(table));
    else if (dtype == RR_F64)
        nco_table<double>(numer, denom, start_phase, static_cast<double *>(table));
    else
        RR_FAIL(RR_ERR_BAD_ARG, "unknown dtype");
    return RR_OK;
}

int rr_filter_design_taps(size_t n, const rr_c64 *resp, const double *window_rel, rr_c64 *taps) {
    RR_GUARD_BEGIN
    std::vector<cd> g(n);
    RR_TRY(rr::filter_design_taps(n, resp, window_rel, g.data()));
    for (size_t i = 0; i < n; ++i) {
        taps[i].re = g[i].real();
        taps[i].im = g[i].imag();
    }
    return RR_OK;
    RR_GUARD_END
}

int rr_downsampler_design(double input_rate, double output_rate, double bandwidth, double quality, size_t *ir_len,
                          double *ir, size_t ir_cap) {
    RR_GUARD_BEGIN
    if (!ir_len) RR_FAIL(RR_ERR_BAD_ARG, "null");
    if (!(output_rate >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "output sample rate must be positive");
    if (!(bandwidth >= 0.0)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be positive");
    if (!(bandwidth < output_rate)) RR_FAIL(RR_ERR_CONTRACT, "bandwidth must be smaller than output sample rate");
    std::vector<double> v;
    RR_TRY(rr::downsampler_design(input_rate, output_rate, bandwidth, quality, v));
    *ir_len = v.size();
    if (ir) {
        if (ir_cap < v.size()) RR_FAIL(RR_ERR_CAPACITY, "ir_cap %zu < %zu", ir_cap, v.size());
        memcpy(ir, v.data(), v.size() * sizeof(double));
    }
    return RR_OK;
    RR_GUARD_END
}

int rr_downsampler_schedule(double input_rate, double output_rate, size_t n_in, double *pos, uint32_t *emit,
                            size_t emit_cap, size_t *count) {
    RR_GUARD_BEGIN
    if (!pos || !count) RR_FAIL(RR_ERR_BAD_ARG, "null");
    if (!(input_rate >= output_rate) || !(output_rate >= 0.0))
        RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be greater than or equal to output sample rate");
    Schedule sc;
    sc.configure(input_rate, output_rate);
    // resume from *pos: replay the integer phase, or set the f64 accumulator
    if (sc.integer_ratio) {
        const uint64_t k = (uint64_t)(*pos / output_rate);
        if (k >= sc.D) RR_FAIL(RR_ERR_BAD_ARG, "pos out of range");
        sc.phase = sc.D - 1 - k;
    }
    sc.pos = *pos;
    std::vector<uint32_t> e;
    const size_t c = sc.advance(n_in, emit ? &e : nullptr);
    if (emit) {
        if (c > emit_cap) RR_FAIL(RR_ERR_CAPACITY, "emit_cap %zu < %zu", emit_cap, c);
        memcpy(emit, e.data(), c * sizeof(uint32_t));
    }
    *count = c;
    *pos = sc.pos;
    return RR_OK;
    RR_GUARD_END
}

int rr_fourier_design_window(size_t n, const double *window_rel, double *values) {
    return rr::fourier_design_window(n, window_rel, values);
}

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
                dtype == RR_F32 ? 8192u : 4096u);
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
static int set_sink(MeterSink &k, double double_percentile, double sample_rate, double *d_bandwidth, double *d_energy,
                    size_t cap_frames, int store_spectra) {
    k = MeterSink{};
    if (!d_bandwidth) return RR_OK;  // off
    if (!(double_percentile == double_percentile)) RR_FAIL(RR_ERR_BAD_ARG, "metering: double_percentile is NaN");
    k.on = true;
    k.dp = double_percentile;
    k.rate = sample_rate;
    k.bw = d_bandwidth;
    k.energy = d_energy;
    k.cap = cap_frames;
    k.store = store_spectra ? 1 : 0;
    return RR_OK;
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
        case FR::BS_LDS: std::snprintf(buf, cap, "bluestein one kernel M=%zu", r.M); break;
        case FR::BS_LAUNCHES: {
            // around the nested power-of-two transform: one launch each (M <= 8192 / 4096: five in all), its two passes with the
            // element-wise stages folded in (four), or its five launches (seventeen)
            const FourierRoute nested = fourier_route(dtype, r.M, false);
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

int rr_synth_iq_dev(int device, void *hip_stream, uint64_t seed, uint64_t t0, size_t n, void *d_out) {
    if (n && !d_out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    RR_HIP(hipSetDevice(device));
    return launch_synth(static_cast<hipStream_t>(hip_stream), seed, t0, n, d_out);
}

}  // extern "C"
