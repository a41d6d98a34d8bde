// rr_api_common.hpp — what the files behind the extern "C" boundary share (rr_api.hip: common entry points and design math;
// rr_api_blocks.hip: FreqShifter, Filter, Downsampler, Upsampler, FmDemod; rr_api_fourier.hip: Fourier, Stft, Channelizer;
// rr_api_chain.hip: Chain, ChainBank, Meter): each holds a block's host logic AND its entry points.  Host code only.
#pragma once
#include "rr_blocks.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

using namespace rr;

template <class T> static inline void cast_to(const double *src, size_t n, std::vector<unsigned char> &dst) {
    dst.resize(n * sizeof(T));
    T *d = reinterpret_cast<T *>(dst.data());
    for (size_t i = 0; i < n; ++i) d[i] = static_cast<T>(src[i]);
}

static inline int upload(DevBuf &buf, const void *src, size_t bytes, hipStream_t s) {
    RR_TRY(buf.reserve(bytes ? bytes : 16));
    if (bytes) {
        RR_HIP(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, s));
        // the source is pageable host memory owned by the handle and may be
        // rewritten by the next (re)design: make the copy complete here.
        RR_HIP(hipStreamSynchronize(s));
    }
    return RR_OK;
}

static inline double gain_as_flt(int dtype, double g) { return dtype == RR_F32 ? (double)(float)g : g; }  // flt!(gain), transform.rs:55

// ---------------------------------------------------------------------------
// host-pointer entry points: H2D -> process_dev -> D2H on the handle's stream
// ---------------------------------------------------------------------------
template <class F>
static inline int host_io(rr_block *h, const void *in, size_t n_in, void *out, size_t need_out, bool blocking, F &&run) {
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

static inline void chain_use_stream(rr_chain *c, hipStream_t st) {
    c->stream = st;
    c->fs->stream = c->fl->stream = c->ds->stream = c->fo->stream = st;
}

static inline int set_sink(MeterSink &k, double double_percentile, double sample_rate, double *d_bandwidth, double *d_energy,
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

// which kernels transform a chunk of `len` points (rr_api_fourier.hip; also behind rr_fourier_route)
struct FourierRoute;

// k_ols4096_f64's tables for combined taps c (complex): G = DFT_4096(c) / 4096 in natural order, tw = e^{-j 2 pi k / 4096}, both f64
inline void ols64_tables(const std::vector<rr::cd> &c, std::vector<double> &G, std::vector<double> &tw) {
    std::vector<rr::cd> gg(4096, rr::cd(0, 0));
    for (size_t i = 0; i < c.size() && i < 4096; ++i) gg[i] = c[i];
    rr::fft_f64(gg, false);
    G.resize(2 * 4096);
    tw.resize(2 * 4096);
    for (size_t i = 0; i < 4096; ++i) {
        G[2 * i] = gg[i].real() / 4096.0;
        G[2 * i + 1] = gg[i].imag() / 4096.0;
        const double ang = -2.0 * M_PI * (double)i / 4096.0;
        tw[2 * i] = std::cos(ang);
        tw[2 * i + 1] = std::sin(ang);
    }
}
