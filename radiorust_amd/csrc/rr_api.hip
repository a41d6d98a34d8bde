// rr_api.hip — the entry points no block owns (include/radiorust_amd.h): version, devices, streams, waiting, pinned host memory,
// the design math of the blocks (windows, Filter taps, Downsampler / Upsampler responses, phase tables), the synthetic source;
// and rr_block's base.  No CPU fallback anywhere: without a HIP device every create returns RR_ERR_HIP.
#include "rr_api_common.hpp"

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

int rr_upsampler_schedule(double input_rate, double output_rate, size_t n_in, double *pos, int32_t *before,
                          size_t before_cap, size_t *count) {
    RR_GUARD_BEGIN
    if (!pos || !count) RR_FAIL(RR_ERR_BAD_ARG, "null");
    if (!(input_rate > 0.0) || !(input_rate <= output_rate))
        RR_FAIL(RR_ERR_CONTRACT, "input sample rate must be smaller than or equal to output sample rate");
    UpSchedule sc;
    sc.configure(input_rate, output_rate);
    sc.pos = *pos;
    std::vector<int32_t> b;
    const size_t c = sc.advance(n_in, before ? &b : nullptr);
    if (before) {
        if (n_in > before_cap) RR_FAIL(RR_ERR_CAPACITY, "before_cap %zu < %zu", before_cap, n_in);
        memcpy(before, b.data(), n_in * sizeof(int32_t));
    }
    *count = c;
    *pos = sc.pos;
    return RR_OK;
    RR_GUARD_END
}

int rr_fourier_design_window(size_t n, const double *window_rel, double *values) {
    return rr::fourier_design_window(n, window_rel, values);
}

int rr_synth_iq_dev(int device, void *hip_stream, uint64_t seed, uint64_t t0, size_t n, void *d_out) {
    if (n && !d_out) RR_FAIL(RR_ERR_BAD_ARG, "null");
    RR_HIP(hipSetDevice(device));
    return launch_synth(static_cast<hipStream_t>(hip_stream), seed, t0, n, d_out);
}

}  // extern "C"
