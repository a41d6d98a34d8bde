"""GPU parity of the on-device chain FreqShifter -> Filter -> Downsampler -> Fourier
(BASELINE configs[1]) against the oracle's block-by-block composition, for the
block-by-block path (allow_fused=False) and the fused path (allow_fused=True).

Tolerance: relative RMS <= 1e-5 per output stream against the f64 oracle
(north_star), and within a small multiple of the f32 oracle's own error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def lowpass(cut):
    return lambda _b, f: 1.0 if abs(f) <= cut else 0.0


def rms_rel(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(b) ** 2)))


@pytest.fixture(scope="module")
def rr():
    import torch

    assert torch.cuda.is_available()
    import radiorust_amd

    return radiorust_amd


CFG2 = dict(shift=25e6, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6, fft_len=4096)


def make(rr, oracle, params, allow_fused, dtype=np.float32, null_bin=2.0):
    g = rr.Chain(**params, fft_window=rr.Kaiser.with_null_at_bin(null_bin), allow_fused=allow_fused, dtype=dtype)
    return g


def oracle_spectra(oracle, x, fs, params, flt, null_bin=2.0, precision=1.0):
    return oracle.run_chain(x, fs, flt=flt, fft_window=oracle.Kaiser.with_null_at_bin(null_bin), precision=precision, **params)[3]


@pytest.mark.parametrize("allow_fused", [False, True])
@pytest.mark.parametrize("pieces", ["one", "ragged", "chunks64"])
def test_chain_cfg2(rr, oracle, allow_fused, pieces):
    fs, n = 200e6, 1 << 18
    x = oracle.synth_iq(1, 0, n)
    t64 = oracle_spectra(oracle, x, fs, CFG2, np.float64)
    t32 = oracle_spectra(oracle, x, fs, CFG2, np.float32)
    assert len(t64) == 15
    g = make(rr, oracle, CFG2, allow_fused)
    if pieces == "one":
        cuts = [0, n]
    elif pieces == "ragged":
        cuts = [0, 1, 63, 64, 65, 5000, 5000, 70001, 200000, n]
    else:
        cuts = list(range(0, 64 * 400 + 1, 64)) + [n]
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        want = g.peek(fs, b - a)
        got = g.process(rr.Samples(fs, x[a:b]))
        assert len(got) == want
        out += got
    assert len(out) == 15
    assert all(s.sample_rate == 50e6 and len(s.chunk) == 4096 for s in out)
    for k in range(15):
        e = rms_rel(out[k].chunk, t64[k])
        eref = rms_rel(t32[k], t64[k])
        assert e <= 1e-5 and e <= max(4 * eref, 5e-7), (k, e, eref)
    if allow_fused and pieces == "one":
        # a long steady-state call must actually take the fused kernels
        g.process(rr.Samples(fs, x))
        assert g.last_path_fused()


@pytest.mark.parametrize("allow_fused", [False, True])
def test_chain_interrupt_and_retune(rr, oracle, allow_fused):
    """An interrupt event drops the Rechunker's partial chunk and the Filter's
    history (chunks.rs:80-88, filters.rs:262-265); Downsampler/Fourier carry on."""
    fs = 200e6
    x = oracle.synth_iq(2, 0, 300000)
    g = make(rr, oracle, CFG2, allow_fused)
    out = g.process(rr.Samples(fs, x[:100000]))  # 100000 = 1562 chunks + 32 leftover
    ev = rr.EventSignal(rr.SamplesLost())
    mid = g.process(ev)  # the Rechunker holds 32 samples: it reports them lost in front of the event (chunks.rs:80-88)
    assert len(mid) == 2 and mid[0].is_event() and mid[0].event.is_interrupt() and mid[1] is ev
    assert g.pending() == 0 and g.process(ev) == [ev]  # nothing pending: the event alone
    g.set_shift(-12.5e6)
    out += g.process(rr.Samples(fs, x[100000:]))
    # oracle: same message sequence through the four blocks + a Rechunker(64)
    sh = oracle.FreqShifter(1.0, 25e6, flt=np.float64)
    fl = oracle.Filter(lowpass(20e6), flt=np.float64)
    ds = oracle.Downsampler(4096, 50e6, 40e6, flt=np.float64)
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), flt=np.float64)
    ref = []

    def feed(mixed):
        for off in range(0, len(mixed) - 63, 64):
            z = fl.process(fs, mixed[off : off + 64])
            if z is not None:
                for c in ds.feed(fs, z):
                    ref.append(fo.process(c))

    m1 = sh.process(fs, x[:100000])
    feed(m1[: 100000 // 64 * 64])  # the 32 leftover samples are lost with the event
    fl.interrupt()
    sh.set_shift(-12.5e6)
    feed(sh.process(fs, x[100000:]))
    assert len(out) == len(ref) and len(ref) >= 17
    for a, b in zip(out, ref):
        assert rms_rel(a.chunk, b) <= 1e-5


@pytest.mark.parametrize("allow_fused", [False, True])
def test_chain_sample_rate_change(rr, oracle, allow_fused):
    """A new sample rate while the Rechunker in front of the Filter holds samples: they are dropped behind a
    SamplesLost (chunks.rs:72-79) - they were mixed with the old rate's NCO table -, the FreqShifter rebuilds its
    table keeping the phase (transform.rs:318-340), Filter and Downsampler redesign (filters.rs:178-187,
    resampling.rs:75-102); the partly filled output chunk of the Downsampler carries on."""
    fs1, fs2 = 200e6, 100e6
    x = oracle.synth_iq(5, 0, 260000)
    g = make(rr, oracle, CFG2, allow_fused)
    out = g.process(rr.Samples(fs1, x[:100000]))  # 1562 chunks of 64 + 32 leftover
    assert g.pending() == 32
    out2 = g.process(rr.Samples(fs2, x[100000:]))
    assert out2[0].is_event() and isinstance(out2[0].event, rr.SamplesLost) and not out2[1].is_event()
    out += out2[1:]
    assert g.pending() == (260000 - 100000) % 64
    sh = oracle.FreqShifter(1.0, 25e6, flt=np.float64)
    fl = oracle.Filter(lowpass(20e6), flt=np.float64)
    ds = oracle.Downsampler(4096, 50e6, 40e6, flt=np.float64)
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), flt=np.float64)
    ref = []

    def feed(fs, mixed):
        for off in range(0, len(mixed) - 63, 64):
            z = fl.process(fs, mixed[off : off + 64])
            if z is not None:
                for c in ds.feed(fs, z):
                    ref.append(fo.process(c))

    m1 = sh.process(fs1, x[:100000])
    feed(fs1, m1[: 100000 // 64 * 64])  # the 32 leftover samples are dropped by the Rechunker
    fl.interrupt()                      # its SamplesLost is an interrupt for the Filter
    feed(fs2, sh.process(fs2, x[100000:]))
    assert len(out) == len(ref) and len(ref) >= 20
    for a, b in zip(out, ref):
        assert rms_rel(a.chunk, b) <= 1e-5
    # a rate change with nothing pending loses nothing
    g2 = make(rr, oracle, CFG2, allow_fused)
    g2.process(rr.Samples(fs1, x[:64000]))
    assert g2.pending() == 0 and not any(s.is_event() for s in g2.process(rr.Samples(fs2, x[64000:128000])))


@pytest.mark.parametrize("allow_fused", [False, True])
def test_chain_other_parameters(rr, oracle, allow_fused):
    """Parameters of the reference's bandwidth_meter example (main.rs:43-69
    without the Overlapper): non-integer-free 10:1 decimation, L = 145, a
    480-entry NCO table — and a 2.5:1 rational ratio that only the generic
    schedule can serve."""
    cases = [
        (1024000.0, dict(shift=200e3, filter_len=128, freq_resp=lowpass(30e3), output_rate=102400.0, bandwidth=60e3, fft_len=1024), 4.0),
        (48000.0, dict(shift=700.0, filter_len=64, freq_resp=lowpass(8e3), output_rate=19200.0, bandwidth=12e3, fft_len=256), 2.0),
        (200e6, dict(shift=12.345e6, precision=1e3, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6, fft_len=512), 2.0),
        # 8 : 1 (L = 288) and 2 : 1 (L = 144): k_ols_wave<8> / k_ols_wave<2> on the fused path
        (384000.0, dict(shift=48e3, filter_len=64, freq_resp=lowpass(20e3), output_rate=48000.0, bandwidth=40000.0, fft_len=1024), 2.0),
        (96000.0, dict(shift=12e3, filter_len=64, freq_resp=lowpass(20e3), output_rate=48000.0, bandwidth=44000.0, fft_len=2048), 2.0),
    ]
    for fs, params, nb in cases:
        n = 150000
        x = oracle.synth_iq(3, 0, n)
        p2 = dict(params)
        prec = p2.pop("precision", 1.0)
        t64 = oracle_spectra(oracle, x, fs, p2, np.float64, nb, precision=prec)
        g = make(rr, oracle, params, allow_fused, null_bin=nb)
        out = g.process(rr.Samples(fs, x[:40000])) + g.process(rr.Samples(fs, x[40000:]))
        assert len(out) == len(t64) and len(out) > 3
        for a, b in zip(out, t64):
            assert rms_rel(a.chunk, b) <= 1e-5


def test_chain_f64(rr, oracle):
    fs, n = 200e6, 1 << 16
    x = oracle.synth_iq(4, 0, n).astype(np.complex128)
    t64 = oracle_spectra(oracle, x, fs, CFG2, np.float64)
    g = make(rr, oracle, CFG2, True, dtype=np.float64)
    out = g.process(rr.Samples(fs, x))
    assert len(out) == len(t64) == 3
    for a, b in zip(out, t64):
        assert rms_rel(a.chunk, b) <= 1e-12


@pytest.mark.parametrize("kernel", ["k_ols4096_f64", "k_decim_poly"])
@pytest.mark.parametrize("shift,precision", [(25e6, 1.0), (12.345e6, 1e3)])
def test_chain_f64_fused_front_end(rr, oracle, monkeypatch, shift, precision, kernel):
    """Complex<f64>: mixer + Filter + Downsampler as ONE pass once the chain is in its steady state - overlap-save in blocks of
    4096 points (k_ols4096_f64, the default) or the polyphase decimator (RR_CHAIN_F64_FUSED=poly), the phase table riding along,
    combined taps - then the register-resident 4096-point transform; ragged calls, an interrupt and a retune, against the f64
    oracle at 1e-11."""
    if kernel == "k_decim_poly":
        monkeypatch.setenv("RR_CHAIN_F64_FUSED", "poly")
    fs, n = 200e6, 1 << 18
    params = dict(CFG2, shift=shift)
    x = oracle.synth_iq(14, 0, n).astype(np.complex128)
    g = rr.Chain(**params, precision=precision, fft_window=rr.Kaiser.with_null_at_bin(2.0), dtype=np.float64)
    ref = oracle.run_chain(x, fs, flt=np.float64, fft_window=oracle.Kaiser.with_null_at_bin(2.0), precision=precision, **params)[3]
    out, fused_calls = [], 0
    cuts = [0, 20000, 20001, 70007, 70071, 150000, 200064, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        out += g.process(rr.Samples(fs, x[a:b]))
        fused_calls += g.last_path_kernel() == kernel
    assert fused_calls >= 4, fused_calls
    assert len(out) == len(ref) == 15
    for a, b in zip(out, ref):
        assert rms_rel(a.chunk, b) <= 1e-11
    # retune + interrupt: against a fresh oracle run is not possible mid-stream; the block-by-block path is the reference
    g2 = rr.Chain(**params, precision=precision, fft_window=rr.Kaiser.with_null_at_bin(2.0), dtype=np.float64, allow_fused=False)
    g3 = rr.Chain(**params, precision=precision, fft_window=rr.Kaiser.with_null_at_bin(2.0), dtype=np.float64)
    y = oracle.synth_iq(15, 0, n).astype(np.complex128)
    outs = {id(g2): [], id(g3): []}
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        for h in (g2, g3):
            if i == 3:
                h.set_shift(-31e6)
            if i == 5:
                h.interrupt()
            outs[id(h)] += h.process(rr.Samples(fs, y[a:b]))
    assert len(outs[id(g2)]) == len(outs[id(g3)]) >= 10
    for a, b in zip(outs[id(g3)], outs[id(g2)]):
        assert rms_rel(a.chunk, b.chunk) <= 1e-11


def test_chain_device_api_and_capacity(rr, oracle):
    import ctypes as C

    import torch

    from radiorust_amd._lib import RR_ERR_CAPACITY

    fs, n = 200e6, 1 << 20
    d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    rr.synth_iq_dev(0, stream, 5, 0, n, d_in.data_ptr())
    g = make(rr, oracle, CFG2, True)
    g.set_stream(stream)
    frames = g.peek(fs, n)
    assert frames == 63
    d_out = torch.empty(frames * 4096, dtype=torch.complex64, device="cuda")
    # too small: refused, nothing consumed
    cnt = C.c_size_t()
    s = rr._lib.lib().rr_chain_process_dev(g._h, fs, d_in.data_ptr(), n, d_out.data_ptr(), 4096, C.byref(cnt))
    assert s == RR_ERR_CAPACITY and cnt.value == 0
    assert g.peek(fs, n) == 63
    assert g.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel()) == frames * 4096
    torch.cuda.synchronize()
    x = oracle.synth_iq(5, 0, n)
    t32 = oracle_spectra(oracle, x, fs, CFG2, np.float32)
    got = d_out.cpu().numpy().reshape(frames, 4096)
    for k in (0, 1, 31, 62):
        assert rms_rel(got[k], t32[k]) <= 1e-5
    # size-independent property at full size: Parseval per frame (window has
    # mean square 1, so sum|X|^2 = n * sum|w v|^2) — checked via linearity:
    # chain(2x) == 2 chain(x)
    g2 = make(rr, oracle, CFG2, True)
    g2.set_stream(stream)
    d_out2 = torch.empty_like(d_out)
    d_in2 = d_in * 2
    g2.process_dev(fs, d_in2.data_ptr(), n, d_out2.data_ptr(), d_out2.numel())
    torch.cuda.synchronize()
    assert torch.allclose(d_out2, d_out * 2, rtol=1e-6, atol=0)


@pytest.mark.parametrize("allow_fused", [False, True])
def test_chain_complex_response(rr, oracle, allow_fused):
    """A one-sided (not even) frequency response gives genuinely complex taps: the fused
    path then runs the overlap-save kernel (k_ols_decim4)."""
    fs, n = 200e6, 1 << 18
    resp = lambda b, f: (1.0 + 0.25j) if 0 <= f <= 20e6 else 0.0  # noqa: E731
    params = dict(shift=25e6, filter_len=64, freq_resp=resp, output_rate=50e6, bandwidth=40e6, fft_len=4096)
    x = oracle.synth_iq(6, 0, n)
    t64 = oracle_spectra(oracle, x, fs, params, np.float64)
    g = make(rr, oracle, params, allow_fused)
    out = g.process(rr.Samples(fs, x[:100000])) + g.process(rr.Samples(fs, x[100000:]))
    assert len(out) == len(t64) == 15
    for a, b in zip(out, t64):
        assert rms_rel(a.chunk, b) <= 1e-5
    if allow_fused:
        assert g.last_path_fused()


@pytest.mark.parametrize("kernel", ["ols", "olsw", "olsf", "direct"])
def test_chain_overlap_save_kernel_forced(rr, oracle, monkeypatch, kernel):
    """RR_FUSED_KERNEL selects the fused FIR implementation: overlap-save with a workgroup per
    4096-block (ols), with a wave per 1024-block (olsw), or the direct form (direct)."""
    monkeypatch.setenv("RR_FUSED_KERNEL", kernel)
    fs, n = 200e6, 1 << 18
    x = oracle.synth_iq(1, 0, n)
    t64 = oracle_spectra(oracle, x, fs, CFG2, np.float64)
    g = make(rr, oracle, CFG2, True)
    out = []
    for a, b in ((0, 70001), (70001, 200000), (200000, n)):
        out += g.process(rr.Samples(fs, x[a:b]))
    assert g.last_path_fused() and len(out) == 15
    assert g.last_path_kernel() == {"ols": "k_ols_decim4", "olsw": "k_ols_wave", "olsf": "k_ols_frame", "direct": "k_mix_fir_decim"}[kernel]
    for a, b in zip(out, t64):
        assert rms_rel(a.chunk, b) <= 1e-5


@pytest.mark.parametrize("bw,filter_len,shift,precision", [(30e6, 64, 25e6, 1.0), (30e6, 6, 12.345e6, 1e3), (20e6, 20, -50e6, 1.0),
                                                           (20e6, 90, 12.5e6, 1.0), (30e6, 70, 3.7e6, 1e5), (40e6, 10, 25e6, 1.0)])
def test_chain_frame_kernel_other_overlaps(rr, oracle, monkeypatch, bw, filter_len, shift, precision):
    """k_ols_frame for combined responses shorter than cfg2's: Downsamplers of L = 60 (bw 30 MHz) and 40 (20 MHz) with Filters
    of 6 .. 90 taps give overlaps of 64 / 128 samples - 240 / 224 results per block, 18 / 19 blocks per frame, the waves of
    the last round partly idle - beside the 192 of cfg2's shape; general and foldable NCO periods, ragged calls."""
    monkeypatch.setenv("RR_FUSED_KERNEL", "olsf")
    fs, n = 200e6, 1 << 18
    params = dict(shift=shift, filter_len=filter_len, freq_resp=lowpass(18e6), output_rate=50e6, bandwidth=bw, fft_len=4096)
    x = oracle.synth_iq(77, 0, n)
    t64 = oracle.run_chain(x, fs, flt=np.float64, fft_window=oracle.Kaiser.with_null_at_bin(2.0), precision=precision, **params)[3]
    g = rr.Chain(**params, precision=precision, fft_window=rr.Kaiser.with_null_at_bin(2.0))
    out, frame_calls = [], 0
    for a, b in ((0, 50001), (50001, 120000), (120000, 120000 + 16384 * 5), (120000 + 16384 * 5, n)):
        out += g.process(rr.Samples(fs, x[a:b]))
        frame_calls += g.last_path_kernel() == "k_ols_frame"
    assert frame_calls >= 3 and len(out) == len(t64) == 15
    for a, b in zip(out, t64):
        assert rms_rel(a.chunk, b) <= 1e-5


@pytest.mark.parametrize("seed", range(6))
def test_chain_wave_kernel_randomised(rr, oracle, seed):
    """k_ols_wave (the default fused kernel for 4x decimation) under random parameters: shifts whose
    NCO period does and does not divide 128 (one phasor pair per lane vs the general table walk),
    filter lengths and transition widths that move the combined tap count Lc between 112 and 513,
    random ragged call sizes (edge blocks, carry, odd block alignment), real and complex responses."""
    rng = np.random.default_rng(100 + seed)
    fs = 200e6
    filter_len = int(rng.choice([64, 128, 256]))
    bw = float(rng.choice([30e6, 40e6, 44e6]))  # L = ceil(200e6 / ((50e6 - bw) / 2) * 3): 60, 120, 200
    shift, precision = [(25e6, 1.0), (12.5e6, 1.0), (12.345e6, 1e3), (-3.7e6, 1e5), (1.5625e6, 1.0), (31e6, 1e6)][seed]
    cut = float(rng.choice([10e6, 20e6]))
    resp = lowpass(cut) if seed % 2 == 0 else (lambda _b, f, c=cut: 1.0 if 0 <= f <= c else 0.0)
    params = dict(shift=shift, filter_len=filter_len, freq_resp=resp, output_rate=50e6, bandwidth=bw, fft_len=4096)
    n = 1 << 17
    x = oracle.synth_iq(30 + seed, 0, n)
    t64 = oracle.run_chain(x, fs, flt=np.float64, fft_window=oracle.Kaiser.with_null_at_bin(2.0), precision=precision, **params)[3]
    g = rr.Chain(**params, precision=precision, fft_window=rr.Kaiser.with_null_at_bin(2.0))
    cuts = sorted({0, n, *(int(v) for v in rng.integers(1, n, size=6))})
    out = []
    fused_calls = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        out += g.process(rr.Samples(fs, x[a:b]))
        fused_calls += g.last_path_kernel() in ("k_ols_wave", "k_ols_frame", "k_ols_decim4")  # 4096-blocks beyond Lc = 385
    assert len(out) == len(t64) and len(out) >= 7
    assert fused_calls >= 2, (cuts, fused_calls)
    for a, b in zip(out, t64):
        assert rms_rel(a.chunk, b) <= 1e-5


def test_chain_enqueue_with_pinned_buffers(rr, oracle):
    """rr_chain_enqueue: the asynchronous host-pointer entry; n_out is final at enqueue time,
    the spectra are there after rr_wait."""
    import ctypes as C

    L = rr._lib.lib()
    fs, n = 200e6, 1 << 17
    x = oracle.synth_iq(1, 0, n)
    t64 = oracle_spectra(oracle, x, fs, CFG2, np.float64)
    g = make(rr, oracle, CFG2, True)
    g._ensure_design(fs)
    p_in, p_out = C.c_void_p(), C.c_void_p()
    cap = 8 * 4096
    assert L.rr_host_alloc(n * 8, C.byref(p_in)) == 0 and L.rr_host_alloc(cap * 8, C.byref(p_out)) == 0
    C.memmove(p_in, x.ctypes.data, n * 8)
    cnt = C.c_size_t()
    assert L.rr_chain_enqueue(g._h, fs, p_in, n, p_out, cap, C.byref(cnt)) == 0
    assert cnt.value == len(t64) * 4096 == 7 * 4096
    g.wait()
    assert g.query()
    y = np.frombuffer((C.c_char * (cnt.value * 8)).from_address(p_out.value), dtype=np.complex64).reshape(-1, 4096)
    for a, b in zip(y, t64):
        assert rms_rel(a, b) <= 1e-5
    assert L.rr_host_free(p_in) == 0 and L.rr_host_free(p_out) == 0


def test_chain_enqueue_back_to_back_matches_blocking_calls(rr, oracle):
    """Several rr_chain_enqueue calls in flight (their copies and kernels overlap on three streams, two sets
    of staging buffers, growing call sizes): after one rr_wait every call's spectra equal those of the same
    stream pushed through the blocking entry, bit for bit."""
    import ctypes as C

    L = rr._lib.lib()
    fs = 200e6
    sizes = [1 << 15, 3000, 1 << 16, 0, 70001, 1 << 17, 1 << 15, 123457]
    x = oracle.synth_iq(5, 0, sum(sizes))
    ref = make(rr, oracle, CFG2, True)
    want, off = [], 0
    for n in sizes:
        got = ref.process(rr.Samples(fs, x[off : off + n]))
        want.append(np.concatenate([np.asarray(s.chunk) for s in got]) if got else np.empty(0, np.complex64))
        off += n
    g = make(rr, oracle, CFG2, True)
    g._ensure_design(fs)
    bufs, off = [], 0
    for n in sizes:
        p_in, p_out = C.c_void_p(), C.c_void_p()
        cap = (n // 4 // 4096 + 2) * 4096
        assert L.rr_host_alloc(max(n, 1) * 8, C.byref(p_in)) == 0 and L.rr_host_alloc(cap * 8, C.byref(p_out)) == 0
        C.memmove(p_in, x[off : off + n].ctypes.data, n * 8)
        off += n
        bufs.append((p_in, p_out, cap, C.c_size_t()))
    for n, (p_in, p_out, cap, cnt) in zip(sizes, bufs):
        assert L.rr_chain_enqueue(g._h, fs, p_in, n, p_out, cap, C.byref(cnt)) == 0
    g.wait()
    assert g.query()
    for w, (p_in, p_out, cap, cnt) in zip(want, bufs):
        assert cnt.value == w.size
        if cnt.value:
            y = np.frombuffer((C.c_char * (cnt.value * 8)).from_address(p_out.value), dtype=np.complex64)
            assert np.array_equal(y.view(np.uint32), np.ascontiguousarray(w).view(np.uint32))
        assert L.rr_host_free(p_in) == 0 and L.rr_host_free(p_out) == 0


def test_handles_run_concurrently_from_different_threads(rr, oracle):
    """SURVEY 8(b) threading: a handle belongs to one task at a time, different handles run concurrently from
    different OS threads (tokio work stealing).  Four chains with their own streams and seeds, each driven by
    its own thread through ragged calls; every result equals the same chain run alone."""
    import threading

    fs, n = 200e6, 1 << 17
    cuts = [0, 1000, 70001, 70064, n]
    xs = [oracle.synth_iq(40 + i, 0, n) for i in range(4)]

    def run(x):
        g = make(rr, oracle, CFG2, True)
        out = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            out += [np.array(s.chunk) for s in g.process(rr.Samples(fs, x[a:b]))]
        return out

    alone = [run(x) for x in xs]
    got = [None] * 4
    errs = []

    def worker(i):
        try:
            for _ in range(3):
                got[i] = run(xs[i])
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for a, g_ in zip(alone, got):
        assert len(a) == len(g_) == 7
        for u, v in zip(a, g_):
            assert np.array_equal(u.view(np.uint32), v.view(np.uint32))


def test_chain_one_call_of_2_pow_28_samples(rr, oracle):
    """Index arithmetic at a size the other tests do not reach (BASELINE's timing batch is 2^26 per step): ONE chain call
    over 2^28 samples (2 GiB in, 16383 spectra out).  The first 256 spectra against the C oracle; every spectrum against
    the same stream fed in 16 calls of 2^24 samples (a size-independent property: the cut of a stream into calls must not
    show), compared on the device."""
    import torch

    n, fs = 1 << 28, 200e6
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 3, 0, n, d_in.data_ptr())
    torch.cuda.synchronize()
    g = make(rr, oracle, CFG2, True)
    g.set_stream(st)
    frames = g.peek(fs, n)
    assert frames == 16383
    d_out = torch.empty(frames * 4096, dtype=torch.complex64, device="cuda")
    assert g.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel()) == frames * 4096
    torch.cuda.synchronize()  # (torch's default stream is the null stream: the handle then runs on its own)
    K = 256
    x = d_in[: (K + 2) * 16384].cpu().numpy()
    ref = oracle.run_chain_c(x, fs, shift=25e6, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6,
                             fft_len=4096, fft_window=oracle.Kaiser.with_null_at_bin(2.0), flt=np.float64, threads=4,
                             max_frames=K + 1)[0][:K]
    y = d_out[: K * 4096].cpu().numpy().reshape(K, 4096)
    for a, b in zip(y, ref):
        assert rms_rel(a, b) <= 1e-5
    g2 = make(rr, oracle, CFG2, True)
    g2.set_stream(st)
    d_out2 = torch.empty_like(d_out)
    off = wrote = 0
    step = 1 << 24
    while off < n:
        wrote += g2.process_dev(fs, d_in.data_ptr() + 8 * off, step, d_out2.data_ptr() + 8 * wrote, d_out2.numel() - wrote)
        off += step
    assert wrote == frames * 4096
    torch.cuda.synchronize()
    a, b = d_out.view(frames, 4096), d_out2.view(frames, 4096)
    err = (torch.linalg.vector_norm(a - b, dim=1) / torch.linalg.vector_norm(b, dim=1)).max().item()
    assert err <= 2e-6, err


def test_chain_frame_kernel_and_two_kernel_calls_in_one_stream(rr, oracle):
    """Calls of 2^23 samples and more run the fused frame kernel (k_ols_frame: FIR and Fourier stage in one kernel, the
    unfinished frame in a pending buffer), shorter ones k_ols_wave + k_fft4096 with the same tables; the pending samples,
    the mixed-sample history and the NCO phase have to pass from one to the other - also across a retune and an
    interrupt.  Every call's spectra against the same stream fed through a chain that never takes the frame kernel
    (compared on the device), the first ones of every section against the C oracle."""
    import torch

    fs = 200e6
    # (the first call of a stream, and the first after an interrupt, runs block by block: the fused kernels need the
    #  Filter's previous chunk and a filled Downsampler window)
    sizes = [50000, (1 << 23) + 12345, 70000, 1 << 23, 1000, 30000, (1 << 23) + 64, 300000]
    n = sum(sizes)
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 5, 0, n, d_in.data_ptr())
    torch.cuda.synchronize()

    def run(env_kernel):
        if env_kernel:
            os.environ["RR_FUSED_KERNEL"] = env_kernel
        else:
            os.environ.pop("RR_FUSED_KERNEL", None)
        g = make(rr, oracle, CFG2, True)
        g.set_stream(st)
        outs, kernels, off = [], [], 0
        for i, m in enumerate(sizes):
            if i == 3:
                g.set_shift(-12.5e6)  # retune between calls (transform.rs:318-340)
            if i == 5:
                g.interrupt()         # the Filter's history and the carry go (filters.rs:262-265)
            cap = (m // 4 // 4096 + 2) * 4096
            d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
            w = g.process_dev(fs, d_in.data_ptr() + 8 * off, m, d_out.data_ptr(), cap)
            torch.cuda.synchronize()
            outs.append(d_out[:w].clone())
            kernels.append(g.last_path_kernel())
            off += m
        return outs, kernels

    import os
    try:
        got, kernels = run(None)
        ref, kref = run("olsw")
    finally:
        os.environ.pop("RR_FUSED_KERNEL", None)
    assert kernels == ["", "k_ols_frame", "k_ols_wave", "k_ols_frame", "k_ols_wave", "", "k_ols_frame", "k_ols_wave"], kernels
    assert "k_ols_frame" not in kref
    for a, b in zip(got, ref):
        assert a.numel() == b.numel()
        if a.numel():
            fa, fb = a.view(-1, 4096), b.view(-1, 4096)
            err = (torch.linalg.vector_norm(fa - fb, dim=1) / torch.linalg.vector_norm(fb, dim=1)).max().item()
            assert err <= 2e-6, err
    # and the first 32 spectra of the stream against the C oracle
    K = 32
    x = d_in[: (K + 2) * 16384].cpu().numpy()
    want = oracle.run_chain_c(x, fs, shift=25e6, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6,
                              fft_len=4096, fft_window=oracle.Kaiser.with_null_at_bin(2.0), flt=np.float64, threads=4,
                              max_frames=K + 1)[0][:K]
    y = torch.cat([got[0], got[1]])[: K * 4096].cpu().numpy().reshape(K, 4096)
    for a, b in zip(y, want):
        assert rms_rel(a, b) <= 1e-5


@pytest.mark.parametrize("kernel", ["olsf", "olsw"])
def test_chain_frame_kernel_with_the_mixer_folded_into_its_tables(rr, oracle, monkeypatch, kernel):
    """NCO periods that divide 8 (shifts by multiples of fs / 8): k_ols_frame<true> (and k_ols_wave<4, true, true>, which the
    chain runs on calls below 2^23 samples) transforms the samples unmixed, the mixer
    sits in the response tables (rr_chain::ensure_mixfold) and in one product per result.  The frame kernel forced onto short
    ragged calls, retunes between shifts with periods 8, 8, 4, 2, 1 and 16 (the last: the GP instances, the mixer behind the
    filter), an
    interrupt; every call against the same stream with the fold switched off, and the whole against the C oracle section by
    section (a retune keeps the phase continuous: transform.rs:322-325)."""
    import torch

    fs = 200e6
    shifts = [25e6, -25e6, 50e6, 75e6, 100e6, 0.0, 12.5e6, 25e6]
    sizes = [40000, 70000 + 4 * 13, 65536, 50000 + 4 * 5, 123456, 30000, 65536 + 4 * 7, 80000, 16384 * 3, 100000, 70000, 90000, 65536, 81920, 60000, 77777 // 4 * 4]
    n = sum(sizes)
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 9, 0, n, d_in.data_ptr())
    torch.cuda.synchronize()
    monkeypatch.setenv("RR_FUSED_KERNEL", kernel)

    def run(fold):
        monkeypatch.setenv("RR_FRAME_MIXFOLD", "1" if fold else "0")
        g = make(rr, oracle, CFG2, True)
        g.set_stream(st)
        outs, off = [], 0
        for i, m in enumerate(sizes):
            if i and i % 2 == 0:
                g.set_shift(shifts[(i // 2) % len(shifts)])
            if i == 11:
                g.interrupt()
            cap = (m // 4 // 4096 + 2) * 4096
            d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
            torch.cuda.synchronize()
            w = g.process_dev(fs, d_in.data_ptr() + 8 * off, m, d_out.data_ptr(), cap)
            torch.cuda.synchronize()
            outs.append(d_out[:w].clone())
            folded.append(g.last_path_mixer_folded())
            assert g.last_path_kernel() in ("", "k_ols_frame" if kernel == "olsf" else "k_ols_wave")
            off += m
        return outs

    folded = []
    got = run(True)
    # folded: every frame call under a period that divides 8, except the first one after a retune (the history in front of it
    # was written under the old table) and the calls that restart block by block (the first of the stream, after the interrupt)
    assert sum(folded) >= 5 and not folded[0] and not folded[12] and folded[13], folded
    folded = []
    ref = run(False)
    assert not any(folded)
    total = 0
    for a, b in zip(got, ref):
        assert a.numel() == b.numel()
        total += a.numel()
        if a.numel():
            fa, fb = a.view(-1, 4096), b.view(-1, 4096)
            err = (torch.linalg.vector_norm(fa - fb, dim=1) / torch.linalg.vector_norm(fb, dim=1)).max().item()
            assert err <= 3e-6, err
    assert total >= 40 * 4096
    # the first section (shift fs / 8, several calls) against the C oracle
    K = 6
    x = d_in[: (K + 2) * 16384].cpu().numpy()
    want = oracle.run_chain_c(x, fs, shift=25e6, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6,
                              fft_len=4096, fft_window=oracle.Kaiser.with_null_at_bin(2.0), flt=np.float64, threads=4,
                              max_frames=K + 1)[0][:K]
    y = torch.cat(got[:2])[: K * 4096].cpu().numpy().reshape(K, 4096)
    for a, b in zip(y, want):
        assert rms_rel(a, b) <= 1e-5


@pytest.mark.parametrize("shift,precision,bw,filter_len", [(25e6, 1.0, 40e6, 64), (-50e6, 1.0, 30e6, 64), (12.345e6, 1e3, 40e6, 64),
                                                           (7e6, 1e3, 20e6, 20), (0.0, 1.0, 30e6, 30)])
def test_chain_frame_kernel_with_1024_point_spectra(rr, oracle, monkeypatch, shift, precision, bw, filter_len):
    """k_ols_frame<.., 1024>: 1024-point spectra, ONE WAVE per frame (five blocks, then the wave-level transform) - the mixer folded
    into the tables (periods 8 and 4), behind the filter (40000, 200) and absent; the three overlaps; ragged calls with the
    frame kernel forced onto short calls, an interrupt, calls without a whole frame; centred and plain; against the f64 oracle."""
    monkeypatch.setenv("RR_FUSED_KERNEL", "olsf")
    fs, n = 200e6, 1 << 18
    params = dict(shift=shift, filter_len=filter_len, freq_resp=lowpass(10e6), output_rate=50e6, bandwidth=bw, fft_len=1024)
    x = oracle.synth_iq(41, 0, n)
    for center in (False, True):
        ref = oracle.run_chain(x, fs, flt=np.float64, fft_window=oracle.Kaiser.with_null_at_bin(2.0), precision=precision,
                               center_dc=center, **params)[3]
        g = rr.Chain(**params, precision=precision, fft_window=rr.Kaiser.with_null_at_bin(2.0), center_dc=center)
        out, frame_calls = [], 0
        cuts = [0, 30000, 30001, 31000, 90007, 90071, 150000, 200064, 200100, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            out += g.process(rr.Samples(fs, x[a:b]))
            frame_calls += g.last_path_kernel() == "k_ols_frame"
        assert frame_calls >= 4, frame_calls
        assert len(out) == len(ref) and len(out) >= 60
        for a, b in zip(out, ref):
            assert len(a.chunk) == 1024 and rms_rel(a.chunk, b) <= 1e-5


@pytest.mark.parametrize("kernel,out_rate,bw", [("olsf", 50e6, 40e6), ("olsw", 50e6, 40e6), ("olsw", 25e6, 20e6), ("olsw", 100e6, 80e6),
                                                ("olsw", 12.5e6, 10e6)])  # (16 : 1: k_ols_wg)
def test_chain_frame_kernel_with_the_mixer_behind_the_filter(rr, oracle, monkeypatch, kernel, out_rate, bw):
    """Every NCO period that does not divide 8 (and every period at 2 : 1 and 8 : 1): k_ols_frame<.., GP> and
    k_ols_wave<D, .., GP> transform the samples as they are with the tables of the
    response c[i] w^-i (rr_chain::ensure_genfold) and multiply their results by the phase table's entries at their positions.
    Periods 40000 (the bench's general_nco leg: 12.345 MHz at precision 1e3), 16, 200 (below the 256 results of a block: the
    index reduction by %), 2000 and 40000 again with another numerator; ragged calls, retunes, an interrupt; every call against
    the same stream with the mixer in front of the transform (RR_FRAME_GENFOLD=0), the first section against the C oracle."""
    import torch

    fs = 200e6
    shifts = [12.345e6, 12.5e6, -7e6, 3.3e6, -61.005e6, 12.345e6]
    sizes = [40000, 70000 + 4 * 13, 65536, 50000 + 4 * 5, 123456, 30000, 65536 + 4 * 7, 80000, 16384 * 3, 100000, 70000, 90000,
             65536, 81920, 60000, 77777 // 4 * 4, 40000, 66000]
    n = sum(sizes)
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 11, 0, n, d_in.data_ptr())
    torch.cuda.synchronize()
    monkeypatch.setenv("RR_FUSED_KERNEL", kernel)
    params = dict(CFG2, shift=shifts[0], precision=1e3, output_rate=out_rate, bandwidth=bw)
    D = int(fs / out_rate)

    def run(fold):
        monkeypatch.setenv("RR_FRAME_GENFOLD", "1" if fold else "0")
        g = make(rr, oracle, params, True)
        g.set_stream(st)
        outs, off = [], 0
        for i, m in enumerate(sizes):
            if i and i % 3 == 0:
                g.set_shift(shifts[(i // 3) % len(shifts)])
            if i == 10:
                g.interrupt()
            cap = (m // D // 4096 + 2) * 4096
            d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
            torch.cuda.synchronize()
            w = g.process_dev(fs, d_in.data_ptr() + 8 * off, m, d_out.data_ptr(), cap)
            torch.cuda.synchronize()
            outs.append(d_out[:w].clone())
            folded.append(g.last_path_mixer_folded())
            assert g.last_path_kernel() in ("", "k_ols_frame" if kernel == "olsf" else "k_ols_wave")
            off += m
        return outs

    folded = []
    got = run(True)
    # (not the first call of the stream or after the interrupt - block by block -, nor the first frame call behind those or behind a
    # retune: the history in front of it was not written under the table in use)
    assert sum(folded) >= 9 and not folded[0] and not folded[1] and folded[2] and not folded[3] and folded[4] and folded[5], folded
    folded = []
    ref = run(False)
    assert not any(folded)
    total = 0
    for a, b in zip(got, ref):
        assert a.numel() == b.numel()
        total += a.numel()
        if a.numel():
            fa, fb = a.view(-1, 4096), b.view(-1, 4096)
            err = (torch.linalg.vector_norm(fa - fb, dim=1) / torch.linalg.vector_norm(fb, dim=1)).max().item()
            assert err <= 3e-6, err
    assert total >= 240 // D * 4096
    K = 20 // D
    x = d_in[: (K + 2) * 4096 * D].cpu().numpy()
    want = oracle.run_chain_c(x, fs, shift=shifts[0], precision=1e3, filter_len=64, freq_resp=lowpass(20e6), output_rate=out_rate,
                              bandwidth=bw, fft_len=4096, fft_window=oracle.Kaiser.with_null_at_bin(2.0), flt=np.float64,
                              threads=4, max_frames=K + 1)[0][:K]
    y = torch.cat(got[:3])[: K * 4096].cpu().numpy().reshape(K, 4096)
    for a, b in zip(y, want):
        assert rms_rel(a, b) <= 1e-5


@pytest.mark.parametrize("dtype,tol", [(np.float32, 1e-5), (np.float64, 1e-11)])
def test_meter_is_the_bandwidth_meter_example(rr, oracle, dtype, tol):
    """rr_meter_*: the reference's own pipeline in its own order (examples/bandwidth_meter/main.rs:53-69) -
    FreqShifter -> Downsampler(1024, 102400, max_bw) (10 : 1, L = 145) -> Filter(|f| <= max_bw / 2) on the Downsampler's
    1024-sample chunks -> Overlapper(4) -> Fourier(Kaiser null at bin 4) -> metering::bandwidth(0.01, ..) - on the device
    without host hops, fed in ragged pieces, against the same blocks of the oracle message by message; an event in the
    middle resets Filter and Overlapper but not the Downsampler's partly filled chunk."""
    from radiorust_amd import metering

    fs, out_rate, max_bw, quality = 1024000.0, 102400.0, 60e3, 4
    resp = lambda _b, f: 1.0 if abs(f) <= max_bw / 2 else 0.0  # noqa: E731
    x = oracle.synth_iq(41, 0, 300000)
    if dtype == np.float64:
        x = x.astype(np.complex128)
    g = rr.Meter(shift=12.5e3, output_rate=out_rate, bandwidth=max_bw, chunk_len=1024, freq_resp=resp, overlap=quality,
                 fft_window=rr.Kaiser.with_null_at_bin(float(quality)), dtype=dtype)
    sh = oracle.FreqShifter(1.0, 12.5e3, flt=np.float64)
    ds = oracle.Downsampler(1024, out_rate, max_bw, flt=np.float64)
    fl = oracle.Filter(resp, flt=np.float64)
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(float(quality)), flt=np.float64)
    hist, ref, got = [], [], []

    def feed(piece):
        for c in ds.feed(fs, sh.process(fs, piece)):
            z = fl.process(out_rate, c)
            if z is not None:
                hist.append(z)
                if len(hist) >= quality:
                    ref.append(fo.process(np.concatenate(hist[-quality:])))
                    del hist[: len(hist) - (quality - 1)]

    cuts = [0, 5000, 5001, 60000, 131072, 200000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        out = g.process(rr.Samples(fs, x[a:b]))
        assert all(s.sample_rate == out_rate and len(s.chunk) == 1024 * quality for s in out)
        # f32 calls of >= 4096 samples: mixer and decimator as one kernel; the 1-sample call and f64: two steps
        assert g.front_fused() == (dtype == np.float32 and b - a >= 4096)
        got += out
        feed(x[a:b])
    assert len(got) == len(ref) and len(ref) >= 12
    # an interrupting event: SamplesLost first, Filter and Overlapper start over, the Downsampler's chunk carries on
    ev = rr.EventSignal(rr.Disconnection())
    mid = g.process(ev)
    assert len(mid) == 2 and isinstance(mid[0].event, rr.SamplesLost) and mid[1] is ev
    fl.interrupt()
    hist.clear()
    got += g.process(rr.Samples(fs, x[200000:]))
    feed(x[200000:])
    assert len(got) == len(ref) and len(ref) >= 17
    for a, b in zip(got, ref):
        assert rms_rel(a.chunk, b) <= tol
        bw_ref = oracle.bandwidth(0.01, out_rate, b, flt=np.float64)
        bw_got = metering.bandwidth(0.01, out_rate, a.chunk, dtype=dtype)
        assert abs(bw_got - bw_ref) <= 1e-3 * bw_ref + 1e-6


def _meter_vs_oracle(rr, oracle, *, fs, out_rate, max_bw, chunk_len, quality, script, seed=43, tol=1e-5):
    """Drives rr.Meter and the oracle's FreqShifter -> Downsampler -> Filter -> Overlapper -> Fourier side by side through
    `script`, a list of ("feed", n) / ("shift", hz) / ("rate", hz) steps; returns the number of spectra compared."""
    resp = lambda _b, f: 1.0 if abs(f) <= max_bw / 2 else 0.0  # noqa: E731
    total = sum(s[1] for s in script if s[0] == "feed")
    x = oracle.synth_iq(seed, 0, total)
    shift0 = next(s[1] for s in script if s[0] == "shift")
    g = rr.Meter(shift=shift0, output_rate=out_rate, bandwidth=max_bw, chunk_len=chunk_len, freq_resp=resp, overlap=quality,
                 fft_window=rr.Kaiser.with_null_at_bin(float(quality)))
    sh = oracle.FreqShifter(1.0, shift0, flt=np.float64)
    ds = oracle.Downsampler(chunk_len, out_rate, max_bw, flt=np.float64)
    fl = oracle.Filter(resp, flt=np.float64)
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(float(quality)), flt=np.float64)
    hist, ref, got = [], [], []
    rate, off, first = fs, 0, True
    for op, v in script:
        if op == "shift":
            if not first:
                g.set_shift(v)
                sh.set_shift(v)
            first = False
            continue
        if op == "rate":
            rate = v
            continue
        piece = x[off : off + v]
        off += v
        out = g.process(rr.Samples(rate, piece))
        if v >= 4096:
            assert g.front_fused(), (op, v)  # mixer and decimator as ONE kernel (k_decim_poly with the table riding along)
        got += out
        for c in ds.feed(rate, sh.process(rate, piece)):
            z = fl.process(out_rate, c)
            if z is not None:
                hist.append(z)
                if len(hist) >= quality:
                    ref.append(fo.process(np.concatenate(hist[-quality:])))
                    del hist[: len(hist) - (quality - 1)]
        assert len(got) == len(ref), (op, v, len(got), len(ref))
    worst = 0.0
    for a, b in zip(got, ref):
        worst = max(worst, rms_rel(a.chunk, b))
    assert worst <= tol, worst
    return len(ref)


@pytest.mark.parametrize("case", ["10to1", "8to3", "tiny_denom", "denom_not_dividing_256"])
def test_meter_retunes_between_ragged_calls(rr, oracle, case):
    """The fused front end (FreqShifter riding on k_decim_poly) after `set_shift`: the table is then rebuilt WITH the phase
    of the current phasor (transform.rs:322-337), so no entry of it is a pure rotation - the kernel's 256-sample steps
    must come from the rotations kept behind the table.  Ratios 10 : 1 (the example's) and 8 : 3 (simple_receiver.rs:28,
    the mixer riding along with Q > 1), a two-entry table, and a period that does not divide 256; every call is >= 4096
    samples and ragged, so the 8-load interior batches, the 4 / 2 / 1 tails and the edge tiles all meet a non-zero
    start phase."""
    if case == "8to3":
        cfg = dict(fs=1024000.0, out_rate=384000.0, max_bw=200e3, chunk_len=256, quality=2)
        shifts = [33e3, -120.5e3, 7e3, 512e3]
    elif case == "tiny_denom":
        cfg = dict(fs=1024000.0, out_rate=102400.0, max_bw=60e3, chunk_len=1024, quality=4)
        shifts = [512e3, 256e3, -256e3, 341e3]  # periods 2, 4, 4 and 1024
    elif case == "denom_not_dividing_256":
        cfg = dict(fs=1024000.0, out_rate=102400.0, max_bw=60e3, chunk_len=1024, quality=4)
        shifts = [204800.0, 3 * 40960.0, 7 * 10240.0, -3 * 8192.0]  # 1/5, 3/25, 7/100, -3/125 of fs = 2^13 5^3
    else:
        cfg = dict(fs=1024000.0, out_rate=102400.0, max_bw=60e3, chunk_len=1024, quality=4)
        shifts = [12.5e3, -40.123e3, 3.0, 99.999e3]
    sizes = [70001, 4096, 65536 + 17, 40000, 123457, 50003, 81920, 60001]
    script = []
    for i, n in enumerate(sizes):
        if i % 2 == 0:
            script.append(("shift", shifts[(i // 2) % len(shifts)]))
        script.append(("feed", n))
    assert _meter_vs_oracle(rr, oracle, script=script, **cfg) >= 20


def test_meter_sample_rate_change_mid_stream(rr, oracle):
    """A new input rate recalculates the phase table from the current phasor's phase (transform.rs:318-340) and redesigns the
    Downsampler (resampling.rs:75-102; 10 : 1 -> 20 : 1 -> 5 : 1); the partly filled output chunk carries on."""
    script = [("shift", 12.5e3), ("feed", 100001), ("rate", 2048000.0), ("feed", 150003), ("feed", 8192), ("shift", -30e3),
              ("feed", 90001), ("rate", 512000.0), ("feed", 70001), ("feed", 65536)]
    n = _meter_vs_oracle(rr, oracle, fs=1024000.0, out_rate=102400.0, max_bw=60e3, chunk_len=1024, quality=4, script=script)
    assert n >= 20


@pytest.mark.parametrize("shift,precision,out_rate,bw", [(25e6, 1.0, 50e6, 40e6), (12.345e6, 1e3, 25e6, 20e6)])
def test_chainbank_small_chunks_stay_in_lockstep(rr, oracle, shift, precision, out_rate, bw):
    """Chunks so small that most calls complete no frame (1024 .. 8192 samples per channel at 4 : 1 / 8 : 1): the bank's step is then
    ONE launch that appends every channel's decimated samples to its pending chunk (plus one launch that moves the pending
    chunks into the chains' own buffers behind a frame-completing step) - bit for bit the stand-alone chains, in lockstep."""
    import torch

    fs, K = 200e6, 6
    sizes = [65536, 65536] + [4096] * 9 + [1024] * 13 + [8192] * 5 + [2048, 4096, 1024, 512, 512, 16384, 4096, 4096]
    n_total = sum(sizes)
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(K * n_total, dtype=torch.complex64, device="cuda")
    for k in range(K):
        rr.synth_iq_dev(0, st, 300 + k, 0, n_total, d_in.data_ptr() + 8 * k * n_total)
    torch.cuda.synchronize()
    params = dict(shift=shift, precision=precision, filter_len=64, freq_resp=lowpass(20e6), output_rate=out_rate, bandwidth=bw,
                  fft_len=4096, fft_window=rr.Kaiser.with_null_at_bin(2.0))
    bank = rr.ChainBank(K, **params)
    bank.set_stream(st)
    solo = [rr.Chain(**params) for _ in range(K)]
    for c in solo:
        c.set_stream(st)
    cap = 1 << 16
    out_b = torch.zeros(K * cap, dtype=torch.complex64, device="cuda")
    out_s = torch.zeros(K * cap, dtype=torch.complex64, device="cuda")
    pos, lock, frames = 0, [], 0
    for i, n in enumerate(sizes):
        wb = bank.process_dev(fs, d_in.data_ptr() + 8 * pos, n_total, n, out_b.data_ptr(), cap, cap)
        lock.append(bank.last_path_lockstep())
        for k, c in enumerate(solo):
            ws = c.process_dev(fs, d_in.data_ptr() + 8 * (k * n_total + pos), n, out_s.data_ptr() + 8 * k * cap, cap)
            assert ws == wb, (i, k, ws, wb)
        torch.cuda.synchronize()
        for k in range(K):
            assert torch.equal(out_b[k * cap : k * cap + wb], out_s[k * cap : k * cap + wb]), (i, k)
        frames += wb // 4096
        pos += n
    assert frames >= 6
    # lane by lane only while the histories fill; every small call afterwards in lockstep, with or without a frame
    assert all(lock[2:]), lock


@pytest.mark.parametrize("shift,precision,out_rate,bw", [(25e6, 1.0, 50e6, 40e6), (12.345e6, 1e3, 50e6, 40e6), (12.345e6, 1e3, 25e6, 20e6)])
def test_chainbank_lockstep_is_bit_identical_to_stand_alone_chains(rr, oracle, shift, precision, out_rate, bw):
    """rr_chainbank: K channels through TWO launches per call once they are in the steady state - every channel's spectra
    bit for bit those of a stand-alone Chain fed the same samples in the same calls (stream start lane by lane, an interrupt
    and a retune in the middle, a ragged call that takes the bank out of lockstep and back).  8 : 1: k_ols_wave2k_bank."""
    import torch

    fs, K = 200e6, 5
    n_total = 1 << 19
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(K * n_total, dtype=torch.complex64, device="cuda")
    for k in range(K):
        rr.synth_iq_dev(0, st, 100 + k, 0, n_total, d_in.data_ptr() + 8 * k * n_total)
    torch.cuda.synchronize()
    params = dict(shift=shift, precision=precision, filter_len=64, freq_resp=lowpass(20e6), output_rate=out_rate, bandwidth=bw,
                  fft_len=4096, fft_window=rr.Kaiser.with_null_at_bin(2.0))
    bank = rr.ChainBank(K, **params)
    bank.set_stream(st)
    solo = [rr.Chain(**params) for _ in range(K)]
    for c in solo:
        c.set_stream(st)
    cap = 1 << 17
    out_b = torch.zeros(K * cap, dtype=torch.complex64, device="cuda")
    out_s = torch.zeros(K * cap, dtype=torch.complex64, device="cuda")
    sizes = [65536, 65536, 32768, 65536, 16384 + 64, 1000, 65536 - 1064, 65536, 65536, 49152]
    pos, lock = 0, []
    for i, n in enumerate(sizes):
        if i == 4:
            bank.set_shift(-12.5e6)
            for c in solo:
                c.set_shift(-12.5e6)
        if i == 7:
            bank.interrupt()
            for c in solo:
                c.interrupt()
        wb = bank.process_dev(fs, d_in.data_ptr() + 8 * pos, n_total, n, out_b.data_ptr(), cap, cap)
        lock.append(bank.last_path_lockstep())
        for k, c in enumerate(solo):
            ws = c.process_dev(fs, d_in.data_ptr() + 8 * (k * n_total + pos), n, out_s.data_ptr() + 8 * k * cap, cap)
            assert ws == wb, (i, k, ws, wb)
        torch.cuda.synchronize()
        for k in range(K):
            assert torch.equal(out_b[k * cap : k * cap + wb], out_s[k * cap : k * cap + wb]), (i, k)
        pos += n
    # lockstep once the histories have filled, lane by lane at the start, after the retune's table change and for the ragged calls
    assert lock[0] is False and lock[2] is True and lock[3] is True and lock[5] is False and lock[-1] is True, lock
    assert sum(lock) >= 4, lock
