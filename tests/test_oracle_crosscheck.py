"""Cross-checks of the C restatement (oracle/rr_oracle.c) against the
independent numpy f64 formulation (oracle/oracle_np.py).

Filter / FreqShifter / Downsampler have no tests in the reference (parity
unpinned); these checks only guard the restatement against
transcription mistakes.  Parameter sets are the BASELINE configs and the block
parameters of the reference's examples (simple_receiver.rs:25-62,
bandwidth_meter/main.rs:51-72).
"""
import math
import os

import numpy as np
import pytest

from oracle import oracle_np as onp


def lowpass(cut):
    return lambda _bin, f: 1.0 if abs(f) <= cut else 0.0


def rms_rel(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return math.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(b) ** 2))


# ---- math ----------------------------------------------------------------
def test_bessel_vs_scipy(oracle):
    for x in [0.0, 1e-3, 0.5, 1.7320508, 2.8284271, 7.5, 15.8, 40.0, 300.0]:
        assert oracle.bessel_I0(x) == pytest.approx(float(onp.bessel_I0(x)), rel=1e-13)


def test_sinc_vs_numpy(oracle):
    for x in np.linspace(-7.3, 7.3, 41):
        assert oracle.sinc(x) == pytest.approx(float(np.sinc(x)), abs=1e-15)


def test_kaiser_quirks(oracle):
    # no factor pi in with_null_at_bin (math.rs:37-39)
    assert oracle.kaiser_null_at_bin_to_beta(2.0) == math.sqrt(3.0)
    assert oracle.kaiser_alpha_to_beta(1.5) == 1.5 * math.pi
    w = oracle.Kaiser.with_null_at_bin(2.0)
    assert w.relative_value_at(0.0) == pytest.approx(float(onp.bessel_I0(math.sqrt(3.0))), rel=1e-14)
    assert w.relative_value_at(1.0) == 1.0
    assert oracle.Rectangular().relative_value_at(0.3) == 1.0
    assert oracle.CustomWindow(lambda x: 1.0 - x * x).relative_value_at(0.5) == 0.75


# ---- FFT convention --------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 12, 64, 96, 128, 1000, 4096])
def test_fft_matches_numpy(oracle, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    X = oracle.fft(x, flt=np.float64)
    assert rms_rel(X, np.fft.fft(x)) < 1e-13
    Xi = oracle.fft(x, inverse=True, flt=np.float64)
    assert rms_rel(Xi, np.fft.ifft(x) * n) < 1e-13  # unnormalised inverse
    X32 = oracle.fft(x, flt=np.float32)
    assert rms_rel(X32, np.fft.fft(x.astype(np.complex64))) < 2e-6


# ---- FreqShifter -----------------------------------------------------------
@pytest.mark.parametrize(
    "fs,prec,shift,want",
    [
        (48000.0, 1.0, 700.0, (7, 480)),  # BASELINE cfg1
        (200e6, 1.0, 25e6, (1, 8)),  # cfg2 primary
        (200e6, 1e3, 12.345e6, (2469, 40000)),  # cfg2 stress
        (1024000.0, 1.0, 200e3, (25, 128)),  # bandwidth_meter example
        (48000.0, 1.0, -700.0, (-7, 480)),
        (48000.0, 1.0, 0.0, (0, 1)),
    ],
)
def test_ratio(oracle, fs, prec, shift, want):
    assert oracle.freq_to_ratio(fs, prec, shift) == want
    assert onp.freq_ratio(fs, prec, shift) == want


@pytest.mark.parametrize("fs,prec,shift", [(48000.0, 1.0, 700.0), (200e6, 1.0, 25e6), (48000.0, 1.0, -1234.0)])
def test_freqshifter_vs_numpy(oracle, fs, prec, shift):
    x = oracle.synth_iq(1, 0, 5000)
    numer, denom = onp.freq_ratio(fs, prec, shift)
    want = onp.freqshift(x, numer, denom)
    # chunk-boundary invariance: feed in ragged pieces
    for flt, tol in ((np.float64, 1e-14), (np.float32, 3e-7)):
        sh = oracle.FreqShifter(prec, shift, flt=flt)
        got = np.concatenate([sh.process(fs, x[a:b]) for a, b in ((0, 1), (1, 1000), (1000, 1000), (1000, 4097), (4097, 5000))])
        assert rms_rel(got, want) < tol
        assert sh.table().size == denom


def test_freqshifter_retune_keeps_phase(oracle):
    fs = 48000.0
    sh = oracle.FreqShifter(1.0, 700.0, flt=np.float64)
    x = np.ones(1000, dtype=np.complex128)
    y1 = sh.process(fs, x[:333])
    sh.set_shift(-2500.0)
    y2 = sh.process(fs, x[:100])
    # phase of the first sample after the retune = phase the old table would
    # have had at that position (transform.rs:322-325)
    numer, denom = onp.freq_ratio(fs, 1.0, 700.0)
    want0 = onp.freqshift(np.ones(334), numer, denom)[333]
    assert abs(y2[0] - want0) < 1e-12
    n2, d2 = onp.freq_ratio(fs, 1.0, -2500.0)
    want = want0 * onp.freqshift(np.ones(100), n2, d2)
    assert rms_rel(y2, want) < 1e-12
    assert y1.size == 333


# ---- Filter ----------------------------------------------------------------
FILTER_CASES = [
    (64, 200e6, 20e6),  # cfg2
    (1024, 2e9, 200e6),  # cfg5
    (4096, 48000.0, 16e3),  # cfg1
    (48, 48000.0, 4e3),  # non power of two
    (33, 48000.0, 5e3),  # odd: swap leaves the last tap alone
]


@pytest.mark.parametrize("n,fs,cut", FILTER_CASES)
def test_filter_is_causal_fir(oracle, n, fs, cut):
    """The overlap-save filter equals a causal n-tap FIR with g = 2n*h, and the
    first chunk after a reset produces no output (filters.rs:240,260)."""
    wv = np.array([oracle.Kaiser.with_null_at_bin(2.0).relative_value_at(p) for p in onp.window_positions(n)])
    g = onp.filter_taps(n, fs, lowpass(cut), wv)
    nchunks = 5
    x = oracle.synth_iq(2, 0, n * nchunks)
    for flt, tol in ((np.float64, 1e-12), (np.float32, 2e-6)):
        f = oracle.Filter(lowpass(cut), flt=flt)
        outs = [f.process(fs, x[i * n : (i + 1) * n]) for i in range(nchunks)]
        assert outs[0] is None
        got = np.concatenate(outs[1:])
        want = onp.fir_causal(x, g, n)
        assert rms_rel(got, want) < tol
        h = f.response()
        assert rms_rel(2 * n * h, g) < 1e-12
    # taps of a real-even response are real to rounding
    assert np.max(np.abs(g.imag)) < 1e-15 * np.max(np.abs(g.real)) * n


def test_filter_interrupt_and_update(oracle):
    n, fs = 64, 200e6
    x = oracle.synth_iq(3, 0, n * 6)
    f = oracle.Filter(lowpass(20e6), flt=np.float64)
    assert f.process(fs, x[:n]) is None
    assert f.process(fs, x[n : 2 * n]) is not None
    f.interrupt()
    assert f.process(fs, x[2 * n : 3 * n]) is None  # history dropped
    y = f.process(fs, x[3 * n : 4 * n])
    wv = np.array([oracle.Kaiser.with_null_at_bin(2.0).relative_value_at(p) for p in onp.window_positions(n)])
    g = onp.filter_taps(n, fs, lowpass(20e6), wv)
    want = onp.fir_causal(x[2 * n : 4 * n], g, n)
    assert rms_rel(y, want) < 1e-12
    f.update(lowpass(5e6))
    assert f.process(fs, x[4 * n : 5 * n]) is None  # redesign drops history
    # a rate change redesigns too
    assert f.process(fs, x[5 * n : 6 * n]) is not None
    assert f.process(fs / 2, x[:n]) is None
    # and so does a chunk-length change
    assert f.process(fs / 2, x[: n // 2]) is None


def test_filter_complex_response(oracle):
    """Single-sideband style response (not even): taps are genuinely complex."""
    n, fs = 64, 48000.0
    resp = lambda b, f: (1.0 + 0.5j) if 0 <= f <= 6000 else 0.0  # noqa: E731
    wv = np.array([oracle.Rectangular().relative_value_at(p) for p in onp.window_positions(n)])
    g = onp.filter_taps(n, fs, resp, wv)
    x = oracle.synth_iq(4, 0, n * 4)
    f = oracle.Filter(resp, oracle.Rectangular(), flt=np.float64)
    outs = [f.process(fs, x[i * n : (i + 1) * n]) for i in range(4)]
    assert rms_rel(np.concatenate(outs[1:]), onp.fir_causal(x, g, n)) < 1e-12
    assert np.max(np.abs(g.imag)) > 1e-3


# ---- Downsampler -------------------------------------------------------------
DS_CASES = [
    (200e6, 50e6, 40e6, 3.0, 120),  # cfg2
    (1024000.0, 384000.0, 200000.0, 3.0, 34),  # simple_receiver.rs:28
    (384000.0, 48000.0, 40000.0, 3.0, 288),  # simple_receiver.rs:52
    (1024000.0, 102400.0, 60000.0, 3.0, 145),  # bandwidth_meter/main.rs:56
]


@pytest.mark.parametrize("fin,fout,bw,q,L", DS_CASES)
def test_downsampler_vs_numpy(oracle, fin, fout, bw, q, L):
    ir = onp.downsampler_ir(fin, fout, bw, q)
    assert ir.size == L
    n = 6000
    x = oracle.synth_iq(5, 0, n)
    emit = onp.emit_indices(n, fin, fout)
    want = onp.downsample(x, ir, emit)
    for flt, tol in ((np.float64, 1e-13), (np.float32, 1e-6)):
        d = oracle.Downsampler(1024, fout, bw, q, flt=flt)
        pieces = [d.process(fin, x[a:b]) for a, b in ((0, 7), (7, 7), (7, 2048), (2048, 2049), (2049, n))]
        got = np.concatenate(pieces)
        assert got.size == emit.size
        assert rms_rel(got, want) < tol
        assert np.max(np.abs(d.ir() - ir)) < (1e-15 if flt == np.float64 else 1e-7)
    assert np.sum(ir * ir) == pytest.approx(1.0, rel=1e-14)  # unit ENERGY, not unit gain


def test_downsampler_contract(oracle):
    with pytest.raises(AssertionError):
        oracle.Downsampler(16, 48000.0, 48000.0)  # bandwidth must be < output rate
    d = oracle.Downsampler(16, 48000.0, 40000.0)
    with pytest.raises(AssertionError):
        d.process(44100.0, np.zeros(8, dtype=np.complex64))  # input rate < output rate


def test_downsampler_output_chunks(oracle):
    d = oracle.Downsampler(100, 50e6, 40e6, flt=np.float32)
    x = oracle.synth_iq(6, 0, 1000)
    chunks = d.feed(200e6, x)
    assert [c.size for c in chunks] == [100, 100]  # 250 outputs -> 2 full chunks, 50 pending
    chunks = d.feed(200e6, x[:200])
    assert [c.size for c in chunks] == [100]


# ---- Fourier ---------------------------------------------------------------
@pytest.mark.parametrize("n,center", [(4096, False), (4096, True), (256, False), (1000, True), (7, True)])
def test_fourier_vs_numpy(oracle, n, center):
    win = oracle.Kaiser.with_null_at_bin(2.0)
    rel = np.array([win.relative_value_at(p) for p in onp.window_positions(n)])
    w = onp.fourier_window(n, rel)
    assert np.mean(w * w) == pytest.approx(1.0, rel=1e-13)
    x = oracle.synth_iq(7, 0, n)
    want = onp.fourier(x, w, center)
    for flt, tol in ((np.float64, 1e-13), (np.float32, 2e-6)):
        f = oracle.Fourier(win, center, flt=flt)
        assert rms_rel(f.process(x), want) < tol
        assert np.max(np.abs(f.window_values() - w)) < (1e-14 if flt == np.float64 else 2e-7)


# ---- synthetic input ---------------------------------------------------------
def test_synth_iq_is_counter_based(oracle):
    a = oracle.synth_iq(1, 0, 1000)
    b = oracle.synth_iq(1, 500, 500)
    assert np.array_equal(a[500:], b)
    c = oracle.synth_iq(2, 0, 1000)
    assert not np.array_equal(a, c)
    assert np.max(np.abs(a.real)) <= 1.0 and np.max(np.abs(a.imag)) <= 1.0
    # the two tones are there: fs/16 and -3fs/32
    X = np.abs(np.fft.fft(oracle.synth_iq(1, 0, 4096)))
    top = set(np.argsort(X)[-2:].tolist())
    assert top == {4096 // 16, 4096 - 3 * 4096 // 32}


def test_threaded_chain_runner_is_bit_equal(oracle):
    """rro_chain_run_mt (one thread per block, capacity-1 hand-off, the CPU baseline of bench.py)
    produces the spectra of the single-threaded runner bit for bit, for several message sizes."""
    lp = lambda _b, f: 1.0 if abs(f) <= 20e6 else 0.0  # noqa: E731
    kw = dict(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
              fft_window=oracle.Kaiser.with_null_at_bin(2.0), flt=np.float32)
    x = oracle.synth_iq(1, 0, (1 << 18) + 100)
    a, fa = oracle.run_chain_c(x, 200e6, **kw)
    assert fa == 16  # (2^18 + 64 - 64 swallowed by the Filter) / 4 = 65536 decimated samples
    for batch in (1, 7, 256):
        b, fb = oracle.run_chain_c(x, 200e6, threads=4, batch=batch, **kw)
        assert fb == fa and np.array_equal(a, b)


def test_native_timing_build_agrees_with_parity_build():
    """bench.py's cpu_baseline times the oracle's sources compiled for the host (`-march=native`, contraction
    allowed; the fastest of a few flag sets).  That build is never the checker, but it must be the same
    computation: its spectra agree with the parity build's to f32 rounding."""
    from oracle import rr_oracle as o

    path, flags, rates = o.pick_native()
    assert flags in o.NATIVE_FLAG_SETS and set(rates) == set(o.NATIVE_FLAG_SETS) and os.path.exists(path)
    x = o.synth_iq(3, 0, 1 << 17)
    kw = dict(shift=25e6, filter_len=64, freq_resp=lambda _b, f: 1.0 if abs(f) <= 20e6 else 0.0, output_rate=50e6,
              bandwidth=40e6, fft_len=4096, fft_window=o.Kaiser.with_null_at_bin(2.0), flt=np.float32)
    a, fa = o.run_chain_c(x, 200e6, **kw)
    with o.native():
        b, fb = o.run_chain_c(x, 200e6, **kw)
    c, _ = o.run_chain_c(x, 200e6, **kw)  # back on the parity build
    assert fa == fb and np.array_equal(a, c)
    err = np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(a) ** 2))
    assert err < 1e-5, err
    assert o.cpu_model()
