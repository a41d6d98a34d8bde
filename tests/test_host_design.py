"""CPU: the product's host-side design math (csrc/rr_design.cpp, reached through
the C ABI) against the oracle.  No GPU involved."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import assert_approx


@pytest.fixture(scope="module")
def L():
    from radiorust_amd import _lib, build

    build.build_library()
    return _lib.lib()


def lowpass(cut):
    return lambda _b, f: 1.0 if abs(f) <= cut else 0.0


def test_math_kats_on_the_product(L):
    """math.rs:55-85 vectors through the product's own implementation."""
    assert L.rr_bessel_i0(0.0) == 1.0
    assert L.rr_bessel_i0(math.inf) == math.inf and L.rr_bessel_i0(-math.inf) == math.inf
    assert math.isnan(L.rr_bessel_i0(math.nan))
    for x, want in [(0.5, 1.06348337074132), (-0.5, 1.06348337074132), (1.23, 1.41552757215846),
                    (15.8, 736184.938479417), (456.0, 2.04094157812291e196)]:
        assert_approx(L.rr_bessel_i0(x), want)
    assert L.rr_bessel_i0(1000.0) == math.inf
    assert L.rr_sinc(0.0) == 1.0
    for x, want in [(0.4, 0.756826728640657), (1.0, 0.0), (2.6, 0.11643488132933186),
                    (-5.8, -0.03225825116512552), (17.0, 0.0), (-2345.0, 0.0)]:
        assert_approx(L.rr_sinc(x), want)


def test_math_bit_equal_to_oracle(L, oracle):
    for x in np.linspace(-20, 20, 101):
        assert L.rr_bessel_i0(x) == oracle.bessel_I0(x)
        assert L.rr_sinc(x) == oracle.sinc(x)
    for b in (0.0, math.sqrt(3), 7.0):
        for x in np.linspace(-1, 1, 33):
            assert L.rr_kaiser_rel_with_beta(b, x) == oracle.kaiser_rel_with_beta(b, x)
    assert L.rr_kaiser_null_at_bin_to_beta(2.0) == oracle.kaiser_null_at_bin_to_beta(2.0)
    assert L.rr_kaiser_alpha_to_beta(1.25) == oracle.kaiser_alpha_to_beta(1.25)


def test_deemphasis_factor(oracle):
    """blocks::filters::deemphasis_factor (filters.rs:20-27): product == oracle bit for bit, both == 1 / (1 + j tau 2 pi f)."""
    import radiorust_amd as rr

    for tau, f in ((50e-6, 0.0), (50e-6, 20.0), (50e-6, 3183.1), (75e-6, -16000.0), (50e-6, 1e9), (1e-3, 1e-3)):
        a, b = rr.deemphasis_factor(tau, f), oracle.deemphasis_factor(tau, f)
        assert a == b
        exact = 1.0 / complex(1.0, tau * 2.0 * math.pi * f)
        assert abs(a - exact) <= 4e-16 * abs(exact)
    assert rr.deemphasis_factor(50e-6, 0.0) == 1.0


@pytest.mark.parametrize("fs,prec,shift", [(48000.0, 1.0, 700.0), (200e6, 1.0, 25e6), (200e6, 1e3, 12.345e6),
                                           (48000.0, 1.0, -700.0), (48000.0, 1.0, 0.0), (1024000.0, 1.0, 200e3),
                                           (44100.0, 0.5, 1234.56)])
def test_ratio_and_table(L, oracle, fs, prec, shift):
    n, d = C.c_int64(), C.c_int64()
    assert L.rr_freqshifter_ratio(fs, prec, shift, C.byref(n), C.byref(d)) == 0
    assert (n.value, d.value) == oracle.freq_to_ratio(fs, prec, shift)
    for code, flt, cdt in ((0, np.float32, np.complex64), (1, np.float64, np.complex128)):
        tab = np.empty(d.value, dtype=cdt)
        assert L.rr_freqshifter_table(code, n.value, d.value, 0.0, tab.ctypes.data) == 0
        sh = oracle.FreqShifter(prec, shift, flt=flt)
        sh.process(fs, np.zeros(1, dtype=cdt))
        if flt == np.float32:
            assert np.array_equal(tab, sh.table())  # same libm, same expression: bit equal
        else:  # gcc and clang differ in fusing sin+cos into sincos: 1 ulp in f64
            assert np.max(np.abs(tab - sh.table())) <= 2.3e-16


def test_ratio_contract(L):
    n, d = C.c_int64(), C.c_int64()
    assert L.rr_freqshifter_ratio(10.0, 100.0, 1.0, C.byref(n), C.byref(d)) == 4  # denom rounds to 0 -> panic


@pytest.mark.parametrize("n,fs,cut", [(64, 200e6, 20e6), (1024, 2e9, 200e6), (4096, 48000.0, 16e3), (48, 48000.0, 4e3),
                                      (33, 48000.0, 5e3), (1, 48000.0, 5e3), (2, 48000.0, 5e3)])
def test_filter_taps_vs_oracle(L, oracle, n, fs, cut):
    import radiorust_amd as rr

    resp = rr.sample_freq_resp(lowpass(cut), n, fs)
    win = rr.Kaiser.with_null_at_bin(2.0).sample(n)
    taps = np.empty(n, dtype=np.complex128)
    assert L.rr_filter_design_taps(n, resp.ctypes.data, win.ctypes.data, taps.ctypes.data) == 0
    f = oracle.Filter(lowpass(cut), flt=np.float64)
    f.process(fs, np.zeros(n, dtype=np.complex128))
    want = 2 * n * f.response()
    assert np.max(np.abs(taps - want)) <= 1e-13 * max(np.max(np.abs(want)), 1e-300)


def test_filter_taps_complex_response(L, oracle):
    import radiorust_amd as rr

    n, fs = 96, 48000.0
    fr = lambda b, f: (0.5 - 1.0j) if (0 < f < 7000) else (0.25 if b == 0 else 0.0)  # noqa: E731
    resp = rr.sample_freq_resp(fr, n, fs)
    win = rr.Rectangular().sample(n)
    taps = np.empty(n, dtype=np.complex128)
    assert L.rr_filter_design_taps(n, resp.ctypes.data, win.ctypes.data, taps.ctypes.data) == 0
    f = oracle.Filter(fr, oracle.Rectangular(), flt=np.float64)
    f.process(fs, np.zeros(n, dtype=np.complex128))
    want = 2 * n * f.response()
    assert np.max(np.abs(taps - want)) <= 1e-13 * np.max(np.abs(want))


@pytest.mark.parametrize("fin,fout,bw,q,Lw", [(200e6, 50e6, 40e6, 3.0, 120), (1024000.0, 384000.0, 200000.0, 3.0, 34),
                                              (384000.0, 48000.0, 40000.0, 3.0, 288), (1024000.0, 102400.0, 60000.0, 3.0, 145),
                                              (48000.0, 48000.0, 20000.0, 1.0, 4)])
def test_downsampler_design_vs_oracle(L, oracle, fin, fout, bw, q, Lw):
    n = C.c_size_t()
    assert L.rr_downsampler_design(fin, fout, bw, q, C.byref(n), None, 0) == 0
    assert n.value == Lw
    ir = np.empty(n.value, dtype=np.float64)
    assert L.rr_downsampler_design(fin, fout, bw, q, C.byref(n), ir.ctypes.data, ir.size) == 0
    d = oracle.Downsampler(16, fout, bw, q, flt=np.float64)
    d.process(fin, np.zeros(1, dtype=np.complex128))
    assert np.array_equal(ir, d.ir())


def test_downsampler_design_contract(L):
    n = C.c_size_t()
    assert L.rr_downsampler_design(48000.0, 48000.0, 48000.0, 3.0, C.byref(n), None, 0) == 4  # bw !< out
    assert L.rr_downsampler_design(44100.0, 48000.0, 20000.0, 3.0, C.byref(n), None, 0) == 4  # in < out
    assert L.rr_downsampler_design(48000.0, -1.0, -2.0, 3.0, C.byref(n), None, 0) == 4


@pytest.mark.parametrize("fin,fout", [(200e6, 50e6), (1024000.0, 384000.0), (384000.0, 48000.0), (1024000.0, 102400.0),
                                      (48000.0, 44100.5), (48000.0, 48000.0), (3.0, 2.0)])
def test_schedule_vs_oracle(L, oracle, fin, fout):
    """The emit schedule equals what the oracle's sample-by-sample loop does,
    also when resumed across ragged chunk boundaries."""
    bw = fout * 0.5
    d = oracle.Downsampler(16, fout, bw, 1.0, flt=np.float64)
    total = 5000
    # reference count per chunk from the oracle
    pieces = [(0, 1), (1, 3), (3, 1000), (1000, 1000), (1000, 4097), (4097, total)]
    pos = C.c_double(0.0)
    all_emit = []
    for a, b in pieces:
        want = len(d.process(fin, np.ones(b - a, dtype=np.complex128)))
        cnt = C.c_size_t()
        emit = np.empty(b - a + 1, dtype=np.uint32)
        assert L.rr_downsampler_schedule(fin, fout, b - a, C.byref(pos), emit.ctypes.data, emit.size, C.byref(cnt)) == 0
        assert cnt.value == want
        all_emit.extend((emit[: cnt.value].astype(np.int64) + a).tolist())
    from oracle import oracle_np as onp

    if float(fin).is_integer() and float(fout).is_integer():
        assert all_emit == onp.emit_indices(total, fin, fout).tolist()


@pytest.mark.parametrize("fin,fout", [(48000.0, 44100.5), (1000.25, 333.125), (3.5, 2.25), (1024000.0, 44100.0), (2.0 ** 40 + 0.5, 2.0 ** 39 + 0.25),
                                      (48000.0, 44100.1)])
def test_schedule_closed_form_for_dyadic_rates(L, fin, fout):
    """Rates that are whole multiples of 2^-s (every f64 has such an s; the sums of resampling.rs:110-112 stay exact while
    (in + out) 2^s <= 2^53) run the closed form: counts and the accumulator after ragged pieces equal the sample loop's
    (the call with an emit list runs the loop).  44100.1 needs s = 37 - beyond 2^53, the loop on both sides."""
    pos_loop, pos_closed = C.c_double(0.0), C.c_double(0.0)
    for n in (1, 2, 997, 1000, 4097, 30000, 1):
        c1, c2 = C.c_size_t(), C.c_size_t()
        emit = np.empty(n + 1, dtype=np.uint32)
        assert L.rr_downsampler_schedule(fin, fout, n, C.byref(pos_loop), emit.ctypes.data, emit.size, C.byref(c1)) == 0
        assert L.rr_downsampler_schedule(fin, fout, n, C.byref(pos_closed), None, 0, C.byref(c2)) == 0
        assert c1.value == c2.value and pos_loop.value == pos_closed.value, (n, c1.value, c2.value, pos_loop.value, pos_closed.value)


@pytest.mark.parametrize("fin,fout", [(44100.0, 48000.0), (44100.5, 48000.0), (3.0, 7.0), (333.125, 1000.25), (1.0, 1.0), (48000.0, 384000.0),
                                      (2.0 ** 39 + 0.25, 2.0 ** 40 + 0.5), (44100.1, 48000.0)])
def test_upsampler_schedule_closed_form_for_dyadic_rates(L, fin, fout):
    """The interpolation schedule (resampling.rs:248-265) for rates on a 2^-s grid: counts and the carried position of the closed
    form (no list) equal the sample loop's (the call with a list), ragged pieces; the list itself is ceil((t out - pos) / in)."""
    pos_loop, pos_closed = C.c_double(0.0), C.c_double(0.0)
    for n in (1, 2, 997, 1000, 4097, 30000, 1, 0, 5):
        c1, c2 = C.c_size_t(), C.c_size_t()
        before = np.empty(n + 1, dtype=np.int32)
        p0 = pos_loop.value
        assert L.rr_upsampler_schedule(fin, fout, n, C.byref(pos_loop), before.ctypes.data, before.size, C.byref(c1)) == 0
        assert L.rr_upsampler_schedule(fin, fout, n, C.byref(pos_closed), None, 0, C.byref(c2)) == 0
        assert c1.value == c2.value and pos_loop.value == pos_closed.value, (n, c1.value, c2.value, pos_loop.value, pos_closed.value)
        if fin == 44100.0 and n:  # integers: the closed form in exact arithmetic
            want = [-((-(t * int(fout) - int(p0))) // int(fin)) for t in range(n)]
            assert before[:n].tolist() == want


@pytest.mark.parametrize("n", [1, 3, 4, 256, 1000, 4096])
def test_fourier_window_vs_oracle(L, oracle, n):
    import radiorust_amd as rr

    for pw, ow in ((rr.Rectangular(), oracle.Rectangular()), (rr.Kaiser.with_null_at_bin(2.0), oracle.Kaiser.with_null_at_bin(2.0)),
                   (rr.CustomWindow(lambda x: 1 - 0.5 * x * x), oracle.CustomWindow(lambda x: 1 - 0.5 * x * x))):
        rel = pw.sample(n)
        vals = np.empty(n, dtype=np.float64)
        assert L.rr_fourier_design_window(n, rel.ctypes.data, vals.ctypes.data) == 0
        f = oracle.Fourier(ow, flt=np.float64)
        f.process(np.zeros(n, dtype=np.complex128))
        assert np.array_equal(vals, f.window_values())


def test_sample_freq_resp_layout():
    import radiorust_amd as rr

    seen = []
    r = rr.sample_freq_resp(lambda b, f: seen.append((b, f)) or (b + 1j * f), 6, 600.0)
    # bins 0, +-1, +-2; Nyquist bin 3 stays zero (filters.rs:190-199)
    assert sorted(b for b, _ in seen) == [-2, -1, 0, 1, 2]
    assert r[3] == 0 and r[1] == 1 + 100j and r[5] == -1 - 100j
    seen.clear()
    rr.sample_freq_resp(lambda b, f: seen.append(b) or 0, 5, 500.0)
    assert sorted(seen) == [-2, -1, 0, 1, 2]
