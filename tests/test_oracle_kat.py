"""The reference's own known-answer tests, run against the CPU oracle.

These are the ONLY vectors the reference holds for the hot path:
  * analysis.rs:139-209  test_fourier   (pins Fourier + the FFT convention)
  * math.rs:55-69        test_bessel_I0
  * math.rs:70-85        test_sinc
Values below are the reference's data, not its code.
"""
import math

import numpy as np

from conftest import assert_approx


def test_bessel_I0(oracle):
    o = oracle
    assert o.bessel_I0(0.0) == 1.0
    assert o.bessel_I0(-math.inf) == math.inf
    assert o.bessel_I0(math.inf) == math.inf
    assert math.isnan(o.bessel_I0(math.nan))
    assert_approx(o.bessel_I0(0.5), 1.06348337074132)
    assert_approx(o.bessel_I0(-0.5), 1.06348337074132)
    assert_approx(o.bessel_I0(1.23), 1.41552757215846)
    assert_approx(o.bessel_I0(15.8), 736184.938479417)
    assert_approx(o.bessel_I0(456.0), 2.04094157812291e196)
    assert o.bessel_I0(1000.0) == math.inf
    assert o.bessel_I0(-1000.0) == math.inf


def test_sinc(oracle):
    o = oracle
    assert o.sinc(0.0) == 1.0
    assert_approx(o.sinc(0.4), 0.756826728640657)
    assert_approx(o.sinc(-0.4), 0.756826728640657)
    assert_approx(o.sinc(1.0), 0.0)
    assert_approx(o.sinc(-1.0), 0.0)
    assert_approx(o.sinc(2.0), 0.0)
    assert_approx(o.sinc(2.6), 0.11643488132933186)
    assert_approx(o.sinc(-2.6), 0.11643488132933186)
    assert_approx(o.sinc(5.8), -0.03225825116512552)
    assert_approx(o.sinc(-5.8), -0.03225825116512552)
    assert_approx(o.sinc(17.0), 0.0)
    assert_approx(o.sinc(2345.0), 0.0)
    assert_approx(o.sinc(-2345.0), 0.0)


# analysis.rs:139-209: two Fourier<f64> blocks (plain / center_dc) fed the same
# two chunks, of length 3 then 4 (the plan has to change between them).
FOURIER_KAT = [
    ([1.0, 1.0, 1.0], [3, 0, 0], [0, 3, 0]),
    ([1.0, 1.5, 1.0, 0.5], [4, -1j, 0, 1j], [0, 1j, 4, -1j]),
]


def test_fourier_kat(oracle):
    f1 = oracle.Fourier(flt=np.float64)
    f2 = oracle.Fourier(center_dc=True, flt=np.float64)
    for chunk, want1, want2 in FOURIER_KAT:
        out1 = f1.process(np.array(chunk, dtype=np.complex128))
        out2 = f2.process(np.array(chunk, dtype=np.complex128))
        for got, want in ((out1, want1), (out2, want2)):
            for g, w in zip(got, want):
                assert_approx(g.real, complex(w).real)
                assert_approx(g.imag, complex(w).imag)


def test_fourier_kat_f32(oracle):
    """Same vectors through the f32 instantiation (what the GPU computes in);
    tolerance is f32 epsilon-scale, not 1e-10."""
    f1 = oracle.Fourier(flt=np.float32)
    f2 = oracle.Fourier(center_dc=True, flt=np.float32)
    for chunk, want1, want2 in FOURIER_KAT:
        out1 = f1.process(np.array(chunk, dtype=np.complex64))
        out2 = f2.process(np.array(chunk, dtype=np.complex64))
        np.testing.assert_allclose(out1, np.array(want1, dtype=np.complex64), atol=1e-6)
        np.testing.assert_allclose(out2, np.array(want2, dtype=np.complex64), atol=1e-6)
