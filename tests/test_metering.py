"""metering::{level, bandwidth, rescale_energy} and GainControl: the reference's own
known-answer tests (src/metering.rs:115-259, src/blocks/transform.rs:396-416) against
the oracle (CPU) and against the GPU functions (-m gpu).  These rows are parity PINNED:
the vectors below are the reference's."""
import math

import numpy as np
import pytest

from conftest import assert_approx

H = 1.0 / math.sqrt(2.0)
S5 = math.sqrt(0.5)
OSC = [1, H + 1j * H, 1j, -H + 1j * H, -1, -H - 1j * H, -1j, H - 1j * H]
BW_CASES = [
    ([0, 0], 0.0),
    ([1, 1, 1, 1, 1, 1, -1, S5 - 1j * S5], 0.99 * 48000.0),
    ([7.4 - 2.1j] * 3, 0.99 * 48000.0),
    ([0, 0, 0, 0, 0, 0, 2.1, 0], 0.99 * 48000.0 / 8.0),
    ([1.5, 0, 0, 0, 0, 0, 1.5, 0], 2.98 * 48000.0 / 8.0),
]
RESCALE_CASES = [
    (3, [0, 2 + 1j, -0.5], [0.0, 5.0, 0.25]),
    (3, [1, 2, 3, 4], [2.3333333333333, 8.6666666666667, 19.0]),
    (4, [1, 2, 3], [0.75, 2.25, 4.25, 6.75]),
]


def run_kats(level, bandwidth, rescale_energy, gain_fn):
    assert_approx(math.log10(level(OSC)) * 10.0, 0.0)
    for bins, want in BW_CASES:
        assert_approx(bandwidth(0.01, 48000.0, bins), want)
    for res, inp, want in RESCALE_CASES:
        out = rescale_energy(res, inp)
        assert len(out) == res
        for g, w in zip(out, want):
            assert_approx(float(g), w, 1e-10 if np.asarray(out).dtype == np.float64 else 1e-6)
    y = gain_fn(0.25, np.array([32 - 1j, 15 - 2j], dtype=np.complex64))
    assert y[0].real == 8.0 and y[0].imag == -0.25 and y[1].real == 3.75 and y[1].imag == -0.5


def test_reference_kats_on_oracle(oracle):
    run_kats(lambda c: oracle.level(c, np.float64), lambda p, fs, b: oracle.bandwidth(p, fs, b, np.float64),
             lambda r, i: oracle.rescale_energy(r, i, np.float64), lambda g, c: oracle.gain(g, c, np.float32))


@pytest.mark.gpu
def test_reference_kats_on_gpu():
    import torch

    assert torch.cuda.is_available()
    from radiorust_amd import metering as m
    from radiorust_amd.signal import Samples

    def gain_fn(g, c):
        return m.GainControl.new(g).process(Samples(48000.0, c))[0].chunk

    run_kats(lambda c: m.level(c, np.float64), lambda p, fs, b: m.bandwidth(p, fs, b, np.float64),
             lambda r, i: m.rescale_energy(r, i, np.float64), gain_fn)


@pytest.mark.gpu
def test_metering_bit_equal_to_oracle_on_spectra(oracle):
    """Same f64 accumulation order as the reference: results are bit-identical to the
    oracle's on real spectra, for f32 and f64 inputs, batched on the device."""
    import ctypes as C

    import torch

    import radiorust_amd as rr
    from radiorust_amd import metering as m

    x = oracle.synth_iq(31, 0, 4096 * 5)
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), flt=np.float32)
    spectra = np.stack([fo.process(x[i * 4096 : (i + 1) * 4096]) for i in range(5)])
    for k in range(5):
        assert m.level(spectra[k]) == oracle.level(spectra[k], np.float32)
        assert m.bandwidth(0.01, 50e6, spectra[k]) == oracle.bandwidth(0.01, 50e6, spectra[k], np.float32)
        assert np.array_equal(m.rescale_energy(300, spectra[k]), oracle.rescale_energy(300, spectra[k], np.float32))
        assert np.array_equal(m.rescale_energy(5000, spectra[k]), oracle.rescale_energy(5000, spectra[k], np.float32))
    s64 = spectra.astype(np.complex128)
    assert m.bandwidth(0.05, 50e6, s64[0], np.float64) == oracle.bandwidth(0.05, 50e6, s64[0], np.float64)
    # batched, device resident
    L = rr._lib.lib()
    d = torch.from_numpy(spectra).cuda()
    out = torch.empty(5, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rr._lib.check(L.rr_bandwidth_dev(0, 0, C.c_void_p(st), 0.01, 50e6, d.data_ptr(), 4096, 5, out.data_ptr()))
    torch.cuda.synchronize()
    assert out.cpu().tolist() == [oracle.bandwidth(0.01, 50e6, spectra[k], np.float32) for k in range(5)]
    g = m.GainControl.new(0.3)
    y = g.process(rr.Samples(1.0, x[:1000]))[0].chunk
    assert np.array_equal(y, oracle.gain(0.3, x[:1000], np.float32))
    g.set(2.0)
    assert g.get() == 2.0
