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


def _serial_bandwidth_dev(rr, spectra_dev, n, count, dp, rate):
    """rr_bandwidth_dev (the serial kernel, the reference's summation order) on device-resident spectra."""
    import ctypes as C

    import torch

    out = torch.empty(count, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    code = 0 if spectra_dev.dtype == torch.complex64 else 1
    rr._lib.check(rr._lib.lib().rr_bandwidth_dev(code, 0, C.c_void_p(st), dp, rate, spectra_dev.data_ptr(), n, count, out.data_ptr()))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.gpu
def test_reference_kats_on_the_parallel_scan():
    """The reference's bandwidth KATs (metering.rs:131-259) through the workgroup-wide parallel form the metered pipelines use
    (rr_bandwidth_fast_dev: any frame length), at the reference's own tolerance (assert_approx: 1e-10)."""
    import ctypes as C

    import torch

    import radiorust_amd as rr

    L = rr._lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for dtype, code in ((np.complex128, 1), (np.complex64, 0)):
        for bins, want in BW_CASES:
            d = torch.from_numpy(np.asarray(bins, dtype=dtype)).cuda()
            bw = torch.empty(1, dtype=torch.float64, device="cuda")
            en = torch.empty(1, dtype=torch.float64, device="cuda")
            rr._lib.check(L.rr_bandwidth_fast_dev(code, 0, C.c_void_p(st), 0.01, 48000.0, d.data_ptr(), len(bins), 1,
                                                  bw.data_ptr(), en.data_ptr()))
            torch.cuda.synchronize()
            assert_approx(float(bw.item()), want, 1e-10 if code == 1 else 1e-6)
            assert abs(en.item() - float(np.sum(np.abs(np.asarray(bins, dtype=dtype)) ** 2))) <= 1e-6 * max(1.0, en.item())


@pytest.mark.gpu
@pytest.mark.parametrize("n,count", [(4096, 7), (1000, 5), (8192, 3), (300, 9), (20000, 2), (5, 4)])
def test_parallel_scan_matches_the_serial_kernel(oracle, n, count):
    """Frames of any length, f32 and f64: the parallel scan against the serial kernel on noise-like spectra with a few
    strong lines (crossings at both ends of the walk), to the last bits of the f64 sums."""
    import ctypes as C

    import torch

    import radiorust_amd as rr

    rng = np.random.default_rng(n)
    x = (rng.standard_normal((count, n)) + 1j * rng.standard_normal((count, n))) * 1e-2
    for k in range(count):
        x[k, rng.integers(0, n)] += 3.0
        x[k, rng.integers(0, n)] += 1.0 - 2.0j
    L = rr._lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for cdt, code, tol in ((np.complex64, 0, 1e-12), (np.complex128, 1, 1e-12)):
        d = torch.from_numpy(x.astype(cdt)).cuda()
        for dp in (0.01, 0.5, 1.9, 2.5):
            if n <= 8192:
                want = _serial_bandwidth_dev(rr, d, n, count, dp, 50e6)
            else:  # (the serial kernel stages a frame's energies in LDS: 8192 bins at most; the oracle's loop instead)
                want = np.array([oracle.bandwidth(dp, 50e6, x[k].astype(cdt), np.float32 if code == 0 else np.float64) for k in range(count)])
            bw = torch.empty(count, dtype=torch.float64, device="cuda")
            rr._lib.check(L.rr_bandwidth_fast_dev(code, 0, C.c_void_p(st), dp, 50e6, d.data_ptr(), n, count, bw.data_ptr(), None))
            torch.cuda.synchronize()
            got = bw.cpu().numpy()
            assert np.all(np.abs(got - want) <= tol * 50e6), (dp, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("hop_chunks,center_dc", [(4, False), (4, True), (1, False), (16, False)])
def test_stft_fused_bandwidth_epilogue(oracle, hop_chunks, center_dc, monkeypatch):
    """rr_stft_set_metering: the bandwidth of every 4096-point spectrum from the kernel that makes it (k_stft4096 / k_fft4096
    with the epilogue) against (a) the serial kernel run on the spectra the same call stored and (b) the oracle's
    metering::bandwidth on the oracle's spectra; then with store_spectra = 0 (no output buffer at all): the same figures;
    then with RR_METER_SERIAL=1: bit-equal to the serial kernel."""
    import torch

    import radiorust_amd as rr

    P = hop_chunks
    M = 4096 // P
    frames = 80
    n = M * (frames + P - 1)
    x = oracle.synth_iq(55, 0, n)
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(x).cuda()
    rate, dp = 102400.0, 0.01

    def run(store, serial=False):
        monkeypatch.setenv("RR_METER_SERIAL", "1" if serial else "0")
        g = rr.Stft(M, P, rr.Kaiser.with_null_at_bin(float(P)), center_dc=center_dc)
        g.set_stream(st)
        bw = torch.zeros(frames, dtype=torch.float64, device="cuda")
        en = torch.zeros(frames, dtype=torch.float64, device="cuda")
        g.set_metering(dp, rate, bw.data_ptr(), frames, en.data_ptr(), store_spectra=store)
        out = torch.zeros(frames * 4096, dtype=torch.complex64, device="cuda") if store else None
        got = g.process_dev(d_in.data_ptr(), n, out.data_ptr() if store else 0, frames * 4096 if store else 0)
        torch.cuda.synchronize()
        assert got == frames * 4096
        return bw.cpu().numpy(), en.cpu().numpy(), out

    bw, en, spectra = run(True)
    want = _serial_bandwidth_dev(rr, spectra, 4096, frames, dp, rate)
    assert np.all(np.abs(bw - want) <= 1e-12 * rate), np.max(np.abs(bw - want))
    e_ref = (spectra.view(frames, 4096).abs().double() ** 2).sum(dim=1).cpu().numpy()
    assert np.all(np.abs(en - e_ref) <= 1e-6 * e_ref)
    # the oracle's own pipeline: Overlapper + Fourier + metering::bandwidth in f64
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(float(P)), center_dc=center_dc, flt=np.float64)
    for k in (0, 1, frames // 2, frames - 1):
        ref = oracle.bandwidth(dp, rate, fo.process(x[k * M : k * M + 4096].astype(np.complex128)), np.float64)
        assert abs(bw[k] - ref) <= 1e-4 * ref + 1e-6, (k, bw[k], ref)
    bw2, en2, _ = run(False)
    assert np.array_equal(bw2, bw) and np.array_equal(en2, en)
    bw3, _, spectra3 = run(True, serial=True)
    assert np.array_equal(bw3, _serial_bandwidth_dev(rr, spectra3, 4096, frames, dp, rate))


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["frame", "wave", "blocks", "f64"])
def test_chain_metering_epilogue(oracle, path, monkeypatch):
    """rr_chain_set_metering on every path of the chain: the fused frame kernel (k_ols_frame with the epilogue), the
    two-kernel path (k_fft4096 with the epilogue), block by block, and Complex<f64> (the parallel scan behind the transform)."""
    import torch

    import radiorust_amd as rr

    if path == "frame":
        monkeypatch.setenv("RR_FUSED_KERNEL", "olsf")
    elif path == "wave":
        monkeypatch.setenv("RR_FUSED_KERNEL", "olsw")
    dtype = np.float64 if path == "f64" else np.float32
    fs, n = 200e6, 1 << 19
    lp = lambda _b, f: 1.0 if abs(f) <= 20e6 else 0.0  # noqa: E731
    g = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                 fft_window=rr.Kaiser.with_null_at_bin(2.0), allow_fused=path in ("frame", "wave"), dtype=dtype)
    st = torch.cuda.current_stream().cuda_stream
    g.set_stream(st)
    x = oracle.synth_iq(3, 0, n)
    tdt = torch.complex128 if path == "f64" else torch.complex64
    d_in = torch.from_numpy(x.astype(np.complex128 if path == "f64" else np.complex64)).cuda()
    cap = 64
    bw = torch.zeros(cap, dtype=torch.float64, device="cuda")
    en = torch.zeros(cap, dtype=torch.float64, device="cuda")
    out = torch.zeros(cap * 4096, dtype=tdt, device="cuda")
    g.set_metering(0.01, bw.data_ptr(), cap, en.data_ptr())
    esz = 16 if path == "f64" else 8
    total, pieces = 0, [(0, 100000), (100000, 300000), (300000, n)]
    for a, b in pieces:
        w = g.process_dev(fs, d_in.data_ptr() + esz * a, b - a, out.data_ptr(), cap * 4096)
        torch.cuda.synchronize()
        frames = w // 4096
        if frames:
            want = _serial_bandwidth_dev(rr, out[:w], 4096, frames, 0.01, 50e6)
            got = bw[:frames].cpu().numpy()
            assert np.all(np.abs(got - want) <= 1e-12 * 50e6), (a, b, np.max(np.abs(got - want)))
            e_ref = (out[:w].view(frames, 4096).abs().double() ** 2).sum(dim=1).cpu().numpy()
            assert np.all(np.abs(en[:frames].cpu().numpy() - e_ref) <= 1e-6 * e_ref)
        total += frames
    assert total == 31
    if path == "frame":
        assert g.last_path_kernel() == "k_ols_frame"
    # switched off again: the arrays stay untouched
    g.set_metering(0.01, 0, 0)
    bw.fill_(-1.0)
    g.process_dev(fs, d_in.data_ptr(), 1 << 17, out.data_ptr(), cap * 4096)
    torch.cuda.synchronize()
    assert float(bw.max().item()) == -1.0


@pytest.mark.gpu
def test_meter_process_bandwidth_is_the_examples_loop(oracle):
    """rr_meter_process_bandwidth = examples/bandwidth_meter/main.rs:53-78 in one call: samples in, one
    metering::bandwidth(0.01, ..) per spectrum out, against the oracle's blocks wired the same way + the oracle's bandwidth."""
    import radiorust_amd as rr

    fs, out_rate, max_bw, quality = 1024000.0, 102400.0, 60e3, 4
    resp = lambda _b, f: 1.0 if abs(f) <= max_bw / 2 else 0.0  # noqa: E731
    x = oracle.synth_iq(41, 0, 400000)
    g = rr.Meter(shift=12.5e3, output_rate=out_rate, bandwidth=max_bw, chunk_len=1024, freq_resp=resp, overlap=quality,
                 fft_window=rr.Kaiser.with_null_at_bin(float(quality)))
    sh = oracle.FreqShifter(1.0, 12.5e3, flt=np.float64)
    ds = oracle.Downsampler(1024, out_rate, max_bw, flt=np.float64)
    fl = oracle.Filter(resp, flt=np.float64)
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(float(quality)), flt=np.float64)
    hist, ref, got = [], [], []
    for a, b in ((0, 150001), (150001, 150002), (150002, 400000)):
        got += list(g.process_bandwidth(rr.Samples(fs, x[a:b]), 0.01))
        for c in ds.feed(fs, sh.process(fs, x[a:b])):
            z = fl.process(out_rate, c)
            if z is not None:
                hist.append(z)
                if len(hist) >= quality:
                    ref.append(oracle.bandwidth(0.01, out_rate, fo.process(np.concatenate(hist[-quality:])), np.float64))
                    del hist[: len(hist) - (quality - 1)]
    assert len(got) == len(ref) and len(ref) >= 30
    for a, b in zip(got, ref):
        assert abs(a - b) <= 1e-3 * b + 1e-6, (a, b)
