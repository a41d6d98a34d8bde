"""GPU parity of the Upsampler (resampling.rs:147-280) and FmDemod (modulation.rs:83-158)
through the C ABI against the CPU oracle.

Upsampler: the gather kernel performs the reference's products and sums in the reference's
order without a*b+c contraction, so Complex<f32> and Complex<f64> outputs are compared
BIT-EXACTLY with the oracle of the same type.
FmDemod: everything but atan2 is exact; the device atan2f and glibc's differ by an ulp or
two, so the f32 comparison allows 4 ulp of pi * factor absolute."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rr():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    import radiorust_amd

    radiorust_amd._lib.lib()
    return radiorust_amd


CUTS = [0, 1, 1, 7, 2048, 2049, 9000, 20000]


@pytest.mark.parametrize("flt", [np.float32, np.float64])
@pytest.mark.parametrize("fi,fo,bw,q", [(48000.0, 384000.0, 40000.0, 3.0), (102400.0, 1024000.0, 60000.0, 3.0),
                                        (44100.0, 48000.0, 30000.0, 3.0), (48000.0, 48000.0, 20000.0, 1.0),
                                        (50e6, 200e6, 40e6, 3.0), (3.0, 7.0, 1.0, 2.0),
                                        (44100.5, 48000.0, 30000.0, 3.0), (333.125, 1000.25, 100.0, 2.0), (44100.1, 48000.0, 30000.0, 2.0)])
def test_upsampler_bit_exact(rr, oracle, fi, fo, bw, q, flt):
    cdt = np.complex64 if flt == np.float32 else np.complex128
    x = oracle.synth_iq(12, 0, CUTS[-1]).astype(cdt)
    g = rr.Upsampler.with_quality(1000, fo, bw, q, dtype=flt)
    o = oracle.Upsampler(1000, fo, bw, q, flt=flt)
    for a, b in zip(CUTS[:-1], CUTS[1:]):
        y, r = g.process_raw(fi, x[a:b]), o.process(fi, x[a:b])
        assert len(y) == len(r)  # identical release schedule, chunk by chunk
        assert np.array_equal(y.view(flt), r.view(flt)), (a, b)
    assert g.ir_len() == len(o.ir())


@pytest.mark.parametrize("fi,fo,bw", [(44100.0, 48000.0, 30000.0), (44100.5, 48000.0, 30000.0), (3.0, 7.0, 1.0)])
def test_upsampler_closed_form_schedule_long_stream(rr, oracle, fi, fo, bw, monkeypatch):
    """Rates on a 2^-s grid: k_upsample_closed (the schedule in closed form on the device, no list from the host) over a long
    ragged stream, bit-equal to the oracle; RR_UPSAMPLER_GENERIC=1 switched on and off in the middle of the stream (the list form
    rebuilds what the closed-form calls did not keep)."""
    n = 300000
    x = oracle.synth_iq(41, 0, n)
    g = rr.Upsampler.with_quality(1000, fo, bw, 3.0)
    o = oracle.Upsampler(1000, fo, bw, 3.0, flt=np.float32)
    cuts = [0, 1, 70001, 70002, 150000, 150007, 220000, 299999, n]
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if i in (3, 4, 6):
            monkeypatch.setenv("RR_UPSAMPLER_GENERIC", "1")
        else:
            monkeypatch.delenv("RR_UPSAMPLER_GENERIC", raising=False)
        y, r = g.process_raw(fi, x[a:b]), o.process(fi, x[a:b])
        assert len(y) == len(r)
        assert np.array_equal(y.view(np.float32), r.view(np.float32)), (i, a, b)


@pytest.mark.parametrize("U", [2, 3, 5, 6, 7, 10, 13, 16])
def test_upsampler_integer_ratio_kernel_bit_exact(rr, oracle, U):
    """Integer ratios 2 .. 16 in f32 run k_upsample_int for calls of >= 4096 outputs (a lane produces the U
    outputs one input releases, inputs staged in LDS, outputs leave through LDS in order): still bit-equal
    to the reference's scatter-add order, also at ragged call sizes and across the switch between kernels."""
    fi, fo = 1000.0, 1000.0 * U
    x = oracle.synth_iq(30 + U, 0, 12000)
    g = rr.Upsampler.with_quality(1000, fo, 420.0, 3.0)
    o = oracle.Upsampler(1000, fo, 420.0, 3.0, flt=np.float32)
    for a, b in zip([0, 5, 3000, 3001, 3300, 9001], [5, 3000, 3001, 3300, 9001, 12000]):
        y, r = g.process_raw(fi, x[a:b]), o.process(fi, x[a:b])
        assert len(y) == len(r) == U * (b - a)
        assert np.array_equal(y.view(np.float32), r.view(np.float32)), (a, b)


def test_upsampler_rate_change_chunks_and_events(rr, oracle):
    g = rr.Upsampler.new(500, 384000.0, 20000.0)
    o = oracle.Upsampler(500, 384000.0, 20000.0, flt=np.float32)
    x = oracle.synth_iq(13, 0, 3000)
    out, ref = [], []
    for rate, a, b in ((48000.0, 0, 1000), (96000.0, 1000, 2200), (48000.0, 2200, 3000)):
        out += g.process(rr.Samples(rate, x[a:b]))
        ref += o.feed(rate, x[a:b])
        ev = rr.EventSignal(rr.Disconnection())
        assert g.process(ev) == [ev]  # resampling.rs:269-271
    assert len(out) == len(ref) and all(len(s.chunk) == 500 and s.sample_rate == 384000.0 for s in out)
    assert np.array_equal(np.concatenate([s.chunk for s in out]).view(np.float32), np.concatenate(ref).view(np.float32))


def test_upsampler_contract(rr):
    from radiorust_amd._lib import ContractViolation

    with pytest.raises(ContractViolation):
        rr.Upsampler.new(16, -1.0, 10.0)
    g = rr.Upsampler.new(16, 48000.0, 20000.0)
    with pytest.raises(ContractViolation):
        g.process_raw(96000.0, np.zeros(8, dtype=np.complex64))
    with pytest.raises(ContractViolation):
        g.process_raw(16000.0, np.zeros(8, dtype=np.complex64))


@pytest.mark.parametrize("flt,tol_ulp", [(np.float32, 4), (np.float64, 4)])
def test_fmdemod_parity(rr, oracle, flt, tol_ulp):
    cdt = np.complex64 if flt == np.float32 else np.complex128
    fs, dev = 384000.0, 75000.0
    n = 50000
    t = np.arange(n)
    msg = 0.7 * np.sin(2 * np.pi * 1000 * t / fs) + 0.2 * np.sin(2 * np.pi * 19000 * t / fs)
    x = (np.exp(1j * np.cumsum(msg * dev / fs * 2 * np.pi)) * (1 + 0.01 * oracle.synth_iq(14, 0, n))).astype(cdt)
    g, o = rr.FmDemod(dev, dtype=flt), oracle.FmDemod(dev, flt=flt)
    factor = fs / dev / (2 * np.pi)
    atol = tol_ulp * np.finfo(flt).eps * np.pi * factor
    for a, b in ((0, 1), (1, 2), (2, 4097), (4097, 4097), (4097, n)):
        y, r = g.process_raw(fs, x[a:b]), o.process(fs, x[a:b])
        assert y.shape == r.shape and not np.any(y.imag)
        assert np.max(np.abs(y.real - r.real), initial=0.0) <= atol
    assert np.max(np.abs(y.real[-1000:] - msg[-1000:])) < 0.05  # and it demodulates


def test_fmdemod_long_call_octants_and_zeros(rr, oracle):
    """The long-call kernel computes arg() by a polynomial (k_fmdemod_pairs): every octant, the axes, zero samples
    (atan2's signed-zero rules) and magnitudes from 1e-12 to 1e12, against the oracle's libm atan2f."""
    rng = np.random.default_rng(5)
    n = 3 * 4096 + 1
    ang = rng.uniform(-np.pi, np.pi, n)
    mag = 10.0 ** rng.uniform(-12, 12, n)
    x = (mag * np.exp(1j * ang)).astype(np.complex64)
    x[100:140] = (1, 1j, -1, -1j, 1 + 1j, -1 + 1j, -1 - 1j, 1 - 1j) * 5   # products on the axes and diagonals
    x[200:204] = 0                                                       # 0 * conj(..) = +-0 +- 0j
    x[300] = complex(-0.0, 0.0)
    fs, dev = 48000.0, 2500.0
    g, o = rr.FmDemod(dev), oracle.FmDemod(dev, flt=np.float32)
    y, r = g.process_raw(fs, x), o.process(fs, x)
    atol = 4 * np.finfo(np.float32).eps * np.pi * fs / dev / (2 * np.pi)
    assert y.shape == r.shape and not np.any(y.imag)
    assert np.max(np.abs(y.real - r.real)) <= atol
    # ... and the short-call kernel (libm's atan2f on the device) agrees on the same stream
    g2 = rr.FmDemod(dev)
    y2 = np.concatenate([g2.process_raw(fs, x[a:a + 1000]) for a in range(0, n, 1000)])
    assert np.max(np.abs(y2.real - y.real)) <= atol


def test_fmdemod_interrupt_deviation_events(rr, oracle):
    fs = 48000.0
    x = oracle.synth_iq(15, 0, 96)
    g, o = rr.FmDemod(2500.0), oracle.FmDemod(2500.0, flt=np.float32)
    (s1,) = g.process(rr.Samples(fs, x[:32]))
    r1 = o.process(fs, x[:32])
    ev = rr.EventSignal(rr.SamplesLost())  # an interrupting event (signal.rs)
    assert g.process(ev) == [ev]
    o.interrupt()
    (s2,) = g.process(rr.Samples(fs, x[32:64]))
    r2 = o.process(fs, x[32:64])
    assert s2.chunk[0] == s1.chunk[-1]  # output_sample repeats (modulation.rs:121-127)
    g.set_deviation(5000.0)
    o.set_deviation(5000.0)
    assert g.deviation() == 5000.0
    (s3,) = g.process(rr.Samples(fs, x[64:]))
    r3 = o.process(fs, x[64:])
    atol = 4 * np.finfo(np.float32).eps * np.pi * fs / 2500.0 / (2 * np.pi)
    for s, r in ((s1, r1), (s2, r2), (s3, r3)):
        assert np.max(np.abs(s.chunk - r)) <= atol


def _rms(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / max(np.sum(np.abs(b) ** 2), 1e-300)))


@pytest.mark.parametrize("dtype,tol", [(np.float32, 1e-6), (np.float64, 1e-13)])
def test_gain_control_folded_into_the_block_in_front(rr, oracle, dtype, tol):
    """rr_*_set_gain: GainControl (transform.rs:29-92) behind a Downsampler (the reference's receiver,
    simple_receiver.rs:52-56), a Filter and an FmDemod without a pass of its own, against the oracle's two blocks one after
    the other; gain changes between calls take effect at the next call and keep the histories (watch semantics)."""
    import torch

    cdt = np.complex64 if dtype == np.float32 else np.complex128
    x = oracle.synth_iq(12, 0, 200000).astype(cdt)
    cuts = [0, 30000, 30001, 90000, 200000]
    gains = [0.25, 0.25, -1.7, 3.0]
    # Downsampler 8 : 1 (384 k -> 48 k, L = 288: the receiver's last stage, k_ols_wave in f32) and 10 : 1 (k_decim_poly)
    for out_rate, in_rate, bw in ((48000.0, 384000.0, 40000.0), (102400.0, 1024000.0, 60e3)):
        g = rr.Downsampler.new(1024, out_rate, bw, dtype=dtype)
        o = oracle.Downsampler(1024, out_rate, bw, flt=np.float64)
        for (a, b), gain in zip(zip(cuts[:-1], cuts[1:]), gains):
            g.set_gain(gain)
            got = g.process_raw(in_rate, x[a:b])
            want = oracle.gain(float(np.float32(gain)) if dtype == np.float32 else gain, o.process(in_rate, x[a:b].astype(np.complex128)), np.float64)
            assert len(got) == len(want)
            if len(want):
                assert _rms(got, want) <= max(tol, 1e-12), (out_rate, a, b, _rms(got, want))
    # Filter, n = 1000: chunk by chunk from the host (short calls), then 40 chunks in one device call (k_filter_blk4096 in f32)
    resp = lambda _b, f: 1.0 if abs(f) <= 9e3 else 0.0  # noqa: E731
    f = rr.Filter.new(resp, dtype=dtype)
    of = oracle.Filter(resp, flt=np.float64)
    n = 1000
    pos = 0
    for gain in (0.5, -2.0, 1.0):
        f.set_gain(gain)
        for _ in range(3):
            chunk = x[pos:pos + n]
            pos += n
            got = f.process(rr.Samples(48000.0, chunk))
            want = of.process(48000.0, chunk.astype(np.complex128))
            assert (want is None) == (len(got) == 0)
            if want is not None:
                assert _rms(got[0].chunk, (float(np.float32(gain)) if dtype == np.float32 else gain) * want) <= 10 * max(tol, 1e-12)
    f.set_gain(0.125)
    k = 40
    d_in = torch.from_numpy(x[pos:pos + k * n]).cuda()
    d_out = torch.empty(k * n, dtype=d_in.dtype, device="cuda")
    f.set_stream(torch.cuda.current_stream().cuda_stream)
    w = f.process_dev(48000.0, n, d_in.data_ptr(), k * n, d_out.data_ptr(), k * n)
    torch.cuda.synchronize()
    assert w == k * n
    want = np.concatenate([of.process(48000.0, x[pos + i * n:pos + (i + 1) * n].astype(np.complex128)) for i in range(k)])
    assert _rms(d_out.cpu().numpy(), 0.125 * want) <= 10 * max(tol, 1e-12)
    # FmDemod: FmDemod -> GainControl exactly (the gain rides on the store, the state keeps the demodulator's own output)
    d = rr.FmDemod(75000.0, dtype=dtype)
    od = oracle.FmDemod(75000.0, flt=dtype)
    for (a, b), gain in zip(zip(cuts[:-1], cuts[1:]), gains):
        d.set_gain(gain)
        got = d.process_raw(384000.0, x[a:b])
        want = oracle.gain(gain, od.process(384000.0, x[a:b]), dtype)
        if dtype == np.float64:
            assert np.max(np.abs(got - want)) <= 1e-12 * abs(gain)
        else:
            assert np.max(np.abs(got - want)) <= 4 * 2.4e-7 * 3.1416 * (384000.0 / 75000.0 / 6.2832) * abs(gain) * 1.01 + 1e-12
