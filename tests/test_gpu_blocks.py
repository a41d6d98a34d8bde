"""GPU parity of the four blocks, called through the C ABI (ctypes), against the
CPU oracle on identical synthetic IQ.

Tolerances
  * Complex<f32> data path: relative RMS error against the f64 oracle
        sqrt(sum |y - y_ref|^2 / sum |y_ref|^2) <= 1e-5        (north_star)
    and, as a regression guard, within a small multiple of the error the f32
    oracle itself makes (a wrong tap or index shows up as >> f32 rounding).
  * Complex<f64> data path: <= 1e-12 relative RMS; the reference's own
    known-answer test (analysis.rs:139-209) at its 1e-10.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_approx

pytestmark = pytest.mark.gpu

TOL = 1e-5


def lowpass(cut):
    return lambda _b, f: 1.0 if abs(f) <= cut else 0.0


def rms_rel(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    den = np.sum(np.abs(b) ** 2)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / den)) if den else float(np.max(np.abs(a - b)))


@pytest.fixture(scope="module")
def rr():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    import radiorust_amd

    radiorust_amd._lib.lib()  # the HIP library must be the thing that runs
    return radiorust_amd


def check(got, truth64, oracle32=None, tol=TOL):
    err = rms_rel(got, truth64)
    assert err <= tol, f"rms {err:g} > {tol:g}"
    if oracle32 is not None:
        ref_err = rms_rel(oracle32, truth64)
        assert err <= max(4 * ref_err, 4e-7), f"gpu rms {err:g} vs cpu-f32 rms {ref_err:g}"
    return err


# ---------------------------------------------------------------- source
def test_synth_source_bit_exact(rr, oracle):
    import torch

    for seed, t0, n in ((1, 0, 100000), (7, 123456789, 4097), (8, 2**40 + 5, 33)):
        buf = torch.empty(n, dtype=torch.complex64, device="cuda")
        rr.synth_iq_dev(0, torch.cuda.current_stream().cuda_stream, seed, t0, n, buf.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), oracle.synth_iq(seed, t0, n))


# ---------------------------------------------------------------- FreqShifter
@pytest.mark.parametrize("fs,prec,shift", [(48000.0, 1.0, 700.0), (200e6, 1.0, 25e6), (200e6, 1e3, 12.345e6),
                                           (48000.0, 1.0, -1234.0), (48000.0, 1.0, 0.0)])
def test_freqshifter_parity(rr, oracle, fs, prec, shift):
    x = oracle.synth_iq(1, 0, 200001)
    cuts = [0, 1, 1000, 1000, 4097, 65536, 200001]  # ragged, one empty chunk
    o64 = oracle.FreqShifter(prec, shift, flt=np.float64)
    o32 = oracle.FreqShifter(prec, shift, flt=np.float32)
    g = rr.FreqShifter.with_precision_and_shift(prec, shift)
    assert g.precision() == prec and g.shift() == shift
    got, t64, t32 = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        (s,) = g.process(rr.Samples(fs, x[a:b]))
        assert s.sample_rate == fs and len(s.chunk) == b - a
        got.append(s.chunk)
        t64.append(o64.process(fs, x[a:b]))
        t32.append(o32.process(fs, x[a:b]))
    check(np.concatenate(got), np.concatenate(t64), np.concatenate(t32))
    # same table, same products: agreement with the f32 oracle is at the ulp level
    assert rms_rel(np.concatenate(got), np.concatenate(t32)) < 1e-7


def test_freqshifter_retune_and_events(rr, oracle):
    fs = 48000.0
    x = oracle.synth_iq(2, 0, 3000)
    g = rr.FreqShifter.with_shift(700.0)
    o64 = oracle.FreqShifter(1.0, 700.0, flt=np.float64)
    o32 = oracle.FreqShifter(1.0, 700.0, flt=np.float32)
    outs, refs = [], []
    outs.append(g.process(rr.Samples(fs, x[:777]))[0].chunk)
    refs.append(o32.process(fs, x[:777]))
    o64.process(fs, x[:777])
    ev = rr.EventSignal(rr.Disconnection())
    assert g.process(ev) == [ev]  # forwarded unchanged, no state change
    g.set_shift(-2500.0)
    o32.set_shift(-2500.0)
    assert g.shift() == -2500.0
    outs.append(g.process(rr.Samples(fs, x[777:2000]))[0].chunk)
    refs.append(o32.process(fs, x[777:2000]))
    g.update_shift(lambda s: s + 100.0)
    o32.set_shift(-2400.0)
    outs.append(g.process(rr.Samples(fs, x[2000:]))[0].chunk)
    refs.append(o32.process(fs, x[2000:]))
    # phase continuity goes through atan2f of an f32 phasor on both sides
    assert rms_rel(np.concatenate(outs), np.concatenate(refs)) < 5e-7
    # sample-rate change alone also rebuilds the table (transform.rs:318-319)
    y = g.process(rr.Samples(44100.0, x[:100]))[0].chunk
    assert rms_rel(y, o32.process(44100.0, x[:100])) < 5e-7


def test_freqshifter_f64(rr, oracle):
    x = oracle.synth_iq(3, 0, 50000).astype(np.complex128)
    g = rr.FreqShifter.with_shift(700.0, dtype=np.float64)
    o64 = oracle.FreqShifter(1.0, 700.0, flt=np.float64)
    y = np.concatenate([g.process(rr.Samples(48000.0, x[a:b]))[0].chunk for a, b in ((0, 12345), (12345, 50000))])
    assert y.dtype == np.complex128
    assert rms_rel(y, o64.process(48000.0, x)) < 1e-15


def test_freqshifter_errors(rr):
    from radiorust_amd._lib import RR_ERR_BAD_ARG, RR_ERR_CAPACITY, BackendError, ContractViolation

    g = rr.FreqShifter.with_precision_and_shift(100.0, 1.0)
    with pytest.raises(ContractViolation):  # 10 / 100 rounds to denom 0: Ratio::new panics
        g.process(rr.Samples(10.0, np.zeros(4, dtype=np.complex64)))
    g2 = rr.FreqShifter.with_precision_and_shift(1e-3, 1.001)  # 1001 / 2e11 does not reduce
    with pytest.raises(BackendError) as e:  # 2e11-entry table
        g2.process(rr.Samples(200e6, np.zeros(4, dtype=np.complex64)))
    assert e.value.status == RR_ERR_BAD_ARG
    L = rr._lib.lib()
    g3 = rr.FreqShifter.with_shift(1.0)
    x = np.zeros(8, dtype=np.complex64)
    n = C.c_size_t(99)
    assert L.rr_freqshifter_process(g3._h, 48000.0, x.ctypes.data, 8, x.ctypes.data, 4, C.byref(n)) == RR_ERR_CAPACITY
    assert n.value == 0
    with pytest.raises(TypeError):
        rr.FreqShifter.with_shift(1.0, dtype=np.float16)


# ---------------------------------------------------------------- Filter
FILTER_CASES = [(64, 200e6, 20e6, 40), (1024, 2e9, 200e6, 6), (4096, 48000.0, 16e3, 3), (48, 48000.0, 4e3, 9), (33, 48000.0, 5e3, 9),
                (1000, 48000.0, 6e3, 5), (1537, 96000.0, 9e3, 4),          # any length up to 2048: the 4096-point block kernel
                (3000, 48000.0, 7e3, 4), (8192, 1024000.0, 100e3, 4),      # longer: partitions of 2048 taps
                (16384, 1024000.0, 100e3, 3)]                             # examples/simple_receiver.rs:28-37: chunks of 16384


@pytest.mark.parametrize("n,fs,cut,chunks", FILTER_CASES)
def test_filter_parity(rr, oracle, n, fs, cut, chunks):
    x = oracle.synth_iq(4, 0, n * chunks)
    g = rr.Filter.new(lowpass(cut))
    o64 = oracle.Filter(lowpass(cut), flt=np.float64)
    o32 = oracle.Filter(lowpass(cut), flt=np.float32)
    got, t64, t32 = [], [], []
    for i in range(chunks):
        c = x[i * n : (i + 1) * n]
        out = g.process(rr.Samples(fs, c))
        r64, r32 = o64.process(fs, c), o32.process(fs, c)
        if i == 0:
            assert out == [] and r64 is None  # delay of one chunk (filters.rs:240,260)
            continue
        assert len(out) == 1 and len(out[0].chunk) == n
        got.append(out[0].chunk)
        t64.append(r64)
        t32.append(r32)
    check(np.concatenate(got), np.concatenate(t64), np.concatenate(t32))


def test_filter_interrupt_update_and_rate_change(rr, oracle):
    n, fs = 64, 200e6
    x = oracle.synth_iq(5, 0, n * 10)
    g = rr.Filter.new(lowpass(20e6))
    o = oracle.Filter(lowpass(20e6), flt=np.float64)

    def both(rate, c):
        out = g.process(rr.Samples(rate, c))
        ref = o.process(rate, c)
        assert (out == []) == (ref is None)
        if ref is not None:
            check(out[0].chunk, ref)

    both(fs, x[:n])
    both(fs, x[n : 2 * n])
    ev = rr.EventSignal(rr.SamplesLost())
    assert g.process(ev) == [ev]
    o.interrupt()
    both(fs, x[2 * n : 3 * n])  # swallowed again
    both(fs, x[3 * n : 4 * n])
    ev2 = rr.EventSignal(rr.Event())  # not an interrupt: history survives
    assert g.process(ev2) == [ev2]
    both(fs, x[4 * n : 5 * n])
    g.update(lowpass(5e6))
    o.update(lowpass(5e6))
    both(fs, x[5 * n : 6 * n])
    both(fs, x[6 * n : 7 * n])
    g.update_with_window(lowpass(5e6), rr.Rectangular())
    o.update(lowpass(5e6), oracle.Rectangular())
    both(fs, x[7 * n : 8 * n])
    both(fs, x[8 * n : 9 * n])
    both(fs / 2, x[:n])  # rate change -> redesign -> swallowed
    both(fs / 2, x[n : 2 * n])
    both(fs / 2, x[: n // 2])  # chunk length change -> redesign
    both(fs / 2, x[n // 2 : n])


def test_filter_complex_taps_and_windows(rr, oracle):
    n, fs = 64, 48000.0
    fr = lambda b, f: (1.0 + 0.5j) if 0 <= f <= 6000 else 0.0  # noqa: E731
    x = oracle.synth_iq(6, 0, n * 8)
    for gw, ow in ((rr.Rectangular(), oracle.Rectangular()), (rr.Kaiser.with_alpha(1.5), oracle.Kaiser.with_alpha(1.5)),
                   (rr.CustomWindow(lambda v: 1 - 0.9 * v * v), oracle.CustomWindow(lambda v: 1 - 0.9 * v * v))):
        g = rr.Filter.with_window(fr, gw)
        o64 = oracle.Filter(fr, ow, flt=np.float64)
        o32 = oracle.Filter(fr, ow, flt=np.float32)
        got, t64, t32 = [], [], []
        for i in range(8):
            c = x[i * n : (i + 1) * n]
            out = g.process(rr.Samples(fs, c))
            r64, r32 = o64.process(fs, c), o32.process(fs, c)
            if out:
                got.append(out[0].chunk)
                t64.append(r64)
                t32.append(r32)
        check(np.concatenate(got), np.concatenate(t64), np.concatenate(t32))


@pytest.mark.parametrize("n", [8192, 16384])
def test_filter_f64_long(rr, oracle, n):
    """Complex<f64> responses beyond one LDS tile of the direct-form kernel (6144 taps): the taps are taken in
    several passes.  The reference has no limit on the chunk length."""
    fs = 1024000.0
    x = oracle.synth_iq(9, 0, n * 3).astype(np.complex128)
    g = rr.Filter.new(lowpass(100e3), dtype=np.float64)
    o = oracle.Filter(lowpass(100e3), flt=np.float64)
    got, ref = [], []
    for i in range(3):
        out = g.process(rr.Samples(fs, x[i * n : (i + 1) * n]))
        r = o.process(fs, x[i * n : (i + 1) * n])
        if out:
            got.append(out[0].chunk)
            ref.append(r)
    assert len(got) == 2
    assert rms_rel(np.concatenate(got), np.concatenate(ref)) < 1e-12


@pytest.mark.parametrize("n,dtype,tol", [(4096, np.float32, 1e-5), (8192, np.float32, 1e-5), (16384, np.float32, 1e-5), (3000, np.float32, 1e-5),
                                          (8192, np.float64, 1e-11), (5000, np.float64, 1e-11)])
def test_filter_long_responses_at_n_log_n(rr, oracle, n, dtype, tol, monkeypatch):
    """Responses beyond 2048 taps (simple_receiver.rs:28-37 builds a Filter on 16384-sample chunks, i.e. 16384 taps): overlap-
    save with blocks of 2^14 .. 2^16 points through the two-pass tile transform (rr_filter::process_conv) instead of partitions
    of 2048 taps - several chunks per call, single chunks, an interrupt in between, device pointers; against the f64 oracle."""
    import torch

    monkeypatch.setenv("RR_FILTER_CONV_MIN", "2049")  # (the default threshold for f32 is 16384 taps: below it the partitions are ahead)
    fs = 1024000.0
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    chunks = 13
    x = oracle.synth_iq(19, 0, n * chunks).astype(cdt)
    g = rr.Filter.new(lowpass(100e3), dtype=dtype)
    o = oracle.Filter(lowpass(100e3), flt=np.float64)
    st = torch.cuda.current_stream().cuda_stream
    g.set_stream(st)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros(n * chunks, dtype=d_in.dtype, device="cuda")
    esz = x.itemsize
    pos = 0
    for k, interrupt in ((3, False), (1, False), (5, False), (2, True), (2, False)):
        if interrupt:
            g.process(rr.EventSignal(rr.SamplesLost()))
            o.interrupt()
        w = g.process_dev(fs, n, d_in.data_ptr() + esz * pos, k * n, d_out.data_ptr(), k * n)
        torch.cuda.synchronize()
        assert g.last_kernel() == 4, g.last_kernel()
        ref = [o.process(fs, x[pos + i * n:pos + (i + 1) * n].astype(np.complex128)) for i in range(k)]
        ref = np.concatenate([r for r in ref if r is not None] or [np.empty(0, np.complex128)])
        assert w == len(ref)
        if w:
            assert rms_rel(d_out[:w].cpu().numpy(), ref) <= tol, (k, rms_rel(d_out[:w].cpu().numpy(), ref))
        pos += k * n


@pytest.mark.parametrize("n", [2049, 3000, 4096, 5000, 8192])
def test_filter_blocks_of_16384_points(rr, oracle, n):
    """2049 .. 8192 taps in f32: k_filter_blk16k - blocks of 16 384 points in LDS, one forward and one inverse transform per
    block (kernel 5).  Several chunks per call, single chunks, an interrupt in between, a ragged tail of blocks; against the f64
    oracle, call by call and chunk by chunk."""
    import torch

    fs = 1024000.0
    chunks = 23
    x = oracle.synth_iq(23, 0, n * chunks).astype(np.complex64)
    g = rr.Filter.new(lowpass(100e3))
    o = oracle.Filter(lowpass(100e3), flt=np.float64)
    st = torch.cuda.current_stream().cuda_stream
    g.set_stream(st)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros(n * chunks, dtype=d_in.dtype, device="cuda")
    pos = 0
    for k, interrupt in ((3, False), (1, False), (9, False), (2, True), (1, False), (7, False)):
        if interrupt:
            g.process(rr.EventSignal(rr.SamplesLost()))
            o.interrupt()
        w = g.process_dev(fs, n, d_in.data_ptr() + 8 * pos, k * n, d_out.data_ptr(), k * n)
        torch.cuda.synchronize()
        assert g.last_kernel() == 5, g.last_kernel()
        ref = [o.process(fs, x[pos + i * n:pos + (i + 1) * n].astype(np.complex128)) for i in range(k)]
        ref = np.concatenate([r for r in ref if r is not None] or [np.empty(0, np.complex128)])
        assert w == len(ref)
        if w:
            e = rms_rel(d_out[:w].cpu().numpy(), ref)
            assert e <= 1e-5, (k, e)
            # every chunk on its own too (a wrong block would hide in the overall RMS)
            got = d_out[:w].cpu().numpy()
            for i in range(0, w, n):
                assert rms_rel(got[i:i + n], ref[i:i + n]) <= 1e-5, (k, i)
        pos += k * n


@pytest.mark.parametrize("n,onesided", [(64, False), (100, True), (1000, False), (2048, True), (2049, False)])
def test_filter_f64_blocks_of_4096_points(rr, oracle, n, onesided):
    """Complex<f64>, 2 .. 2049 taps, calls of >= 4096 outputs: k_ols4096_f64 (kernel 6) - real and complex (one-sided response)
    taps, several chunks per call, short calls on the other kernels in between, an interrupt; against the f64 oracle."""
    import torch

    fs = 1024000.0
    per = max(1, (9000 + n - 1) // n)  # chunks per long call
    plan = [(per, False), (1, False), (3 * per, False), (per, True), (2, False), (2 * per, False)]
    chunks = sum(k for k, _ in plan)
    x = oracle.synth_iq(29, 0, n * chunks).astype(np.complex128)
    resp = (lambda b, f: 1.0 if 0 <= f <= 100e3 else 0.0) if onesided else lowpass(100e3)
    g = rr.Filter.new(resp, dtype=np.float64)
    o = oracle.Filter(resp, flt=np.float64)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros(n * chunks, dtype=d_in.dtype, device="cuda")
    pos, seen = 0, set()
    for k, interrupt in plan:
        if interrupt:
            g.process(rr.EventSignal(rr.SamplesLost()))
            o.interrupt()
        w = g.process_dev(fs, n, d_in.data_ptr() + 16 * pos, k * n, d_out.data_ptr(), k * n)
        torch.cuda.synchronize()
        seen.add(g.last_kernel())
        ref = [o.process(fs, x[pos + i * n:pos + (i + 1) * n]) for i in range(k)]
        ref = np.concatenate([r for r in ref if r is not None] or [np.empty(0, np.complex128)])
        assert w == len(ref)
        if w >= 4096:
            assert g.last_kernel() == 6, g.last_kernel()
        if w:
            assert rms_rel(d_out[:w].cpu().numpy(), ref) <= 1e-12, (k, rms_rel(d_out[:w].cpu().numpy(), ref))
        pos += k * n
    assert 6 in seen


def test_filter_deemphasis_of_simple_receiver(rr, oracle):
    """examples/relm_app/simple_receiver.rs:43-49: the audio Filter behind the FM demodulator - rectangular window,
    complex response built from blocks::filters::deemphasis_factor(50e-6, f) on 20 Hz .. 16 kHz, DC bin blocked -
    at 384 kS/s in chunks of 16384 (the Downsampler in front of it emits those): complex taps through the
    partitioned block kernel."""
    n, fs = 16384, 384000.0

    def make_resp(mod):
        return lambda b, f: (mod.deemphasis_factor(50e-6, f) if abs(b) >= 1 and 20.0 <= abs(f) <= 16000.0 else 0.0)

    x = oracle.synth_iq(31, 0, n * 3)
    g = rr.Filter.new_rectangular(make_resp(rr))
    o64 = oracle.Filter(make_resp(oracle), oracle.Rectangular(), flt=np.float64)
    o32 = oracle.Filter(make_resp(oracle), oracle.Rectangular(), flt=np.float32)
    got, t64, t32 = [], [], []
    for i in range(3):
        c = x[i * n : (i + 1) * n]
        out = g.process(rr.Samples(fs, c))
        r64, r32 = o64.process(fs, c), o32.process(fs, c)
        if out:
            got.append(out[0].chunk)
            t64.append(r64)
            t32.append(r32)
    assert len(got) == 2
    check(np.concatenate(got), np.concatenate(t64), np.concatenate(t32))


def test_filter_f64(rr, oracle):
    n, fs = 64, 200e6
    x = oracle.synth_iq(7, 0, n * 6).astype(np.complex128)
    g = rr.Filter.new(lowpass(20e6), dtype=np.float64)
    o = oracle.Filter(lowpass(20e6), flt=np.float64)
    got, ref = [], []
    for i in range(6):
        out = g.process(rr.Samples(fs, x[i * n : (i + 1) * n]))
        r = o.process(fs, x[i * n : (i + 1) * n])
        if out:
            got.append(out[0].chunk)
            ref.append(r)
    assert rms_rel(np.concatenate(got), np.concatenate(ref)) < 1e-12


def test_filter_overlap_save_f64_and_batched(rr, oracle):
    """Long power-of-two filters run as overlap-save fast convolution (the
    reference's own recipe, filters.rs:240-259); BASELINE configs[4]: n = 1024 at 2 GS/s."""
    import torch

    n, fs, k = 256, 2e9, 7
    x = oracle.synth_iq(16, 0, n * k).astype(np.complex128)
    g = rr.Filter.new(lowpass(200e6), dtype=np.float64)
    o = oracle.Filter(lowpass(200e6), flt=np.float64)
    got, ref = [], []
    for i in range(k):
        out = g.process(rr.Samples(fs, x[i * n : (i + 1) * n]))
        r = o.process(fs, x[i * n : (i + 1) * n])
        if out:
            got.append(out[0].chunk)
            ref.append(r)
    assert rms_rel(np.concatenate(got), np.concatenate(ref)) < 1e-12
    # device-resident batch of chunks, f32, n = 1024 (cfg5)
    n, k = 1024, 64
    x = oracle.synth_iq(17, 0, n * k)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    g = rr.Filter.new(lowpass(200e6))
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    wrote = g.process_dev(fs, n, d_in.data_ptr(), n * k, d_out.data_ptr(), n * k)
    assert wrote == n * (k - 1)  # first chunk swallowed
    torch.cuda.synchronize()
    o64 = oracle.Filter(lowpass(200e6), flt=np.float64)
    o32 = oracle.Filter(lowpass(200e6), flt=np.float32)
    r64 = [o64.process(fs, x[i * n : (i + 1) * n]) for i in range(k)][1:]
    r32 = [o32.process(fs, x[i * n : (i + 1) * n]) for i in range(k)][1:]
    check(d_out.cpu().numpy()[:wrote], np.concatenate(r64), np.concatenate(r32))


@pytest.mark.parametrize("n,forced", [(64, None), (128, None), (48, None), (256, None), (385, "wave"), (300, "wave"), (64, "ols4096"), (128, "ols4096")])
def test_filter_short_power_of_two_long_calls(rr, oracle, n, forced, monkeypatch):
    """Short filters in f32: calls that produce >= 16384 samples run k_filter_wave (a wave per 1024-sample
    block; RR_FILTER_KERNEL=ols4096: k_filter_ols4096 for n = 64 / 128), shorter ones the small-call kernels;
    one stream through a mix of both, against the chunk-by-chunk oracle."""
    import torch

    if forced:
        monkeypatch.setenv("RR_FILTER_KERNEL", forced)
    fs = 200e6
    kb = -(-20000 // n)  # chunks of a call that is long enough for the big-call kernel
    ks = [3, kb + 7, 1, kb, 2, 2]  # chunks per call
    x = oracle.synth_iq(21, 0, n * sum(ks))
    g = rr.Filter.new(lowpass(20e6))
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    off = wrote = 0
    kernels = []
    for k in ks:
        wrote += g.process_dev(fs, n, d_in.data_ptr() + 8 * off, n * k, d_out.data_ptr() + 8 * wrote, n * k)
        kernels.append(g.last_kernel())
        off += n * k
    torch.cuda.synchronize()
    big = 2 if forced == "ols4096" else 3  # (k_filter_wave by default up to 256 taps; RR_FILTER_KERNEL=wave: up to its 385)
    assert kernels[1] == big and kernels[3] == big and 3 not in (kernels[0], kernels[2], kernels[4]), kernels
    assert wrote == n * (sum(ks) - 1)
    o64 = oracle.Filter(lowpass(20e6), flt=np.float64)
    o32 = oracle.Filter(lowpass(20e6), flt=np.float32)
    r64 = [o64.process(fs, x[i * n : (i + 1) * n]) for i in range(sum(ks))][1:]
    r32 = [o32.process(fs, x[i * n : (i + 1) * n]) for i in range(sum(ks))][1:]
    got = d_out.cpu().numpy()[:wrote]
    check(got, np.concatenate(r64), np.concatenate(r32))
    # the seams between the kernels
    r = np.concatenate(r64)
    for edge in (n * 2, n * (kb + 9), n * (kb + 10), n * (2 * kb + 10)):
        check(got[edge - n : edge + n], r[edge - n : edge + n])


def test_filter_needs_design_status(rr):
    from radiorust_amd._lib import RR_ERR_NEED_DESIGN

    L = rr._lib.lib()
    g = rr.Filter.new(lowpass(1e3))
    x = np.zeros(64, dtype=np.complex64)
    n = C.c_size_t()
    assert L.rr_filter_process(g._h, 48000.0, x.ctypes.data, 64, x.ctypes.data, 64, C.byref(n)) == RR_ERR_NEED_DESIGN
    g.process(rr.Samples(48000.0, x))
    assert L.rr_filter_process(g._h, 48000.0, x.ctypes.data, 32, x.ctypes.data, 64, C.byref(n)) == RR_ERR_NEED_DESIGN
    assert L.rr_filter_process(g._h, 44100.0, x.ctypes.data, 64, x.ctypes.data, 64, C.byref(n)) == RR_ERR_NEED_DESIGN
    assert L.rr_filter_process(g._h, 48000.0, x.ctypes.data, 64, x.ctypes.data, 64, C.byref(n)) == 0 and n.value == 64


# ---------------------------------------------------------------- Downsampler
DS_CASES = [(200e6, 50e6, 40e6, 3.0), (1024000.0, 384000.0, 200000.0, 3.0), (384000.0, 48000.0, 40000.0, 3.0),
            (1024000.0, 102400.0, 60000.0, 3.0), (48000.0, 44100.5, 30000.0, 2.0), (48000.0, 48000.0, 20000.0, 1.0)]


@pytest.mark.parametrize("fin,fout,bw,q", DS_CASES)
def test_downsampler_parity(rr, oracle, fin, fout, bw, q):
    n = 60000
    x = oracle.synth_iq(8, 0, n)
    cuts = [0, 7, 7, 2048, 2049, 30000, n]
    g = rr.Downsampler.with_quality(1000, fout, bw, q)
    o64 = oracle.Downsampler(1000, fout, bw, q, flt=np.float64)
    o32 = oracle.Downsampler(1000, fout, bw, q, flt=np.float32)
    got, t64, t32 = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        y = g.process_raw(fin, x[a:b])
        r64, r32 = o64.process(fin, x[a:b]), o32.process(fin, x[a:b])
        assert len(y) == len(r64)  # identical emission schedule, chunk by chunk
        got.append(y)
        t64.append(r64)
        t32.append(r32)
    assert g.ir_len() == len(o64.ir())
    check(np.concatenate(got), np.concatenate(t64), np.concatenate(t32))


# integer ratios 2, 4, 8 in f32: calls of >= 4096 samples run the chain's fused kernels with an all-ones
# NCO table (rr_downsampler_last_kernel: 1 k_mix_fir_decim, 2 k_ols_decim4, 3 k_ols_wave), other periodic
# schedules k_decim_poly (5), shorter calls k_fir; the stream of outputs must not notice the switches
FAST_CASES = [
    (200e6, 50e6, 40e6, 3.0, 3),      # cfg2's Downsampler: L = 120, D = 4 -> k_ols_wave
    (200e6, 50e6, 47e6, 3.0, 2),      # L = 400: beyond k_ols_wave's overlap -> k_ols_decim4
    (200e6, 50e6, 30e6, 3.0, 3),      # L = 60: short responses too (k_mix_fir_decim only on request since round 2)
    (384000.0, 48000.0, 40000.0, 3.0, 3),   # the reference's simple_receiver second stage: D = 8, L = 288 -> k_ols_wave<8>
    (384000.0, 48000.0, 43000.0, 3.0, 3),   # D = 8, L = 461: overlap 464 of k_ols_wave2k's 2048 (512 of k_ols_wave<8>'s 1024)
    (384000.0, 48000.0, 45500.0, 3.0, 3),   # D = 8, L = 922: k_ols_wave2k alone reaches it
    (1638400.0, 102400.0, 61440.0, 3.0, 3),   # D = 16, L = 240 -> k_ols_wg (a workgroup of four waves per block of 4096 samples)
    (1638400.0, 102400.0, 92160.0, 3.0, 3),   # D = 16, L = 960
    (3276800.0, 102400.0, 81920.0, 3.0, 3),   # D = 32, L = 960: blocks of 8192 samples, two runs per wave
    (6553600.0, 102400.0, 92160.0, 3.0, 3),   # D = 64, L = 3840: blocks of 16 384 samples, four runs per wave
    (6553600.0, 102400.0, 98304.0, 3.0, 0),   # D = 64, L = 9600: beyond half a block (and k_decim_poly's tile) -> k_fir
    (96000.0, 48000.0, 30000.0, 3.0, 3),    # D = 2, L = 32
    (96000.0, 48000.0, 44000.0, 3.0, 3),    # D = 2, L = 144 -> k_ols_wave<2>
    (96000.0, 48000.0, 46500.0, 3.0, 3),    # D = 2, L = 384
    # every other integer ratio and rational ratios with a short period: k_decim_poly (5), rr_decim.hip
    (1024000.0, 102400.0, 60000.0, 3.0, 5),   # examples/bandwidth_meter/main.rs:56: 10 : 1, L = 145
    (1024000.0, 384000.0, 200000.0, 3.0, 5),  # examples/simple_receiver.rs:28: 8 : 3, L = 34
    (300000.0, 100000.0, 60000.0, 3.0, 5),    # 3 : 1
    (48000.0, 32000.0, 20000.0, 2.0, 5),      # 3 : 2
    (700000.0, 300000.0, 100000.0, 3.5, 5),   # 7 : 3
    (2560000.0, 40000.0, 30000.0, 1.5, 3),    # 64 : 1, L = 768: k_ols_wg (k_decim_poly with tiles of 64 periods on request: the test below)
    (45000.0, 40000.0, 30000.0, 1.0, 5),      # 9 : 8, the longest period served
    (512000.0, 1000.0, 700.0, 2.0, 0),        # 512 : 1, L = 6827: beyond k_decim_poly's LDS -> k_fir, 8 outputs per workgroup, the taps in passes
    # every other pair of integer rates: the response at every position, the releasing ones stored - k_filter_wave<true> (10)
    (48000.0, 44100.0, 30000.0, 2.0, 10),     # 160 : 147, L = 14
    (48000.0, 44100.0, 40000.0, 3.0, 10),     # 160 : 147, L = 71
    (1024000.0, 44100.0, 20000.0, 3.0, 10),   # 10240 : 441, L = 255
    (220500.0, 48000.0, 40000.0, 2.0, 10),    # 147 : 32, L = 111
    (48000.0, 44100.5, 30000.0, 2.0, 10),     # rates on the 2^-1 grid: 96000 : 88201, closed form (the sample loop before)
    (1000.25, 333.125, 200.0, 2.0, 10),       # 2^-3 grid: 8002 : 2665
    (1024000.0, 44100.0, 30000.0, 3.0, 10),   # L = 436: beyond k_filter_wave's overlap -> the 4096-point blocks (k_filter_blk4096<.., SEL>)
    (1024000.0, 44100.0, 41000.0, 3.0, 10),   # L = 1982
    (1024000.0, 44100.0, 42000.0, 3.0, 0),    # L = 2926: beyond those too -> k_fir, the periodic schedule in closed form
]


@pytest.mark.parametrize("fin,fout,bw,q,kernel", FAST_CASES)
def test_downsampler_fast_paths(rr, oracle, fin, fout, bw, q, kernel):
    n = 150000
    x = oracle.synth_iq(11, 0, n)
    cuts = [0, 5000, 5003, 9099, 9100, 60000, 60001, 140001, n]
    g = rr.Downsampler.with_quality(1000, fout, bw, q)
    o64 = oracle.Downsampler(1000, fout, bw, q, flt=np.float64)
    o32 = oracle.Downsampler(1000, fout, bw, q, flt=np.float32)
    got, t64, t32, kernels = [], [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        y = g.process_raw(fin, x[a:b])
        kernels.append(g.last_kernel())
        r64, r32 = o64.process(fin, x[a:b]), o32.process(fin, x[a:b])
        assert len(y) == len(r64)
        got.append(y)
        t64.append(r64)
        t32.append(r32)
    assert kernels == [kernel, 0, kernel, 0, kernel, 0, kernel, kernel], (kernels, g.ir_len())
    check(np.concatenate(got), np.concatenate(t64), np.concatenate(t32))
    # every piece on its own too (a wrong history hand-over would hide in the overall RMS)
    for y, r in zip(got, t64):
        if len(y) > 50:
            check(y, r)


@pytest.mark.parametrize("fin,fout,bw,q", [(1024000.0, 102400.0, 60000.0, 3.0), (1024000.0, 384000.0, 200000.0, 3.0), (48000.0, 32000.0, 20000.0, 2.0),
                                          (45000.0, 40000.0, 30000.0, 1.0), (2560000.0, 40000.0, 30000.0, 1.5), (300000.0, 100000.0, 97000.0, 3.0),
                                          (500000.0, 100000.0, 10000.0, 1.0)])
def test_downsampler_polyphase_kernel_one_period_per_lane_on_request(rr, oracle, fin, fout, bw, q, monkeypatch):
    """RR_DECIM_PAIR=0: k_decim_poly with one period per lane (8-byte LDS reads, one product each) instead of two neighbouring
    periods per lane (16-byte reads, four products each): both against the oracle and against each other, ragged calls (the tap
    table is rebuilt when a call starts elsewhere in the period).  The last two cases: a long response at 3 : 1 (many columns per
    row) and one shorter than the period (rows without taps)."""
    n = 400000
    x = oracle.synth_iq(23, 0, n)
    ref = oracle.Downsampler(1000, fout, bw, q, flt=np.float64).process(fin, x)
    monkeypatch.setenv("RR_OLS_WG", "0")  # (64 : 1 would otherwise run k_ols_wg)
    outs = []
    for env in ("0", None):
        if env is None:
            monkeypatch.delenv("RR_DECIM_PAIR", raising=False)
        else:
            monkeypatch.setenv("RR_DECIM_PAIR", env)
        g = rr.Downsampler.with_quality(1000, fout, bw, q)
        y = np.concatenate([g.process_raw(fin, x[:150001]), g.process_raw(fin, x[150001:150001 + 99999]), g.process_raw(fin, x[250000:])])
        assert g.last_kernel() == 5, g.ir_len()
        check(y, ref)
        outs.append(y)
    assert rms_rel(outs[0], outs[1]) < 2e-6


@pytest.mark.parametrize("D,bwf", [(16, 0.6), (16, 0.9), (32, 0.8), (64, 0.6), (64, 0.9),
                                   # even ratios that are no powers of two (from 16 taps per period on): 3 .. 16 waves, the last one
                                   # with half of its lanes empty where D = 4 NW - 2
                                   (5, 0.9), (7, 0.9), (9, 0.9), (15, 0.9), (25, 0.9), (63, 0.9),  # odd ratios: the block staged sample by sample
                                   (6, 0.9), (10, 0.9), (12, 0.8), (14, 0.9), (20, 0.9), (22, 0.8), (24, 0.8), (34, 0.8), (48, 0.8), (62, 0.9)])
def test_downsampler_power_of_two_ratios_polyphase_decimator_on_request(rr, oracle, D, bwf, monkeypatch):
    """RR_OLS_WG=0: 16 / 32 / 64 : 1 through k_decim_poly (direct form) instead of k_ols_wg (overlap-save, a workgroup per block of
    256 D samples): the same results against the oracle, and the two kernels against each other; ragged calls."""
    fo = 102400.0
    fi, bw = fo * D, fo * bwf
    n = 700000
    x = oracle.synth_iq(29, 0, n)
    ref = oracle.Downsampler(1000, fo, bw, 3.0, flt=np.float64).process(fi, x)
    outs = []
    for env, want in (("0", 5), (None, 3)):
        if env is None:
            monkeypatch.delenv("RR_OLS_WG", raising=False)
        else:
            monkeypatch.setenv("RR_OLS_WG", env)
        g = rr.Downsampler.with_quality(1000, fo, bw, 3.0)
        y = np.concatenate([g.process_raw(fi, x[:300001]), g.process_raw(fi, x[300001:300001 + 131071]), g.process_raw(fi, x[431072:])])
        assert g.last_kernel() == want, (g.last_kernel(), g.ir_len())
        check(y, ref)
        outs.append(y)
    assert rms_rel(outs[0], outs[1]) < 2e-6


@pytest.mark.parametrize("bw", [30000.0, 40000.0, 43000.0])
def test_downsampler_8_to_1_blocks_of_1024_on_request(rr, oracle, bw, monkeypatch):
    """RR_OLSW_2K=0: 8 : 1 through k_ols_wave<8> (a wave per 1024-sample block) instead of k_ols_wave2k (2048-sample blocks): the
    same results against the oracle, and the two kernels against each other."""
    n = 200000
    x = oracle.synth_iq(19, 0, n)
    o64 = oracle.Downsampler(1000, 48000.0, bw, 3.0, flt=np.float64)
    ref = o64.process(384000.0, x)
    outs = []
    for env in ("0", None):
        if env is None:
            monkeypatch.delenv("RR_OLSW_2K", raising=False)
        else:
            monkeypatch.setenv("RR_OLSW_2K", env)
        g = rr.Downsampler.with_quality(1000, 48000.0, bw, 3.0)
        y = np.concatenate([g.process_raw(384000.0, x[:70001]), g.process_raw(384000.0, x[70001:])])
        assert g.last_kernel() == 3
        check(y, ref)
        outs.append(y)
    assert rms_rel(outs[0], outs[1]) < 2e-6


@pytest.mark.parametrize("fin,fout,bw,q", [(200e6, 50e6, 40e6, 3.0), (1024000.0, 102400.0, 60000.0, 3.0), (1024000.0, 384000.0, 200000.0, 3.0),
                                          (48000.0, 32000.0, 20000.0, 2.0)])
@pytest.mark.parametrize("poly", [False, True])
def test_downsampler_f64_polyphase_kernel(rr, oracle, monkeypatch, fin, fout, bw, q, poly):
    """Complex<f64>: integer ratios run k_ols4096_f64 (kernel 11; RR_DOWNSAMPLER_POLY=1: the decimator), every other periodic
    ratio the LDS tile fits k_decim_poly_f64 (kernel 5) in long calls, k_fir in short ones; the history is handed over between
    them."""
    if poly:
        monkeypatch.setenv("RR_DOWNSAMPLER_POLY", "1")
    want = 11 if (fin / fout).is_integer() and not poly else 5
    n = 90000
    x = oracle.synth_iq(17, 0, n).astype(np.complex128)
    g = rr.Downsampler.with_quality(1000, fout, bw, q, dtype=np.float64)
    o = oracle.Downsampler(1000, fout, bw, q, flt=np.float64)
    cuts = [0, 5000, 5003, 40000, 40001, n]
    kernels = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        y, r = g.process_raw(fin, x[a:b]), o.process(fin, x[a:b])
        kernels.append(g.last_kernel())
        assert len(y) == len(r)
        if len(y) > 8:
            assert rms_rel(y, r) <= 1e-12
    assert kernels == [want, 0, want, 0, want], kernels


@pytest.mark.parametrize("fin,fout,bw,q", [(200e6, 50e6, 40e6, 3.0), (1024000.0, 384000.0, 200000.0, 3.0), (48000.0, 32000.0, 20000.0, 2.0),
                                          (1024000.0, 102400.0, 60000.0, 3.0), (45000.0, 44999.0, 30000.0, 1.0), (2.0e9, 1.9e9, 1.0e9, 2.0),
                                          (200e6, 50e6, 47e6, 3.0), (2560000.0, 40000.0, 30000.0, 1.5), (48000.0, 32000.0, 31900.0, 1.0)])  # L = 400, 768, 960: the 4096-point blocks
def test_downsampler_select_kernel_on_request(rr, oracle, monkeypatch, fin, fout, bw, q):
    """RR_DOWNSAMPLER_SELECT=1: k_filter_wave<true> for every pair of integer rates it takes - the ratios the other kernels
    serve by default, a period of 45 000 inputs (almost every position releases) and rates close to the kernel's 2^31 limit;
    ragged calls, a short one on k_fir in between."""
    monkeypatch.setenv("RR_DOWNSAMPLER_SELECT", "1")
    n = 200000
    x = oracle.synth_iq(12, 0, n)
    cuts = [0, 4096, 4100, 9099, 60000, 60001, 140001, n]
    g = rr.Downsampler.with_quality(1000, fout, bw, q)
    o64 = oracle.Downsampler(1000, fout, bw, q, flt=np.float64)
    kernels = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        y, r = g.process_raw(fin, x[a:b]), o64.process(fin, x[a:b])
        kernels.append(g.last_kernel())
        assert len(y) == len(r)
        if len(y) > 50:
            check(y, r)
    assert kernels == [10, 0, 10, 10, 0, 10, 10], (kernels, g.ir_len())


def test_downsampler_fast_path_can_be_switched_off(rr, oracle, monkeypatch):
    monkeypatch.setenv("RR_DOWNSAMPLER_GENERIC", "1")
    g = rr.Downsampler.new(1000, 50e6, 40e6)
    x = oracle.synth_iq(12, 0, 20000)
    y = g.process_raw(200e6, x)
    assert g.last_kernel() == 0
    check(y, oracle.Downsampler(1000, 50e6, 40e6, flt=np.float64).process(200e6, x))


@pytest.mark.parametrize("fin,fout,bw", [(200e6, 50e6, 30e6), (96000.0, 48000.0, 30000.0), (384000.0, 48000.0, 30000.0)])
def test_downsampler_direct_form_on_request(rr, oracle, monkeypatch, fin, fout, bw):
    """k_mix_fir_decim (the direct form, D = 2 / 4 / 8) is no default any more; RR_FUSED_KERNEL=direct keeps it reachable."""
    monkeypatch.setenv("RR_FUSED_KERNEL", "direct")
    g = rr.Downsampler.new(1000, fout, bw)
    x = oracle.synth_iq(13, 0, 70000)
    r = oracle.Downsampler(1000, fout, bw, flt=np.float64)
    for a, b in ((0, 30000), (30000, 30007), (30007, 70000)):
        y = g.process_raw(fin, x[a:b])
        assert g.last_kernel() == (1 if b - a >= 4096 else 0)
        want = r.process(fin, x[a:b])
        assert len(y) == len(want)
        if len(y) > 8:
            check(y, want)


def test_downsampler_output_chunks_and_events(rr, oracle):
    g = rr.Downsampler.new(100, 50e6, 40e6)
    x = oracle.synth_iq(9, 0, 1200)
    out = g.process(rr.Samples(200e6, x[:1000]))
    assert [len(s.chunk) for s in out] == [100, 100] and all(s.sample_rate == 50e6 for s in out)
    ev = rr.EventSignal(rr.Disconnection())
    assert g.process(ev) == [ev]  # no reset (resampling.rs:135-137)
    out2 = g.process(rr.Samples(200e6, x[1000:]))
    assert [len(s.chunk) for s in out2] == [100]
    o = oracle.Downsampler(100, 50e6, 40e6, flt=np.float64)
    ref = o.feed(200e6, x[:1000]) + o.feed(200e6, x[1000:])
    check(np.concatenate([s.chunk for s in out + out2]), np.concatenate(ref))


def test_downsampler_contract(rr):
    from radiorust_amd._lib import ContractViolation

    with pytest.raises(ContractViolation):
        rr.Downsampler.new(16, 48000.0, 48000.0)
    with pytest.raises(ContractViolation):
        rr.Downsampler.new(16, -1.0, -2.0)
    g = rr.Downsampler.new(16, 48000.0, 40000.0)
    with pytest.raises(ContractViolation):
        g.process_raw(44100.0, np.zeros(8, dtype=np.complex64))


def test_downsampler_rate_change_resets(rr, oracle):
    g = rr.Downsampler.new(16, 48000.0, 40000.0)
    o = oracle.Downsampler(16, 48000.0, 40000.0, flt=np.float64)
    x = oracle.synth_iq(10, 0, 4000)
    for rate, a, b in ((96000.0, 0, 1500), (192000.0, 1500, 3000), (96000.0, 3000, 4000)):
        check(g.process_raw(rate, x[a:b]), o.process(rate, x[a:b]))


@pytest.mark.parametrize("fin,fout,bw,q", [(48000.0, 44100.0, 40000.0, 3.0), (48000.0, 44100.5, 30000.0, 2.0), (48000.0, 44100.1, 30000.0, 2.0)])
def test_downsampler_f64_long_periods(rr, oracle, fin, fout, bw, q):
    """Complex<f64> beyond k_decim_poly_f64's 8 phases: k_fir with the periodic schedule in closed form (160 : 147, 96000 : 88201
    on the 2^-1 grid) and with the emission list (44100.1: off every grid the f64 sums stay exact on); ragged calls."""
    n = 60000
    x = oracle.synth_iq(18, 0, n).astype(np.complex128)
    g = rr.Downsampler.with_quality(1000, fout, bw, q, dtype=np.float64)
    o = oracle.Downsampler(1000, fout, bw, q, flt=np.float64)
    cuts = [0, 5000, 5003, 40000, 40001, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        y, r = g.process_raw(fin, x[a:b]), o.process(fin, x[a:b])
        assert len(y) == len(r) and g.last_kernel() == 0
        if len(y) > 8:
            assert rms_rel(y, r) <= 1e-12


def test_downsampler_f64(rr, oracle):
    x = oracle.synth_iq(11, 0, 20000).astype(np.complex128)
    g = rr.Downsampler.new(16, 50e6, 40e6, dtype=np.float64)
    o = oracle.Downsampler(16, 50e6, 40e6, flt=np.float64)
    assert rms_rel(g.process_raw(200e6, x), o.process(200e6, x)) < 1e-13


# ---------------------------------------------------------------- Fourier
def test_fourier_reference_kat_on_gpu(rr):
    """analysis.rs:139-209 run on the device in f64 at the reference's 1e-10."""
    f1 = rr.Fourier.new(dtype=np.float64)
    f2 = rr.Fourier.new_center_dc(dtype=np.float64)
    for chunk, want1, want2 in [([1.0, 1.0, 1.0], [3, 0, 0], [0, 3, 0]),
                                ([1.0, 1.5, 1.0, 0.5], [4, -1j, 0, 1j], [0, 1j, 4, -1j])]:
        sig = rr.Samples(48000.0, np.array(chunk, dtype=np.complex128))
        (o1,), (o2,) = f1.process(sig), f2.process(sig)
        for got, want in ((o1.chunk, want1), (o2.chunk, want2)):
            for gv, wv in zip(got, want):
                assert_approx(gv.real, complex(wv).real)
                assert_approx(gv.imag, complex(wv).imag)
    g1, g2 = rr.Fourier.new(), rr.Fourier.new_center_dc()  # and in f32
    sig = rr.Samples(48000.0, np.array([1.0, 1.5, 1.0, 0.5], dtype=np.complex64))
    np.testing.assert_allclose(g1.process(sig)[0].chunk, [4, -1j, 0, 1j], atol=1e-6)
    np.testing.assert_allclose(g2.process(sig)[0].chunk, [0, 1j, 4, -1j], atol=1e-6)


@pytest.mark.parametrize("n,center", [(4096, False), (4096, True), (8192, False), (8192, True), (256, True), (256, False), (1024, False),
                                      (1024, True), (2048, False), (2048, True), (2, False), (1, True), (1000, True), (7, True), (4095, False), (12000, False),
                                      (16384, True), (65536, False), (65536, True), (1 << 18, False),   # four-step through HBM
                                      (1 << 15, False), (1 << 17, True), (1 << 19, False), (1 << 20, True),  # (two passes, k_fft_tile, up to 512 x 512; the transposes beyond)
                                      (1 << 21, False),                                                  # (beyond the tile kernel: transposes around the row kernels)
                                      (20000, False), (20000, True), (5000, False), (31, True),          # Bluestein beyond 4096 points; direct below 32
                                      (513, False), (1025, True), (1999, True), (2047, False),           # 513 .. 2048: Bluestein in one kernel (k_bluestein4096)
                                      (2049, False), (3001, True), (4093, False), (4001, True),          # 2049 .. 4096 with a prime factor beyond 13: k_bluestein_big<8192>
                                      (4099, True), (5003, False), (6007, True), (8191, False), (8191, True),  # 4097 .. 8192 likewise: k_bluestein_big<16384>
                                      (33, False), (100, True), (255, False), (300, True), (511, True)])  # 32 .. 512: a wave per chunk (k_bluestein1024)
def test_fourier_parity(rr, oracle, n, center):
    x = oracle.synth_iq(12, 0, n)
    gw, ow = rr.Kaiser.with_null_at_bin(2.0), oracle.Kaiser.with_null_at_bin(2.0)
    g = rr.Fourier(gw, center)
    (out,) = g.process(rr.Samples(1e6, x))
    t64 = oracle.Fourier(ow, center, flt=np.float64).process(x)
    t32 = oracle.Fourier(ow, center, flt=np.float32).process(x)
    check(out.chunk, t64, t32)


@pytest.mark.parametrize("n,center", [(1000, True), (20000, False), (8192, False), (32768, True), (65536, False), (33, False),
                                      (1 << 14, True), (1 << 17, False), (1 << 20, True),
                                      (1999, True), (2039, False), (67, True), (2048 - 1, False), (2053, True)])  # k_bluestein_lds up to 2048 points; five launches beyond
def test_fourier_f64_lengths(rr, oracle, n, center):
    """Complex<f64>: powers of two beyond the LDS kernel (4096) by the four-step transform, every other length >= 32 by
    Bluestein over f64 power-of-two transforms (the twiddles of the four-step split are evaluated in f64 with the
    phase reduced exactly)."""
    x = oracle.synth_iq(13, 0, n).astype(np.complex128)
    g = rr.Fourier(rr.Kaiser.with_null_at_bin(2.0), center, dtype=np.float64)
    (out,) = g.process(rr.Samples(1e6, x))
    ref = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), center, flt=np.float64).process(x)
    assert rms_rel(out.chunk, ref) < 1e-11


@pytest.mark.parametrize("center", [False, True])
def test_fourier_f64_4096_register_kernel(rr, oracle, center, monkeypatch):
    """Complex<f64>, 4096 points: k_fft4096_f64 (the lane's 16 values in registers, radix 16 x 16 x 16) against the f64 oracle
    at 1e-12, many chunks in one call, overlapping frames through the Stft (hop 1024) as well, and against the Stockham
    kernel it replaces (RR_FOURIER_GENERIC=1)."""
    n = 4096
    x = oracle.synth_iq(21, 0, 9 * n).astype(np.complex128)
    g = rr.Fourier(rr.Kaiser.with_null_at_bin(2.0), center, dtype=np.float64)
    of = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), center, flt=np.float64)
    import torch

    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty(9 * n, dtype=torch.complex128, device="cuda")
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    assert g.process_dev(n, d_in.data_ptr(), 9 * n, d_out.data_ptr(), 9 * n) == 9 * n
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for k in range(9):
        assert rms_rel(got[k * n:(k + 1) * n], of.process(x[k * n:(k + 1) * n])) < 1e-12
    st = rr.Stft(1024, 4, rr.Kaiser.with_null_at_bin(2.0), center_dc=center, dtype=np.float64)
    out = st.process(rr.Samples(1e6, x))
    assert len(out) == 33
    for k in (0, 1, 17, 32):
        assert rms_rel(out[k].chunk, of.process(x[k * 1024:k * 1024 + 4096])) < 1e-12


@pytest.mark.parametrize("n,center", [(6, False), (12, True), (60, False), (96, True), (360, False), (1536, True), (2000, False), (3000, True),
                                      (3072, False), (3125, True), (3840, False), (4000, True), (4050, False), (4095, True),
                                      (4800, False), (6000, True), (7776, False), (8000, True),   # (beyond 4096: f32 only, Bluestein in f64)
                                      (77, False), (343, True), (1001, False), (1331, True), (2002, False), (2401, True), (4004, False),
                                      (6006, True), (8008, False)])   # radix 7 / 11 / 13 passes
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fourier_mixed_radix_lengths(rr, oracle, monkeypatch, n, center, dtype):
    """Chunk lengths 2^a 3^b 5^c that are not powers of two run k_fft_mixed (radix 5 / 4 / 3 / 2 passes in one LDS image); every
    other length keeps Bluestein (4095 = 3^2 5 7 13 here).  Several chunks per call, against the oracle and against the Bluestein
    form of the same handle type (RR_FOURIER_MIXED=0)."""
    import torch

    chunks = 7
    x = oracle.synth_iq(23, 0, n * chunks).astype(np.complex128 if dtype == np.float64 else np.complex64)
    o = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), center, flt=np.float64)
    ref = np.concatenate([o.process(x[i * n:(i + 1) * n]) for i in range(chunks)])
    outs = []
    for mixed in ("2", "0"):  # 2: wherever it applies (by default only where it was measured faster than Bluestein's kernels)
        monkeypatch.setenv("RR_FOURIER_MIXED", mixed)
        g = rr.Fourier(rr.Kaiser.with_null_at_bin(2.0), center, dtype=dtype)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        assert g.process_dev(n, d_in.data_ptr(), n * chunks, d_out.data_ptr(), n * chunks) == n * chunks
        torch.cuda.synchronize()
        outs.append(d_out.cpu().numpy())
    tol = 1e-11 if dtype == np.float64 else 2e-6
    assert rms_rel(outs[0], ref) < tol, rms_rel(outs[0], ref)
    assert rms_rel(outs[1], ref) < tol, rms_rel(outs[1], ref)


@pytest.mark.parametrize("n,center", [(20000, True), (48000, False), (10000, True), (30375, True), (12000, False), (100000, True),
                                      (8640, False), (262144 // 2 * 2 - 12144, False), (5000, True),
                                      (77000, True), (91091, False)])   # 275 x 280 with radix 7 / 11 passes; 7^2 11 13^2 has no split up to 512: Bluestein
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fourier_mixed_radix_two_pass(rr, oracle, monkeypatch, n, center, dtype):
    """Lengths 2^a 3^b 5^c beyond one LDS image (8192 points in f32, 4096 in f64) up to 512 x 512: two passes, k_fft_tilem - bundles
    that are partly empty (N1 or N2 not a multiple of 16 / 8), odd lengths (30375 = 3^5 5^3: center_dc rotates by n // 2), several
    chunks per call; against the oracle and against Bluestein (RR_FOURIER_MIXED=0)."""
    import torch

    from oracle import oracle_np as onp

    chunks = 3
    x = oracle.synth_iq(29, 0, n * chunks).astype(np.complex128 if dtype == np.float64 else np.complex64)
    # the C oracle's transform of a length that is not a power of two is the O(n^2) sum: the reference here is its window
    # (analysis.rs:88-101, oracle's own Kaiser) and numpy's f64 transform - the formulation tests/test_oracle_crosscheck.py
    # checks the oracle against
    win = oracle.Kaiser.with_null_at_bin(2.0)
    w = onp.fourier_window(n, np.array([win.relative_value_at(p) for p in onp.window_positions(n)]))
    ref = np.concatenate([onp.fourier(x[i * n:(i + 1) * n], w, center) for i in range(chunks)])
    outs = []
    for mixed in ("1", "0"):
        monkeypatch.setenv("RR_FOURIER_MIXED", mixed)
        g = rr.Fourier(rr.Kaiser.with_null_at_bin(2.0), center, dtype=dtype)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        assert g.process_dev(n, d_in.data_ptr(), n * chunks, d_out.data_ptr(), n * chunks) == n * chunks
        torch.cuda.synchronize()
        outs.append(d_out.cpu().numpy())
    tol = 1e-11 if dtype == np.float64 else 2e-6
    assert rms_rel(outs[0], ref) < tol, rms_rel(outs[0], ref)
    assert rms_rel(outs[1], ref) < tol, rms_rel(outs[1], ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fourier_big_two_pass_against_the_transposes(rr, oracle, monkeypatch, dtype):
    """Powers of two beyond one LDS tile, several chunks per device call: the two-pass form (k_fft_tile, the default up to
    2^18 points) against the five-launch form (RR_FOURIER_BIG=transpose) and the oracle."""
    import torch

    monkeypatch.setenv("RR_FOURIER_16K", "0")  # (16 384 points in f32 otherwise run k_fft16384, one pass)
    for n, chunks, center in ((1 << 14, 5, False), (1 << 15, 3, True), (1 << 16, 3, True)):
        x = oracle.synth_iq(21, 0, n * chunks).astype(np.complex128 if dtype == np.float64 else np.complex64)
        o = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), center, flt=np.float64)
        ref = np.concatenate([o.process(x[i * n:(i + 1) * n]) for i in range(chunks)])
        outs = []
        for form in ("", "transpose"):
            monkeypatch.setenv("RR_FOURIER_BIG", form)
            g = rr.Fourier(rr.Kaiser.with_null_at_bin(2.0), center, dtype=dtype)
            d_in = torch.from_numpy(x).cuda()
            d_out = torch.empty_like(d_in)
            assert g.process_dev(n, d_in.data_ptr(), n * chunks, d_out.data_ptr(), n * chunks) == n * chunks
            torch.cuda.synchronize()
            outs.append(d_out.cpu().numpy())
        tol = 1e-11 if dtype == np.float64 else 2e-6
        assert rms_rel(outs[0], ref) < tol and rms_rel(outs[1], ref) < tol
        assert rms_rel(outs[0], outs[1]) < tol


def test_fourier_bluestein_one_kernel_many_chunks(rr, oracle, monkeypatch):
    """n = 1000 and 1536 (k_bluestein4096), 300 and 97 (k_bluestein1024): 40 chunks in one device call, against the oracle
    and against the five-launch form (RR_FOURIER_GENERIC=1)."""
    import torch

    for n, center in ((1000, False), (1536, True), (300, True), (97, False)):
        x = oracle.synth_iq(14, 0, n * 40)
        o = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), center, flt=np.float64)
        ref = np.concatenate([o.process(x[i * n:(i + 1) * n]) for i in range(40)])
        outs = []
        for generic in ("0", "1"):
            monkeypatch.setenv("RR_FOURIER_GENERIC", generic)
            g = rr.Fourier(rr.Kaiser.with_null_at_bin(2.0), center)
            d_in = torch.from_numpy(x).cuda()
            d_out = torch.empty_like(d_in)
            g.set_stream(torch.cuda.current_stream().cuda_stream)
            assert g.process_dev(n, d_in.data_ptr(), n * 40, d_out.data_ptr(), n * 40) == n * 40
            torch.cuda.synchronize()
            outs.append(d_out.cpu().numpy())
            check(outs[-1], ref)
        assert rms_rel(outs[0], outs[1]) < 2e-6


def test_fourier_custom_window_and_length_change(rr, oracle):
    fn = lambda v: 1 - 0.8 * v * v  # noqa: E731
    g = rr.Fourier.with_window(rr.CustomWindow(fn))
    o = oracle.Fourier(oracle.CustomWindow(fn), flt=np.float64)
    for n in (512, 300, 512):
        x = oracle.synth_iq(13, n, n)
        check(g.process(rr.Samples(1.0, x))[0].chunk, o.process(x))
    ev = rr.EventSignal(rr.Disconnection())
    assert g.process(ev) == [ev]


def test_fourier_batched_device_api(rr, oracle):
    import torch

    n, k = 4096, 9
    x = oracle.synth_iq(14, 0, n * k)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    g = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    assert g.process_dev(n, d_in.data_ptr(), n * k, d_out.data_ptr(), n * k) == n * k
    torch.cuda.synchronize()
    o = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), flt=np.float64)
    ref = np.concatenate([o.process(x[i * n : (i + 1) * n]) for i in range(k)])
    check(d_out.cpu().numpy(), ref)


def test_fourier_8192_register_kernel_on_request(rr, oracle, monkeypatch):
    """RR_FOURIER_8K=regs: k_fft8192 (256 lanes with 32 values each), the default before k_fft_big<8192>."""
    import torch

    monkeypatch.setenv("RR_FOURIER_8K", "regs")
    n, k = 8192, 9
    x = oracle.synth_iq(16, 0, n * k)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    for center in (False, True):
        g = rr.Fourier(rr.Kaiser.with_null_at_bin(2.0), center)
        assert g.process_dev(n, d_in.data_ptr(), n * k, d_out.data_ptr(), n * k) == n * k
        torch.cuda.synchronize()
        o64 = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), center, flt=np.float64)
        ref = np.concatenate([o64.process(x[i * n:(i + 1) * n]) for i in range(k)])
        check(d_out.cpu().numpy(), ref)


@pytest.mark.parametrize("n,center", [(1024, False), (1024, True), (256, False), (2048, False), (2048, True), (512, False), (512, True), (8192, True), (64, False), (64, True), (128, False), (128, True),
                                      (16384, False), (16384, True)])
def test_fourier_wave_kernels_batched(rr, oracle, n, center, monkeypatch):
    """Chunks of 512 / 1024 (k_fft512 / k_fft1024: a wave per chunk), 2048 (k_fft2048: 128 lanes per chunk), 16 384 (k_fft16384: a
    workgroup of 1024 lanes per chunk, one pass over HBM) and 256 (the channelizer's one-branch case) in f32: many chunks per call on the device, every chunk against the f64 oracle."""
    import torch

    k = 301 if n < 8192 else 40  # (301: the last wave of the 64- / 128-point kernels is partly empty)
    x = oracle.synth_iq(15, 0, n * k)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    win, owin = rr.Kaiser.with_null_at_bin(2.0), oracle.Kaiser.with_null_at_bin(2.0)
    g = rr.Fourier(win, center)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    assert g.process_dev(n, d_in.data_ptr(), n * k, d_out.data_ptr(), n * k) == n * k
    torch.cuda.synchronize()
    o64 = oracle.Fourier(owin, center, flt=np.float64)
    o32 = oracle.Fourier(owin, center, flt=np.float32)
    got = d_out.cpu().numpy().reshape(k, n)
    for i in (0, 1, 127, 128, 129, 255, 299, 300) if k == 301 else (0, 1, 17, 39):
        check(got[i], o64.process(x[i * n : (i + 1) * n]), o32.process(x[i * n : (i + 1) * n]))
    ref = np.concatenate([o64.process(x[i * n : (i + 1) * n]) for i in range(k)])
    check(got.reshape(-1), ref)


@pytest.mark.parametrize("n,center", [(2049, True), (3001, False), (4093, True), (4095, False)])
def test_fourier_bluestein_8192_register_transform_on_request(rr, oracle, n, center, monkeypatch):
    """RR_FOURIER_BS8K=regs: 2049 .. 4096 points around the two 8192-point register transforms of k_bluestein8192 (the default is
    k_bluestein_big<8192>, the workgroup transforms of rr_fft_big.hpp): same results against the oracle."""
    monkeypatch.setenv("RR_FOURIER_BS8K", "regs")
    monkeypatch.setenv("RR_FOURIER_MIXED", "0")
    assert rr.fourier_route(n) == "bluestein one kernel M=8192"
    k = 21
    x = oracle.synth_iq(31, 0, n * k)
    gw, ow = rr.Kaiser.with_null_at_bin(2.0), oracle.Kaiser.with_null_at_bin(2.0)
    g = rr.Fourier(gw, center)
    o64 = oracle.Fourier(ow, center, flt=np.float64)
    o32 = oracle.Fourier(ow, center, flt=np.float32)
    for i in (0, 7, k - 1):
        (out,) = g.process(rr.Samples(1e6, x[i * n : (i + 1) * n]))
        check(out.chunk, o64.process(x[i * n : (i + 1) * n]), o32.process(x[i * n : (i + 1) * n]))


@pytest.mark.parametrize("n,center,k", [(1000, True, 2100), (33, False, 500), (1500, False, 64), (4095, True, 9), (100, True, 300),
                                        (5003, True, 300), (8191, False, 33)])
def test_fourier_bluestein_batched(rr, oracle, n, center, k):
    """Chunk lengths that are not powers of two (32 and more) run Bluestein's algorithm over the power-of-two
    kernels, in passes of at most 2^22 workspace elements (n = 1000: 2048 chunks per pass, so 2100 chunks take
    two): chunks from the start, the pass boundary and the end against the f64 oracle."""
    import torch

    x = oracle.synth_iq(16, 0, n * k)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    win, owin = rr.Kaiser.with_null_at_bin(2.0), oracle.Kaiser.with_null_at_bin(2.0)
    g = rr.Fourier(win, center)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    assert g.process_dev(n, d_in.data_ptr(), n * k, d_out.data_ptr(), n * k) == n * k
    torch.cuda.synchronize()
    o64 = oracle.Fourier(owin, center, flt=np.float64)
    got = d_out.cpu().numpy().reshape(k, n)
    for i in sorted({0, 1, k // 2, min(2047, k - 1), min(2048, k - 1), k - 1}):
        check(got[i], o64.process(x[i * n : (i + 1) * n]))


def test_fourier_errors(rr):
    from radiorust_amd._lib import BackendError

    g = rr.Fourier.new()
    with pytest.raises(BackendError):
        g.process(rr.Samples(1.0, np.zeros(0, dtype=np.complex64)))
    with pytest.raises(BackendError):
        g.process(rr.Samples(1.0, np.zeros((1 << 23) + 1, dtype=np.complex64)))  # beyond 2^23 points: says so


# ---------------------------------------------------------------- async + pinned
def test_enqueue_wait_with_pinned_buffers(rr, oracle):
    L = rr._lib.lib()
    n = 1 << 16
    p_in, p_out = C.c_void_p(), C.c_void_p()
    assert L.rr_host_alloc(n * 8, C.byref(p_in)) == 0 and L.rr_host_alloc(n * 8, C.byref(p_out)) == 0
    x = oracle.synth_iq(15, 0, n)
    C.memmove(p_in, x.ctypes.data, n * 8)
    g = rr.FreqShifter.with_shift(700.0)
    cnt = C.c_size_t()
    assert L.rr_freqshifter_enqueue(g._h, 48000.0, p_in, n, p_out, n, C.byref(cnt)) == 0
    assert cnt.value == n  # final at enqueue time
    g.wait()
    assert g.query()
    y = np.frombuffer((C.c_char * (n * 8)).from_address(p_out.value), dtype=np.complex64).copy()
    assert rms_rel(y, oracle.FreqShifter(1.0, 700.0, flt=np.float32).process(48000.0, x)) < 1e-7
    # registering an existing allocation (a pooled Vec) works too
    buf = np.zeros(n, dtype=np.complex64)
    assert L.rr_host_register(buf.ctypes.data, buf.nbytes) == 0
    assert L.rr_host_unregister(buf.ctypes.data) == 0
    assert L.rr_host_free(p_in) == 0 and L.rr_host_free(p_out) == 0


# ---------------------------------------------------------------- Filter, half-precision points (cfg5)
@pytest.mark.parametrize("resp16", [False, True])
def test_filter_f16_output(rr, oracle, resp16):
    """rr_filter_process_dev_f16: the 1024-tap overlap-save Filter with IEEE-half output pairs (and,
    second case, a half-precision response table).  Tolerances are the formats' own: half has an
    11-bit significand, so rounding the output alone gives a relative RMS error of about
    2^-11 / sqrt(3) = 2.8e-4; rounding the response adds about as much again."""
    import torch

    n, N, fs = 1024, 1 << 16, 2e9
    x = oracle.synth_iq(21, 0, N)
    f = rr.Filter.new(lowpass(200e6))
    st = torch.cuda.current_stream().cuda_stream
    f.set_stream(st)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros(2 * N, dtype=torch.float16, device="cuda")
    got = f.process_dev_f16(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N, response_f16=resp16)
    torch.cuda.synchronize()
    assert got == N - n  # first chunk swallowed
    y = d_out[: 2 * got].cpu().numpy().astype(np.float64)
    y = y[0::2] + 1j * y[1::2]
    o = oracle.Filter(lowpass(200e6), flt=np.float64)
    ref = [o.process(fs, x[a:a + n].astype(np.complex128)) for a in range(0, N, n)]
    ref = np.concatenate([r for r in ref if r is not None])
    assert len(ref) == got
    err = rms_rel(y, ref)
    assert 5e-5 < err < (8e-4 if resp16 else 4e-4), err


@pytest.mark.parametrize("n", [1024, 256, 2048])
def test_filter_block4096_ragged_calls(rr, oracle, n, monkeypatch):
    """The 4096-point block kernel of the long Filters (rr_filter_ols.hip; BASELINE configs[4]: n = 1024 at
    2 GS/s) on one stream cut into calls of different sizes (first chunk swallowed, calls shorter than a block,
    blocks reaching into the history and past the input, the history handed from call to call by the kernel
    itself), against the chunk-by-chunk oracle; nothing is written past the produced samples."""
    import torch

    fs = 2e9
    ks = [1, 2, 37, 1, 5, 64, 3]  # chunks per call
    x = oracle.synth_iq(33, 0, n * sum(ks))
    o64 = oracle.Filter(lowpass(200e6), flt=np.float64)
    o32 = oracle.Filter(lowpass(200e6), flt=np.float32)
    r64 = np.concatenate([o64.process(fs, x[i * n : (i + 1) * n]) for i in range(sum(ks))][1:])
    r32 = np.concatenate([o32.process(fs, x[i * n : (i + 1) * n]) for i in range(sum(ks))][1:])
    d_in = torch.from_numpy(x).cuda()
    monkeypatch.setenv("RR_FILTER_KERNEL", "ols4096")  # (n = 256 would otherwise take k_filter_wave for the long calls)
    g = rr.Filter.new(lowpass(200e6))
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    d_out = torch.zeros_like(d_in)
    off = wrote = 0
    for k in ks:
        made = g.process_dev(fs, n, d_in.data_ptr() + 8 * off, n * k, d_out.data_ptr() + 8 * wrote, n * k)
        assert g.last_kernel() == (2 if made else 0)
        wrote += made
        off += n * k
    torch.cuda.synchronize()
    assert wrote == n * (sum(ks) - 1)
    got = d_out.cpu().numpy()
    assert not np.any(got[wrote:]), "wrote past the produced samples"
    check(got[:wrote], r64, r32)
    # the seams between calls (history written by the previous call's kernel)
    edge = 0
    for k in ks[1:-1]:
        edge += n * k
        check(got[edge - n : edge + n], r64[edge - n : edge + n])


def test_filter_f16_unsupported(rr):
    import torch
    from radiorust_amd._lib import BackendError

    f = rr.Filter.new(lowpass(10e6))
    d = torch.zeros(128, dtype=torch.complex64, device="cuda")
    o = torch.zeros(256, dtype=torch.float16, device="cuda")
    with pytest.raises(BackendError):
        f.process_dev_f16(200e6, 64, d.data_ptr(), 128, o.data_ptr(), 128)  # 64 taps: direct-form kernel
