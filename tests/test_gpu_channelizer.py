"""GPU parity of the polyphase FFT channelizer (BASELINE configs[2]: 256 bins, 4 taps
per branch) against the reference composition it is defined by:
Rechunker(M) -> Overlapper(P) -> Fourier::with_window(Kaiser null-at-bin P) over P*M
samples -> every P-th bin (chunks.rs:42-242, analysis.rs:60-132), built from the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def reference_frames(oracle, chunks, M, P, flt, history=None):
    """Overlapper(P) + Fourier + bin selection, chunk by chunk."""
    hist = [] if history is None else history
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(float(P)), flt=flt)
    out = []
    for c in chunks:
        hist.append(c)
        if len(hist) >= P:
            X = fo.process(np.concatenate(hist[-P:]))
            out.append(X[::P])
            del hist[: len(hist) - (P - 1)]
    return out, hist


@pytest.mark.parametrize("M,P,dtype,tol", [(256, 4, np.float32, 1e-5), (64, 3, np.float32, 1e-5), (256, 1, np.float32, 1e-5),
                                           (128, 4, np.float64, 1e-12), (256, 2, np.float32, 1e-5), (256, 3, np.float32, 1e-5),
                                           (256, 6, np.float32, 1e-5), (256, 8, np.float32, 1e-5), (256, 5, np.float32, 1e-5),
                                           (256, 4, np.float64, 1e-12),
                                           # 512 / 1024 / 2048 / 4096 bins: the Fourier kernels with the fold at the load (k_fft512<true> .. k_fft4096<true>)
                                           (1024, 8, np.float32, 1e-5), (1024, 3, np.float32, 1e-5), (1024, 1, np.float32, 1e-5),
                                           (4096, 2, np.float32, 1e-5), (4096, 5, np.float32, 1e-5), (512, 4, np.float32, 1e-5), (512, 1, np.float32, 1e-5),
                                           (2048, 3, np.float32, 1e-5), (2048, 8, np.float32, 1e-5)])
def test_channelizer_parity(rr, oracle, M, P, dtype, tol):
    nchunks = 37
    x = oracle.synth_iq(21, 0, M * nchunks)
    if dtype == np.float64:
        x = x.astype(np.complex128)
    chunks = [x[i * M : (i + 1) * M] for i in range(nchunks)]
    ref, _ = reference_frames(oracle, chunks, M, P, np.float64)
    g = rr.Channelizer(M, P, dtype=dtype)
    got = []
    # ragged multiples of M per call, including one call that is too short to emit anything
    cuts = [0, 1, 2, 10, 11, 30, nchunks]
    for a, b in zip(cuts[:-1], cuts[1:]):
        got += g.process(rr.Samples(1e9, x[a * M : b * M]))
    assert len(got) == len(ref) == nchunks - (P - 1)
    assert all(len(s.chunk) == M for s in got)
    for a, b in zip(got, ref):
        assert rms_rel(a.chunk, b) <= tol


@pytest.mark.parametrize("M,P,hop,dtype,tol", [(256, 4, 128, np.float32, 1e-5), (256, 4, 64, np.float32, 1e-5),
                                               (256, 8, 128, np.float32, 1e-5), (256, 2, 64, np.float32, 1e-5), (256, 3, 128, np.float32, 1e-5),
                                               (64, 3, 32, np.float64, 1e-12), (1024, 2, 512, np.float32, 1e-5),
                                               (100, 4, 100, np.float32, 1e-5), (100, 4, 50, np.float32, 1e-5),
                                               (1000, 2, 250, np.float64, 1e-11), (16384, 2, 8192, np.float32, 1e-5),
                                               # bin counts 2^a 3^b 5^c: fold + mixed-radix transform in one kernel (k_fft_mixed)
                                               (300, 3, 100, np.float32, 1e-5), (1500, 2, 1500, np.float32, 1e-5),
                                               (3000, 2, 1000, np.float64, 1e-11), (6000, 2, 3000, np.float32, 1e-5),
                                               (77, 2, 77, np.float32, 1e-5)])   # (7 x 11: the fold to a workspace + Bluestein)
def test_channelizer_oversampled_and_any_bins(rr, oracle, M, P, hop, dtype, tol):
    """The general form (rr_channelizer_create_ex): `hop` < bins samples between frames - the oversampled
    filterbank - and bin counts that are not powers of two, against the composition it is defined by:
    Rechunker(hop) -> Overlapper(P M / hop) -> Fourier::with_window(Kaiser null-at-bin P) over P M samples ->
    every P-th bin (chunks.rs:194-242, analysis.rs:60-132)."""
    K = P * M // hop
    nchunks = 3 * K + 5
    x = oracle.synth_iq(26, 0, hop * nchunks)
    if dtype == np.float64:
        x = x.astype(np.complex128)
    chunks = [x[i * hop : (i + 1) * hop] for i in range(nchunks)]
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(float(P)), flt=np.float64)
    ref = [fo.process(np.concatenate(chunks[i : i + K]))[::P] for i in range(nchunks - K + 1)]
    g = rr.Channelizer(M, P, dtype=dtype, hop=hop)
    got = []
    cuts = [0, 1, 2, K + 1, K + 2, 2 * K + 3, nchunks]  # ragged multiples of the hop per call
    for a, b in zip(cuts[:-1], cuts[1:]):
        got += g.process(rr.Samples(1e9, x[a * hop : b * hop]))
    assert len(got) == len(ref) and all(len(s.chunk) == M for s in got)
    for a, b in zip(got, ref):
        assert rms_rel(a.chunk, b) <= tol


def test_channelizer_event_resets_history(rr, oracle):
    M, P = 256, 4
    x = oracle.synth_iq(22, 0, M * 12)
    g = rr.Channelizer(M, P)
    out1 = g.process(rr.Samples(1e9, x[: M * 6]))
    assert len(out1) == 3
    ev = rr.EventSignal(rr.Event())
    res = g.process(ev)
    assert len(res) == 2 and res[0].event.is_interrupt() and res[1] is ev  # SamplesLost first (chunks.rs:225-233)
    out2 = g.process(rr.Samples(1e9, x[M * 6 :]))
    assert len(out2) == 3  # the history starts again from empty
    ref, _ = reference_frames(oracle, [x[i * M : (i + 1) * M] for i in range(6, 12)], M, P, np.float64)
    for a, b in zip(out2, ref):
        assert rms_rel(a.chunk, b) <= 1e-5


def test_channelizer_device_batch_and_errors(rr, oracle):
    import torch

    from radiorust_amd._lib import BackendError

    M, P, k = 256, 4, 4096
    n = M * k
    d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rr.synth_iq_dev(0, st, 23, 0, n, d_in.data_ptr())
    d_out = torch.empty(n, dtype=torch.complex64, device="cuda")
    g = rr.Channelizer(M, P)
    g.set_stream(st)
    wrote = g.process_dev(d_in.data_ptr(), n, d_out.data_ptr(), n)
    assert wrote == (k - (P - 1)) * M
    torch.cuda.synchronize()
    x = oracle.synth_iq(23, 0, n)
    got = d_out.cpu().numpy()[:wrote].reshape(-1, M)
    ref, _ = reference_frames(oracle, [x[i * M : (i + 1) * M] for i in range(40)], M, P, np.float64)
    for j in (0, 1, 17, 36):
        assert rms_rel(got[j], ref[j]) <= 1e-5
    with pytest.raises(BackendError):
        g.process(rr.Samples(1e9, x[: M + 1]))  # not whole chunks
    with pytest.raises(BackendError):
        rr.Channelizer(256, 4, hop=96)  # the hop must divide bins * taps_per_branch
    with pytest.raises(BackendError):
        rr.Channelizer(256, 4, hop=512)  # .. and not exceed bins


def overlapped_spectra(oracle, chunks, P, window, center_dc, flt, history=None):
    """Overlapper(P) + Fourier, chunk by chunk (chunks.rs:194-242, analysis.rs:60-132)."""
    hist = [] if history is None else history
    fo = oracle.Fourier(window, center_dc=center_dc, flt=flt)
    out = []
    for c in chunks:
        hist.append(c)
        if len(hist) >= P:
            out.append(fo.process(np.concatenate(hist[-P:])))
            del hist[: len(hist) - (P - 1)]
    return out, hist


@pytest.mark.parametrize("M,P,center,dtype,tol", [(1024, 4, False, np.float32, 1e-5), (256, 4, True, np.float32, 1e-5),
                                                  (4096, 1, False, np.float32, 1e-5), (2048, 2, True, np.float32, 1e-5),
                                                  (64, 8, False, np.float64, 1e-12), (512, 16, False, np.float32, 1e-5),
                                                  (512, 2, False, np.float32, 1e-5), (256, 1, False, np.float32, 1e-5),
                                                  (1024, 1, True, np.float32, 1e-5), (1024, 2, False, np.float32, 1e-5), (512, 4, True, np.float32, 1e-5), (128, 4, True, np.float32, 1e-5), (256, 2, False, np.float32, 1e-5), (2048, 4, True, np.float32, 1e-5), (250, 4, True, np.float32, 1e-5), (100, 3, False, np.float32, 1e-5),
                                                  (100, 3, True, np.float64, 1e-11), (1500, 4, False, np.float32, 1e-5),
                                                  # spans 2^a 3^b 5^c (7 ..): overlapping frames through k_fft_mixed / the two passes of k_fft_tilem
                                                  (3000, 4, True, np.float32, 1e-5), (2500, 2, False, np.float64, 1e-11), (1001, 4, False, np.float32, 1e-5),
                                                  # a span of 5001 = 3 x 1667 points: k_bluestein_big<16384> with a hop
                                                  (1667, 3, True, np.float32, 1e-5),
                                                  # 16 384-point spans: k_fft16384 with a hop
                                                  (4096, 4, True, np.float32, 1e-5), (2048, 8, False, np.float32, 1e-5)])
def test_stft_parity(rr, oracle, M, P, center, dtype, tol):
    """rr_stft_*: Rechunker -> Overlapper -> Fourier on the device (4096-point spans run k_fft4096
    with a hop, the others the generic power-of-two kernel) against the oracle composition."""
    nchunks = 23
    x = oracle.synth_iq(22, 0, M * nchunks)
    if dtype == np.float64:
        x = x.astype(np.complex128)
    chunks = [x[i * M : (i + 1) * M] for i in range(nchunks)]
    win = oracle.Kaiser.with_null_at_bin(2.0)
    ref, _ = overlapped_spectra(oracle, chunks, P, win, center, np.float64)
    g = rr.Stft(M, P, rr.Kaiser.with_null_at_bin(2.0), center_dc=center, dtype=dtype)
    got = []
    cuts = [0, 1, 2, 10, 11, 20, nchunks]
    for a, b in zip(cuts[:-1], cuts[1:]):
        got += g.process(rr.Samples(1e6, x[a * M : b * M]))
    assert len(got) == len(ref) == max(nchunks - (P - 1), 0)
    for a, b in zip(got, ref):
        assert len(a.chunk) == M * P and rms_rel(a.chunk, b) <= tol


@pytest.mark.parametrize("P,center", [(2, False), (4, True), (8, False), (16, True)])
def test_stft_4096_point_spans_in_long_calls(rr, oracle, P, center):
    """4096-point spans with 64 and more frames per call run k_stft4096 (a workgroup per run of frames, the
    sliding window in registers); shorter calls k_fft4096 with a hop.  Both against the oracle composition,
    with the history handed from call to call."""
    M = 4096 // P
    nchunks = 300 + P
    x = oracle.synth_iq(26, 0, M * nchunks)
    chunks = [x[i * M : (i + 1) * M] for i in range(nchunks)]
    win = oracle.Kaiser.with_null_at_bin(2.0)
    ref, _ = overlapped_spectra(oracle, chunks, P, win, center, np.float64)
    g = rr.Stft(M, P, rr.Kaiser.with_null_at_bin(2.0), center_dc=center)
    got = []
    cuts = [0, 3, 100, 101, 180, 290, nchunks]   # calls of 3, 97, 1, 79, 110, 10 + P chunks
    for a, b in zip(cuts[:-1], cuts[1:]):
        got += g.process(rr.Samples(1e6, x[a * M : b * M]))
    assert len(got) == len(ref) == nchunks - (P - 1)
    for a, b in zip(got, ref):
        assert rms_rel(a.chunk, b) <= 1e-5


def test_stft_event_resets_history(rr, oracle):
    M, P = 256, 4
    x = oracle.synth_iq(23, 0, M * 12)
    chunks = [x[i * M : (i + 1) * M] for i in range(12)]
    g = rr.Stft(M, P)
    a = g.process(rr.Samples(1e6, x[: 5 * M]))
    ev = rr.EventSignal(rr.Disconnection())
    mid = g.process(ev)
    assert len(mid) == 2 and mid[0].is_event() and mid[1] is ev  # SamplesLost first (chunks.rs:225-233)
    b = g.process(rr.Samples(1e6, x[5 * M :]))
    r1, _ = overlapped_spectra(oracle, chunks[:5], P, oracle.Rectangular(), False, np.float64)
    r2, _ = overlapped_spectra(oracle, chunks[5:], P, oracle.Rectangular(), False, np.float64)
    assert len(a) == len(r1) == 2 and len(b) == len(r2) == 4
    for s, r in zip(a + b, r1 + r2):
        assert rms_rel(s.chunk, r) <= 1e-5


def test_stft_contract(rr):
    from radiorust_amd._lib import BackendError, ContractViolation

    with pytest.raises(ContractViolation):
        rr.Stft(0, 4)
    with pytest.raises(ContractViolation):
        rr.Stft(64, 0)
    with pytest.raises(BackendError):
        rr.Stft(5, 3)  # 15 points: overlapping chunks need a power-of-two span (one LDS tile) or 32 points and more (Bluestein)
    with pytest.raises(BackendError):
        rr.Stft(8192, 4)  # 32768 points: a power of two beyond the overlapped kernels (one image in LDS: up to 16384 in f32)
    with pytest.raises(BackendError):
        rr.Stft(4096, 2, dtype=np.float64)  # (f64: up to 4096)


def test_stft_rechunks_arbitrary_input(rr, oracle):
    """The Rechunker in front (chunks.rs:42-177; the reference's own test feeds 4096-sample chunks into
    Rechunker(1024), chunks.rs:247-271): input messages of any length give the same spectra as whole
    chunks, and a change of sample rate drops the patchwork and the history behind a SamplesLost."""
    M, P = 1024, 2
    x = oracle.synth_iq(24, 0, M * 9 + 300)
    chunks = [x[i * M : (i + 1) * M] for i in range(9)]
    ref, _ = overlapped_spectra(oracle, chunks, P, oracle.Rectangular(), False, np.float64)
    g = rr.Stft(M, P)
    got = []
    for a, b in ((0, 4096), (4096, 4097), (4097, 4100), (4100, 9000), (9000, len(x))):
        got += g.process(rr.Samples(1.0, x[a:b]))
    assert len(got) == len(ref) == 8 and all(len(s.chunk) == M * P for s in got)
    for s, r in zip(got, ref):
        assert rms_rel(s.chunk, r) <= 1e-5
    assert g.pending() == 300
    out = g.process(rr.Samples(2.0, x[: 3 * M]))  # new rate: 300 pending samples and the history are dropped
    # the Rechunker's SamplesLost (chunks.rs:72-79) reaches the Overlapper as an event: it sends one of its own first
    assert out[0].is_event() and out[1].is_event() and len(out) == 2 + 2
    ref2, _ = overlapped_spectra(oracle, chunks[:3], P, oracle.Rectangular(), False, np.float64)
    for s, r in zip(out[2:], ref2):
        assert s.sample_rate == 2.0 and rms_rel(s.chunk, r) <= 1e-5


def test_stft_rate_change_and_events_like_the_composition(rr, oracle):
    """Rechunker -> Overlapper -> Fourier, message by message (chunks.rs:62-92, 207-233):
    a new sample rate with NOTHING pending loses nothing - the Overlapper keeps its history and labels each
    output with the length-weighted average rate of its chunks; an event while samples are pending is preceded
    by the Rechunker's SamplesLost, and the Overlapper puts a SamplesLost of its own in front of every event."""
    M, P = 256, 4
    x = oracle.synth_iq(25, 0, M * 10 + 100)
    chunks = [x[i * M : (i + 1) * M] for i in range(10)]
    ref, _ = overlapped_spectra(oracle, chunks, P, oracle.Rectangular(), False, np.float64)
    g = rr.Stft(M, P)
    a = g.process(rr.Samples(1000.0, x[: 5 * M]))           # chunks 0 .. 4 at rate 1000: spectra 0, 1
    assert g.pending() == 0
    b = g.process(rr.Samples(2000.0, x[5 * M : 10 * M]))    # chunks 5 .. 9 at rate 2000: spectra 2 .. 6, nothing lost
    assert not any(s.is_event() for s in a + b) and len(a) == 2 and len(b) == 5
    for s, r in zip(a + b, ref):
        assert rms_rel(s.chunk, r) <= 1e-5
    # rates: windows (2,3,4,5), (3,4,5,6), (4,5,6,7) mix the two rates
    assert [s.sample_rate for s in a + b] == [1000.0, 1000.0, 1250.0, 1500.0, 1750.0, 2000.0, 2000.0]
    c = g.process(rr.Samples(2000.0, x[10 * M :]))           # 100 samples: no chunk yet
    assert c == [] and g.pending() == 100
    ev = rr.EventSignal(rr.Disconnection())
    out = g.process(ev)
    assert [s.is_event() for s in out] == [True] * 4 and out[3] is ev and g.pending() == 0
    assert all(isinstance(s.event, rr.SamplesLost) for s in out[:3])
    again = g.process(rr.Samples(2000.0, x[: 4 * M]))       # the history is gone: P chunks for the first spectrum
    assert len(again) == 1 and rms_rel(again[0].chunk, ref[0]) <= 1e-5
