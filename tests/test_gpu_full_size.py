"""BASELINE's full timing sizes (2^26 samples per call) for the configs beside the chain: configs[4] (the 1024-tap Filter at
2 GS/s) and configs[2] (the 256-bin, 4-tap/branch channelizer at 1 GS/s).  At this size the C oracle covers a prefix; the
rest is checked through properties that do not depend on the size: the cut of a stream into calls must not show, the blocks
are linear, and the Fourier stage keeps the energy its window was scaled for (analysis.rs:97).  (The chain's own full-size
test, 2^28 samples in one call, is tests/test_gpu_chain.py::test_chain_one_call_of_2_pow_28_samples.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1 << 26


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


def ready():
    """torch's work (fills, arithmetic) is done before a handle touches the buffers: with torch's null stream as the
    current one a handle runs on a stream of its own, unordered against torch's"""
    import torch

    torch.cuda.synchronize()


def dev_err(a, b):
    import torch

    return (torch.linalg.vector_norm((a - b).view(-1)) / torch.linalg.vector_norm(b.view(-1))).item()


def lowpass200(_b, f):
    return 1.0 if abs(f) <= 200e6 else 0.0


def test_cfg5_filter_full_size(rr, oracle):
    """Filter n = 1024 at 2 GS/s over 2^26 samples in one call (k_filter_blk4096): a prefix against the f64 oracle, the whole
    output against the same stream fed in 16 calls of 2^22 samples, and linearity F(a x + y) = a F(x) + F(y)."""
    import torch

    n, fs = 1024, 2e9
    st = torch.cuda.current_stream().cuda_stream
    d_x = torch.empty(N, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 5, 0, N, d_x.data_ptr())
    torch.cuda.synchronize()
    g = rr.Filter.new(lowpass200)
    g.set_stream(st)
    d_y = torch.zeros(N, dtype=torch.complex64, device="cuda")
    ready()
    wrote = g.process_dev(fs, n, d_x.data_ptr(), N, d_y.data_ptr(), N)
    assert wrote == N - n  # the first chunk after a reset is swallowed (filters.rs:187, 240, 260)
    torch.cuda.synchronize()
    assert g.last_kernel() == 2  # k_filter_blk4096
    # prefix against the oracle, chunk by chunk as the reference runs it
    K = 64
    x = d_x[: (K + 1) * n].cpu().numpy()
    o = oracle.Filter(lowpass200, flt=np.float64)
    ref = []
    for i in range(K + 1):
        r = o.process(fs, x[i * n : (i + 1) * n])
        if r is not None and len(r):
            ref.append(r)
    ref = np.concatenate(ref)
    assert rms_rel(d_y[: len(ref)].cpu().numpy(), ref) <= 1e-5
    # the cut into calls does not show
    g2 = rr.Filter.new(lowpass200)
    g2.set_stream(st)
    d_y2 = torch.zeros(N, dtype=torch.complex64, device="cuda")
    ready()
    off = w2 = 0
    step = 1 << 22
    while off < N:
        w2 += g2.process_dev(fs, n, d_x.data_ptr() + 8 * off, step, d_y2.data_ptr() + 8 * w2, N - w2)
        off += step
    assert w2 == wrote
    torch.cuda.synchronize()
    assert dev_err(d_y[:wrote], d_y2[:wrote]) <= 2e-6
    # linearity on a quarter of the stream
    M = 1 << 24
    a = 0.75 - 0.5j
    d_z = torch.empty(M, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 6, 0, M, d_z.data_ptr())
    torch.cuda.synchronize()
    d_mix = a * d_x[:M] + d_z
    outs = []
    for src in (d_x[:M].contiguous(), d_z, d_mix):
        h = rr.Filter.new(lowpass200)
        h.set_stream(st)
        d_o = torch.zeros(M, dtype=torch.complex64, device="cuda")
        ready()
        assert h.process_dev(fs, n, src.data_ptr(), M, d_o.data_ptr(), M) == M - n
        torch.cuda.synchronize()
        outs.append(d_o[: M - n])
    assert dev_err(outs[2], a * outs[0] + outs[1]) <= 3e-6


def test_cfg3_channelizer_full_size(rr, oracle):
    """256 bins x 4 taps/branch over 2^26 samples in one call (k_channelizer256): a prefix against the composition it is defined
    by (Overlapper(4) -> Fourier over 1024 samples -> every 4th bin), the whole output against 8 calls of 2^23 samples, and
    linearity."""
    import torch

    M, P = 256, 4
    st = torch.cuda.current_stream().cuda_stream
    d_x = torch.empty(N, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 7, 0, N, d_x.data_ptr())
    torch.cuda.synchronize()
    g = rr.Channelizer(M, P)
    g.set_stream(st)
    frames = N // M - (P - 1)
    d_y = torch.zeros(frames * M, dtype=torch.complex64, device="cuda")
    ready()
    assert g.process_dev(d_x.data_ptr(), N, d_y.data_ptr(), d_y.numel()) == frames * M
    torch.cuda.synchronize()
    K = 40
    x = d_x[: (K + P - 1) * M].cpu().numpy()
    fo = oracle.Fourier(oracle.Kaiser.with_null_at_bin(float(P)), flt=np.float64)
    ref = np.concatenate([fo.process(x[i * M : (i + P) * M])[::P] for i in range(K)])
    assert rms_rel(d_y[: K * M].cpu().numpy(), ref) <= 1e-5
    g2 = rr.Channelizer(M, P)
    g2.set_stream(st)
    d_y2 = torch.zeros_like(d_y)
    ready()
    off = w2 = 0
    step = 1 << 23
    while off < N:
        w2 += g2.process_dev(d_x.data_ptr() + 8 * off, step, d_y2.data_ptr() + 8 * w2, d_y2.numel() - w2)
        off += step
    assert w2 == frames * M
    torch.cuda.synchronize()
    assert dev_err(d_y, d_y2) <= 2e-6
    Mq = 1 << 24
    a = -0.5 + 1.25j
    d_z = torch.empty(Mq, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 8, 0, Mq, d_z.data_ptr())
    torch.cuda.synchronize()
    d_mix = a * d_x[:Mq] + d_z
    outs = []
    fq = Mq // M - (P - 1)
    for src in (d_x[:Mq].contiguous(), d_z, d_mix):
        h = rr.Channelizer(M, P)
        h.set_stream(st)
        d_o = torch.zeros(fq * M, dtype=torch.complex64, device="cuda")
        ready()
        assert h.process_dev(src.data_ptr(), Mq, d_o.data_ptr(), d_o.numel()) == fq * M
        torch.cuda.synchronize()
        outs.append(d_o)
    assert dev_err(outs[2], a * outs[0] + outs[1]) <= 3e-6


def test_fourier_full_size_energy_and_linearity(rr, oracle):
    """Fourier 4096 with the chain's window over 2^26 samples (k_fft4096): the window is scaled to unit mean square
    (analysis.rs:97), so for white input the energy of the spectra is n times the energy of the samples within the statistics
    of 2^26 samples; the first chunks against the oracle; linearity on 2^24 samples."""
    import torch

    n = 4096
    st = torch.cuda.current_stream().cuda_stream
    d_x = torch.randn(N, dtype=torch.complex64, device="cuda")
    g = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
    g.set_stream(st)
    d_y = torch.empty_like(d_x)
    torch.cuda.synchronize()
    assert g.process_dev(n, d_x.data_ptr(), N, d_y.data_ptr(), N) == N
    torch.cuda.synchronize()
    ex = torch.sum(d_x.real.double() ** 2 + d_x.imag.double() ** 2).item()
    ey = torch.sum(d_y.real.double() ** 2 + d_y.imag.double() ** 2).item()
    assert abs(ey / (n * ex) - 1.0) < 2e-3
    o = oracle.Fourier(oracle.Kaiser.with_null_at_bin(2.0), False, flt=np.float64)
    x = d_x[: 8 * n].cpu().numpy()
    ref = np.concatenate([o.process(x[i * n : (i + 1) * n]) for i in range(8)])
    assert rms_rel(d_y[: 8 * n].cpu().numpy(), ref) <= 1e-5
    Mq = 1 << 24
    a = 1.5 + 0.25j
    d_z = torch.randn(Mq, dtype=torch.complex64, device="cuda")
    d_mix = a * d_x[:Mq] + d_z
    outs = []
    for src in (d_x[:Mq].contiguous(), d_z, d_mix):
        d_o = torch.empty(Mq, dtype=torch.complex64, device="cuda")
        torch.cuda.synchronize()
        assert g.process_dev(n, src.data_ptr(), Mq, d_o.data_ptr(), Mq) == Mq
        torch.cuda.synchronize()
        outs.append(d_o)
    assert dev_err(outs[2], a * outs[0] + outs[1]) <= 3e-6
