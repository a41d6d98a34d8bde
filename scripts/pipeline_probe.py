#!/usr/bin/env python3
"""Do two chains on two HIP streams fill each other's launch boundaries?  One chain alone pays ~13 us per launch for the
ramp-up of its first round of workgroups and the drain of its last (2^24 samples = one round 51 us, every further round
38 us: scripts/callsize_probe.py); two independent channels issued alternately on two streams should hide most of it.
Prints ms per 2^26 samples for: one chain; two chains on one stream; two chains on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

N, fs, K = 1 << 26, 200e6, 200
lp = lambda _b, f: 1.0 if abs(f) <= 20e6 else 0.0


def mk(stream, seed):
    c = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                 fft_window=rr.Kaiser.with_null_at_bin(2.0), device=0)
    c.set_stream(stream.cuda_stream)
    d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, stream.cuda_stream, seed, 0, N, d_in.data_ptr())
    cap = (N // 16 // 4096 + 2) * 4096 * 4
    d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
    return c, d_in, d_out, cap


def run(chains, k):
    for _ in range(k):
        for c, d_in, d_out, cap in chains:
            c.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), cap)


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, streams in (("one chain", [s0]), ("two chains, one stream", [s0, s0]), ("two chains, two streams", [s1, s2])):
    chains = [mk(s, i + 1) for i, s in enumerate(streams)]
    torch.cuda.synchronize()
    run(chains, 300 // len(chains))  # settle
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(chains, K // len(chains))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    per = dt / K * 1e3
    print(f"{name:28s} {per:.4f} ms per 2^26 samples = {N / per / 1e6:.1f} GSamples/s = {N * 10 / per / 1e6 / 8000 * 100:.1f} %", flush=True)
    del chains
for lg in (27, 28):
    n2 = 1 << lg
    c = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                 fft_window=rr.Kaiser.with_null_at_bin(2.0), device=0)
    c.set_stream(s0.cuda_stream)
    d_in = torch.empty(n2, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, s0.cuda_stream, 1, 0, n2, d_in.data_ptr())
    cap = (n2 // 4 // 4096 + 2) * 4096
    d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
    k2 = 200 >> (lg - 26)
    for _ in range(k2):
        c.process_dev(fs, d_in.data_ptr(), n2, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k2):
        c.process_dev(fs, d_in.data_ptr(), n2, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t) / k2 / (n2 >> 26) * 1e3
    print(f"one chain, 2^{lg} per call        {per:.4f} ms per 2^26 samples = {N * 10 / per / 1e6 / 8000 * 100:.1f} %", flush=True)
    del c, d_in, d_out
