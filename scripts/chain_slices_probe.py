#!/usr/bin/env python3
"""The chain's two-kernel shapes: one call of 2^26 samples against the same stream in 2, 4, 8, 16 calls.  With smaller calls the
decimated samples the Fourier kernel reads may still sit in the memory-side cache (256 MB) when it asks for them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

N, fs = 1 << 26, 200e6
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
for name, kw, D in (("chain 8:1 / FFT 1024", dict(shift=25e6, filter_len=64, output_rate=25e6, bandwidth=20e6, fft_len=1024), 8),
                    ("chain 2:1 / FFT 8192", dict(shift=25e6, filter_len=64, output_rate=100e6, bandwidth=80e6, fft_len=8192), 2),
                    ("chain 8:1 / FFT 4096", dict(shift=25e6, filter_len=64, output_rate=25e6, bandwidth=20e6, fft_len=4096), 8),
                    ("chain 10:1 / FFT 4096", dict(shift=12.5e6, filter_len=64, output_rate=20e6, bandwidth=12e6, fft_len=4096), 10),
                    ("chain 4:1 / FFT 4096 (cfg2)", dict(shift=25e6, filter_len=64, output_rate=50e6, bandwidth=40e6, fft_len=4096), 4)):
    for parts in (1, 2, 4, 8, 16):
        ch = rr.Chain(freq_resp=lp, fft_window=rr.Kaiser.with_null_at_bin(2.0), **kw)
        ch.set_stream(st)
        n = N // parts
        cap = n // D + 2 * kw["fft_len"]
        co = torch.empty(cap * parts, dtype=torch.complex64, device="cuda")

        def step():
            for p in range(parts):
                ch.process_dev(fs, d_in.data_ptr() + 8 * n * p, n, co.data_ptr() + 8 * cap * p, cap)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        K = 10
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / K
        b = 8 + 8 / D
        print(f"{name:30s} {parts:2d} calls of 2^{n.bit_length() - 1}: {dt * 1e3:7.3f} ms = {100 * b * N / dt / 8e12:4.1f} %   ({ch.last_path_kernel()})", flush=True)
        del ch, co
