"""Bluestein around the workgroup transforms of rr_fft_big.hpp (k_bluestein_big<M>) against the routes it replaces: ms per 2^24
samples and the fraction of the HBM bound (16 bytes per sample).  Run on the GPU box: python scripts/bs_big_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radiorust_amd as rr  # noqa: E402

N = 1 << 24
d_in = torch.randn(N, dtype=torch.complex64, device="cuda")
d_out = torch.empty_like(d_in)
st = torch.cuda.current_stream().cuda_stream


def timed(f, k=5):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(k):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for nf, env in ((3001, {}), (3001, {"RR_FOURIER_BS8K": "big"}), (4093, {}), (4093, {"RR_FOURIER_BS8K": "big"}),
                (2049, {}), (2049, {"RR_FOURIER_BS8K": "big"}),
                (5003, {}), (5003, {"RR_FOURIER_BS_BIG": "0"}), (6007, {}), (8191, {}), (8191, {"RR_FOURIER_BS_BIG": "0"}), (4099, {})):
    for k, v in env.items():
        os.environ[k] = v
    g = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
    g.set_stream(st)
    n = N // nf * nf
    dt = timed(lambda: g.process_dev(nf, d_in.data_ptr(), n, d_out.data_ptr(), n))
    print(f"Fourier n = {nf:5d} {str(env):36s} {rr.fourier_route(nf):34s} {dt * 1e3 * N / n:8.3f} ms / 2^24  {16 * n / dt / 8e12 * 100:5.1f} %", flush=True)
    for k in env:
        os.environ.pop(k)
