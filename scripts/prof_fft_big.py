#!/usr/bin/env python3
"""Fourier beyond one LDS image for the profiler: 2^16 and 20000 points, 20 calls of 2^24 samples each (which pass of the
two-pass kernels takes what)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

st = torch.cuda.current_stream().cuda_stream
N = 1 << 24
d_in = torch.randn(N, dtype=torch.complex64, device="cuda")
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
for nf in (1 << 14, 1 << 16, 1 << 18, 20000, 250000, 3000):
    g = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
    g.set_stream(st)
    n = N // nf * nf
    for _ in range(20):
        g.process_dev(nf, d_in.data_ptr(), n, d_out.data_ptr(), n)
    torch.cuda.synchronize()
