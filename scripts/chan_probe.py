#!/usr/bin/env python3
"""The 512 / 1024 / 2048 / 4096-bin channelizers by taps per branch (2^26 samples per call, device-resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
for M, P in ((512, 2), (512, 4), (512, 8), (1024, 2), (1024, 4), (1024, 6), (1024, 8), (1024, 16), (2048, 2), (2048, 4), (2048, 8), (4096, 2), (4096, 4), (4096, 8)):
    c = rr.Channelizer(M, P)
    c.set_stream(st)
    for _ in range(3): c.process_dev(d_in.data_ptr(), N, d_out.data_ptr(), N)
    torch.cuda.synchronize()
    K = 10
    t = time.perf_counter()
    for _ in range(K): c.process_dev(d_in.data_ptr(), N, d_out.data_ptr(), N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"channelizer {M} bins x {P} taps/branch: {dt*1e3:.3f} ms per 2^26 samples = {100*16*N/dt/8e12:.1f} % of 16 B/sample")
