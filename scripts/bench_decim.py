#!/usr/bin/env python3
"""The Downsampler alone for a list of rate pairs (2^26 device-resident samples): ms, GSamples/s, % of its 8 + 8 Q / P
roofline, which kernel ran.  RR_DOWNSAMPLER_POLY=1 routes the ratios 2, 4, 8 through k_decim_poly as well."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
cases = [(200e6, 50e6, 40e6), (384000.0, 48000.0, 40000.0), (96000.0, 48000.0, 30000.0), (1024000.0, 102400.0, 60000.0),
         (1024000.0, 384000.0, 200000.0), (300000.0, 100000.0, 60000.0), (48000.0, 32000.0, 20000.0), (2560000.0, 40000.0, 30000.0)]
for fin, fout, bw in cases:
    ds = rr.Downsampler.new(4096, fout, bw)
    ds.set_stream(st)
    for _ in range(3):
        ds.process_dev(fin, d_in.data_ptr(), N, d_out.data_ptr(), N)
    torch.cuda.synchronize()
    K = 10
    t = time.perf_counter()
    for _ in range(K):
        ds.process_dev(fin, d_in.data_ptr(), N, d_out.data_ptr(), N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    bps = 8 + 8 * fout / fin
    print(f"{fin:.0f} -> {fout:.0f} (L = {ds.ir_len()}): {dt*1e3:.3f} ms  {N/dt/1e9:.1f} GSamples/s  {100*bps*N/dt/8e12:.1f} % of {bps:.2f} B/sample  kernel {ds.last_kernel()}")
