#!/usr/bin/env python3
"""The f64 Downsampler and the f64 chain alone, for rocprofv3 --kernel-trace --stats (RR_DECIM_F64_R4=0: the first kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radiorust_amd as rr

N2 = 1 << 24
st = torch.cuda.current_stream().cuda_stream
d64 = torch.randn(N2, dtype=torch.complex128, device="cuda")
o64 = torch.empty(N2, dtype=torch.complex128, device="cuda")
g = rr.Downsampler.new(4096, 50e6, 40e6, dtype=np.float64); g.set_stream(st)
for _ in range(12):
    g.process_dev(200e6, d64.data_ptr(), N2, o64.data_ptr(), N2)
torch.cuda.synchronize()
lp20 = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
ch = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp20, output_rate=50e6, bandwidth=40e6, fft_len=4096,
              fft_window=rr.Kaiser.with_null_at_bin(2.0), dtype=np.float64)
ch.set_stream(st)
co = torch.empty(N2 // 4 + 8192, dtype=torch.complex128, device="cuda")
for _ in range(12):
    ch.process_dev(200e6, d64.data_ptr(), N2, co.data_ptr(), co.numel())
torch.cuda.synchronize()
print("done", g.last_kernel(), ch.last_path_kernel())
