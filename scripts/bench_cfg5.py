#!/usr/bin/env python3
"""BASELINE configs[4] (SURVEY cfg5): 1024-tap Filter at 2 GS/s on one MI355X, device-resident stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
n, N, fs = 1024, 1 << 26, 2e9
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
f = rr.Filter.new(lambda b, fr: 1.0 if abs(fr) <= 200e6 else 0.0)
f.set_stream(st)
for _ in range(3):
    f.process_dev(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N)
torch.cuda.synchronize()
K = 10
t = time.perf_counter()
for _ in range(K):
    f.process_dev(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"cfg5 Filter n=1024 f32: {dt*1e3:.3f} ms per 2^26 samples = {N/dt/1e6:.0f} MSamples/s, "
      f"{16*N/dt/1e9:.0f} GB/s algorithmic (16 B/sample) = {100*16*N/dt/8e12:.1f} % of HBM roofline")
