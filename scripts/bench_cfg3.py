#!/usr/bin/env python3
"""BASELINE configs[2] (SURVEY cfg3): 256-bin polyphase FFT channelizer, 4 taps/branch, device-resident stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
M, P, N = 256, 4, 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
c = rr.Channelizer(M, P)
c.set_stream(st)
for _ in range(3):
    c.process_dev(d_in.data_ptr(), N, d_out.data_ptr(), N)
torch.cuda.synchronize()
K = 10
t = time.perf_counter()
for _ in range(K):
    c.process_dev(d_in.data_ptr(), N, d_out.data_ptr(), N)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"cfg3 channelizer M=256 P=4 f32: {dt*1e3:.3f} ms per 2^26 samples = {N/dt/1e6:.0f} MSamples/s, "
      f"{16*N/dt/1e9:.0f} GB/s algorithmic (16 B/sample) = {100*16*N/dt/8e12:.1f} % of HBM roofline")
