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
# (200 calls = 50 ms of load first: the power controller's first tens of milliseconds after an idle gap are a transient - ten calls
#  from idle measured 0.276 ms where the steady state and the rocprofv3 average over 400 launches say 0.24)
for _ in range(200):
    f.process_dev(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N)
torch.cuda.synchronize()
K = 100
t = time.perf_counter()
for _ in range(K):
    f.process_dev(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"cfg5 Filter n=1024 f32: {dt*1e3:.3f} ms per 2^26 samples = {N/dt/1e6:.0f} MSamples/s, "
      f"{16*N/dt/1e9:.0f} GB/s algorithmic (16 B/sample) = {100*16*N/dt/8e12:.1f} % of HBM roofline")

# the half-precision points of SURVEY 8(d): f16 output, and f16 output + f16 response table
import numpy as np
from oracle import rr_oracle as o
M = 1 << 16
xh = o.synth_iq(1, 0, M)
of = o.Filter(lambda b, fr: 1.0 if abs(fr) <= 200e6 else 0.0, flt=np.float64)
ref = [of.process(fs, xh[a:a + n].astype(np.complex128)) for a in range(0, M, n)]
ref = np.concatenate([r for r in ref if r is not None])
d_h = torch.empty(2 * N, dtype=torch.float16, device="cuda")
for resp16 in (False, True):
    g = rr.Filter.new(lambda b, fr: 1.0 if abs(fr) <= 200e6 else 0.0)
    g.set_stream(st)
    got = g.process_dev_f16(fs, n, d_in.data_ptr(), M, d_h.data_ptr(), N, response_f16=resp16)
    torch.cuda.synchronize()
    y = d_h[: 2 * got].cpu().numpy().astype(np.float64)
    y = y[0::2] + 1j * y[1::2]
    err = float(np.sqrt(np.sum(np.abs(y - ref) ** 2) / np.sum(np.abs(ref) ** 2)))
    for _ in range(200):
        g.process_dev_f16(fs, n, d_in.data_ptr(), N, d_h.data_ptr(), N, response_f16=resp16)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        g.process_dev_f16(fs, n, d_in.data_ptr(), N, d_h.data_ptr(), N, response_f16=resp16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"cfg5 Filter n=1024 f16 out{' + f16 response' if resp16 else ''}: {dt*1e3:.3f} ms = {N/dt/1e6:.0f} MSamples/s, "
          f"{12*N/dt/1e9:.0f} GB/s algorithmic (12 B/sample) = {100*12*N/dt/8e12:.1f} % of HBM roofline, "
          f"rms error vs f64 oracle {err:.2e}")
