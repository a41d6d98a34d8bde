#!/usr/bin/env python3
"""A few calls of the cfg5 Filter (n = 1024, 2^26 samples) for profiler runs; RR_FILTER4K_VARIANT picks the form."""
import os, sys
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
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    f.process_dev(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N)
torch.cuda.synchronize()
