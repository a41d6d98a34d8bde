#!/usr/bin/env python3
"""rr_meter alone for the profiler: 20 calls of 2^26 input samples (rocprofv3 --kernel-trace --stats -- python3 scripts/prof_meter.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
N, fs, out_rate, bw, q = 1 << 26, 1024000.0, 102400.0, 60e3, 4
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
resp = lambda b, f: 1.0 if abs(f) <= bw / 2 else 0.0
m = rr.Meter(shift=12.5e3, output_rate=out_rate, bandwidth=bw, chunk_len=1024, freq_resp=resp, overlap=q,
             fft_window=rr.Kaiser.with_null_at_bin(float(q)))
m.set_stream(st)
cap = (N // 10 // 1024 + 8) * 4096
d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
for _ in range(20):
    m.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), cap)
torch.cuda.synchronize()
