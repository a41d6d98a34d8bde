#!/usr/bin/env python3
"""k_decim_poly alone, for rocprofv3 (--kernel-trace --stats, or a --pmc pass): the chain at 10 : 1 and 5 : 1 and the stand-alone
Downsampler at 10 : 1, 20 calls of 2^26 samples each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

N, fs = 1 << 26, 200e6
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for kw, D in ((dict(shift=12.5e6, filter_len=64, output_rate=20e6, bandwidth=12e6, fft_len=4096), 10),
              (dict(shift=12.5e6, filter_len=64, output_rate=40e6, bandwidth=30e6, fft_len=4096), 5)):
    ch = rr.Chain(freq_resp=lp, fft_window=rr.Kaiser.with_null_at_bin(2.0), **kw)
    ch.set_stream(st)
    cap = N // D + 2 * 4096
    for _ in range(K):
        ch.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
ds = rr.Downsampler.new(4096, 102400.0, 60000.0)
ds.set_stream(st)
for _ in range(K):
    ds.process_dev(1024000.0, d_in.data_ptr(), N, d_out.data_ptr(), N)
torch.cuda.synchronize()
print("done")
