#!/usr/bin/env python3
"""Does k_fft4096 (stream B) run concurrently with k_ols_wave (stream A)?  Times the chain's fused
step alone, 4096 independent Fourier frames alone, and both issued together on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

fs, n = 200e6, 1 << 26
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ch = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
              fft_window=rr.Kaiser.with_null_at_bin(2.0))
ch.set_stream(sa.cuda_stream)
fo = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
fo.set_stream(sb.cuda_stream)
d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, sa.cuda_stream, 1, 0, n, d_in.data_ptr())
cap = (n // 4 // 4096 + 2) * 4096
d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
m = n // 4
d_fin = torch.randn(m, dtype=torch.complex64, device="cuda")
d_fout = torch.empty(m, dtype=torch.complex64, device="cuda")
torch.cuda.synchronize()

def run(chain, fourier, k=10):
    for _ in range(3):
        if chain: ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)
        if fourier: fo.process_dev(4096, d_fin.data_ptr(), m, d_fout.data_ptr(), m)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        if chain: ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)
        if fourier: fo.process_dev(4096, d_fin.data_ptr(), m, d_fout.data_ptr(), m)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3

a, b, c = run(True, False), run(False, True), run(True, True)
print(f"chain step alone {a:.4f} ms, 4096 Fourier frames alone {b:.4f} ms, both on two streams {c:.4f} ms (sum {a+b:.4f})")
