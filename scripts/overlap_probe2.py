#!/usr/bin/env python3
"""Would the chain gain from running the Fourier stage of step t beside the FIR stage of step t + 1?  k_ols_wave (as
the 4 : 1 Downsampler, stream A, 2^26 samples) and k_fft4096 (stream B, 2^24 samples) alone, back to back on one
stream, and issued together on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

n = 1 << 26
m = n // 4
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ds = rr.Downsampler.new(4096, 50e6, 40e6)
fo = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, sa.cuda_stream, 1, 0, n, d_in.data_ptr())
d_dec = torch.empty(m + 4096, dtype=torch.complex64, device="cuda")
d_fin = torch.randn(m, dtype=torch.complex64, device="cuda")
d_fout = torch.empty(m, dtype=torch.complex64, device="cuda")
torch.cuda.synchronize()


def run(a, b, two, k=20):
    ds.set_stream(sa.cuda_stream)
    fo.set_stream(sb.cuda_stream if two else sa.cuda_stream)
    def body():
        if a: ds.process_dev(200e6, d_in.data_ptr(), n, d_dec.data_ptr(), m + 4096)
        if b: fo.process_dev(4096, d_fin.data_ptr(), m, d_fout.data_ptr(), m)
    for _ in range(5): body()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k): body()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3


for rnd in range(3):
    a, b, s1, s2 = run(True, False, False), run(False, True, False), run(True, True, False), run(True, True, True)
    print(f"k_ols_wave alone {a:.4f}  k_fft4096 alone {b:.4f}  one stream {s1:.4f}  two streams {s2:.4f} ms", flush=True)
