#!/usr/bin/env python3
"""Chain step time for NCO tables of different periods (1/8 -> 8 entries; 2469/40000 ->
40 000 entries = 320 KB; a period that does not divide 128 takes the general table walk of k_ols_wave)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
fs, n = 200e6, 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, n, d_in.data_ptr())
cap = (n // 4 // 4096 + 2) * 4096
d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
for shift, prec in ((25e6, 1.0), (12.5e6, 1.0), (12.345e6, 1e3), (12.345678e6, 1.0), (0.0, 1.0)):
    ch = rr.Chain(shift=shift, precision=prec, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                  fft_window=rr.Kaiser.with_null_at_bin(2.0))
    for _ in range(300): ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100): ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 100
    print(f"shift {shift/1e6:.6f} MHz precision {prec:g} Hz: {ch.last_path_kernel()} {dt*1e3:.4f} ms/step = {n/dt/1e9:.1f} GSamples/s")
