#!/usr/bin/env python3
"""Runs the chain bench loop with a RR_STAMP build and prints per-phase cycle shares."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
L = rr._lib.lib()
fs, n = 200e6, 1 << 26
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
ch = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096, fft_window=rr.Kaiser.with_null_at_bin(2.0))
st = torch.cuda.current_stream().cuda_stream
ch.set_stream(st)
d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, n, d_in.data_ptr())
cap = (n // 4 // 4096 + 2) * 4096
d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
for _ in range(3):
    ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
L.rr_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
L.rr_debug_read_stamps(buf, 1)
K = 5
for _ in range(K):
    ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)
torch.cuda.synchronize()
L.rr_debug_read_stamps(buf, 0)
names = ["stage", "barrier1", "prefetch_issue", "fir_loop", "barrier2", "store"]
waves = buf[7]
tot = sum(buf[i] for i in range(6))
print("waves", waves, "fused", ch.last_path_fused())
for i, nm in enumerate(names):
    print(f"{nm:16s} {buf[i]/waves/16:10.0f} cycles/tile/wave  {100*buf[i]/tot:5.1f}%")
print(f"total {tot/waves/16:.0f} cycles/tile/wave")
