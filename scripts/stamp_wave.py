#!/usr/bin/env python3
"""RR_STAMP build + RR_FUSED_KERNEL=olsw: average s_memtime cycles per phase of a k_ols_wave block."""
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
import numpy as np
nb = 80664
arr = np.zeros(nb * 8, dtype=np.uint32)
L.rr_debug_read_wave_stamps.argtypes = [C.c_void_p, C.c_uint]
assert L.rr_debug_read_wave_stamps(arr.ctypes.data, nb) == 0
rec = arr.reshape(nb, 8)[100:-100].astype(np.int64)
names = ["prologue+load+mix", "pass0+exch1", "pass1", "exch2+pass2+H", "inverse", "store"]
tot = rec[:, :6].sum(axis=1)
print("blocks", len(rec), "fused", ch.last_path_fused())
for i, nm in enumerate(names):
    c = rec[:, i]
    print(f"{nm:20s} mean {c.mean():8.0f}  p10 {np.percentile(c,10):8.0f}  p50 {np.percentile(c,50):8.0f}  p90 {np.percentile(c,90):8.0f}  {100*c.sum()/tot.sum():5.1f}%")
print(f"total mean {tot.mean():.0f} p50 {np.percentile(tot,50):.0f} ticks/block")
start = rec[:, 6]
span = (start.max() - start.min()) & 0xffffffff
print("start-time span (ticks, low 32 bits):", span)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "wave_stamps.npy"), arr.reshape(nb, 8))
