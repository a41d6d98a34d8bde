#!/usr/bin/env python3
"""Integer-ratio Downsamplers (2, 4, 8 : 1) through each kernel that can serve them: the default choice, k_ols_wave<D>
(RR_FUSED_KERNEL=olsw), the direct form k_mix_fir_decim (direct) and k_decim_poly (RR_DOWNSAMPLER_POLY=1).
Where pick_fused_kernel's thresholds come from.  2^26 device-resident samples per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
# (input rate, output rate, bandwidth): L = ceil(fin / ((fout - bw) / 2) * 3)
cases = [(96000.0, 48000.0, 30000.0), (96000.0, 48000.0, 40000.0), (96000.0, 48000.0, 44000.0), (96000.0, 48000.0, 46500.0),
         (200e6, 50e6, 30e6), (200e6, 50e6, 36e6), (200e6, 50e6, 40e6),
         (384000.0, 48000.0, 20000.0), (384000.0, 48000.0, 30000.0), (384000.0, 48000.0, 40000.0), (384000.0, 48000.0, 43000.0)]
modes = [("default", {}), ("olsw", {"RR_FUSED_KERNEL": "olsw"}), ("direct", {"RR_FUSED_KERNEL": "direct"}), ("poly", {"RR_DOWNSAMPLER_POLY": "1"})]
for fin, fout, bw in cases:
    line = []
    for name, env in modes:
        for k in ("RR_FUSED_KERNEL", "RR_DOWNSAMPLER_POLY"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ds = rr.Downsampler.new(4096, fout, bw)
        ds.set_stream(st)
        for _ in range(3):
            ds.process_dev(fin, d_in.data_ptr(), N, d_out.data_ptr(), N)
        torch.cuda.synchronize()
        K = 10
        t = time.perf_counter()
        for _ in range(K):
            ds.process_dev(fin, d_in.data_ptr(), N, d_out.data_ptr(), N)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / K
        line.append(f"{name} {dt*1e3:.3f} (k{ds.last_kernel()})")
        L = ds.ir_len()
    print(f"{int(fin / fout)} : 1, L = {L}: " + "  ".join(line), flush=True)
