#!/usr/bin/env python3
"""Chain throughput against the number of samples per call (device-resident input, back-to-back calls).
usage: callsize_probe.py [log2 sizes ..]            one chain
       callsize_probe.py bank K [log2 sizes ..]     a ChainBank of K channels in lockstep (aggregate rate over all channels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
fs = 200e6
st = torch.cuda.current_stream().cuda_stream
ARGS = sys.argv[1:]
BANK = 0
if ARGS and ARGS[0] == "bank":
    BANK = int(ARGS[1])
    ARGS = ARGS[2:]
LGS = [int(a) for a in ARGS] or ([12, 14, 16, 18, 20] if BANK else [14, 16, 18, 20, 22, 24, 26])  # (up to 28: 2 GiB of input)
N = min(1 << 28, (BANK or 1) << max(LGS))
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N // 4 + 8192, dtype=torch.complex64, device="cuda")
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
if BANK:
    # K channels side by side in the buffers (channel k at k * n); every call gives each channel n samples
    for lg in LGS:
        n = 1 << lg
        K = min(BANK, N // n)
        bank = rr.ChainBank(K, shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                            fft_window=rr.Kaiser.with_null_at_bin(2.0))
        bank.set_stream(st)
        ocap = n // 4 + 8192
        d_o = torch.empty(K * ocap, dtype=torch.complex64, device="cuda")
        calls = max(100, min(5000, (1 << 31) // (n * K)))
        for _ in range(min(calls, 200)): bank.process_dev(fs, d_in.data_ptr(), n, n, d_o.data_ptr(), ocap, ocap)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(calls): bank.process_dev(fs, d_in.data_ptr(), n, n, d_o.data_ptr(), ocap, ocap)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / calls
        print(f"bank of {K} channels x 2^{lg} samples per call: {dt*1e6:9.1f} us per call = {K*n/dt/1e9:7.1f} GSamples/s aggregate "
              f"(lockstep: {bank.last_path_lockstep()})")
        del bank, d_o
    sys.exit(0)
for lg in LGS:
    n = 1 << lg
    ch = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                  fft_window=rr.Kaiser.with_null_at_bin(2.0))
    calls = max(50 if lg > 26 else 200, min(20000, (1 << 30) // n))
    for _ in range(min(calls, 300)): ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel())
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(calls): ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / calls
    print(f"2^{lg} samples per call: {dt*1e6:9.1f} us per call = {n/dt/1e9:7.1f} GSamples/s ({ch.last_path_kernel()})")
