#!/usr/bin/env python3
"""The chain shapes of scripts/bench_blocks.py alone, for `rocprofv3 --kernel-trace --stats`: per-kernel averages of the
two-kernel shapes (k_ols_wave<D, ..> + the Fourier kernel).  RR_FRAME_GENFOLD=0 / RR_FRAME_MIXFOLD=0 for A/B runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

N, fs = 1 << 26, 200e6
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
only = sys.argv[1:]
for name, kw, D in (("8:1/1024", dict(shift=25e6, filter_len=64, output_rate=25e6, bandwidth=20e6, fft_len=1024), 8),
                    ("2:1/8192", dict(shift=25e6, filter_len=64, output_rate=100e6, bandwidth=80e6, fft_len=8192), 2),
                    ("4:1/1024", dict(shift=25e6, filter_len=64, output_rate=50e6, bandwidth=40e6, fft_len=1024), 4),
                    ("8:1/4096", dict(shift=25e6, filter_len=64, output_rate=25e6, bandwidth=20e6, fft_len=4096), 8),
                    ("8:1/1024g", dict(shift=12.345e6, precision=1e3, filter_len=64, output_rate=25e6, bandwidth=20e6, fft_len=1024), 8),
                    ("2:1/8192g", dict(shift=12.345e6, precision=1e3, filter_len=64, output_rate=100e6, bandwidth=80e6, fft_len=8192), 2)):
    if only and name not in only:
        continue
    ch = rr.Chain(freq_resp=lp, fft_window=rr.Kaiser.with_null_at_bin(2.0), **kw)
    ch.set_stream(st)
    cap = N // D + 2 * kw["fft_len"]
    co = torch.empty(cap, dtype=torch.complex64, device="cuda")
    for _ in range(23):
        ch.process_dev(fs, d_in.data_ptr(), N, co.data_ptr(), cap)
    torch.cuda.synchronize()
    print(name, ch.last_path_kernel(), ch.last_path_mixer_folded(), flush=True)
    del ch, co
