#!/usr/bin/env python3
"""k_ols_wave2k (8 : 1, a wave per 2048-sample block) against k_ols_wave<8> (RR_OLSW_2K=0): the stand-alone Downsampler at two
response lengths and the chain's 8 : 1 shapes, ms per 2^26 samples.  RR_LIB picks a build variant (scripts/build_variant.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")


def timed(call, K=10):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / K


lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
for env in ("0", None):
    if env is None:
        os.environ.pop("RR_OLSW_2K", None)
    else:
        os.environ["RR_OLSW_2K"] = env
    row = []
    for bw in (30000.0, 40000.0, 43000.0):
        ds = rr.Downsampler.new(4096, 48000.0, bw)
        ds.set_stream(st)
        dt = timed(lambda: ds.process_dev(384000.0, d_in.data_ptr(), N, d_out.data_ptr(), N))
        row.append(f"L={ds.ir_len()} {dt * 1e3:.3f}")
    for fft_len in (1024, 4096):
        ch = rr.Chain(freq_resp=lp, fft_window=rr.Kaiser.with_null_at_bin(2.0), shift=25e6, filter_len=64, output_rate=25e6, bandwidth=20e6, fft_len=fft_len)
        ch.set_stream(st)
        cap = N // 8 + 2 * fft_len
        dt = timed(lambda: ch.process_dev(200e6, d_in.data_ptr(), N, d_out.data_ptr(), cap))
        row.append(f"chain/{fft_len} {dt * 1e3:.3f}")
    print(("k_ols_wave<8> " if env == "0" else "k_ols_wave2k  ") + os.environ.get("RR_LIB", "default")[-24:] + ": " + "  ".join(row), flush=True)
