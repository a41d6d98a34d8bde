#!/usr/bin/env python3
"""k_filter_blkbig<8192> (responses of 1025 .. 4096 taps in blocks of 8192 points; RR_FILTER_BLOCK=8192 forces it) beside the
default choice per length: ms per 2^26 samples.  RR_LIB picks a build variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
lp = lambda b, f: 1.0 if abs(f) <= 0.2e9 else 0.0
def timed(call, K=10):
    for _ in range(3): call()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K): call()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / K
row = []
for nt in (1280, 1536, 2048, 3072, 4096):
    for blk in (None, "8192"):
        if blk: os.environ["RR_FILTER_BLOCK"] = blk
        else: os.environ.pop("RR_FILTER_BLOCK", None)
        g = rr.Filter.new(lp); g.set_stream(st)
        n = N // nt * nt
        dt = timed(lambda: g.process_dev(2e9, nt, d_in.data_ptr(), n, d_out.data_ptr(), n))
        row.append(f"n={nt}{'/8192' if blk else ''} k{g.last_kernel()} {dt*1e3:.3f}")
os.environ.pop("RR_FILTER_BLOCK", None)
print(os.environ.get("RR_LIB", "default")[-16:], "  ".join(row), flush=True)
