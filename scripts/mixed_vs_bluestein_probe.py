import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import radiorust_amd as rr
N = 1 << 24
st = torch.cuda.current_stream().cuda_stream
d_in = torch.randn(N, dtype=torch.complex64, device="cuda"); d_out = torch.empty_like(d_in)
def timed(f, k=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/k
for nf in (2100, 2500, 2800, 3000, 3072, 3600, 4000, 4004, 4096-96, 4800, 5000, 6000, 6144, 6561, 7000, 7200, 8000):
    row = []
    for env in ({}, {"RR_FOURIER_MIXED": "2"}, {"RR_FOURIER_MIXED": "0"}):
        for k, v in env.items(): os.environ[k] = v
        try:
            g = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0)); g.set_stream(st)
            n = N // nf * nf
            dt = timed(lambda: g.process_dev(nf, d_in.data_ptr(), n, d_out.data_ptr(), n))
            row.append(f"{rr.fourier_route(nf)[:30]:30s} {dt*1e3*N/n:.3f}")
        except Exception as e:
            row.append("n/a")
        for k in env: os.environ.pop(k)
    print(f"{nf:5d}: default {row[0]} | mixed {row[1]} | bluestein {row[2]}", flush=True)
