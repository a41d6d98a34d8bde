import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
for n in (64, 128, 192, 256, 320, 384):
    for forced in (None, "ols4096"):
        if forced: os.environ["RR_FILTER_KERNEL"] = forced
        else: os.environ.pop("RR_FILTER_KERNEL", None)
        fl = rr.Filter.new(lp); fl.set_stream(st)
        M = N // n * n
        for _ in range(3): fl.process_dev(200e6, n, d_in.data_ptr(), M, d_out.data_ptr(), M)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fl.process_dev(200e6, n, d_in.data_ptr(), M, d_out.data_ptr(), M)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(f"n={n} {forced or 'default'} kernel {fl.last_kernel()}: {dt*1e3:.3f} ms = {100*16*M/dt/8e12:.1f} %", flush=True)
