import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 1000):
    fo = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
    fo.set_stream(st)
    M = N // n * n
    if n == 1000: M = 1000 * 4000
    for _ in range(2): fo.process_dev(n, d_in.data_ptr(), M, d_out.data_ptr(), M)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): fo.process_dev(n, d_in.data_ptr(), M, d_out.data_ptr(), M)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"Fourier n={n}: {dt*1e3:.3f} ms per {M} samples = {M/dt/1e9:.1f} GSamples/s = {100*16*M/dt/8e12:.1f} %")
