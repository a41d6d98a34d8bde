import sys, time
sys.path.insert(0, '/root/repo')
import torch
import radiorust_amd as rr
# 4096 frames of 4096 bins whose inputs overlap almost completely (hop 1): the reads hit L1/L2, the
# writes (134 MB) go to HBM -> the write-bound speed of the Fourier stage
g = rr.Stft(1, 4096, rr.Kaiser.with_null_at_bin(2.0))
n = 4096 + 4095
a = torch.randn(n, dtype=torch.complex64, device="cuda")
b = torch.empty(4096 * 4096, dtype=torch.complex64, device="cuda")
g.process_dev(a.data_ptr(), 4095, b.data_ptr(), b.numel())  # fill the history
for _ in range(50): g.process_dev(a.data_ptr(), 4096, b.data_ptr(), b.numel())
torch.cuda.synchronize()
K = 200
t = time.perf_counter()
for _ in range(K): w = g.process_dev(a.data_ptr(), 4096, b.data_ptr(), b.numel())
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"wrote {w} bins per call; {dt*1e3:.4f} ms per 4096 frames (reads cached) = {w*8/dt/1e9:.0f} GB/s of writes")
