import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import radiorust_amd as rr
st = torch.cuda.current_stream().cuda_stream
N = 1 << 22
d_in = torch.randn(N, dtype=torch.complex64, device="cuda"); d_out = torch.empty(N * 17, dtype=torch.complex64, device="cuda")
d_in64 = torch.randn(N, dtype=torch.complex128, device="cuda"); d_out64 = torch.empty(N * 9, dtype=torch.complex128, device="cuda")
def timed(f, k=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/k
for env in (None, "1"):
    if env: os.environ["RR_UPSAMPLER_GENERIC"] = env
    else: os.environ.pop("RR_UPSAMPLER_GENERIC", None)
    row = []
    for U in (8, 10, 16):
        up = rr.Upsampler.new(4096, 102400.0 * U, 40000.0); up.set_stream(st)
        row.append(f"f32 U={U} {timed(lambda: up.process_dev(102400.0, d_in.data_ptr(), N, d_out.data_ptr(), N * U))*1e3:.3f}")
    for U in (4, 8):
        up = rr.Upsampler.new(4096, 102400.0 * U, 40000.0, dtype=np.float64); up.set_stream(st)
        row.append(f"f64 U={U} {timed(lambda: up.process_dev(102400.0, d_in64.data_ptr(), N, d_out64.data_ptr(), N * U))*1e3:.3f}")
    print("generic" if env else "default", "  ".join(row), flush=True)
