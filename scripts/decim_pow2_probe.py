"""The stand-alone Downsampler at 8 .. 64 : 1 (even ratios) with responses of 15 / 30 / 60 taps per period: ms per 2^26 samples.
RR_OLS_WG=0: k_decim_poly for the ratios from 10 : 1 on; =2: k_ols_wg whatever the length."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import radiorust_amd as rr
N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
def timed(call, K=10):
    for _ in range(3): call()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K): call()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / K
for D in (5, 6, 7, 8, 9, 10, 12, 15, 16, 20, 24, 32, 48, 64):
    for bwf in (0.6, 0.8, 0.9):
        fo = 102400.0
        ds = rr.Downsampler.new(4096, fo, fo * bwf); ds.set_stream(st)
        dt = timed(lambda: ds.process_dev(fo * D, d_in.data_ptr(), N, d_out.data_ptr(), N))
        print(f"D={D:2d} bw {bwf}: L={ds.ir_len():5d} kernel {ds.last_kernel()} {dt*1e3:.3f} ms = {100*(8+8/D)*N/dt/8e12:.1f} %", flush=True)
