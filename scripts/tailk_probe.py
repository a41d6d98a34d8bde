import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import radiorust_amd as rr
N, fs = 1 << 26, 200e6
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
def timed(call, K=20):
    for _ in range(5): call()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K): call()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / K
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
row = []
for n in (64, 128, 256):
    fl = rr.Filter.new(lp); fl.set_stream(st)
    row.append(f"Filter{n} {timed(lambda: fl.process_dev(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N))*1e3:.3f}")
for (fo, bw) in ((50e6, 40e6), (50e6, 46e6), (100e6, 80e6), (100e6, 94e6)):
    ds = rr.Downsampler.new(4096, fo, bw); ds.set_stream(st)
    row.append(f"Down{int(200e6/fo)}:1 L={ds.ir_len()} {timed(lambda: ds.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), N))*1e3:.3f}")
ch = rr.Chain(freq_resp=lp, fft_window=rr.Kaiser.with_null_at_bin(2.0), shift=25e6, filter_len=64, output_rate=100e6, bandwidth=80e6, fft_len=8192); ch.set_stream(st)
cap = N // 2 + 2 * 8192
co = torch.empty(cap, dtype=torch.complex64, device="cuda")
row.append(f"chain2:1 {timed(lambda: ch.process_dev(fs, d_in.data_ptr(), N, co.data_ptr(), cap))*1e3:.3f}")
print(os.environ.get("RR_LIB", "default")[-12:], "  ".join(row), flush=True)
