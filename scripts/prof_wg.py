import os, sys
sys.path.insert(0, os.getcwd())
import torch
import radiorust_amd as rr
N = 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
for D in (8, 16, 32, 64):
    fo = 102400.0
    ds = rr.Downsampler.new(4096, fo, fo * 0.8); ds.set_stream(st)
    for _ in range(6):
        ds.process_dev(fo * D, d_in.data_ptr(), N, d_out.data_ptr(), N)
    torch.cuda.synchronize()
print("done")
