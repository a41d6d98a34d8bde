import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import radiorust_amd as rr
fs, n = 200e6, 1 << 26
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, n, d_in.data_ptr())
cap = (n // 4 // 4096 + 2) * 4096
d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
for flen, bw in ((64, 40e6), (128, 44e6), (256, 40e6), (256, 44e6)):
    for kern in ("olsw", "ols", "direct"):
        os.environ["RR_FUSED_KERNEL"] = kern
        lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
        ch = rr.Chain(shift=25e6, filter_len=flen, freq_resp=lp, output_rate=50e6, bandwidth=bw, fft_len=4096,
                      fft_window=rr.Kaiser.with_null_at_bin(2.0))
        for _ in range(300): ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(100): ch.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 100
        print(f"filter_len {flen} bw {bw/1e6:.0f} MHz forced {kern:6s} ran {ch.last_path_kernel():16s} {dt*1e3:.4f} ms/step")
