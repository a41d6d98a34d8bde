"""k_fft4096 on working sets from 128 to 4096 frames: does data that is still in L2 / MALL make the pass faster?"""
import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import radiorust_amd as rr
fo = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
st = torch.cuda.current_stream().cuda_stream
fo.set_stream(st)
for frames in (4096, 1024, 512, 256, 128):
    m = frames * 4096
    a = torch.randn(m, dtype=torch.complex64, device="cuda")
    b = torch.empty(m, dtype=torch.complex64, device="cuda")
    for _ in range(200): fo.process_dev(4096, a.data_ptr(), m, b.data_ptr(), m)
    torch.cuda.synchronize()
    K = 2000 * 4096 // frames // 8
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): fo.process_dev(4096, a.data_ptr(), m, b.data_ptr(), m)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / K * 1e3
    print(f"{frames} frames: {t*1e3:.1f} us per launch, {t/frames*1e6:.2f} ns/frame, {2*m*8/t/1e6:.0f} GB/s")
