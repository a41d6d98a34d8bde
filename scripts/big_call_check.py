#!/usr/bin/env python3
"""One chain call of 2^28 samples (2 GiB in, 16383 spectra out): index arithmetic at a size the tests do not reach.
The first spectra against the C oracle, all of them against the same stream fed in 16 calls of 2^24 samples."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radiorust_amd as rr
from oracle import rr_oracle as o

n, fs = 1 << int(os.environ.get('LOG2N', '28')), 200e6
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
cfg = dict(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096)
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 3, 0, n, d_in.data_ptr())
g = rr.Chain(**cfg, fft_window=rr.Kaiser.with_null_at_bin(2.0))
g.set_stream(st)
frames = g.peek(fs, n)
d_out = torch.empty(frames * 4096, dtype=torch.complex64, device="cuda")
t = time.perf_counter()
got = g.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel())
torch.cuda.synchronize()
print(f"GPU: {frames} spectra in {(time.perf_counter() - t)*1e3:.1f} ms (first call: includes the block-by-block start), path {g.last_path_kernel()}")
assert got == frames * 4096
y = d_out.cpu().numpy().reshape(frames, 4096)
# (a) the first spectra against the C oracle (it keeps at most `max_frames` spectra of the whole run)
K = 512
x = d_in[: (K + 2) * 16384].cpu().numpy()
ref = o.run_chain_c(x, fs, fft_window=o.Kaiser.with_null_at_bin(2.0), threads=4, max_frames=K + 1, **cfg)[0][:K]  # (the runner's last kept slot is reused for the later frames)
e = np.sqrt(np.sum(np.abs(y[:K].astype(np.complex128) - ref) ** 2, axis=1) / np.sum(np.abs(ref.astype(np.complex128)) ** 2, axis=1))
print(f"first {K} spectra against the oracle: max relative RMS {e.max():.2e}"); bad = np.nonzero(e > 2e-6)[0]; print("bad frames:", bad[:20], len(bad))
assert e.max() <= 2e-6
# (b) every spectrum against the same stream fed in 16 calls of 2^24 samples (the size the tests and the bench run at)
g2 = rr.Chain(**cfg, fft_window=rr.Kaiser.with_null_at_bin(2.0))
g2.set_stream(st)
d_out2 = torch.empty_like(d_out)
off = wrote = 0
step = 1 << 24
while off < n:
    wrote += g2.process_dev(fs, d_in.data_ptr() + 8 * off, step, d_out2.data_ptr() + 8 * wrote, d_out2.numel() - wrote)
    off += step
torch.cuda.synchronize()
assert wrote == frames * 4096
y2 = d_out2.cpu().numpy().reshape(frames, 4096)
num = np.sum(np.abs(y.astype(np.complex128) - y2) ** 2, axis=1)
den = np.sum(np.abs(y2.astype(np.complex128)) ** 2, axis=1)
e2 = np.sqrt(num / den)
print(f"all {frames} spectra, one call against 16 calls: max relative RMS {e2.max():.2e} at frame {int(e2.argmax())}, mean {e2.mean():.2e}")
assert e2.max() <= 2e-6
print("ok")
