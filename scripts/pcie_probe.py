#!/usr/bin/env python3
"""PCIe-inclusive rate of the chain: host buffers in, spectra out (pinned vs pageable memory)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radiorust_amd as rr
from oracle import rr_oracle as o  # only for the synthetic input
L = rr._lib.lib()
fs, n = 200e6, 1 << int(os.environ.get('LOG2N', '24'))
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
g = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp, output_rate=50e6, bandwidth=40e6, fft_len=4096,
             fft_window=rr.Kaiser.with_null_at_bin(2.0))
g._ensure_design(fs)
x = o.synth_iq(1, 0, n)
cap = n // 4 + 8192
for pinned in (True, False):
    if pinned:
        p_in, p_out = C.c_void_p(), C.c_void_p()
        assert L.rr_host_alloc(n * 8, C.byref(p_in)) == 0 and L.rr_host_alloc(cap * 8, C.byref(p_out)) == 0
        C.memmove(p_in, x.ctypes.data, n * 8)
        a_in, a_out = p_in, p_out
    else:
        out = np.empty(cap, dtype=np.complex64)
        a_in, a_out = C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data)
    cnt = C.c_size_t()
    for _ in range(3):
        assert L.rr_chain_enqueue(g._h, fs, a_in, n, a_out, cap, C.byref(cnt)) == 0
    g.wait()
    K = max(10, (10 << 24) // n)
    t = time.perf_counter()
    for _ in range(K):
        assert L.rr_chain_enqueue(g._h, fs, a_in, n, a_out, cap, C.byref(cnt)) == 0
    g.wait()
    dt = (time.perf_counter() - t) / K
    print(f"{'pinned' if pinned else 'pageable'} host buffers: {dt*1e3:.2f} ms per {n} samples = {n/dt/1e9:.2f} GSamples/s "
          f"({(n*8 + cnt.value*8)/dt/1e9:.1f} GB/s over the link)")
