#!/usr/bin/env python3
"""The reference's bandwidth_meter pipeline (examples/bandwidth_meter/main.rs:53-69) device-resident through rr_meter:
1.024 MS/s -> shift -> 10 : 1 (L = 145) -> Filter 1024 -> Overlapper(4) -> Fourier 4096, on 2^26 input samples per call;
against the same stages called block by block (FreqShifter, Downsampler, Filter, Stft handles, device pointers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr
N, fs, out_rate, bw, q = 1 << 26, 1024000.0, 102400.0, 60e3, 4
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
resp = lambda b, f: 1.0 if abs(f) <= bw / 2 else 0.0
m = rr.Meter(shift=12.5e3, output_rate=out_rate, bandwidth=bw, chunk_len=1024, freq_resp=resp, overlap=q,
             fft_window=rr.Kaiser.with_null_at_bin(float(q)))
m.set_stream(st)
cap = (N // 10 // 1024 + 8) * 4096
d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")
def timeit(fn, K=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / K
dt = timeit(lambda: m.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), cap))
alg = 8 + 0.8 + 0.8 + 0.8 + 0.8 + 3.2  # in, decimated out/in, filtered out/in, spectra out (x4 overlap), per input sample
print(f"rr_meter (front end fused: {m.front_fused()}): {dt*1e3:.3f} ms per 2^26 input samples = {N/dt/1e9:.1f} GSamples/s; "
      f"{100*alg*N/dt/8e12:.1f} % of the unfused-stage traffic model {alg:.1f} B/sample, {100*(8+3.2)*N/dt/8e12:.1f} % of in + spectra out (11.2 B)")
# the example's last step (main.rs:78): metering::bandwidth per spectrum.  (a) fused behind the transform, spectra still written;
# (b) fused, spectra not written at all (the example never looks at them); (c) the serial kernel behind the pipeline (round 2's form)
import ctypes as C
frames = cap // 4096
d_bw = torch.empty(frames, dtype=torch.float64, device="cuda")
L = rr._lib.lib()
def plain():
    m.set_metering(0.01, 0, 0)
    return timeit(lambda: m.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), cap), K=30)
def fused():
    m.set_metering(0.01, d_bw.data_ptr(), frames)
    return timeit(lambda: m.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), cap), K=30)
def fused_nostore():
    m.set_metering(0.01, d_bw.data_ptr(), frames, store_spectra=False)
    return timeit(lambda: m.process_dev(fs, d_in.data_ptr(), N, 0, 0), K=30)
# (the three forms in turn, five rounds, medians: the card's power state moves single runs by several per cent)
rounds = [(plain(), fused(), fused_nostore()) for _ in range(5)]
med = lambda i: sorted(r[i] for r in rounds)[len(rounds) // 2]
dt0, dt_a, dt_b = med(0), med(1), med(2)
m.set_metering(0.01, 0, 0)
def serial():
    w = m.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), cap)
    rr._lib.check(L.rr_bandwidth_dev(0, 0, C.c_void_p(st), 0.01, out_rate, d_out.data_ptr(), 4096, w // 4096, d_bw.data_ptr()))
dt_c = timeit(serial, K=3)
print(f"pipeline alone {dt0*1e3:.4f} ms (median of 5 rounds of 30 calls, the forms in turn); + metering::bandwidth per spectrum: "
      f"fused epilogue {dt_a*1e3:.4f} ms = {dt_a/dt0:.3f} x; fused, spectra not written {dt_b*1e3:.4f} ms = {dt_b/dt0:.3f} x; "
      f"serial kernel behind the pipeline {dt_c*1e3:.3f} ms = {dt_c/dt0:.3f} x")
sh = rr.FreqShifter.with_shift(12.5e3); sh.set_stream(st)
ds = rr.Downsampler.new(1024, out_rate, bw); ds.set_stream(st)
fl = rr.Filter.new(resp); fl.set_stream(st)
sf = rr.Stft(1024, q, rr.Kaiser.with_null_at_bin(float(q))); sf.set_stream(st)
d_a = torch.empty(N, dtype=torch.complex64, device="cuda")
d_b = torch.empty(N // 8, dtype=torch.complex64, device="cuda")
d_c = torch.empty(N // 8, dtype=torch.complex64, device="cuda")
def blocks():
    sh.process_dev(fs, d_in.data_ptr(), N, d_a.data_ptr(), N)
    k = ds.process_dev(fs, d_a.data_ptr(), N, d_b.data_ptr(), N // 8)
    k = k // 1024 * 1024
    k2 = fl.process_dev(out_rate, 1024, d_b.data_ptr(), k, d_c.data_ptr(), N // 8)
    sf.process_dev(d_c.data_ptr(), k2, d_out.data_ptr(), cap)
dt2 = timeit(blocks)
print(f"the four handles one after the other (device pointers): {dt2*1e3:.3f} ms = {N/dt2/1e9:.1f} GSamples/s")
