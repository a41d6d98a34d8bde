#!/usr/bin/env python3
"""Throughput of the paths round 2 added beside the headline ones (device-resident input, ms per call and GSamples/s):
oversampled / any-bin channelizers, Filters beyond 2048 taps (partitions), Fourier beyond one LDS tile (four-step),
Bluestein lengths, the f64 blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radiorust_amd as rr

st = torch.cuda.current_stream().cuda_stream


def timed(fn, k=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k


def line(name, n, dt, bps):
    print(f"{name:58s} {dt*1e3:8.3f} ms  {n/dt/1e9:7.1f} GSamples/s  {100*bps*n/dt/8e12:5.1f} % of {bps:g} B/sample", flush=True)


N = 1 << 26
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(2 * N, dtype=torch.complex64, device="cuda")

for M, P, hop in ((256, 4, 256), (256, 4, 128), (256, 4, 64), (100, 4, 100), (1000, 2, 1000), (1024, 8, 1024), (512, 4, 512), (2048, 4, 2048), (4096, 4, 4096)):
    c = rr.Channelizer(M, P, hop=hop)
    c.set_stream(st)
    n = N // hop * hop
    if hop < M:
        n = (1 << 24) // hop * hop
    outs = []
    dt = timed(lambda: outs.append(c.process_dev(d_in.data_ptr(), n, d_out.data_ptr(), 2 * N)))
    line(f"Channelizer {M} bins x {P} taps/branch, hop {hop}", n, dt, 8 + 8 * M / hop)

lp = lambda cut: (lambda b, f: 1.0 if abs(f) <= cut else 0.0)
for nt, conv_min in ((1024, None), (2048, None), (4096, None), (8192, None), (16384, None), (32768, None), (65536, None),
                     (4096, 2049), (8192, 2049), (16384, 1 << 20), (32768, 1 << 20)):
    # (kernel 2: k_filter_blk4096, partitions of 2048 taps beyond 2048; kernel 4: overlap-save through the tile passes)
    if conv_min is None:
        os.environ.pop("RR_FILTER_CONV_MIN", None)
    else:
        os.environ["RR_FILTER_CONV_MIN"] = str(conv_min)
    g = rr.Filter.new(lp(0.2e9))
    g.set_stream(st)
    n = N // nt * nt
    dt = timed(lambda: g.process_dev(2e9, nt, d_in.data_ptr(), n, d_out.data_ptr(), n), k=3)
    line(f"Filter n = {nt} (kernel {g.last_kernel()}{'' if conv_min is None else ', RR_FILTER_CONV_MIN=' + str(conv_min)})", n, dt, 16)
os.environ.pop("RR_FILTER_CONV_MIN", None)
# the block length of k_filter_blkbig (kernel 5) against the default's; RR_FILTER_KERNEL=parts: the partitions of 2048 taps
for nt, blk, kern in ((1536, 8192, None), (2048, 8192, None), (2048, 16384, None), (3072, 8192, None), (3072, 16384, None), (4096, 8192, None), (4096, None, "parts"), (8192, None, "parts")):
    os.environ.pop("RR_FILTER_BLOCK", None)
    os.environ.pop("RR_FILTER_KERNEL", None)
    if blk:
        os.environ["RR_FILTER_BLOCK"] = str(blk)
    if kern:
        os.environ["RR_FILTER_KERNEL"] = kern
    g = rr.Filter.new(lp(0.2e9))
    g.set_stream(st)
    n = N // nt * nt
    dt = timed(lambda: g.process_dev(2e9, nt, d_in.data_ptr(), n, d_out.data_ptr(), n), k=3)
    line(f"Filter n = {nt} (kernel {g.last_kernel()}, " + (f"RR_FILTER_BLOCK={blk}" if blk else f"RR_FILTER_KERNEL={kern}") + ")", n, dt, 16)
os.environ.pop("RR_FILTER_BLOCK", None)
os.environ.pop("RR_FILTER_KERNEL", None)

for nf in (8192, 16384, 32768, 65536, 96, 300, 500, 720, 1000, 1001, 1536, 1999, 2000, 3000, 3001, 4004, 4800, 5000, 5003, 8000, 8191, 20000, 250000, 20011):
    g = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
    g.set_stream(st)
    n = min(N, 1 << 24) // nf * nf
    dt = timed(lambda: g.process_dev(nf, d_in.data_ptr(), n, d_out.data_ptr(), n))
    line(f"Fourier n = {nf}", n, dt, 16)

# f64
N2 = 1 << 24
d64 = torch.randn(N2, dtype=torch.complex128, device="cuda")
o64 = torch.empty(N2, dtype=torch.complex128, device="cuda")
g = rr.FreqShifter.with_shift(25e6, dtype=np.float64); g.set_stream(st)
dt = timed(lambda: g.process_dev(200e6, d64.data_ptr(), N2, o64.data_ptr(), N2)); line("f64 FreqShifter", N2, dt, 32)
g = rr.Filter.new(lp(20e6), dtype=np.float64); g.set_stream(st)
dt = timed(lambda: g.process_dev(200e6, 64, d64.data_ptr(), N2, o64.data_ptr(), N2)); line("f64 Filter n = 64", N2, dt, 32)
g = rr.Downsampler.new(4096, 50e6, 40e6, dtype=np.float64); g.set_stream(st)
dt = timed(lambda: g.process_dev(200e6, d64.data_ptr(), N2, o64.data_ptr(), N2)); line("f64 Downsampler 4 : 1", N2, dt, 20)
g = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0), dtype=np.float64); g.set_stream(st)
dt = timed(lambda: g.process_dev(4096, d64.data_ptr(), N2, o64.data_ptr(), N2)); line("f64 Fourier 4096", N2, dt, 32)
for nf in (1000, 1999, 20011):
    g = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0), dtype=np.float64); g.set_stream(st)
    n = N2 // nf * nf
    dt = timed(lambda: g.process_dev(nf, d64.data_ptr(), n, o64.data_ptr(), n)); line(f"f64 Fourier {nf} ({rr.fourier_route(nf, np.float64)})", n, dt, 32)
# the f64 chain (cfg2's parameters): mixer + Filter + Downsampler as one pass of k_decim_poly_f64, then k_fft4096_f64
lp20 = lp(20e6)
for fused in (True, False):
    ch = rr.Chain(shift=25e6, filter_len=64, freq_resp=lp20, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                  fft_window=rr.Kaiser.with_null_at_bin(2.0), dtype=np.float64, allow_fused=fused)
    ch.set_stream(st)
    co = torch.empty(N2 // 4 + 8192, dtype=torch.complex128, device="cuda")
    dt = timed(lambda: ch.process_dev(200e6, d64.data_ptr(), N2, co.data_ptr(), co.numel()))
    line(f"f64 chain cfg2 ({'fused front end: ' + ch.last_path_kernel() if fused else 'block by block'})", N2, dt, 20)
