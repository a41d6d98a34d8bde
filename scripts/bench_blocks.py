#!/usr/bin/env python3
"""The four blocks of the path one at a time (device-resident stream of 2^26 samples, cfg2's parameters):
ms per call, GSamples/s and the fraction of the block's own HBM roofline (FreqShifter 16 B,
Filter 16 B, Downsampler 8 + 8/D B, Fourier 16 B per input sample)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radiorust_amd as rr

N, fs = 1 << 26, 200e6
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")


def run(name, bytes_per_sample, call, K=10):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"{name:44s} {dt*1e3:7.3f} ms  {N/dt/1e9:6.1f} GSamples/s  {bytes_per_sample*N/dt/1e9:5.0f} GB/s algorithmic "
          f"({bytes_per_sample} B/sample) = {100*bytes_per_sample*N/dt/8e12:4.1f} % of the HBM roofline")


sh = rr.FreqShifter.with_shift(25e6)
sh.set_stream(st)
run("FreqShifter 25 MHz", 16, lambda: sh.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), N))
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
for n in (64, 128, 256, 1024):
    fl = rr.Filter.new(lp)
    fl.set_stream(st)
    run(f"Filter n={n} lowpass 20 MHz", 16, lambda: fl.process_dev(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N))
for generic in (False, True):
    if generic:
        os.environ["RR_DOWNSAMPLER_GENERIC"] = "1"
    ds = rr.Downsampler.new(4096, 50e6, 40e6)
    ds.set_stream(st)
    run(f"Downsampler 200->50 MS/s L=120{' (k_fir)' if generic else ''}", 10, lambda: ds.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), N))
    if not generic:
        print("   kernel:", ds.last_kernel())
os.environ.pop("RR_DOWNSAMPLER_GENERIC", None)
ds8 = rr.Downsampler.new(4096, 48000.0, 40000.0)
ds8.set_stream(st)
run("Downsampler 384->48 kS/s L=288 (D=8)", 9, lambda: ds8.process_dev(384000.0, d_in.data_ptr(), N, d_out.data_ptr(), N))
ds8l = rr.Downsampler.new(4096, 48000.0, 43000.0)
ds8l.set_stream(st)
run("Downsampler 384->48 kS/s L=461 (D=8)", 9, lambda: ds8l.process_dev(384000.0, d_in.data_ptr(), N, d_out.data_ptr(), N))
for D16, fo16, bw16 in ((16, 102400.0, 81920.0), (64, 102400.0, 81920.0)):
    dsw = rr.Downsampler.new(4096, fo16, bw16)
    dsw.set_stream(st)
    run(f"Downsampler {D16}:1 (bandwidth 0.8)", 8 + 8 / D16, lambda: dsw.process_dev(fo16 * D16, d_in.data_ptr(), N, d_out.data_ptr(), N))
    print("   kernel:", dsw.last_kernel(), " L =", dsw.ir_len())
# the reference's own pipelines: bandwidth_meter/main.rs:56 (10 : 1, L = 145) and simple_receiver.rs:28 (8 : 3, L = 34)
for name, fin, fout, bw, bps in (("Downsampler 1024->102.4 kS/s L=145 (10:1)", 1024000.0, 102400.0, 60000.0, 8.8),
                                 ("Downsampler 1024->384 kS/s L=34 (8:3)", 1024000.0, 384000.0, 200000.0, 11.0),
                                 ("Downsampler 300->100 kS/s (3:1)", 300000.0, 100000.0, 60000.0, 8 + 8 / 3),
                                 ("Downsampler 48->32 kS/s (3:2)", 48000.0, 32000.0, 20000.0, 8 + 16 / 3),
                                 ("Downsampler 48->44.1 kS/s L=71 (160:147)", 48000.0, 44100.0, 40000.0, 8 + 8 * 147 / 160),
                                 ("Downsampler 1024->44.1 kS/s L=255 (10240:441)", 1024000.0, 44100.0, 20000.0, 8 + 8 * 441 / 10240),
                                 ("Downsampler 1024->44.1 kS/s L=436 (10240:441)", 1024000.0, 44100.0, 30000.0, 8 + 8 * 441 / 10240)):
    for generic in (False, True):
        if generic:
            os.environ["RR_DOWNSAMPLER_GENERIC"] = "1"
        dsx = rr.Downsampler.new(4096, fout, bw)
        dsx.set_stream(st)
        run(name + (" (k_fir)" if generic else ""), round(bps, 2), lambda: dsx.process_dev(fin, d_in.data_ptr(), N, d_out.data_ptr(), N),
            K=3 if generic else 10)
        if not generic:
            print("   kernel:", dsx.last_kernel(), " L =", dsx.ir_len())
        os.environ.pop("RR_DOWNSAMPLER_GENERIC", None)
fo = rr.Fourier.with_window(rr.Kaiser.with_null_at_bin(2.0))
fo.set_stream(st)
run("Fourier 4096 Kaiser(null@2)", 16, lambda: fo.process_dev(4096, d_in.data_ptr(), N, d_out.data_ptr(), N))
fm = rr.FmDemod(75000.0)
fm.set_stream(st)
run("FmDemod", 16, lambda: fm.process_dev(fs, d_in.data_ptr(), N, d_out.data_ptr(), N))
up = rr.Upsampler.new(4096, 200e6, 40e6)
up.set_stream(st)
NU, NO = N // 4, N
def up_call():
    up.process_dev(50e6, d_in.data_ptr(), NU, d_out.data_ptr(), NO)
N = NU  # (the rates of this line are per input sample: 8 B in + 32 B out)
run("Upsampler 50->200 MS/s", 40, up_call)
# 44 100 -> 48 000 (147 : 160): the schedule in closed form on the device (k_upsample_closed); RR_UPSAMPLER_GENERIC=1: a list from the host per call
for generic in (False, True):
    if generic:
        os.environ["RR_UPSAMPLER_GENERIC"] = "1"
    up2 = rr.Upsampler.new(4096, 48000.0, 40000.0)
    up2.set_stream(st)
    N = 1 << 24
    n_o = N * 160 // 147 + 16
    run(f"Upsampler 44.1->48 kS/s{' (list from the host)' if generic else ''}", round(8 + 8 * 160 / 147, 2), lambda: up2.process_dev(44100.0, d_in.data_ptr(), N, d_out.data_ptr(), n_o), K=3)
os.environ.pop("RR_UPSAMPLER_GENERIC", None)
N = NO
# the reference example's analysis stage (bandwidth_meter/main.rs:66-69): chunks of 1024, Overlapper(4), Fourier with
# Kaiser(null at bin 4): one 4096-point spectrum per 1024 new samples: 8 B in + 32 B out per input sample
NS = N // 4
d_big = torch.empty(N + 8192, dtype=torch.complex64, device="cuda")
sf = rr.Stft(1024, 4, rr.Kaiser.with_null_at_bin(4.0))
sf.set_stream(st)
N = NS
run("Stft 1024 x 4 (example's analysis)", 40, lambda: sf.process_dev(d_in.data_ptr(), NS, d_big.data_ptr(), NO + 8192))
N = NO
# chain shapes beside cfg2's: other decimations, Fourier lengths and response lengths; roofline = 8 + 8 / D B
lp = lambda b, f: 1.0 if abs(f) <= 20e6 else 0.0
for name, kw, D in (("chain 4:1 / FFT 4096, Lc 183 (cfg2)", dict(shift=25e6, filter_len=64, output_rate=50e6, bandwidth=40e6, fft_len=4096), 4),
                    ("chain 4:1 / FFT 4096, general NCO (2469/40000)", dict(shift=12.345e6, precision=1e3, filter_len=64, output_rate=50e6, bandwidth=40e6, fft_len=4096), 4),
                    ("chain 4:1 / FFT 4096, Lc 123 (V = 128)", dict(shift=25e6, filter_len=64, output_rate=50e6, bandwidth=30e6, fft_len=4096), 4),
                    ("chain 4:1 / FFT 4096, Lc 59 (V = 64)", dict(shift=25e6, filter_len=20, output_rate=50e6, bandwidth=20e6, fft_len=4096), 4),
                    ("chain 8:1 / FFT 1024", dict(shift=25e6, filter_len=64, output_rate=25e6, bandwidth=20e6, fft_len=1024), 8),
                    ("chain 2:1 / FFT 8192", dict(shift=25e6, filter_len=64, output_rate=100e6, bandwidth=80e6, fft_len=8192), 2),
                    ("chain 4:1 / FFT 1024", dict(shift=25e6, filter_len=64, output_rate=50e6, bandwidth=40e6, fft_len=1024), 4),
                    ("chain 8:1 / FFT 4096", dict(shift=25e6, filter_len=64, output_rate=25e6, bandwidth=20e6, fft_len=4096), 8),
                    ("chain 10:1 / FFT 4096 (the example's ratio)", dict(shift=12.5e6, filter_len=64, output_rate=20e6, bandwidth=12e6, fft_len=4096), 10),
                    ("chain 5:1 / FFT 4096", dict(shift=12.5e6, filter_len=64, output_rate=40e6, bandwidth=30e6, fft_len=4096), 5),
                    ("chain 16:1 / FFT 4096 (k_ols_wg)", dict(shift=12.345e6, precision=1e3, filter_len=64, output_rate=12.5e6, bandwidth=10e6, fft_len=4096), 16),
                    ("chain 20:1 / FFT 1024 (k_ols_wg)", dict(shift=12.5e6, filter_len=64, output_rate=10e6, bandwidth=8e6, fft_len=1024), 20)):
    ch = rr.Chain(freq_resp=lp, fft_window=rr.Kaiser.with_null_at_bin(2.0), **kw)
    ch.set_stream(st)
    cap = N // D + 2 * kw["fft_len"]
    co = torch.empty(cap, dtype=torch.complex64, device="cuda")
    run(name, 8 + 8 / D, lambda: ch.process_dev(fs, d_in.data_ptr(), N, co.data_ptr(), cap))
    print("   kernel:", ch.last_path_kernel(), " mixer folded:", ch.last_path_mixer_folded())
    del ch, co
