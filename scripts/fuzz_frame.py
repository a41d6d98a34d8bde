#!/usr/bin/env python3
"""Randomised parity sweep of the fused frame kernel (k_ols_frame: the whole chain in one kernel) against the f64
oracle: the shapes it is compiled for (4 : 1, 4096- or 1024-point Fourier stage, combined response of up to 193 taps), random
shifts and NCO periods, real / one-sided / complex responses, ragged call sizes.  RR_FUSED_KERNEL=olsf is set here,
so that it also runs on the short calls of a test stream (by default it takes calls of 2^23 samples and more).
usage: fuzz_frame.py [cases] [seed]"""
import os, sys
os.environ["RR_FUSED_KERNEL"] = "olsf"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radiorust_amd as rr
from oracle import rr_oracle as o

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
worst, used = 0.0, {}
for case in range(cases):
    fs, out_rate = 200e6, 50e6
    bw = float(rng.choice([40e6, 30e6, 20e6]))     # Downsampler: L = 120 / 60 / 40
    L = {40e6: 120, 30e6: 60, 20e6: 40}[bw]
    filter_len = int(rng.integers(2, 194 - L + 1))  # Lc = L + n - 1 up to 193: overlaps of 64 / 128 / 192 samples
    precision = float(rng.choice([1.0, 1e3, 1e5, 12345.0]))
    shift = float(rng.uniform(-60e6, 60e6)) if rng.random() < 0.6 else float(rng.choice([25e6, 12.5e6, 0.0, -50e6]))
    cut = float(rng.uniform(2e6, 24e6))
    kind = int(rng.integers(0, 3))
    resp = [lambda b, f, c=cut: 1.0 if abs(f) <= c else 0.0,
            lambda b, f, c=cut: 1.0 if 0 <= f <= c else 0.0,
            lambda b, f, c=cut: complex(np.exp(-abs(f) / c), 0.3 * np.sign(f) * np.exp(-abs(f) / c))][kind]
    center = bool(rng.integers(0, 2))
    fft_len = int(rng.choice([4096, 1024]))  # (1024: a wave per frame)
    params = dict(shift=shift, filter_len=filter_len, freq_resp=resp, output_rate=out_rate, bandwidth=bw, fft_len=fft_len)
    n = int(rng.integers(1 << 16, 1 << 18))
    x = o.synth_iq(300 + case, 0, n)
    ref = o.run_chain(x, fs, flt=np.float64, fft_window=o.Kaiser.with_null_at_bin(2.0), precision=precision, center_dc=center, **params)[3]
    g = rr.Chain(**params, precision=precision, fft_window=rr.Kaiser.with_null_at_bin(2.0), center_dc=center)
    k = int(rng.integers(1, 7))
    cuts = sorted({0, n, *(int(v) for v in rng.integers(1, n, size=k))})
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        out += g.process(rr.Samples(fs, x[a:b]))
        nm = g.last_path_kernel() or "blocks"
        used[nm] = used.get(nm, 0) + 1
    assert len(out) == len(ref), (case, len(out), len(ref))
    for i, (s, r) in enumerate(zip(out, ref)):
        den = np.sum(np.abs(r) ** 2)
        e = float(np.sqrt(np.sum(np.abs(s.chunk.astype(np.complex128) - r) ** 2) / den)) if den > 0 else 0.0
        worst = max(worst, e)
        assert e <= 1e-5, (case, i, e, filter_len, shift, precision, cuts)
print(f"{cases} cases ok, worst relative RMS error {worst:.3g}, calls per path {used}")
