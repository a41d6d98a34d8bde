#!/usr/bin/env python3
"""Randomised parity sweep of the Chain (all fused kernels the selector may pick) against the f64 oracle.
usage: fuzz_chain.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radiorust_amd as rr
from oracle import rr_oracle as o

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst, used = 0.0, {}
for case in range(cases):
    fs = 200e6
    filter_len = int(rng.choice([64, 128, 256]))
    # (20 MS/s and 40 MS/s: 10 : 1 and 5 : 1 - the chain's front end through k_decim_poly)
    out_rate, bw = [(50e6, 30e6), (50e6, 40e6), (50e6, 44e6), (100e6, 80e6), (25e6, 20e6), (20e6, 12e6), (40e6, 30e6)][int(rng.integers(0, 7))]
    precision = float(rng.choice([1.0, 1e3, 1e5, 12345.0]))
    shift = float(rng.uniform(-60e6, 60e6)) if rng.random() < 0.7 else float(rng.choice([25e6, 12.5e6, 0.0, -50e6]))
    cut = float(rng.uniform(2e6, 24e6))
    kind = int(rng.integers(0, 3))
    resp = [lambda b, f, c=cut: 1.0 if abs(f) <= c else 0.0,
            lambda b, f, c=cut: 1.0 if 0 <= f <= c else 0.0,
            lambda b, f, c=cut: complex(np.exp(-abs(f) / c), 0.3 * np.sign(f) * np.exp(-abs(f) / c))][kind]
    fft_len = int(rng.choice([4096, 1024, 2048]))
    center = bool(rng.integers(0, 2))
    params = dict(shift=shift, filter_len=filter_len, freq_resp=resp, output_rate=out_rate, bandwidth=bw, fft_len=fft_len)
    n = int(rng.integers(1 << 15, 1 << 17))
    x = o.synth_iq(100 + case, 0, n)
    ref = o.run_chain(x, fs, flt=np.float64, fft_window=o.Kaiser.with_null_at_bin(2.0), precision=precision, center_dc=center, **params)[3]
    g = rr.Chain(**params, precision=precision, fft_window=rr.Kaiser.with_null_at_bin(2.0), center_dc=center)
    k = int(rng.integers(1, 9))
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
        assert e <= 1e-5, (case, i, e, params, precision, cuts)
print(f"{cases} cases ok, worst relative RMS error {worst:.3g}, calls per path {used}")
