#!/usr/bin/env python3
"""Randomised parity sweep of the round-2 kernels against the f64 oracle: Downsampler with random integral rate
pairs P : Q (k_decim_poly; ragged calls mixed with short ones), Filter with 386 .. 5000 taps (k_filter_blk4096,
one launch or accumulating partitions) and Fourier at random lengths up to 20000 in f32 and f64 (Bluestein,
four-step).  usage: fuzz_r2.py [cases] [seed]"""
import os, sys
from math import gcd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radiorust_amd as rr
from oracle import rr_oracle as o

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)


def rms(a, b):
    a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
    d = np.sum(np.abs(b) ** 2)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / d)) if d else float(np.max(np.abs(a - b), initial=0.0))


worst = {"down": 0.0, "filter": 0.0, "fourier32": 0.0, "fourier64": 0.0}
seen = {"down": set(), "filter": set()}
for case in range(cases):
    # ---- Downsampler, P : Q ------------------------------------------------------------------------
    while True:
        Q = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 7, 8]))
        P = int(rng.integers(Q + 1, 48)) if rng.random() < 0.8 else int(rng.choice([64, 100, 128, 200, 512]))
        if gcd(P, Q) == 1 and P > Q:
            break
    unit = float(rng.choice([1.0, 1000.0, 12800.0]))
    fi, fo = P * unit, Q * unit
    bw = fo * float(rng.uniform(0.3, 0.95))
    q = float(rng.choice([1.0, 2.0, 3.0]))
    n = int(rng.integers(9000, 200000))
    x = o.synth_iq(900 + case, 0, n)
    g, r = rr.Downsampler.with_quality(64, fo, bw, q), o.Downsampler(64, fo, bw, q, flt=np.float64)
    k = int(rng.integers(1, 7))
    cuts = sorted({0, n, *(int(v) for v in rng.integers(0, n + 1, size=k))})
    ys, rs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        y, yr = g.process_raw(fi, x[a:b]), r.process(fi, x[a:b].astype(np.complex128))
        seen["down"].add(g.last_kernel())
        assert len(y) == len(yr), ("down", case, P, Q)
        ys.append(y); rs.append(yr)
    e = rms(np.concatenate(ys), np.concatenate(rs))
    worst["down"] = max(worst["down"], e)
    assert e <= 1e-5, ("down", case, P, Q, g.ir_len(), e)
    # ---- Filter, long responses ------------------------------------------------------------------------
    nt = int(rng.choice([386, 500, 512, 1000, 1024, 1025, 2047, 2048, 2049, 3000, 4096, 5000, int(rng.integers(386, 5000))]))
    fs = 2e9
    cut = fs * float(rng.uniform(0.05, 0.45))
    resp = (lambda b, f: 1.0 if abs(f) <= cut else 0.0) if rng.random() < 0.7 else (lambda b, f: (1.0 + 0.5j) if 0 <= f <= cut else 0.0)
    ks = [int(rng.integers(1, 3)), -(-int(rng.integers(20000, 90000)) // nt), 1, -(-9000 // nt)]
    rng.shuffle(ks)
    total = sum(ks)
    x = o.synth_iq(1300 + case, 0, nt * total)
    g, r = rr.Filter.new(resp), o.Filter(resp, flt=np.float64)
    ref = [r.process(fs, x[i * nt:(i + 1) * nt].astype(np.complex128)) for i in range(total)]
    ref = np.concatenate([v for v in ref if v is not None])
    got, off = [], 0
    for kk in ks:
        d_in = torch.from_numpy(x[off * nt:(off + kk) * nt]).cuda()
        d_out = torch.empty_like(d_in)
        g.set_stream(torch.cuda.current_stream().cuda_stream)
        w = g.process_dev(fs, nt, d_in.data_ptr(), nt * kk, d_out.data_ptr(), nt * kk)
        torch.cuda.synchronize()
        seen["filter"].add(g.last_kernel())
        got.append(d_out.cpu().numpy()[:w])
        off += kk
    got = np.concatenate(got)
    assert len(got) == len(ref), ("filter", case, nt)
    e = rms(got, ref)
    worst["filter"] = max(worst["filter"], e)
    assert e <= 1e-5, ("filter", case, nt, ks, e)
    # ---- Fourier, any length ---------------------------------------------------------------------------
    nf = int(rng.choice([33, 100, 1000, 4097, 12000, 16384, 20000, 32768, int(rng.integers(32, 20000))]))
    chunks = int(rng.integers(1, 4))
    for flt, key, tol in ((np.float32, "fourier32", 1e-5), (np.float64, "fourier64", 1e-12)):
        cd = np.complex64 if flt is np.float32 else np.complex128
        xs = o.synth_iq(1700 + case, 0, nf * chunks).astype(cd)
        gw, ow = rr.Kaiser.with_null_at_bin(2.0), o.Kaiser.with_null_at_bin(2.0)
        g, r = rr.Fourier.with_window(gw, dtype=flt), o.Fourier(ow, flt=np.float64)
        for c in range(chunks):
            y = g.process(rr.Samples(1.0, xs[c * nf:(c + 1) * nf]))[0].chunk
            yr = r.process(xs[c * nf:(c + 1) * nf].astype(np.complex128))
            e = rms(y, yr)
            worst[key] = max(worst[key], e)
            assert e <= tol, (key, case, nf, e)
print(f"{cases} cases ok; worst: {worst}; kernels seen: {seen}")
