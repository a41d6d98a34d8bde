#!/usr/bin/env python3
"""Randomised parity sweep of the long-call kernels of the stand-alone blocks: Downsampler with integer
ratios 2 / 4 / 8 / 16 / 32 / 64 (k_mix_fir_decim / k_ols_wave / k_ols_wave2k / k_ols_wg / k_ols_decim4 by ratio and L) and Filter with n <= 385 (k_filter_wave),
mixed with short calls, against the f64 oracle.  usage: fuzz_fast.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radiorust_amd as rr
from oracle import rr_oracle as o

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)


def rms(a, b):
    a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
    d = np.sum(np.abs(b) ** 2)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / d)) if d else float(np.max(np.abs(a - b), initial=0.0))


worst = {"down": 0.0, "filter": 0.0}
seen = {"down": set(), "filter": set()}
for case in range(cases):
    # ---- Downsampler -------------------------------------------------------------------------------
    D = int(rng.choice([2, 4, 4, 4, 8, 8, 16, 32, 64, 10, 12, 14, 20, 24, 30, 48, 62, 5, 7, 9, 15, 21, 33]))
    fo = float(rng.choice([48000.0, 50e6, 1.0]))
    fi = fo * D
    if rng.random() < 0.4:  # aim at the selection boundaries (L = ceil(fi / margin * q), margin = (fo - bw) / 2)
        L_t = int(rng.choice([110, 111, 112, 113, 384, 385, 386, 387, 60, 500]))
        bw = fo - 2.0 * fi * 3.0 / (L_t - 0.5)
        if bw <= 0.05 * fo:  # (the larger ratios: aim at lengths that scale with the ratio instead)
            bw = fo - 2.0 * fi * 3.0 / (L_t * D / 4 - 0.5)
        if bw <= 0.05 * fo:
            bw = fo * float(rng.uniform(0.3, 0.97))
    else:
        bw = fo * float(rng.uniform(0.3, 0.97))
    q = 3.0
    n = int(rng.integers(9000, 120000))
    x = o.synth_iq(400 + case, 0, n)
    g, r = rr.Downsampler.with_quality(64, fo, bw, q), o.Downsampler(64, fo, bw, q, flt=np.float64)
    k = int(rng.integers(1, 7))
    cuts = sorted({0, n, *(int(v) for v in rng.integers(0, n + 1, size=k))})
    ys, rs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        y, yr = g.process_raw(fi, x[a:b]), r.process(fi, x[a:b].astype(np.complex128))
        seen["down"].add(g.last_kernel())
        assert len(y) == len(yr), ("down", case)
        if len(y) > 64:
            e = rms(y, yr)
            assert e <= 1e-5, ("down piece", case, D, g.ir_len(), a, b, g.last_kernel(), e)
        ys.append(y); rs.append(yr)
    e = rms(np.concatenate(ys), np.concatenate(rs))
    worst["down"] = max(worst["down"], e)
    assert e <= 1e-5, ("down", case, D, g.ir_len(), e)
    # ---- Filter ------------------------------------------------------------------------------------
    nt = int(rng.choice([2, 3, 17, 33, 48, 64, 64, 65, 100, 128, 129, 200, 256, 257, 384, 385, int(rng.integers(2, 386))]))
    fs = 200e6
    cut = fs * float(rng.uniform(0.05, 0.45))
    resp = (lambda b, f: 1.0 if abs(f) <= cut else 0.0) if rng.random() < 0.7 else (lambda b, f: (1.0 + 0.5j) if 0 <= f <= cut else 0.0)
    chunks_big = -(-int(rng.integers(16384, 60000)) // nt)
    ks = [int(rng.integers(1, 4)), chunks_big, int(rng.integers(1, 3)), -(-17000 // nt), 1]
    rng.shuffle(ks)
    total = sum(ks)
    x = o.synth_iq(600 + case, 0, nt * total)
    g, r = rr.Filter.new(resp), o.Filter(resp, flt=np.float64)
    ref = [r.process(fs, x[i * nt:(i + 1) * nt].astype(np.complex128)) for i in range(total)]
    ref = np.concatenate([v for v in ref if v is not None])
    got, off = [], 0
    for kk in ks:
        # the host entry takes one chunk per call; the device entry whole runs of chunks
        import torch
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
    assert e <= 1e-5, ("filter", case, nt, e)
    pos = 0
    for kk, piece in zip(ks, range(len(ks))):
        ln = nt * (kk - (1 if piece == 0 else 0))
        if ln > 64:
            ep = rms(got[pos:pos + ln], ref[pos:pos + ln])
            assert ep <= 1e-5, ("filter piece", case, nt, piece, ep)
        pos += ln
print(f"{cases} cases ok; worst: {worst}; kernels seen: {seen}")
