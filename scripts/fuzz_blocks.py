#!/usr/bin/env python3
"""Randomised parity sweep of the stand-alone blocks (Upsampler, Downsampler, FmDemod, Stft, Channelizer)
against the oracle.  usage: fuzz_blocks.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radiorust_amd as rr
from oracle import rr_oracle as o

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)


def rms(a, b):
    a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
    d = np.sum(np.abs(b) ** 2)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / d)) if d else float(np.max(np.abs(a - b), initial=0.0))


def cuts_of(n):
    k = int(rng.integers(0, 6))
    return sorted({0, n, *(int(v) for v in rng.integers(0, n + 1, size=k))})


worst = {"up": 0.0, "down": 0.0, "fm": 0.0, "stft": 0.0, "chan": 0.0}
for case in range(cases):
    n = int(rng.integers(500, 20000))
    x = o.synth_iq(200 + case, 0, n)
    flt = np.float32 if rng.random() < 0.7 else np.float64
    cdt = np.complex64 if flt == np.float32 else np.complex128
    xs = x.astype(cdt)
    # Upsampler: bit-exact
    fi = float(rng.choice([8000.0, 44100.0, 48000.0, 1e6, 3.0]))
    fo = fi * float(rng.choice([1, 2, 3, 8])) if rng.random() < 0.6 else fi * float(rng.uniform(1.0, 6.0))
    bw = fi * float(rng.uniform(0.2, 0.9))
    q = float(rng.choice([1.0, 2.0, 3.0]))
    g, r = rr.Upsampler.with_quality(64, fo, bw, q, dtype=flt), o.Upsampler(64, fo, bw, q, flt=flt)
    c = cuts_of(n)
    for a, b in zip(c[:-1], c[1:]):
        y, yr = g.process_raw(fi, xs[a:b]), r.process(fi, xs[a:b])
        assert len(y) == len(yr) and np.array_equal(y.view(flt), yr.view(flt)), ("up", case, fi, fo, bw, q, a, b)
    # Downsampler
    fo2 = fi / float(rng.choice([1, 2, 4, 5])) if rng.random() < 0.6 else fi / float(rng.uniform(1.0, 7.0))
    bw2 = fo2 * float(rng.uniform(0.3, 0.9))
    g, r = rr.Downsampler.with_quality(64, fo2, bw2, q, dtype=flt), o.Downsampler(64, fo2, bw2, q, flt=np.float64)
    ys, rs = [], []
    for a, b in zip(c[:-1], c[1:]):
        y, yr = g.process_raw(fi, xs[a:b]), r.process(fi, xs[a:b].astype(np.complex128))
        assert len(y) == len(yr), ("down", case)
        ys.append(y); rs.append(yr)
    e = rms(np.concatenate(ys), np.concatenate(rs)) if sum(len(v) for v in rs) else 0.0
    worst["down"] = max(worst["down"], e)
    assert e <= (1e-5 if flt == np.float32 else 1e-12), ("down", case, e)
    # FmDemod
    dev = float(rng.uniform(0.01, 0.4)) * fi
    g, r = rr.FmDemod(dev, dtype=flt), o.FmDemod(dev, flt=flt)
    atol = 4 * np.finfo(flt).eps * np.pi * fi / dev / (2 * np.pi)
    for i, (a, b) in enumerate(zip(c[:-1], c[1:])):
        if i == 2:
            g.process(rr.EventSignal(rr.SamplesLost())); r.interrupt()
        y, yr = g.process_raw(fi, xs[a:b]), r.process(fi, xs[a:b])
        d = float(np.max(np.abs(y - yr), initial=0.0))
        worst["fm"] = max(worst["fm"], d / atol)
        assert d <= atol, ("fm", case, d, atol)
    # Stft / Channelizer (f32 paths incl. the wave channelizer)
    M = int(rng.choice([64, 256, 256, 1024])); P = int(rng.choice([1, 2, 4, 8]))
    if M * P <= 8192:
        nch = max(P + 2, min(n // M, 40))
        xc = o.synth_iq(300 + case, 0, M * nch)
        chunks = [xc[i * M:(i + 1) * M] for i in range(nch)]
        fou = o.Fourier(o.Kaiser.with_null_at_bin(float(P)), flt=np.float64)
        ref = [fou.process(np.concatenate(chunks[i:i + P])) for i in range(nch - P + 1)]
        gs = rr.Stft(M, P, rr.Kaiser.with_null_at_bin(float(P)))
        gc = rr.Channelizer(M, P)
        outs, outc = [], []
        cc = sorted({0, nch, *(int(v) for v in rng.integers(0, nch + 1, size=3))})
        for a, b in zip(cc[:-1], cc[1:]):
            outs += gs.process(rr.Samples(1e6, xc[a * M:b * M])); outc += gc.process(rr.Samples(1e6, xc[a * M:b * M]))
        assert len(outs) == len(outc) == len(ref)
        for s, cch, rf in zip(outs, outc, ref):
            e1, e2 = rms(s.chunk, rf), rms(cch.chunk, rf[::P])
            worst["stft"] = max(worst["stft"], e1); worst["chan"] = max(worst["chan"], e2)
            assert e1 <= 1e-5 and e2 <= 1e-5, ("stft/chan", case, M, P, e1, e2)
print(f"{cases} cases ok; worst: {worst}")
