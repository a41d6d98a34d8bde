"""Generates tests/golden/*.npz from the f64 CPU oracle (oracle/rr_oracle.c).

Run from the repo root:  python tests/golden/make_golden.py

What the fixtures are: designed coefficients and short input/output excerpts of
the BASELINE configs, computed by the f64 instantiation of the oracle and
cross-checked here against the independent numpy formulation
(oracle/oracle_np.py).  They are NOT outputs of the reference itself (Rust,
unbuildable in this image; its only hot-path vectors are the known-answer tests
ported in tests/test_oracle_kat.py) — they anchor the oracle and the GPU path to
each other and guard against regressions.  Inputs are the counter-based
synthetic IQ (seed, t0, n), so only outputs are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_np as onp  # noqa: E402
from oracle import rr_oracle as o  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def lowpass(cut):
    return lambda _b, f: 1.0 if abs(f) <= cut else 0.0


def rms_rel(a, b):
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(b) ** 2)))


def design_fixture():
    d = {}
    for name, n, fs, cut in (("cfg2", 64, 200e6, 20e6), ("cfg5", 1024, 2e9, 200e6), ("bwmeter", 1024, 102400.0, 30e3)):
        f = o.Filter(lowpass(cut), flt=np.float64)
        f.process(fs, np.zeros(n, dtype=np.complex128))
        g = 2 * n * f.response()
        wv = np.array([o.Kaiser.with_null_at_bin(2.0).relative_value_at(p) for p in onp.window_positions(n)])
        assert rms_rel(g, onp.filter_taps(n, fs, lowpass(cut), wv)) < 1e-12
        d[f"filter_taps_{name}"] = g
        d[f"filter_params_{name}"] = np.array([n, fs, cut])
    for name, fin, fout, bw in (("cfg2", 200e6, 50e6, 40e6), ("rx1", 1024000.0, 384000.0, 200000.0),
                                ("rx2", 384000.0, 48000.0, 40000.0), ("bwmeter", 1024000.0, 102400.0, 60000.0)):
        ds = o.Downsampler(16, fout, bw, 3.0, flt=np.float64)
        ds.process(fin, np.zeros(1, dtype=np.complex128))
        ir = ds.ir()
        assert np.max(np.abs(ir - onp.downsampler_ir(fin, fout, bw, 3.0))) < 1e-14
        d[f"downsampler_ir_{name}"] = ir
        d[f"downsampler_params_{name}"] = np.array([fin, fout, bw, 3.0])
    for n in (4, 4096):
        fo = o.Fourier(o.Kaiser.with_null_at_bin(2.0), flt=np.float64)
        fo.process(np.zeros(n, dtype=np.complex128))
        d[f"fourier_window_kaiser2_{n}"] = fo.window_values()
    for name, fs, prec, shift in (("cfg1", 48000.0, 1.0, 700.0), ("cfg2", 200e6, 1.0, 25e6)):
        for flt, tag in ((np.float32, "f32"), (np.float64, "f64")):
            sh = o.FreqShifter(prec, shift, flt=flt)
            sh.process(fs, np.zeros(1, dtype=np.complex64))
            d[f"nco_table_{name}_{tag}"] = sh.table()
    np.savez_compressed(os.path.join(OUT, "designs.npz"), **d)


def chain_fixture():
    """BASELINE configs[1]: 200 MS/s, shift 25 MHz, 64-tap low-pass 20 MHz,
    Downsampler(4096, 50 MS/s, 40 MHz), Fourier Kaiser(null at bin 2)."""
    fs, n_in = 200e6, 64 * 1200  # 76 800 samples -> 4 spectra
    x = o.synth_iq(1, 0, n_in)
    kw = dict(shift=25e6, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6,
              fft_len=4096, fft_window=o.Kaiser.with_null_at_bin(2.0))
    mixed, filtered, decim, spectra = o.run_chain(x, fs, flt=np.float64, **kw)
    assert len(spectra) == 4
    # independent formulation of the same chain
    numer, denom = onp.freq_ratio(fs, 1.0, 25e6)
    m2 = onp.freqshift(x, numer, denom)
    wv = np.array([o.Kaiser.with_null_at_bin(2.0).relative_value_at(p) for p in onp.window_positions(64)])
    g = onp.filter_taps(64, fs, lowpass(20e6), wv)
    z2 = onp.fir_causal(m2, g, 64)
    ir = onp.downsampler_ir(fs, 50e6, 40e6)
    v2 = onp.downsample(z2, ir, onp.emit_indices(len(z2), fs, 50e6))
    w = onp.fourier_window(4096, np.array([o.Kaiser.with_null_at_bin(2.0).relative_value_at(p) for p in onp.window_positions(4096)]))
    assert rms_rel(mixed, m2) < 1e-13 and rms_rel(filtered, z2) < 1e-12
    assert rms_rel(decim, v2[: len(decim)]) < 1e-12
    for k, s in enumerate(spectra):
        assert rms_rel(s, onp.fourier(v2[k * 4096 : (k + 1) * 4096], w)) < 1e-12
    np.savez_compressed(
        os.path.join(OUT, "chain_cfg2.npz"),
        params=np.array([fs, 25e6, 64, 20e6, 50e6, 40e6, 3.0, 4096, 2.0]),
        seed_t0_n=np.array([1, 0, n_in], dtype=np.int64),
        mixed_head=mixed[:512],
        filtered_head=filtered[:512],
        decimated_head=decim[:512],
        decimated_tail=decim[-512:],
        spectrum0_every8=spectra[0][::8],
        spectrum3_every8=spectra[3][::8],
        spectra_energy=np.array([np.sum(np.abs(s) ** 2) for s in spectra]),
    )


def blocks_fixture():
    """Per-block excerpts for cfg1 (48 kS/s FreqShifter + 4096-tap low-pass)."""
    fs = 48000.0
    x = o.synth_iq(2, 0, 4096 * 3)
    sh = o.FreqShifter(1.0, 700.0, flt=np.float64)
    fl = o.Filter(lowpass(16e3), flt=np.float64)
    outs = []
    for i in range(3):
        m = sh.process(fs, x[i * 4096 : (i + 1) * 4096])
        y = fl.process(fs, m)
        if y is not None:
            outs.append(y)
    y = np.concatenate(outs)
    np.savez_compressed(os.path.join(OUT, "blocks_cfg1.npz"), seed_t0_n=np.array([2, 0, 4096 * 3], dtype=np.int64),
                        filtered_head=y[:1024], filtered_tail=y[-1024:])


def resample_demod_fixture():
    """Upsampler (resampling.rs:147-280) 48 kS/s -> 384 kS/s (L = 288) and 44.1 -> 48 kS/s (L = 21), and
    FmDemod (modulation.rs:83-158), in f32 (the Upsampler's f32 arithmetic order is part of the contract:
    the GPU kernel is compared bit for bit) and f64.  Cross-check: the numpy scatter-add formulation."""
    x = o.synth_iq(4, 0, 2048)
    d = {"seed_t0_n": np.array([4, 0, 2048], dtype=np.int64)}
    for name, fi, fo, bw in (("int8", 48000.0, 384000.0, 40000.0), ("frac", 44100.0, 48000.0, 30000.0)):
        for flt, tag in ((np.float32, "f32"), (np.float64, "f64")):
            u = o.Upsampler(1024, fo, bw, 3.0, flt=flt)
            y = np.concatenate([u.process(fi, x[:1000].astype(np.complex64 if flt == np.float32 else np.complex128)),
                                u.process(fi, x[1000:].astype(np.complex64 if flt == np.float32 else np.complex128))])
            if flt == np.float64:
                ir, L = u.ir(), len(u.ir())
                pos, cnt, ref = 0.0, 0, np.zeros(len(y) + L, dtype=np.complex128)
                for xv in x.astype(np.complex128):
                    ref[cnt:cnt + L] += xv * ir
                    while pos < fo:
                        cnt += 1
                        pos += fi
                    pos -= fo
                assert cnt == len(y) and np.array_equal(ref[:cnt], y)
                d[f"upsampler_ir_{name}"] = ir
            d[f"upsampler_{name}_{tag}_head"] = y[:1024]
            d[f"upsampler_{name}_{tag}_tail"] = y[-1024:]
            d[f"upsampler_{name}_{tag}_count"] = np.array([len(y)], dtype=np.int64)
        d[f"upsampler_params_{name}"] = np.array([fi, fo, bw, 3.0])
    fs, dev = 384000.0, 75000.0
    t = np.arange(2048)
    msg = 0.7 * np.sin(2 * np.pi * 1000 * t / fs)
    xm = (np.exp(1j * np.cumsum(msg * dev / fs * 2 * np.pi)) * (1 + 0.01 * x)).astype(np.complex64)
    dm = o.FmDemod(dev, flt=np.float64)
    y = dm.process(fs, xm.astype(np.complex128))
    ref = np.concatenate([[0.0], np.angle(xm[1:].astype(np.complex128) * np.conj(xm[:-1].astype(np.complex128))) * (fs / dev / (2 * np.pi))])
    assert np.max(np.abs(y.real - ref)) < 1e-12 and not np.any(y.imag)
    d["fmdemod_params"] = np.array([fs, dev])
    d["fmdemod_input_f32"] = xm
    d["fmdemod_output_f64"] = y.real
    np.savez_compressed(os.path.join(OUT, "resample_demod.npz"), **d)


if __name__ == "__main__":
    o.build()
    design_fixture()
    chain_fixture()
    blocks_fixture()
    resample_demod_fixture()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
