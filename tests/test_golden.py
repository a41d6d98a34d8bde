"""Committed golden fixtures (tests/golden/*.npz, made by make_golden.py from the
f64 oracle): the CPU oracle must reproduce them, and the GPU path must match
them within the north-star tolerance (1e-5 RMS; tighter where f32 allows)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def lowpass(cut):
    return lambda _b, f: 1.0 if abs(f) <= cut else 0.0


def rms_rel(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(b) ** 2)))


# ---------------------------------------------------------------- CPU: oracle
def test_oracle_reproduces_designs(oracle):
    d = load("designs.npz")
    n, fs, cut = d["filter_params_cfg2"]
    f = oracle.Filter(lowpass(cut), flt=np.float64)
    f.process(fs, np.zeros(int(n), dtype=np.complex128))
    assert np.array_equal(2 * int(n) * f.response(), d["filter_taps_cfg2"])
    fin, fout, bw, q = d["downsampler_params_cfg2"]
    ds = oracle.Downsampler(16, fout, bw, q, flt=np.float64)
    ds.process(fin, np.zeros(1, dtype=np.complex128))
    assert np.array_equal(ds.ir(), d["downsampler_ir_cfg2"])
    assert len(d["downsampler_ir_cfg2"]) == 120
    assert len(d["downsampler_ir_rx1"]) == 34 and len(d["downsampler_ir_rx2"]) == 288 and len(d["downsampler_ir_bwmeter"]) == 145
    assert len(d["nco_table_cfg1_f32"]) == 480 and len(d["nco_table_cfg2_f32"]) == 8


def test_oracle_reproduces_chain(oracle):
    c = load("chain_cfg2.npz")
    seed, t0, n = (int(v) for v in c["seed_t0_n"])
    x = oracle.synth_iq(seed, t0, n)
    mixed, filtered, decim, spectra = oracle.run_chain(
        x, 200e6, shift=25e6, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6, fft_len=4096,
        fft_window=oracle.Kaiser.with_null_at_bin(2.0), flt=np.float64)
    assert np.array_equal(mixed[:512], c["mixed_head"])
    assert np.array_equal(filtered[:512], c["filtered_head"])
    assert np.array_equal(decim[:512], c["decimated_head"])
    assert np.array_equal(spectra[0][::8], c["spectrum0_every8"])
    assert np.array_equal(spectra[3][::8], c["spectrum3_every8"])
    # f32 instantiation stays within f32 rounding of the f64 fixture
    _, _, d32, s32 = oracle.run_chain(
        x, 200e6, shift=25e6, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6, fft_len=4096,
        fft_window=oracle.Kaiser.with_null_at_bin(2.0), flt=np.float32)
    assert rms_rel(d32[:512], c["decimated_head"]) < 2e-6
    assert rms_rel(s32[0][::8], c["spectrum0_every8"]) < 2e-6


# ---------------------------------------------------------------- CPU: product host math
def test_product_designs_match_fixtures():
    import ctypes as C

    import radiorust_amd as rr

    L = rr._lib.lib()
    d = load("designs.npz")
    for name in ("cfg2", "cfg5", "bwmeter"):
        n, fs, cut = d[f"filter_params_{name}"]
        n = int(n)
        resp = rr.sample_freq_resp(lowpass(cut), n, fs)
        win = rr.Kaiser.with_null_at_bin(2.0).sample(n)
        taps = np.empty(n, dtype=np.complex128)
        assert L.rr_filter_design_taps(n, resp.ctypes.data, win.ctypes.data, taps.ctypes.data) == 0
        want = d[f"filter_taps_{name}"]
        assert np.max(np.abs(taps - want)) <= 1e-13 * np.max(np.abs(want))
    for name in ("cfg2", "rx1", "rx2", "bwmeter"):
        fin, fout, bw, q = d[f"downsampler_params_{name}"]
        cnt = C.c_size_t()
        ir = np.empty(len(d[f"downsampler_ir_{name}"]), dtype=np.float64)
        assert L.rr_downsampler_design(fin, fout, bw, q, C.byref(cnt), ir.ctypes.data, ir.size) == 0
        assert np.array_equal(ir, d[f"downsampler_ir_{name}"])
    for n in (4, 4096):
        rel = rr.Kaiser.with_null_at_bin(2.0).sample(n)
        vals = np.empty(n)
        assert L.rr_fourier_design_window(n, rel.ctypes.data, vals.ctypes.data) == 0
        assert np.array_equal(vals, d[f"fourier_window_kaiser2_{n}"])
    for name, numer, denom in (("cfg1", 7, 480), ("cfg2", 1, 8)):
        tab = np.empty(denom, dtype=np.complex64)
        assert L.rr_freqshifter_table(0, numer, denom, 0.0, tab.ctypes.data) == 0
        assert np.array_equal(tab, d[f"nco_table_{name}_f32"])


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("allow_fused", [False, True])
def test_gpu_chain_matches_fixture(oracle, allow_fused):
    import radiorust_amd as rr

    c = load("chain_cfg2.npz")
    seed, t0, n = (int(v) for v in c["seed_t0_n"])
    x = oracle.synth_iq(seed, t0, n)
    ch = rr.Chain(shift=25e6, filter_len=64, freq_resp=lowpass(20e6), output_rate=50e6, bandwidth=40e6, fft_len=4096,
                  fft_window=rr.Kaiser.with_null_at_bin(2.0), allow_fused=allow_fused)
    # feed in two ragged pieces to exercise the carries
    out = ch.process(rr.Samples(200e6, x[:30001])) + ch.process(rr.Samples(200e6, x[30001:]))
    assert len(out) == 4
    assert rms_rel(out[0].chunk[::8], c["spectrum0_every8"]) < 1e-5
    assert rms_rel(out[3].chunk[::8], c["spectrum3_every8"]) < 1e-5
    energy = np.array([np.sum(np.abs(s.chunk.astype(np.complex128)) ** 2) for s in out])
    np.testing.assert_allclose(energy, c["spectra_energy"], rtol=1e-5)


@pytest.mark.gpu
def test_gpu_blocks_match_cfg1_fixture(oracle):
    import radiorust_amd as rr

    b = load("blocks_cfg1.npz")
    seed, t0, n = (int(v) for v in b["seed_t0_n"])
    x = oracle.synth_iq(seed, t0, n)
    sh = rr.FreqShifter.with_shift(700.0)
    fl = rr.Filter.new(lowpass(16e3))
    outs = []
    for i in range(3):
        (m,) = sh.process(rr.Samples(48000.0, x[i * 4096 : (i + 1) * 4096]))
        outs += fl.process(m)
    assert len(outs) == 2  # first chunk swallowed
    y = np.concatenate([s.chunk for s in outs])
    assert rms_rel(y[:1024], b["filtered_head"]) < 1e-5
    assert rms_rel(y[-1024:], b["filtered_tail"]) < 1e-5


def _cuts(x, cdt):
    return x[:1000].astype(cdt), x[1000:].astype(cdt)


@pytest.mark.parametrize("name", ["int8", "frac"])
def test_oracle_reproduces_upsampler_fixture(oracle, name):
    g = load("resample_demod.npz")
    seed, t0, n = (int(v) for v in g["seed_t0_n"])
    x = oracle.synth_iq(seed, t0, n)
    fi, fo, bw, q = g[f"upsampler_params_{name}"]
    for flt, tag, cdt in ((np.float32, "f32", np.complex64), (np.float64, "f64", np.complex128)):
        u = oracle.Upsampler(1024, fo, bw, q, flt=flt)
        y = np.concatenate([u.process(fi, c) for c in _cuts(x, cdt)])
        assert len(y) == int(g[f"upsampler_{name}_{tag}_count"][0])
        assert np.array_equal(y[:1024], g[f"upsampler_{name}_{tag}_head"]) and np.array_equal(y[-1024:], g[f"upsampler_{name}_{tag}_tail"])
    u.process(fi, np.zeros(1, dtype=np.complex128))
    assert np.array_equal(u.ir(), g[f"upsampler_ir_{name}"])


def test_oracle_reproduces_fmdemod_fixture(oracle):
    g = load("resample_demod.npz")
    fs, dev = g["fmdemod_params"]
    d = oracle.FmDemod(dev, flt=np.float64)
    y = d.process(fs, g["fmdemod_input_f32"].astype(np.complex128))
    assert np.array_equal(y.real, g["fmdemod_output_f64"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["int8", "frac"])
def test_gpu_upsampler_matches_fixture_bit_for_bit(oracle, name):
    import radiorust_amd as rr

    g = load("resample_demod.npz")
    seed, t0, n = (int(v) for v in g["seed_t0_n"])
    x = oracle.synth_iq(seed, t0, n)
    fi, fo, bw, q = g[f"upsampler_params_{name}"]
    for flt, tag, cdt in ((np.float32, "f32", np.complex64), (np.float64, "f64", np.complex128)):
        u = rr.Upsampler.with_quality(1024, fo, bw, q, dtype=flt)
        y = np.concatenate([u.process_raw(fi, c) for c in _cuts(x, cdt)])
        assert len(y) == int(g[f"upsampler_{name}_{tag}_count"][0])
        assert np.array_equal(y[:1024], g[f"upsampler_{name}_{tag}_head"]) and np.array_equal(y[-1024:], g[f"upsampler_{name}_{tag}_tail"])


@pytest.mark.gpu
def test_gpu_fmdemod_matches_fixture(oracle):
    import radiorust_amd as rr

    g = load("resample_demod.npz")
    fs, dev = g["fmdemod_params"]
    d = rr.FmDemod(dev)
    y = d.process_raw(fs, g["fmdemod_input_f32"])
    atol = 4 * np.finfo(np.float32).eps * np.pi * fs / dev / (2 * np.pi)  # device atan2f vs the f64 result
    assert np.max(np.abs(y.real - g["fmdemod_output_f64"])) <= atol and not np.any(y.imag)
