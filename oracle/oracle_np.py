"""Independent numpy/scipy f64 formulation of the same four blocks.

TEST INFRASTRUCTURE ONLY.  It exists to catch transcription mistakes in
oracle/rr_oracle.c: every function here reaches the result by a *different
route* (closed forms, np.fft, np.convolve, scipy.special) than the line-by-line
restatement, so agreement between the two is evidence that the restatement
computes what the reference source says.  It is not run against the reference
itself (Rust, unbuildable here) — see the pinning statement in rr_oracle.h.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np
from scipy import special


def bessel_I0(x):
    return special.i0(x)


def sinc(x):
    return np.sinc(x)  # numpy's sinc is the normalised sin(pi x)/(pi x)


def kaiser(beta, x):
    return special.i0(beta * np.sqrt(1.0 - np.asarray(x, dtype=np.float64) ** 2))


def window_positions(n):
    """Sampling positions used by Filter (filters.rs:209-212) and Fourier
    (analysis.rs:93-94): x_i = 2 (i + 0.5) / n - 1."""
    return 2.0 * (np.arange(n, dtype=np.float64) + 0.5) / n - 1.0


# ---- FreqShifter ---------------------------------------------------------
def freq_ratio(sample_rate, precision, shift):
    """transform.rs:298-302 with exact rational arithmetic for the reduction.
    Python's round() is half-to-even, Rust's f64::round is half-away-from-
    zero, so round explicitly."""

    def rnd(v):
        return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)

    denom = rnd(sample_rate / precision)
    numer = rnd(denom * shift / sample_rate)
    fr = Fraction(numer, denom)
    return fr.numerator, fr.denominator


def freqshift(x, numer, denom, idx0=0, start_phase=0.0):
    """y[t] = x[t] * e^{j(start + 2 pi numer (idx0+t) / denom)} in f64."""
    t = (np.arange(len(x), dtype=np.int64) + idx0) % denom
    k = (t * numer) % denom
    return np.asarray(x, dtype=np.complex128) * np.exp(1j * (start_phase + 2.0 * np.pi * k / denom))


# ---- Filter --------------------------------------------------------------
def filter_taps(n, sample_rate, freq_resp, window_values):
    """Causal FIR taps g[k] equivalent to the reference's overlap-save filter:
    out[t] = sum_k g[k] x[t-k]  (filters.rs:184-259; g = 2n * h).
    `window_values` = window sampled at window_positions(n)."""
    R = np.zeros(n, dtype=np.complex128)
    for i in range((n - 1) // 2 + 1):
        R[i] = freq_resp(i, i * sample_rate / n)
        if i > 0:
            R[n - i] = freq_resp(-i, -i * sample_rate / n)
    h = np.fft.ifft(R)  # = unnormalised inverse / n
    # swap(i, i + n//2) for i < n//2  (odd n leaves the last element alone)
    half = n // 2
    hs = h.copy()
    hs[:half], hs[half : 2 * half] = h[half : 2 * half].copy(), h[:half].copy()
    e_pre = np.sum(np.abs(hs) ** 2)
    hw = hs * window_values
    e_post = np.sum(np.abs(hw) ** 2)
    return hw * math.sqrt(e_pre / e_post)


def fir_causal(x, g, t0):
    """out[t] for t in [t0, len(x)): sum_k g[k] x[t-k], x[<0] = 0."""
    y = np.convolve(np.asarray(x, dtype=np.complex128), g)[: len(x)]
    return y[t0:]


# ---- Downsampler ---------------------------------------------------------
def downsampler_ir(input_rate, output_rate, bandwidth, quality=3.0):
    margin = (output_rate - bandwidth) / 2.0
    L = int(math.ceil(input_rate / margin * quality))
    beta = math.sqrt((L * margin / input_rate) ** 2 - 1.0)
    x = np.arange(L, dtype=np.float64) + 0.5 - L / 2.0
    y = np.sinc(x * output_rate / input_rate) * kaiser(beta, 2.0 * x / L)
    return y / math.sqrt(np.sum(y * y))


def emit_indices(n_in, input_rate, output_rate):
    """Indices t (0-based input sample) after which an output is emitted, by
    the closed form floor((t+1) r) > floor(t r) with exact rationals."""
    r = Fraction(output_rate) / Fraction(input_rate)
    out = []
    prev = 0
    for t in range(n_in):
        cur = math.floor((t + 1) * r)
        if cur > prev:
            out.append(t)
        prev = cur
    return np.asarray(out, dtype=np.int64)


def downsample(z, ir, emit):
    """v[m] = sum_j ir[j] z[emit[m] - L + 1 + j], z[<0] = 0."""
    L = len(ir)
    zp = np.concatenate([np.zeros(L - 1, dtype=np.complex128), np.asarray(z, dtype=np.complex128)])
    return np.array([np.dot(ir, zp[t : t + L]) for t in emit], dtype=np.complex128)


# ---- Fourier -------------------------------------------------------------
def fourier_window(n, rel_values):
    rel_values = np.asarray(rel_values, dtype=np.float64)
    return rel_values * math.sqrt(n / np.sum(rel_values**2))


def fourier(x, w, center_dc=False):
    X = np.fft.fft(np.asarray(x, dtype=np.complex128) * w)
    return np.roll(X, len(X) // 2) if center_dc else X
