"""ctypes front-end of the CPU oracle (oracle/rr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (radiorust_amd/) never
imports this module.

Every class mirrors one reference block and keeps that block's per-task state
(see rr_oracle.h for the file:line map).  Parity pinning: Fourier, bessel_I0
and sinc are pinned by the reference's own known-answer tests; Filter,
FreqShifter and Downsampler are PARITY UNPINNED (the reference has no tests
for them and cannot be built here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librr_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  No-op when the .so is
    newer than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("rr_oracle.c", "rr_oracle_impl.inc", "rr_oracle.h")]
    if (
        not force
        and os.path.exists(_LIB_PATH)
        and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)
    ):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None

WIN_RECT, WIN_KAISER, WIN_CUSTOM = 0, 1, 2
_WINFN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)
_RESPFN = C.CFUNCTYPE(None, C.c_int64, C.c_double, C.POINTER(C.c_double), C.c_void_p)


class _CWindow(C.Structure):
    _fields_ = [("kind", C.c_int), ("beta", C.c_double), ("fn", _WINFN), ("ud", C.c_void_p)]


NATIVE_FLAG_SETS = ("-O3", "-O3 -march=native", "-O3 -march=native -mprefer-vector-width=256")


def _host_tag() -> str:
    import hashlib
    import platform

    return hashlib.sha1((cpu_model() + platform.machine()).encode()).hexdigest()[:10]


def build_native(flags: str | None = None) -> str:
    """The same sources compiled for THIS host's cores (`-march=native`, a*b+c contraction allowed): the timing
    build of bench.py's cpu_baseline.  It is built where it runs (the GPU box's host CPU is not this
    container's), into a file named after the host's CPU model so that a copy made elsewhere is never loaded.
    Never used for parity: results may differ from the parity build in the last bits.
    flags = None: the fastest of NATIVE_FLAG_SETS on a short probe of the cfg2 chain (the auto-vectoriser's
    choices differ a lot between hosts: 15 / 21 / 28 MSamples/s for the three sets on one Xeon)."""
    if flags is None:
        return pick_native()[0]
    import hashlib

    out = os.path.join(_HERE, "_build", f"librr_oracle_native_{_host_tag()}_{hashlib.sha1(flags.encode()).hexdigest()[:6]}.so")
    srcs = [os.path.join(_HERE, f) for f in ("rr_oracle.c", "rr_oracle_impl.inc", "rr_oracle.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run([os.environ.get("CC", "gcc"), *flags.split(), "-fPIC", "-std=gnu11", "-shared", "-o", out, srcs[0], "-lm",
                    "-lpthread"], check=True)
    return out


_native_choice = None


def pick_native():
    """(path, flags, {flags: MSamples/s}) of the fastest timing build on this host (single-threaded cfg2 probe)."""
    global _native_choice, _lib
    if _native_choice is not None:
        return _native_choice
    import time

    x = synth_iq(1, 0, 1 << 19)
    rates = {}
    saved = _lib
    try:
        for fl in NATIVE_FLAG_SETS:
            _lib = _bind(C.CDLL(build_native(fl)))
            best = 0.0
            for _ in range(2):
                t = time.perf_counter()
                run_chain_c(x, 200e6, shift=25e6, filter_len=64, freq_resp=lambda _b, f: 1.0 if abs(f) <= 20e6 else 0.0,
                            output_rate=50e6, bandwidth=40e6, fft_len=4096, fft_window=Kaiser.with_null_at_bin(2.0),
                            flt=np.float32, max_frames=1)
                best = max(best, x.size / (time.perf_counter() - t) / 1e6)
            rates[fl] = round(best, 2)
    finally:
        _lib = saved
    fl = max(rates, key=rates.get)
    _native_choice = (build_native(fl), fl, rates)
    return _native_choice


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class native:
    """Context manager: inside it every oracle call goes to the `-march=native` timing build."""

    def __enter__(self):
        global _lib
        self._saved = _lib
        _lib = _bind(C.CDLL(build_native()))
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    _lib = _bind(C.CDLL(build()))
    return _lib


def _bind(L):
    d = C.c_double
    for name in ("rro_bessel_i0", "rro_kaiser_alpha_to_beta", "rro_kaiser_null_at_bin_to_beta", "rro_sinc"):
        getattr(L, name).restype = d
        getattr(L, name).argtypes = [d]
    L.rro_deemphasis_factor.restype = None
    L.rro_deemphasis_factor.argtypes = [d, d, C.POINTER(d)]
    L.rro_kaiser_rel_with_beta.restype = d
    L.rro_kaiser_rel_with_beta.argtypes = [d, d]
    L.rro_window_value.restype = d
    L.rro_window_value.argtypes = [C.POINTER(_CWindow), d]
    L.rro_freq_to_ratio.restype = None
    L.rro_freq_to_ratio.argtypes = [d, d, d, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.rro_synth_iq_f32.restype = None
    L.rro_synth_iq_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
    vp, sz = C.c_void_p, C.c_size_t
    for suf in ("f32", "f64"):
        g = lambda n: getattr(L, f"{n}_{suf}")  # noqa: E731
        g("rro_fft").restype = None
        g("rro_fft").argtypes = [vp, sz, C.c_int]
        g("rro_freqshifter_new").restype = vp
        g("rro_freqshifter_new").argtypes = [d, d]
        g("rro_freqshifter_set_shift").restype = None
        g("rro_freqshifter_set_shift").argtypes = [vp, d]
        g("rro_freqshifter_process").restype = None
        g("rro_freqshifter_process").argtypes = [vp, d, vp, sz, vp]
        g("rro_freqshifter_table").restype = sz
        g("rro_freqshifter_table").argtypes = [vp, vp, sz]
        g("rro_freqshifter_free").restype = None
        g("rro_freqshifter_free").argtypes = [vp]
        g("rro_filter_new").restype = vp
        g("rro_filter_new").argtypes = [_RESPFN, vp, C.POINTER(_CWindow)]
        g("rro_filter_update").restype = None
        g("rro_filter_update").argtypes = [vp, _RESPFN, vp, C.POINTER(_CWindow)]
        g("rro_filter_process").restype = sz
        g("rro_filter_process").argtypes = [vp, d, vp, sz, vp]
        g("rro_filter_interrupt").restype = None
        g("rro_filter_interrupt").argtypes = [vp]
        g("rro_filter_response").restype = sz
        g("rro_filter_response").argtypes = [vp, vp, sz]
        g("rro_filter_free").restype = None
        g("rro_filter_free").argtypes = [vp]
        g("rro_downsampler_new").restype = vp
        g("rro_downsampler_new").argtypes = [d, d, d]
        g("rro_downsampler_process").restype = sz
        g("rro_downsampler_process").argtypes = [vp, d, vp, sz, vp, sz]
        g("rro_downsampler_ir").restype = sz
        g("rro_downsampler_ir").argtypes = [vp, vp, sz]
        g("rro_downsampler_free").restype = None
        g("rro_downsampler_free").argtypes = [vp]
        g("rro_fourier_new").restype = vp
        g("rro_fourier_new").argtypes = [C.POINTER(_CWindow), C.c_int]
        g("rro_fourier_process").restype = None
        g("rro_fourier_process").argtypes = [vp, vp, sz, vp]
        g("rro_fourier_window").restype = sz
        g("rro_fourier_window").argtypes = [vp, vp, sz]
        g("rro_fourier_free").restype = None
        g("rro_fourier_free").argtypes = [vp]
        g("rro_level").restype = d
        g("rro_level").argtypes = [vp, sz]
        g("rro_bandwidth").restype = d
        g("rro_bandwidth").argtypes = [d, d, vp, sz]
        g("rro_rescale_energy").restype = C.c_int
        g("rro_rescale_energy").argtypes = [vp, sz, vp, sz]
        g("rro_gain").restype = None
        g("rro_gain").argtypes = [d, vp, sz, vp]
        g("rro_upsampler_new").restype = vp
        g("rro_upsampler_new").argtypes = [d, d, d]
        g("rro_upsampler_process").restype = sz
        g("rro_upsampler_process").argtypes = [vp, d, vp, sz, vp, sz]
        g("rro_upsampler_ir").restype = sz
        g("rro_upsampler_ir").argtypes = [vp, vp, sz]
        g("rro_upsampler_free").restype = None
        g("rro_upsampler_free").argtypes = [vp]
        g("rro_fmdemod_new").restype = vp
        g("rro_fmdemod_new").argtypes = [d]
        g("rro_fmdemod_set_deviation").restype = None
        g("rro_fmdemod_set_deviation").argtypes = [vp, d]
        g("rro_fmdemod_interrupt").restype = None
        g("rro_fmdemod_interrupt").argtypes = [vp]
        g("rro_fmdemod_process").restype = None
        g("rro_fmdemod_process").argtypes = [vp, d, vp, sz, vp]
        g("rro_fmdemod_free").restype = None
        g("rro_fmdemod_free").argtypes = [vp]
        g("rro_chain_run").restype = sz
        g("rro_chain_run").argtypes = [vp, sz, d, d, d, sz, _RESPFN, vp, C.POINTER(_CWindow), d, d, d, sz,
                                       C.POINTER(_CWindow), C.c_int, vp, sz]
        g("rro_chain_run_mt").restype = sz
        g("rro_chain_run_mt").argtypes = [vp, sz, d, d, d, sz, _RESPFN, vp, C.POINTER(_CWindow), d, d, d, sz,
                                          C.POINTER(_CWindow), C.c_int, vp, sz, sz]
    return L


# --------------------------------------------------------------------------
# design math
# --------------------------------------------------------------------------
def bessel_I0(x: float) -> float:
    return lib().rro_bessel_i0(float(x))


def sinc(x: float) -> float:
    return lib().rro_sinc(float(x))


def deemphasis_factor(tau: float, frequency: float) -> complex:
    out = (C.c_double * 2)()
    lib().rro_deemphasis_factor(float(tau), float(frequency), out)
    return complex(out[0], out[1])


def kaiser_rel_with_beta(beta: float, x: float) -> float:
    return lib().rro_kaiser_rel_with_beta(float(beta), float(x))


def kaiser_alpha_to_beta(alpha: float) -> float:
    return lib().rro_kaiser_alpha_to_beta(float(alpha))


def kaiser_null_at_bin_to_beta(n: float) -> float:
    return lib().rro_kaiser_null_at_bin_to_beta(float(n))


def freq_to_ratio(sample_rate: float, precision: float, frequency: float):
    n, d = C.c_int64(), C.c_int64()
    lib().rro_freq_to_ratio(float(sample_rate), float(precision), float(frequency), C.byref(n), C.byref(d))
    return n.value, d.value


# --------------------------------------------------------------------------
# windows (windowing.rs)
# --------------------------------------------------------------------------
class Window:
    """Base: `relative_value_at(x)`, x in [-1, 1] (windowing.rs:6-10)."""

    def _c(self) -> _CWindow:
        raise NotImplementedError

    def relative_value_at(self, x: float) -> float:
        cw = self._c()
        return lib().rro_window_value(C.byref(cw), float(x))


class Rectangular(Window):
    def _c(self):
        return _CWindow(WIN_RECT, 0.0, _WINFN(0), None)


class Kaiser(Window):
    def __init__(self, beta: float):
        self.beta = float(beta)

    @classmethod
    def with_beta(cls, beta):
        return cls(beta)

    @classmethod
    def with_alpha(cls, alpha):
        return cls(kaiser_alpha_to_beta(alpha))

    @classmethod
    def with_null_at_bin(cls, n):
        return cls(kaiser_null_at_bin_to_beta(n))

    def _c(self):
        return _CWindow(WIN_KAISER, self.beta, _WINFN(0), None)


class CustomWindow(Window):
    def __init__(self, fn):
        self._fn = fn
        self._cfn = _WINFN(lambda x, _ud: float(fn(x)))

    def _c(self):
        return _CWindow(WIN_CUSTOM, 0.0, self._cfn, None)


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _dt(flt):
    flt = np.dtype(flt)
    if flt == np.float32:
        return "f32", np.complex64, np.float32
    if flt == np.float64:
        return "f64", np.complex128, np.float64
    raise TypeError(flt)


def _cin(a, cdt):
    return np.ascontiguousarray(a, dtype=cdt)


def fft(x, inverse: bool = False, flt=np.float64):
    """Unnormalised DFT with the rustfft sign convention, any length."""
    suf, cdt, _ = _dt(flt)
    a = _cin(x, cdt).copy()
    getattr(lib(), f"rro_fft_{suf}")(a.ctypes.data, a.size, int(bool(inverse)))
    return a


def synth_iq(seed: int, t0: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.complex64)
    lib().rro_synth_iq_f32(int(seed), int(t0), int(n), out.ctypes.data)
    return out


# --------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------
class FreqShifter:
    """transform.rs:266-391 (per-task state + one Samples message)."""

    def __init__(self, precision=1.0, shift=0.0, flt=np.float32):
        self._suf, self._cdt, _ = _dt(flt)
        self._h = getattr(lib(), f"rro_freqshifter_new_{self._suf}")(float(precision), float(shift))

    def set_shift(self, shift):
        getattr(lib(), f"rro_freqshifter_set_shift_{self._suf}")(self._h, float(shift))

    def process(self, sample_rate, chunk):
        x = _cin(chunk, self._cdt)
        y = np.empty_like(x)
        getattr(lib(), f"rro_freqshifter_process_{self._suf}")(
            self._h, float(sample_rate), x.ctypes.data, x.size, y.ctypes.data
        )
        return y

    def table(self):
        f = getattr(lib(), f"rro_freqshifter_table_{self._suf}")
        n = f(self._h, None, 0)
        out = np.empty(n, dtype=self._cdt)
        f(self._h, out.ctypes.data, n)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            getattr(lib(), f"rro_freqshifter_free_{self._suf}")(self._h)
            self._h = None


class Filter:
    """filters.rs:110-298.  `freq_resp(bin, freq) -> complex`."""

    def __init__(self, freq_resp, window: Window | None = None, flt=np.float32):
        self._suf, self._cdt, _ = _dt(flt)
        self._set_fn(freq_resp)
        self._window = window if window is not None else Kaiser.with_null_at_bin(2.0)
        cw = self._window._c()
        self._h = getattr(lib(), f"rro_filter_new_{self._suf}")(self._cfn, None, C.byref(cw))

    def _set_fn(self, freq_resp):
        def tramp(bin_, freq, out, _ud):
            v = complex(freq_resp(int(bin_), float(freq)))
            out[0] = v.real
            out[1] = v.imag

        self._cfn = _RESPFN(tramp)

    def update(self, freq_resp, window: Window | None = None):
        self._set_fn(freq_resp)
        if window is not None:
            self._window = window
            cw = window._c()
            getattr(lib(), f"rro_filter_update_{self._suf}")(self._h, self._cfn, None, C.byref(cw))
        else:
            getattr(lib(), f"rro_filter_update_{self._suf}")(self._h, self._cfn, None, None)

    def interrupt(self):
        getattr(lib(), f"rro_filter_interrupt_{self._suf}")(self._h)

    def process(self, sample_rate, chunk):
        """Returns the output chunk, or None (first chunk after a reset)."""
        x = _cin(chunk, self._cdt)
        y = np.empty_like(x)
        n = getattr(lib(), f"rro_filter_process_{self._suf}")(
            self._h, float(sample_rate), x.ctypes.data, x.size, y.ctypes.data
        )
        return y if n else None

    def response(self):
        """Windowed, energy-normalised impulse response h (f64, before the
        cast to Flt); the equivalent causal FIR taps are g = 2n*h."""
        f = getattr(lib(), f"rro_filter_response_{self._suf}")
        n = f(self._h, None, 0)
        out = np.empty(n, dtype=np.complex128)
        f(self._h, out.ctypes.data, n)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            getattr(lib(), f"rro_filter_free_{self._suf}")(self._h)
            self._h = None


class Downsampler:
    """resampling.rs:14-146.  `process` returns the outputs produced by this
    input chunk; regrouping into `output_chunk_len` chunks is `feed`."""

    def __init__(self, output_chunk_len, output_rate, bandwidth, quality=3.0, flt=np.float32):
        self._suf, self._cdt, self._rdt = _dt(flt)
        self.output_chunk_len = int(output_chunk_len)
        self.output_rate = float(output_rate)
        self._h = getattr(lib(), f"rro_downsampler_new_{self._suf}")(
            float(output_rate), float(bandwidth), float(quality)
        )
        if not self._h:
            raise AssertionError("Downsampler contract violated (resampling.rs:51-56)")
        self._pending = np.empty(0, dtype=self._cdt)

    def process(self, input_rate, chunk):
        x = _cin(chunk, self._cdt)
        cap = x.size + 1
        y = np.empty(cap, dtype=self._cdt)
        n = getattr(lib(), f"rro_downsampler_process_{self._suf}")(
            self._h, float(input_rate), x.ctypes.data, x.size, y.ctypes.data, cap
        )
        if n == C.c_size_t(-1).value:
            raise AssertionError("Downsampler contract violated (resampling.rs:77-81)")
        assert n <= cap
        return y[:n].copy()

    def feed(self, input_rate, chunk):
        """Like the block: list of full output chunks of output_chunk_len."""
        self._pending = np.concatenate([self._pending, self.process(input_rate, chunk)])
        out = []
        L = self.output_chunk_len
        while self._pending.size >= L:
            out.append(self._pending[:L].copy())
            self._pending = self._pending[L:]
        return out

    def ir(self):
        f = getattr(lib(), f"rro_downsampler_ir_{self._suf}")
        n = f(self._h, None, 0)
        out = np.empty(n, dtype=self._rdt)
        f(self._h, out.ctypes.data, n)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            getattr(lib(), f"rro_downsampler_free_{self._suf}")(self._h)
            self._h = None


class Upsampler:
    """resampling.rs:147-280.  `process` returns the outputs produced by this input chunk;
    regrouping into `output_chunk_len` chunks is `feed`."""

    def __init__(self, output_chunk_len, output_rate, bandwidth, quality=3.0, flt=np.float32):
        self._suf, self._cdt, self._rdt = _dt(flt)
        self.output_chunk_len = int(output_chunk_len)
        self.output_rate = float(output_rate)
        self._h = getattr(lib(), f"rro_upsampler_new_{self._suf}")(float(output_rate), float(bandwidth), float(quality))
        if not self._h:
            raise AssertionError("Upsampler contract violated (resampling.rs:185-186)")
        self._pending = np.empty(0, dtype=self._cdt)

    def process(self, input_rate, chunk):
        x = _cin(chunk, self._cdt)
        ratio = self.output_rate / float(input_rate) if input_rate > 0 else 1.0
        cap = int(x.size * ratio) + 8
        y = np.empty(cap, dtype=self._cdt)
        n = getattr(lib(), f"rro_upsampler_process_{self._suf}")(
            self._h, float(input_rate), x.ctypes.data, x.size, y.ctypes.data, cap
        )
        if n == C.c_size_t(-1).value:
            raise AssertionError("Upsampler contract violated (resampling.rs:205-214)")
        assert n <= cap
        return y[:n].copy()

    def feed(self, input_rate, chunk):
        self._pending = np.concatenate([self._pending, self.process(input_rate, chunk)])
        out = []
        L = self.output_chunk_len
        while self._pending.size >= L:
            out.append(self._pending[:L].copy())
            self._pending = self._pending[L:]
        return out

    def ir(self):
        f = getattr(lib(), f"rro_upsampler_ir_{self._suf}")
        n = f(self._h, None, 0)
        out = np.empty(n, dtype=self._rdt)
        f(self._h, out.ctypes.data, n)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            getattr(lib(), f"rro_upsampler_free_{self._suf}")(self._h)
            self._h = None


class FmDemod:
    """modulation.rs:83-158."""

    def __init__(self, deviation, flt=np.float32):
        self._suf, self._cdt, self._rdt = _dt(flt)
        self._h = getattr(lib(), f"rro_fmdemod_new_{self._suf}")(float(deviation))

    def set_deviation(self, deviation):
        getattr(lib(), f"rro_fmdemod_set_deviation_{self._suf}")(self._h, float(deviation))

    def interrupt(self):
        getattr(lib(), f"rro_fmdemod_interrupt_{self._suf}")(self._h)

    def process(self, sample_rate, chunk):
        x = _cin(chunk, self._cdt)
        y = np.empty_like(x)
        getattr(lib(), f"rro_fmdemod_process_{self._suf}")(self._h, float(sample_rate), x.ctypes.data, x.size, y.ctypes.data)
        return y

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            getattr(lib(), f"rro_fmdemod_free_{self._suf}")(self._h)
            self._h = None


class Fourier:
    """analysis.rs:26-133."""

    def __init__(self, window: Window | None = None, center_dc=False, flt=np.float32):
        self._suf, self._cdt, self._rdt = _dt(flt)
        self._window = window if window is not None else Rectangular()
        cw = self._window._c()
        self._h = getattr(lib(), f"rro_fourier_new_{self._suf}")(C.byref(cw), int(bool(center_dc)))

    def process(self, chunk):
        x = _cin(chunk, self._cdt)
        y = np.empty_like(x)
        getattr(lib(), f"rro_fourier_process_{self._suf}")(self._h, x.ctypes.data, x.size, y.ctypes.data)
        return y

    def window_values(self):
        f = getattr(lib(), f"rro_fourier_window_{self._suf}")
        n = f(self._h, None, 0)
        out = np.empty(n, dtype=self._rdt)
        f(self._h, out.ctypes.data, n)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            getattr(lib(), f"rro_fourier_free_{self._suf}")(self._h)
            self._h = None


# --------------------------------------------------------------------------
# the chain of BASELINE configs[1] as the reference would wire it:
#   source(chunks of n_filter) -> FreqShifter -> Filter -> Downsampler ->
#   Fourier   (examples/bandwidth_meter/main.rs:51-72 is the template)
# --------------------------------------------------------------------------
def run_chain(
    x,
    sample_rate,
    *,
    shift,
    precision=1.0,
    filter_len,
    freq_resp,
    filter_window=None,
    output_rate,
    bandwidth,
    quality=3.0,
    fft_len,
    fft_window=None,
    center_dc=False,
    flt=np.float32,
):
    """Feeds x in chunks of `filter_len` through the four blocks; returns
    (mixed, filtered, decimated, spectra[list]) so each stage can be compared."""
    suf, cdt, _ = _dt(flt)
    x = _cin(x, cdt)
    fs = FreqShifter(precision, shift, flt=flt)
    fl = Filter(freq_resp, filter_window, flt=flt)
    ds = Downsampler(fft_len, output_rate, bandwidth, quality, flt=flt)
    fo = Fourier(fft_window, center_dc, flt=flt)
    mixed, filtered, decimated, spectra = [], [], [], []
    n = int(filter_len)
    for off in range(0, x.size - n + 1, n):
        m = fs.process(sample_rate, x[off : off + n])
        mixed.append(m)
        f = fl.process(sample_rate, m)
        if f is None:
            continue
        filtered.append(f)
        for chunk in ds.feed(sample_rate, f):
            decimated.append(chunk)
            spectra.append(fo.process(chunk))
    cat = lambda l: np.concatenate(l) if l else np.empty(0, dtype=cdt)  # noqa: E731
    return cat(mixed), cat(filtered), cat(decimated), spectra


def run_chain_c(x, sample_rate, *, shift, precision=1.0, filter_len, freq_resp, filter_window=None, output_rate,
                bandwidth, quality=3.0, fft_len, fft_window=None, center_dc=False, flt=np.float32, max_frames=None,
                threads=1, batch=256):
    """Same wiring as run_chain, but the chunk loop runs in C (rro_chain_run):
    returns the spectra as an array [frames, fft_len].  With `max_frames` only
    the first frames are kept (all the work is still done) — used for timing.
    threads=4: one thread per block with capacity-1 hand-off of `batch` Filter
    chunks per message (rro_chain_run_mt); same spectra, bit for bit."""
    suf, cdt, _ = _dt(flt)
    x = _cin(x, cdt)
    fw = filter_window if filter_window is not None else Kaiser.with_null_at_bin(2.0)
    ffw = fft_window if fft_window is not None else Rectangular()

    def tramp(bin_, freq, out, _ud):
        v = complex(freq_resp(int(bin_), float(freq)))
        out[0] = v.real
        out[1] = v.imag

    cfn = _RESPFN(tramp)
    cap = max(1, x.size // int(filter_len) if max_frames is None else int(max_frames))
    cap = min(cap, x.size // int(fft_len) + 1)
    out = np.empty((cap, int(fft_len)), dtype=cdt)
    cfw, cffw = fw._c(), ffw._c()
    args = (x.ctypes.data, x.size, float(sample_rate), float(precision), float(shift), int(filter_len), cfn, None,
            C.byref(cfw), float(output_rate), float(bandwidth), float(quality), int(fft_len), C.byref(cffw),
            int(bool(center_dc)), out.ctypes.data, cap)
    if int(threads) == 4:
        frames = getattr(lib(), f"rro_chain_run_mt_{suf}")(*args, int(batch))
    elif int(threads) == 1:
        frames = getattr(lib(), f"rro_chain_run_{suf}")(*args)
    else:
        raise ValueError("threads must be 1 or 4 (one per block)")
    if frames == C.c_size_t(-1).value:
        raise AssertionError("chain contract violated")
    return out[: min(frames, cap)], frames


# --------------------------------------------------------------------------
# metering (metering.rs) and GainControl (transform.rs:29-92)
# --------------------------------------------------------------------------
def level(chunk, flt=np.float64) -> float:
    suf, cdt, _ = _dt(flt)
    x = _cin(chunk, cdt)
    return getattr(lib(), f"rro_level_{suf}")(x.ctypes.data, x.size)


def bandwidth(double_percentile, sample_rate, bins, flt=np.float64) -> float:
    suf, cdt, _ = _dt(flt)
    x = _cin(bins, cdt)
    return getattr(lib(), f"rro_bandwidth_{suf}")(float(double_percentile), float(sample_rate), x.ctypes.data, x.size)


def rescale_energy(resolution, input_, flt=np.float64):
    suf, cdt, rdt = _dt(flt)
    x = _cin(input_, cdt)
    out = np.empty(int(resolution), dtype=rdt)
    if getattr(lib(), f"rro_rescale_energy_{suf}")(out.ctypes.data, int(resolution), x.ctypes.data, x.size) != 0:
        raise AssertionError("assert!(n > 0)")
    return out


def gain(g, chunk, flt=np.float32):
    suf, cdt, _ = _dt(flt)
    x = _cin(chunk, cdt)
    y = np.empty_like(x)
    getattr(lib(), f"rro_gain_{suf}")(float(g), x.ctypes.data, x.size, y.ctypes.data)
    return y
