"""Host-side mirror of the reference's hot-path blocks over the C ABI.

Same constructor names, argument meaning and error behaviour as
`radiorust::blocks::{FreqShifter, Filter, Downsampler, Fourier}`; each block's
`process(signal)` is the body of the reference block's task loop for one
received message (it returns the list of messages the task would `send`).
All arithmetic happens in the HIP library; this module only evaluates the user's
closures (frequency response, custom windows) and moves messages.

Python is the test/bench harness language here (the reference's own host
language, Rust, has no toolchain in this image): see INTEGRATION.md for the
Rust shim a maintainer would add on top of the same C ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .signal import EventSignal, Samples
from .windowing import Kaiser, Rectangular, Window


def _dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return _lib.RR_F32, np.complex64
    if dtype == np.float64:
        return _lib.RR_F64, np.complex128
    raise TypeError(f"Flt must be float32 or float64, not {dtype}")


class _Block:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    # -- plumbing shared by all handles ---------------------------------
    def set_stream(self, hip_stream: int | None):
        """Run on the caller's hipStream_t (e.g. torch's current stream)."""
        _lib.check(_lib.lib().rr_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def wait(self):
        _lib.check(_lib.lib().rr_wait(self._h))

    def query(self) -> bool:
        s = _lib.lib().rr_query(self._h)
        if s == _lib.RR_ERR_NOT_READY:
            return False
        _lib.check(s)
        return True

    def close(self):
        if self._h:
            getattr(_lib.lib(), self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _host_call(self, fn, rate_args, chunk, cap):
        x = np.ascontiguousarray(chunk, dtype=self._cdt)
        out = np.empty(cap, dtype=self._cdt)
        n_out = C.c_size_t()
        _lib.check(fn(self._h, *rate_args, x.ctypes.data, x.size, out.ctypes.data, cap, C.byref(n_out)))
        return out[: n_out.value]


class FreqShifter(_Block):
    """Complex oscillator and mixer (transform.rs:266-391)."""

    _destroy = "rr_freqshifter_destroy"

    def __init__(self, precision: float = 1.0, shift: float = 0.0, dtype=np.float32, device: int = 0):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        _lib.check(_lib.lib().rr_freqshifter_create(self._code, float(precision), float(shift), device, C.byref(self._h)))

    # transform.rs:282-297
    @classmethod
    def new(cls, **kw):
        return cls(1.0, 0.0, **kw)

    @classmethod
    def with_shift(cls, shift, **kw):
        return cls(1.0, shift, **kw)

    @classmethod
    def with_precision(cls, precision, **kw):
        return cls(precision, 0.0, **kw)

    @classmethod
    def with_precision_and_shift(cls, precision, shift, **kw):
        return cls(precision, shift, **kw)

    def precision(self) -> float:
        v = C.c_double()
        _lib.check(_lib.lib().rr_freqshifter_precision(self._h, C.byref(v)))
        return v.value

    def shift(self) -> float:
        v = C.c_double()
        _lib.check(_lib.lib().rr_freqshifter_shift(self._h, C.byref(v)))
        return v.value

    def set_shift(self, shift: float):
        _lib.check(_lib.lib().rr_freqshifter_set_shift(self._h, float(shift)))

    def update_shift(self, modify):
        """transform.rs:388-390: `modify` maps the old shift to the new one."""
        self.set_shift(modify(self.shift()))

    def process(self, signal):
        if signal.is_event():
            return [signal]  # transform.rs:357-359
        y = self._host_call(_lib.lib().rr_freqshifter_process, (float(signal.sample_rate),), signal.chunk, len(signal.chunk))
        return [Samples(signal.sample_rate, y)]

    def process_dev(self, sample_rate, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_freqshifter_process_dev(self._h, float(sample_rate), d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


def deemphasis_factor(tau: float, frequency: float) -> complex:
    """blocks::filters::deemphasis_factor (filters.rs:20-27): 1 / (1 + j tau 2 pi f)."""
    out = (C.c_double * 2)()
    _lib.check(_lib.lib().rr_deemphasis_factor(float(tau), float(frequency), out))
    return complex(out[0], out[1])


def fourier_route(n: int, dtype=np.float32) -> str:
    """Which kernels transform a chunk of n samples (rr_fourier_route: host only, the decision rr_fourier_process takes)."""
    buf = C.create_string_buffer(128)
    _lib.check(_lib.lib().rr_fourier_route(0 if np.dtype(dtype) == np.float32 else 1, int(n), buf, 128))
    return buf.value.decode()


def sample_freq_resp(freq_resp, n: int, sample_rate: float) -> np.ndarray:
    """Evaluates the user's closure exactly where the reference does
    (filters.rs:188-199): bins 0..=(n-1)/2 and their negatives; for even n the
    Nyquist bin stays zero."""
    resp = np.zeros(n, dtype=np.complex128)
    step = sample_rate / n
    for i in range((n - 1) // 2 + 1):
        resp[i] = complex(freq_resp(i, i * step))
        if i > 0:
            resp[n - i] = complex(freq_resp(-i, -(i * step)))
    return resp


class Filter(_Block):
    """General purpose frequency filter (filters.rs:110-298).  `freq_resp(bin,
    freq) -> complex` stays on the host; the library gets its samples."""

    _destroy = "rr_filter_destroy"

    def __init__(self, freq_resp, window: Window | None = None, dtype=np.float32, device: int = 0):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        self._freq_resp = freq_resp
        self._window = window if window is not None else Kaiser.with_null_at_bin(2.0)
        _lib.check(_lib.lib().rr_filter_create(self._code, device, C.byref(self._h)))

    def set_gain(self, gain: float):
        """A GainControl wired behind this block (transform.rs:29-92), folded into the block's tables: no pass of its own."""
        _lib.check(_lib.lib().rr_filter_set_gain(self._h, float(gain)))
        return self

    # filters.rs:128-152
    @classmethod
    def new(cls, freq_resp, **kw):
        return cls(freq_resp, Kaiser.with_null_at_bin(2.0), **kw)

    @classmethod
    def new_rectangular(cls, freq_resp, **kw):
        return cls(freq_resp, Rectangular(), **kw)

    @classmethod
    def with_window(cls, freq_resp, window, **kw):
        return cls(freq_resp, window, **kw)

    def update(self, freq_resp):
        self._freq_resp = freq_resp
        _lib.check(_lib.lib().rr_filter_mark_params_changed(self._h))

    def update_with_window(self, freq_resp, window):
        self._freq_resp = freq_resp
        self._window = window
        _lib.check(_lib.lib().rr_filter_mark_params_changed(self._h))

    def _ensure_design(self, sample_rate: float, n: int):
        needed = C.c_int()
        _lib.check(_lib.lib().rr_filter_needs_design(self._h, sample_rate, n, C.byref(needed)))
        if needed.value:
            resp = sample_freq_resp(self._freq_resp, n, sample_rate)
            win = self._window.sample(n)
            _lib.check(_lib.lib().rr_filter_design(self._h, sample_rate, n, resp.ctypes.data, win.ctypes.data))

    def process(self, signal):
        if signal.is_event():
            if signal.event.is_interrupt():  # filters.rs:262-265
                _lib.check(_lib.lib().rr_filter_reset(self._h))
            return [signal]
        n = len(signal.chunk)
        self._ensure_design(float(signal.sample_rate), n)
        y = self._host_call(_lib.lib().rr_filter_process, (float(signal.sample_rate),), signal.chunk, n)
        return [Samples(signal.sample_rate, y)] if len(y) else []

    def process_dev(self, sample_rate, chunk_len: int, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        """n_in = k * chunk_len consecutive chunks resident on the device."""
        self._ensure_design(float(sample_rate), chunk_len)
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_filter_process_dev(self._h, float(sample_rate), d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


    def last_kernel(self) -> int:
        """0 k_fir, 1 k_filter_ols, 2 k_filter_ols4096, 3 k_filter_wave."""
        v = C.c_int()
        _lib.check(_lib.lib().rr_filter_last_kernel(self._h, C.byref(v)))
        return v.value

    def process_dev_f16(self, sample_rate, chunk_len: int, d_in: int, n_in: int, d_out_f16: int, cap: int,
                        response_f16: bool = False) -> int:
        """As process_dev with half-precision output pairs (and optionally a half-precision response table)."""
        self._ensure_design(float(sample_rate), int(chunk_len))
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_filter_process_dev_f16(self._h, float(sample_rate), d_in, n_in, d_out_f16, cap,
                                                         C.byref(n_out), int(bool(response_f16))))
        return n_out.value


class Downsampler(_Block):
    """Reduce sample rate (resampling.rs:14-146)."""

    _destroy = "rr_downsampler_destroy"

    def __init__(self, output_chunk_len: int, output_rate: float, bandwidth: float, quality: float = 3.0,
                 dtype=np.float32, device: int = 0):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        self.output_chunk_len = int(output_chunk_len)
        self.output_rate = float(output_rate)
        _lib.check(_lib.lib().rr_downsampler_create(self._code, float(output_rate), float(bandwidth), float(quality), device, C.byref(self._h)))
        self._pending = np.empty(0, dtype=self._cdt)  # the partly filled output_chunk

    @classmethod
    def new(cls, output_chunk_len, output_rate, bandwidth, **kw):
        return cls(output_chunk_len, output_rate, bandwidth, 3.0, **kw)

    @classmethod
    def with_quality(cls, output_chunk_len, output_rate, bandwidth, quality, **kw):
        return cls(output_chunk_len, output_rate, bandwidth, quality, **kw)

    def set_gain(self, gain: float):
        """A GainControl wired behind this block (examples/relm_app/simple_receiver.rs:52-56), folded into the impulse response."""
        _lib.check(_lib.lib().rr_downsampler_set_gain(self._h, float(gain)))
        return self

    def ir_len(self) -> int:
        v = C.c_size_t()
        _lib.check(_lib.lib().rr_downsampler_ir_len(self._h, C.byref(v)))
        return v.value

    def last_kernel(self) -> int:
        """0 = k_fir; 1 / 2 / 3 = the chain's fused kernels (k_mix_fir_decim / k_ols_decim4 / k_ols_wave)."""
        v = C.c_int()
        _lib.check(_lib.lib().rr_downsampler_last_kernel(self._h, C.byref(v)))
        return v.value

    def process_raw(self, sample_rate, chunk) -> np.ndarray:
        """Outputs produced by this input chunk, not yet regrouped."""
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_downsampler_peek(self._h, float(sample_rate), len(chunk), C.byref(n_out)))
        return self._host_call(_lib.lib().rr_downsampler_process, (float(sample_rate),), chunk, n_out.value)

    def process(self, signal):
        if signal.is_event():
            return [signal]  # resampling.rs:135-137 (no reset)
        self._pending = np.concatenate([self._pending, self.process_raw(signal.sample_rate, signal.chunk)])
        out = []
        L = self.output_chunk_len
        while len(self._pending) >= L:  # resampling.rs:121-131
            out.append(Samples(self.output_rate, self._pending[:L].copy()))
            self._pending = self._pending[L:]
        return out

    def process_dev(self, sample_rate, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_downsampler_process_dev(self._h, float(sample_rate), d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


class Upsampler(_Block):
    """Increase sample rate (resampling.rs:147-280)."""

    _destroy = "rr_upsampler_destroy"

    def __init__(self, output_chunk_len: int, output_rate: float, bandwidth: float, quality: float = 3.0,
                 dtype=np.float32, device: int = 0):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        self.output_chunk_len = int(output_chunk_len)
        self.output_rate = float(output_rate)
        _lib.check(_lib.lib().rr_upsampler_create(self._code, float(output_rate), float(bandwidth), float(quality), device, C.byref(self._h)))
        self._pending = np.empty(0, dtype=self._cdt)  # the partly filled output_chunk

    @classmethod
    def new(cls, output_chunk_len, output_rate, bandwidth, **kw):
        return cls(output_chunk_len, output_rate, bandwidth, 3.0, **kw)

    @classmethod
    def with_quality(cls, output_chunk_len, output_rate, bandwidth, quality, **kw):
        return cls(output_chunk_len, output_rate, bandwidth, quality, **kw)

    def ir_len(self) -> int:
        v = C.c_size_t()
        _lib.check(_lib.lib().rr_upsampler_ir_len(self._h, C.byref(v)))
        return v.value

    def process_raw(self, sample_rate, chunk) -> np.ndarray:
        """Outputs released by this input chunk, not yet regrouped."""
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_upsampler_peek(self._h, float(sample_rate), len(chunk), C.byref(n_out)))
        return self._host_call(_lib.lib().rr_upsampler_process, (float(sample_rate),), chunk, n_out.value)

    def process(self, signal):
        if signal.is_event():
            return [signal]  # resampling.rs:269-271
        self._pending = np.concatenate([self._pending, self.process_raw(signal.sample_rate, signal.chunk)])
        out = []
        L = self.output_chunk_len
        while len(self._pending) >= L:  # resampling.rs:251-261
            out.append(Samples(self.output_rate, self._pending[:L].copy()))
            self._pending = self._pending[L:]
        return out

    def process_dev(self, sample_rate, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_upsampler_process_dev(self._h, float(sample_rate), d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


class FmDemod(_Block):
    """FM demodulator (modulation.rs:83-158)."""

    _destroy = "rr_fmdemod_destroy"

    def __init__(self, deviation: float, dtype=np.float32, device: int = 0):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        _lib.check(_lib.lib().rr_fmdemod_create(self._code, float(deviation), device, C.byref(self._h)))

    def deviation(self) -> float:
        v = C.c_double()
        _lib.check(_lib.lib().rr_fmdemod_deviation(self._h, C.byref(v)))
        return v.value

    def set_deviation(self, deviation: float):
        _lib.check(_lib.lib().rr_fmdemod_set_deviation(self._h, float(deviation)))
        return self

    def set_gain(self, gain: float):
        """A GainControl wired behind the demodulator, applied on the store (bit-equal to the two blocks one after the other)."""
        _lib.check(_lib.lib().rr_fmdemod_set_gain(self._h, float(gain)))
        return self

    def process_raw(self, sample_rate, chunk) -> np.ndarray:
        return self._host_call(_lib.lib().rr_fmdemod_process, (float(sample_rate),), chunk, len(chunk))

    def process(self, signal):
        if signal.is_event():
            if signal.event.is_interrupt():  # modulation.rs:145-149
                _lib.check(_lib.lib().rr_fmdemod_reset(self._h))
            return [signal]
        return [Samples(signal.sample_rate, self.process_raw(signal.sample_rate, signal.chunk))]

    def process_dev(self, sample_rate, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_fmdemod_process_dev(self._h, float(sample_rate), d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


class Fourier(_Block):
    """Windowed Fourier analysis (analysis.rs:26-133)."""

    _destroy = "rr_fourier_destroy"

    def __init__(self, window: Window | None = None, center_dc: bool = False, dtype=np.float32, device: int = 0):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        self._window = window if window is not None else Rectangular()
        spec = self._window._spec()
        self._sampled_for = None
        if spec is None:
            spec = _lib.Window(_lib.RR_WIN_SAMPLED, 0.0)
        _lib.check(_lib.lib().rr_fourier_create(self._code, spec, int(bool(center_dc)), device, C.byref(self._h)))
        self._is_sampled = spec.kind == _lib.RR_WIN_SAMPLED

    # analysis.rs:39-59
    @classmethod
    def new(cls, **kw):
        return cls(Rectangular(), False, **kw)

    @classmethod
    def new_center_dc(cls, **kw):
        return cls(Rectangular(), True, **kw)

    @classmethod
    def with_window(cls, window, **kw):
        return cls(window, False, **kw)

    @classmethod
    def with_window_center_dc(cls, window, **kw):
        return cls(window, True, **kw)

    def _ensure_window(self, n: int):
        if self._is_sampled and self._sampled_for != n:
            vals = self._window.sample(n)
            _lib.check(_lib.lib().rr_fourier_set_sampled_window(self._h, n, vals.ctypes.data))
            self._sampled_for = n

    def process(self, signal):
        if signal.is_event():
            return [signal]  # analysis.rs:122-124
        n = len(signal.chunk)
        self._ensure_window(n)
        y = self._host_call(_lib.lib().rr_fourier_process, (), signal.chunk, n)
        return [Samples(signal.sample_rate, y)]

    def process_dev(self, chunk_len: int, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        self._ensure_window(chunk_len)
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_fourier_process_dev(self._h, chunk_len, d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


class Chain(_Block):
    """FreqShifter -> [Rechunker(filter_len)] -> Filter -> Downsampler(fft_len)
    -> Fourier on one device without host hops (the wiring of
    examples/bandwidth_meter/main.rs:51-72).  Emits spectra of `fft_len` bins."""

    _destroy = "rr_chain_destroy"

    def __init__(self, *, shift: float, precision: float = 1.0, filter_len: int, freq_resp,
                 filter_window: Window | None = None, output_rate: float, bandwidth: float, quality: float = 3.0,
                 fft_len: int, fft_window: Window | None = None, center_dc: bool = False, dtype=np.float32,
                 device: int = 0, allow_fused: bool = True):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        self._freq_resp = freq_resp
        self._filter_window = filter_window if filter_window is not None else Kaiser.with_null_at_bin(2.0)
        fw = fft_window if fft_window is not None else Rectangular()
        spec = fw._spec()
        if spec is None:
            raise TypeError("Chain needs a built-in fft_window (Rectangular or Kaiser)")
        self.filter_len = int(filter_len)
        self.fft_len = int(fft_len)
        self.output_rate = float(output_rate)
        self._rate = None  # sample rate of the last Samples message (what the Rechunker's patchwork carries)
        p = _lib.ChainParams(self._code, float(precision), float(shift), self.filter_len, float(output_rate),
                             float(bandwidth), float(quality), self.fft_len, spec, int(bool(center_dc)),
                             int(bool(allow_fused)))
        _lib.check(_lib.lib().rr_chain_create(p, device, C.byref(self._h)))

    def set_shift(self, shift: float):
        _lib.check(_lib.lib().rr_chain_set_shift(self._h, float(shift)))

    def update_filter(self, freq_resp, window: Window | None = None):
        self._freq_resp = freq_resp
        if window is not None:
            self._filter_window = window
        _lib.check(_lib.lib().rr_chain_filter_mark_params_changed(self._h))

    def interrupt(self):
        _lib.check(_lib.lib().rr_chain_interrupt(self._h))

    def _ensure_design(self, sample_rate: float):
        needed = C.c_int()
        _lib.check(_lib.lib().rr_chain_filter_needs_design(self._h, sample_rate, C.byref(needed)))
        if needed.value:
            resp = sample_freq_resp(self._freq_resp, self.filter_len, sample_rate)
            win = self._filter_window.sample(self.filter_len)
            _lib.check(_lib.lib().rr_chain_filter_design(self._h, sample_rate, resp.ctypes.data, win.ctypes.data))

    def peek(self, sample_rate, n_in: int) -> int:
        self._ensure_design(float(sample_rate))
        v = C.c_size_t()
        _lib.check(_lib.lib().rr_chain_peek(self._h, float(sample_rate), n_in, C.byref(v)))
        return v.value

    def last_path_fused(self) -> bool:
        v = C.c_int()
        _lib.check(_lib.lib().rr_chain_last_path(self._h, C.byref(v)))
        return bool(v.value)

    def set_metering(self, double_percentile, d_bandwidth: int, cap_frames: int, d_energy: int = 0, store_spectra: bool = True):
        """As Meter.set_metering: bandwidth (and energy) of every spectrum the chain produces, from the kernel that makes it."""
        _lib.check(_lib.lib().rr_chain_set_metering(self._h, float(double_percentile), d_bandwidth or None, d_energy or None,
                                                    int(cap_frames), int(bool(store_spectra))))

    def last_path_kernel(self) -> str:
        """Name of the mix + FIR + decimate kernel the last call ran ("" = block-by-block)."""
        v = C.c_int()
        _lib.check(_lib.lib().rr_chain_last_path(self._h, C.byref(v)))
        return ["", "k_mix_fir_decim", "k_ols_decim4", "k_ols_wave", "k_ols_frame", "k_decim_poly", "k_ols_frame", "k_ols_wave", "k_ols_frame", "k_ols_wave", "", "k_ols4096_f64"][v.value]

    def last_path_mixer_folded(self) -> bool:
        """True if the last call ran k_ols_frame<true> or k_ols_wave<4, true, true>: the mixer folded into the response tables
        (NCO periods that divide 8), or the GP instances of the two: the mixer behind the filter (every other period)."""
        v = C.c_int()
        _lib.check(_lib.lib().rr_chain_last_path(self._h, C.byref(v)))
        return v.value in (6, 7, 8, 9)

    def pending(self) -> int:
        """Samples the Rechunker in front of the Filter holds (chunks.rs:62-64)."""
        n = C.c_size_t()
        _lib.check(_lib.lib().rr_chain_pending(self._h, C.byref(n)))
        return n.value

    def process(self, signal):
        """One message through the four blocks.  The Rechunker in front of the Filter drops its leftover samples and
        sends SamplesLost - an interrupt for the Filter - when an event arrives or the sample rate changes while it
        holds some (chunks.rs:72-92); an interrupting event resets the Filter (filters.rs:262-265)."""
        from .signal import SamplesLost

        if signal.is_event():
            out = []
            if self.pending():
                self.interrupt()
                out.append(EventSignal(SamplesLost()))
            elif signal.event.is_interrupt():
                self.interrupt()
            return out + [signal]
        pre = []
        rate = float(signal.sample_rate)
        if self._rate is not None and rate != self._rate and self.pending():
            self.interrupt()
            pre = [EventSignal(SamplesLost())]
        self._rate = rate
        frames = self.peek(rate, len(signal.chunk))
        y = self._host_call(_lib.lib().rr_chain_process, (rate,), signal.chunk, frames * self.fft_len)
        return pre + [Samples(self.output_rate, y[i * self.fft_len : (i + 1) * self.fft_len]) for i in range(frames)]

    def process_dev(self, sample_rate, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        self._ensure_design(float(sample_rate))
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_chain_process_dev(self._h, float(sample_rate), d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


class Channelizer(_Block):
    """Polyphase FFT channelizer (BASELINE configs[2]): the reference composition
    Rechunker(bins) -> Overlapper(taps_per_branch) -> Fourier.with_window(window) ->
    every taps_per_branch-th bin (chunks.rs:42-242, analysis.rs:60-132) as one
    fold + `bins`-point FFT per hop.  Input chunks must be multiples of the hop.
    hop < bins: the oversampled filterbank = Rechunker(hop) -> Overlapper(bins * taps_per_branch / hop) -> the same
    Fourier and bin selection; any number of bins works (powers of two with hop = bins take the fused kernels)."""

    _destroy = "rr_channelizer_destroy"

    def __init__(self, bins: int, taps_per_branch: int, window: Window | None = None, dtype=np.float32, device: int = 0,
                 hop: int | None = None):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        self.bins, self.taps_per_branch = int(bins), int(taps_per_branch)
        self.hop = int(hop) if hop else self.bins
        w = window if window is not None else Kaiser.with_null_at_bin(float(taps_per_branch))
        spec = w._spec()
        if spec is None:
            raise TypeError("Channelizer needs a built-in window (Rectangular or Kaiser)")
        _lib.check(_lib.lib().rr_channelizer_create_ex(self._code, self.bins, self.taps_per_branch, self.hop, spec, device,
                                                       C.byref(self._h)))

    def process(self, signal):
        """Each output message is one frame of `bins` channel samples; the frame rate is
        sample_rate / bins.  Any event resets the overlap history and is preceded by
        SamplesLost, as the Overlapper does (chunks.rs:225-233)."""
        from .signal import SamplesLost

        if signal.is_event():
            _lib.check(_lib.lib().rr_channelizer_reset(self._h))
            return [EventSignal(SamplesLost()), signal]
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_channelizer_peek(self._h, len(signal.chunk), C.byref(n_out)))
        y = self._host_call(_lib.lib().rr_channelizer_process, (), signal.chunk, n_out.value)
        M = self.bins
        return [Samples(signal.sample_rate, y[i * M : (i + 1) * M]) for i in range(len(y) // M)]

    def process_dev(self, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_channelizer_process_dev(self._h, d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


class Stft(_Block):
    """Overlapped Fourier analysis: Rechunker(chunk_len) -> Overlapper(chunk_count) ->
    Fourier.with_window(window) (chunks.rs:42-242, analysis.rs:26-133; the wiring of
    examples/bandwidth_meter/main.rs:66-69).  Input chunks may have any length."""

    _destroy = "rr_stft_destroy"

    def __init__(self, chunk_len: int, chunk_count: int, window: Window | None = None, center_dc: bool = False,
                 dtype=np.float32, device: int = 0):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        self.chunk_len, self.chunk_count = int(chunk_len), int(chunk_count)
        self._rate = None
        self._rates = []  # sample rates of the chunks in the Overlapper's history
        w = window if window is not None else Rectangular()
        spec = w._spec()
        if spec is None:
            raise TypeError("Stft needs a built-in window (Rectangular or Kaiser)")
        _lib.check(_lib.lib().rr_stft_create(self._code, self.chunk_len, self.chunk_count, spec, int(bool(center_dc)),
                                             device, C.byref(self._h)))

    def pending(self) -> int:
        """Samples the Rechunker holds (chunks.rs:62-64)."""
        n = C.c_size_t()
        _lib.check(_lib.lib().rr_stft_pending(self._h, C.byref(n)))
        return n.value

    def process(self, signal):
        """One output message per overlapped chunk (chunk_len * chunk_count bins), events as the composition
        Rechunker -> Overlapper -> Fourier passes them on:
        * an event: the Rechunker sends SamplesLost first if it holds samples, and drops them (chunks.rs:80-88); the
          Overlapper answers EVERY event with a reset of its history and a SamplesLost of its own in front of it
          (chunks.rs:225-233);
        * a new sample rate: the Rechunker drops what it holds - if it holds anything - and sends SamplesLost
          (chunks.rs:72-79), which resets the Overlapper; with nothing pending nothing is lost, the Overlapper keeps
          its history and labels each output with the length-weighted average rate of its chunks (chunks.rs:207-213)."""
        from .signal import SamplesLost

        if signal.is_event():
            out = []
            if self.pending():
                out += [EventSignal(SamplesLost()), EventSignal(SamplesLost())]  # the Overlapper's, the Rechunker's
            _lib.check(_lib.lib().rr_stft_reset(self._h))
            self._rates = []
            return out + [EventSignal(SamplesLost()), signal]
        pre = []
        rate = float(signal.sample_rate)
        if self._rate is not None and rate != self._rate and self.pending():
            _lib.check(_lib.lib().rr_stft_reset(self._h))
            self._rates = []
            pre = [EventSignal(SamplesLost()), EventSignal(SamplesLost())]
        self._rate = rate
        chunks = (self.pending() + len(signal.chunk)) // self.chunk_len  # chunks this message completes
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_stft_peek(self._h, len(signal.chunk), C.byref(n_out)))
        y = self._host_call(_lib.lib().rr_stft_process, (), signal.chunk, n_out.value)
        N = self.chunk_len * self.chunk_count
        out, k = [], 0
        for _ in range(chunks):
            self._rates.append(rate)
            if len(self._rates) >= self.chunk_count:
                acc, cnt = 0.0, 0
                for r in self._rates:  # chunks.rs:207-213, same order of operations
                    cnt += self.chunk_len
                    acc += r * float(self.chunk_len)
                out.append(Samples(acc / float(cnt), y[k * N : (k + 1) * N]))
                k += 1
                self._rates.pop(0)
        assert k == len(y) // N
        return pre + out

    def process_dev(self, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_stft_process_dev(self._h, d_in, n_in, d_out or None, cap, C.byref(n_out)))
        return n_out.value

    def set_metering(self, double_percentile, sample_rate, d_bandwidth: int, cap_frames: int, d_energy: int = 0,
                     store_spectra: bool = True):
        """metering::bandwidth(double_percentile, sample_rate, spectrum) per spectrum, written to the device array
        d_bandwidth (f64) by the kernel that makes the spectrum (see Meter.set_metering)."""
        _lib.check(_lib.lib().rr_stft_set_metering(self._h, float(double_percentile), float(sample_rate), d_bandwidth or None,
                                                   d_energy or None, int(cap_frames), int(bool(store_spectra))))


class ChainBank(_Block):
    """K chains with the same parameters whose streams advance in lockstep (rr_chainbank_*): a call gives every channel
    the same number of samples (device pointers, channel k at d_in + k * in_stride samples) and, once all channels are in the
    steady fused state, runs as two launches for all of them.  Each channel's spectra are bit-identical to a stand-alone
    Chain's."""

    _destroy = "rr_chainbank_destroy"

    def __init__(self, channels: int, *, shift: float, precision: float = 1.0, filter_len: int, freq_resp,
                 filter_window: Window | None = None, output_rate: float, bandwidth: float, quality: float = 3.0,
                 fft_len: int, fft_window: Window | None = None, center_dc: bool = False, dtype=np.float32,
                 device: int = 0, allow_fused: bool = True):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        self._freq_resp = freq_resp
        self._filter_window = filter_window if filter_window is not None else Kaiser.with_null_at_bin(2.0)
        fw = fft_window if fft_window is not None else Rectangular()
        spec = fw._spec()
        if spec is None:
            raise TypeError("ChainBank needs a built-in fft_window (Rectangular or Kaiser)")
        self.channels, self.filter_len, self.fft_len = int(channels), int(filter_len), int(fft_len)
        p = _lib.ChainParams(self._code, float(precision), float(shift), self.filter_len, float(output_rate),
                             float(bandwidth), float(quality), self.fft_len, spec, int(bool(center_dc)),
                             int(bool(allow_fused)))
        _lib.check(_lib.lib().rr_chainbank_create(p, self.channels, device, C.byref(self._h)))

    def set_shift(self, shift: float):
        _lib.check(_lib.lib().rr_chainbank_set_shift(self._h, float(shift)))

    def interrupt(self):
        _lib.check(_lib.lib().rr_chainbank_interrupt(self._h))

    def _ensure_design(self, sample_rate: float):
        needed = C.c_int()
        _lib.check(_lib.lib().rr_chainbank_filter_needs_design(self._h, sample_rate, C.byref(needed)))
        if needed.value:
            resp = sample_freq_resp(self._freq_resp, self.filter_len, sample_rate)
            win = self._filter_window.sample(self.filter_len)
            _lib.check(_lib.lib().rr_chainbank_filter_design(self._h, sample_rate, resp.ctypes.data, win.ctypes.data))

    def peek(self, sample_rate, n_in: int) -> int:
        self._ensure_design(float(sample_rate))
        v = C.c_size_t()
        _lib.check(_lib.lib().rr_chainbank_peek(self._h, float(sample_rate), n_in, C.byref(v)))
        return v.value

    def process_dev(self, sample_rate, d_in: int, in_stride: int, n_in: int, d_out: int, out_stride: int, cap: int) -> int:
        """Bins written per channel."""
        self._ensure_design(float(sample_rate))
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_chainbank_process_dev(self._h, float(sample_rate), d_in, in_stride, n_in, d_out, out_stride,
                                                       cap, C.byref(n_out)))
        return n_out.value

    def last_path_lockstep(self) -> bool:
        v = C.c_int()
        _lib.check(_lib.lib().rr_chainbank_last_path(self._h, C.byref(v)))
        return bool(v.value)


class Meter(_Block):
    """The reference's own hot-path caller in ITS order (examples/bandwidth_meter/main.rs:53-69) on one device without
    host hops: FreqShifter(shift) -> Downsampler(chunk_len, output_rate, bandwidth, quality) -> Filter(freq_resp; it
    sees chunks of chunk_len at output_rate) -> Overlapper(overlap) -> Fourier.with_window(fft_window).  Output: one
    message per overlapped spectrum (chunk_len * overlap bins) at the output rate; `metering.bandwidth` on each of
    them is the example's last step (rr_bandwidth_dev keeps that on the device too)."""

    _destroy = "rr_meter_destroy"

    def __init__(self, *, shift: float, precision: float = 1.0, output_rate: float, bandwidth: float, quality: float = 3.0,
                 chunk_len: int, freq_resp, filter_window: Window | None = None, overlap: int, fft_window: Window | None = None,
                 center_dc: bool = False, dtype=np.float32, device: int = 0):
        super().__init__()
        self._code, self._cdt = _dtype_code(dtype)
        fw = fft_window if fft_window is not None else Rectangular()
        spec = fw._spec()
        if spec is None:
            raise TypeError("Meter needs a built-in fft_window (Rectangular or Kaiser)")
        self.chunk_len, self.overlap, self.output_rate = int(chunk_len), int(overlap), float(output_rate)
        p = _lib.MeterParams(self._code, float(precision), float(shift), float(output_rate), float(bandwidth), float(quality),
                             self.chunk_len, self.overlap, spec, int(bool(center_dc)))
        _lib.check(_lib.lib().rr_meter_create(p, device, C.byref(self._h)))
        self._filter_window = filter_window if filter_window is not None else Kaiser.with_null_at_bin(2.0)
        self.update_filter(freq_resp)

    def set_shift(self, shift: float):
        _lib.check(_lib.lib().rr_meter_set_shift(self._h, float(shift)))

    def update_filter(self, freq_resp, window: Window | None = None):
        """Filter::update / update_with_window: the closure is sampled at the rate and chunk length the Filter sees."""
        if window is not None:
            self._filter_window = window
        resp = sample_freq_resp(freq_resp, self.chunk_len, self.output_rate)
        win = self._filter_window.sample(self.chunk_len)
        _lib.check(_lib.lib().rr_meter_filter_design(self._h, resp.ctypes.data, win.ctypes.data))

    def peek(self, sample_rate, n_in: int) -> int:
        n = C.c_size_t()
        _lib.check(_lib.lib().rr_meter_peek(self._h, float(sample_rate), n_in, C.byref(n)))
        return n.value

    def process(self, signal):
        from .signal import SamplesLost

        if signal.is_event():
            _lib.check(_lib.lib().rr_meter_event(self._h, int(signal.event.is_interrupt())))
            return [EventSignal(SamplesLost()), signal]  # the Overlapper's answer to every event (chunks.rs:225-233)
        frames = self.peek(signal.sample_rate, len(signal.chunk))
        N = self.chunk_len * self.overlap
        y = self._host_call(_lib.lib().rr_meter_process, (float(signal.sample_rate),), signal.chunk, frames * N)
        return [Samples(self.output_rate, y[i * N : (i + 1) * N]) for i in range(frames)]

    def set_metering(self, double_percentile, d_bandwidth: int, cap_frames: int, d_energy: int = 0, store_spectra: bool = True):
        """metering::bandwidth(double_percentile, output_rate, spectrum) per spectrum as the pipeline's last step
        (examples/bandwidth_meter/main.rs:78), written to the device array d_bandwidth (f64) by the kernel that makes the
        spectrum; d_bandwidth = 0 switches it off; store_spectra = False: the spectra themselves are not written."""
        _lib.check(_lib.lib().rr_meter_set_metering(self._h, float(double_percentile), d_bandwidth or None, d_energy or None,
                                                    int(cap_frames), int(bool(store_spectra))))

    def process_bandwidth(self, signal, double_percentile: float) -> np.ndarray:
        """The example's loop body as one call: samples in, one bandwidth per spectrum out; the spectra never leave the chip."""
        x = np.ascontiguousarray(signal.chunk, dtype=self._cdt)
        frames = self.peek(signal.sample_rate, len(x))
        out = np.empty(max(frames, 1), dtype=np.float64)
        n = C.c_size_t()
        _lib.check(_lib.lib().rr_meter_process_bandwidth(self._h, float(signal.sample_rate), x.ctypes.data, len(x),
                                                         float(double_percentile), out.ctypes.data, len(out), C.byref(n)))
        return out[: n.value]

    def front_fused(self) -> bool:
        """True when the last call ran FreqShifter + Downsampler as one kernel."""
        v = C.c_int()
        _lib.check(_lib.lib().rr_meter_last_path(self._h, C.byref(v)))
        return bool(v.value)

    def process_dev(self, sample_rate, d_in: int, n_in: int, d_out: int, cap: int) -> int:
        n_out = C.c_size_t()
        _lib.check(_lib.lib().rr_meter_process_dev(self._h, float(sample_rate), d_in, n_in, d_out, cap, C.byref(n_out)))
        return n_out.value


def synth_iq_dev(device: int, hip_stream: int | None, seed: int, t0: int, n: int, d_out: int):
    """Fills n complex64 samples of the synthetic IQ source on the device."""
    _lib.check(_lib.lib().rr_synth_iq_dev(device, C.c_void_p(hip_stream or 0), seed, t0, n, d_out))
