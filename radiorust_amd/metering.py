"""Host-side mirror of src/metering.rs and of the GainControl block
(src/blocks/transform.rs:29-92) over the C ABI; the arithmetic runs on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .blocks import _dtype_code
from .signal import Samples


def _arr(x, dtype):
    code, cdt = _dtype_code(dtype)
    return code, np.ascontiguousarray(x, dtype=cdt)


def level(chunk, dtype=np.float32, device: int = 0) -> float:
    """metering.rs:21-30 — average |x|^2 (f64)."""
    code, x = _arr(chunk, dtype)
    out = C.c_double()
    _lib.check(_lib.lib().rr_level(code, device, x.ctypes.data, x.size, C.byref(out)))
    return out.value


def bandwidth(double_percentile: float, sample_rate: float, bins, dtype=np.float32, device: int = 0) -> float:
    """metering.rs:41-80."""
    code, x = _arr(bins, dtype)
    out = C.c_double()
    _lib.check(_lib.lib().rr_bandwidth(code, device, float(double_percentile), float(sample_rate), x.ctypes.data, x.size, C.byref(out)))
    return out.value


def rescale_energy(resolution: int, input_, dtype=np.float32, device: int = 0) -> np.ndarray:
    """metering.rs:89-109 — returns the `output` vector."""
    code, x = _arr(input_, dtype)
    out = np.empty(int(resolution), dtype=np.dtype(dtype))
    _lib.check(_lib.lib().rr_rescale_energy(code, device, x.ctypes.data, x.size, int(resolution), out.ctypes.data))
    return out


class GainControl:
    """Block which applies a configurable gain (transform.rs:29-92)."""

    def __init__(self, gain: float, dtype=np.float32, device: int = 0):
        self._gain = float(gain)
        self._dtype = dtype
        self._device = device

    @classmethod
    def new(cls, gain, **kw):
        return cls(gain, **kw)

    def get(self) -> float:
        return self._gain

    def set(self, gain: float):
        self._gain = float(gain)

    def process(self, signal):
        if signal.is_event():
            return [signal]
        code, x = _arr(signal.chunk, self._dtype)
        y = np.empty_like(x)
        _lib.check(_lib.lib().rr_gain(code, self._device, self._gain, x.ctypes.data, x.size, y.ctypes.data))
        return [Samples(signal.sample_rate, y)]
