"""Window functions — host-side mirror of src/windowing.rs (windowing.rs:6-67).

Values come from the backend's own f64 design math (rr_kaiser_rel_with_beta,
csrc/rr_design.cpp), not from the test oracle.
"""
from __future__ import annotations

import numpy as np

from . import _lib


class Window:
    """`relative_value_at(x)`: value at x in [-1, 1], times an unknown constant
    (windowing.rs:6-10)."""

    def relative_value_at(self, x: float) -> float:
        raise NotImplementedError

    def sample(self, n: int) -> np.ndarray:
        """Values at the positions the blocks use: 2 (i + 0.5) / n - 1
        (filters.rs:209-212, analysis.rs:93-94)."""
        spec = self._spec()
        out = np.empty(n, dtype=np.float64)
        if spec is not None:
            _lib.check(_lib.lib().rr_window_sample(spec, n, out.ctypes.data))
        else:
            for i in range(n):
                out[i] = self.relative_value_at(2.0 * (i + 0.5) / n - 1.0)
        return out

    def _spec(self):
        """Built-in description for the C ABI, or None (sampled by the host)."""
        return None


class Rectangular(Window):
    def relative_value_at(self, x: float) -> float:
        return 1.0

    def _spec(self):
        return _lib.Window(_lib.RR_WIN_RECTANGULAR, 0.0)


class Kaiser(Window):
    def __init__(self, beta: float):
        self.beta = float(beta)

    @classmethod
    def with_beta(cls, beta: float) -> "Kaiser":
        return cls(beta)

    @classmethod
    def with_alpha(cls, alpha: float) -> "Kaiser":
        return cls(_lib.lib().rr_kaiser_alpha_to_beta(float(alpha)))

    @classmethod
    def with_null_at_bin(cls, n: float) -> "Kaiser":
        return cls(_lib.lib().rr_kaiser_null_at_bin_to_beta(float(n)))

    def relative_value_at(self, x: float) -> float:
        return _lib.lib().rr_kaiser_rel_with_beta(self.beta, float(x))

    def _spec(self):
        return _lib.Window(_lib.RR_WIN_KAISER, self.beta)


class CustomWindow(Window):
    """Window given by a closure `f(x) -> float` (windowing.rs:58-67)."""

    def __init__(self, fn):
        self.fn = fn

    def relative_value_at(self, x: float) -> float:
        return float(self.fn(x))
