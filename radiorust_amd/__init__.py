"""radiorust_amd — MI355X (gfx950) backend for radiorust's IQ-stream hot path.

The product is the HIP library behind include/radiorust_amd.h
(radiorust_amd/csrc/, built into radiorust_amd/lib/libradiorust_amd.so).  This
package is the thin host-side mirror of the reference's block API used by the
tests and the benchmark.  Importing it does not need a GPU; creating a block does.
"""
from . import _lib  # noqa: F401
from .blocks import (Chain, ChainBank, Channelizer, Downsampler, Filter, FmDemod, Fourier, FreqShifter, Meter, Stft, Upsampler,  # noqa: F401
                     deemphasis_factor, fourier_route, sample_freq_resp, synth_iq_dev)
from .signal import Disconnection, Event, EventSignal, Samples, SamplesLost  # noqa: F401
from .windowing import CustomWindow, Kaiser, Rectangular, Window  # noqa: F401

__all__ = [
    "Chain", "ChainBank", "Channelizer", "Downsampler", "Filter", "FmDemod", "Fourier", "FreqShifter", "Meter", "Stft", "Upsampler", "deemphasis_factor", "fourier_route", "sample_freq_resp",
    "synth_iq_dev",
    "Disconnection", "Event", "EventSignal", "Samples", "SamplesLost",
    "CustomWindow", "Kaiser", "Rectangular", "Window",
]
