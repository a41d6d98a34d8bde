"""ctypes binding of libradiorust_amd.so (include/radiorust_amd.h).

There is no fallback: if the shared object is missing this module raises, and
if no HIP device is usable every `*_create` returns RR_ERR_HIP, surfaced here as
`BackendError`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RR_LIB selects a diagnostic build (ablation/stamp variants of the same sources)
LIB_PATH = os.environ.get("RR_LIB") or os.path.join(_HERE, "lib", "libradiorust_amd.so")

RR_OK, RR_ERR_BAD_ARG, RR_ERR_CAPACITY, RR_ERR_HIP, RR_ERR_CONTRACT, RR_ERR_NEED_DESIGN, RR_ERR_NOT_READY = range(7)
RR_F32, RR_F64 = 0, 1
RR_WIN_RECTANGULAR, RR_WIN_KAISER, RR_WIN_SAMPLED = 0, 1, 2

_STATUS_NAMES = {
    1: "RR_ERR_BAD_ARG",
    2: "RR_ERR_CAPACITY",
    3: "RR_ERR_HIP",
    4: "RR_ERR_CONTRACT",
    5: "RR_ERR_NEED_DESIGN",
    6: "RR_ERR_NOT_READY",
}


class BackendError(RuntimeError):
    """Non-zero status from the C ABI."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{_STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class ContractViolation(BackendError, AssertionError):
    """RR_ERR_CONTRACT: the reference would `panic!` here."""


class c64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class Window(C.Structure):
    _fields_ = [("kind", C.c_int), ("beta", C.c_double)]


class ChainParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int),
        ("precision", C.c_double),
        ("shift", C.c_double),
        ("filter_len", C.c_size_t),
        ("output_rate", C.c_double),
        ("bandwidth", C.c_double),
        ("quality", C.c_double),
        ("fft_len", C.c_size_t),
        ("fft_window", Window),
        ("center_dc", C.c_int),
        ("allow_fused", C.c_int),
    ]


class MeterParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int),
        ("precision", C.c_double),
        ("shift", C.c_double),
        ("output_rate", C.c_double),
        ("bandwidth", C.c_double),
        ("quality", C.c_double),
        ("chunk_len", C.c_size_t),
        ("overlap", C.c_size_t),
        ("fft_window", Window),
        ("center_dc", C.c_int),
    ]


_vp, _sz, _d, _i = C.c_void_p, C.c_size_t, C.c_double, C.c_int
_psz = C.POINTER(C.c_size_t)

# name -> (restype, argtypes); every symbol declared in include/radiorust_amd.h
SIGNATURES = {
    "rr_version": (_i, []),
    "rr_last_error_string": (C.c_char_p, []),
    "rr_device_count": (_i, [C.POINTER(_i)]),
    "rr_device_pci_bus_id": (_i, [_i, C.c_char_p, _sz]),
    "rr_set_stream": (_i, [_vp, _vp]),
    "rr_wait": (_i, [_vp]),
    "rr_query": (_i, [_vp]),
    "rr_host_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "rr_host_free": (_i, [_vp]),
    "rr_host_register": (_i, [_vp, _sz]),
    "rr_host_unregister": (_i, [_vp]),
    "rr_bessel_i0": (_d, [_d]),
    "rr_kaiser_rel_with_beta": (_d, [_d, _d]),
    "rr_kaiser_alpha_to_beta": (_d, [_d]),
    "rr_kaiser_null_at_bin_to_beta": (_d, [_d]),
    "rr_sinc": (_d, [_d]),
    "rr_deemphasis_factor": (_i, [_d, _d, _vp]),
    "rr_window_sample": (_i, [C.POINTER(Window), _sz, _vp]),
    "rr_freqshifter_ratio": (_i, [_d, _d, _d, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rr_freqshifter_table": (_i, [_i, C.c_int64, C.c_int64, _d, _vp]),
    "rr_filter_design_taps": (_i, [_sz, _vp, _vp, _vp]),
    "rr_downsampler_design": (_i, [_d, _d, _d, _d, _psz, _vp, _sz]),
    "rr_downsampler_schedule": (_i, [_d, _d, _sz, C.POINTER(_d), _vp, _sz, _psz]),
    "rr_upsampler_schedule": (_i, [_d, _d, _sz, C.POINTER(_d), _vp, _sz, _psz]),
    "rr_fourier_design_window": (_i, [_sz, _vp, _vp]),
    "rr_fourier_route": (_i, [_i, _sz, C.c_char_p, _sz]),
    "rr_freqshifter_create": (_i, [_i, _d, _d, _i, C.POINTER(_vp)]),
    "rr_freqshifter_set_shift": (_i, [_vp, _d]),
    "rr_freqshifter_shift": (_i, [_vp, C.POINTER(_d)]),
    "rr_freqshifter_precision": (_i, [_vp, C.POINTER(_d)]),
    "rr_freqshifter_process": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_freqshifter_enqueue": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_freqshifter_process_dev": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_freqshifter_destroy": (_i, [_vp]),
    "rr_filter_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "rr_filter_needs_design": (_i, [_vp, _d, _sz, C.POINTER(_i)]),
    "rr_filter_mark_params_changed": (_i, [_vp]),
    "rr_filter_design": (_i, [_vp, _d, _sz, _vp, _vp]),
    "rr_filter_reset": (_i, [_vp]),
    "rr_filter_set_gain": (_i, [_vp, _d]),
    "rr_filter_process": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_filter_enqueue": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_filter_process_dev": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_filter_process_dev_f16": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz, _i]),
    "rr_filter_last_kernel": (_i, [_vp, C.POINTER(_i)]),
    "rr_filter_destroy": (_i, [_vp]),
    "rr_downsampler_create": (_i, [_i, _d, _d, _d, _i, C.POINTER(_vp)]),
    "rr_downsampler_set_gain": (_i, [_vp, _d]),
    "rr_downsampler_peek": (_i, [_vp, _d, _sz, _psz]),
    "rr_downsampler_process": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_downsampler_enqueue": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_downsampler_process_dev": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_downsampler_ir_len": (_i, [_vp, _psz]),
    "rr_downsampler_last_kernel": (_i, [_vp, C.POINTER(_i)]),
    "rr_downsampler_destroy": (_i, [_vp]),
    "rr_fourier_create": (_i, [_i, C.POINTER(Window), _i, _i, C.POINTER(_vp)]),
    "rr_fourier_set_sampled_window": (_i, [_vp, _sz, _vp]),
    "rr_fourier_process": (_i, [_vp, _vp, _sz, _vp, _sz, _psz]),
    "rr_fourier_enqueue": (_i, [_vp, _vp, _sz, _vp, _sz, _psz]),
    "rr_fourier_process_dev": (_i, [_vp, _sz, _vp, _sz, _vp, _sz, _psz]),
    "rr_fourier_destroy": (_i, [_vp]),
    "rr_chain_create": (_i, [C.POINTER(ChainParams), _i, C.POINTER(_vp)]),
    "rr_chain_set_shift": (_i, [_vp, _d]),
    "rr_chain_filter_needs_design": (_i, [_vp, _d, C.POINTER(_i)]),
    "rr_chain_filter_mark_params_changed": (_i, [_vp]),
    "rr_chain_filter_design": (_i, [_vp, _d, _vp, _vp]),
    "rr_chain_interrupt": (_i, [_vp]),
    "rr_chain_pending": (_i, [_vp, _psz]),
    "rr_chain_peek": (_i, [_vp, _d, _sz, _psz]),
    "rr_chain_process": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_chain_enqueue": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_chain_process_dev": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_chain_set_metering": (_i, [_vp, _d, _vp, _vp, _sz, _i]),
    "rr_chain_last_path": (_i, [_vp, C.POINTER(_i)]),
    "rr_chain_destroy": (_i, [_vp]),
    "rr_chainbank_create": (_i, [C.POINTER(ChainParams), _sz, _i, C.POINTER(_vp)]),
    "rr_chainbank_channels": (_i, [_vp, _psz]),
    "rr_chainbank_channel": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "rr_chainbank_set_shift": (_i, [_vp, _d]),
    "rr_chainbank_filter_needs_design": (_i, [_vp, _d, C.POINTER(_i)]),
    "rr_chainbank_filter_mark_params_changed": (_i, [_vp]),
    "rr_chainbank_filter_design": (_i, [_vp, _d, _vp, _vp]),
    "rr_chainbank_interrupt": (_i, [_vp]),
    "rr_chainbank_peek": (_i, [_vp, _d, _sz, _psz]),
    "rr_chainbank_process_dev": (_i, [_vp, _d, _vp, _sz, _sz, _vp, _sz, _sz, _psz]),
    "rr_chainbank_last_path": (_i, [_vp, C.POINTER(_i)]),
    "rr_chainbank_destroy": (_i, [_vp]),
    "rr_chain_timing_enable": (_i, [_vp, _i]),
    "rr_chain_timing_every": (_i, [_vp, C.c_uint]),
    "rr_chain_timing_reset": (_i, [_vp]),
    "rr_chain_timing_read": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(C.c_uint64)]),
    "rr_chain_timing_stage_name": (C.c_char_p, [_i]),
    "rr_channelizer_create": (_i, [_i, _sz, _sz, C.POINTER(Window), _i, C.POINTER(_vp)]),
    "rr_channelizer_create_ex": (_i, [_i, _sz, _sz, _sz, C.POINTER(Window), _i, C.POINTER(_vp)]),
    "rr_channelizer_reset": (_i, [_vp]),
    "rr_channelizer_peek": (_i, [_vp, _sz, _psz]),
    "rr_channelizer_process": (_i, [_vp, _vp, _sz, _vp, _sz, _psz]),
    "rr_channelizer_process_dev": (_i, [_vp, _vp, _sz, _vp, _sz, _psz]),
    "rr_channelizer_destroy": (_i, [_vp]),
    "rr_stft_create": (_i, [_i, _sz, _sz, C.POINTER(Window), _i, _i, C.POINTER(_vp)]),
    "rr_stft_reset": (_i, [_vp]),
    "rr_stft_pending": (_i, [_vp, _psz]),
    "rr_meter_create": (_i, [C.POINTER(MeterParams), _i, C.POINTER(_vp)]),
    "rr_meter_set_shift": (_i, [_vp, _d]),
    "rr_meter_filter_design": (_i, [_vp, _vp, _vp]),
    "rr_meter_event": (_i, [_vp, _i]),
    "rr_meter_peek": (_i, [_vp, _d, _sz, _psz]),
    "rr_meter_process": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_meter_process_dev": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_meter_set_metering": (_i, [_vp, _d, _vp, _vp, _sz, _i]),
    "rr_meter_process_bandwidth": (_i, [_vp, _d, _vp, _sz, _d, _vp, _sz, _psz]),
    "rr_meter_last_path": (_i, [_vp, C.POINTER(_i)]),
    "rr_meter_destroy": (_i, [_vp]),
    "rr_stft_peek": (_i, [_vp, _sz, _psz]),
    "rr_stft_process": (_i, [_vp, _vp, _sz, _vp, _sz, _psz]),
    "rr_stft_process_dev": (_i, [_vp, _vp, _sz, _vp, _sz, _psz]),
    "rr_stft_set_metering": (_i, [_vp, _d, _d, _vp, _vp, _sz, _i]),
    "rr_stft_destroy": (_i, [_vp]),
    "rr_upsampler_create": (_i, [_i, _d, _d, _d, _i, C.POINTER(_vp)]),
    "rr_upsampler_peek": (_i, [_vp, _d, _sz, _psz]),
    "rr_upsampler_process": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_upsampler_enqueue": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_upsampler_process_dev": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_upsampler_ir_len": (_i, [_vp, _psz]),
    "rr_upsampler_destroy": (_i, [_vp]),
    "rr_upsampler_design": (_i, [_d, _d, _d, _d, _psz, _vp, _sz]),
    "rr_fmdemod_create": (_i, [_i, _d, _i, C.POINTER(_vp)]),
    "rr_fmdemod_set_deviation": (_i, [_vp, _d]),
    "rr_fmdemod_set_gain": (_i, [_vp, _d]),
    "rr_fmdemod_deviation": (_i, [_vp, C.POINTER(_d)]),
    "rr_fmdemod_reset": (_i, [_vp]),
    "rr_fmdemod_process": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_fmdemod_enqueue": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_fmdemod_process_dev": (_i, [_vp, _d, _vp, _sz, _vp, _sz, _psz]),
    "rr_fmdemod_destroy": (_i, [_vp]),
    "rr_level_dev": (_i, [_i, _i, _vp, _vp, _sz, _sz, _vp]),
    "rr_level": (_i, [_i, _i, _vp, _sz, C.POINTER(_d)]),
    "rr_bandwidth_dev": (_i, [_i, _i, _vp, _d, _d, _vp, _sz, _sz, _vp]),
    "rr_bandwidth": (_i, [_i, _i, _d, _d, _vp, _sz, C.POINTER(_d)]),
    "rr_bandwidth_fast_dev": (_i, [_i, _i, _vp, _d, _d, _vp, _sz, _sz, _vp, _vp]),
    "rr_rescale_energy_dev": (_i, [_i, _i, _vp, _vp, _sz, _sz, _sz, _vp]),
    "rr_rescale_energy": (_i, [_i, _i, _vp, _sz, _sz, _vp]),
    "rr_gain_dev": (_i, [_i, _i, _vp, _d, _vp, _sz, _vp]),
    "rr_gain": (_i, [_i, _i, _d, _vp, _sz, _vp]),
    "rr_synth_iq_dev": (_i, [_i, _vp, C.c_uint64, C.c_uint64, _sz, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Loads the HIP backend.  Raises if it has not been built: there is no
    pure-Python or CPU path behind this package."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m radiorust_amd.build` "
            "(or __graft_entry__.build()); radiorust_amd has no CPU fallback"
        )
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own
    # libamdhip64 and load it by file name, so if our library pulled in the
    # system runtime first, torch.cuda would later fail to initialise.  When
    # torch is installed, let it load its runtime first; ours then binds to the
    # same libamdhip64.so.7 by soname.  (Plumbing only: no torch API is used.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status: int) -> None:
    if status == RR_OK:
        return
    msg = lib().rr_last_error_string().decode("utf-8", "replace")
    if status == RR_ERR_CONTRACT:
        raise ContractViolation(status, msg)
    raise BackendError(status, msg)
