"""ctypes binding of include/spectro_hip.h — the same symbols a Rust `fft_backend::hip_backend` would bind.

The library is REQUIRED: importing this module without a built libspectro_hip.so raises; there is no
CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# SGX_LIB_PATH lets kernel A/B experiments (tools/ablate.sh) load an alternative build of the same ABI
LIB_PATH = os.environ.get("SGX_LIB_PATH") or os.path.join(_PKG, "libspectro_hip.so")

SGX_OK, SGX_INVALID_INPUT, SGX_DIM_MISMATCH, SGX_BACKEND, SGX_INTERNAL = range(5)
WIN_RECTANGULAR, WIN_HANNING, WIN_HAMMING, WIN_BLACKMAN, WIN_KAISER, WIN_GAUSSIAN, WIN_CUSTOM = range(7)
FREQ_LINEAR, FREQ_MEL, FREQ_LOGHZ, FREQ_ERB, FREQ_CHROMA = 0, 1, 2, 3, 4
MELNORM_NONE, MELNORM_SLANEY, MELNORM_L1, MELNORM_L2 = range(4)
AMP_POWER, AMP_MAGNITUDE, AMP_DECIBELS, AMP_COMPLEX = range(4)
F32, F64 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
DEVICE_CURRENT, DEVICE_HOST_ONLY = -1, -2

# every symbol include/spectro_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "sgx_plan_create", "sgx_plan_destroy", "sgx_output_shape", "sgx_execute", "sgx_execute_timed", "sgx_axes",
    "sgx_r2c", "sgx_c2r", "sgx_istft", "sgx_istft_length", "sgx_window", "sgx_mel_weights", "sgx_shard_range", "sgx_last_error", "sgx_last_create_error",
    "sgx_kernel_name", "sgx_abi_version", "sgx_device_count",
    "sgx_fft2d_create", "sgx_fft2d_destroy", "sgx_fft2d_forward", "sgx_fft2d_inverse", "sgx_fft2d_convolve",
    "sgx_fft2d_filter", "sgx_fft2d_last_error", "sgx_fft2d_reserve", "sgx_fft2d_device",
    "sgx_reserve", "sgx_plan_device", "sgx_last_dim_mismatch",
    "sgx_c2c_create", "sgx_c2c_destroy", "sgx_c2c_forward", "sgx_c2c_inverse", "sgx_c2c_last_error",
    "sgx_comm_unique_id", "sgx_comm_create", "sgx_comm_adopt", "sgx_comm_destroy", "sgx_comm_last_error", "sgx_gather", "sgx_shard_execute", "sgx_shard_execute_chunked",
    "sgx_membench", "sgx_clock_probe",
]


class SgxParams(C.Structure):
    _fields_ = [
        ("n_fft", C.c_uint32), ("hop_size", C.c_uint32), ("centre", C.c_int32), ("window_kind", C.c_int32),
        ("window_param", C.c_double), ("custom_window", C.POINTER(C.c_double)), ("custom_window_len", C.c_uint32),
        ("sample_rate_hz", C.c_double), ("freq_scale", C.c_int32), ("n_mels", C.c_uint32), ("f_min", C.c_double),
        ("f_max", C.c_double), ("mel_norm", C.c_int32), ("amp_scale", C.c_int32), ("has_log_params", C.c_int32),
        ("floor_db", C.c_double), ("dtype", C.c_int32), ("device", C.c_int32),
        ("n_mfcc", C.c_uint32), ("mfcc_include_c0", C.c_int32), ("mfcc_lifter", C.c_uint32), ("erb_spacing", C.c_int32),
        ("chroma_tuning", C.c_double), ("chroma_norm", C.c_int32),
    ]


class SpectrogramError(Exception):
    """Base error (src/python/error.rs:10-66)."""


class InvalidInputError(SpectrogramError):
    pass


class DimensionMismatchError(SpectrogramError):
    """DimensionMismatch { expected, got } (src/error.rs:19-21): the two numbers are attributes when the library reported them."""

    def __init__(self, msg="", expected=None, got=None):
        super().__init__(msg)
        self.expected, self.got = expected, got


class FFTBackendError(SpectrogramError):
    pass


class InternalError(SpectrogramError):
    pass


_ERR = {SGX_INVALID_INPUT: InvalidInputError, SGX_DIM_MISMATCH: DimensionMismatchError,
        SGX_BACKEND: FFTBackendError, SGX_INTERNAL: InternalError}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FFTBackendError(
            f"hip -- FFT backend error: {LIB_PATH} is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python -m spectrograms_amd.build`); this package has no CPU fallback")
    try:
        # torch wheels bundle their own HIP/HSA runtime.  Load it first so libspectro_hip.so binds to that same
        # copy (same SONAME) instead of pulling /opt/rocm's: two HIP runtimes in one process cannot both see the GPU.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.sgx_plan_create.argtypes = [C.POINTER(SgxParams), C.POINTER(vp)]
    L.sgx_plan_destroy.argtypes = [vp]
    L.sgx_plan_destroy.restype = None
    L.sgx_output_shape.argtypes = [vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.sgx_execute.argtypes = [vp, vp, sz, sz, sz, vp, sz, C.c_int32, vp]
    L.sgx_execute_timed.argtypes = [vp, vp, sz, sz, sz, vp, sz, vp, C.c_int32, C.POINTER(C.c_float)]
    L.sgx_axes.argtypes = [vp, sz, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.sgx_r2c.argtypes = [vp, vp, sz, vp, sz]
    L.sgx_c2r.argtypes = [vp, vp, sz, vp, sz]
    L.sgx_istft_length.argtypes = [vp, sz, C.POINTER(sz)]
    L.sgx_istft.argtypes = [vp, vp, sz, sz, sz, vp, sz, C.c_int32, vp]
    L.sgx_window.argtypes = [vp, C.POINTER(C.c_double)]
    L.sgx_mel_weights.argtypes = [vp, C.POINTER(sz), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    L.sgx_shard_range.argtypes = [sz, C.c_int32, C.c_int32, C.POINTER(sz), C.POINTER(sz)]
    L.sgx_last_error.argtypes = [vp]
    L.sgx_last_error.restype = C.c_char_p
    L.sgx_last_create_error.restype = C.c_char_p
    L.sgx_kernel_name.argtypes = [vp]
    L.sgx_kernel_name.restype = C.c_char_p
    L.sgx_abi_version.restype = C.c_int32
    L.sgx_device_count.restype = C.c_int32
    L.sgx_fft2d_create.argtypes = [sz, sz, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.sgx_fft2d_destroy.argtypes = [vp]
    L.sgx_fft2d_destroy.restype = None
    L.sgx_fft2d_forward.argtypes = [vp, vp, sz, vp, C.c_int32, vp]
    L.sgx_fft2d_inverse.argtypes = [vp, vp, sz, vp, C.c_int32, vp]
    L.sgx_fft2d_convolve.argtypes = [vp, vp, sz, vp, sz, sz, vp, C.c_int32, vp]
    L.sgx_fft2d_filter.argtypes = [vp, vp, sz, C.c_int32, C.c_double, C.c_double, vp, C.c_int32, vp]
    L.sgx_fft2d_last_error.argtypes = [vp]
    L.sgx_fft2d_last_error.restype = C.c_char_p
    L.sgx_fft2d_reserve.argtypes = [vp, sz, C.c_int32]
    L.sgx_fft2d_device.argtypes = [vp]
    L.sgx_fft2d_device.restype = C.c_int32
    L.sgx_reserve.argtypes = [vp, sz, sz, C.c_int32, C.c_int32]
    L.sgx_plan_device.argtypes = [vp]
    L.sgx_plan_device.restype = C.c_int32
    L.sgx_last_dim_mismatch.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    L.sgx_c2c_create.argtypes = [sz, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.sgx_c2c_destroy.argtypes = [vp]
    L.sgx_c2c_destroy.restype = None
    L.sgx_c2c_forward.argtypes = [vp, vp, sz]
    L.sgx_c2c_inverse.argtypes = [vp, vp, sz]
    L.sgx_c2c_last_error.argtypes = [vp]
    L.sgx_c2c_last_error.restype = C.c_char_p
    L.sgx_comm_unique_id.argtypes = [vp]
    L.sgx_comm_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.sgx_comm_adopt.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.sgx_comm_destroy.argtypes = [vp]
    L.sgx_comm_destroy.restype = None
    L.sgx_comm_last_error.argtypes = [vp]
    L.sgx_comm_last_error.restype = C.c_char_p
    L.sgx_gather.argtypes = [vp, vp, vp, sz, sz, C.c_int32, vp]
    L.sgx_shard_execute.argtypes = [vp, vp, vp, sz, sz, sz, vp, vp, vp]
    L.sgx_shard_execute_chunked.argtypes = [vp, vp, vp, sz, sz, sz, vp, vp, C.c_int32, vp]
    L.sgx_membench.argtypes = [C.c_int32, sz, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    L.sgx_clock_probe.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
    L.sgx_clock_probe.restype = C.c_int32
    _lib = L
    return L


def raise_status(status: int, plan=None) -> None:
    if status == SGX_OK:
        return
    L = lib()
    msg = (L.sgx_last_error(plan) if plan else L.sgx_last_create_error()) or b""
    text = msg.decode() or f"status {status}"
    if status == SGX_DIM_MISMATCH and plan:
        e, g = C.c_size_t(), C.c_size_t()
        if L.sgx_last_dim_mismatch(plan, C.byref(e), C.byref(g)) == SGX_OK:
            raise DimensionMismatchError(text, e.value, g.value)
    raise _ERR.get(status, InternalError)(text)
