"""2-D FFT path (BASELINE config 5): the reference's names and argument order (src/python/fft2d.rs; src/fft2d.rs,
src/image_ops.rs) over the sgx_fft2d_* C ABI, plus batched entry points ([B, R, C] in one call)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _ffi
from .params import parse_dtype


class Fft2dPlan:
    """One sgx_fft2d (shape + dtype).  Not thread-safe, like the reference's `&mut self` plans."""

    def __init__(self, nrows: int, ncols: int, dtype: Optional[str] = None, device: int = _ffi.DEVICE_CURRENT):
        self._lib = _ffi.lib()
        self.nrows, self.ncols = int(nrows), int(ncols)
        self._dt = parse_dtype(dtype)
        self._np = np.float32 if self._dt == _ffi.F32 else np.float64
        self._cnp = np.complex64 if self._dt == _ffi.F32 else np.complex128
        h = C.c_void_p()
        st = self._lib.sgx_fft2d_create(self.nrows, self.ncols, self._dt, device, C.byref(h))
        if st:
            raise _ffi._ERR.get(st, _ffi.InternalError)((self._lib.sgx_fft2d_last_error(None) or b"").decode())
        self._h = h
        self._device = int(self._lib.sgx_fft2d_device(h))

    @property
    def device(self) -> int:
        return self._device

    def reserve(self, batch: int, host_staging: bool = True) -> None:
        """Pre-size the plan-owned intermediates so that later calls of up to `batch` images do not allocate."""
        self._check(self._lib.sgx_fft2d_reserve(self._h, int(batch), int(host_staging)))

    def _device_images(self, t, what: str, shape_tail, tdt):
        """A wrong shape, dtype, layout or device would make the kernels read or write out of bounds: check before the call."""
        if not t.is_cuda or t.device.index != self._device:
            raise ValueError(f"{what} is on {t.device}, the plan is bound to cuda:{self._device}")
        if t.dtype != tdt or not t.is_contiguous():
            raise ValueError(f"{what} must be a contiguous tensor of the plan's dtype")
        if t.dim() != len(shape_tail) + 1 or tuple(t.shape[1:]) != tuple(shape_tail) or t.shape[0] == 0:
            raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected (batch, {', '.join(map(str, shape_tail))}), got {tuple(t.shape)}",
                                              tuple(shape_tail), tuple(t.shape[1:]))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.sgx_fft2d_destroy(h)
            self._h = None

    def _check(self, st):
        if st:
            raise _ffi._ERR.get(st, _ffi.InternalError)((self._lib.sgx_fft2d_last_error(self._h) or b"").decode())

    def _images(self, a) -> np.ndarray:
        x = np.ascontiguousarray(a, dtype=self._np)
        if x.ndim == 2:
            x = x[None]
        if x.ndim != 3 or x.shape[1:] != (self.nrows, self.ncols):
            raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {(self.nrows, self.ncols)}, got {x.shape[1:]}")
        return x

    def forward(self, images) -> np.ndarray:
        x = self._images(images)
        out = np.empty((x.shape[0], self.nrows, self.ncols // 2 + 1), self._cnp)
        self._check(self._lib.sgx_fft2d_forward(self._h, x.ctypes.data, x.shape[0], out.ctypes.data, _ffi.MEM_HOST, None))
        return out

    def inverse(self, spectra) -> np.ndarray:
        s = np.ascontiguousarray(spectra, dtype=self._cnp)
        if s.ndim == 2:
            s = s[None]
        if s.ndim != 3 or s.shape[1] != self.nrows:
            raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {self.nrows} rows, got {s.shape[1:]}")
        if s.shape[2] != self.ncols // 2 + 1:  # ifft2d: src/fft2d.rs:157-163
            raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {self.ncols // 2 + 1}, got {s.shape[2]}")
        out = np.empty((s.shape[0], self.nrows, self.ncols), self._np)
        self._check(self._lib.sgx_fft2d_inverse(self._h, s.ctypes.data, s.shape[0], out.ctypes.data, _ffi.MEM_HOST, None))
        return out

    def convolve(self, images, kernel) -> np.ndarray:
        x = self._images(images)
        k = np.ascontiguousarray(kernel, dtype=self._np)
        if k.ndim != 2:
            raise ValueError("kernel must be 2-D")
        out = np.empty_like(x)
        self._check(self._lib.sgx_fft2d_convolve(self._h, x.ctypes.data, x.shape[0], k.ctypes.data, k.shape[0], k.shape[1],
                                                 out.ctypes.data, _ffi.MEM_HOST, None))
        return out

    def filter(self, images, kind: int, lo: float, hi: float = 0.0) -> np.ndarray:
        x = self._images(images)
        out = np.empty_like(x)
        self._check(self._lib.sgx_fft2d_filter(self._h, x.ctypes.data, x.shape[0], kind, float(lo), float(hi), out.ctypes.data,
                                               _ffi.MEM_HOST, None))
        return out

    # device-resident batched entry points (torch CUDA tensors), used by the benchmarks
    def forward_torch(self, x, out=None):
        import torch
        cdt = torch.float32 if self._dt == _ffi.F32 else torch.float64
        self._device_images(x, "images", (self.nrows, self.ncols), cdt)
        b = x.shape[0]
        if out is None:
            out = torch.empty((b, self.nrows, self.ncols // 2 + 1, 2), dtype=cdt, device=x.device)
        else:
            self._device_images(out, "out", (self.nrows, self.ncols // 2 + 1, 2), cdt)
            if out.shape[0] != b:
                raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {b}, got {out.shape[0]}", b, out.shape[0])
        s = torch.cuda.current_stream(x.device).cuda_stream
        self._check(self._lib.sgx_fft2d_forward(self._h, x.data_ptr(), b, out.data_ptr(), _ffi.MEM_DEVICE, C.c_void_p(s)))
        return out

    def inverse_torch(self, spec, out=None):
        """ifft2d of [B, nrows, ncols/2+1, 2] half spectra (interleaved re, im) -> [B, nrows, ncols], device resident."""
        import torch
        tdt = torch.float32 if self._dt == _ffi.F32 else torch.float64
        self._device_images(spec, "spectra", (self.nrows, self.ncols // 2 + 1, 2), tdt)
        b = spec.shape[0]
        if out is None:
            out = torch.empty((b, self.nrows, self.ncols), dtype=tdt, device=spec.device)
        else:
            self._device_images(out, "out", (self.nrows, self.ncols), tdt)
            if out.shape[0] != b:
                raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {b}, got {out.shape[0]}", b, out.shape[0])
        s = torch.cuda.current_stream(spec.device).cuda_stream
        self._check(self._lib.sgx_fft2d_inverse(self._h, spec.data_ptr(), b, out.data_ptr(), _ffi.MEM_DEVICE, C.c_void_p(s)))
        return out

    def convolve_torch(self, x, kernel, out=None):
        import torch
        k = np.ascontiguousarray(kernel, dtype=self._np)
        if k.ndim != 2:
            raise ValueError("kernel must be 2-D")
        tdt = torch.float32 if self._dt == _ffi.F32 else torch.float64
        self._device_images(x, "images", (self.nrows, self.ncols), tdt)
        if out is None:
            out = torch.empty_like(x)
        else:
            self._device_images(out, "out", (self.nrows, self.ncols), tdt)
            if out.shape[0] != x.shape[0]:
                raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {x.shape[0]}, got {out.shape[0]}", x.shape[0], out.shape[0])
        s = torch.cuda.current_stream(x.device).cuda_stream
        self._check(self._lib.sgx_fft2d_convolve(self._h, x.data_ptr(), x.shape[0], k.ctypes.data, k.shape[0], k.shape[1],
                                                 out.data_ptr(), _ffi.MEM_DEVICE, C.c_void_p(s)))
        return out

    def filter_torch(self, x, kind: int, lo: float, hi: float = 0.0, out=None):
        """lowpass (0) / highpass (1) / bandpass (2) of [B, nrows, ncols] device images, device resident (src/image_ops.rs:301-430)."""
        import torch
        tdt = torch.float32 if self._dt == _ffi.F32 else torch.float64
        self._device_images(x, "images", (self.nrows, self.ncols), tdt)
        if out is None:
            out = torch.empty_like(x)
        else:
            self._device_images(out, "out", (self.nrows, self.ncols), tdt)
            if out.shape[0] != x.shape[0]:
                raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {x.shape[0]}, got {out.shape[0]}", x.shape[0], out.shape[0])
        s = torch.cuda.current_stream(x.device).cuda_stream
        self._check(self._lib.sgx_fft2d_filter(self._h, x.data_ptr(), x.shape[0], kind, float(lo), float(hi), out.data_ptr(),
                                               _ffi.MEM_DEVICE, C.c_void_p(s)))
        return out


class C2cPlan:
    """1-D complex-to-complex plan: C2cPlan<T> of the reference (src/fft_backend.rs:113-137).  `forward` / `inverse` return a
    new array; neither normalises (the caller divides by n after an inverse)."""

    def __init__(self, n: int, dtype: Optional[str] = None, device: int = _ffi.DEVICE_CURRENT):
        self._lib = _ffi.lib()
        self.n = int(n)
        self._dt = parse_dtype(dtype)
        self._cnp = np.complex64 if self._dt == _ffi.F32 else np.complex128
        h = C.c_void_p()
        st = self._lib.sgx_c2c_create(self.n, self._dt, device, C.byref(h))
        if st:
            raise _ffi._ERR.get(st, _ffi.InternalError)((self._lib.sgx_c2c_last_error(None) or b"").decode())
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.sgx_c2c_destroy(h)
            self._h = None

    def _run(self, fn, x) -> np.ndarray:
        buf = np.array(x, dtype=self._cnp, copy=True).ravel()
        st = fn(self._h, buf.ctypes.data, buf.size)
        if st:
            msg = (self._lib.sgx_c2c_last_error(self._h) or b"").decode()
            if st == _ffi.SGX_DIM_MISMATCH:
                raise _ffi.DimensionMismatchError(msg, self.n, buf.size)
            raise _ffi._ERR.get(st, _ffi.InternalError)(msg)
        return buf

    def forward(self, x) -> np.ndarray:
        return self._run(self._lib.sgx_c2c_forward, x)

    def inverse(self, x) -> np.ndarray:
        return self._run(self._lib.sgx_c2c_inverse, x)


class Fft2dPlanner:
    """Plan cache keyed by shape (src/fft2d.rs:491-657)."""

    def __init__(self, dtype=None):
        self._plans = {}
        self._dtype = "float32" if parse_dtype(dtype) == _ffi.F32 else "float64"  # fixed for the planner's lifetime

    @property
    def dtype(self) -> str:
        return self._dtype

    def _plan(self, shape, dtype) -> Fft2dPlan:
        dtype = self._dtype if dtype is None else dtype
        key = (int(shape[0]), int(shape[1]), parse_dtype(dtype))
        if key not in self._plans:
            self._plans[key] = Fft2dPlan(key[0], key[1], dtype)
        return self._plans[key]

    def fft2d(self, data, dtype=None):
        a = _as2d(data)
        return self._plan(a.shape, dtype).forward(a)[0]

    def ifft2d(self, spectrum, output_ncols: int, dtype=None):
        s = np.asarray(spectrum)
        return self._plan((s.shape[0], output_ncols), dtype).inverse(s)[0]

    def power_spectrum_2d(self, data):
        s = self.fft2d(data)
        return (s.real * s.real + s.imag * s.imag).astype(s.real.dtype)

    def magnitude_spectrum_2d(self, data):
        return np.abs(self.fft2d(data))


def _as2d(a) -> np.ndarray:
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("expected a 2-D array")
    if a.shape[0] == 0 or a.shape[1] == 0:
        raise _ffi.InvalidInputError("Invalid input: array dimensions must be > 0")
    return a


def fft2d(data, dtype=None):
    a = _as2d(data)
    return Fft2dPlan(a.shape[0], a.shape[1], dtype).forward(a)[0]


def ifft2d(spectrum, output_ncols: int, dtype=None):
    s = np.asarray(spectrum)
    if s.ndim != 2 or s.shape[0] == 0 or output_ncols == 0:
        raise _ffi.InvalidInputError("Invalid input: dimensions must be > 0")
    return Fft2dPlan(s.shape[0], output_ncols, dtype).inverse(s)[0]


def power_spectrum_2d(data, dtype=None):
    s = fft2d(data, dtype)
    return (s.real * s.real + s.imag * s.imag).astype(s.real.dtype)  # norm_sqr (src/fft2d.rs:214-218)


def magnitude_spectrum_2d(data, dtype=None):
    return np.abs(fft2d(data, dtype))


def gaussian_kernel_2d(size: int, sigma: float, dtype=None) -> np.ndarray:
    """src/image_ops.rs:188-220 — geometry in f64, normalised, cast to T."""
    if size <= 0 or size % 2 == 0:
        raise _ffi.InvalidInputError("Invalid input: kernel size must be odd and > 0")
    if sigma <= 0.0:
        raise _ffi.InvalidInputError("Invalid input: sigma must be > 0")
    center = float(size // 2)
    variance = sigma * sigma
    coeff = 1.0 / (2.0 * math.pi * variance)
    idx = np.arange(size, dtype=np.float64) - center
    k = coeff * np.exp(-(idx[:, None] ** 2 + idx[None, :] ** 2) / (2.0 * variance))
    total = 0.0
    for v in k.reshape(-1):  # sequential f64 sum, as ndarray's .sum() over the standard layout
        total += float(v)
    return (k / total).astype(np.float32 if parse_dtype(dtype) == _ffi.F32 else np.float64)


def convolve_fft(image, kernel, dtype=None):
    a = _as2d(image)
    return Fft2dPlan(a.shape[0], a.shape[1], dtype).convolve(a, kernel)[0]


def lowpass_filter(image, cutoff_fraction: float, dtype=None):
    a = _as2d(image)
    return Fft2dPlan(a.shape[0], a.shape[1], dtype).filter(a, 0, cutoff_fraction)[0]


def highpass_filter(image, cutoff_fraction: float, dtype=None):
    a = _as2d(image)
    return Fft2dPlan(a.shape[0], a.shape[1], dtype).filter(a, 1, cutoff_fraction)[0]


def bandpass_filter(image, low_cutoff: float, high_cutoff: float, dtype=None):
    a = _as2d(image)
    return Fft2dPlan(a.shape[0], a.shape[1], dtype).filter(a, 2, low_cutoff, high_cutoff)[0]


def detect_edges_fft(image, dtype=None):
    return highpass_filter(image, 0.1, dtype)  # src/image_ops.rs:467-470


def sharpen_fft(image, amount: float, dtype=None):
    if amount < 0.0:
        raise _ffi.InvalidInputError("Invalid input: amount must be >= 0")
    hp = highpass_filter(image, 0.2, dtype)
    return (np.asarray(image, dtype=hp.dtype) + hp * hp.dtype.type(amount)).astype(hp.dtype)  # :486-514


def fftshift(arr, dtype=None):
    return np.fft.fftshift(np.asarray(arr, dtype=np.float32 if parse_dtype(dtype) == _ffi.F32 else np.float64))


def ifftshift(arr, dtype=None):
    return np.fft.ifftshift(np.asarray(arr, dtype=np.float32 if parse_dtype(dtype) == _ffi.F32 else np.float64))


def fftshift_1d(arr, dtype=None):
    """1-D fftshift (src/fft2d.rs fftshift_1d): DC to the centre."""
    a = np.asarray(arr, dtype=np.float32 if parse_dtype(dtype) == _ffi.F32 else np.float64)
    if a.ndim != 1:
        raise ValueError("fftshift_1d expects a 1-D array")
    return np.fft.fftshift(a)


def ifftshift_1d(arr, dtype=None):
    a = np.asarray(arr, dtype=np.float32 if parse_dtype(dtype) == _ffi.F32 else np.float64)
    if a.ndim != 1:
        raise ValueError("ifftshift_1d expects a 1-D array")
    return np.fft.ifftshift(a)


def fftfreq(n: int, d: float = 1.0, dtype=None):
    return np.fft.fftfreq(n, d).astype(np.float32 if parse_dtype(dtype) == _ffi.F32 else np.float64)


def rfftfreq(n: int, d: float = 1.0, dtype=None):
    return np.fft.rfftfreq(n, d).astype(np.float32 if parse_dtype(dtype) == _ffi.F32 else np.float64)
