"""ctypes front-end for the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY — see oracle/spectro_oracle.h.  Import this module only
from tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``spectrograms_amd/`` may import it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

WINDOWS = {"rectangular": 0, "hanning": 1, "hamming": 2, "blackman": 3, "kaiser": 4,
           "gaussian": 5, "custom": 6}
MEL_NORMS = {None: 0, "none": 0, "slaney": 1, "l1": 2, "l2": 3}
AMPS = {"power": 0, "magnitude": 1, "db": 2, "decibels": 2}


class _Params(C.Structure):
    _fields_ = [
        ("n_fft", C.c_uint32), ("hop", C.c_uint32), ("centre", C.c_int32),
        ("window_kind", C.c_int32), ("window_param", C.c_double),
        ("custom_window", C.POINTER(C.c_double)), ("sample_rate", C.c_double),
        ("freq_scale", C.c_int32), ("n_mels", C.c_uint32), ("f_min", C.c_double),
        ("f_max", C.c_double), ("mel_norm", C.c_int32), ("amp_scale", C.c_int32),
        ("has_db", C.c_int32), ("floor_db", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        sz, dp, fp = C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.orc_validate.argtypes = [C.POINTER(_Params), C.c_char_p, sz]
        L.orc_frame_count.restype = sz
        L.orc_frame_count.argtypes = [sz, sz, sz, C.c_int]
        L.orc_istft_length.restype = sz
        L.orc_istft_length.argtypes = [sz, sz, sz, C.c_int]
        L.orc_make_window.argtypes = [C.c_int, C.c_double, dp, sz, dp]
        L.orc_mel_filterbank.restype = C.c_long
        L.orc_mel_filterbank.argtypes = [C.c_double, sz, sz, C.c_double, C.c_double, C.c_int,
                                         C.POINTER(sz), C.POINTER(C.c_uint32), dp, sz]
        L.orc_axes.argtypes = [C.POINTER(_Params), sz, dp, dp]
        L.orc_hz_to_mel.restype = C.c_double
        L.orc_hz_to_mel.argtypes = [C.c_double]
        L.orc_mel_to_hz.restype = C.c_double
        L.orc_mel_to_hz.argtypes = [C.c_double]
        for suf, p in (("f32", fp), ("f64", dp)):
            getattr(L, f"orc_rfft_{suf}").argtypes = [p, sz, p]
            getattr(L, f"orc_stft_{suf}").argtypes = [C.POINTER(_Params), p, sz, p]
            getattr(L, f"orc_spectrogram_{suf}").argtypes = [C.POINTER(_Params), p, sz, p]
            getattr(L, f"orc_spectrogram_batch_{suf}").argtypes = [C.POINTER(_Params), p, sz, sz, sz, p, C.c_int]
            getattr(L, f"orc_stft_batch_{suf}").argtypes = [C.POINTER(_Params), p, sz, sz, sz, p, C.c_int]
            getattr(L, f"orc_mfcc_{suf}").argtypes = [C.POINTER(_Params), C.c_uint32, C.c_int, C.c_uint32, p, sz, p]
            getattr(L, f"orc_chromagram_{suf}").argtypes = [C.POINTER(_Params), C.c_double, C.c_double, C.c_double, C.c_int, p, sz, p]
            getattr(L, f"orc_irfft_{suf}").argtypes = [p, sz, sz, p]
            getattr(L, f"orc_istft_{suf}").argtypes = [p, sz, sz, sz, sz, C.c_int, C.c_double, dp, C.c_int, p]
            getattr(L, f"orc_fft2d_{suf}").argtypes = [p, sz, sz, p]
            getattr(L, f"orc_ifft2d_{suf}").argtypes = [p, sz, sz, p]
            getattr(L, f"orc_convolve_fft_{suf}").argtypes = [p, sz, sz, p, sz, sz, p]
            getattr(L, f"orc_filter2d_{suf}").argtypes = [p, sz, sz, C.c_int, C.c_double, C.c_double, p]
        L.orc_gaussian_kernel_2d.argtypes = [sz, C.c_double, dp]
        L.orc_lowpass_mask.argtypes = [sz, sz, C.c_double, dp]
        L.orc_lowpass_mask.restype = None
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code: int, msg: str = ""):
        super().__init__(f"oracle status {code}: {msg}")
        self.code = code


@dataclass
class Params:
    n_fft: int
    hop: int
    window: str = "hanning"
    window_param: float = 0.0
    custom_window: Optional[np.ndarray] = None
    centre: bool = True
    sample_rate: float = 16000.0
    n_mels: int = 0            # 0 -> linear
    loghz: bool = False        # True: n_mels log-spaced bins between f_min and f_max (LogHzParams)
    erb: bool = False          # True: n_mels frequency-domain gammatone filters (ErbParams)
    erb_spacing: int = 0       # 0 linear on the ERB scale, 1 Apple TR #35
    f_min: float = 0.0
    f_max: float = 8000.0
    mel_norm: Optional[str] = None
    amp: str = "power"
    floor_db: Optional[float] = None
    _keep: list = field(default_factory=list, repr=False)

    def c(self) -> _Params:
        p = _Params()
        p.n_fft, p.hop, p.centre = self.n_fft, self.hop, int(self.centre)
        p.window_kind = WINDOWS[self.window]
        p.window_param = float(self.window_param)
        if self.custom_window is not None:
            cw = np.ascontiguousarray(self.custom_window, dtype=np.float64)
            self._keep.append(cw)
            p.custom_window = cw.ctypes.data_as(C.POINTER(C.c_double))
        p.sample_rate = float(self.sample_rate)
        p.freq_scale = (3 if self.erb else 2 if self.loghz else 1) if self.n_mels else 0
        p.n_mels = int(self.n_mels)
        p.f_min, p.f_max = float(self.f_min), float(self.f_max)
        p.mel_norm = int(self.erb_spacing) if self.erb else MEL_NORMS[self.mel_norm]
        p.amp_scale = AMPS[self.amp]
        p.has_db = int(self.floor_db is not None)
        p.floor_db = float(self.floor_db) if self.floor_db is not None else 0.0
        return p

    @property
    def n_bins(self) -> int:
        return self.n_mels if self.n_mels else self.n_fft // 2 + 1


def _suf(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(dtype)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float if a.dtype == np.float32 else C.c_double))


def validate(p: Params) -> None:
    buf = C.create_string_buffer(256)
    cp = p.c()
    rc = lib().orc_validate(C.byref(cp), buf, 256)
    if rc:
        raise OracleError(rc, buf.value.decode())


def frame_count(n_samples: int, n_fft: int, hop: int, centre: bool) -> int:
    return int(lib().orc_frame_count(n_samples, n_fft, hop, int(centre)))


def make_window(kind: str, n: int, param: float = 0.0, custom=None) -> np.ndarray:
    out = np.empty(n, np.float64)
    cw = None if custom is None else np.ascontiguousarray(custom, np.float64)
    rc = lib().orc_make_window(WINDOWS[kind], float(param),
                               None if cw is None else _ptr(cw), n, _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def mel_filterbank(sr, n_fft, n_mels, f_min, f_max, norm=None):
    """Returns (row_ptr, cols, vals) CSR and a dense (n_mels, n_fft//2+1) f64 matrix."""
    nb = n_fft // 2 + 1
    cap = n_mels * nb
    row_ptr = np.zeros(n_mels + 1, np.uintp)
    cols = np.zeros(cap, np.uint32)
    vals = np.zeros(cap, np.float64)
    nnz = lib().orc_mel_filterbank(float(sr), n_fft, n_mels, float(f_min), float(f_max),
                                   MEL_NORMS[norm], row_ptr.ctypes.data_as(C.POINTER(C.c_size_t)),
                                   cols.ctypes.data_as(C.POINTER(C.c_uint32)), _ptr(vals), cap)
    if nnz < 0:
        raise OracleError(-nnz)
    dense = np.zeros((n_mels, nb), np.float64)
    for m in range(n_mels):
        a, b = int(row_ptr[m]), int(row_ptr[m + 1])
        dense[m, cols[a:b]] = vals[a:b]
    return row_ptr, cols[:nnz].copy(), vals[:nnz].copy(), dense


def axes(p: Params, n_frames: int):
    freqs = np.empty(p.n_bins, np.float64)
    times = np.empty(n_frames, np.float64)
    cp = p.c()
    rc = lib().orc_axes(C.byref(cp), n_frames, _ptr(freqs), _ptr(times))
    if rc:
        raise OracleError(rc)
    return freqs, times


def rfft(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x)
    suf = _suf(x.dtype)
    out = np.empty((x.size // 2 + 1, 2), x.dtype)
    rc = getattr(lib(), f"orc_rfft_{suf}")(_ptr(x), x.size, _ptr(out))
    if rc:
        raise OracleError(rc)
    return out[:, 0] + 1j * out[:, 1]


def stft(p: Params, x: np.ndarray) -> np.ndarray:
    """Complex STFT (n_fft/2+1, n_frames) in x.dtype precision."""
    x = np.ascontiguousarray(x)
    suf = _suf(x.dtype)
    nf = frame_count(x.size, p.n_fft, p.hop, p.centre)
    out = np.empty((p.n_fft // 2 + 1, nf, 2), x.dtype)
    cp = p.c()
    rc = getattr(lib(), f"orc_stft_{suf}")(C.byref(cp), _ptr(x), x.size, _ptr(out))
    if rc:
        raise OracleError(rc)
    return out.view(np.complex64 if suf == "f32" else np.complex128)[..., 0]


def spectrogram(p: Params, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x)
    suf = _suf(x.dtype)
    nf = frame_count(x.size, p.n_fft, p.hop, p.centre)
    out = np.empty((p.n_bins, nf), x.dtype)
    cp = p.c()
    rc = getattr(lib(), f"orc_spectrogram_{suf}")(C.byref(cp), _ptr(x), x.size, _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def spectrogram_batch(p: Params, x: np.ndarray, nthreads: int = 1, out: Optional[np.ndarray] = None) -> np.ndarray:
    x = np.ascontiguousarray(x)
    assert x.ndim == 2
    suf = _suf(x.dtype)
    b, n = x.shape
    nf = frame_count(n, p.n_fft, p.hop, p.centre)
    if out is None:
        out = np.empty((b, p.n_bins, nf), x.dtype)
    assert out.shape == (b, p.n_bins, nf) and out.dtype == x.dtype and out.flags.c_contiguous
    cp = p.c()
    rc = getattr(lib(), f"orc_spectrogram_batch_{suf}")(C.byref(cp), _ptr(x), b, n, n, _ptr(out), nthreads)
    if rc:
        raise OracleError(rc)
    return out


def stft_batch(p: Params, x: np.ndarray, nthreads: int = 1) -> np.ndarray:
    x = np.ascontiguousarray(x)
    assert x.ndim == 2
    suf = _suf(x.dtype)
    b, n = x.shape
    nf = frame_count(n, p.n_fft, p.hop, p.centre)
    out = np.empty((b, p.n_fft // 2 + 1, nf, 2), x.dtype)
    cp = p.c()
    rc = getattr(lib(), f"orc_stft_batch_{suf}")(C.byref(cp), _ptr(x), b, n, n, _ptr(out), nthreads)
    if rc:
        raise OracleError(rc)
    return out.view(np.complex64 if suf == "f32" else np.complex128)[..., 0]


def mfcc(p: Params, x: np.ndarray, n_mfcc: int = 13, include_c0: bool = True, lifter: int = 22) -> np.ndarray:
    """src/mfcc.rs:224-316 over the Mel-dB spectrogram described by p (which must be Mel + dB)."""
    x = np.ascontiguousarray(x)
    suf = _suf(x.dtype)
    nf = frame_count(x.size, p.n_fft, p.hop, p.centre)
    rows = n_mfcc - (0 if include_c0 or n_mfcc <= 1 else 1)
    out = np.empty((rows, nf), x.dtype)
    cp = p.c()
    rc = getattr(lib(), f"orc_mfcc_{suf}")(C.byref(cp), n_mfcc, int(include_c0), lifter, _ptr(x), x.size, _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


CHROMA_NORMS = {None: 0, "none": 0, "l1": 1, "l2": 2, "max": 3}


def chroma_filterbank(sr: float, n_fft: int, tuning: float = 440.0, f_min: float = 32.7, f_max: float = 4186.0) -> np.ndarray:
    fb = np.empty((12, n_fft // 2 + 1), np.float64)
    L = lib()
    L.orc_chroma_filterbank.argtypes = [C.c_double, C.c_size_t, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
    rc = L.orc_chroma_filterbank(sr, n_fft, tuning, f_min, f_max, fb.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise OracleError(rc, "chroma_filterbank")
    return fb


def chromagram(p: Params, x: np.ndarray, tuning: float = 440.0, f_min: float = 32.7, f_max: float = 4186.0, norm="l2") -> np.ndarray:
    """chromagram() (src/chroma.rs:470-505) of one signal in x's precision; p carries the STFT parameters."""
    x = np.ascontiguousarray(x)
    rdt = np.float32 if x.dtype == np.float32 else np.float64
    x = x.astype(rdt)
    out = np.empty((12, frame_count(x.size, p.n_fft, p.hop, p.centre)), rdt)
    rc = getattr(lib(), f"orc_chromagram_{_suf(rdt)}")(C.byref(p.c()), tuning, f_min, f_max, CHROMA_NORMS[norm], _ptr(x), x.size, _ptr(out))
    if rc != 0:
        raise OracleError(rc, "chromagram")
    return out


def irfft(spec: np.ndarray, n_fft: int) -> np.ndarray:
    """irfft (spectrogram.rs:4789-4811) in spec's precision (complex64 -> f32, complex128 -> f64)."""
    spec = np.ascontiguousarray(spec)
    rdt = np.float32 if spec.dtype == np.complex64 else np.float64
    spec = spec.astype(np.complex64 if rdt == np.float32 else np.complex128)
    out = np.empty(n_fft, rdt)
    rc = getattr(lib(), f"orc_irfft_{_suf(rdt)}")(_ptr(spec.view(rdt)), spec.shape[0], n_fft, _ptr(out))
    if rc != 0:
        raise OracleError(rc, "irfft")
    return out


def istft_length(n_frames: int, n_fft: int, hop: int, centre: bool) -> int:
    return int(lib().orc_istft_length(n_frames, n_fft, hop, int(centre)))


def istft(stft_matrix: np.ndarray, n_fft: int, hop: int, window: str = "hanning", centre: bool = True,
          window_param: float = 0.0, custom=None) -> np.ndarray:
    """istft (spectrogram.rs:4860-4946) of one (n_bins, n_frames) complex matrix, in its precision."""
    m = np.ascontiguousarray(stft_matrix)
    rdt = np.float32 if m.dtype == np.complex64 else np.float64
    m = m.astype(np.complex64 if rdt == np.float32 else np.complex128)
    nb, nf = m.shape
    out = np.empty(istft_length(nf, n_fft, hop, centre), rdt)
    cw = None
    if custom is not None:
        cw = np.ascontiguousarray(custom, dtype=np.float64)
    rc = getattr(lib(), f"orc_istft_{_suf(rdt)}")(_ptr(m.view(rdt)), nb, nf, n_fft, hop, WINDOWS[window], float(window_param),
                                                   cw.ctypes.data_as(C.POINTER(C.c_double)) if cw is not None else None,
                                                   int(centre), _ptr(out))
    if rc != 0:
        raise OracleError(rc, "istft")
    return out


def fft2d(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img)
    suf = _suf(img.dtype)
    r, c = img.shape
    out = np.empty((r, c // 2 + 1, 2), img.dtype)
    rc = getattr(lib(), f"orc_fft2d_{suf}")(_ptr(img), r, c, _ptr(out))
    if rc:
        raise OracleError(rc)
    return out.view(np.complex64 if suf == "f32" else np.complex128)[..., 0]


def ifft2d(spec: np.ndarray, ncols: int) -> np.ndarray:
    spec = np.ascontiguousarray(spec)
    rdt = np.float32 if spec.dtype == np.complex64 else np.float64
    suf = _suf(rdt)
    r = spec.shape[0]
    assert spec.shape[1] == ncols // 2 + 1
    out = np.empty((r, ncols), rdt)
    rc = getattr(lib(), f"orc_ifft2d_{suf}")(_ptr(spec.view(rdt)), r, ncols, _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def convolve_fft(img: np.ndarray, ker: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img)
    ker = np.ascontiguousarray(ker, dtype=img.dtype)
    suf = _suf(img.dtype)
    out = np.empty_like(img)
    rc = getattr(lib(), f"orc_convolve_fft_{suf}")(_ptr(img), img.shape[0], img.shape[1], _ptr(ker), ker.shape[0], ker.shape[1], _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def filter2d(img: np.ndarray, kind: int, lo: float, hi: float = 0.0) -> np.ndarray:
    img = np.ascontiguousarray(img)
    suf = _suf(img.dtype)
    out = np.empty_like(img)
    rc = getattr(lib(), f"orc_filter2d_{suf}")(_ptr(img), img.shape[0], img.shape[1], kind, float(lo), float(hi), _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def gaussian_kernel_2d(size: int, sigma: float) -> np.ndarray:
    out = np.empty((size, size), np.float64)
    rc = lib().orc_gaussian_kernel_2d(size, float(sigma), _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def lowpass_mask(nrows: int, ncols: int, cutoff: float) -> np.ndarray:
    out = np.empty((nrows, ncols), np.float64)
    lib().orc_lowpass_mask(nrows, ncols, float(cutoff), _ptr(out))
    return out


def max_threads() -> int:
    return int(lib().orc_max_threads())
