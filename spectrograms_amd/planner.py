"""Plan objects over the C ABI: the Python mirror of SpectrogramPlanner / *Plan (src/python/planner.rs:107-350,
671-750) plus the batched entry point the HIP engine adds (`compute_batch`)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _ffi
from .params import ChromaParams, ErbParams, LogHzParams, LogParams, MelParams, MfccParams, SpectrogramParams, parse_dtype


class Spectrogram:
    """Result container (src/spectrogram.rs:2547-2832; Python class src/python/spectrogram.rs)."""

    def __init__(self, data: np.ndarray, frequencies: np.ndarray, times: np.ndarray, params: SpectrogramParams,
                 db_floor: Optional[float]):
        self._data, self._freqs, self._times, self._params, self._db_floor = data, frequencies, times, params, db_floor

    data = property(lambda s: s._data)
    dtype = property(lambda s: "float32" if s._data.dtype == np.float32 else "float64")
    frequencies = property(lambda s: s._freqs.tolist())
    times = property(lambda s: s._times.tolist())
    n_bins = property(lambda s: s._data.shape[0])
    n_frames = property(lambda s: s._data.shape[1])
    shape = property(lambda s: s._data.shape)
    params = property(lambda s: s._params)
    T = property(lambda s: s._data.T)

    def frequency_range(self) -> Tuple[float, float]:
        return float(self._freqs[0]), float(self._freqs[-1])

    def duration(self) -> float:
        return float(self._times[-1]) if self._times.size else 0.0

    def db_range(self):
        return None if self._db_floor is None else (float(self._data.min()), float(self._data.max()))

    def __len__(self) -> int:
        return self._data.shape[0]

    def __getitem__(self, idx):
        return self._data[idx]

    def __array__(self, dtype=None, copy=None):
        return self._data if dtype is None else self._data.astype(dtype)

    def astype(self, dtype):
        return self._data.astype(dtype)

    # DLPack protocol (src/python/dlpack.rs, spectrogram.rs `real_dlpack`): zero-copy export of the host array
    def __dlpack__(self, **kwargs):
        return self._data.__dlpack__(**kwargs)

    def __dlpack_device__(self):
        return self._data.__dlpack_device__()


class SpectrogramBatch:
    """Device-resident batch of spectrograms: what `Plan.compute_batch_resident` returns.

    `tensor` is a torch tensor [batch, n_bins, n_frames] living where the kernel wrote it (frame axis contiguous, S9).  The
    DLPack export hands that memory to any consumer without a copy (device type kDLROCM), which is what makes the
    reference's `spectrograms.torch.batch` (python/spectrograms/torch.py:200-285: torch.stack over per-signal host arrays,
    then `.to(device)`) a zero-copy view here.
    """

    def __init__(self, tensor, frequencies: np.ndarray, times: np.ndarray, params, db_floor):
        self.tensor, self._freqs, self._times, self._params, self._db_floor = tensor, frequencies, times, params, db_floor

    frequencies = property(lambda s: s._freqs.tolist())
    times = property(lambda s: s._times.tolist())
    params = property(lambda s: s._params)
    shape = property(lambda s: tuple(s.tensor.shape))
    batch_size = property(lambda s: s.tensor.shape[0])
    n_bins = property(lambda s: s.tensor.shape[1])
    n_frames = property(lambda s: s.tensor.shape[2])
    dtype = property(lambda s: str(s.tensor.dtype).replace("torch.", ""))

    def __len__(self) -> int:
        return self.tensor.shape[0]

    def __getitem__(self, i):
        """One signal's spectrogram as a host `Spectrogram` (copies that slice to the host)."""
        return Spectrogram(self.tensor[i].cpu().numpy(), self._freqs, self._times, self._params, self._db_floor)

    def __dlpack__(self, **kwargs):
        return self.tensor.__dlpack__(**kwargs)

    def __dlpack_device__(self):
        return self.tensor.__dlpack_device__()


class StftResult:
    """StftResult (src/spectrogram.rs:534-630)."""

    def __init__(self, data: np.ndarray, frequencies: np.ndarray, sample_rate: float, params):
        self._data, self._freqs, self.sample_rate, self.params = data, frequencies, sample_rate, params

    data = property(lambda s: s._data)
    dtype = property(lambda s: "float32" if s._data.dtype == np.complex64 else "float64")
    frequencies = property(lambda s: s._freqs.tolist())
    n_bins = property(lambda s: s._data.shape[0])
    n_frames = property(lambda s: s._data.shape[1])
    shape = property(lambda s: s._data.shape)

    @property
    def frequency_resolution(self) -> float:
        return self.sample_rate / self.params.n_fft

    @property
    def time_resolution(self) -> float:
        return self.params.hop_size / self.sample_rate

    def norm(self) -> np.ndarray:
        return np.abs(self._data)

    def __array__(self, dtype=None, copy=None):
        return self._data if dtype is None else self._data.astype(dtype)

    def __dlpack__(self, **kwargs):
        return self._data.__dlpack__(**kwargs)

    def __dlpack_device__(self):
        return self._data.__dlpack_device__()


class Chromagram:
    """Chromagram result (src/chroma.rs:132-190): `.data` is (12, n_frames)."""

    LABELS = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]

    def __init__(self, data: np.ndarray, params: "ChromaParams"):
        self._data, self.params = data, params

    data = property(lambda s: s._data)
    dtype = property(lambda s: "float32" if s._data.dtype == np.float32 else "float64")
    n_bins = property(lambda s: s._data.shape[0])
    n_frames = property(lambda s: s._data.shape[1])
    shape = property(lambda s: s._data.shape)

    @staticmethod
    def labels():
        return list(Chromagram.LABELS)

    def __array__(self, dtype=None, copy=None):
        return self._data if dtype is None else self._data.astype(dtype)

    def __dlpack__(self, **kwargs):
        return self._data.__dlpack__(**kwargs)

    def __dlpack_device__(self):
        return self._data.__dlpack_device__()


class Mfcc:
    """Mfcc result (src/mfcc.rs:96-140): `.data` is (n_coefficients, n_frames)."""

    def __init__(self, data: np.ndarray, params: MfccParams):
        self._data, self.params = data, params

    data = property(lambda s: s._data)
    dtype = property(lambda s: "float32" if s._data.dtype == np.float32 else "float64")
    n_bins = property(lambda s: s._data.shape[0])
    n_frames = property(lambda s: s._data.shape[1])
    shape = property(lambda s: s._data.shape)

    def __array__(self, dtype=None, copy=None):
        return self._data if dtype is None else self._data.astype(dtype)

    def __dlpack__(self, **kwargs):
        return self._data.__dlpack__(**kwargs)

    def __dlpack_device__(self):
        return self._data.__dlpack_device__()


class Plan:
    """One sgx_plan.  Not thread-safe (mirrors `&mut self`; reference plan classes are `unsendable`)."""

    def __init__(self, params: SpectrogramParams, amp: int, mel: Optional[MelParams] = None,
                 db: Optional[LogParams] = None, dtype: Optional[str] = None, device: int = _ffi.DEVICE_CURRENT,
                 mfcc: Optional[MfccParams] = None):
        self._lib = _ffi.lib()
        self._params, self._mel, self._db = params, mel, db
        self._dt = parse_dtype(dtype)
        self._np = np.float32 if self._dt == _ffi.F32 else np.float64
        self._amp = amp
        st = params.stft
        p = _ffi.SgxParams()
        p.n_fft, p.hop_size, p.centre = st.n_fft, st.hop_size, int(st.centre)
        p.window_kind, p.window_param = st.window.kind, st.window.param
        self._cw = None
        if st.window.kind == _ffi.WIN_CUSTOM:
            self._cw = np.ascontiguousarray(st.window.coefficients, np.float64)
            p.custom_window = self._cw.ctypes.data_as(C.POINTER(C.c_double))
            p.custom_window_len = self._cw.size
        p.sample_rate_hz = params.sample_rate
        if isinstance(mel, ChromaParams):
            p.freq_scale = _ffi.FREQ_CHROMA
            p.f_min, p.f_max, p.chroma_tuning, p.chroma_norm = mel.f_min, mel.f_max, mel.tuning, mel.norm.code
        elif isinstance(mel, ErbParams):
            p.freq_scale = _ffi.FREQ_ERB
            p.n_mels, p.f_min, p.f_max = mel.n_filters, mel.f_min, mel.f_max
            p.erb_spacing = 1 if mel.spacing == "apple_tr35" else 0
        elif isinstance(mel, LogHzParams):
            p.freq_scale = _ffi.FREQ_LOGHZ
            p.n_mels, p.f_min, p.f_max = mel.n_bins, mel.f_min, mel.f_max
        elif mel is not None:
            p.freq_scale = _ffi.FREQ_MEL
            p.n_mels, p.f_min, p.f_max, p.mel_norm = mel.n_mels, mel.f_min, mel.f_max, mel.norm.code
        else:
            p.freq_scale = _ffi.FREQ_LINEAR
        p.amp_scale = amp
        p.has_log_params = int(db is not None)
        p.floor_db = db.floor_db if db is not None else 0.0
        p.dtype, p.device = self._dt, device
        self._mfcc = mfcc
        if mfcc is not None:
            p.n_mfcc, p.mfcc_include_c0, p.mfcc_lifter = mfcc.n_mfcc, int(mfcc.include_c0), mfcc.lifter
        h = C.c_void_p()
        _ffi.raise_status(self._lib.sgx_plan_create(C.byref(p), C.byref(h)))
        self._h = h
        self.n_fft = st.n_fft
        self._device = int(self._lib.sgx_plan_device(h))  # resolved ordinal (DEVICE_CURRENT was bound at creation); -2: host only
        self._frame_plan = None  # no-centre sibling for compute_frame, created on first use

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.sgx_plan_destroy(h)
            self._h = None

    # ---- queries
    @property
    def dtype(self) -> str:
        return "float32" if self._dt == _ffi.F32 else "float64"

    @property
    def is_complex(self) -> bool:
        return self._amp == _ffi.AMP_COMPLEX

    @property
    def device(self) -> int:
        """HIP device ordinal the plan's tables live on (-2: host-only plan)."""
        return self._device

    def reserve(self, batch: int, n_samples: int, host_staging: bool = True, inverse: bool = False) -> None:
        """Pre-size the plan-owned scratch so that later calls of up to this size do not allocate (sgx_reserve)."""
        _ffi.raise_status(self._lib.sgx_reserve(self._h, int(batch), int(n_samples), int(host_staging), int(inverse)), self._h)

    def _check_device(self, t, what: str) -> None:
        # the kernels run on the plan's device with the plan's tables: a tensor that lives elsewhere would be a wild pointer there
        if not t.is_cuda or t.device.index != self._device:
            raise ValueError(f"{what} is on {t.device}, the plan is bound to cuda:{self._device}")

    @property
    def kernel_name(self) -> str:
        return self._lib.sgx_kernel_name(self._h).decode()

    def output_shape(self, signal_length: int) -> Tuple[int, int]:
        if int(signal_length) <= 0:
            raise ValueError("signal_length must be > 0")
        nb, nf = C.c_size_t(), C.c_size_t()
        _ffi.raise_status(self._lib.sgx_output_shape(self._h, int(signal_length), C.byref(nb), C.byref(nf)), self._h)
        return nb.value, nf.value

    def axes(self, n_frames: int):
        nb = self.output_shape(self.n_fft)[0]
        f, t = np.empty(nb, np.float64), np.empty(n_frames, np.float64)
        _ffi.raise_status(self._lib.sgx_axes(self._h, n_frames, f.ctypes.data_as(C.POINTER(C.c_double)),
                                             t.ctypes.data_as(C.POINTER(C.c_double))), self._h)
        return f, t

    def window(self) -> np.ndarray:
        w = np.empty(self.n_fft, np.float64)
        _ffi.raise_status(self._lib.sgx_window(self._h, w.ctypes.data_as(C.POINTER(C.c_double))), self._h)
        return w

    def mel_weights(self):
        nnz = C.c_size_t()
        _ffi.raise_status(self._lib.sgx_mel_weights(self._h, C.byref(nnz), None, None, None), self._h)
        rows = 12 if isinstance(self._mel, ChromaParams) else (getattr(self._mel, "n_mels", None) or getattr(self._mel, "n_bins", None)
                                                                 or self._mel.n_filters)
        ptr = np.empty(rows + 1, np.uint32)
        col = np.empty(nnz.value, np.uint32)
        val = np.empty(nnz.value, np.float64)
        _ffi.raise_status(self._lib.sgx_mel_weights(self._h, None, ptr.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                    col.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                    val.ctypes.data_as(C.POINTER(C.c_double))), self._h)
        return ptr, col, val

    # ---- compute
    def _host_samples(self, samples, ndim: int) -> np.ndarray:
        x = np.ascontiguousarray(samples, dtype=self._np)  # np.ascontiguousarray(samples, T), functions.rs:29-59
        if x.ndim != ndim:
            raise ValueError(f"samples must be a {ndim}-D array")
        if x.size == 0:
            raise ValueError("samples must be non-empty")  # functions.rs:52-54
        return x

    def compute_batch(self, samples, out=None, stream: int = 0):
        """[B, N] -> [B, n_bins, n_frames] in one launch.  numpy in -> numpy out (host path, plan-owned staging);
        torch CUDA tensor in -> torch CUDA tensor out (zero-copy device path on `stream`/torch's current stream)."""
        if type(samples).__module__.startswith("torch"):
            return self._compute_batch_torch(samples, out, stream)
        x = self._host_samples(samples, 2)
        b, n = x.shape
        nb, nf = self.output_shape(n)
        cdt = (np.complex64 if self._dt == _ffi.F32 else np.complex128) if self.is_complex else self._np
        if out is None:
            out = np.empty((b, nb, nf), cdt)
        elif out.dtype != cdt or not out.flags.c_contiguous:
            raise ValueError("out must be C-contiguous with the plan's dtype")
        elems = out.size * (2 if self.is_complex else 1)
        _ffi.raise_status(self._lib.sgx_execute(self._h, x.ctypes.data, b, n, n, out.ctypes.data, elems,
                                                _ffi.MEM_HOST, None), self._h)
        return out

    def _compute_batch_torch(self, x, out, stream):
        import torch
        tdt = torch.float32 if self._dt == _ffi.F32 else torch.float64
        if not x.is_cuda or x.dtype != tdt or x.dim() != 2 or x.stride(1) != 1:
            raise ValueError("device path needs a 2-D CUDA tensor of the plan's dtype with unit inner stride")
        if x.numel() == 0:
            raise ValueError("samples must be non-empty")
        self._check_device(x, "samples")
        b, n = x.shape
        nb, nf = self.output_shape(n)
        shape = (b, nb, nf, 2) if self.is_complex else (b, nb, nf)
        if out is None:
            out = torch.empty(shape, dtype=tdt, device=x.device)
        else:
            if tuple(out.shape) != shape:
                raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {shape}, got {tuple(out.shape)}", shape, tuple(out.shape))
            if out.dtype != tdt or not out.is_contiguous():
                raise ValueError("out must be a contiguous tensor of the plan's dtype")
            self._check_device(out, "out")
        s = stream or torch.cuda.current_stream(x.device).cuda_stream
        _ffi.raise_status(self._lib.sgx_execute(self._h, x.data_ptr(), b, n, x.stride(0), out.data_ptr(), out.numel(),
                                                _ffi.MEM_DEVICE, C.c_void_p(s)), self._h)
        return torch.view_as_complex(out) if self.is_complex else out

    def compute_batch_resident(self, samples, stream: int = 0) -> "SpectrogramBatch":
        """[B, N] signals (torch device tensor, or a numpy array that is uploaded once) -> a device-resident
        `SpectrogramBatch` with axes and DLPack export; nothing comes back to the host."""
        import torch
        if not type(samples).__module__.startswith("torch"):
            dev = torch.device("cuda", torch.cuda.current_device())
            samples = torch.from_numpy(self._host_samples(samples, 2)).to(dev)
        t = self._compute_batch_torch(samples, None, stream)
        freqs, times = self.axes(t.shape[2])
        return SpectrogramBatch(t, freqs, times, self._params, self._db.floor_db if self._db else None)

    def time_batch_torch(self, x, out, iters: int, stream: int = 0) -> float:
        """Mean device milliseconds per launch over `iters` back-to-back launches (hipEvents on the launch stream)."""
        import torch
        self._check_device(x, "samples")
        self._check_device(out, "out")
        s = stream or torch.cuda.current_stream(x.device).cuda_stream
        ms = C.c_float()
        b, n = x.shape
        _ffi.raise_status(self._lib.sgx_execute_timed(self._h, x.data_ptr(), b, n, x.stride(0), out.data_ptr(),
                                                      out.numel(), C.c_void_p(s), iters, C.byref(ms)), self._h)
        return float(ms.value)

    def compute(self, samples):
        """Per-signal compute (SpectrogramPlan::compute :240-294 / StftPlan::compute :1424-1458)."""
        x = self._host_samples(samples, 1)
        data = self.compute_batch(x[None, :])[0]
        freqs, times = self.axes(data.shape[1])
        if self.is_complex:
            return StftResult(data, freqs, self._params.sample_rate, self._params.stft)
        if self._mfcc is not None:
            return Mfcc(data, self._mfcc)
        if isinstance(self._mel, ChromaParams):
            return Chromagram(data, self._mel)
        return Spectrogram(data, freqs, times, self._params, self._db.floor_db if self._db else None)

    def compute_frame(self, samples, frame_idx: int) -> np.ndarray:
        """SpectrogramPlan::compute_frame (:335-372): one column, computed from the n_fft-sample span it covers.  The reference
        reuses the plan's FFT / window / mapping for this; here the one-frame launch runs on a no-centre sibling plan that is
        created on first use and kept for the life of this plan (same device, same tables)."""
        x = self._host_samples(samples, 1)
        st = self._params.stft
        pad = st.n_fft // 2 if st.centre else 0
        lo = int(frame_idx) * st.hop_size - pad
        seg = np.zeros(st.n_fft, self._np)
        a, b = max(lo, 0), min(lo + st.n_fft, x.size)
        if b > a:
            seg[a - lo:b - lo] = x[a:b]
        sib = self if not st.centre else self._frame_plan
        if sib is None:
            sib = self._frame_plan = Plan(SpectrogramParams(type(st)(st.n_fft, st.hop_size, st.window, False), self._params.sample_rate),
                                          self._amp, self._mel, self._db, self.dtype, self._device, self._mfcc)
        return sib.compute_batch(seg[None, :])[0][:, 0]

    def r2c(self, frame) -> np.ndarray:
        """Conforming R2cPlan::process (src/fft_backend.rs:423-431) on one frame of n_fft reals."""
        x = np.ascontiguousarray(frame, dtype=self._np)
        out = np.empty(self.n_fft // 2 + 1, np.complex64 if self._dt == _ffi.F32 else np.complex128)
        _ffi.raise_status(self._lib.sgx_r2c(self._h, x.ctypes.data, x.size, out.ctypes.data, out.size), self._h)
        return out


    def c2r(self, spectrum) -> np.ndarray:
        """Conforming C2rPlan::process (src/fft_backend.rs:526-565): n_fft/2+1 complex bins -> n_fft reals, scaled 1/n_fft."""
        cdt = np.complex64 if self._dt == _ffi.F32 else np.complex128
        x = np.ascontiguousarray(spectrum, dtype=cdt)
        out = np.empty(self.n_fft, self._np)
        _ffi.raise_status(self._lib.sgx_c2r(self._h, x.ctypes.data, x.size, out.ctypes.data, out.size), self._h)
        return out

    def istft_length(self, n_frames: int) -> int:
        n = C.c_size_t()
        _ffi.raise_status(self._lib.sgx_istft_length(self._h, n_frames, C.byref(n)), self._h)
        return n.value

    def istft_batch(self, stft, out=None, stream: int = 0):
        """Batched istft (src/spectrogram.rs:4860-4946) of [batch][n_bins][n_frames] complex matrices with this plan's
        n_fft / hop / window / centre.  numpy in -> numpy out (plan-owned staging); torch device tensor in -> device tensor
        out, asynchronous on the current (or given) stream."""
        cdt = np.complex64 if self._dt == _ffi.F32 else np.complex128
        if isinstance(stft, np.ndarray) or not hasattr(stft, "data_ptr"):
            m = np.ascontiguousarray(stft, dtype=cdt)
            if m.ndim != 3:
                raise ValueError("stft must be a 3-D array (batch, n_bins, n_frames)")
            b, nb, nf = m.shape
            if b == 0 or nf == 0:
                raise ValueError("stft must be non-empty")
            if out is None and nb == self.n_fft // 2 + 1:
                out = np.empty((b, self.istft_length(nf)), self._np)
            elif out is None:
                out = np.empty((b, 0), self._np)
            _ffi.raise_status(self._lib.sgx_istft(self._h, m.ctypes.data, b, nb, nf, out.ctypes.data, out.size, _ffi.MEM_HOST,
                                                  C.c_void_p(stream)), self._h)
            return out
        import torch
        want = torch.complex64 if self._dt == _ffi.F32 else torch.complex128
        if stft.dtype != want or not stft.is_contiguous() or stft.dim() != 3:
            raise ValueError("stft must be a contiguous 3-D tensor of the plan's complex dtype")
        self._check_device(stft, "stft")
        b, nb, nf = stft.shape
        rdt = torch.float32 if self._dt == _ffi.F32 else torch.float64
        if out is None:
            n_out = self.istft_length(nf) if nb == self.n_fft // 2 + 1 else 0
            out = torch.empty((b, n_out), dtype=rdt, device=stft.device)
        else:
            if out.dtype != rdt or not out.is_contiguous() or out.dim() != 2 or out.shape[0] != b:
                raise ValueError("out must be a contiguous (batch, n_samples) tensor of the plan's real dtype")
            self._check_device(out, "out")
        s = stream or torch.cuda.current_stream(stft.device).cuda_stream
        _ffi.raise_status(self._lib.sgx_istft(self._h, stft.data_ptr(), b, nb, nf, out.data_ptr(), out.numel(), _ffi.MEM_DEVICE,
                                              C.c_void_p(s)), self._h)
        return out

    def istft(self, stft_matrix) -> np.ndarray:
        m = np.asarray(stft_matrix.data if isinstance(stft_matrix, StftResult) else stft_matrix)
        if m.ndim != 2:
            raise ValueError("stft_matrix must be a 2-D array (n_bins, n_frames)")
        return self.istft_batch(m[None])[0]


class SpectrogramPlanner:
    """src/python/planner.rs:107-350 — same method names and argument order."""

    def __init__(self, device: int = _ffi.DEVICE_CURRENT):
        self._device = device

    def linear_power_plan(self, params, dtype=None):
        return Plan(params, _ffi.AMP_POWER, None, None, dtype, self._device)

    def linear_magnitude_plan(self, params, dtype=None):
        return Plan(params, _ffi.AMP_MAGNITUDE, None, None, dtype, self._device)

    def linear_db_plan(self, params, db_params, dtype=None):
        return Plan(params, _ffi.AMP_DECIBELS, None, db_params, dtype, self._device)

    def mel_power_plan(self, params, mel_params, dtype=None):
        return Plan(params, _ffi.AMP_POWER, mel_params, None, dtype, self._device)

    def mel_magnitude_plan(self, params, mel_params, dtype=None):
        return Plan(params, _ffi.AMP_MAGNITUDE, mel_params, None, dtype, self._device)

    def mel_db_plan(self, params, mel_params, db_params, dtype=None):
        return Plan(params, _ffi.AMP_DECIBELS, mel_params, db_params, dtype, self._device)

    def loghz_power_plan(self, params, loghz_params, dtype=None):
        return Plan(params, _ffi.AMP_POWER, loghz_params, None, dtype, self._device)

    def loghz_magnitude_plan(self, params, loghz_params, dtype=None):
        return Plan(params, _ffi.AMP_MAGNITUDE, loghz_params, None, dtype, self._device)

    def loghz_db_plan(self, params, loghz_params, db_params, dtype=None):
        return Plan(params, _ffi.AMP_DECIBELS, loghz_params, db_params, dtype, self._device)

    def erb_power_plan(self, params, erb_params, dtype=None):  # src/python/planner.rs:343-365
        return Plan(params, _ffi.AMP_POWER, erb_params, None, dtype, self._device)

    def erb_magnitude_plan(self, params, erb_params, dtype=None):
        return Plan(params, _ffi.AMP_MAGNITUDE, erb_params, None, dtype, self._device)

    def erb_db_plan(self, params, erb_params, db_params, dtype=None):
        return Plan(params, _ffi.AMP_DECIBELS, erb_params, db_params, dtype, self._device)

    def chroma_plan(self, stft_params, sample_rate, chroma_params, dtype=None):
        """Plan for chromagram() (src/chroma.rs:470-505): reusable, with `compute_batch`."""
        return Plan(SpectrogramParams(stft_params, sample_rate), _ffi.AMP_MAGNITUDE, chroma_params, None, dtype, self._device)

    def mfcc_plan(self, stft_params, sample_rate, n_mels, mfcc_params, dtype=None):
        """Plan form of `mfcc()` (src/mfcc.rs:359-379): Mel 0..sr/2, floor -80 dB, then DCT-II + lifter."""
        params = SpectrogramParams(stft_params, sample_rate)
        return Plan(params, _ffi.AMP_DECIBELS, MelParams(n_mels, 0.0, sample_rate / 2.0), LogParams(-80.0), dtype,
                    self._device, mfcc_params)

    def stft_plan(self, params, dtype=None):
        """StftPlan::new (src/spectrogram.rs:1204-1228)."""
        return Plan(params, _ffi.AMP_COMPLEX, None, None, dtype, self._device)
