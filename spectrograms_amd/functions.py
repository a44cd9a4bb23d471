"""One-shot functions with the reference's names and argument order (src/python/functions.rs:92-266, 757-773)."""
from __future__ import annotations

import numpy as np

from . import _ffi
from .params import LogParams, MelParams, SpectrogramParams
from .planner import Plan as _Plan

# ---- plan cache for the one-shot functions ------------------------------------------------------------------------------
# The reference caches FFT plans process-wide (src/fft_backend.rs:946-1076; clear_fft_plan_cache / fft_plan_cache_info in the
# Python module).  Here a "plan" also owns its device tables (window, twiddles, filterbank), so a one-shot compute_* call that
# repeats earlier parameters reuses the whole plan instead of rebuilding and re-uploading them.  Least-recently-used, bounded.
_PLAN_CACHE = {}
_PLAN_CACHE_MAX = 32


def _key(obj):
    """Hashable identity of a parameter object by VALUE: every field that reaches the plan, array contents included (two custom
    windows of equal length must not share a plan).  Unknown types are an error — never repr(), which may omit fields or be an
    address that a later object reuses."""
    if obj is None or isinstance(obj, (int, float, str, bool)):
        return obj
    if isinstance(obj, np.generic):
        return obj.item()
    if isinstance(obj, np.ndarray):
        return (obj.dtype.str, obj.shape, obj.tobytes())
    if isinstance(obj, (list, tuple)):
        return tuple(_key(v) for v in obj)
    if isinstance(obj, dict):
        return tuple(sorted((k, _key(v)) for k, v in obj.items()))
    fields = {}
    for klass in type(obj).__mro__:  # __slots__ classes have no __dict__: walk the declared slots of every base
        slots = klass.__dict__.get("__slots__", ())
        for name in ((slots,) if isinstance(slots, str) else slots):
            if name not in ("__dict__", "__weakref__") and hasattr(obj, name):
                fields[name] = getattr(obj, name)
    if hasattr(obj, "__dict__"):
        fields.update(vars(obj))
    if not fields and not hasattr(obj, "__dict__") and not any("__slots__" in k.__dict__ for k in type(obj).__mro__):
        raise TypeError(f"cannot build a plan-cache key for {type(obj).__name__}")
    return (type(obj).__module__, type(obj).__qualname__) + tuple((k, _key(v)) for k, v in sorted(fields.items()))


def Plan(params, amp, mapping, db, dtype, mfcc=None, inverse=False):
    """Cached constructor used by every one-shot function below (same arguments as planner.Plan)."""
    from .params import parse_dtype
    import torch
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    key = (_key(params), amp, _key(mapping), _key(db), parse_dtype(dtype), _key(mfcc), bool(inverse), dev)
    plan = _PLAN_CACHE.pop(key, None)
    if plan is None:
        plan = _Plan(params, amp, mapping, db, dtype, mfcc=mfcc)
        while len(_PLAN_CACHE) >= _PLAN_CACHE_MAX:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
    _PLAN_CACHE[key] = plan  # most recently used last
    return plan


def clear_fft_plan_cache() -> None:
    """Drop every cached plan (and its device tables)."""
    _PLAN_CACHE.clear()


def fft_plan_cache_info():
    """(n_forward_plans, n_inverse_plans) currently cached."""
    inv = sum(1 for k in _PLAN_CACHE if k[6])
    return len(_PLAN_CACHE) - inv, inv


def compute_linear_power_spectrogram(samples, params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_POWER, None, db, dtype).compute(samples)


def compute_linear_magnitude_spectrogram(samples, params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_MAGNITUDE, None, db, dtype).compute(samples)


def compute_linear_db_spectrogram(samples, params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_DECIBELS, None, db, dtype).compute(samples)


def compute_mel_power_spectrogram(samples, params, mel_params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_POWER, mel_params, db, dtype).compute(samples)


def compute_mel_magnitude_spectrogram(samples, params, mel_params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_MAGNITUDE, mel_params, db, dtype).compute(samples)


def compute_mel_db_spectrogram(samples, params, mel_params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_DECIBELS, mel_params, db, dtype).compute(samples)


def compute_stft(samples, params, dtype=None):
    return Plan(params, _ffi.AMP_COMPLEX, None, None, dtype).compute(samples)


def compute_mfcc(samples, stft_params, sample_rate, n_mels, mfcc_params, dtype=None):
    """src/python/functions.rs:606-640 -> mfcc() src/mfcc.rs:359-379."""
    params = SpectrogramParams(stft_params, sample_rate)
    return Plan(params, _ffi.AMP_DECIBELS, MelParams(n_mels, 0.0, sample_rate / 2.0), LogParams(-80.0), dtype,
                mfcc=mfcc_params).compute(samples)


def compute_loghz_power_spectrogram(samples, params, loghz_params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_POWER, loghz_params, db, dtype).compute(samples)


def compute_loghz_magnitude_spectrogram(samples, params, loghz_params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_MAGNITUDE, loghz_params, db, dtype).compute(samples)


def compute_loghz_db_spectrogram(samples, params, loghz_params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_DECIBELS, loghz_params, db, dtype).compute(samples)


def compute_erb_power_spectrogram(samples, params, erb_params, db=None, dtype=None):  # src/python/functions.rs:274-300
    return Plan(params, _ffi.AMP_POWER, erb_params, db, dtype).compute(samples)


def compute_erb_magnitude_spectrogram(samples, params, erb_params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_MAGNITUDE, erb_params, db, dtype).compute(samples)


def compute_erb_db_spectrogram(samples, params, erb_params, db=None, dtype=None):
    return Plan(params, _ffi.AMP_DECIBELS, erb_params, db, dtype).compute(samples)


def compute_irfft(spectrum, n_fft, dtype=None):
    """irfft (src/spectrogram.rs:4789-4811; Python src/python/functions.rs:970-983): n_fft/2+1 bins -> n_fft samples."""
    from .params import StftParams, WindowType
    if int(n_fft) <= 0:
        raise ValueError("n_fft must be > 0")
    params = SpectrogramParams(StftParams(int(n_fft), int(n_fft), WindowType.rectangular, False), 1.0)
    return Plan(params, _ffi.AMP_COMPLEX, None, None, dtype, inverse=True).c2r(spectrum)


def compute_istft(stft_matrix, n_fft, hop_size, window, center=True, dtype=None):
    """istft (src/spectrogram.rs:4860-4946; Python src/python/functions.rs:1018-1038)."""
    from .params import StftParams
    if int(n_fft) <= 0 or int(hop_size) <= 0:
        raise ValueError("n_fft and hop_size must be > 0")
    m = np.asarray(getattr(stft_matrix, "data", stft_matrix))
    if m.ndim == 2 and m.shape[0] != int(n_fft) // 2 + 1:  # checked before hop_size (:4876-4882)
        raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {int(n_fft) // 2 + 1}, got {m.shape[0]}")
    params = SpectrogramParams(StftParams(int(n_fft), int(hop_size), window, bool(center)), 1.0)
    return Plan(params, _ffi.AMP_COMPLEX, None, None, dtype, inverse=True).istft(m)


def compute_chromagram(samples, stft_params, sample_rate, chroma_params, dtype=None):
    """chromagram() (src/chroma.rs:470-505; Python src/python/functions.rs:551-567): 12 pitch classes x n_frames."""
    return Plan(SpectrogramParams(stft_params, sample_rate), _ffi.AMP_MAGNITUDE, chroma_params, None, dtype).compute(samples)


# ---- single-frame transforms sharing the backend (src/spectrogram.rs:4490-4693) -----------------------
def _one_frame(samples, n_fft, window, amp, dtype):
    """Zero-pad `samples` to n_fft (`fft` :4479-4503 / `power_spectrum` :4611-4633), optional window, one R2C on the GPU."""
    from .params import StftParams, WindowType
    n_fft = int(n_fft)
    if n_fft <= 0:
        raise ValueError("n_fft must be non-zero positive integer")
    x = np.ascontiguousarray(samples).reshape(-1)
    if x.size == 0:
        raise ValueError("samples must be non-empty")
    if x.size > n_fft:
        raise _ffi.InvalidInputError(f"Invalid input: Input length ({x.size}) exceeds FFT size ({n_fft})")
    params = SpectrogramParams(StftParams(n_fft, n_fft, window if window is not None else WindowType.rectangular, False), 1.0)
    plan = Plan(params, amp, None, None, dtype)
    padded = np.zeros(n_fft, plan._np)
    padded[:x.size] = x
    return plan.compute_batch(padded[None, :])[0][:, 0]


def compute_fft(samples, n_fft=None, dtype=None):
    """fft (src/spectrogram.rs:4475-4506; Python src/python/functions.rs:787-805): n_fft/2+1 complex bins of the zero-padded input."""
    n = np.size(samples) if n_fft is None else n_fft
    return _one_frame(samples, n, None, _ffi.AMP_COMPLEX, dtype)


def compute_rfft(samples, n_fft, dtype=None):
    """rfft (:4535-4541): |fft|."""
    return np.abs(_one_frame(samples, n_fft, None, _ffi.AMP_COMPLEX, dtype))


def compute_power_spectrum(samples, n_fft, window=None, dtype=None):
    """power_spectrum (:4611-4643): |X|^2 of the zero-padded, optionally windowed frame."""
    return _one_frame(samples, n_fft, window, _ffi.AMP_POWER, dtype)


def compute_magnitude_spectrum(samples, n_fft, window=None, dtype=None):
    """magnitude_spectrum (:4684-4693): sqrt of the power spectrum."""
    return _one_frame(samples, n_fft, window, _ffi.AMP_MAGNITUDE, dtype)
