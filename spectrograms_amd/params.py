"""Parameter types mirroring the reference's Python classes (src/python/params.rs; names and argument order from
python/spectrograms/__init__.pyi).  Validation happens at the same API point and with the same message text as the
reference constructors (src/spectrogram.rs:3479-3506, 3793-3813, 4071-4077, 4129-4140; src/window.rs:134-203)."""
from __future__ import annotations

import math
import re
from typing import Optional

import numpy as np

from . import _ffi


class WindowType:
    """src/window.rs:19-50; Python surface src/python/params.rs:43-174."""

    __slots__ = ("kind", "param", "coefficients")

    def __init__(self, kind: int, param: float = 0.0, coefficients: Optional[np.ndarray] = None):
        self.kind, self.param, self.coefficients = kind, float(param), coefficients

    @classmethod
    def kaiser(cls, beta: float) -> "WindowType":
        return cls(_ffi.WIN_KAISER, beta)

    # window generators of the Python class (src/python/params.rs:100-174 -> make_window, src/spectrogram.rs:2159-2235);
    # the coefficients come from the engine's own table builder (host-only plan: no GPU needed)
    @staticmethod
    def _make(window: "WindowType", n: int, dtype) -> np.ndarray:
        from .planner import Plan
        if int(n) <= 0:
            raise ValueError("n must be > 0")
        params = SpectrogramParams(StftParams(int(n), int(n), window, False), 1.0)
        w = Plan(params, _ffi.AMP_POWER, None, None, dtype, device=_ffi.DEVICE_HOST_ONLY).window()
        return w.astype(np.float32 if parse_dtype(dtype) == _ffi.F32 else np.float64)

    @classmethod
    def make_hanning(cls, n: int, dtype=None) -> np.ndarray:
        return cls._make(cls.hanning, n, dtype)

    @classmethod
    def make_hamming(cls, n: int, dtype=None) -> np.ndarray:
        return cls._make(cls.hamming, n, dtype)

    @classmethod
    def make_blackman(cls, n: int, dtype=None) -> np.ndarray:
        return cls._make(cls.blackman, n, dtype)

    @classmethod
    def make_kaiser(cls, n: int, beta: float, dtype=None) -> np.ndarray:
        return cls._make(cls.kaiser(beta), n, dtype)

    @classmethod
    def make_gaussian(cls, n: int, std: float, dtype=None) -> np.ndarray:
        return cls._make(cls.gaussian(std), n, dtype)

    @classmethod
    def gaussian(cls, std: float) -> "WindowType":
        return cls(_ffi.WIN_GAUSSIAN, std)

    @classmethod
    def custom(cls, coefficients, normalize: Optional[str] = None) -> "WindowType":
        # WindowType::custom_with_normalization, src/window.rs:134-203
        c = np.array(coefficients, dtype=np.float64).reshape(-1)
        if c.size == 0:
            raise ValueError("Custom window coefficients cannot be empty")
        bad = np.flatnonzero(~np.isfinite(c))
        if bad.size:
            raise ValueError(f"Window coefficient at index {int(bad[0])} is not finite: {c[bad[0]]}")
        if normalize is not None:
            if normalize == "sum":
                s = float(c.sum())
                if s == 0.0:
                    raise ValueError("Cannot normalize window by sum: sum is zero")
                c = c / s
            elif normalize in ("peak", "max"):
                m = float(c.max())
                if m == 0.0:
                    raise ValueError("Cannot normalize window by peak: maximum is zero")
                c = c / m
            elif normalize in ("energy", "rms"):
                e = float((c * c).sum())
                if e == 0.0:
                    raise ValueError("Cannot normalize window by energy: energy is zero")
                c = c / math.sqrt(e)
            else:
                raise ValueError(f"Unknown normalization mode '{normalize}'. Valid modes: 'sum', 'peak', 'energy'")
        return cls(_ffi.WIN_CUSTOM, 0.0, np.ascontiguousarray(c))

    # "hann" / "hanning" / "hamm" / "hamming" / "rect" / "rectangle" / "blackman" / "kaiser=<x>" / "gaussian=<x>", any case,
    # surrounding blanks ignored, <x> = digits[.digits] — FromStr for WindowType, src/window.rs:276-338 (same grammar, same
    # message texts)
    _SPEC = re.compile(r"^(?:(?P<name>rect|rectangle|hann|hanning|hamm|hamming|blackman)|(?P<param_name>kaiser|gaussian)=(?P<param>\d+(\.\d+)?))$",
                       re.IGNORECASE | re.ASCII)

    @classmethod
    def from_str(cls, s: str) -> "WindowType":
        if not isinstance(s, str):
            raise TypeError("window specification must be a str")
        if s == "":
            raise _ffi.InvalidInputError("Invalid input: Input must not be empty. Must be one of ['rectangular', 'hanning', "
                                         "'hamming', 'blackman', 'gaussian', 'kaiser']")
        # (str::trim strips Unicode White_Space; str.strip() does the same for str)
        m = cls._SPEC.match(s.strip())
        if m is None:
            raise _ffi.InvalidInputError(f"Invalid input: Invalid window specification '{s}'")
        if m.group("name") is not None:
            name = m.group("name").lower()
            return {"rect": cls.rectangular, "rectangle": cls.rectangular, "hann": cls.hanning, "hanning": cls.hanning,
                    "hamm": cls.hamming, "hamming": cls.hamming, "blackman": cls.blackman}[name]
        value = float(m.group("param"))
        return cls.kaiser(value) if m.group("param_name").lower() == "kaiser" else cls.gaussian(value)

    def __repr__(self) -> str:
        names = ["Rectangular", "Hanning", "Hamming", "Blackman", f"Kaiser(beta={self.param})",
                 f"Gaussian(std={self.param})", f"Custom(n={0 if self.coefficients is None else self.coefficients.size})"]
        return f"WindowType.{names[self.kind]}"


WindowType.rectangular = WindowType(_ffi.WIN_RECTANGULAR)
WindowType.hanning = WindowType(_ffi.WIN_HANNING)
WindowType.hamming = WindowType(_ffi.WIN_HAMMING)
WindowType.blackman = WindowType(_ffi.WIN_BLACKMAN)


class StftParams:
    """StftParams(n_fft, hop_size, window, centre=True) — src/python/params.rs:487-503."""

    def __init__(self, n_fft: int, hop_size: int, window: WindowType, centre: bool = True):
        n_fft, hop_size = int(n_fft), int(hop_size)
        if n_fft <= 0 or hop_size <= 0:
            raise ValueError("n_fft and hop_size must be > 0")  # NonZeroUsize
        if hop_size > n_fft:
            raise _ffi.InvalidInputError("Invalid input: hop_size must be <= n_fft")
        if window.kind == _ffi.WIN_CUSTOM and window.coefficients.size != n_fft:
            raise _ffi.InvalidInputError(
                f"Invalid input: Custom window size ({window.coefficients.size}) must match n_fft ({n_fft})")
        self.n_fft, self.hop_size, self.window, self.centre = n_fft, hop_size, window, bool(centre)


class LogParams:
    """LogParams(floor_db) — src/python/params.rs:583-585."""

    def __init__(self, floor_db: float):
        if not math.isfinite(floor_db):
            raise _ffi.InvalidInputError("Invalid input: floor_db must be finite")
        self.floor_db = float(floor_db)


class SpectrogramParams:
    """SpectrogramParams(stft, sample_rate) — src/python/params.rs:630-635."""

    def __init__(self, stft: StftParams, sample_rate: float):
        if not (sample_rate > 0.0 and math.isfinite(sample_rate)):
            raise _ffi.InvalidInputError("Invalid input: sample_rate_hz must be finite and > 0")
        self.stft, self.sample_rate = stft, float(sample_rate)

    @classmethod
    def speech_default(cls, sample_rate: float) -> "SpectrogramParams":
        return cls(StftParams(512, 160, WindowType.hanning, True), sample_rate)

    @classmethod
    def music_default(cls, sample_rate: float) -> "SpectrogramParams":
        return cls(StftParams(2048, 512, WindowType.hanning, True), sample_rate)


class MelNorm:
    """MelNorm — src/spectrogram.rs:2385-2429."""

    def __init__(self, code: int):
        self.code = code


MelNorm.none = MelNorm(_ffi.MELNORM_NONE)
MelNorm.slaney = MelNorm(_ffi.MELNORM_SLANEY)
MelNorm.l1 = MelNorm(_ffi.MELNORM_L1)
MelNorm.l2 = MelNorm(_ffi.MELNORM_L2)


class MelParams:
    """MelParams(n_mels, f_min, f_max, norm=None) — src/python/params.rs:812-850."""

    def __init__(self, n_mels: int, f_min: float, f_max: float, norm: Optional[MelNorm] = None):
        if int(n_mels) <= 0:
            raise ValueError("n_mels must be > 0")
        if f_min < 0.0:
            raise _ffi.InvalidInputError("Invalid input: f_min must be >= 0")
        if f_max <= f_min:
            raise _ffi.InvalidInputError("Invalid input: f_max must be > f_min")
        self.n_mels, self.f_min, self.f_max = int(n_mels), float(f_min), float(f_max)
        self.norm = norm if norm is not None else MelNorm.none


class LogHzParams:
    """LogHzParams(n_bins, f_min, f_max) — src/spectrogram.rs:3935-3990."""

    def __init__(self, n_bins: int, f_min: float, f_max: float):
        if int(n_bins) <= 0:
            raise ValueError("n_bins must be > 0")
        if not (f_min > 0.0 and math.isfinite(f_min)):
            raise _ffi.InvalidInputError("Invalid input: f_min must be finite and > 0")
        if f_max <= f_min:
            raise _ffi.InvalidInputError("Invalid input: f_max must be > f_min")
        self.n_bins, self.f_min, self.f_max = int(n_bins), float(f_min), float(f_max)


class ErbParams:
    """ErbParams(n_filters, f_min, f_max) — src/erb.rs:27-92, Python class src/python/params.rs:910-980.

    `spacing`: "linear" (uniform on the Glasberg & Moore ERB scale, default) or "apple_tr35" (ErbSpacing, erb.rs:14-25).
    """

    def __init__(self, n_filters: int, f_min: float, f_max: float, spacing: str = "linear"):
        if int(n_filters) < 2:
            raise _ffi.InvalidInputError("Invalid input: n_filters must be >= 2 (single filter would cause division by zero)")
        if f_min < 0.0 or math.isinf(f_min):
            raise _ffi.InvalidInputError("Invalid input: f_min must be finite and >= 0")
        if f_max <= f_min:
            raise _ffi.InvalidInputError("Invalid input: f_max must be > f_min")
        if spacing not in ("linear", "apple_tr35"):
            raise ValueError("spacing must be 'linear' or 'apple_tr35'")
        self.n_filters, self.f_min, self.f_max, self.spacing = int(n_filters), float(f_min), float(f_max), spacing

    def with_spacing(self, spacing: str) -> "ErbParams":
        return ErbParams(self.n_filters, self.f_min, self.f_max, spacing)

    @staticmethod
    def speech_standard() -> "ErbParams":  # erb.rs:170-173
        return ErbParams(40, 0.0, 8000.0)

    @staticmethod
    def music_standard(sample_rate: float) -> "ErbParams":  # erb.rs:190-192
        return ErbParams(64, 0.0, sample_rate / 2.0)

    def __repr__(self):
        return f"ErbParams(n_filters={self.n_filters}, f_min={self.f_min}, f_max={self.f_max})"


GammatoneParams = ErbParams


class ChromaNorm:
    """ChromaNorm (src/chroma.rs:24-31; Python class src/python/params.rs:1101-1158): `ChromaNorm.none/l1/l2/max`."""

    def __init__(self, name: str, code: int):
        self.name, self.code = name, code

    def __repr__(self):
        return f"ChromaNorm.{self.name}"

    def __eq__(self, other):
        return isinstance(other, ChromaNorm) and other.code == self.code

    def __hash__(self):
        return hash(self.code)


ChromaNorm.none = ChromaNorm("none", 0)
ChromaNorm.l1 = ChromaNorm("l1", 1)
ChromaNorm.l2 = ChromaNorm("l2", 2)
ChromaNorm.max = ChromaNorm("max", 3)


class ChromaParams:
    """ChromaParams(tuning=440.0, f_min=32.7, f_max=4186.0, norm=None) — src/chroma.rs:12-130, src/python/params.rs:1161-1240.
    `norm=None` is the enum default, L2 (`#[default] L2`, chroma.rs:28)."""

    def __init__(self, tuning: float = 440.0, f_min: float = 32.7, f_max: float = 4186.0, norm: "ChromaNorm" = None):
        if not (tuning > 0.0 and math.isfinite(tuning)):
            raise _ffi.InvalidInputError("Invalid input: tuning must be finite and > 0")
        if not (f_min > 0.0 and math.isfinite(f_min)):
            raise _ffi.InvalidInputError("Invalid input: f_min must be finite and > 0")
        if f_max <= f_min:
            raise _ffi.InvalidInputError("Invalid input: f_max must be > f_min")
        self.tuning, self.f_min, self.f_max = float(tuning), float(f_min), float(f_max)
        self.norm = norm if norm is not None else ChromaNorm.l2
        self.n_octaves = max(int(math.ceil(math.log2(f_max / f_min))), 1)

    @classmethod
    def music_standard(cls) -> "ChromaParams":
        p = cls(440.0, 32.7, 4186.0, ChromaNorm.l2)
        p.n_octaves = 7
        return p

    def with_norm(self, norm: "ChromaNorm") -> "ChromaParams":
        return ChromaParams(self.tuning, self.f_min, self.f_max, norm)

    def __repr__(self):
        return f"ChromaParams(tuning={self.tuning}, f_min={self.f_min}, f_max={self.f_max})"


class MfccParams:
    """MfccParams(n_mfcc=13) — src/mfcc.rs:20-90 (defaults include_c0=True, lifter=22; `with_c0` / `with_lifter`)."""

    def __init__(self, n_mfcc: int = 13, include_c0: bool = True, lifter: int = 22):
        if int(n_mfcc) <= 0:
            raise ValueError("n_mfcc must be > 0")
        self.n_mfcc, self.include_c0, self.lifter = int(n_mfcc), bool(include_c0), int(lifter)

    @classmethod
    def speech_standard(cls) -> "MfccParams":
        return cls(13)

    def with_c0(self, include_c0: bool) -> "MfccParams":
        return MfccParams(self.n_mfcc, include_c0, self.lifter)

    def with_lifter(self, lifter: int) -> "MfccParams":
        return MfccParams(self.n_mfcc, self.include_c0, lifter)


def parse_dtype(dtype: Optional[str]) -> int:
    """src/python/dtype.rs:34-42."""
    d = "float64" if dtype is None else dtype
    if d in ("float64", "f64", "double"):
        return _ffi.F64
    if d in ("float32", "f32", "single"):
        return _ffi.F32
    raise ValueError(f"unsupported dtype {d!r}; expected 'float32' or 'float64'")
