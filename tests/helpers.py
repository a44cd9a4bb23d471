"""Shared test helpers: BASELINE workload generators + independent numpy restatements."""
import numpy as np


def cfg2_signal(b: int, n: int = 160000, sr: float = 16000.0) -> np.ndarray:
    """BASELINE.md §2 config-2/3/4 generator, row b (same code as tests/golden/make_golden.py)."""
    if b % 2 == 0:
        f = 110.0 * 2.0 ** ((b % 48) / 12.0)
        i = np.arange(n, dtype=np.float64)
        return (0.5 * np.sin(2.0 * np.pi * f * i / sr)).astype(np.float32)
    rng = np.random.default_rng(1234 + b)
    return (0.1 * rng.standard_normal(n)).astype(np.float32)


def cfg2_batch(batch: int, n: int = 160000) -> np.ndarray:
    return np.stack([cfg2_signal(b, n) for b in range(batch)])


def np_frames(x, n_fft, hop, centre):
    """Zero-padded framing, semantics S1/S2 of SURVEY.md §0, written independently with numpy."""
    x = np.asarray(x)
    pad = n_fft // 2 if centre else 0
    xp = np.concatenate([np.zeros(pad, x.dtype), x, np.zeros(pad, x.dtype)])
    if xp.size < n_fft:
        xp = np.concatenate([xp, np.zeros(n_fft - xp.size, x.dtype)])
        n_frames = 1
    else:
        n_frames = (xp.size - n_fft) // hop + 1
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return xp[idx]


def np_stft(x, n_fft, hop, window, centre=True):
    """f64 complex STFT (bins, frames) via numpy pocketfft."""
    fr = np_frames(np.asarray(x, np.float64), n_fft, hop, centre) * np.asarray(window, np.float64)[None, :]
    return np.fft.rfft(fr, axis=-1).T


def slaney_hz_to_mel(f):
    f = np.asarray(f, np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def slaney_mel_to_hz(m):
    m = np.asarray(m, np.float64)
    lin = m * (200.0 / 3.0)
    log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def np_mel_filterbank(sr, n_fft, n_mels, f_min, f_max, norm=None):
    """Independent vectorised restatement of the Slaney/Hz-space triangular bank (S7)."""
    mels = np.linspace(slaney_hz_to_mel(f_min), slaney_hz_to_mel(f_max), n_mels + 2)
    hz = slaney_mel_to_hz(mels)
    bins = np.arange(n_fft // 2 + 1) * (sr / n_fft)
    lower = (bins[None, :] - hz[:-2, None]) / (hz[1:-1] - hz[:-2])[:, None]
    upper = (hz[2:, None] - bins[None, :]) / (hz[2:] - hz[1:-1])[:, None]
    fb = np.clip(np.minimum(lower, upper), 0.0, 1.0)
    fb[fb <= 1e-10] = 0.0
    if norm == "slaney":
        fb *= (2.0 / (hz[2:] - hz[:-2]))[:, None]
    elif norm == "l1":
        s = fb.sum(1, keepdims=True)
        fb = np.where(s > 0, fb / np.where(s > 0, s, 1), fb)
    elif norm == "l2":
        s = np.sqrt((fb ** 2).sum(1, keepdims=True))
        fb = np.where(s > 0, fb / np.where(s > 0, s, 1), fb)
    return fb


def rel_err(a, b):
    """max |a-b| / max |b| (scale-relative error)."""
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))
