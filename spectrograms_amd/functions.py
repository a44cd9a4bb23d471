"""One-shot functions with the reference's names and argument order (src/python/functions.rs:92-266, 757-773)."""
from __future__ import annotations

from . import _ffi
from .params import LogParams, MelParams, SpectrogramParams
from .planner import Plan


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
