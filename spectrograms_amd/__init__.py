"""spectrograms_amd — MI355X (gfx950) engine for the batched STFT / Mel-spectrogram hot path of the
`spectrograms` crate, behind the reference's own parameter / plan / function names.

The compute path is hand-written HIP in libspectro_hip.so reached through the C ABI of
include/spectro_hip.h.  There is no CPU fallback: if the library is not built, or no HIP device is present,
compute calls raise FFTBackendError.
"""
from ._ffi import DimensionMismatchError, FFTBackendError, InternalError, InvalidInputError, SpectrogramError
from .fft2d import (C2cPlan, Fft2dPlan, Fft2dPlanner, bandpass_filter, convolve_fft, detect_edges_fft, fft2d, fftfreq, fftshift,
                    fftshift_1d, gaussian_kernel_2d, highpass_filter, ifft2d, ifftshift, ifftshift_1d, lowpass_filter,
                    magnitude_spectrum_2d, power_spectrum_2d, rfftfreq, sharpen_fft)
from .functions import (clear_fft_plan_cache, compute_chromagram, compute_erb_db_spectrogram,
                        compute_erb_magnitude_spectrogram, compute_erb_power_spectrogram, compute_fft, compute_irfft,
                        compute_istft, compute_linear_db_spectrogram, compute_linear_magnitude_spectrogram,
                        compute_linear_power_spectrogram, compute_loghz_db_spectrogram,
                        compute_loghz_magnitude_spectrogram, compute_loghz_power_spectrogram,
                        compute_magnitude_spectrum, compute_mel_db_spectrogram, compute_mel_magnitude_spectrogram,
                        compute_mel_power_spectrogram, compute_mfcc, compute_power_spectrum, compute_rfft, compute_stft,
                        fft_plan_cache_info)
from .params import (ChromaNorm, ChromaParams, ErbParams, GammatoneParams, LogHzParams, LogParams, MelNorm, MelParams,
                     MfccParams, SpectrogramParams, StftParams, WindowType)
from .planner import Chromagram, Mfcc, Plan, Spectrogram, SpectrogramBatch, SpectrogramPlanner, StftResult

# The reference exposes one plan class per (frequency scale, amplitude scale) (python/spectrograms/__init__.pyi:1087-1245);
# here they are one class parameterised at construction, exported under the same names.
LinearPowerPlan = LinearMagnitudePlan = LinearDbPlan = Plan
MelPowerPlan = MelMagnitudePlan = MelDbPlan = Plan
ErbPowerPlan = ErbMagnitudePlan = ErbDbPlan = Plan
LogHzPowerPlan = LogHzMagnitudePlan = LogHzDbPlan = Plan

__all__ = [
    "SpectrogramError", "InvalidInputError", "DimensionMismatchError", "FFTBackendError", "InternalError",
    "WindowType", "StftParams", "SpectrogramParams", "MelParams", "MelNorm", "LogParams", "SpectrogramPlanner",
    "Plan", "Spectrogram", "StftResult", "compute_linear_power_spectrogram", "compute_linear_magnitude_spectrogram",
    "compute_linear_db_spectrogram", "compute_mel_power_spectrogram", "compute_mel_magnitude_spectrogram",
    "compute_mel_db_spectrogram", "compute_stft", "compute_mfcc", "MfccParams", "Mfcc", "clear_fft_plan_cache",
    "fft_plan_cache_info", "compute_fft", "compute_rfft", "compute_power_spectrum", "compute_magnitude_spectrum",
    "compute_chromagram", "ChromaParams", "ChromaNorm", "Chromagram", "SpectrogramBatch", "compute_irfft",
    "compute_istft", "ErbParams", "GammatoneParams", "compute_erb_power_spectrogram",
    "compute_erb_magnitude_spectrogram", "compute_erb_db_spectrogram", "LogHzParams",
    "compute_loghz_power_spectrogram", "compute_loghz_magnitude_spectrogram", "compute_loghz_db_spectrogram",
    "fft2d", "ifft2d", "C2cPlan", "Fft2dPlan", "Fft2dPlanner", "convolve_fft", "gaussian_kernel_2d", "lowpass_filter",
    "highpass_filter", "bandpass_filter", "detect_edges_fft", "sharpen_fft", "power_spectrum_2d",
    "magnitude_spectrum_2d", "fftshift", "ifftshift", "fftfreq", "rfftfreq", "fftshift_1d", "ifftshift_1d",
    "LinearPowerPlan", "LinearMagnitudePlan", "LinearDbPlan", "MelPowerPlan", "MelMagnitudePlan", "MelDbPlan",
    "ErbPowerPlan", "ErbMagnitudePlan", "ErbDbPlan", "LogHzPowerPlan", "LogHzMagnitudePlan", "LogHzDbPlan",
]
