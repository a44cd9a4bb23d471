"""PyTorch interop: DLPack export of the result classes and the device-resident batch (spectrograms_amd/torch.py)."""
import numpy as np
import pytest
import torch

import spectrograms_amd as sg
import spectrograms_amd.torch as sgt
from oracle import oracle as orc
from spectrograms_amd import _ffi
from spectrograms_amd.planner import Spectrogram
from tests import helpers as H


def _spec(n_frames, seed=0):
    rng = np.random.default_rng(seed)
    data = rng.random((5, n_frames))
    params = sg.SpectrogramParams(sg.StftParams(8, 4, sg.WindowType.hanning, True), 16000.0)
    return Spectrogram(data, np.arange(5.0), np.arange(n_frames) * 0.1, params, None)


def test_host_results_export_dlpack_without_a_copy():
    s = _spec(7)
    t = torch.from_dlpack(s)
    assert t.shape == (5, 7) and t.dtype == torch.float64 and t.data_ptr() == s.data.ctypes.data


def test_batch_takes_resident_batches_only():
    with pytest.raises(TypeError, match="SpectrogramBatch"):
        sgt.batch([_spec(7, 1), _spec(7, 2)])


@pytest.mark.gpu
def test_resident_batch_is_zero_copy_and_matches_oracle():
    x = H.cfg2_batch(4)
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    plan = sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32")
    sb = sgt.batch_signals(plan, x)
    assert sb.shape == (4, 80, 626) and sb.__dlpack_device__()[0] == 10  # kDLROCM
    t = sgt.batch(sb, device="cuda")
    assert t.data_ptr() == sb.tensor.data_ptr() and t.is_cuda
    assert np.array_equal(t.cpu().numpy(), plan.compute_batch(x))  # same launch as the host-pointer path
    ref = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=256, n_mels=80, amp="db", floor_db=-80.0), x.astype(np.float64))
    near = ref > ref.max() - 40.0  # tolerance tiers of tests/test_gpu_parity.py: 1e-3 dB within 40 dB of the peak
    assert np.max(np.abs(t.cpu().numpy() - ref)[near]) < 1e-3
    one = sb[2]
    assert one.shape == (80, 626) and len(one.times) == 626 and one.db_range() is not None
    host, axes = sgt.batch_with_axes(sb, device="cpu")
    assert not host.is_cuda and axes.item_shape == (80, 626) and axes.times.shape == (626,) and axes.frequencies.shape == (80,)
    assert sgt.batch(sb, dtype=torch.float64).dtype == torch.float64
    # tensors on another device than the plan's are refused before they reach the kernels
    with pytest.raises(ValueError, match="CUDA tensor|plan is bound"):
        plan.compute_batch(torch.zeros((2, 4000), dtype=torch.float32))


def test_shift_helpers_and_plan_aliases():
    a = np.arange(7.0)
    assert np.array_equal(sg.fftshift_1d(a), np.fft.fftshift(a)) and np.array_equal(sg.ifftshift_1d(sg.fftshift_1d(a)), a)
    assert sg.fftshift_1d(a, dtype="float32").dtype == np.float32
    with pytest.raises(ValueError):
        sg.fftshift_1d(np.zeros((2, 2)))
    assert sg.MelDbPlan is sg.Plan and sg.LogHzPowerPlan is sg.Plan


@pytest.mark.gpu
def test_one_shot_functions_reuse_cached_plans():
    sg.clear_fft_plan_cache()
    assert sg.fft_plan_cache_info() == (0, 0)
    x = np.random.default_rng(0).standard_normal(4000)
    params = sg.SpectrogramParams(sg.StftParams(512, 128, sg.WindowType.hanning, True), 16000.0)
    a = sg.compute_mel_power_spectrogram(x, params, sg.MelParams(40, 0.0, 8000.0))
    b = sg.compute_mel_power_spectrogram(2 * x, sg.SpectrogramParams(sg.StftParams(512, 128, sg.WindowType.hanning, True), 16000.0),
                                         sg.MelParams(40, 0.0, 8000.0))
    assert sg.fft_plan_cache_info() == (1, 0) and np.allclose(b.data, 4 * a.data, rtol=1e-12)
    sg.compute_mel_power_spectrogram(x, params, sg.MelParams(40, 0.0, 8000.0), dtype="float32")  # another dtype: another plan
    S = sg.compute_stft(x, params)
    y = sg.compute_istft(S.data, 512, 128, sg.WindowType.hanning, True)
    assert sg.fft_plan_cache_info() == (3, 1) and np.max(np.abs(y[256:3500] - x[256:3500])) < 1e-10
    sg.clear_fft_plan_cache()
    assert sg.fft_plan_cache_info() == (0, 0)


def test_window_generators_and_result_members():
    """WindowType.make_* (python/spectrograms/__init__.pyi:141-190) and the resolution / DLPack members of the result classes."""
    from oracle import oracle as orc
    from spectrograms_amd.planner import Mfcc, StftResult
    for name, fn, args in (("hanning", sg.WindowType.make_hanning, ()), ("hamming", sg.WindowType.make_hamming, ()),
                           ("blackman", sg.WindowType.make_blackman, ()), ("kaiser", sg.WindowType.make_kaiser, (8.6,)),
                           ("gaussian", sg.WindowType.make_gaussian, (20.0,))):
        w = fn(64, *args)
        assert w.dtype == np.float64 and np.array_equal(w, orc.make_window(name, 64, *(args or (0.0,))))
        assert fn(64, *args, dtype="float32").dtype == np.float32
    st = sg.StftParams(512, 128, sg.WindowType.hanning, True)
    r = StftResult(np.zeros((257, 4), np.complex64), np.arange(257.0), 16000.0, st)
    assert r.frequency_resolution == 16000.0 / 512 and r.time_resolution == 128 / 16000.0
    assert torch.from_dlpack(r).shape == (257, 4)
    m = Mfcc(np.zeros((13, 4), np.float32), sg.MfccParams(13))
    assert torch.from_dlpack(m).dtype == torch.float32
    assert sg.Fft2dPlanner(dtype="float32").dtype == "float32" and sg.Fft2dPlanner().dtype == "float64"
