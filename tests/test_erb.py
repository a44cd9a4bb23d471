"""ERB / gammatone frequency-domain mapping (src/erb.rs:266-401): a dense n_filters x n_bins matrix
of |H(f)|^2 applied to the power spectrum, sequential accumulation in T."""
import os

import numpy as np
import pytest

import spectrograms_amd as sg
from oracle import oracle as orc
from spectrograms_amd import _ffi
from tests import helpers as H


def np_erb(sr, n_fft, n_filters, f_min, f_max, spacing=0):
    if spacing == 0:
        e = lambda f: 24.7 * (4.37 * f / 1000.0 + 1.0)
        cf = (np.linspace(e(f_min), e(f_max), n_filters) / 24.7 - 1.0) * 1000.0 / 4.37
    else:
        shift = 9.26449 * 24.7
        i = np.arange(1, n_filters + 1)
        cf = (-shift + np.exp(i * (np.log(f_min + shift) - np.log(f_max + shift)) / n_filters) * (f_max + shift))[::-1]
    freqs = np.arange(n_fft // 2 + 1) * sr / n_fft
    bw = 1.019 * 24.7 * (4.37 * cf / 1000.0 + 1.0)
    x = (freqs[None, :] - cf[:, None]) / bw[:, None]
    return 1.0 / (1.0 + x * x) ** 4, cf


@pytest.mark.parametrize("spacing", [0, 1])
@pytest.mark.parametrize("n_fft,hop,nf,fmin,fmax", [(512, 256, 40, 0.0, 8000.0), (1024, 256, 64, 50.0, 8000.0), (400, 160, 32, 100.0, 4000.0)])
def test_oracle_erb_matches_numpy(n_fft, hop, nf, fmin, fmax, spacing):
    x = np.random.default_rng(1).standard_normal(6000)
    p = orc.Params(n_fft=n_fft, hop=hop, n_mels=nf, erb=True, erb_spacing=spacing, f_min=fmin, f_max=fmax)
    got = orc.spectrogram(p, x)
    m, cf = np_erb(16000.0, n_fft, nf, fmin, fmax, spacing)
    ref = m @ (np.abs(H.np_stft(x, n_fft, hop, np.hanning(n_fft))) ** 2)
    assert got.shape == ref.shape and H.rel_err(got, ref) < 1e-11
    assert np.allclose(orc.axes(p, 3)[0], cf, rtol=1e-12)
    assert np.all(np.diff(cf) > 0)  # low -> high in both spacings (erb.rs:221-238)


def test_host_erb_tables_and_validation():
    params = sg.SpectrogramParams(sg.StftParams(512, 256, sg.WindowType.hanning, True), 16000.0)
    for spacing in ("linear", "apple_tr35"):
        erb = sg.ErbParams(40, 0.0, 8000.0, spacing)
        pl = sg.Plan(params, _ffi.AMP_POWER, erb, None, "float64", device=_ffi.DEVICE_HOST_ONLY)
        assert pl.output_shape(16000) == (40, 63)
        ptr, col, val = pl.mel_weights()
        m, cf = np_erb(16000.0, 512, 40, 0.0, 8000.0, int(spacing != "linear"))
        assert np.array_equal(ptr, np.arange(41) * 257) and np.array_equal(col, np.tile(np.arange(257), 40))
        assert np.allclose(val.reshape(40, 257), m, rtol=1e-11)
        assert np.allclose(pl.axes(4)[0], cf, rtol=1e-12)
    assert sg.ErbParams.speech_standard().n_filters == 40 and sg.ErbParams.music_standard(44100.0).f_max == 22050.0
    with pytest.raises(sg.InvalidInputError, match="n_filters must be >= 2"):
        sg.ErbParams(1, 0.0, 100.0)
    with pytest.raises(sg.InvalidInputError, match="f_max must be > f_min"):
        sg.ErbParams(8, 100.0, 100.0)
    with pytest.raises(sg.InvalidInputError, match="exceeds Nyquist"):
        sg.Plan(params, _ffi.AMP_POWER, sg.ErbParams(10, 20.0, 9000.0), None, "float64", device=_ffi.DEVICE_HOST_ONLY)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop,amp,floor,spacing", [(1024, 256, "power", None, "linear"), (512, 256, "magnitude", None, "apple_tr35"),
                                                         (1024, 256, "db", -80.0, "linear"), (400, 160, "power", None, "linear"),
                                                         # n_fft 512 at the tuned kernel's hops: a dense bank has no band schedule, so the
                                                         # plan leaves the two-frames-per-transform mode for the generic kernel
                                                         (512, 128, "power", None, "linear"), (512, 160, "db", -80.0, "linear")])
def test_gpu_erb_matches_oracle(n_fft, hop, amp, floor, spacing, dtype):
    npdt = np.float32 if dtype == "float32" else np.float64
    x = (0.3 * np.random.default_rng(2).standard_normal((3, 7000))).astype(npdt)
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    erb = sg.ErbParams(64, 0.0, 8000.0, spacing)
    code = {"power": _ffi.AMP_POWER, "magnitude": _ffi.AMP_MAGNITUDE, "db": _ffi.AMP_DECIBELS}[amp]
    plan = sg.Plan(params, code, erb, sg.LogParams(floor) if floor is not None else None, dtype)
    got = plan.compute_batch(x)
    op = orc.Params(n_fft=n_fft, hop=hop, n_mels=64, erb=True, erb_spacing=int(spacing != "linear"), f_min=0.0, f_max=8000.0,
                    amp=amp, floor_db=floor)
    ref = orc.spectrogram_batch(op, x.astype(np.float64))
    if amp == "db":
        assert np.max(np.abs(got - ref)) < (1e-8 if dtype == "float64" else 1e-3)
    else:
        assert H.rel_err(got, ref) < (1e-10 if dtype == "float64" else 2e-5)
    s = sg.compute_erb_power_spectrogram(x[0], params, erb, dtype=dtype)
    assert s.shape == (64, got.shape[2]) and np.allclose(s.frequencies, orc.axes(op, 1)[0], rtol=1e-12)


@pytest.mark.gpu
def test_gpu_erb_full_batch_tuned_kernel():
    """config-2 input through the tuned 1024/256 f32 kernel with the dense 64 x 513 gammatone bank."""
    x = H.cfg2_batch(8)
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_POWER, sg.ErbParams(64, 0.0, 8000.0), None, "float32")
    assert plan.kernel_name.startswith("r32x16")
    got = plan.compute_batch(x)
    ref = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=256, n_mels=64, erb=True, f_min=0.0, f_max=8000.0), x.astype(np.float64))
    assert H.rel_err(got, ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("b", [0, 1])
def test_gpu_erb_matches_reference_fixture(golden_dir, b, dtype):
    """HIP against tests/golden/erb_ref.npz — outputs of the reference's own numpy_impls.erb_centers / gammatone_response /
    erb_spectrogram (python/examples/numpy_impls.py:126-159) on config-2 rows; f64 <= 1e-10, f32 <= 1e-4 relative within 40 dB of
    the peak (the dense bank runs on v_mfma_f32_16x16x4_f32 in f32)."""
    g = np.load(os.path.join(golden_dir, "erb_ref.npz"))
    nf, f_min, f_max = int(g["params"][0]), float(g["params"][1]), float(g["params"][2])
    x = H.cfg2_signal(b).astype(np.float32 if dtype == "float32" else np.float64)
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_POWER, sg.ErbParams(nf, f_min, f_max, "linear"), None, dtype)
    got = plan.compute_batch(x[None])[0].astype(np.float64)
    ref = g[f"c2_b{b}_erb_power"]
    sel = got[:, g[f"c2_b{b}_frames"]]
    if dtype == "float64":
        assert np.max(np.abs(sel - ref)) <= 1e-10 * ref.max()
    else:
        assert np.max(np.abs(sel - ref)) <= 1e-4 * ref.max()
        near = ref > 1e-4 * ref.max()
        assert np.max(np.abs(sel - ref)[near] / ref[near]) <= 1e-4
    rs = g[f"c2_b{b}_erb_rowsum"]
    assert np.max(np.abs(got.sum(axis=1) - rs)) <= (1e-10 if dtype == "float64" else 2e-5) * rs.max()
    assert np.allclose(plan.axes(4)[0], g["centres"], rtol=1e-12)
