"""LogHz frequency mapping (src/spectrogram.rs:2438-2508, 3935-3990): sparse <=2-nnz interpolation rows
through the same mapping slot as Mel."""
import os

import numpy as np
import pytest

import spectrograms_amd as sg
from oracle import oracle as orc
from spectrograms_amd import _ffi
from tests import helpers as H


def np_loghz(sr, n_fft, n_bins, f_min, f_max):
    nb = n_fft // 2 + 1
    freqs = np.exp(np.log(f_min) + np.arange(n_bins) * (np.log(f_max) - np.log(f_min)) / (n_bins - 1))
    m = np.zeros((n_bins, nb))
    for b, f in enumerate(freqs):
        e = f / (sr / n_fft)
        lo, hi = int(np.floor(e)), min(int(np.ceil(e)), nb - 1)
        if lo >= nb:
            continue
        if lo == hi:
            m[b, lo] = 1.0
        else:
            m[b, lo] = 1.0 - (e - lo)
            m[b, hi] = e - lo
    m[np.abs(m) <= 1e-10] = 0.0
    return m, freqs


@pytest.mark.parametrize("n_fft,hop,n_bins,fmin,fmax", [(512, 256, 128, 20.0, 8000.0), (1024, 256, 64, 50.0, 7000.0), (400, 160, 40, 100.0, 4000.0)])
def test_oracle_loghz_matches_numpy(n_fft, hop, n_bins, fmin, fmax):
    x = np.random.default_rng(1).standard_normal(6000)
    p = orc.Params(n_fft=n_fft, hop=hop, n_mels=n_bins, loghz=True, f_min=fmin, f_max=fmax)
    got = orc.spectrogram(p, x)
    m, freqs = np_loghz(16000.0, n_fft, n_bins, fmin, fmax)
    ref = m @ (np.abs(H.np_stft(x, n_fft, hop, np.hanning(n_fft))) ** 2)
    assert got.shape == ref.shape and H.rel_err(got, ref) < 1e-11
    f, _ = orc.axes(p, 3)
    assert np.allclose(f, freqs, rtol=1e-13)
    assert (m != 0).sum(axis=1).max() <= 2  # src/spectrogram.rs:5385: 1-2 non-zeros per row


def test_host_loghz_tables_and_validation():
    params = sg.SpectrogramParams(sg.StftParams(512, 256, sg.WindowType.hanning, True), 16000.0)
    pl = sg.Plan(params, _ffi.AMP_POWER, sg.LogHzParams(128, 20.0, 8000.0), None, "float64", device=_ffi.DEVICE_HOST_ONLY)
    assert pl.output_shape(16000) == (128, 63)
    ptr, col, val = pl.mel_weights()
    m, freqs = np_loghz(16000.0, 512, 128, 20.0, 8000.0)
    dense = np.zeros_like(m)
    for b in range(128):
        dense[b, col[ptr[b]:ptr[b + 1]]] = val[ptr[b]:ptr[b + 1]]
    assert np.allclose(dense, m, rtol=1e-9, atol=1e-9)  # exp(log f) rounding moves the interpolation fraction by ~1e-13
    assert np.allclose(pl.axes(4)[0], freqs, rtol=1e-13)
    with pytest.raises(sg.InvalidInputError, match="f_min must be finite and > 0"):
        sg.LogHzParams(10, 0.0, 100.0)
    with pytest.raises(sg.InvalidInputError, match="Nyquist"):
        sg.Plan(params, _ffi.AMP_POWER, sg.LogHzParams(10, 20.0, 9000.0), None, "float64", device=_ffi.DEVICE_HOST_ONLY)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop,amp,floor", [(1024, 256, "power", None), (512, 256, "magnitude", None), (1024, 256, "db", -80.0), (400, 160, "power", None)])
def test_gpu_loghz_matches_oracle(n_fft, hop, amp, floor, dtype):
    npdt = np.float32 if dtype == "float32" else np.float64
    x = (0.3 * np.random.default_rng(2).standard_normal((3, 7000))).astype(npdt)
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    lp = sg.LogHzParams(96, 30.0, 7900.0)
    code = {"power": _ffi.AMP_POWER, "magnitude": _ffi.AMP_MAGNITUDE, "db": _ffi.AMP_DECIBELS}[amp]
    plan = sg.Plan(params, code, lp, sg.LogParams(floor) if floor is not None else None, dtype)
    got = plan.compute_batch(x)
    op = orc.Params(n_fft=n_fft, hop=hop, n_mels=96, loghz=True, f_min=30.0, f_max=7900.0, amp=amp, floor_db=floor)
    ref = orc.spectrogram_batch(op, x.astype(np.float64))
    if amp == "db":
        assert np.max(np.abs(got - ref)) < (1e-8 if dtype == "float64" else 1e-3)  # noise input: every bin within range
    else:
        assert H.rel_err(got, ref) < (1e-10 if dtype == "float64" else 2e-5)
    s = sg.compute_loghz_power_spectrogram(x[0], params, lp, dtype=dtype)
    assert s.shape == (96, got.shape[2]) and np.allclose(s.frequencies, orc.axes(op, 1)[0], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("b", [0, 1])
def test_gpu_loghz_matches_reference_fixture(golden_dir, b, dtype):
    """HIP against tests/golden/loghz_ref.npz — outputs of the reference's own numpy_impls.log_frequency_matrix / logfreq_spectrogram
    (python/examples/numpy_impls.py:94-123) on config-2 rows; f64 <= 1e-10, f32 <= 1e-4 relative within 40 dB of the peak."""
    g = np.load(os.path.join(golden_dir, "loghz_ref.npz"))
    n_bins, f_min, f_max = int(g["params"][0]), float(g["params"][1]), float(g["params"][2])
    x = H.cfg2_signal(b).astype(np.float32 if dtype == "float32" else np.float64)
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_POWER, sg.LogHzParams(n_bins, f_min, f_max), None, dtype)
    got = plan.compute_batch(x[None])[0].astype(np.float64)
    ref = g[f"c2_b{b}_loghz_power"]
    sel = got[:, g[f"c2_b{b}_frames"]]
    if dtype == "float64":
        assert np.max(np.abs(sel - ref)) <= 1e-10 * ref.max()
    else:
        assert np.max(np.abs(sel - ref)) <= 1e-4 * ref.max()
        near = ref > 1e-4 * ref.max()
        assert np.max(np.abs(sel - ref)[near] / ref[near]) <= 1e-4
    rs = g[f"c2_b{b}_loghz_rowsum"]
    assert np.max(np.abs(got.sum(axis=1) - rs)) <= (1e-10 if dtype == "float64" else 2e-5) * rs.max()
