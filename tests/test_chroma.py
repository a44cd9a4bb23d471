"""Chromagram (SURVEY.md §8f-4 "further consumer"; src/chroma.rs): magnitude spectrogram -> 12-row pitch-class bank ->
per-frame normalisation.  KATs are the reference's own (tests/chroma_tests.rs)."""
import numpy as np
import pytest

import spectrograms_amd as sg
from oracle import oracle as orc
from spectrograms_amd import _ffi
from tests import helpers as H


def sine(freq, sr=16000.0, n=16000):
    return np.sin(2.0 * np.pi * freq * np.arange(n) / sr)


def np_chroma_bank(sr, n_fft, tuning=440.0, f_min=32.7, f_max=4186.0):
    freqs = np.arange(n_fft // 2 + 1) * sr / n_fft
    fb = np.zeros((12, freqs.size))
    ok = (freqs >= f_min) & (freqs <= f_max) & (freqs > 0)
    pc = np.mod(69.0 + 12.0 * np.log2(freqs[ok] / tuning), 12.0)
    d = np.abs(pc[None, :] - np.arange(12.0)[:, None])
    fb[:, ok] = np.exp(-0.5 * np.minimum(d, 12.0 - d) ** 2)
    s = fb.sum(axis=1, keepdims=True)
    return np.where(s > 0, fb / np.where(s > 0, s, 1.0), fb)


@pytest.mark.parametrize("norm", ["none", "l1", "l2", "max"])
def test_oracle_chromagram_matches_numpy_and_reference_kats(norm):
    x = sine(440.0)
    p = orc.Params(n_fft=2048, hop=512)
    got = orc.chromagram(p, x, norm=norm)
    assert np.allclose(orc.chroma_filterbank(16000.0, 2048), np_chroma_bank(16000.0, 2048), rtol=1e-11, atol=1e-16)
    mag = np.abs(H.np_stft(x, 2048, 512, np.hanning(2048)))
    ref = np_chroma_bank(16000.0, 2048) @ mag
    if norm == "l1":
        ref = ref / ref.sum(axis=0, keepdims=True)
    elif norm == "l2":
        ref = ref / np.sqrt((ref ** 2).sum(axis=0, keepdims=True))
    elif norm == "max":
        ref = ref / ref.max(axis=0, keepdims=True)
    assert got.shape == (12, 32) and np.max(np.abs(got - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref)))
    assert np.all(np.isfinite(got)) and np.all(got >= 0.0)            # chroma_tests.rs:12-29
    assert got.sum(axis=1).argmax() == 9                               # A440 -> pitch class A (chroma_tests.rs:32-67)
    assert orc.chromagram(p, sine(261.63), norm=norm).sum(axis=1).argmax() == 0   # C4 (chroma_tests.rs:70-102)
    if norm == "l1":
        assert np.allclose(got.sum(axis=0), 1.0)
    if norm == "l2":
        assert np.allclose(np.sqrt((got ** 2).sum(axis=0)), 1.0)
    if norm == "max":
        assert np.allclose(got.max(axis=0), 1.0)


def test_chroma_params_validation_and_host_tables():
    assert sg.ChromaParams(442.0, 50.0, 8000.0, sg.ChromaNorm.l2).norm == sg.ChromaNorm.l2   # chroma_tests.rs:105-117
    for bad in ((0.0, 50.0, 8000.0), (-440.0, 50.0, 8000.0), (440.0, 1000.0, 500.0)):
        with pytest.raises(sg.InvalidInputError):
            sg.ChromaParams(*bad)
    assert sg.ChromaParams().norm == sg.ChromaNorm.l2 and sg.ChromaParams.music_standard().n_octaves == 7
    params = sg.SpectrogramParams(sg.StftParams(2048, 512, sg.WindowType.hanning, True), 16000.0)
    pl = sg.Plan(params, _ffi.AMP_MAGNITUDE, sg.ChromaParams(), None, "float64", device=_ffi.DEVICE_HOST_ONLY)
    assert pl.output_shape(16000) == (12, 32)
    ptr, col, val = pl.mel_weights()
    dense = np.zeros((12, 1025))
    for c in range(12):
        dense[c, col[ptr[c]:ptr[c + 1]]] = val[ptr[c]:ptr[c + 1]]
    assert np.allclose(dense, np_chroma_bank(16000.0, 2048), rtol=1e-11, atol=1e-16)
    with pytest.raises(sg.InvalidInputError, match="magnitude"):
        sg.Plan(params, _ffi.AMP_POWER, sg.ChromaParams(), None, "float64", device=_ffi.DEVICE_HOST_ONLY)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop,norm", [(2048, 512, "l2"), (1024, 256, "l1"), (1024, 256, "max"), (1024, 256, "none"), (400, 160, "l2"), (512, 128, "l2"),
                                            (512, 64, "max")])
def test_gpu_chromagram_matches_oracle(n_fft, hop, norm, dtype):
    rdt = np.float32 if dtype == "float32" else np.float64
    rng = np.random.default_rng(4)
    x = np.stack([sine(440.0, n=12000), sine(261.63, n=12000) + 0.1 * rng.standard_normal(12000), rng.standard_normal(12000)]).astype(rdt)
    cp = sg.ChromaParams(440.0, 32.7, 4186.0, getattr(sg.ChromaNorm, norm))
    st = sg.StftParams(n_fft, hop, sg.WindowType.hanning, True)
    plan = sg.SpectrogramPlanner().chroma_plan(st, 16000.0, cp, dtype=dtype)
    got = plan.compute_batch(x)
    ref = np.stack([orc.chromagram(orc.Params(n_fft=n_fft, hop=hop), r.astype(np.float64), norm=norm) for r in x])
    assert got.shape == ref.shape == (3, 12, ref.shape[2])
    assert np.max(np.abs(got - ref)) < (1e-10 if dtype == "float64" else 3e-5) * max(1.0, np.max(np.abs(ref)))
    c = sg.compute_chromagram(x[0], st, 16000.0, cp, dtype=dtype)
    assert isinstance(c, sg.Chromagram) and c.shape == (12, got.shape[2]) and c.labels()[9] == "A"
    assert c.data.sum(axis=1).argmax() == 9
