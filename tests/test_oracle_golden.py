"""Pins the CPU oracle (oracle/) against (i) golden vectors produced by the reference's own
numpy_impls.py (tests/golden/*.npz, generator: tests/golden/make_golden.py) and (ii) the numeric
known-answer tests of the reference's test-suite.  CPU only."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import helpers as H


@pytest.fixture(scope="module")
def c1(golden_dir):
    return np.load(os.path.join(golden_dir, "config1_ref.npz"))


@pytest.fixture(scope="module")
def c2(golden_dir):
    return np.load(os.path.join(golden_dir, "config2_ref.npz"))


@pytest.fixture(scope="module")
def short(golden_dir):
    return np.load(os.path.join(golden_dir, "short_ref.npz"))


def sine440():
    return np.sin(2.0 * np.pi * 440.0 * np.arange(16000, dtype=np.float64) / 16000.0)


# ---------------------------------------------------------------- golden vectors (reference numpy_impls)
@pytest.mark.parametrize("n_fft,hop,shape", [(512, 256, (257, 63)), (256, 128, (129, 126))])
def test_config1_f64_matches_reference(c1, n_fft, hop, shape):
    x = sine440()
    p = orc.Params(n_fft=n_fft, hop=hop, window="hanning", centre=True, sample_rate=16000.0)
    S = orc.stft(p, x)
    assert S.shape == shape  # doc-test KAT src/spectrogram.rs:505-507
    ref = c1[f"c1_{n_fft}_{hop}_stft"]
    # f64 tolerance: 1e-10 abs on unit-scale input (the reference's own roundtrip bound, fft_backend.rs:1886-1907)
    assert np.max(np.abs(S - ref)) < 1e-10
    P = orc.spectrogram(p, x)
    assert np.max(np.abs(P - c1[f"c1_{n_fft}_{hop}_power"])) < 1e-10 * np.max(ref.real ** 2 + ref.imag ** 2)
    pm = orc.Params(n_fft=n_fft, hop=hop, amp="magnitude")
    assert np.max(np.abs(orc.spectrogram(pm, x) - c1[f"c1_{n_fft}_{hop}_magnitude"])) < 1e-10
    pn = orc.Params(n_fft=n_fft, hop=hop, centre=False)
    Pn = orc.spectrogram(pn, x)
    refn = c1[f"c1_{n_fft}_{hop}_nocentre_power"]
    assert Pn.shape == refn.shape
    assert np.max(np.abs(Pn - refn)) < 1e-10 * refn.max()
    freqs, times = orc.axes(p, shape[1])
    assert np.allclose(freqs, c1[f"c1_{n_fft}_{hop}_freqs"], rtol=0, atol=1e-9)
    assert np.allclose(times, np.arange(shape[1]) * hop / 16000.0)  # S10: no centre offset


@pytest.mark.parametrize("n", [8, 256, 512, 1024])
def test_hann_matches_reference(c1, n):
    assert np.array_equal(orc.make_window("hanning", n), c1[f"hann_{n}"]) or \
        np.max(np.abs(orc.make_window("hanning", n) - c1[f"hann_{n}"])) < 2e-16


@pytest.mark.parametrize("b", [0, 1])
def test_config2_f32_and_f64_match_reference(c2, b):
    x = H.cfg2_signal(b)
    assert np.array_equal(x[:64], c2[f"c2_b{b}_x_head"])
    assert abs(float(x.astype(np.float64).sum()) - float(c2[f"c2_b{b}_x_sum"])) < 1e-9
    p = orc.Params(n_fft=1024, hop=256)
    sub = c2[f"c2_b{b}_frames"]
    ref = c2[f"c2_b{b}_stft"]
    scale = np.max(np.abs(ref))
    S64 = orc.stft(p, x.astype(np.float64))
    assert S64.shape == (513, 626)
    assert np.max(np.abs(S64[:, sub] - ref)) < 1e-10 * max(scale, 1.0)
    # f32 path (window cast to f32, sample*window in f32, FFT in f32): 1e-4 * max|X| abs
    S32 = orc.stft(p, x)
    assert S32.dtype == np.complex64
    assert np.max(np.abs(S32[:, sub].astype(np.complex128) - ref)) < 1e-4 * scale
    P32 = orc.spectrogram(p, x).astype(np.float64)
    rs = c2[f"c2_b{b}_power_rowsum"]
    assert np.max(np.abs(P32.sum(axis=1) - rs)) < 1e-4 * rs.max()


@pytest.mark.parametrize("n", [5, 300, 511, 512, 513, 1000])
def test_short_and_ragged_inputs(short, n):
    x = short[f"short_{n}_x"]
    ref = short[f"short_{n}_stft"]
    p = orc.Params(n_fft=512, hop=256)
    S = orc.stft(p, x)
    assert S.shape == ref.shape
    assert np.max(np.abs(S - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref)))


# ---------------------------------------------------------------- reference test-suite KATs
def test_five_samples_one_frame():
    # tests/spectrogram_tests.rs:112-121
    assert orc.frame_count(5, 512, 256, True) == 1
    assert orc.frame_count(5, 512, 256, False) == 1  # padded_len < n_fft -> exactly 1 frame
    P = orc.spectrogram(orc.Params(n_fft=512, hop=256, centre=False), np.ones(5))
    assert P.shape == (257, 1)


def test_frame_count_kats():
    assert orc.frame_count(16000, 512, 256, True) == 63   # src/spectrogram.rs:505-507
    assert orc.frame_count(160000, 1024, 256, True) == 626
    assert orc.frame_count(16000, 256, 128, True) == 126
    assert orc.frame_count(1024, 1024, 1024, False) == 1
    assert orc.frame_count(2047, 1024, 1024, False) == 1
    assert orc.frame_count(2048, 1024, 1024, False) == 2


def test_dc_of_zero_padded_ones_is_three():
    # tests/fft_padding_tests.rs:149-158: [1,1,1] zero-padded to n_fft=8 -> DC bin == 3
    x = np.zeros(8)
    x[:3] = 1.0
    X = orc.rfft(x)
    assert abs(X[0] - 3.0) < 1e-12
    X32 = orc.rfft(x.astype(np.float32))
    assert abs(X32[0] - 3.0) < 1e-6


def test_dc_of_ones_is_n_and_definition():
    # src/fft_backend.rs:1880-1907 (DC of ones = N); DFT definition vs direct sum
    rng = np.random.default_rng(0)
    for n in (1, 2, 4, 8, 16, 64, 400, 1024, 3, 5, 7, 10, 100):
        x = np.ones(n)
        X = orc.rfft(x)
        assert abs(X[0] - n) < 1e-9
        assert X.size == 1 or np.max(np.abs(X[1:])) < 1e-9
        y = rng.standard_normal(n)
        k = np.arange(n // 2 + 1)[:, None]
        j = np.arange(n)[None, :]
        direct = (y[None, :] * np.exp(-2j * np.pi * k * j / n)).sum(axis=1)
        assert np.max(np.abs(orc.rfft(y) - direct)) < 1e-9 * max(1, n)
        assert np.max(np.abs(orc.rfft(y.astype(np.float32)) - direct)) < 2e-5 * max(1, np.max(np.abs(direct)))


def test_f32_tone_peaks_at_bin_128():
    # tests/f32_smoke_tests.rs:28-50: 8-sample-period tone, n_fft=1024 -> peak bin 128 +- 1
    n = 4096
    x = np.sin(2 * np.pi * np.arange(n) / 8.0).astype(np.float32)
    P = orc.spectrogram(orc.Params(n_fft=1024, hop=256), x)
    peak = int(np.argmax(P[:, P.shape[1] // 2]))
    assert abs(peak - 128) <= 1


def test_f32_mel_power_vs_f64_within_reference_bound():
    # src/spectrogram.rs:5308-5363: two-tone, 512/256, 40 mels, f32 vs f64 max-rel < 5e-3
    sr = 16000.0
    i = np.arange(16000)
    x = 0.5 * np.sin(2 * np.pi * 440 * i / sr) + 0.3 * np.sin(2 * np.pi * 1200 * i / sr)
    p = orc.Params(n_fft=512, hop=256, n_mels=40, f_min=0.0, f_max=8000.0)
    a = orc.spectrogram(p, x)
    b = orc.spectrogram(p, x.astype(np.float32)).astype(np.float64)
    m = a > 1e-6 * a.max()
    assert np.max(np.abs(a[m] - b[m]) / a[m]) < 5e-3


def test_db_floor_and_db_none_semantics():
    # tests/spectrogram_tests.rs:44-61,91-109 (all >= floor); S6: Decibels without LogParams returns power
    x = np.sin(2 * np.pi * 440 * np.arange(16000) / 16000.0)
    pdb = orc.Params(n_fft=512, hop=256, amp="db", floor_db=-80.0)
    D = orc.spectrogram(pdb, x)
    assert D.min() >= -80.0 - 1e-9
    P = orc.spectrogram(orc.Params(n_fft=512, hop=256), x)
    assert np.allclose(D, 10 * np.log10(np.maximum(P, 1e-8)), atol=1e-9)
    Dn = orc.spectrogram(orc.Params(n_fft=512, hop=256, amp="db", floor_db=None), x)
    assert np.array_equal(Dn, P)
    mdb = orc.Params(n_fft=512, hop=256, n_mels=40, amp="db", floor_db=-80.0)
    assert orc.spectrogram(mdb, x).min() >= -80.0 - 1e-9
    # Mel magnitude = sqrt(M . power), not M . magnitude (S5)
    mm = orc.spectrogram(orc.Params(n_fft=512, hop=256, n_mels=40, amp="magnitude"), x)
    mp = orc.spectrogram(orc.Params(n_fft=512, hop=256, n_mels=40), x)
    assert np.allclose(mm, np.sqrt(mp), rtol=1e-15)


# ---------------------------------------------------------------- windows (S3) vs numpy's symmetric windows
@pytest.mark.parametrize("n", [2, 8, 255, 1024])
def test_windows_match_numpy(n):
    assert np.max(np.abs(orc.make_window("hanning", n) - np.hanning(n))) < 1e-15
    assert np.max(np.abs(orc.make_window("hamming", n) - np.hamming(n))) < 1e-15
    assert np.max(np.abs(orc.make_window("blackman", n) - np.blackman(n))) < 1e-15
    assert np.array_equal(orc.make_window("rectangular", n), np.ones(n))
    # Kaiser uses the A&S polynomial I0 (src/spectrogram.rs:2237-2259): for beta <= 3.75 only the small-argument
    # branch is used and it is within ~1e-7 of the exact Bessel window
    k = orc.make_window("kaiser", n, 3.0)
    assert np.max(np.abs(k - np.kaiser(n, 3.0))) < 5e-7
    assert np.allclose(k, k[::-1], atol=1e-15)       # symmetric (tests/window_tests.rs:333-367)
    if n % 2 == 1:
        assert abs(k.max() - 1.0) < 1e-12            # peak-normalised at the centre sample
    # Reference quirk copied on purpose: the large-argument branch (:2247-2257) divides by sqrt(2*pi) although
    # A&S 9.8.2's polynomial already contains 1/sqrt(2*pi), so for beta > 3.75 I0 is low by 2.5066x there.
    k5 = orc.make_window("kaiser", n, 5.0)
    exact_edge = 1.0 / np.i0(5.0)
    assert abs(k5[0] - exact_edge * np.sqrt(2 * np.pi)) < 1e-6
    g = orc.make_window("gaussian", n, 0.4 * n)
    c = (n - 1) / 2.0
    assert np.allclose(g, np.exp(-0.5 * ((np.arange(n) - c) / (0.4 * n)) ** 2), rtol=1e-15)
    cw = np.linspace(0, 1, n)
    assert np.array_equal(orc.make_window("custom", n, custom=cw), cw)


# ---------------------------------------------------------------- Mel filterbank (S7) vs independent numpy restatement
@pytest.mark.parametrize("norm", [None, "slaney", "l1", "l2"])
@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(16000, 1024, 80, 0.0, 8000.0), (16000, 512, 40, 0.0, 8000.0),
                                                        (22050, 400, 64, 20.0, 7600.0), (48000, 2048, 128, 0.0, 24000.0)])
def test_mel_filterbank_matches_numpy(sr, n_fft, n_mels, fmin, fmax, norm):
    row_ptr, cols, vals, dense = orc.mel_filterbank(sr, n_fft, n_mels, fmin, fmax, norm)
    ref = H.np_mel_filterbank(sr, n_fft, n_mels, fmin, fmax, norm)
    assert dense.shape == ref.shape
    assert np.max(np.abs(dense - ref)) < 1e-12 * max(1.0, ref.max())
    assert np.all(np.diff(row_ptr.astype(np.int64)) >= 0)
    for m in range(n_mels):  # ascending column order within a row (accumulation order S8)
        c = cols[int(row_ptr[m]):int(row_ptr[m + 1])]
        assert np.all(np.diff(c.astype(np.int64)) > 0)


def test_mel_structure_kats():
    # 16 kHz / 1024 / 80 mels / 0-8000 Hz -> 1001 non-zeros, 4..37 per row
    row_ptr, cols, vals, dense = orc.mel_filterbank(16000, 1024, 80, 0.0, 8000.0)
    per_row = np.diff(row_ptr.astype(np.int64))
    assert int(row_ptr[-1]) == 1001
    assert per_row.min() == 4 and per_row.max() == 37
    # src/spectrogram.rs:5365-5449: > 80 % sparse, each row < bins/2 non-zeros, peak <= 1 with MelNorm::None
    assert 1.0 - 1001 / (80 * 513) > 0.8
    assert per_row.max() < 513 // 2
    assert vals.max() <= 1.0 and vals.min() > 1e-10


def test_mel_axis_ignores_fmin_fmax():
    # S10: mel axis = band centres over 0..Nyquist regardless of MelParams f_min/f_max
    p = orc.Params(n_fft=512, hop=256, n_mels=40, f_min=100.0, f_max=4000.0)
    freqs, _ = orc.axes(p, 4)
    mels = np.linspace(H.slaney_hz_to_mel(0.0), H.slaney_hz_to_mel(8000.0), 42)[1:-1]
    assert np.allclose(freqs, H.slaney_mel_to_hz(mels), rtol=1e-12)


def test_mel_spectrogram_matches_numpy_pipeline():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(8000)
    p = orc.Params(n_fft=400, hop=160, n_mels=64, f_min=0.0, f_max=8000.0, window="hamming")  # tests/mfcc_tests.rs:133 sizes
    out = orc.spectrogram(p, x)
    S = H.np_stft(x, 400, 160, np.hamming(400))
    ref = H.np_mel_filterbank(16000, 400, 64, 0.0, 8000.0) @ (np.abs(S) ** 2)
    assert out.shape == ref.shape
    assert H.rel_err(out, ref) < 1e-11


# ---------------------------------------------------------------- validation / error paths
@pytest.mark.parametrize("kw", [
    dict(n_fft=512, hop=513),                                         # hop > n_fft (spectrogram.rs:3485-3487)
    dict(n_fft=512, hop=256, sample_rate=0.0),                        # :4130
    dict(n_fft=512, hop=256, sample_rate=float("inf")),
    dict(n_fft=512, hop=256, n_mels=40, f_min=-1.0),                  # :3799
    dict(n_fft=512, hop=256, n_mels=40, f_min=100.0, f_max=50.0),     # :3803
    dict(n_fft=512, hop=256, n_mels=40, f_max=9000.0),                # tests/spectrogram_tests.rs:147-158
    dict(n_fft=512, hop=256, n_mels=10001),                           # :1696
    dict(n_fft=512, hop=256, amp="db", floor_db=float("nan")),        # :4072
    dict(n_fft=0, hop=1),
    dict(n_fft=8, hop=0),
])
def test_invalid_params_rejected(kw):
    with pytest.raises(orc.OracleError) as e:
        orc.validate(orc.Params(**kw))
    assert e.value.code == 1


def test_batch_matches_single_and_threads():
    x = H.cfg2_batch(4, 8000)
    p = orc.Params(n_fft=256, hop=64, n_mels=20, amp="db", floor_db=-80.0)
    one = np.stack([orc.spectrogram(p, r) for r in x])
    assert np.array_equal(orc.spectrogram_batch(p, x, 1), one)
    assert np.array_equal(orc.spectrogram_batch(p, x, 4), one)
    s1 = np.stack([orc.stft(p, r) for r in x])
    assert np.array_equal(orc.stft_batch(p, x, 2), s1)


# ---------------------------------------------------------------- frequency mappings pinned by the reference's numpy_impls
# (tests/golden/make_golden.py: log_frequency_matrix / logfreq_spectrogram and erb_centers / gammatone_response / erb_spectrogram,
# python/examples/numpy_impls.py:94-159 — the semantics of src/spectrogram.rs:2438-2508 and src/erb.rs:266-403, linear spacing)
@pytest.mark.parametrize("b", [0, 1])
def test_loghz_matches_reference(golden_dir, b):
    g = np.load(os.path.join(golden_dir, "loghz_ref.npz"))
    n_bins, f_min, f_max = int(g["params"][0]), float(g["params"][1]), float(g["params"][2])
    x = H.cfg2_signal(b).astype(np.float64)
    p = orc.Params(n_fft=1024, hop=256, n_mels=n_bins, loghz=True, f_min=f_min, f_max=f_max)
    got = orc.spectrogram(p, x)
    ref = g[f"c2_b{b}_loghz_power"]
    assert got.shape == (n_bins, 626)
    assert np.max(np.abs(got[:, g[f"c2_b{b}_frames"]] - ref)) <= 1e-10 * ref.max()
    rs = g[f"c2_b{b}_loghz_rowsum"]
    assert np.max(np.abs(got.sum(axis=1) - rs)) <= 1e-10 * rs.max()
    # the interpolation matrix itself, through the host plan's CSR export
    import spectrograms_amd as sg
    from spectrograms_amd import _ffi
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    pl = sg.Plan(params, _ffi.AMP_POWER, sg.LogHzParams(n_bins, f_min, f_max), None, "float64", device=_ffi.DEVICE_HOST_ONLY)
    ptr, col, val = pl.mel_weights()
    dense = np.zeros((n_bins, 513))
    for r in range(n_bins):
        dense[r, col[ptr[r]:ptr[r + 1]]] = val[ptr[r]:ptr[r + 1]]
    assert np.max(np.abs(dense - g["matrix"])) <= 1e-9  # exp(log f) rounding moves the interpolation fraction by ~1e-13


@pytest.mark.parametrize("b", [0, 1])
def test_erb_matches_reference(golden_dir, b):
    g = np.load(os.path.join(golden_dir, "erb_ref.npz"))
    nf, f_min, f_max = int(g["params"][0]), float(g["params"][1]), float(g["params"][2])
    x = H.cfg2_signal(b).astype(np.float64)
    p = orc.Params(n_fft=1024, hop=256, n_mels=nf, erb=True, erb_spacing=0, f_min=f_min, f_max=f_max)
    got = orc.spectrogram(p, x)
    ref = g[f"c2_b{b}_erb_power"]
    assert got.shape == (nf, 626)
    assert np.max(np.abs(got[:, g[f"c2_b{b}_frames"]] - ref)) <= 1e-10 * ref.max()
    rs = g[f"c2_b{b}_erb_rowsum"]
    assert np.max(np.abs(got.sum(axis=1) - rs)) <= 1e-10 * rs.max()
    assert np.allclose(orc.axes(p, 3)[0], g["centres"], rtol=1e-12)
    import spectrograms_amd as sg
    from spectrograms_amd import _ffi
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    pl = sg.Plan(params, _ffi.AMP_POWER, sg.ErbParams(nf, f_min, f_max, "linear"), None, "float64", device=_ffi.DEVICE_HOST_ONLY)
    _, _, val = pl.mel_weights()
    assert np.allclose(val.reshape(nf, 513)[:, g["matrix_cols"]], g["matrix"], rtol=1e-11)
    assert np.allclose(pl.axes(4)[0], g["centres"], rtol=1e-12)


@pytest.mark.parametrize("n", [2048 + 1, 9001, 16385, 20000, 44100, 65536, 100003])
def test_oracle_long_lengths_pinned_against_numpy_fft(n):
    """From 2048 points on the oracle's non-power-of-two transforms are chirp-z in f64 (oracle/spectro_oracle.c, ORC_CZT_MIN) instead
    of the O(n^2) definition; pinned here against numpy.fft (pocketfft) — forward, inverse and the complex transform the 2-D path uses."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    ref = np.fft.rfft(x)
    X = orc.rfft(x)
    assert np.max(np.abs(X - ref)) < 1e-12 * np.max(np.abs(ref))
    assert np.max(np.abs(orc.irfft(ref, n) - x)) < 1e-12
    X32 = orc.rfft(x.astype(np.float32))  # non-powers of two: the f32 oracle accumulates in f64 and rounds once, like its direct sum did; powers of two: radix-2 in f32
    tol32 = 5e-6 if n & (n - 1) == 0 else 2e-7
    assert X32.dtype == np.complex64 and np.max(np.abs(X32 - np.fft.rfft(x.astype(np.float32).astype(np.float64)))) < tol32 * np.max(np.abs(ref))
