"""Inverse 1-D path: irfft / C2rPlan::process (src/fft_backend.rs:526-565, src/spectrogram.rs:4789-4811)
and batched istft (src/spectrogram.rs:4860-4946).  CPU tests pin the oracle; GPU tests compare the HIP path with it."""
import os

import numpy as np
import pytest

import spectrograms_amd as sg
from oracle import oracle as orc
from spectrograms_amd import _ffi
from tests import helpers as H


def np_istft(S, n_fft, hop, w, centre):
    nb, nf = S.shape
    out_len = (nf - 1) * hop + n_fft
    acc, nrm = np.zeros(out_len), np.zeros(out_len)
    for f in range(nf):
        col = S[:, f].copy()
        col[0] = col[0].real
        if n_fft % 2 == 0:
            col[-1] = col[-1].real
        acc[f * hop:f * hop + n_fft] += np.fft.irfft(col, n_fft) * w
        nrm[f * hop:f * hop + n_fft] += w * w
    ok = nrm > 1e-10
    acc[ok] /= nrm[ok]
    pad = n_fft // 2 if centre else 0
    unp = max(out_len - 2 * pad, 0)
    return acc[pad:pad + unp] if centre and unp > 0 else acc


CASES = [(512, 128, True, "hanning"), (512, 256, True, "hanning"), (400, 100, False, "hamming"), (1024, 256, True, "hanning"),
         (400, 160, True, "blackman"), (256, 256, False, "rectangular"), (8, 3, True, "hanning"), (15, 4, True, "hamming"),
         # lengths without a pass split: chirp-z rows (odd: the half spectrum has no Nyquist bin)
         (251, 62, True, "hanning"), (1009, 252, True, "hamming"), (1023, 256, False, "hamming"), (1006, 300, True, "blackman"),
         # short 2 x prime lengths the forward cost model leaves on the direct / two-factor kernels: their inverse rows take the LDS-tile
         # rows (launch_c2r_rows), not half-length chirp-z tables
         (34, 9, True, "hanning"), (62, 16, False, "hamming"), (46, 46, True, "rectangular"),
         # n_fft 2048: the fused tuned kernel k_istft2048 from hop 128 (round 4), the register-tiled rows below
         (2048, 512, True, "hanning"), (2048, 256, False, "hamming"), (2048, 300, True, "blackman"), (2048, 128, True, "hanning"),
         (2048, 1024, True, "hamming"), (2048, 2048, False, "rectangular"), (2048, 100, True, "hanning")]


@pytest.mark.parametrize("n_fft,hop,centre,window", CASES)
def test_oracle_istft_matches_numpy(n_fft, hop, centre, window):
    x = np.random.default_rng(3).standard_normal(3000)
    S = orc.stft(orc.Params(n_fft=n_fft, hop=hop, centre=centre, window=window), x)
    got = orc.istft(S, n_fft, hop, window, centre)
    ref = np_istft(S, n_fft, hop, orc.make_window(window, n_fft), centre)
    assert got.shape == ref.shape and np.max(np.abs(got - ref)) < 1e-11
    assert len(got) == orc.istft_length(S.shape[1], n_fft, hop, centre)


def test_oracle_istft_roundtrip_and_edge_lengths():
    x = np.sin(0.02 * np.arange(2048))  # tests/f32_smoke_tests.rs:66-76
    for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-6)):
        S = orc.stft(orc.Params(n_fft=256, hop=128), x.astype(dt))
        y = orc.istft(S, 256, 128, "hanning", True)
        assert y.dtype == dt and np.all(np.isfinite(y))
        n = min(len(y), len(x))
        assert np.max(np.abs(y[128:n - 128] - x[128:n - 128])) < tol
    # one centred frame of even n_fft: the trimmed length would be 0, so the untrimmed n_fft samples come back (:4933)
    assert orc.istft_length(1, 512, 128, True) == 512 and orc.istft_length(1, 15, 4, True) == 1
    assert orc.istft_length(3, 512, 128, True) == 256 and orc.istft_length(3, 512, 128, False) == 768
    with pytest.raises(orc.OracleError):  # n_bins mismatch (:4876-4879)
        orc.istft(np.zeros((100, 4), np.complex128), 512, 128)
    with pytest.raises(orc.OracleError):  # tests/fft_padding_tests.rs:133-138
        orc.irfft(np.ones(4, np.complex128), 8)
    with pytest.raises(orc.OracleError):  # realfft rejects a complex DC bin
        orc.irfft(np.array([1 + 1j, 0, 0, 0, 0]), 8)
    z = np.random.default_rng(0).standard_normal(100)
    assert np.max(np.abs(orc.irfft(np.fft.rfft(z), 100) - z)) < 1e-12


def test_host_istft_validation():
    params = sg.SpectrogramParams(sg.StftParams(512, 128, sg.WindowType.hanning, True), 16000.0)
    pl = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float64", device=_ffi.DEVICE_HOST_ONLY)
    assert pl.istft_length(3) == 256 and pl.istft_length(1) == 512
    with pytest.raises(sg.DimensionMismatchError, match="expected 257, got 100"):
        pl.istft_batch(np.zeros((1, 100, 4), np.complex128))
    with pytest.raises(sg.DimensionMismatchError):
        pl.istft_batch(np.zeros((1, 257, 4), np.complex128), out=np.zeros((1, 5)))
    with pytest.raises(sg.SpectrogramError, match="no HIP device"):  # no CPU fallback
        pl.istft_batch(np.zeros((1, 257, 4), np.complex128))
    with pytest.raises(sg.DimensionMismatchError, match="expected 5, got 4"):
        pl8 = sg.Plan(sg.SpectrogramParams(sg.StftParams(8, 8, sg.WindowType.rectangular, False), 1.0), _ffi.AMP_COMPLEX, None, None,
                      "float64", device=_ffi.DEVICE_HOST_ONLY)
        pl8.c2r(np.ones(4, np.complex128))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop,centre,window", CASES + [(4096, 1024, True, "hanning"), (8192, 2048, True, "hanning"),
                                                         # even lengths whose own chirp-z does not fit LDS in f64 (6000) / in either type (8200):
                                                         # half-length complex form on the chirp-z kernel where ITS convolution fits
                                                         # (8200 in f64: past every on-chip tile — the global-memory chirp-z of bigfft.hip, like 12000 / 9001)
                                                         (6000, 1500, True, "hanning"), (8200, 2050, True, "hamming"), (12000, 3000, True, "hanning"),
                                                         (9001, 4500, False, "hamming")])
def test_gpu_istft_matches_oracle(n_fft, hop, centre, window, dtype):
    rdt, cdt = (np.float32, np.complex64) if dtype == "float32" else (np.float64, np.complex128)
    n = max(3000, 3 * n_fft) if n_fft < 6000 else n_fft + 2 * hop  # (the oracle's non-power-of-two transforms are O(n^2))
    x = np.random.default_rng(5).standard_normal((3 if n_fft < 6000 else 2, n)).astype(rdt)
    wt = getattr(sg.WindowType, window)
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, wt, centre), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, dtype)
    S = np.stack([orc.stft(orc.Params(n_fft=n_fft, hop=hop, centre=centre, window=window), r) for r in x]).astype(cdt)
    got = plan.istft_batch(S)
    ref = np.stack([orc.istft(s, n_fft, hop, window, centre) for s in S])
    assert got.dtype == rdt and got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < (1e-11 if dtype == "float64" else 2e-5 * max(1.0, np.max(np.abs(ref))))
    one = sg.compute_istft(S[1], n_fft, hop, wt, centre, dtype=dtype)
    assert np.array_equal(one, got[1])


@pytest.mark.gpu
@pytest.mark.parametrize("hop,centre,window,n", [(64, True, "hanning", 9000), (93, True, "hanning", 9000), (94, False, "hanning", 9000),
                                                 (100, True, "hamming", 9000), (128, True, "hanning", 20000), (256, False, "hanning", 20000),
                                                 (300, True, "blackman", 20000), (512, True, "hanning", 20000),
                                                 (1024, True, "rectangular", 20000), (256, True, "hanning", 700)])
def test_gpu_istft_fused_1024_kernel(hop, centre, window, n):
    """f32, n_fft = 1024: the fused tuned kernel (halo frames recomputed per tile) for hops that do and do not divide n_fft,
    with tile edges, a single-tile signal and the untrimmed / trimmed output windows; below hop 94 (more than 10 halo frames of
    16) the plan takes the register-tiled rows + overlap-add instead (hop 64 here)."""
    x = np.random.default_rng(11).standard_normal((5, n)).astype(np.float32)
    wt = getattr(sg.WindowType, window)
    plan = sg.Plan(sg.SpectrogramParams(sg.StftParams(1024, hop, wt, centre), 16000.0), _ffi.AMP_COMPLEX, None, None, "float32")
    S = np.stack([orc.stft(orc.Params(n_fft=1024, hop=hop, centre=centre, window=window), r) for r in x])
    got = plan.istft_batch(S)
    ref = np.stack([orc.istft(s, 1024, hop, window, centre) for s in S])
    assert got.shape == ref.shape
    # out = sum(y w) / sum(w w): where the window sum is tiny (the first / last samples of an uncentred Hann frame) the
    # division amplifies the f32 rounding of y by 1/w in ANY implementation, so the error is weighed by sqrt(norm) there
    w = orc.make_window(window, 1024)
    nf = S.shape[2]
    nrm = np.zeros((nf - 1) * hop + 1024)
    for f in range(nf):
        nrm[f * hop:f * hop + 1024] += w * w
    nrm = nrm[512:512 + ref.shape[1]] if centre and ref.shape[1] != nrm.size else nrm
    scale = np.minimum(1.0, np.sqrt(nrm))[None, :]
    assert np.max(np.abs(got - ref) * scale) < 2e-5 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,hop,centre,window,n", [(512, 128, True, "hanning", 20000), (512, 100, True, "hamming", 20000),
                                                       (512, 171, False, "hanning", 20000), (512, 512, True, "rectangular", 9000),
                                                       (400, 160, True, "hanning", 30000), (400, 133, False, "blackman", 9000),
                                                       (256, 64, True, "hanning", 40000), (128, 32, True, "hanning", 300),
                                                       (512, 64, True, "hanning", 9000), (512, 63, False, "hamming", 9000),
                                                       (400, 50, True, "hanning", 9000), (512, 32, True, "hanning", 5000),
                                                       (256, 255, False, "hanning", 5000), (256, 256, True, "hamming", 5000),
                                                       (1000, 200, True, "hanning", 30000)])
def test_gpu_istft_fused_register_tiled_kernel(n_fft, hop, centre, window, n):
    """f32, lengths with a pass split: the fused register-tiled kernel (windowed frames kept in LDS, halo frames recomputed per
    tile) over many tiles, hops that do not divide n_fft (and do or do not divide the workgroup's 256 threads: the overlap-add
    walks the tile's positions with all threads), a signal shorter than one tile, both output windows; up to half a tile of halo
    frames is fused (hop = n_fft / 8: 7 of 16), hop = n_fft / 16 takes the frame-scratch path."""
    x = np.random.default_rng(12).standard_normal((4, n)).astype(np.float32)
    wt = getattr(sg.WindowType, window)
    plan = sg.Plan(sg.SpectrogramParams(sg.StftParams(n_fft, hop, wt, centre), 16000.0), _ffi.AMP_COMPLEX, None, None, "float32")
    S = np.stack([orc.stft(orc.Params(n_fft=n_fft, hop=hop, centre=centre, window=window), r) for r in x])
    got = plan.istft_batch(S)
    ref = np.stack([orc.istft(s, n_fft, hop, window, centre) for s in S])
    assert got.shape == ref.shape
    w = orc.make_window(window, n_fft)
    nf = S.shape[2]
    nrm = np.zeros((nf - 1) * hop + n_fft)
    for f in range(nf):
        nrm[f * hop:f * hop + n_fft] += w * w
    nrm = nrm[n_fft // 2:n_fft // 2 + ref.shape[1]] if centre and ref.shape[1] != nrm.size else nrm
    scale = np.minimum(1.0, np.sqrt(nrm))[None, :]  # see test_gpu_istft_fused_1024_kernel
    assert np.max(np.abs(got - ref) * scale) < 2e-5 * max(1.0, np.max(np.abs(ref)))
    assert np.array_equal(plan.istft_batch(S[2:3])[0], got[2])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("float32", 3e-6), ("float64", 1e-12)])
def test_gpu_stft_istft_roundtrip_device(dtype, tol):
    """forward on the GPU, inverse on the GPU, device-resident end to end (config-2 shaped rows)."""
    import torch
    x = H.cfg2_batch(4).astype(np.float32 if dtype == "float32" else np.float64)
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, dtype)
    xd = torch.from_numpy(x).cuda()
    S = plan.compute_batch(xd)
    Sc = torch.view_as_complex(S.reshape(S.shape[0], 513, -1, 2)) if not S.is_complex() else S
    y = plan.istft_batch(Sc.contiguous())
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    n = min(y.shape[1], x.shape[1])
    assert y.shape[1] == plan.istft_length(Sc.shape[2])
    assert np.max(np.abs(y[:, 512:n - 512] - x[:, 512:n - 512])) < tol


@pytest.mark.gpu
def test_gpu_irfft_and_dc_check():
    z = np.random.default_rng(0).standard_normal(2048)
    spec = np.fft.rfft(z)
    o64 = sg.compute_irfft(spec, 2048, dtype="float64")  # python/tests/test_dtype_ops.py:81-88
    o32 = sg.compute_irfft(spec, 2048, dtype="float32")
    assert o64.dtype == np.float64 and o32.dtype == np.float32
    np.testing.assert_allclose(o64, z, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(o32, z, rtol=1e-3, atol=1e-3)
    assert np.max(np.abs(o64 - orc.irfft(spec, 2048))) < 1e-12
    z100 = z[:100]
    assert np.max(np.abs(sg.compute_irfft(np.fft.rfft(z100), 100) - z100)) < 1e-12
    with pytest.raises(sg.DimensionMismatchError):  # tests/fft_padding_tests.rs:133-138
        sg.compute_irfft(np.ones(4, np.complex128), 8)
    with pytest.raises(sg.SpectrogramError, match="DC or Nyquist"):
        sg.compute_irfft(np.array([1 + 1j, 0, 0, 0, 0]), 8)
    with pytest.raises(sg.SpectrogramError, match="DC or Nyquist"):
        bad = np.zeros((257, 4), np.complex128)
        bad[256, 2] = 1j
        sg.compute_istft(bad, 512, 128, sg.WindowType.hanning, True)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("float32", 2e-5), ("float64", 1e-10)])
def test_gpu_single_frame_helpers(dtype, tol):
    """fft / rfft / power_spectrum / magnitude_spectrum (src/spectrogram.rs:4475-4693): zero padding to n_fft, optional window;
    known answers from tests/fft_padding_tests.rs."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal(300)
    for n_fft in (512, 400, 300):
        ref = np.fft.rfft(np.concatenate([x, np.zeros(n_fft - 300)]))
        got = sg.compute_fft(x, n_fft, dtype=dtype)
        assert got.shape == (n_fft // 2 + 1,) and got.dtype == (np.complex64 if dtype == "float32" else np.complex128)
        assert np.max(np.abs(got - ref)) < tol * np.max(np.abs(ref))
        assert np.max(np.abs(sg.compute_rfft(x, n_fft, dtype=dtype) - np.abs(ref))) < tol * np.max(np.abs(ref))
        w = orc.make_window("hanning", n_fft)
        refw = np.abs(np.fft.rfft(np.concatenate([x, np.zeros(n_fft - 300)]) * w)) ** 2
        p = sg.compute_power_spectrum(x, n_fft, sg.WindowType.hanning, dtype=dtype)
        assert np.max(np.abs(p - refw)) < 2 * tol * np.max(refw)
        m = sg.compute_magnitude_spectrum(x, n_fft, sg.WindowType.hanning, dtype=dtype)
        assert np.max(np.abs(m - np.sqrt(refw))) < 2 * tol * np.max(np.sqrt(refw))
    assert np.size(sg.compute_fft(x, dtype=dtype)) == 151  # n_fft defaults to len(samples)
    dc = sg.compute_fft(np.array([1.0, 1.0, 1.0]), 8, dtype=dtype)  # tests/fft_padding_tests.rs:149-158: DC of zero-padded ones = 3
    assert abs(dc[0] - 3.0) < 1e-6 and sg.compute_magnitude_spectrum(np.array([1.0, 2.0, 3.0]), 8, sg.WindowType.hanning).shape == (5,)
    with pytest.raises(sg.InvalidInputError, match="exceeds FFT size"):
        sg.compute_fft(np.zeros(10), 8)


@pytest.mark.gpu
def test_gpu_istft_fuzz_shapes():
    """Seeded random sweep of the inverse path over every row kernel (fused n_fft = 1024, register-tiled power-of-two and mixed
    radix, LDS radix-2, direct) against the oracle: hops that do not divide n_fft, short inputs, both dtypes."""
    rng = np.random.default_rng(int(os.environ.get("SGX_FUZZ_SEED", 77)))
    pool = [8, 16, 32, 64, 128, 256, 512, 1024, 2048, 80, 160, 200, 240, 320, 400, 480, 640, 800, 960, 1200, 30, 100, 441, 97]
    wins = ["hanning", "hamming", "blackman", "rectangular"]
    for case in range(40):
        n_fft = int(pool[rng.integers(len(pool))])
        hop = int(rng.integers(max(1, n_fft // 8), n_fft + 1))
        centre = bool(rng.integers(2))
        window = wins[rng.integers(len(wins))]
        dtype = ["float32", "float64"][rng.integers(2)]
        rdt, cdt = (np.float32, np.complex64) if dtype == "float32" else (np.float64, np.complex128)
        n = int(rng.integers(n_fft, 8 * n_fft + 100))
        x = rng.standard_normal((int(rng.integers(1, 4)), n)).astype(rdt)
        params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, getattr(sg.WindowType, window), centre), 16000.0)
        plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, dtype)
        S = np.stack([orc.stft(orc.Params(n_fft=n_fft, hop=hop, centre=centre, window=window), r) for r in x]).astype(cdt)
        got = plan.istft_batch(S)
        ref = np.stack([orc.istft(s, n_fft, hop, window, centre) for s in S])
        assert got.shape == ref.shape, (n_fft, hop, centre, window, dtype)
        # the division by sum(w^2) is ill-conditioned where the windows barely overlap (hop close to n_fft with a tapered
        # window): weigh the error by min(1, sum w^2) there
        w = orc.make_window(window, n_fft)
        nf = S.shape[2]
        nrm = np.zeros((nf - 1) * hop + n_fft)
        for f in range(nf):
            nrm[f * hop:f * hop + n_fft] += w * w
        pad = n_fft // 2 if centre else 0
        nrm = nrm[pad:pad + ref.shape[1]] if ref.shape[1] != len(nrm) else nrm
        wgt = np.minimum(1.0, nrm)[None, :]
        tol = 1e-10 if dtype == "float64" else 3e-5
        assert np.max(np.abs(got - ref) * wgt) < tol * max(1.0, np.max(np.abs(ref) * wgt)), (n_fft, hop, centre, window, dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("hop,centre,n,batch", [(512, True, 100000, 5), (512, False, 33333, 3), (640, True, 70001, 2), (128, True, 20000, 2),
                                               (2048, True, 50000, 4), (512, True, 2047, 2), (512, True, 1, 1)])
def test_gpu_istft_2048_long_signals(hop, centre, n, batch):
    """k_istft2048 over many tiles per signal (runs that start inside a signal build their carry from the tile in front), frame
    counts that are not multiples of 16, signals shorter than a frame; round trip and the oracle."""
    rng = np.random.default_rng(11)
    if not centre and n < 2048:
        n = 2048 + n
    x = rng.standard_normal((batch, n)).astype(np.float32)
    # without centring (or with hop = n_fft) a Hann envelope falls to 5e-12 at the frame ends and the division amplifies f32
    # rounding there: Hamming then
    wname = "hanning" if centre and hop < 2048 else "hamming"
    params = sg.SpectrogramParams(sg.StftParams(2048, hop, getattr(sg.WindowType, wname), centre), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
    S = plan.compute_batch(x)
    y = plan.istft_batch(np.ascontiguousarray(S))
    ref = np.stack([orc.istft(s.astype(np.complex128), 2048, hop, wname, centre) for s in np.asarray(S)])
    assert y.shape == ref.shape
    assert np.max(np.abs(y - ref)) < 2e-5 * max(1.0, np.max(np.abs(ref)))
    one = plan.istft_batch(np.ascontiguousarray(np.asarray(S)[batch - 1:]))
    assert np.array_equal(one[0], y[batch - 1])


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,hop,n,batch", [(1024, 1024, 300000, 2), (2048, 2048, 400000, 2), (1024, 256, 300000, 3), (2048, 512, 400000, 1),
                                               (1024, 512, 40000, 40), (2048, 1024, 160000, 64)])
def test_gpu_istft_run_cuts(n_fft, hop, n, batch):
    """Few long signals: the launcher cuts them into many short runs (istft_carry_runs); hop = n_fft has no carry and no warm-up tile.
    Whatever the cut, a signal's samples are the same bits as when it is inverted alone (one run per signal at batch 1 ... many)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((batch, n)).astype(np.float32)
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hamming, True), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
    S = np.ascontiguousarray(plan.compute_batch(x))
    y = plan.istft_batch(S)
    ref = orc.istft(S[0].astype(np.complex128), n_fft, hop, "hamming", True)
    assert np.max(np.abs(y[0] - ref)) < 2e-5 * max(1.0, np.max(np.abs(ref)))
    m = min(y.shape[1], n)
    assert np.max(np.abs(y[:, :m] - x[:, :m])) < 1e-4
    big = plan.istft_batch(np.ascontiguousarray(np.concatenate([S] * 3)[: 3 * batch - 1]))  # another batch size: another cut
    assert np.array_equal(big[:batch], y) and np.array_equal(big[batch:2 * batch], y)


@pytest.mark.gpu
def test_gpu_istft_fuzz_tuned_shapes():
    """Random hops, lengths and batch sizes through the fused inverse kernels (n_fft 1024 from hop 64, 2048 from hop 128: every
    run cut istft_carry_runs can produce, general and compile-time overlap-add) and the register-tiled fallback below those hops."""
    rng = np.random.default_rng(int(os.environ.get("SGX_FUZZ_SEED", 2024)))
    for it in range(40):
        n_fft = int(rng.choice([1024, 2048]))
        hop = int(rng.choice([n_fft // 8, n_fft // 4, n_fft // 2, n_fft, int(rng.integers(40, n_fft + 1))]))
        centre = bool(rng.integers(0, 2))
        batch = int(rng.choice([1, 2, 3, 7, 33, 130]))
        n = int(rng.integers(n_fft, 40 * n_fft)) if batch > 7 else int(rng.integers(n_fft, 400 * n_fft))
        x = rng.standard_normal((batch, n)).astype(np.float32)
        params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hamming, centre), 16000.0)
        plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
        S = np.ascontiguousarray(plan.compute_batch(x))
        y = plan.istft_batch(S)
        for b in sorted({0, batch - 1, batch // 2}):
            ref = orc.istft(S[b].astype(np.complex128), n_fft, hop, "hamming", centre)
            assert y[b].shape == ref.shape, (n_fft, hop, centre, batch, n)
            assert np.max(np.abs(y[b] - ref)) < 2e-5 * max(1.0, np.max(np.abs(ref))), (it, n_fft, hop, centre, batch, n, b)


@pytest.mark.gpu
@pytest.mark.parametrize("hop,centre,n,batch", [(256, True, 100000, 5), (256, False, 33333, 3), (512, True, 70001, 2), (64, True, 20000, 2), (100, True, 20000, 2),
                                               (1024, True, 50000, 4), (300, True, 1023, 2), (256, True, 1, 1), (128, False, 9000, 7), (256, True, 160000, 64)])
def test_gpu_istft_f64_1024_tuned(hop, centre, n, batch):
    """k_istft_d1024 (f64 n_fft 1024, hop >= 64): lane-pair fold and trade, carried overlap-add in f64 with the compile-time form at hops 256 / 512 /
    1024, runs that start inside a signal, ragged frame counts, signals shorter than a frame; the oracle and bit-equality of a signal alone."""
    rng = np.random.default_rng(17)
    if not centre and n < 1024:
        n = 1024 + n
    x = rng.standard_normal((batch, n))
    wname = "hanning" if centre and hop < 1024 else "hamming"  # (see test_gpu_istft_2048_long_signals)
    params = sg.SpectrogramParams(sg.StftParams(1024, hop, getattr(sg.WindowType, wname), centre), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float64")
    S = np.ascontiguousarray(plan.compute_batch(x))
    y = plan.istft_batch(S)
    for b in sorted({0, batch - 1, batch // 2}):
        ref = orc.istft(S[b], 1024, hop, wname, centre)
        assert y[b].shape == ref.shape
        assert np.max(np.abs(y[b] - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref))), (hop, centre, n, batch, b)
    m = min(y.shape[1], n)
    assert np.max(np.abs(y[:, 600:m - 600] - x[:, 600:m - 600])) < 1e-9 if m > 1300 else True
    one = plan.istft_batch(np.ascontiguousarray(S[batch - 1:]))
    assert np.array_equal(one[0], y[batch - 1])


@pytest.mark.gpu
@pytest.mark.parametrize("hop,centre,n,batch", [(160, True, 100000, 5), (128, False, 33333, 3), (256, True, 70001, 2), (32, True, 9000, 2), (100, True, 20000, 2),
                                               (512, True, 50000, 4), (300, True, 511, 2), (160, True, 1, 1), (64, False, 9000, 7), (160, True, 160000, 64),
                                               (128, True, 160000, 64), (256, True, 40000, 130)])
def test_gpu_istft_f64_512_tuned(hop, centre, n, batch):
    """k_istft_d512 (f64 n_fft 512 — the reference's speech default 512 / 160 in its default type —, two frames per transform, hop >= 32): odd and even
    frame counts (a slot's second frame may not exist), the compile-time overlap-add at hops 128 / 256 / 512 and the general walk, runs that start
    inside a signal; the oracle and bit-equality of a signal alone."""
    rng = np.random.default_rng(23)
    if not centre and n < 512:
        n = 512 + n
    x = rng.standard_normal((batch, n))
    wname = "hanning" if centre and hop < 512 else "hamming"  # (see test_gpu_istft_2048_long_signals)
    params = sg.SpectrogramParams(sg.StftParams(512, hop, getattr(sg.WindowType, wname), centre), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float64")
    S = np.ascontiguousarray(plan.compute_batch(x))
    y = plan.istft_batch(S)
    for b in sorted({0, batch - 1, batch // 2}):
        ref = orc.istft(S[b], 512, hop, wname, centre)
        assert y[b].shape == ref.shape
        assert np.max(np.abs(y[b] - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref))), (hop, centre, n, batch, b)
    one = plan.istft_batch(np.ascontiguousarray(S[batch - 1:]))
    assert np.array_equal(one[0], y[batch - 1])
    bad = S.copy()
    bad[0, 0, 0] += 1e-3j  # an imaginary part in a DC bin: realfft's C2R reports it (fft_backend.rs:559-563)
    with pytest.raises(Exception):
        plan.istft_batch(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,n_fft,hop", [("float32", 1024, 256), ("float32", 1024, 80), ("float32", 2048, 512), ("float64", 1024, 256), ("float64", 512, 160),
                                             ("float64", 512, 128), ("float32", 512, 128), ("float64", 2048, 512)])
def test_gpu_istft_non_finite_first_frame_stays_in_its_own_samples(dtype, n_fft, hop):
    """A NaN / Inf in frame 0 poisons the samples frame 0 covers and nothing else (the reference's overlap-add, spectrogram.rs:4906-4925).
    The fused inverses load frame 0 in place of the frames past the end of a ragged last tile; those stand-ins are replaced by zeros with a
    select — as a product with 0 they would carry the NaN into the tail of the signal."""
    rng = np.random.default_rng(31)
    n = 50000 + hop * 5  # a frame count that leaves stand-in frames in the last tile of 16 (32 at f64 512)
    x = rng.standard_normal((3, n)).astype(dtype)
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, dtype)
    S = np.ascontiguousarray(plan.compute_batch(x))
    assert S.shape[2] % 32 != 0
    clean = plan.istft_batch(S)
    bad = S.copy()
    bad[1, 5, 0] = np.nan
    bad[1, 9, 0] = np.inf
    y = plan.istft_batch(bad)
    ref = orc.istft(bad[1].astype(np.complex128), n_fft, hop, "hanning", True)
    fin = np.isfinite(ref)
    assert not fin[:n_fft // 2].any() and fin[n_fft // 2:].all()  # frame 0 covers output samples [0, n_fft / 2) of a centred signal
    if dtype == "float64" and n_fft == 512:
        # k_istft_d512 inverts frames 2 p and 2 p + 1 as the real and imaginary parts of ONE complex transform: a non-finite value in
        # frame 0 reaches frame 1 through the twiddle products, i.e. up to sample hop + n_fft / 2 — and no further
        fin_y = np.isfinite(y[1])
        assert not fin_y[:n_fft // 2].any() and fin_y[hop + n_fft // 2:].all()
        fin = fin & fin_y
    else:
        assert np.array_equal(np.isfinite(y[1]), fin)
    tol = (1e-10 if dtype == "float64" else 2e-5) * max(1.0, float(np.abs(ref[fin]).max()))
    assert np.max(np.abs(y[1][fin] - ref[fin])) < tol
    assert np.array_equal(y[0], clean[0]) and np.array_equal(y[2], clean[2])
