"""2-D FFT path (BASELINE config 5).  CPU: the oracle against numpy's rfft2/irfft2 and the
reference's known-answer tests (tests/fft2d_tests.rs, tests/images_ops_tests.rs, src/image_ops.rs:560-621).
GPU: the HIP path through the C ABI against the oracle."""
import os

import numpy as np
import pytest

import spectrograms_amd as sg
from oracle import oracle as orc
from spectrograms_amd import _ffi

SHAPES = [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (8, 8), (16, 32), (10, 12), (7, 9), (64, 64), (33, 20), (100, 50), (128, 128)]


def img(shape, seed=0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    r, c = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    return (np.sin(0.01 * r) + np.cos(0.02 * c) + 0.05 * rng.standard_normal(shape)).astype(dtype)


# ------------------------------------------------------------------ oracle pinned on CPU
@pytest.mark.parametrize("shape", SHAPES)
def test_oracle_fft2d_matches_numpy_and_roundtrips(shape):
    x = img(shape)
    S = orc.fft2d(x)
    ref = np.fft.rfft2(x)
    assert S.shape == (shape[0], shape[1] // 2 + 1)  # S13
    assert np.max(np.abs(S - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref)))
    assert np.max(np.abs(orc.ifft2d(S, shape[1]) - x)) < 1e-10          # tests/fft2d_tests.rs:67-143 round trips
    assert np.max(np.abs(orc.ifft2d(ref.copy(), shape[1]) - np.fft.irfft2(ref, s=shape))) < 1e-10
    x32 = x.astype(np.float32)
    assert np.max(np.abs(orc.fft2d(x32) - ref)) < 1e-5 * max(1.0, np.max(np.abs(ref)))


def test_oracle_fft2d_kats():
    # tests/fft2d_tests.rs:160-203: ones -> DC = rows*cols, rest < 1e-10; delta -> all 1+0i; Parseval; linearity
    ones = np.ones((8, 12))
    S = orc.fft2d(ones)
    assert abs(S[0, 0] - 96.0) < 1e-10
    S[0, 0] = 0
    assert np.max(np.abs(S)) < 1e-10
    d = np.zeros((8, 12))
    d[0, 0] = 1.0
    assert np.max(np.abs(orc.fft2d(d) - 1.0)) < 1e-12
    a, b = img((16, 16), 1), img((16, 16), 2)
    assert np.max(np.abs(orc.fft2d(2 * a + 3 * b) - (2 * orc.fft2d(a) + 3 * orc.fft2d(b)))) < 1e-10
    S = orc.fft2d(a)
    full = np.fft.fft2(a)
    assert abs((np.abs(full) ** 2).sum() / a.size - (a ** 2).sum()) < 1e-6  # Parseval (full spectrum)
    assert np.allclose(S, full[:, :9], atol=1e-10)


def test_oracle_inverse_forces_dc_and_nyquist_real():
    # src/fft_backend.rs:782-793: imag parts of the DC / Nyquist columns are dropped after the column IFFT
    rng = np.random.default_rng(3)
    S = rng.standard_normal((6, 5)) + 1j * rng.standard_normal((6, 5))  # ncols = 8 -> Nyquist column 4
    cols = np.fft.ifft(S, axis=0) * 6
    cols[:, 0] = cols[:, 0].real
    cols[:, 4] = cols[:, 4].real
    ref = np.fft.irfft(cols, n=8, axis=1) * 8 / 48.0
    assert np.max(np.abs(orc.ifft2d(S, 8) - ref)) < 1e-12


def test_oracle_image_ops_kats():
    # src/image_ops.rs:560-621 and tests/images_ops_tests.rs
    k = orc.gaussian_kernel_2d(5, 1.0)
    assert abs(k.sum() - 1.0) < 1e-10 and np.allclose(k, k.T) and np.allclose(k, k[::-1, ::-1])
    assert np.allclose(sg.gaussian_kernel_2d(5, 1.0), k, rtol=1e-15) and np.allclose(sg.gaussian_kernel_2d(9, 2.0), orc.gaussian_kernel_2d(9, 2.0), rtol=1e-15)
    image = np.add.outer(np.arange(64.0), np.arange(64.0))
    ident = np.zeros((3, 3))
    ident[1, 1] = 1.0
    res = orc.convolve_fft(image, ident)
    assert np.max(np.abs(res[1:63, 1:63] - image[1:63, 1:63])) < 1e-6
    wav = (np.sin(0.5 * np.arange(64))[:, None] + np.cos(0.5 * np.arange(64))[None, :]) * 50.0
    lp = orc.filter2d(wav, 0, 0.2)
    assert (lp ** 2).mean() < (wav ** 2).mean()
    hp = orc.filter2d(np.full((32, 32), 7.0), 1, 0.1)
    assert np.max(np.abs(hp)) < 1e-9  # highpass(const) ~ 0
    with pytest.raises(orc.OracleError):
        orc.filter2d(wav, 2, 0.5, 0.2)
    # convolution against a direct circular sum
    rng = np.random.default_rng(4)
    a, kk = rng.standard_normal((12, 10)), rng.standard_normal((3, 5))
    direct = np.zeros_like(a)
    for i in range(3):
        for j in range(5):
            direct += kk[i, j] * np.roll(np.roll(a, i - 1, axis=0), j - 2, axis=1)
    assert np.max(np.abs(orc.convolve_fft(a, kk) - direct)) < 1e-10


def test_lowpass_mask_uses_half_spectrum_dims():
    # S14: the radius / wrap use (nrows, ncols/2+1) as if it were the full width
    m = orc.lowpass_mask(64, 33, 0.5)
    assert m.shape == (64, 33) and m[0, 0] == 1.0
    r = min(64 // 2, 33 // 2) * 0.5
    assert m[0, int(r)] == 1.0 and m[0, int(r) + 1] == 0.0
    assert m[0, 33 - int(r)] == 1.0  # "wrapped" on the half width — the quirk


def test_rank1_kernel_spectrum_is_the_outer_product_of_two_1d_spectra():
    """The identity behind the rank-1 routes of convolve_fft (fft2d.hip outer_product_spectrum), checked without a GPU against
    the oracle: a kernel u v^T, wrapped as pad_kernel_for_fft does (centre to (0, 0): image_ops.rs:123-152), has the half spectrum
    U[k] V[c] with U, V the 1-D transforms of the wrapped u and v — so convolving with it equals convolving the rows with v and the
    columns with u.  Even and odd kernel sizes (the centre index is size // 2 in both), a 1 x n and an n x 1 kernel."""
    rng = np.random.default_rng(12)
    R, C = 64, 48
    x = img((R, C), 3, np.float64)
    for kr, kc in ((5, 5), (4, 6), (1, 7), (9, 1), (8, 3)):
        u, v = rng.standard_normal(kr), rng.standard_normal(kc)
        k = np.outer(u, v)

        def wrapped_spectrum(w, n, bins):
            t = (np.arange(w.size) - w.size // 2) % n
            kk = np.arange(bins)[:, None]
            return (w[None, :] * np.exp(-2j * np.pi * ((kk * t[None, :]) % n) / n)).sum(axis=1)

        U, V = wrapped_spectrum(u, R, R), wrapped_spectrum(v, C, C // 2 + 1)
        ref = orc.convolve_fft(x, k)
        X = orc.fft2d(x)
        got = orc.ifft2d(X * np.outer(U, V), C)
        assert np.max(np.abs(got - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref))), (kr, kc)
        # ... and the two separable passes: rows with v, then columns with u (full-length spectra, circular)
        Vf, Uf = wrapped_spectrum(v, C, C), wrapped_spectrum(u, R, R)
        rows = np.fft.ifft(np.fft.fft(x, axis=1) * Vf[None, :], axis=1).real
        sep = np.fft.ifft(np.fft.fft(rows, axis=0) * Uf[:, None], axis=0).real
        assert np.max(np.abs(sep - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref))), (kr, kc)


def test_host_validation():
    with pytest.raises(sg.InvalidInputError):
        sg.Fft2dPlan(0, 4, device=_ffi.DEVICE_HOST_ONLY)
    with pytest.raises(sg.InvalidInputError):
        sg.gaussian_kernel_2d(4, 1.0)
    with pytest.raises(sg.InvalidInputError):
        sg.gaussian_kernel_2d(5, 0.0)
    pl = sg.Fft2dPlan(8, 8, "float32", device=_ffi.DEVICE_HOST_ONLY)
    with pytest.raises(sg.FFTBackendError):
        pl.forward(np.zeros((8, 8), np.float32))


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("shape", SHAPES + [(256, 256), (1024, 1024), (512, 1024), (1024, 64), (100, 1024), (37, 1024), (4096, 8), (8, 4096),
                                            (8192, 4),
                                            # lengths without a register-tiled split: two-factor transform in LDS (1000 = 25 x 40, 1023 = 31 x 33,
                                            # 1001 = 13 x 77), the direct sum where its two buffers do not fit (6000 in f64) or the length is prime
                                            (1000, 6), (6, 1023), (63, 35), (1001, 4), (6000, 2), (509, 3),
                                            # neither a power of two nor a listed size, 16 ... 8192 (f64: 4096): chirp-z columns (length =
                                            # rows) and chirp-z inverse rows (length = columns), odd and even, primes and composites
                                            (251, 509), (509, 251), (1009, 12), (12, 1009), (1023, 1023), (127, 90), (90, 127), (2003, 5),
                                            (5, 2003), (4093, 3), (3, 4093), (17, 19), (6000, 3), (3, 6000), (98, 94),
                                            # (3, 6000) in f64: inverse rows in half-length complex form (their own chirp-z does not fit LDS); (2, 8200) in f32 likewise
                                            (2, 8200),
                                            # 1000 / 1200 / 1280 as column lengths and as (halved) row lengths: register-tiled, 40-point second pass
                                            (1200, 5), (1280, 3), (5, 2000), (3, 2400), (1080, 6), (6, 1280), (640, 4)])
def test_gpu_fft2d_matches_oracle(shape, dtype):
    npdt = np.float32 if dtype == "float32" else np.float64
    x = img(shape, 5, npdt)
    S = sg.fft2d(x, dtype=dtype)
    ref = orc.fft2d(x.astype(np.float64))
    tol = 1e-10 if dtype == "float64" else 2e-5
    assert S.shape == ref.shape and S.dtype == (np.complex64 if dtype == "float32" else np.complex128)
    assert np.max(np.abs(S - ref)) <= tol * max(1.0, np.max(np.abs(ref)))
    y = sg.ifft2d(S, shape[1], dtype=dtype)
    assert np.max(np.abs(y - x)) <= (1e-10 if dtype == "float64" else 1e-5) * max(1.0, np.max(np.abs(x)))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_gpu_image_ops_match_oracle(dtype):
    npdt = np.float32 if dtype == "float32" else np.float64
    tol = 1e-9 if dtype == "float64" else 2e-5
    for shape, ksz in (((64, 64), 9), ((100, 50), 5), ((256, 128), 9)):
        x = img(shape, 7, npdt)
        k = sg.gaussian_kernel_2d(ksz, 2.0, dtype=dtype)
        got = sg.convolve_fft(x, k, dtype=dtype)
        ref = orc.convolve_fft(x.astype(np.float64), k.astype(np.float64))
        assert np.max(np.abs(got - ref)) <= tol * max(1.0, np.max(np.abs(ref)))
        for kind, fn, args in ((0, sg.lowpass_filter, (0.3,)), (1, sg.highpass_filter, (0.2,)), (2, sg.bandpass_filter, (0.1, 0.6))):
            got = fn(x, *args, dtype=dtype)
            ref = orc.filter2d(x.astype(np.float64), kind, *args)
            assert np.max(np.abs(got - ref)) <= tol * max(1.0, np.max(np.abs(ref)))
    with pytest.raises(sg.InvalidInputError, match="high_cutoff must be greater"):  # tests/images_ops_tests.rs:409-427
        sg.bandpass_filter(np.zeros((16, 16)), 0.5, 0.2)
    with pytest.raises(sg.InvalidInputError, match="must not exceed"):
        sg.convolve_fft(np.zeros((4, 4)), np.zeros((5, 5)))
    with pytest.raises(sg.DimensionMismatchError):
        sg.ifft2d(np.zeros((8, 4), np.complex128), 8)  # 8 // 2 + 1 = 5 columns expected


@pytest.mark.gpu
def test_gpu_fft2d_batched_config5_subset():
    # BASELINE config 5 shape, 4 of the 512 images: batched forward + Gaussian 9x9 convolve vs the oracle per image
    imgs = np.stack([img((1024, 1024), 7 + k, np.float32) for k in range(4)])
    plan = sg.Fft2dPlan(1024, 1024, "float32")
    S = plan.forward(imgs)
    k = sg.gaussian_kernel_2d(9, 2.0, dtype="float32")
    Y = plan.convolve(imgs, k)
    for i in range(4):
        ref = orc.fft2d(imgs[i].astype(np.float64))
        assert np.max(np.abs(S[i] - ref)) <= 2e-5 * np.max(np.abs(ref))
        rc = orc.convolve_fft(imgs[i].astype(np.float64), k.astype(np.float64))
        assert np.max(np.abs(Y[i] - rc)) <= 2e-5 * np.max(np.abs(rc))


@pytest.mark.gpu
def test_gpu_fft2d_config5_full_batch_properties():
    """BASELINE configs[4] at its full size: 512 x 1024 x 1024 f32 in one call per op (the XCD-aware batch mapping at B = 512).
    The oracle covers 4 images elsewhere; here every image is checked through size-independent properties: Parseval with the
    half spectrum's Hermitian weights, ifft2d(fft2d(x)) = x, convolution with a delta = identity, and two images against the
    oracle (first and last)."""
    torch = pytest.importorskip("torch")
    B, R, Cn = 512, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(7)
    r = torch.arange(R, device="cuda", dtype=torch.float32)[:, None]
    c = torch.arange(Cn, device="cuda", dtype=torch.float32)[None, :]
    x = (torch.sin(0.01 * r) + torch.cos(0.02 * c))[None] + 0.05 * torch.randn((B, R, Cn), generator=g, device="cuda", dtype=torch.float32)
    x *= 1.0 + 0.001 * torch.arange(B, device="cuda", dtype=torch.float32)[:, None, None]  # no two images alike
    plan = sg.Fft2dPlan(R, Cn, "float32")
    plan.reserve(B, host_staging=False)
    S = plan.forward_torch(x)
    torch.cuda.synchronize()
    assert tuple(S.shape) == (B, R, Cn // 2 + 1, 2) and bool(torch.isfinite(S).all())
    p = (S.double() ** 2).sum(dim=3)                      # |S|^2, [B, R, 513]
    w = torch.full((Cn // 2 + 1,), 2.0, device="cuda", dtype=torch.float64)
    w[0] = w[-1] = 1.0                                   # columns 0 and N/2 have no mirror
    lhs = (p * w).sum(dim=(1, 2))
    rhs = (x.double() ** 2).sum(dim=(1, 2)) * (R * Cn)
    assert float(((lhs - rhs).abs() / rhs).max()) < 1e-5
    y = plan.inverse_torch(S)
    assert float((y - x).abs().max()) <= 2e-5 * float(x.abs().max())
    z = plan.convolve_torch(x, np.ones((1, 1), np.float32))
    assert float((z - x).abs().max()) <= 2e-5 * float(x.abs().max())
    for i in (0, B - 1):
        ref = orc.fft2d(x[i].cpu().numpy().astype(np.float64))
        got = torch.view_as_complex(S[i]).cpu().numpy()
        assert np.max(np.abs(got - ref)) <= 2e-5 * np.max(np.abs(ref))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1024, 1024), (1024, 64), (1024, 100), (1024, 7)])
def test_gpu_fused_column_stage_1024_rows(shape):
    """f32 images with 1024 rows take the fused column kernel (forward FFT x kernel spectrum / mask x inverse FFT in one
    pass) for convolve_fft and the radial filters; the column count is free (tuned or generic row kernels)."""
    x = np.stack([img(shape, 11 + k, np.float32) for k in range(3)])
    plan = sg.Fft2dPlan(shape[0], shape[1], "float32")
    ksz = 9 if shape[1] >= 9 else 5
    k = sg.gaussian_kernel_2d(ksz, 2.0, dtype="float32")
    got = plan.convolve(x, k)
    for i in range(3):
        ref = orc.convolve_fft(x[i].astype(np.float64), k.astype(np.float64))
        assert np.max(np.abs(got[i] - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref)))
    for kind, args in ((0, (0.3, 0.0)), (1, (0.2, 0.0)), (2, (0.1, 0.6))):
        got = plan.filter(x, kind, *args)
        for i in range(3):
            ref = orc.filter2d(x[i].astype(np.float64), kind, *args)
            assert np.max(np.abs(got[i] - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref)))
    kq = np.random.default_rng(5).standard_normal((min(7, shape[0]), min(5, shape[1]))).astype(np.float32)  # not an outer product: the full 2-D kernel spectrum
    got = plan.convolve(x, kq)
    for i in range(3):
        ref = orc.convolve_fft(x[i].astype(np.float64), kq.astype(np.float64))
        assert np.max(np.abs(got[i] - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1024, 1024), (1024, 100), (1024, 7)])
def test_gpu_rank1_kernels_multiply_in_their_two_factors(shape, monkeypatch):
    """A kernel that is an outer product u v^T to f32 rounding (gaussian_kernel_2d, image_ops.rs:188-220; a box; any separable filter)
    has the 2-D spectrum U[k] V[col]: the fused column kernel takes the two 1-D factors (12 KB) instead of building and re-reading a
    4.2 MB spectrum per image (fft2d.hip outer_product_spectrum, k_colconv1024<MUL_OUTER>).  Against the oracle, and against the same
    call with the detection switched off (SGX_CONV_RANK1=0: the general path) — even kernel sizes and a 1 x n kernel check the
    centring of pad_kernel_for_fft (image_ops.rs:123-152) in both factors; a kernel one element away from rank 1 stays general."""
    rng = np.random.default_rng(9)
    x = np.stack([img(shape, 21 + k, np.float32) for k in range(3)])
    kernels = [sg.gaussian_kernel_2d(9 if shape[1] >= 9 else 5, 2.0, dtype="float32"),
               np.outer(rng.standard_normal(4), rng.standard_normal(6 if shape[1] >= 6 else 3)).astype(np.float32),
               np.ones((1, 5), np.float32) / 5, np.ones((31, 1), np.float32) / 31,
               np.outer(np.hanning(shape[0] // 2), np.hanning(min(64, shape[1]) - 1)).astype(np.float32)]
    near = kernels[0].copy()
    near[1, 2] *= 1.001
    for k in kernels + [near]:
        monkeypatch.delenv("SGX_CONV_RANK1", raising=False)
        monkeypatch.delenv("SGX_CONV_SEPARABLE", raising=False)
        got = sg.Fft2dPlan(shape[0], shape[1], "float32").convolve(x, k)  # (1024 x 1024: the two separable passes over pairs of real rows)
        monkeypatch.setenv("SGX_CONV_SEPARABLE", "0")
        outer = sg.Fft2dPlan(shape[0], shape[1], "float32").convolve(x, k)  # three passes, the outer-product multiplier in the column kernel
        monkeypatch.setenv("SGX_CONV_RANK1", "0")
        general = sg.Fft2dPlan(shape[0], shape[1], "float32").convolve(x, k)
        scale = 0.0
        for i in range(3):
            ref = orc.convolve_fft(x[i].astype(np.float64), k.astype(np.float64))
            scale = max(1.0, float(np.max(np.abs(ref))))
            assert np.max(np.abs(got[i] - ref)) <= 2e-5 * scale, k.shape
            assert np.max(np.abs(general[i] - ref)) <= 2e-5 * scale, k.shape
            assert np.max(np.abs(outer[i] - ref)) <= 2e-5 * scale, k.shape
        assert np.max(np.abs(got - general)) <= 4e-6 * scale and np.max(np.abs(outer - general)) <= 4e-6 * scale
        if k is near:
            assert np.array_equal(got, general) and np.array_equal(outer, general)  # not rank 1 within 2^-22: the same path either way
        elif shape != (1024, 1024):
            assert np.array_equal(got, outer)  # the separable passes need 1024 points along both axes
    monkeypatch.delenv("SGX_CONV_RANK1", raising=False)
    monkeypatch.delenv("SGX_CONV_SEPARABLE", raising=False)


@pytest.mark.gpu
def test_gpu_convolve_config5_full_batch_gaussian(monkeypatch):
    """BASELINE configs[4] at its full size through convolve_fft with the 9 x 9 Gaussian: 512 x 1024 x 1024 f32 in one call (one launch
    of each of the two separable passes), first and last image against the oracle, and every image against the same plan convolving it alone."""
    torch = pytest.importorskip("torch")
    B = 512
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((B, 1024, 1024), generator=g, device="cuda", dtype=torch.float32)
    k = sg.gaussian_kernel_2d(9, 2.0, dtype="float32")
    plan = sg.Fft2dPlan(1024, 1024, "float32")
    y = plan.convolve_torch(x, k)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())
    for i in (0, B - 1):
        ref = orc.convolve_fft(x[i].cpu().numpy().astype(np.float64), k.astype(np.float64))
        assert np.max(np.abs(y[i].cpu().numpy() - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref)))
    small = sg.Fft2dPlan(1024, 1024, "float32")
    for i in (0, 63, 64, 300, B - 1):
        assert torch.equal(small.convolve_torch(x[i:i + 1], k)[0], y[i])
    # every image of the batch: repeated launches give the same bits (a workgroup walks 64 tiles here: a timing-dependent fault in the
    # prefetched tiles, like the one the first form of the real-pair loads had, shows as a launch that differs), and the three-pass
    # route (SGX_CONV_SEPARABLE=0) agrees within the f32 tolerance of the two routes
    for rep in range(3):
        assert torch.equal(plan.convolve_torch(x, k), y)
    monkeypatch.setenv("SGX_CONV_SEPARABLE", "0")
    three = sg.Fft2dPlan(1024, 1024, "float32").convolve_torch(x, k)
    monkeypatch.delenv("SGX_CONV_SEPARABLE", raising=False)
    assert float((three - y).abs().max()) <= 4e-6 * max(1.0, float(y.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["general", "gaussian"])
def test_gpu_convolve_large_batch_runs_in_chunks_on_two_streams(kernel, monkeypatch):
    """From 128 images of 1024 x 1024 f32 on, convolve_fft runs as chunks of 64 alternating between the caller's stream and a
    plan-owned second one (fft2d.hip: fused_product_dev) — a rank-1 kernel (the Gaussian) as groups of up to 512 images (here 100, by
    SGX_SEP_GROUP) through its two separable passes on the caller's stream.  The result must be, bit for bit, what the same images give in small
    batches (which take the single-stream schedule); the call must behave as ONE operation of the caller's stream — work queued
    behind it sees all of its output, work queued before it is seen by all of its chunks — and must replay from a hipGraph."""
    torch = pytest.importorskip("torch")
    B, R, Cn = 168, 1024, 1024  # 64 + 64 + 40: both streams, a ragged last chunk (100 + 68 for the separable passes)
    monkeypatch.setenv("SGX_SEP_GROUP", "100")
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn((B, R, Cn), generator=g, device="cuda", dtype=torch.float32)
    k = sg.gaussian_kernel_2d(9, 2.0, dtype="float32") if kernel == "gaussian" else np.random.default_rng(3).standard_normal((5, 7)).astype(np.float32)
    small = sg.Fft2dPlan(R, Cn, "float32")
    ref = torch.cat([small.convolve_torch(x[i:i + 24], k) for i in range(0, B, 24)])
    plan = sg.Fft2dPlan(R, Cn, "float32")
    plan.reserve(B, host_staging=False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        xin = torch.zeros_like(x)
        xin.copy_(x)                       # queued before the call, on the caller's stream
        y = plan.convolve_torch(xin, k)
        tot = y.sum(dim=(1, 2))            # queued behind the call
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    assert torch.equal(tot, ref.sum(dim=(1, 2)))
    # same kernel on the same stream again -> no host copy in the call -> capturable; the replay works on new input in the same buffers
    out = torch.empty_like(x)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        plan.convolve_torch(xin, k, out=out)
    xin.copy_(x.flip(0))
    out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref.flip(0))
    # the radial filters share the schedule
    f = plan.filter_torch(x, 1, 0.2, 0.0)
    fr = torch.cat([small.filter_torch(x[i:i + 24], 1, 0.2, 0.0) for i in range(0, B, 24)])
    assert torch.equal(f, fr)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((1024, 64), "float32"), ((64, 48), "float32"), ((40, 64), "float64")])
def test_gpu_kernel_spectrum_and_mask_are_reused_only_when_unchanged(shape, dtype):
    """A plan keeps the last kernel's spectrum / the last filter mask on the device and prepares them again only when the kernel
    bytes, its shape, the cut-offs or the stream differ: every call must equal a fresh plan's answer bit for bit."""
    torch = pytest.importorskip("torch")
    npdt = np.float32 if dtype == "float32" else np.float64
    x = np.stack([img(shape, 3 + k, npdt) for k in range(2)])
    k1 = sg.gaussian_kernel_2d(5, 1.0, dtype=dtype)
    k2 = sg.gaussian_kernel_2d(5, 2.5, dtype=dtype)       # same shape, other values
    k3 = np.ascontiguousarray(k2[:3, :])                  # same leading bytes, other shape
    fresh = lambda k: sg.Fft2dPlan(shape[0], shape[1], dtype).convolve(x, k)
    plan = sg.Fft2dPlan(shape[0], shape[1], dtype)
    for k in (k1, k1, k2, k2, k3, k1):
        assert np.array_equal(plan.convolve(x, k), fresh(k))
    fresh_f = lambda *a: sg.Fft2dPlan(shape[0], shape[1], dtype).filter(x, *a)
    for a in ((0, 0.3, 0.0), (0, 0.3, 0.0), (1, 0.3, 0.0), (2, 0.3, 0.6), (2, 0.3, 0.7), (0, 0.3, 0.0)):
        assert np.array_equal(plan.filter(x, *a), fresh_f(*a))
    # the cached spectrum belongs to the stream that produced it: another stream prepares its own
    xd = torch.from_numpy(x).cuda()
    ref = fresh(k1)
    a = plan.convolve_torch(xd, k1).cpu().numpy()
    with torch.cuda.stream(torch.cuda.Stream()):
        b = plan.convolve_torch(xd, k1)
        torch.cuda.current_stream().synchronize()
    c = plan.convolve_torch(xd, k1).cpu().numpy()
    assert np.array_equal(a, ref) and np.array_equal(b.cpu().numpy(), ref) and np.array_equal(c, ref)


@pytest.mark.gpu
def test_gpu_planner_spectrum_helpers():
    pl = sg.Fft2dPlanner(dtype="float32")
    x = img((64, 48), 3, np.float32)
    S = pl.fft2d(x)
    assert S.dtype == np.complex64
    assert np.allclose(pl.power_spectrum_2d(x), np.abs(S) ** 2, rtol=1e-6) and np.allclose(pl.magnitude_spectrum_2d(x), np.abs(S), rtol=1e-6)


@pytest.mark.gpu
def test_gpu_fft2d_fuzz_shapes():
    """Seeded random sweep over image shapes — every column / row kernel family (tuned 1024, register-tiled power-of-two and
    mixed-radix lengths, LDS radix-2, direct) for fft2d, ifft2d, convolve_fft and a filter, both dtypes."""
    rng = np.random.default_rng(int(os.environ.get("SGX_FUZZ_SEED", 4242)))
    rows = [16, 32, 64, 128, 256, 512, 40, 60, 80, 100, 120, 200, 240, 480, 600, 12, 30, 97, 1024]
    cols = [32, 64, 128, 256, 512, 80, 160, 200, 240, 320, 400, 480, 640, 30, 50, 99, 101, 1024]
    for case in range(24):
        shape = (int(rows[rng.integers(len(rows))]), int(cols[rng.integers(len(cols))]))
        dtype = ["float32", "float64"][rng.integers(2)]
        npdt = np.float32 if dtype == "float32" else np.float64
        x = img(shape, 100 + case, npdt)
        S = sg.fft2d(x, dtype=dtype)
        ref = orc.fft2d(x.astype(np.float64))
        tol = 1e-10 if dtype == "float64" else 3e-5
        assert np.max(np.abs(S - ref)) <= tol * max(1.0, np.max(np.abs(ref))), (shape, dtype)
        y = sg.ifft2d(S, shape[1], dtype=dtype)
        assert np.max(np.abs(y - x)) <= (1e-10 if dtype == "float64" else 2e-5) * max(1.0, np.max(np.abs(x))), (shape, dtype)
        ksz = int([3, 5, 9][rng.integers(3)])
        if ksz <= min(shape):
            k = sg.gaussian_kernel_2d(ksz, 1.5, dtype=dtype)
            got = sg.convolve_fft(x, k, dtype=dtype)
            refc = orc.convolve_fft(x.astype(np.float64), k.astype(np.float64))
            assert np.max(np.abs(got - refc)) <= (1e-9 if dtype == "float64" else 3e-5) * max(1.0, np.max(np.abs(refc))), (shape, dtype)
        got = sg.lowpass_filter(x, 0.4, dtype=dtype)
        reff = orc.filter2d(x.astype(np.float64), 0, 0.4)
        assert np.max(np.abs(got - reff)) <= (1e-9 if dtype == "float64" else 3e-5) * max(1.0, np.max(np.abs(reff))), (shape, dtype)


@pytest.mark.gpu
def test_gpu_fft2d_chirpz_fuzz_batched():
    """Seeded sweep over shapes whose column and / or row length has neither a power of two nor a listed split — chirp-z columns
    (`k_bs_c2c`) and chirp-z inverse rows (`HERM`) — in BATCHES (several images per call, more images than one tile holds sequences,
    sequence counts that leave partial tiles): fft2d, ifft2d, convolve_fft and a filter per image against the oracle, both dtypes."""
    rng = np.random.default_rng(int(os.environ.get("SGX_FUZZ_SEED", 777)))
    odd = [17, 19, 23, 31, 37, 53, 61, 97, 101, 127, 131, 211, 251, 257, 331, 509, 521, 1009, 1021, 2003, 98, 94, 202, 1006, 1023]
    easy = [16, 20, 64, 100, 128, 256]
    for case in range(20):
        pick = rng.integers(3)
        r = int(odd[rng.integers(len(odd))]) if pick != 1 else int(easy[rng.integers(len(easy))])
        c = int(odd[rng.integers(len(odd))]) if pick != 0 else int(easy[rng.integers(len(easy))])
        if r * c > 600000:
            c = int(easy[rng.integers(len(easy))])
        dtype = ["float32", "float64"][rng.integers(2)]
        npdt = np.float32 if dtype == "float32" else np.float64
        batch = int(rng.integers(1, 6))
        x = np.stack([img((r, c), 900 + 7 * case + b, npdt) for b in range(batch)])
        plan = sg.Fft2dPlan(r, c, dtype)
        S = plan.forward(x)
        y = plan.inverse(S)
        k = sg.gaussian_kernel_2d(3, 1.0, dtype=dtype)
        Y = plan.convolve(x, k)
        F = plan.filter(x, 1, 0.3, 0.0)
        tol = 1e-10 if dtype == "float64" else 3e-5
        for b in range(batch):
            ref = orc.fft2d(x[b].astype(np.float64))
            assert np.max(np.abs(S[b] - ref)) <= tol * max(1.0, np.max(np.abs(ref))), (r, c, dtype, b)
            assert np.max(np.abs(y[b] - x[b])) <= tol * max(1.0, np.max(np.abs(x[b]))), (r, c, dtype, b)
            refc = orc.convolve_fft(x[b].astype(np.float64), k.astype(np.float64))
            assert np.max(np.abs(Y[b] - refc)) <= 10 * tol * max(1.0, np.max(np.abs(refc))), (r, c, dtype, b)
            reff = orc.filter2d(x[b].astype(np.float64), 1, 0.3, 0.0)
            assert np.max(np.abs(F[b] - reff)) <= 10 * tol * max(1.0, np.max(np.abs(reff))), (r, c, dtype, b)


def test_c2c_plan_host_validation():
    p = sg.C2cPlan(16, "float32", device=_ffi.DEVICE_HOST_ONLY)
    with pytest.raises(sg.DimensionMismatchError) as ei:
        p.forward(np.zeros(15, np.complex64))
    assert ei.value.expected == 16 and ei.value.got == 15
    with pytest.raises(sg.FFTBackendError):  # no CPU fallback
        p.forward(np.zeros(16, np.complex64))
    with pytest.raises(sg.InvalidInputError):
        sg.C2cPlan(0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dtype,tol", [(8, "float64", 1e-12), (1024, "float32", 2e-6), (1000, "float64", 1e-11), (100, "float32", 2e-6),
                                         (1009, "float32", 2e-6), (1009, "float64", 1e-11), (251, "float32", 2e-6), (97, "float64", 1e-11), (4093, "float32", 2e-6),
                                         (5003, "float32", 2e-6), (3000, "float64", 1e-11), (17, "float32", 2e-6),
                                         (7, "float64", 1e-12), (4096, "float64", 1e-11)])
def test_gpu_c2c_plan_matches_numpy(n, dtype, tol):
    """C2cPlan<T>::forward / inverse (src/fft_backend.rs:113-137): unnormalised both ways; inverse(forward(x)) = n x
    (the reference's own round-trip KAT, fft_backend.rs:1909-1927)."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    p = sg.C2cPlan(n, dtype)
    X = p.forward(x)
    ref = np.fft.fft(x)
    assert np.max(np.abs(X - ref)) <= tol * np.max(np.abs(ref)) * max(1.0, np.log2(n))
    y = p.inverse(X)
    assert np.max(np.abs(y / n - x)) <= tol * np.max(np.abs(x)) * max(1.0, np.log2(n))
    ones = p.forward(np.ones(n))
    assert abs(ones[0] - n) <= tol * n and np.max(np.abs(ones[1:])) <= tol * n * max(1.0, np.log2(n))
