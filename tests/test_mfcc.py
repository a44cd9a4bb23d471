"""MFCC epilogue (src/mfcc.rs): oracle pinned on CPU against an independent scipy DCT restatement and the
reference's own mfcc tests' properties; GPU parity through the C ABI."""
import numpy as np
import pytest
import scipy.fft

import spectrograms_amd as sg
from oracle import oracle as orc
from spectrograms_amd import _ffi


def sig(n=16000, seed=0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    return (0.5 * np.sin(2 * np.pi * 440 * t) + 0.1 * rng.standard_normal(n)).astype(dtype)


def np_mfcc(logmel, n_mfcc, include_c0, lifter):
    c = scipy.fft.dct(logmel, type=2, axis=0)[:n_mfcc] / 2.0  # unnormalised sum x_i cos(pi k (i+.5)/n)
    if lifter > 0:
        c = c * (1.0 + (lifter / 2.0) * np.sin(np.pi * np.arange(n_mfcc) / lifter))[:, None]
    return c[1:] if (not include_c0 and n_mfcc > 1) else c


@pytest.mark.parametrize("n_fft,hop,n_mels,n_mfcc,c0,lifter", [(512, 160, 40, 13, True, 22), (400, 160, 64, 20, False, 22),
                                                               (1024, 256, 80, 13, True, 0), (256, 128, 26, 1, False, 22)])
def test_oracle_mfcc_matches_scipy(n_fft, hop, n_mels, n_mfcc, c0, lifter):
    x = sig()
    p = orc.Params(n_fft=n_fft, hop=hop, n_mels=n_mels, f_min=0.0, f_max=8000.0, amp="db", floor_db=-80.0)
    got = orc.mfcc(p, x, n_mfcc, c0, lifter)
    ref = np_mfcc(orc.spectrogram(p, x), n_mfcc, c0, lifter)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 1e-9
    # tests/mfcc_tests.rs:11-160: dims, finite, deterministic
    assert np.all(np.isfinite(got)) and np.array_equal(got, orc.mfcc(p, x, n_mfcc, c0, lifter))
    got32 = orc.mfcc(p, x.astype(np.float32), n_mfcc, c0, lifter)
    assert np.max(np.abs(got32 - ref)) < 2e-3 * max(1.0, np.max(np.abs(ref)))


def test_oracle_mfcc_rejects_too_many_coefficients():
    p = orc.Params(n_fft=512, hop=160, n_mels=20, amp="db", floor_db=-80.0)
    with pytest.raises(orc.OracleError):
        orc.mfcc(p, sig(2000), 21)


def test_host_validation_and_shape():
    st = sg.StftParams(512, 160, sg.WindowType.hanning, True)
    params = sg.SpectrogramParams(st, 16000.0)
    pl = sg.Plan(params, _ffi.AMP_DECIBELS, sg.MelParams(40, 0.0, 8000.0), sg.LogParams(-80.0), "float32",
                 device=_ffi.DEVICE_HOST_ONLY, mfcc=sg.MfccParams(13))
    assert pl.output_shape(16000) == (13, 101)
    pl2 = sg.Plan(params, _ffi.AMP_DECIBELS, sg.MelParams(40, 0.0, 8000.0), sg.LogParams(-80.0), "float32",
                  device=_ffi.DEVICE_HOST_ONLY, mfcc=sg.MfccParams(13).with_c0(False))
    assert pl2.output_shape(16000) == (12, 101)
    with pytest.raises(sg.InvalidInputError, match="n_mfcc must be <= n_mels"):  # src/mfcc.rs:231-233
        sg.Plan(params, _ffi.AMP_DECIBELS, sg.MelParams(12, 0.0, 8000.0), sg.LogParams(-80.0), "float32",
                device=_ffi.DEVICE_HOST_ONLY, mfcc=sg.MfccParams(13))
    with pytest.raises(sg.InvalidInputError, match="MFCC requires"):
        sg.Plan(params, _ffi.AMP_POWER, None, None, "float32", device=_ffi.DEVICE_HOST_ONLY, mfcc=sg.MfccParams(13))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop,n_mels,mp", [(512, 160, 40, sg.MfccParams(13)), (1024, 256, 80, sg.MfccParams(13)),
                                                 (400, 160, 64, sg.MfccParams(20).with_c0(False)),
                                                 (1024, 256, 80, sg.MfccParams(40).with_lifter(0))])
def test_gpu_mfcc_matches_oracle(n_fft, hop, n_mels, mp, dtype):
    npdt = np.float32 if dtype == "float32" else np.float64
    x = sig(20000, 3, npdt)
    st = sg.StftParams(n_fft, hop, sg.WindowType.hanning, True)
    got = sg.compute_mfcc(x, st, 16000.0, n_mels, mp, dtype=dtype)
    p = orc.Params(n_fft=n_fft, hop=hop, n_mels=n_mels, f_min=0.0, f_max=8000.0, amp="db", floor_db=-80.0)
    ref = orc.mfcc(p, x.astype(np.float64), mp.n_mfcc, mp.include_c0, mp.lifter)
    assert got.shape == ref.shape and got.data.dtype == npdt
    # MFCC sums n_mels dB values (each within 1e-3 dB for f32): tolerance n_mels * 1e-3 * lifter gain, f64 1e-8
    tol = 1e-7 if dtype == "float64" else 2e-2
    assert np.max(np.abs(got.data - ref)) < tol * max(1.0, np.max(np.abs(ref)) / 100)
    # batched plan == one-shot (tests/stft_plan_tests.rs:60-82 idiom)
    plan = sg.SpectrogramPlanner().mfcc_plan(st, 16000.0, n_mels, mp, dtype=dtype)
    xb = np.stack([x, x[::-1].copy()])
    yb = plan.compute_batch(xb)
    assert np.array_equal(yb[0], got.data)
    assert np.all(np.isfinite(yb))


@pytest.mark.gpu
@pytest.mark.parametrize("hop,mp,nm", [(256, sg.MfccParams(13), 80), (200, sg.MfccParams(20).with_c0(False), 80), (320, sg.MfccParams(40).with_lifter(0), 80),
                                       (256, sg.MfccParams(64), 96), (441, sg.MfccParams(20), 40), (256, sg.MfccParams(33).with_c0(False), 64),
                                       (256, sg.MfccParams(24), 24), (255, sg.MfccParams(13), 48)])
def test_gpu_fused_mfcc_keeps_the_reference_chain(hop, mp, nm):
    """f32, n_fft 1024: the DCT-II runs inside the Mel-dB launch on the matrix cores (kernels_r32x16.hip mfcc_tile; staged hop 256 /
    other staged hops / per-lane loads).  v_mfma_f32_16x16x4_f32 is an exact fused-multiply-add chain in ascending band order, i.e. the
    reference's `val.mul_add(basis, acc)` fold (src/mfcc.rs:278-292): checked against that chain evaluated on the host from the SAME
    kernel's Mel-dB output (f64 product + sum rounded to f32 = fmaf up to rare double roundings), several tiles per signal."""
    x = np.stack([sig(50000, s, np.float32) for s in range(5)])
    st = sg.StftParams(1024, hop, sg.WindowType.hanning, True)
    params = sg.SpectrogramParams(st, 16000.0)
    mel = sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(nm, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32").compute_batch(x)
    plan = sg.SpectrogramPlanner().mfcc_plan(st, 16000.0, nm, mp, dtype="float32")  # (the menu of chain lengths: 12 / 16 / 20 / 24 steps of 4 bands; 1 .. 4 tiles of 16 coefficients)
    assert plan.kernel_name == "r32x16_f32"
    got = plan.compute_batch(x)
    basis = np.cos(np.pi * np.arange(mp.n_mfcc)[:, None] * (np.arange(nm)[None, :] + 0.5) / nm).astype(np.float32)
    acc = np.zeros((5, mp.n_mfcc, mel.shape[2]), np.float32)
    for i in range(nm):
        acc = (mel[:, i, :][:, None, :].astype(np.float64) * basis[None, :, i, None].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    if mp.lifter > 0:
        w = (1.0 + (mp.lifter / 2.0) * np.sin(np.pi * np.arange(mp.n_mfcc) / mp.lifter)).astype(np.float32)
        acc = acc * w[None, :, None]
    ref = acc[:, 1:] if (not mp.include_c0 and mp.n_mfcc > 1) else acc
    assert got.shape == ref.shape
    same = np.mean(got == ref)
    assert same > 0.999 and np.max(np.abs(got - ref)) <= 4 * np.spacing(np.max(np.abs(ref)).astype(np.float32)), (same, np.max(np.abs(got - ref)))
