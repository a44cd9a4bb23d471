"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU oracle
and the committed golden vectors.

Tolerances (SURVEY.md §8c / BASELINE.json north_star), all against the f64 oracle:
  f64            : <= 1e-10 * max(1, max|ref|)   (the reference's own f64 round-trip bound, fft_backend.rs:1886-1907)
  f32 complex    : <= 1e-4 * max|X| abs          (north_star: within 1e-4 rel. of RustFFT)
  f32 power/mel  : <= 1e-4 * max(ref) abs everywhere, <= 1e-4 relative where ref is within 40 dB of the peak
                   (ref > 1e-4 * max), and <= 5e-3 relative where ref > 1e-6 * max — the reference's own accepted f32 bound
                   (src/spectrogram.rs:5359-5362).  Bins 40-60 dB below the peak sit at the rounding noise of ANY f32 FFT
                   (eps * sqrt(log N) * |X|max ~ 2e-7 * |X|max, i.e. ~4e-4 relative in power at -60 dB), RustFFT included.
  dB             : <= 1e-3 dB abs where the power is within 40 dB of the peak, <= 0.05 dB within 60 dB; >= floor everywhere
A tighter regression guard (GUARD) is asserted as well so that a numerically sloppy kernel cannot hide in the slack.
"""
import os

import numpy as np
import pytest

import spectrograms_amd as sg
from oracle import oracle as orc
from spectrograms_amd import _ffi
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL32 = 1e-4
GUARD32 = 2e-5
TOL64 = 1e-10

WINDOWS = {
    "hanning": (sg.WindowType.hanning, {}),
    "hamming": (sg.WindowType.hamming, {}),
    "blackman": (sg.WindowType.blackman, {}),
    "rectangular": (sg.WindowType.rectangular, {}),
    "kaiser": (sg.WindowType.kaiser(5.0), {"window_param": 5.0}),
    "gaussian": (sg.WindowType.gaussian(60.0), {"window_param": 60.0}),
}


def signals(batch, n, dtype, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    rows = []
    for b in range(batch):
        if b % 2 == 0:
            rows.append(0.5 * np.sin(2 * np.pi * (110.0 * 2 ** (b / 3.0)) * t) + 0.05 * rng.standard_normal(n))
        else:
            rows.append(0.3 * rng.standard_normal(n))
    return np.stack(rows).astype(dtype)


def make(n_fft, hop, window="hanning", centre=True, n_mels=0, fmin=0.0, fmax=8000.0, norm=None, amp="power",
         floor=None, dtype="float32", sr=16000.0):
    w, okw = WINDOWS[window]
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, w, centre), sr)
    mel = None
    if n_mels:
        mn = {None: None, "slaney": sg.MelNorm.slaney, "l1": sg.MelNorm.l1, "l2": sg.MelNorm.l2}[norm]
        mel = sg.MelParams(n_mels, fmin, fmax, mn)
    db = sg.LogParams(floor) if floor is not None else None
    code = {"power": _ffi.AMP_POWER, "magnitude": _ffi.AMP_MAGNITUDE, "db": _ffi.AMP_DECIBELS, "complex": _ffi.AMP_COMPLEX}[amp]
    plan = sg.Plan(params, code, mel, db, dtype)
    op = orc.Params(n_fft=n_fft, hop=hop, window=window, centre=centre, sample_rate=sr, n_mels=n_mels, f_min=fmin,
                    f_max=fmax, mel_norm=norm, amp="power" if amp == "complex" else amp, floor_db=floor, **okw)
    return plan, op


def check(got, ref64, kind, dtype, floor=None, pow64=None):
    got = np.asarray(got)
    assert got.shape == ref64.shape, (got.shape, ref64.shape)
    assert np.all(np.isfinite(got))
    scale = max(float(np.max(np.abs(ref64))), 1e-300)
    if dtype == "float64":
        if kind == "db":
            m = pow64 > 1e-12 * pow64.max()
            assert np.max(np.abs(got[m] - ref64[m])) < 1e-8
        else:
            assert np.max(np.abs(got - ref64)) <= TOL64 * max(1.0, scale)
        return
    g = got.astype(np.complex128 if np.iscomplexobj(got) else np.float64)
    if kind == "complex":
        err = np.max(np.abs(g - ref64)) / scale
        assert err <= TOL32, err
        assert err <= GUARD32, f"regression guard: {err}"
    elif kind == "db":
        assert g.min() >= floor - 1e-3
        m = pow64 > 1e-4 * pow64.max()
        assert not m.any() or np.max(np.abs(g[m] - ref64[m])) <= 1e-3
        m = pow64 > 1e-6 * pow64.max()
        assert not m.any() or np.max(np.abs(g[m] - ref64[m])) <= 0.05
    else:
        err_abs = np.max(np.abs(g - ref64)) / scale
        # relative checks on the POWER scale: within 40 dB (1e-4) and 60 dB (5e-3, the reference's own f32 bound) of the
        # peak; magnitude = sqrt(power), so its thresholds are the square roots and its relative errors half as large
        t40, t60 = (1e-2, 1e-3) if kind == "magnitude" else (1e-4, 1e-6)
        m = ref64 > t40 * scale
        err_rel = np.max(np.abs(g[m] - ref64[m]) / ref64[m]) if m.any() else 0.0
        m = ref64 > t60 * scale
        err_rel60 = np.max(np.abs(g[m] - ref64[m]) / ref64[m]) if m.any() else 0.0
        assert err_abs <= TOL32 and err_rel <= TOL32 and err_rel60 <= 5e-3, (err_abs, err_rel, err_rel60)
        assert err_abs <= GUARD32, f"regression guard: {err_abs}"


def run_case(n, batch=3, seed=0, **kw):
    dtype = kw.get("dtype", "float32")
    amp = kw.get("amp", "power")
    plan, op = make(**kw)
    x = signals(batch, n, np.float32 if dtype == "float32" else np.float64, seed)
    got = plan.compute_batch(x)
    x64 = x.astype(np.float64)
    if amp == "complex":
        ref = orc.stft_batch(op, x64)
        check(got, ref, "complex", dtype)
    else:
        ref = orc.spectrogram_batch(op, x64)
        pow64 = None
        if amp == "db" and kw.get("floor") is None:
            amp = "power"  # S6: Decibels without LogParams returns power
        if amp == "db":
            pop = orc.Params(**{**op.__dict__, "amp": "power", "floor_db": None, "_keep": []})
            pow64 = orc.spectrogram_batch(pop, x64)
        check(got, ref, amp, dtype, kw.get("floor"), pow64)
    return plan, got


# ------------------------------------------------------------------ pow2 sizes, both kernels, both dtypes
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop", [(4, 2), (8, 3), (16, 5), (32, 7), (64, 16), (128, 33), (256, 128), (512, 256), (512, 171),
                                       (1024, 256), (1024, 512), (1024, 333), (2048, 512), (4096, 1024), (8192, 2048)])
@pytest.mark.parametrize("amp", ["complex", "power"])
def test_pow2_sizes(n_fft, hop, amp, dtype):
    """Every radix split of the register-tiled kernel (two passes up to n_fft 512 in f32 / 128 in f64, three above), odd hops
    (unaligned frames), the LDS radix-2 kernel below 32 points."""
    plan, _ = run_case(n=max(6000, 3 * n_fft), n_fft=n_fft, hop=hop, amp=amp, dtype=dtype)
    fits = n_fft <= (4096 if dtype == "float64" else 8192)  # an 8192-point f64 tile (164 KB with its tables) exceeds the CU's LDS
    tuned = dtype == "float32" and (n_fft in (1024, 2048, 4096) or (n_fft == 512 and hop in (64, 128, 160, 256)))  # (2048 / 4096: any hop since round 5)
    tuned64 = dtype == "float64" and (n_fft in (1024, 2048) or (n_fft == 512 and hop <= 260))  # (any hop since round 5)
    if tuned64:
        assert plan.kernel_name == {1024: "d32x16_f64", 512: "d512_f64", 2048: "d32x32_f64"}[n_fft]
    elif 32 <= n_fft and fits and not tuned:
        assert plan.kernel_name == "reg_radix"


# ------------------------------------------------------------------ n_fft 2048, f32: the tuned kernel k_r32x32 (round 4)
@pytest.mark.parametrize("hop", [512, 256, 1024, 2048, 100, 544, 546, 600, 2, 441, 543, 545, 1023, 1])
@pytest.mark.parametrize("amp,floor,n_mels", [("complex", None, None), ("power", None, None), ("magnitude", None, None), ("db", -80.0, None),
                                              ("power", None, 80), ("db", -80.0, 80), ("magnitude", None, 40), ("power", None, 128),
                                              ("db", -80.0, 24)])
def test_tuned_2048(hop, amp, floor, n_mels):
    """The reference's music default n_fft 2048 / hop 512 (src/spectrogram.rs:4243-4248) and its neighbours on k_r32x32: staged samples up to
    hop 544, per-lane columns above; every output mode; frame counts that are not multiples of the 16-frame tile; centre on and off; a
    signal's bits independent of its batch; odd hops since round 5 (staged pairs by ds_read2_b32, the row-start pair patched in the direct path)."""
    n = 23 * 1024 + 77 if hop >= 100 else 6000
    kw = dict(n_fft=2048, hop=hop, amp=amp, floor=floor, dtype="float32")
    if n_mels:
        kw.update(n_mels=n_mels, fmin=0.0, fmax=8000.0)
    plan, got = run_case(n=n, batch=3, **kw)
    assert plan.kernel_name == "r32x32_f32"
    x = signals(3, n, np.float32, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[2:3]))[0], np.asarray(got)[2])
    run_case(n=n, batch=2, centre=False, **kw)



# ------------------------------------------------------------------ n_fft 1024, f64: the tuned kernel k_d32x16 (round 4)
@pytest.mark.parametrize("hop", [256, 128, 512, 1024, 64, 272, 274, 600, 2, 255, 273, 441, 1])
@pytest.mark.parametrize("amp,floor,n_mels", [("complex", None, None), ("power", None, None), ("magnitude", None, None), ("db", -80.0, None),
                                              ("power", None, 80), ("db", -80.0, 80), ("magnitude", None, 40), ("power", None, 128),
                                              ("db", -80.0, 24)])
def test_tuned_f64_1024(hop, amp, floor, n_mels):
    """BASELINE's shape in f64 on k_d32x16 (half rows in lane pairs traded with v_permlane32_swap): staged samples up to hop 272, per-lane
    columns above; per-bin and complex outputs; frame counts that are not multiples of the 16-frame tile; centre on and off; a signal's
    bits independent of its batch; filterbank outputs through the scheduled band stage (8-byte weights).  Odd hops stay on the register-tiled kernel."""
    n = 23 * 512 + 77 if hop >= 64 else 3000
    kw = dict(n_fft=1024, hop=hop, amp=amp, floor=floor, dtype="float64")
    if n_mels:
        kw.update(n_mels=n_mels, fmin=0.0, fmax=8000.0)
    plan, got = run_case(n=n, batch=3, **kw)
    assert plan.kernel_name == "d32x16_f64"
    x = signals(3, n, np.float64, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[2:3]))[0], np.asarray(got)[2])
    run_case(n=n, batch=2, centre=False, **kw)



# ------------------------------------------------------------------ n_fft 512, f64: two frames per transform on k_d512 (round 4)
@pytest.mark.parametrize("hop", [256, 128, 64, 160, 148, 150, 260, 2, 147, 149, 259, 1, 255])
@pytest.mark.parametrize("amp,floor,n_mels", [("complex", None, None), ("power", None, None), ("magnitude", None, None), ("db", -80.0, None),
                                              ("power", None, 80), ("db", -80.0, 40), ("magnitude", None, 128)])
def test_tuned_f64_512(hop, amp, floor, n_mels):
    """f64 n_fft 512 (the reference's Mel benchmark shape 512 / 256 in its default type) on k_d512: both staging depths (hop <= 148 / <= 260),
    odd and even frame counts (a slot's second frame may not exist), centre on and off, a signal's bits independent of its batch; odd hops
    too since round 5 (the staged samples are read one by one); hops above 260 stay on the register-tiled kernel."""
    n = 37 * 256 + 77 if hop >= 64 else 2500
    kw = dict(n_fft=512, hop=hop, amp=amp, floor=floor, dtype="float64")
    if n_mels:
        kw.update(n_mels=n_mels, fmin=0.0, fmax=8000.0)
    plan, got = run_case(n=n, batch=3, **kw)
    assert plan.kernel_name == "d512_f64"
    x = signals(3, n, np.float64, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[2:3]))[0], np.asarray(got)[2])
    run_case(n=n + hop, batch=2, centre=False, **kw)
    assert make(512, 262, dtype="float64")[0].kernel_name == "reg_radix"


@pytest.mark.parametrize("n", [1, 5, 255, 256, 257, 511, 512, 513, 767, 768, 769, 8191, 8192, 8193, 8447, 8448, 8449])
@pytest.mark.parametrize("centre", [True, False])
def test_ragged_lengths_f64_512(n, centre):
    if not centre and n < 512:
        n += 512
    run_case(n=n, batch=2, n_fft=512, hop=256, centre=centre, amp="complex", dtype="float64")
    run_case(n=n, batch=3, n_fft=512, hop=128, centre=centre, amp="power", dtype="float64")
    run_case(n=n, batch=2, n_fft=512, hop=256, centre=centre, n_mels=80, fmin=0.0, fmax=8000.0, amp="db", floor=-80.0, dtype="float64")

@pytest.mark.parametrize("n", [1, 5, 511, 512, 513, 1023, 1024, 1025, 1279, 1280, 1281, 4097, 5119, 5120, 5121])
@pytest.mark.parametrize("centre", [True, False])
def test_ragged_lengths_f64_1024(n, centre):
    if not centre and n < 1024:
        n += 1024
    run_case(n=n, batch=2, n_fft=1024, hop=256, centre=centre, amp="complex", dtype="float64")
    run_case(n=n, batch=3, n_fft=1024, hop=256, centre=centre, amp="power", dtype="float64")
    run_case(n=n, batch=2, n_fft=1024, hop=256, centre=centre, n_mels=80, fmin=0.0, fmax=8000.0, amp="db", floor=-80.0, dtype="float64")


@pytest.mark.parametrize("dtype,n_fft,hop,amp,n_mels", [
    ("float64", 1024, 512, "complex", None), ("float64", 1024, 1024, "complex", None), ("float64", 1024, 600, "complex", None),
    ("float64", 1024, 274, "complex", None), ("float64", 1024, 256, "complex", None), ("float64", 1024, 512, "power", None),
    ("float64", 1024, 512, "power", 80), ("float64", 1024, 256, "power", 128),
    ("float32", 2048, 1024, "complex", None), ("float32", 2048, 2048, "power", None), ("float32", 2048, 600, "complex", None),
    ("float32", 2048, 1024, "power", 80), ("float32", 512, 128, "complex", None), ("float32", 512, 256, "complex", None),
    ("float32", 1024, 512, "complex", None), ("float32", 1024, 600, "power", 80), ("float32", 512, 256, "power", 80), ("float32", 512, 256, "power", 128),
    # the generic kernels at the same size: register-tiled (powers of two, composite), chirp-z, f64 complex (16-byte stores)
    ("float32", 4096, 1024, "complex", None), ("float32", 4096, 1024, "power", None), ("float32", 4096, 2048, "power", None), ("float32", 4096, 4096, "complex", None),
    ("float32", 4096, 1024, "power", 80), ("float64", 2048, 512, "complex", None), ("float64", 2048, 1024, "power", None), ("float64", 2048, 2048, "complex", None),
    ("float64", 2048, 512, "power", 80), ("float64", 400, 160, "complex", None),
    ("float32", 400, 160, "complex", None), ("float32", 1009, 252, "complex", None), ("float64", 509, 128, "complex", None),
    ("float64", 512, 128, "power", 40), ("float64", 512, 256, "complex", None), ("float64", 512, 128, "complex", None), ("float64", 512, 256, "power", 80),
    ("float64", 512, 64, "power", None), ("float64", 512, 260, "power", 128)])
def test_tuned_kernels_many_tiles_per_workgroup(dtype, n_fft, hop, amp, n_mels):
    """64 x 10 s through the tuned kernels' per-lane-column (unstaged), complex and filterbank variants and through the generic kernels: several
    tiles per workgroup, every element of every row against the oracle.  Round 4 found 0.06 % of k_d32x16's complex STFT wrong at hop >= 274
    ONLY at this size (a 16-byte store whose data registers were rewritten by the next instruction, kernels_d32x16.hip: emit): the small-shape
    tests run one tile per workgroup."""
    x = H.cfg2_batch(64).astype(np.float64 if dtype == "float64" else np.float32)
    kw = dict(n_fft=n_fft, hop=hop, dtype=dtype, amp=amp)
    if n_mels:
        kw.update(n_mels=n_mels, fmin=0.0, fmax=8000.0)
    plan, op = make(**kw)
    x64 = x.astype(np.float64)
    ref = orc.stft_batch(op, x64, nthreads=orc.max_threads()) if amp == "complex" else orc.spectrogram_batch(op, x64, nthreads=orc.max_threads())
    for rep in range(2):
        got = np.asarray(plan.compute_batch(x))
        err = np.max(np.abs(got - ref))
        assert err <= (1e-10 if dtype == "float64" else 2e-4) * max(1.0, float(np.max(np.abs(ref)))), (rep, err)



# ------------------------------------------------------------------ n_fft 4096, f32: the tuned kernel k_r64x32 (round 4)
@pytest.mark.parametrize("hop", [1024, 512, 2048, 4096, 100, 1170, 1172, 2050, 2, 1023, 1171, 2051, 441, 1])
@pytest.mark.parametrize("amp,floor,n_mels", [("complex", None, None), ("power", None, None), ("magnitude", None, None), ("db", -80.0, None),
                                              ("power", None, 80), ("db", -80.0, 128), ("magnitude", None, 40), ("power", None, 20)])
def test_tuned_4096(hop, amp, floor, n_mels):
    """f32 n_fft 4096 on k_r64x32 (8-frame tiles, half rows of 32 points in lane pairs): both staging depths and the per-lane columns above hop
    2048; per-bin and complex outputs; frame counts that are not multiples of 8; centre on and off; a signal's bits independent of its batch."""
    n = 27 * 1024 + 77 if hop >= 100 else 9000
    kw = dict(n_fft=4096, hop=hop, amp=amp, floor=floor, dtype="float32")
    if n_mels:  # fused band stage up to hop 1170, per-bin power + bank rows above
        kw.update(n_mels=n_mels, fmin=0.0, fmax=8000.0)
    plan, got = run_case(n=n, batch=3, **kw)
    assert plan.kernel_name == "r64x32_f32"
    x = signals(3, n, np.float32, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[2:3]))[0], np.asarray(got)[2])
    run_case(n=n, batch=2, centre=False, **kw)


@pytest.mark.parametrize("n", [1, 5, 2047, 2048, 2049, 4095, 4096, 4097, 5119, 5120, 5121, 12287, 12288, 12289])
@pytest.mark.parametrize("centre", [True, False])
def test_ragged_lengths_4096(n, centre):
    if not centre and n < 4096:
        n += 4096
    run_case(n=n, batch=2, n_fft=4096, hop=1024, centre=centre, amp="complex")
    run_case(n=n, batch=3, n_fft=4096, hop=512, centre=centre, amp="power")
    run_case(n=n, batch=2, n_fft=4096, hop=1024, centre=centre, n_mels=80, fmin=0.0, fmax=8000.0, amp="db", floor=-80.0)


# ------------------------------------------------------------------ n_fft 2048, f64: the tuned kernel k_d32x32 (round 4)
@pytest.mark.parametrize("hop", [1024, 512, 256, 2048, 100, 584, 586, 1026, 2, 511, 585, 1023, 441, 1])
@pytest.mark.parametrize("amp,floor,n_mels", [("complex", None, None), ("power", None, None), ("magnitude", None, None), ("db", -80.0, None),
                                              ("power", None, 80), ("db", -80.0, 128), ("magnitude", None, 40), ("power", None, 20)])
def test_tuned_f64_2048(hop, amp, floor, n_mels):
    """f64 n_fft 2048 (the reference's Criterion shape 2048 / 1024 in its type) on k_d32x32: two lanes per column in pass 1, half rows in lane
    pairs in pass 2; both staging depths and the per-lane columns above hop 1024; frame counts that are not multiples of 8; centre on and off."""
    n = 27 * 512 + 77 if hop >= 100 else 5000
    kw = dict(n_fft=2048, hop=hop, amp=amp, floor=floor, dtype="float64")
    if n_mels:  # fused band stage up to hop 585, per-bin power + bank rows above
        kw.update(n_mels=n_mels, fmin=0.0, fmax=8000.0)
    plan, got = run_case(n=n, batch=3, **kw)
    assert plan.kernel_name == "d32x32_f64"
    x = signals(3, n, np.float64, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[2:3]))[0], np.asarray(got)[2])
    run_case(n=n, batch=2, centre=False, **kw)


@pytest.mark.parametrize("n", [1, 5, 1023, 1024, 1025, 2047, 2048, 2049, 2559, 2560, 2561, 6143, 6144, 6145])
@pytest.mark.parametrize("centre", [True, False])
def test_ragged_lengths_f64_2048(n, centre):
    if not centre and n < 2048:
        n += 2048
    run_case(n=n, batch=2, n_fft=2048, hop=512, centre=centre, amp="complex", dtype="float64")
    run_case(n=n, batch=3, n_fft=2048, hop=1024, centre=centre, amp="power", dtype="float64")
    run_case(n=n, batch=2, n_fft=2048, hop=512, centre=centre, n_mels=80, fmin=0.0, fmax=8000.0, amp="db", floor=-80.0, dtype="float64")

@pytest.mark.parametrize("n", [1, 5, 1023, 1024, 1025, 2047, 2048, 2049, 2559, 2560, 2561, 8193, 10239, 10240, 10241])
@pytest.mark.parametrize("centre", [True, False])
def test_ragged_lengths_2048(n, centre):
    if not centre and n < 2048:
        n += 2048
    run_case(n=n, batch=2, n_fft=2048, hop=512, centre=centre, amp="complex")
    run_case(n=n, batch=2, n_fft=2048, hop=512, centre=centre, n_mels=80, fmin=0.0, fmax=8000.0, amp="db", floor=-80.0)


def test_tuned_2048_full_size():
    """64 x 10 s at n_fft 2048 / hop 512 (the shape of profiles/bench_r04_sweep.txt): oracle on four rows incl. the last; every row
    equal to its own B = 1 launch, bit for bit."""
    torch = pytest.importorskip("torch")
    base = H.cfg2_batch(64)
    x = torch.from_numpy(base).cuda()
    params = sg.SpectrogramParams(sg.StftParams(2048, 512, sg.WindowType.hanning, True), 16000.0)
    for plan, op in ((sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32"), orc.Params(n_fft=2048, hop=512)),
                     (sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32"),
                      orc.Params(n_fft=2048, hop=512, n_mels=80, amp="db", floor_db=-80.0))):
        assert plan.kernel_name == "r32x32_f32"
        out = plan.compute_batch(x)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.all(np.isfinite(got))
        for b in (0, 1, 31, 63):
            ref = orc.spectrogram_batch(op, base[b:b + 1].astype(np.float64))[0]
            if op.amp == "db":
                pw = orc.spectrogram_batch(orc.Params(n_fft=2048, hop=512, n_mels=80), base[b:b + 1].astype(np.float64))[0]
                m = pw > 1e-4 * pw.max()
                assert np.max(np.abs(got[b][m] - ref[m])) < 1e-3
            else:
                m = ref > 1e-4 * ref.max()
                assert np.max(np.abs(got[b][m] - ref[m]) / ref[m]) < 1e-4
            one = plan.compute_batch(x[b:b + 1])
            assert torch.equal(one[0], out[b])


@pytest.mark.parametrize("amp,floor", [("power", None), ("magnitude", None), ("db", -80.0)])
@pytest.mark.parametrize("n_fft,hop,dtype,n_mels,norm", [(4096, 1024, "float32", 80, None), (8192, 2048, "float32", 128, "slaney"),
                                                         (2048, 512, "float64", 80, None), (4096, 1000, "float64", 40, "l1"),
                                                         (1920, 480, "float64", 64, None),
                                                         # other kernels: LDS radix-2 (beyond the register-tiled sizes), two factors (its tile
                                                         # with the |X|^2 rows would not fit: the plan used to fall to the direct sum)
                                                         (16384, 4096, "float32", 80, None), (8192, 2048, "float64", 80, None),
                                                         (3000, 750, "float32", 40, None), (6000, 1500, "float32", 40, None)])
def test_long_frames_filterbank_in_a_second_launch(n_fft, hop, dtype, n_mels, norm, amp, floor):
    """Filterbank outputs at n_fft * sizeof(T) >= 16 KiB on the register-tiled kernel run as two launches (per-bin power to a
    plan-owned tensor, then one wave per (band, 64 frames)): the same terms in the reference's order; frame counts that are not
    multiples of 64, one signal vs the batch, and a second call on the same plan (the tensor is reused)."""
    plan, got = run_case(n=5 * n_fft + 123, batch=3, n_fft=n_fft, hop=hop, n_mels=n_mels, norm=norm, amp=amp, floor=floor, dtype=dtype)
    assert plan.kernel_name == {(16384, "float32"): "lds_radix2", (8192, "float64"): "lds_radix2", (3000, "float32"): "bluestein",
                                (6000, "float32"): "bluestein", (4096, "float32"): "r64x32_f32", (2048, "float64"): "d32x32_f64"}.get((n_fft, dtype), "reg_radix")  # (4096 f32, 2048 f64: per-bin power on the tuned kernels)
    x = signals(3, 5 * n_fft + 123, np.float32 if dtype == "float32" else np.float64, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[1:2]))[0], np.asarray(got)[1])
    assert np.array_equal(np.asarray(plan.compute_batch(x)), np.asarray(got))


@pytest.mark.parametrize("n_fft,hop,dtype,kernel", [(6000, 1500, "float32", "bluestein"), (6000, 2000, "float64", "bluestein"), (8200, 2050, "float32", "bluestein"),
                                                    (6003, 2000, "float64", "big_chirpz"),
                                                    (3000, 700, "float64", "bluestein"), (5003, 2000, "float32", "bluestein"),
                                                    (8191, 2048, "float32", "bluestein"), (4099, 1000, "float64", "big_chirpz")])
@pytest.mark.parametrize("amp", ["complex", "power"])
def test_long_frames_outside_the_register_tiled_lists(n_fft, hop, dtype, kernel, amp):
    """Frames of 2049 ... 8192 samples that are not a listed size: chirp-z with one 8192- / 16384-point sequence per workgroup in
    LDS (f32 up to n_fft 8192, f64 up to 4096 — 128 KiB of LDS either way); even lengths up to twice that take the half-length complex
    form on the same kernels (one frame per sequence of n_fft / 2 points); odd ones past the LDS chirp-z go through global memory
    (round 5, bigfft.hip: chirp-z on four-step transforms; before: the two-factor kernel / the direct sum)."""
    plan, got = run_case(n=3 * n_fft + 77, batch=2, n_fft=n_fft, hop=hop, amp=amp, dtype=dtype)
    assert plan.kernel_name == kernel
    x = signals(2, 3 * n_fft + 77, np.float32 if dtype == "float32" else np.float64, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[1:2]))[0], np.asarray(got)[1])


MIXED = [80, 120, 160, 200, 240, 320, 400, 480, 600, 640, 800, 960, 1000, 1200, 1440, 1600, 1280, 1920, 2000, 2160, 2400, 2560]


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft", MIXED)
@pytest.mark.parametrize("amp", ["complex", "power"])
def test_mixed_radix_sizes(n_fft, amp, dtype):
    """Even composite sizes on the register-tiled kernel (in-register passes with factors 2, 3 and 5)."""
    hop = n_fft * 2 // 5
    plan, _ = run_case(n=max(6000, 4 * n_fft), n_fft=n_fft, hop=hop, amp=amp, dtype=dtype)
    assert plan.kernel_name == "reg_radix"


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop", [(400, 160), (400, 161), (480, 160), (800, 320)])
def test_mixed_radix_mel_and_odd_hop(n_fft, hop, dtype):
    run_case(n=9000, n_fft=n_fft, hop=hop, n_mels=80, fmin=0.0, fmax=8000.0, amp="db", floor=-80.0, dtype=dtype)
    for n in (1, n_fft // 2, n_fft + 1):
        run_case(n=n, batch=2, n_fft=n_fft, hop=hop, amp="power", dtype=dtype)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop", [(64, 16), (512, 128), (2048, 512)])
def test_pow2_short_and_ragged(n_fft, hop, dtype):
    """Signals shorter than one frame, exactly one frame, and lengths that leave a partial last tile."""
    for n in (1, n_fft // 2, n_fft, n_fft + 1, 5 * n_fft + 17):
        for centre in (True, False):
            if not centre and n < n_fft:
                continue
            run_case(n=n, batch=2, n_fft=n_fft, hop=hop, amp="power", dtype=dtype, centre=centre)


# ------------------------------------------------------------------ arbitrary sizes (reference accepts any n_fft)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop", [(1, 1), (2, 1), (3, 2), (7, 3), (10, 10), (100, 33), (400, 160), (1000, 250), (251, 100), (480, 160), (15, 4), (2 * 3 * 5 * 7 * 11, 500)])
def test_non_pow2_sizes(n_fft, hop, dtype):
    window = "rectangular" if n_fft == 1 else "hanning"  # symmetric Hann of length 1 is 0/0 = NaN in the reference too
    run_case(n=3000, n_fft=n_fft, hop=hop, amp="complex", dtype=dtype, window=window)
    run_case(n=3000, n_fft=n_fft, hop=hop, amp="power", dtype=dtype, window=window)


# ------------------------------------------------------------------ windows / centre / amp scales
@pytest.mark.parametrize("window", sorted(WINDOWS))
@pytest.mark.parametrize("centre", [True, False])
def test_windows_and_centre(window, centre):
    run_case(n=5000, n_fft=1024, hop=256, window=window, centre=centre, amp="complex")
    run_case(n=5000, n_fft=512, hop=128, window=window, centre=centre, amp="magnitude", dtype="float64")


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("amp,floor", [("power", None), ("magnitude", None), ("db", -80.0), ("db", -100.0), ("db", None)])
@pytest.mark.parametrize("n_fft,hop", [(1024, 256), (512, 256), (400, 160)])
def test_linear_amp_scales(n_fft, hop, amp, floor, dtype):
    run_case(n=8000, n_fft=n_fft, hop=hop, amp=amp, floor=floor, dtype=dtype)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("amp,floor", [("power", None), ("magnitude", None), ("db", -80.0)])
@pytest.mark.parametrize("n_fft,hop,n_mels,fmin,fmax,norm", [
    (1024, 256, 80, 0.0, 8000.0, None), (1024, 256, 128, 20.0, 7600.0, "slaney"), (512, 256, 40, 0.0, 8000.0, "l1"),
    (400, 160, 64, 0.0, 8000.0, "l2"), (2048, 512, 80, 0.0, 8000.0, None)])
def test_mel(n_fft, hop, n_mels, fmin, fmax, norm, amp, floor, dtype):
    run_case(n=9000, n_fft=n_fft, hop=hop, n_mels=n_mels, fmin=fmin, fmax=fmax, norm=norm, amp=amp, floor=floor,
             dtype=dtype)


@pytest.mark.parametrize("amp,floor", [("power", None), ("magnitude", None), ("db", -80.0)])
@pytest.mark.parametrize("n_mels,norm", [(40, None), (80, None), (128, "slaney"), (24, "l1")])
def test_mel_512_hop256_tuned(n_mels, norm, amp, floor):
    """The reference's Mel benchmark shape n_fft 512 / hop 256 x {40, 80, 128} bands (benches/spectrogram_benchmarks.rs:105-141) on k_r32x16's
    two-frames-per-transform mode with the larger LDS halves of r32x16_layout.h: odd and even frame counts, ragged batch, centre off,
    a signal's bits independent of its batch."""
    kw = dict(n_fft=512, hop=256, n_mels=n_mels, fmin=0.0, fmax=8000.0, norm=norm, amp=amp, floor=floor, dtype="float32")
    plan, got = run_case(n=9000, batch=3, **kw)
    assert plan.kernel_name == "r32x16_f32"
    x = signals(3, 9000, np.float32, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[2:3]))[0], np.asarray(got)[2])
    run_case(n=9100, batch=2, centre=False, **kw)
    run_case(n=300, batch=5, **kw)


@pytest.mark.parametrize("dtype,n_fft,hop,n", [("float64", 1024, 256, 1024), ("float64", 1024, 256, 2000), ("float32", 2048, 512, 1500), ("float32", 2048, 512, 3000)])
def test_short_signal_batches_leave_the_one_signal_tiles(dtype, n_fft, hop, n):
    """Batches of signals of a few frames: the 16-frame tiles of k_d32x16 / k_r32x32 hold ONE signal, so below 8 (5) frames per signal the call
    runs on the register-tiled kernel (same plan, decided per call); results do not depend on which."""
    plan, got = run_case(n=n, batch=33, n_fft=n_fft, hop=hop, amp="power", dtype=dtype)
    x = signals(33, n, np.float32 if dtype == "float32" else np.float64, 0)
    one = np.asarray(plan.compute_batch(x[5:6]))[0]  # batch 1: the tuned kernel
    ref = np.asarray(got)[5]
    assert np.max(np.abs(one - ref)) <= (1e-10 if dtype == "float64" else 2e-4) * max(1.0, float(np.max(np.abs(ref))))
    run_case(n=n, batch=7, n_fft=n_fft, hop=hop, amp="complex", dtype=dtype)
    if dtype == "float64":
        run_case(n=n, batch=7, n_fft=n_fft, hop=hop, n_mels=40, fmin=0.0, fmax=8000.0, amp="db", floor=-80.0, dtype=dtype)


# ------------------------------------------------------------------ golden vectors (reference numpy_impls) through the C ABI
def test_golden_config1_f64(golden_dir):
    g = np.load(os.path.join(golden_dir, "config1_ref.npz"))
    x = np.sin(2.0 * np.pi * 440.0 * np.arange(16000, dtype=np.float64) / 16000.0)
    for n_fft, hop in ((512, 256), (256, 128)):
        params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
        S = sg.compute_stft(x, params)  # dtype default = float64, as in the reference
        assert S.data.dtype == np.complex128
        assert np.max(np.abs(S.data - g[f"c1_{n_fft}_{hop}_stft"])) < 1e-10
        P = sg.compute_linear_power_spectrogram(x, params)
        assert np.max(np.abs(P.data - g[f"c1_{n_fft}_{hop}_power"])) < 1e-10 * g[f"c1_{n_fft}_{hop}_power"].max()
        M = sg.compute_linear_magnitude_spectrogram(x, params)
        assert np.max(np.abs(M.data - g[f"c1_{n_fft}_{hop}_magnitude"])) < 1e-10
        assert np.allclose(P.frequencies, g[f"c1_{n_fft}_{hop}_freqs"])
        assert np.allclose(P.times, np.arange(P.n_frames) * hop / 16000.0)
        pn = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, False), 16000.0)
        Pn = sg.compute_linear_power_spectrogram(x, pn)
        ref = g[f"c1_{n_fft}_{hop}_nocentre_power"]
        assert Pn.shape == ref.shape and np.max(np.abs(Pn.data - ref)) < 1e-10 * ref.max()


@pytest.mark.parametrize("b", [0, 1])
def test_golden_config2_f32(golden_dir, b):
    g = np.load(os.path.join(golden_dir, "config2_ref.npz"))
    x = H.cfg2_signal(b)
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    S = sg.compute_stft(x, params, dtype="float32")
    assert S.shape == (513, 626) and S.data.dtype == np.complex64
    ref = g[f"c2_b{b}_stft"]
    err = np.max(np.abs(S.data[:, g[f"c2_b{b}_frames"]].astype(np.complex128) - ref)) / np.max(np.abs(ref))
    assert err <= GUARD32, err
    P = sg.compute_linear_power_spectrogram(x, params, dtype="float32")
    rs = g[f"c2_b{b}_power_rowsum"]
    assert np.max(np.abs(P.data.astype(np.float64).sum(axis=1) - rs)) <= GUARD32 * rs.max()
    # bin for bin on the stored frames: power = |stft|^2 of the reference's own complex values (S5)
    pref = np.abs(ref) ** 2
    got = P.data[:, g[f"c2_b{b}_frames"]].astype(np.float64)
    assert np.max(np.abs(got - pref)) <= 1e-4 * pref.max()
    near = pref > 1e-4 * pref.max()
    assert np.max(np.abs(got - pref)[near] / pref[near]) <= 1e-4


def test_golden_short_inputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "short_ref.npz"))
    params = sg.SpectrogramParams(sg.StftParams(512, 256, sg.WindowType.hanning, True), 16000.0)
    for n in (5, 300, 511, 512, 513, 1000):
        S = sg.compute_stft(g[f"short_{n}_x"], params)
        ref = g[f"short_{n}_stft"]
        assert S.shape == ref.shape
        assert np.max(np.abs(S.data - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref)))


# ------------------------------------------------------------------ edge cases of the boundary
@pytest.mark.parametrize("n", [1, 5, 255, 256, 257, 1023, 1024, 1025, 1279, 1280, 1281, 4097])
@pytest.mark.parametrize("centre", [True, False])
def test_ragged_lengths_1024(n, centre):
    run_case(n=n, batch=2, n_fft=1024, hop=256, centre=centre, amp="complex")
    run_case(n=n, batch=2, n_fft=1024, hop=256, centre=centre, n_mels=80, amp="db", floor=-80.0)


def test_hop_equals_n_fft_and_odd_hop():
    run_case(n=10000, n_fft=1024, hop=1024, amp="power")
    run_case(n=10000, n_fft=1024, hop=255, amp="complex")  # odd hop: 8-byte loads impossible -> generic kernel
    run_case(n=10001, n_fft=1024, hop=256, amp="complex")  # odd length: last float2 straddles the end


@pytest.mark.parametrize("amp,floor", [("power", None), ("magnitude", None), ("db", -80.0), ("complex", None)])
@pytest.mark.parametrize("n,centre", [(1, True), (127, True), (128, True), (129, True), (512, False), (640, False), (4095, True), (4096, True),
                                      (4224, True), (8321, False), (40000, True)])
def test_tuned_kernel_512_two_frames_per_transform(n, centre, amp, floor):
    """f32 n_fft 512 / hop 128, per-bin outputs: the tuned kernel packs two consecutive frames into one 512-point complex transform.
    Odd and even frame counts (a slot whose second frame does not exist), signals shorter than a frame, exactly one and several
    32-frame tiles, both centring modes, every amplitude scale and the complex STFT."""
    plan, got = run_case(n=n, batch=3, n_fft=512, hop=128, centre=centre, amp=amp, floor=floor, dtype="float32")
    assert plan.kernel_name == "r32x16_f32"
    one = plan.compute_batch(signals(3, n, np.float32, 0)[1:2])
    assert np.array_equal(np.asarray(one)[0], np.asarray(got)[1])


@pytest.mark.parametrize("amp,floor", [("power", None), ("magnitude", None), ("db", -80.0)])
@pytest.mark.parametrize("n,n_mels,fmin,fmax,norm", [(40000, 80, 0.0, 8000.0, None), (4100, 40, 20.0, 7600.0, "slaney"), (129, 128, 0.0, 8000.0, "l1"),
                                                      (8321, 23, 100.0, 4000.0, None), (16000, 80, 0.0, 8000.0, "l2")])
def test_tuned_kernel_512_mel(n, n_mels, fmin, fmax, norm, amp, floor):
    """f32 n_fft 512 / hop 128 filterbank outputs: the two-frames-per-transform mode with the band schedule walked over 32-frame
    tiles (|X|^2 of both frames of a slot written as one LDS pair)."""
    plan, got = run_case(n=n, batch=3, n_fft=512, hop=128, n_mels=n_mels, fmin=fmin, fmax=fmax, norm=norm, amp=amp, floor=floor, dtype="float32")
    assert plan.kernel_name == "r32x16_f32"
    one = plan.compute_batch(signals(3, n, np.float32, 0)[2:3])
    assert np.array_equal(np.asarray(one)[0], np.asarray(got)[2])


@pytest.mark.parametrize("amp,floor", [("power", None), ("db", -80.0), ("complex", None)])
@pytest.mark.parametrize("hop", [64, 160, 256])
@pytest.mark.parametrize("n,centre", [(1, True), (159, True), (160, True), (512, False), (703, False), (2047, True), (4960, True), (5121, True), (21000, False)])
def test_tuned_kernel_512_other_hops(n, centre, hop, amp, floor):
    """The two-frames-per-transform mode at hops 64 and 160 (the staging padding, the pass-1 read offsets and the number of chunk
    rounds depend on the hop): odd / even frame counts, one and several 32-frame tiles, both centring modes."""
    plan, got = run_case(n=n, batch=3, n_fft=512, hop=hop, centre=centre, amp=amp, floor=floor, dtype="float32")
    assert plan.kernel_name == "r32x16_f32"
    one = plan.compute_batch(signals(3, n, np.float32, 0)[1:2])
    assert np.array_equal(np.asarray(one)[0], np.asarray(got)[1])


@pytest.mark.parametrize("amp,floor", [("power", None), ("db", -80.0), ("complex", None), ("magnitude", None)])
@pytest.mark.parametrize("hop", [2, 30, 100, 200, 320, 384, 510, 512])
@pytest.mark.parametrize("n,centre,batch", [(1, True, 1), (703, False, 3), (5121, True, 1), (21000, False, 3), (40001, True, 5)])
def test_tuned_kernel_512_every_even_hop(n, centre, batch, hop, amp, floor):
    """n_fft 512 at an even hop without a staged variant (round 5): the packed form of the two-frames-per-transform mode — per-lane loads,
    tiles that run on into the next signal — whatever the slot fill and for a single signal too; per-bin and complex outputs (filterbanks
    at such hops stay on the register-tiled kernel).  Against the oracle; a signal alone gives the bits of its row in the batch."""
    plan, got = run_case(n=n, batch=batch, n_fft=512, hop=hop, centre=centre, amp=amp, floor=floor, dtype="float32")
    assert plan.kernel_name == "r32x16_f32"
    one = plan.compute_batch(signals(batch, n, np.float32, 0)[batch - 1:batch])
    assert np.array_equal(np.asarray(one)[0], np.asarray(got)[batch - 1])
    assert make(512, hop, n_mels=40, dtype="float32")[0].kernel_name == "reg_radix"
    assert make(512, 101, dtype="float32")[0].kernel_name == "reg_radix"  # odd hops: the register-tiled kernel


@pytest.mark.parametrize("hop", [64, 160])
@pytest.mark.parametrize("n,n_mels,norm,amp,floor", [(40000, 80, None, "db", -80.0), (4100, 40, "slaney", "power", None), (161, 128, "l1", "magnitude", None)])
def test_tuned_kernel_512_mel_other_hops(n, n_mels, norm, amp, floor, hop):
    plan, got = run_case(n=n, batch=3, n_fft=512, hop=hop, n_mels=n_mels, norm=norm, amp=amp, floor=floor, dtype="float32")
    assert plan.kernel_name == "r32x16_f32"
    one = plan.compute_batch(signals(3, n, np.float32, 0)[2:3])
    assert np.array_equal(np.asarray(one)[0], np.asarray(got)[2])


@pytest.mark.parametrize("hop", [2, 66, 130, 258, 270, 272, 274, 510, 1022])
def test_tuned_kernel_even_hops(hop):
    """Every even hop runs on the tuned kernel: staged loads up to hop 272 (a tile of 15 hop + 1024 samples need not be a whole
    number of 16-byte chunks), per-lane loads above; interior, edge and ragged last tiles."""
    run_case(n=9000 + hop, batch=3, n_fft=1024, hop=hop, amp="complex")
    run_case(n=5000, batch=2, n_fft=1024, hop=hop, n_mels=40, amp="power")


@pytest.mark.parametrize("n_fft,hop", [(1024, 256), (512, 128), (512, 160), (1024, 160), (400, 160), (251, 63), (1009, 250), (1023, 255)])  # (the last three: chirp-z)
def test_strided_rows_and_device_path_match_host_path(n_fft, hop):
    torch = pytest.importorskip("torch")
    plan, op = make(n_fft, hop, n_mels=80, amp="db", floor=-80.0)
    x = signals(5, 7000, np.float32, 3)
    host = plan.compute_batch(x)
    big = torch.zeros((5, 7424), dtype=torch.float32, device="cuda")
    big[:, :7000] = torch.from_numpy(x).cuda()
    dev = plan.compute_batch(big[:, :7000])  # row stride 7424 != n_samples
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy(), host)
    # odd row stride and a base that is only 4-byte aligned: the same kernel (its buffer loads need no more), the same bits
    odd = torch.zeros((5 * 7001 + 1,), dtype=torch.float32, device="cuda")[1:].view(5, 7001)
    odd[:, :7000] = torch.from_numpy(x).cuda()
    dev2 = plan.compute_batch(odd[:, :7000]).cpu().numpy()
    assert np.array_equal(dev2, host)
    ref = orc.spectrogram_batch(op, x.astype(np.float64))
    pw = orc.spectrogram_batch(orc.Params(n_fft=n_fft, hop=hop, n_mels=80), x.astype(np.float64))
    near = pw > 1e-4 * pw.max()
    assert np.max(np.abs(dev2 - ref)[near]) < 1e-3  # dB, the normal bound within 40 dB of the peak
    sp, _ = make(n_fft, hop, amp="complex")
    a = sp.compute_batch(big[:, :7000]).cpu().numpy()
    b = sp.compute_batch(odd[:, :7000]).cpu().numpy()
    assert np.array_equal(a, b)


def test_dimension_mismatch_and_r2c():
    plan, _ = make(1024, 256)
    x = signals(2, 5000, np.float32)
    with pytest.raises(sg.DimensionMismatchError):
        plan.compute_batch(x, out=np.empty((2, 513, 5), np.float32))
    # conforming R2cPlan::process: DC of ones = N (fft_backend.rs:1880-1907), [1,1,1] padded -> 3 (fft_padding_tests.rs:149-158)
    for dtype, tol in (("float32", 1e-5), ("float64", 1e-12)):
        for n in (8, 1024, 2048, 400):  # (1024: the tuned f32 and f64 kernels, 2048: the tuned f32 kernel — with a window of ones)
            p, _ = make(n, max(1, n // 4), dtype=dtype)
            X = p.r2c(np.ones(n))
            assert abs(X[0] - n) < tol * n and np.max(np.abs(X[1:])) < tol * n
            rng = np.random.default_rng(n)
            v = rng.standard_normal(n)
            assert np.max(np.abs(p.r2c(v) - np.fft.rfft(v))) < tol * 40
        p8, _ = make(8, 4, dtype=dtype)
        v = np.zeros(8)
        v[:3] = 1
        assert abs(p8.r2c(v)[0] - 3.0) < tol
        with pytest.raises(sg.DimensionMismatchError):
            p8.r2c(np.ones(9))


def test_plan_reuse_and_one_shot_agree():
    # tests/stft_plan_tests.rs:60-82: plan == one-shot
    params = sg.SpectrogramParams(sg.StftParams(512, 256, sg.WindowType.hanning, True), 16000.0)
    x = signals(1, 16000, np.float64)[0]
    plan = sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(40, 0.0, 8000.0), sg.LogParams(-80.0))
    a = plan.compute(x).data
    b = plan.compute(x).data
    c = sg.compute_mel_db_spectrogram(x, params, sg.MelParams(40, 0.0, 8000.0), sg.LogParams(-80.0)).data
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert a.min() >= -80.0
    fr = plan.compute_frame(x, 7)
    assert np.max(np.abs(fr - a[:, 7])) < 1e-9


def test_f32_tone_peak_bin():
    # tests/f32_smoke_tests.rs:28-50
    x = np.sin(2 * np.pi * np.arange(4096) / 8.0).astype(np.float32)
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    P = sg.compute_linear_power_spectrogram(x, params, dtype="float32").data
    assert abs(int(np.argmax(P[:, P.shape[1] // 2])) - 128) <= 1


# ------------------------------------------------------------------ BASELINE full sizes: whole-output parity + properties
@pytest.fixture(scope="module")
def cfg2_x():
    return H.cfg2_batch(256)


def test_config2_full_size_linear_power(cfg2_x):
    plan, op = make(1024, 256)
    got = plan.compute_batch(cfg2_x)
    assert got.shape == (256, 513, 626)
    ref32 = orc.spectrogram_batch(op, cfg2_x, nthreads=orc.max_threads())  # the f32 CPU restatement, all rows
    scale = ref32.max(axis=(1, 2), keepdims=True)
    assert np.max(np.abs(got - ref32) / scale) <= GUARD32
    # Parseval per frame: sum_k c_k |X_k|^2 = N * sum_n (x_n w_n)^2  (size-independent property)
    w = orc.make_window("hanning", 1024).astype(np.float32)
    for b in (0, 1, 128, 255):
        fr = H.np_frames(cfg2_x[b], 1024, 256, True).astype(np.float64) * w[None, :].astype(np.float64)
        lhs = 2.0 * got[b].astype(np.float64).sum(axis=0) - got[b, 0] - got[b, 512]
        rhs = 1024.0 * (fr ** 2).sum(axis=1)
        assert np.max(np.abs(lhs - rhs)) <= 1e-4 * rhs.max()


def test_config3_full_size_mel_db(cfg2_x):
    plan, op = make(1024, 256, n_mels=80, amp="db", floor=-80.0)
    got = plan.compute_batch(cfg2_x)
    assert got.shape == (256, 80, 626) and got.min() >= -80.0
    ref = orc.spectrogram_batch(op, cfg2_x.astype(np.float64), nthreads=orc.max_threads())
    pop = orc.Params(n_fft=1024, hop=256, n_mels=80)
    pw = orc.spectrogram_batch(pop, cfg2_x.astype(np.float64), nthreads=orc.max_threads())
    m = pw > 1e-4 * pw.max(axis=(1, 2), keepdims=True)
    assert np.max(np.abs(got[m] - ref[m])) <= 1e-3
    mp, _ = make(1024, 256, n_mels=80)
    gp = mp.compute_batch(cfg2_x)
    rel = np.abs(gp[m] - pw[m]) / pw[m]
    assert rel.max() <= TOL32
    m6 = pw > 1e-6 * pw.max(axis=(1, 2), keepdims=True)
    assert np.max(np.abs(got[m6] - ref[m6])) <= 0.05 and (np.abs(gp[m6] - pw[m6]) / pw[m6]).max() <= 5e-3


def test_linearity_full_size(cfg2_x):
    sp, _ = make(1024, 256, amp="complex")
    a, b = cfg2_x[:8], cfg2_x[8:16]
    Sa, Sb = sp.compute_batch(a), sp.compute_batch(b)
    Sab = sp.compute_batch((0.5 * a + 0.25 * b).astype(np.float32))
    ref = 0.5 * Sa.astype(np.complex128) + 0.25 * Sb.astype(np.complex128)
    assert np.max(np.abs(Sab - ref)) <= GUARD32 * np.max(np.abs(ref))


@pytest.mark.parametrize("n_fft,hop,dtype", [(400, 160, "float32"), (512, 128, "float32"), (400, 160, "float64"), (2048, 512, "float32"),
                                             (1024, 256, "float64"), (512, 160, "float32"), (2000, 500, "float32")])
def test_full_size_register_tiled_kernel(cfg2_x, n_fft, hop, dtype):
    """256 x 10 s through the persistent register-tiled kernel (more tiles than resident workgroups): whole-output parity with
    the CPU restatement in the same precision, plus Parseval per frame."""
    plan, op = make(n_fft, hop, dtype=dtype)
    x = cfg2_x if dtype == "float32" else cfg2_x.astype(np.float64)
    got = plan.compute_batch(x)
    # (f32 512 at hops 64 / 128 / 160 takes the tuned kernel's two-frames-per-transform mode)
    assert plan.kernel_name == {(512, "float32"): "r32x16_f32", (2048, "float32"): "r32x32_f32", (1024, "float64"): "d32x16_f64", (512, "float64"): "d512_f64"}.get((n_fft, dtype), "reg_radix")
    nf = (160000 + 2 * (n_fft // 2) - n_fft) // hop + 1
    assert got.shape == (256, n_fft // 2 + 1, nf)
    ref = orc.spectrogram_batch(op, x, nthreads=orc.max_threads())
    scale = ref.max(axis=(1, 2), keepdims=True)
    assert np.max(np.abs(got - ref) / scale) <= (GUARD32 if dtype == "float32" else 1e-12)
    w = orc.make_window("hanning", n_fft).astype(np.float64)
    for b in (0, 1, 255):
        fr = H.np_frames(x[b], n_fft, hop, True).astype(np.float64) * w[None, :]
        lhs = 2.0 * got[b].astype(np.float64).sum(axis=0) - got[b, 0] - got[b, n_fft // 2]
        rhs = float(n_fft) * (fr ** 2).sum(axis=1)
        assert np.max(np.abs(lhs - rhs)) <= (1e-4 if dtype == "float32" else 1e-10) * rhs.max()


@pytest.mark.parametrize("n_fft,hop,dtype,batch", [(400, 160, "float32", 37), (1024, 256, "float64", 19), (512, 128, "float64", 23)])
def test_register_tiled_mel_in_parts_ragged_rounds(cfg2_x, n_fft, hop, dtype, batch):
    """Filterbank outputs of the register-tiled kernel: the |X|^2 rows of a tile are produced in parts (f32 n_fft 400: 32-frame
    tiles, two parts), and a batch whose tile count is not a multiple of the grid ends in a partial round of the XCD-ordered
    tile walk.  Whole-output parity with the CPU restatement, and every utterance equals its own single-utterance launch."""
    plan, op = make(n_fft, hop, dtype=dtype, n_mels=80, fmin=0.0, fmax=8000.0, amp="db", floor=-80.0)
    x = cfg2_x[:batch] if dtype == "float32" else cfg2_x[:batch].astype(np.float64)
    got = plan.compute_batch(x)
    assert plan.kernel_name == {(1024, "float64"): "d32x16_f64", (512, "float64"): "d512_f64"}.get((n_fft, dtype), "reg_radix")  # (f64 1024 / 512: the tuned kernels, round 4)
    ref = orc.spectrogram_batch(op, x.astype(np.float64), nthreads=orc.max_threads())
    pop = orc.Params(**{**op.__dict__, "amp": "power", "floor_db": None, "_keep": []})
    check(got, ref, "db", dtype, -80.0, orc.spectrogram_batch(pop, x.astype(np.float64), nthreads=orc.max_threads()))
    for b in (0, batch // 2, batch - 1):
        assert np.array_equal(plan.compute_batch(x[b:b + 1])[0], got[b])


def test_fuzz_shapes():
    """Seeded random sweep over (n_fft, hop, centre, window, length, dtype, output) — every kernel family (tuned, register-tiled
    power-of-two / mixed radix, LDS radix-2, two-factor, direct) against the oracle, including hops that do not divide n_fft,
    odd hops (unaligned frames), signals shorter than a frame and lengths that leave partial tiles."""
    rng = np.random.default_rng(int(os.environ.get("SGX_FUZZ_SEED", 20260)))
    pool = [4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048] + MIXED + [6, 10, 30, 50, 100, 250, 330, 441, 97, 127, 509, 1006, 1009, 7, 11, 13, 3000]
    seen = set()
    for case in range(90):
        n_fft = int(pool[rng.integers(len(pool))])
        hop = int(rng.integers(1, n_fft + 1))
        centre = bool(rng.integers(2))
        window = sorted(WINDOWS)[rng.integers(len(WINDOWS))]
        dtype = ["float32", "float64"][rng.integers(2)]
        kind = rng.integers(4)
        n = int(rng.integers(1, 6 * n_fft + 50))
        if not centre and n < n_fft:
            n = n_fft + int(rng.integers(0, 3 * n_fft))
        kw = dict(n_fft=n_fft, hop=hop, centre=centre, window=window, dtype=dtype)
        if kind == 0:
            kw["amp"] = "complex"
        elif kind == 1:
            kw["amp"] = "power"
        elif kind == 2:
            kw.update(amp="db", floor=-80.0)
        else:
            if n_fft < 32:
                kw["amp"] = "magnitude"
            else:
                kw.update(n_mels=int(rng.integers(4, 41)), fmin=0.0, fmax=8000.0, amp="power")
        plan, _ = run_case(n=n, batch=int(rng.integers(1, 4)), seed=case, **kw)
        seen.add(plan.kernel_name)
    assert "SGX_FUZZ_SEED" in os.environ or {"reg_radix", "direct_dft", "lds_radix2", "bluestein"} <= seen, seen  # (two-factor: its own tests above; the coverage claim is the default sweep's)


def test_fuzz_tuned_kernels():
    """Seeded random sweep over the six (n_fft, type) pairs with shape-specific kernels, EVERY hop from 1 to n_fft (round 5: odd hops and
    f32 512 at unlisted hops stay on the tuned kernels; the few that do not — f32 512 odd, f64 512 above 260, f64 2048 odd above 1024,
    filterbanks at f32 512's unlisted hops — run the register-tiled kernel through the same checks): every staging depth and the
    per-lane-column variants, windows, centre on / off, signals from shorter than a frame to dozens of tiles, batches up to several
    tiles per workgroup, every output mode — against the oracle; the batch's last signal against its own launch, bit for bit.
    SGX_FUZZ_SEED draws another sweep (tools: hunting runs on the GPU box)."""
    rng = np.random.default_rng(int(os.environ.get("SGX_FUZZ_SEED", 424242)))
    shapes = [("float32", 512), ("float32", 1024), ("float32", 2048), ("float32", 4096), ("float64", 512), ("float64", 1024), ("float64", 2048)]
    seen = set()
    for case in range(84):
        dtype, n_fft = shapes[case % len(shapes)]
        hop = int(rng.integers(1, n_fft + 1)) if rng.integers(4) else int(rng.choice([n_fft // 8, n_fft // 4, n_fft // 2, 160]))
        centre = bool(rng.integers(2))
        window = sorted(WINDOWS)[rng.integers(len(WINDOWS))]
        n = int(rng.integers(1, 60 * n_fft)) if rng.integers(3) else int(rng.integers(1, 3 * n_fft))
        if not centre and n < n_fft:
            n = n_fft + int(rng.integers(0, 3 * n_fft))
        batch = int(rng.choice([1, 2, 3, 9, 40]))
        n = max(1 if centre else n_fft, min(n, hop * max(2, 12_000_000 // (batch * (n_fft // 2 + 1)))))  # (a short hop: bounded output)
        kind = rng.integers(4)
        kw = dict(n_fft=n_fft, hop=hop, centre=centre, window=window, dtype=dtype)
        if kind == 0:
            kw["amp"] = "complex"
        elif kind == 1:
            kw["amp"] = "power"
        elif kind == 2:
            kw.update(amp="db", floor=-80.0)
        else:
            kw.update(n_mels=int(rng.choice([24, 40, 80, 128])), fmin=0.0, fmax=8000.0, amp=str(rng.choice(["power", "magnitude"])))
        plan, got = run_case(n=n, batch=batch, seed=1000 + case, **kw)
        seen.add(plan.kernel_name)
        x = signals(batch, n, np.float32 if dtype == "float32" else np.float64, 1000 + case)
        assert np.array_equal(np.asarray(plan.compute_batch(x[batch - 1:]))[0], np.asarray(got)[batch - 1]), (case, kw, n, batch)
    assert "SGX_FUZZ_SEED" in os.environ or {"r32x16_f32", "r32x32_f32", "r64x32_f32", "d512_f64", "d32x16_f64", "d32x32_f64"} <= seen, seen


def test_fuzz_chirpz_lengths():
    """Seeded sweep over frame lengths that take the chirp-z kernel (primes, 2 x prime, odd composites, unlisted even sizes; every
    convolution length from 64 to 16384): random hop, centre, window, signal length (shorter than a frame included), batch, output
    mode, both dtypes — against the oracle; a batch's last signal against its own launch."""
    rng = np.random.default_rng(int(os.environ.get("SGX_FUZZ_SEED", 31337)))
    pool = [17, 23, 37, 61, 97, 127, 129, 251, 257, 509, 521, 1009, 1021, 1023, 1025, 2003, 2039, 2049, 3000, 4093, 100, 441, 98, 1006, 2900]
    seen_m = set()
    for case in range(60):
        n_fft = int(pool[rng.integers(len(pool))])
        dtype = ["float32", "float64"][rng.integers(2)]
        if case % 15 == 14:
            n_fft, dtype = [5003, 6000, 8191][(case // 15) % 3], "float32"
        hop = int(rng.integers(1, n_fft + 1)) if rng.integers(3) else max(1, n_fft // 4)
        centre = bool(rng.integers(2))
        window = sorted(WINDOWS)[rng.integers(len(WINDOWS))]
        n = int(rng.integers(1, 5 * n_fft + 50))
        if not centre and n < n_fft:
            n = n_fft + int(rng.integers(0, 3 * n_fft))
        n = min(n, 40000 + 2 * n_fft)  # (the oracle's prime lengths are O(n^2))
        hop = max(hop, (n + n_fft) // 400 + 1)  # at most ~400 frames per signal
        batch = int(rng.integers(1, 5))
        kind = rng.integers(4)
        kw = dict(n_fft=n_fft, hop=hop, centre=centre, window=window, dtype=dtype)
        if kind == 0:
            kw["amp"] = "complex"
        elif kind == 1:
            kw["amp"] = "power"
        elif kind == 2:
            kw.update(amp="db", floor=-80.0)
        else:
            kw.update(n_mels=int(rng.integers(4, 41)), fmin=0.0, fmax=8000.0, amp="power")
        plan, got = run_case(n=n, batch=batch, seed=1000 + case, **kw)
        # the plan's own cost predicate (plan.hip, SGX_BS_COST / SGX_BS_COST_BANK32): chirp-z where the direct sum (primes: n / 2
        # multiply-adds per sample) or the two-factor kernel (a + b / 2) costs more than cost * log2(M) * M / n — short f32 frames
        # into a filterbank (n_fft 17 Mel: 14.7 > 8.5) stay on the direct sum, whatever the seed draws
        m, l2 = 1, 0
        while m < 2 * n_fft - 1:
            m, l2 = 2 * m, l2 + 1
        fa = max([d for d in range(1, int(n_fft ** 0.5) + 1) if n_fft % d == 0])
        per_sample = n_fft / 2.0 if fa == 1 else fa + (n_fft // fa) / 2.0
        cost = 0.65 if ("n_mels" in kw and dtype == "float32") else 0.25
        if per_sample > cost * l2 * m / n_fft or n_fft > 4096:
            assert plan.kernel_name == "bluestein", (n_fft, dtype, kw)
        else:
            assert plan.kernel_name in ("bluestein", "direct_dft", "two_factor_dft"), (plan.kernel_name, n_fft, dtype, kw)
        x = signals(batch, n, np.float32 if dtype == "float32" else np.float64, 1000 + case)
        assert np.array_equal(np.asarray(plan.compute_batch(x[batch - 1:]))[0], np.asarray(got)[batch - 1]), (n_fft, dtype, kw)
        seen_m.add(m)
    assert "SGX_FUZZ_SEED" in os.environ or {64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384} <= seen_m, sorted(seen_m)


def test_config4_shard_full_size_mel_power():
    """BASELINE configs[3], one GPU's shard: 1024 x 10 s utterances, Mel-80 power, ONE launch (40 960 tiles: a different
    persistent-grid regime from configs 2/3).  Oracle on eight rows including the last; over the whole output: finite,
    non-negative, and every utterance equal — bit for bit — to its own B = 1 launch (the result of an utterance must not depend
    on where it sits in the batch or on what its neighbours are)."""
    torch = pytest.importorskip("torch")
    B = 1024
    base = H.cfg2_batch(256)
    x = torch.from_numpy(base).cuda().repeat(4, 1)
    x[256:] += 1e-3 * torch.arange(B - 256, device="cuda", dtype=torch.float32)[:, None] / B  # the repeats are not copies
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    plan = sg.SpectrogramPlanner().mel_power_plan(params, sg.MelParams(80, 0.0, 8000.0), dtype="float32")
    out = plan.compute_batch(x)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (B, 80, 626)
    assert bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0
    rows = [0, 1, 255, 256, 511, 777, 1022, 1023]
    ref = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=256, n_mels=80), x[rows].cpu().numpy().astype(np.float64))
    got = out[rows].cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 1e-4 * ref.max()
    near = ref > 1e-4 * ref.max()
    assert np.max(np.abs(got - ref)[near] / ref[near]) <= 1e-4
    for r in (0, 313, 1023):
        one = plan.compute_batch(x[r:r + 1])
        assert torch.equal(one[0], out[r])
    # sharding is by utterance with no halo: two half batches give the same bits as the whole
    lo, hi = plan.compute_batch(x[:512]), plan.compute_batch(x[512:])
    assert torch.equal(torch.cat([lo, hi]), out)


# ------------------------------------------------------------------ tiles that continue into the next signal (batches of short signals)
@pytest.mark.parametrize("amp,n_mels,floor", [("power", 0, None), ("complex", 0, None), ("db", 80, -80.0), ("magnitude", 40, None)])
@pytest.mark.parametrize("n,centre", [(1, True), (300, True), (700, True), (1000, True), (1279, True), (1024, False), (1800, False),
                                       (4000, True), (5120, True)])
def test_packed_tiles_short_signals(n, centre, amp, n_mels, floor):
    """n_fft 1024 / hop 256 over many short signals: 1 ... 21 frames per signal, so a 16-frame tile of the tuned kernel holds frames
    of up to 16 different signals (the reference's per-frame loop knows no tiles, src/spectrogram.rs:240-294, 1230-1250).  Every
    signal of the batch must come out bit for bit as from its own B = 1 launch, and agree with the oracle; odd lengths put the row
    end inside a sample pair, odd batch sizes leave the last tile partly empty."""
    batch = 37
    plan, got = run_case(n=n, batch=batch, seed=n, n_fft=1024, hop=256, centre=centre, amp=amp, n_mels=n_mels, floor=floor)
    assert plan.kernel_name == "r32x16_f32"
    x = signals(batch, n, np.float32, n)
    for b in (0, 1, 17, 35, 36):
        one = plan.compute_batch(x[b:b + 1])
        assert np.array_equal(np.asarray(one[0]), np.asarray(got[b])), b


def test_packed_tiles_large_batch_linear():
    """65 536 signals of 4 frames (the shape of profiles/bench_r02_sweep_misc.txt's slowest rows) through one launch: spot-checked
    against the oracle and against B = 1 launches, the whole output finite; a row stride larger than the row length."""
    import torch

    batch, n, stride = 65536, 1000, 1008
    rng = np.random.default_rng(5)
    buf = torch.zeros((batch, stride), dtype=torch.float32, device="cuda")
    xh = (0.2 * rng.standard_normal((batch, n))).astype(np.float32)
    buf[:, :n] = torch.from_numpy(xh).cuda()
    buf[:, n:] = 1e30  # the stride padding must never be read as samples
    plan, op = make(n_fft=1024, hop=256)
    out = plan.compute_batch(buf[:, :n])
    torch.cuda.synchronize()
    assert tuple(out.shape) == (batch, 513, 4) and bool(torch.isfinite(out).all())
    idx = [0, 1, 2, 3, 4, 5, 4095, 4096, 32767, 65534, 65535]
    ref = orc.spectrogram_batch(op, xh[idx].astype(np.float64))
    check(out[idx].cpu().numpy(), ref, "power", "float32")
    for b in (3, 65535):
        one = plan.compute_batch(xh[b:b + 1])
        assert np.array_equal(one[0], out[b].cpu().numpy())


# ------------------------------------------------------------------ lengths with a large prime factor: chirp-z (Bluestein)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n_fft,hop", [(251, 100), (401, 160), (509, 127), (1006, 251), (1009, 256), (2003, 500), (4093, 1024)])
def test_bluestein_sizes(n_fft, hop, dtype):
    """Primes and 2 x prime through the chirp-z path (the reference's RustFFT plans these with Rader / Bluestein,
    src/fft_backend.rs:376-385): complex, power, dB and Mel outputs against the oracle at the usual tolerances, the batch bit for
    bit equal to single-signal launches, centred and not."""
    for amp, extra in (("complex", {}), ("power", {}), ("db", {"floor": -80.0}), ("power", {"n_mels": 40})):
        plan, got = run_case(n=5 * n_fft + 321, batch=3, n_fft=n_fft, hop=hop, amp=amp, dtype=dtype, **extra)
        assert plan.kernel_name == "bluestein"
        x = signals(3, 5 * n_fft + 321, np.float32 if dtype == "float32" else np.float64, 0)
        assert np.array_equal(np.asarray(plan.compute_batch(x[2:3]))[0], np.asarray(got)[2])
    run_case(n=3 * n_fft, batch=2, n_fft=n_fft, hop=hop, amp="power", dtype=dtype, centre=False, window="kaiser")
    # the conforming per-frame R2cPlan::process on the same plan kind
    p1, _ = make(n_fft, hop, amp="complex", dtype=dtype)
    fr = signals(1, n_fft, np.float32 if dtype == "float32" else np.float64, 3)[0]
    ref = np.fft.rfft(fr.astype(np.float64))
    err = np.max(np.abs(p1.r2c(fr) - ref)) / np.max(np.abs(ref))
    assert err < (2e-5 if dtype == "float32" else 1e-10), err


def test_bluestein_many_frames_one_launch():
    """64 x 10 s at n_fft 1009 / hop 16 = 640 k frames (320 k frame pairs) through the one-kernel chirp-z path (M = 2048: the
    sequence stays in LDS, one launch whatever the frame count); first, a middle and the last signal against the oracle, a signal
    against its own launch."""
    n_fft, hop, batch = 1009, 16, 64
    plan, op = make(n_fft, hop)
    x = H.cfg2_batch(batch)
    got = plan.compute_batch(x)
    assert plan.kernel_name == "bluestein" and got.shape == (batch, 505, 10000)
    idx = [0, 3, 31, 63]
    ref = orc.spectrogram_batch(op, x[idx].astype(np.float64), nthreads=orc.max_threads())
    check(got[idx], ref, "power", "float32")
    assert np.array_equal(plan.compute_batch(x[3:4])[0], got[3])


def test_bluestein_long_frames_many_pairs():
    """n_fft 4093 (M = 8192: one frame pair per workgroup, 64 KiB of LDS, 32-point first and last pass), 3 x 10 s at hop 12 = 20 001
    pairs.  The last signal against its own launch bit for bit; frames of every signal against a float64 rfft of the same
    windowed samples."""
    n_fft, hop, batch, n = 4093, 12, 3, 160000
    plan, op = make(n_fft, hop)
    x = signals(batch, n, np.float32, 5)
    got = plan.compute_batch(x)
    nf = n // hop + 1
    assert plan.kernel_name == "bluestein" and got.shape == (batch, n_fft // 2 + 1, nf)
    assert np.array_equal(plan.compute_batch(x[2:3])[0], got[2])
    w = orc.make_window("hanning", n_fft)
    pad = n_fft // 2
    xp = np.pad(x.astype(np.float64), ((0, 0), (pad, pad)))
    for b, f in ((0, 0), (0, 1), (1, nf - 1), (2, 6099), (2, 6100), (2, nf - 2), (2, nf - 1)):
        fr = (xp[b, f * hop:f * hop + n_fft].astype(np.float32) * w.astype(np.float32)).astype(np.float64)
        ref = np.abs(np.fft.rfft(fr)) ** 2
        assert np.max(np.abs(got[b, :, f] - ref)) <= GUARD32 * max(1.0, float(ref.max())), (b, f)


@pytest.mark.parametrize("amp", ["power", "complex", "db"])
@pytest.mark.parametrize("n,hop", [(1, 128), (100, 128), (400, 128), (600, 128), (1000, 128), (1500, 64), (700, 160), (900, 256)])
def test_packed_tiles_short_signals_n512(n, hop, amp):
    """The n_fft 512 mode (two frames per transform) over many short signals: a tile packs SLOTS — the frame pairs (2p, 2p + 1) of one
    signal — of consecutive signals, so the pairing of every frame is that of a single-signal launch and every signal must come
    out bit for bit as from its own B = 1 launch; against the oracle at the usual tolerances."""
    batch = 41
    floor = -80.0 if amp == "db" else None
    plan, got = run_case(n=n, batch=batch, seed=n + hop, n_fft=512, hop=hop, amp=amp, floor=floor)
    assert plan.kernel_name == "r32x16_f32"
    x = signals(batch, n, np.float32, n + hop)
    for b in (0, 1, 20, 39, 40):
        one = np.asarray(plan.compute_batch(x[b:b + 1])[0])
        assert np.array_equal(one, np.asarray(got[b])), b


@pytest.mark.parametrize("amp,n_mels,floor", [("power", 0, None), ("complex", 0, None), ("db", 80, -80.0)])
@pytest.mark.parametrize("hop,centre", [(255, True), (257, False), (441, True), (1, True), (1023, False), (271, True), (273, True), (101, False)])
def test_odd_hops_on_the_tuned_kernel(hop, centre, amp, n_mels, floor):
    """Odd hops at n_fft 1024 (44.1 kHz / 10 ms = 441 samples is one): odd frames start on odd samples, so their sample pairs sit at
    4-byte-aligned LDS addresses (read with ds_read2_b32 in the staged path) and, in the direct path, the pair (x[-1], x[0]) straddles
    the row start.  Staged (hop <= 272) and direct (above) paths, centred and not, against the oracle; batch vs single-signal bits."""
    n = 3000 if hop == 1 else 20011
    plan, got = run_case(n=n, batch=3, n_fft=1024, hop=hop, centre=centre, amp=amp, n_mels=n_mels, floor=floor)
    assert plan.kernel_name == "r32x16_f32"
    x = signals(3, n, np.float32, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[1:2]))[0], np.asarray(got)[1])


@pytest.mark.parametrize("amp,n_mels,floor", [("power", 0, None), ("complex", 0, None), ("db", 80, -80.0)])
@pytest.mark.parametrize("hop,centre", [(255, True), (257, False), (441, True), (1, True), (1023, False), (271, True), (273, True), (275, False), (101, False)])
def test_odd_hops_on_the_tuned_f64_kernel(hop, centre, amp, n_mels, floor):
    """Round 5: odd hops at n_fft 1024 in f64 stay on k_d32x16 — staged path (hop <= 272): an odd frame's pairs sit at 8-byte-aligned LDS
    addresses and are read as two doubles; direct path (above): the pair (x[-1], x[0]) straddles the row start and x[0] is put back."""
    n = 3000 if hop == 1 else 20011
    plan, got = run_case(n=n, batch=3, n_fft=1024, hop=hop, centre=centre, amp=amp, n_mels=n_mels, floor=floor, dtype="float64")
    assert plan.kernel_name == "d32x16_f64"
    x = signals(3, n, np.float64, 0)
    assert np.array_equal(np.asarray(plan.compute_batch(x[1:2]))[0], np.asarray(got)[1])


@pytest.mark.gpu
def test_f64_db_epilogue():
    """f64 dB outputs use db_f64.h (frexp + atanh series) instead of libm's log10: against the oracle's `10 * log10` over 600 dB of
    dynamic range (powers 1e-300 .. 1e300, a floor below them), on the tuned f64 kernel (n_fft 1024), the register-tiled one (256) and
    through a filterbank; exact at the floor."""
    rng = np.random.default_rng(8)
    scales = 10.0 ** np.linspace(-150, 150, 31)
    for n_fft, hop in ((1024, 256), (256, 64)):
        x = rng.standard_normal((31, 6 * n_fft)) * scales[:, None]
        params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
        D = sg.SpectrogramPlanner().linear_db_plan(params, sg.LogParams(-3050.0), dtype="float64").compute_batch(x)
        ref = orc.spectrogram_batch(orc.Params(n_fft=n_fft, hop=hop, amp="db", floor_db=-3050.0), x)
        assert np.all(np.isfinite(D)) and np.max(np.abs(D - ref)) < 1e-9, np.max(np.abs(D - ref))
        assert D.min() < -2900 and D.max() > 2900
    x = rng.standard_normal((4, 8000)) * 1e-3
    params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
    M = sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-20.0), dtype="float64").compute_batch(x)
    refm = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=256, n_mels=80, amp="db", floor_db=-20.0), x)
    assert np.max(np.abs(M - refm)) < 1e-9 and np.min(M) < -20.0 + 1e-9  # (some bands sit at the floor)
