"""CPU tests: the C-ABI library loads, exports every symbol include/spectro_hip.h declares, and its host logic
(validation, window / Mel tables, framing, axes, sharding) agrees with the oracle.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import spectrograms_amd as sg
from oracle import oracle as orc
from spectrograms_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = _ffi.DEVICE_HOST_ONLY


def host_plan(n_fft=512, hop=256, window=None, centre=True, sr=16000.0, mel=None, amp=_ffi.AMP_POWER, db=None,
              dtype="float64"):
    p = sg.SpectrogramParams(sg.StftParams(n_fft, hop, window or sg.WindowType.hanning, centre), sr)
    return sg.Plan(p, amp, mel, db, dtype, device=HOST)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "spectro_hip.h")).read()
    declared = set(re.findall(r"\b(sgx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sgx_plan", "sgx_comm", "sgx_fft2d", "sgx_c2c"}
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    L = C.CDLL(_ffi.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert _ffi.lib().sgx_abi_version() == 7


def test_params_struct_layout_matches_header():
    # field order / types of sgx_params as declared in the header
    hdr = open(os.path.join(ROOT, "include", "spectro_hip.h")).read()
    body = hdr[hdr.index("typedef struct {"):hdr.index("} sgx_params;")]
    names = re.findall(r"\b(?:uint32_t|int32_t|double|const double \*)\s*\*?([a-z_0-9, ]+);", body)
    flat = [n.strip() for grp in names for n in grp.split(",")]
    assert flat == [f[0] for f in _ffi.SgxParams._fields_]


@pytest.mark.parametrize("kind,param", [("rectangular", 0), ("hanning", 0), ("hamming", 0), ("blackman", 0),
                                        ("kaiser", 5.0), ("kaiser", 2.5), ("gaussian", 100.0)])
@pytest.mark.parametrize("n", [1, 2, 8, 400, 1024])
def test_window_tables_match_oracle(kind, param, n):
    w = {"rectangular": sg.WindowType.rectangular, "hanning": sg.WindowType.hanning, "hamming": sg.WindowType.hamming,
         "blackman": sg.WindowType.blackman, "kaiser": sg.WindowType.kaiser(param),
         "gaussian": sg.WindowType.gaussian(param)}[kind]
    got = host_plan(n, max(1, n // 2), w).window()
    ref = orc.make_window(kind, n, param)
    assert np.max(np.abs(got - ref)) <= 4e-16 or np.allclose(got, ref, rtol=1e-15, atol=0, equal_nan=True)


def test_custom_window_roundtrip_and_normalisation():
    c = np.hamming(64)
    assert np.array_equal(host_plan(64, 32, sg.WindowType.custom(c)).window(), c)
    assert abs(sg.WindowType.custom(c, "sum").coefficients.sum() - 1.0) < 1e-15
    assert abs(sg.WindowType.custom(c, "peak").coefficients.max() - 1.0) < 1e-15
    assert abs((sg.WindowType.custom(c, "energy").coefficients ** 2).sum() - 1.0) < 1e-15
    with pytest.raises(ValueError):
        sg.WindowType.custom(c, "bogus")
    with pytest.raises(ValueError):
        sg.WindowType.custom([1.0, float("nan")])
    with pytest.raises(sg.InvalidInputError):  # src/spectrogram.rs:3490-3498
        sg.StftParams(128, 64, sg.WindowType.custom(c))


def test_window_type_from_str():
    """FromStr for WindowType (src/window.rs:276-338): the names, `kaiser=<x>` / `gaussian=<x>`, case and blanks ignored, the
    reference's three error texts."""
    W = sg.WindowType
    for spec, kind in (("hann", W.hanning), ("Hanning", W.hanning), (" HAMM ", W.hamming), ("hamming", W.hamming), ("rect", W.rectangular),
                       ("Rectangle", W.rectangular), ("blackman", W.blackman)):
        assert W.from_str(spec) is kind
    k = W.from_str("kaiser=5.0")
    assert k.kind == _ffi.WIN_KAISER and k.param == 5.0
    g = W.from_str("  Gaussian=12 ")
    assert g.kind == _ffi.WIN_GAUSSIAN and g.param == 12.0
    with pytest.raises(sg.InvalidInputError, match="Input must not be empty. Must be one of"):
        W.from_str("")
    for bad in ("rectangular", "kaiser", "kaiser=", "kaiser=-1", "kaiser=1e3", "kaiser=.5", "gaussian=1.", "hann=3", "tukey", "   "):
        with pytest.raises(sg.InvalidInputError, match="Invalid window specification '%s'" % bad):
            W.from_str(bad)
    # the parsed window builds the same table as the constructor
    assert np.array_equal(W.make_kaiser(64, 5.0), W._make(W.from_str("kaiser=5.0"), 64, None))


@pytest.mark.parametrize("norm", [None, sg.MelNorm.slaney, sg.MelNorm.l1, sg.MelNorm.l2])
@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(16000, 1024, 80, 0.0, 8000.0), (22050, 400, 64, 20.0, 7600.0),
                                                        (16000, 512, 40, 0.0, 8000.0)])
def test_mel_tables_match_oracle(sr, n_fft, n_mels, fmin, fmax, norm):
    ptr, col, val = host_plan(n_fft, n_fft // 4, sr=float(sr), mel=sg.MelParams(n_mels, fmin, fmax, norm)).mel_weights()
    oname = {None: None, sg.MelNorm.slaney: "slaney", sg.MelNorm.l1: "l1", sg.MelNorm.l2: "l2"}[norm]
    optr, ocol, oval, _ = orc.mel_filterbank(sr, n_fft, n_mels, fmin, fmax, oname)
    assert np.array_equal(ptr.astype(np.int64), optr.astype(np.int64))
    assert np.array_equal(col, ocol)
    assert np.allclose(val, oval, rtol=1e-15, atol=0)


@pytest.mark.parametrize("n,n_fft,hop,centre", [(16000, 512, 256, True), (160000, 1024, 256, True), (5, 512, 256, True),
                                                (5, 512, 256, False), (1000, 400, 160, False), (1023, 1024, 1024, False),
                                                (4096, 1024, 1024, True), (777, 7, 3, True)])
def test_output_shape_and_axes_match_oracle(n, n_fft, hop, centre):
    pl = host_plan(n_fft, hop, centre=centre)
    nb, nf = pl.output_shape(n)
    assert nb == n_fft // 2 + 1
    assert nf == orc.frame_count(n, n_fft, hop, centre)
    f, t = pl.axes(nf)
    of, ot = orc.axes(orc.Params(n_fft=n_fft, hop=hop, centre=centre), nf)
    assert np.array_equal(f, of) and np.array_equal(t, ot)
    plm = host_plan(n_fft if n_fft > 16 else 64, hop if n_fft > 16 else 16, mel=sg.MelParams(12, 50.0, 4000.0))
    fm, _ = plm.axes(3)
    ofm, _ = orc.axes(orc.Params(n_fft=64, hop=16, n_mels=12, f_min=50.0, f_max=4000.0), 3)
    assert np.allclose(fm, ofm, rtol=1e-15)


def test_doc_kat_shape():
    assert host_plan(512, 256).output_shape(16000) == (257, 63)  # src/spectrogram.rs:505-507
    assert host_plan(512, 256, mel=sg.MelParams(40, 0.0, 8000.0)).output_shape(16000) == (40, 63)


def test_constructor_errors_match_reference_points():
    with pytest.raises(sg.InvalidInputError, match="hop_size must be <= n_fft"):
        sg.StftParams(512, 513, sg.WindowType.hanning)
    with pytest.raises(sg.InvalidInputError, match="sample_rate_hz"):
        sg.SpectrogramParams(sg.StftParams(512, 256, sg.WindowType.hanning), 0.0)
    with pytest.raises(sg.InvalidInputError, match="f_min"):
        sg.MelParams(40, -1.0, 100.0)
    with pytest.raises(sg.InvalidInputError, match="f_max must be > f_min"):
        sg.MelParams(40, 100.0, 100.0)
    with pytest.raises(sg.InvalidInputError, match="floor_db"):
        sg.LogParams(float("inf"))
    with pytest.raises(sg.InvalidInputError, match="Nyquist"):  # tests/spectrogram_tests.rs:147-158
        host_plan(512, 256, mel=sg.MelParams(40, 0.0, 9000.0))
    with pytest.raises(sg.InvalidInputError, match="unreasonably large"):
        host_plan(512, 256, mel=sg.MelParams(10001, 0.0, 8000.0))
    with pytest.raises(ValueError):
        sg.Plan(sg.SpectrogramParams(sg.StftParams(512, 256, sg.WindowType.hanning), 16000.0), _ffi.AMP_POWER,
                dtype="float16", device=HOST)


def test_raw_abi_rejects_bad_params_without_aborting():
    L = _ffi.lib()
    p = _ffi.SgxParams()
    p.n_fft, p.hop_size, p.sample_rate_hz, p.device = 512, 600, 16000.0, HOST
    h = C.c_void_p()
    assert L.sgx_plan_create(C.byref(p), C.byref(h)) == _ffi.SGX_INVALID_INPUT and not h.value
    assert b"hop_size" in L.sgx_last_create_error()
    p.hop_size, p.window_kind = 256, _ffi.WIN_CUSTOM
    assert L.sgx_plan_create(C.byref(p), C.byref(h)) == _ffi.SGX_INVALID_INPUT
    assert b"Custom window size (0) must match n_fft (512)" in L.sgx_last_create_error()
    assert L.sgx_plan_create(None, C.byref(h)) == _ffi.SGX_INVALID_INPUT


@pytest.mark.parametrize("n_fft", [2 ** 30 + 2, 2 ** 31 + 6, 2 ** 32 - 1, 2 ** 20 + 1, 2 ** 21 + 2, 2 ** 22])
def test_huge_frame_lengths_fail_fast(n_fft):
    """Lengths above the global-memory transforms' range (2^20; powers of two 2^21 — round 5; before: above every kernel's frame
    tile): SGX_BACKEND at once — (2^30, 2^31] used to spin in the chirp-z length loop (a 32-bit
    M shifted to 0), above 2^31 `2 n` wrapped and a tiny convolution length could be selected."""
    import time
    L = _ffi.lib()
    p = _ffi.SgxParams()
    p.n_fft, p.hop_size, p.sample_rate_hz, p.device, p.centre = n_fft, n_fft // 4, 16000.0, HOST, 1
    h = C.c_void_p()
    t0 = time.perf_counter()
    assert L.sgx_plan_create(C.byref(p), C.byref(h)) == _ffi.SGX_BACKEND and not h.value
    assert b"n_fft too large" in L.sgx_last_create_error() and time.perf_counter() - t0 < 1.0


def test_host_only_plan_refuses_compute_loudly():
    pl = host_plan(256, 64, dtype="float32")
    with pytest.raises(sg.FFTBackendError, match="no HIP device"):
        pl.compute_batch(np.zeros((2, 1000), np.float32))
    with pytest.raises(sg.FFTBackendError):
        pl.r2c(np.zeros(256, np.float32))
    with pytest.raises(sg.DimensionMismatchError, match="expected 256, got 100"):  # validate_fft_io
        pl.r2c(np.zeros(100, np.float32))
    bad = np.empty((2, 129, 3), np.float32)
    with pytest.raises(sg.DimensionMismatchError):  # compute_into shape check comes before the device check
        pl.compute_batch(np.zeros((2, 1000), np.float32), out=bad)
    with pytest.raises(ValueError):
        pl.compute_batch(np.zeros((0, 10), np.float32))


def test_kernel_selection():
    assert host_plan(1024, 256, dtype="float32").kernel_name in ("r32x16_f32", "reg_radix")
    assert host_plan(1024, 256, dtype="float64").kernel_name in ("d32x16_f64", "reg_radix")  # (as for f32: a host-only plan reports the kind it would ask for)
    assert host_plan(1024, 255, dtype="float64").kernel_name in ("d32x16_f64", "reg_radix")  # (odd hops on the tuned f64 kernel since round 5)
    assert host_plan(1024, 256, mel=sg.MelParams(40, 0.0, 8000.0), dtype="float64").kernel_name in ("d32x16_f64", "reg_radix")
    assert host_plan(1024, 256, mel=sg.MelParams(400, 0.0, 8000.0), dtype="float64").kernel_name == "reg_radix"  # schedule too long for the LDS left
    assert host_plan(4096, 1024, dtype="float64").kernel_name == "reg_radix"
    assert host_plan(16, 4, dtype="float32").kernel_name == "lds_radix2"
    assert host_plan(400, 160).kernel_name == "reg_radix"       # m = 200 = 25 x 8
    # lengths outside the register-tiled lists: chirp-z (Bluestein), one kernel with the sequence resident in LDS up to
    # n_fft 2048 (M <= 4096) — faster than the two-factor / direct sums from n_fft ~17 on
    assert host_plan(100, 40).kernel_name == "bluestein"        # 10 x 10
    assert host_plan(441, 160).kernel_name == "bluestein"       # 21 x 21
    assert host_plan(97, 40).kernel_name == "bluestein"         # small prime
    assert host_plan(1023, 256).kernel_name == "bluestein"      # 31 x 33
    assert host_plan(401, 160).kernel_name == "bluestein"       # prime
    assert host_plan(1006, 500).kernel_name == "bluestein"      # 2 x 503
    assert host_plan(2, 1).kernel_name == "direct_dft"
    assert host_plan(11, 4).kernel_name == "direct_dft"         # below 16 points the direct sum stays
    assert host_plan(34, 8, mel=sg.MelParams(8, 0.0, 8000.0), dtype="float32").kernel_name == "two_factor_dft"  # short frames into a filterbank (f32): the two-factor kernel's fused bank wins
    assert host_plan(34, 8).kernel_name == "bluestein"
    # n_fft 2049 ... 8192: one 8192- / 16384-point sequence per workgroup (f32), 8192 only in f64
    assert host_plan(5003, 2000, dtype="float32").kernel_name == "bluestein"
    assert host_plan(3000, 700).kernel_name == "bluestein"       # 50 x 60: M = 8192
    assert host_plan(6000, 1500, dtype="float32").kernel_name == "bluestein"
    assert host_plan(6000, 1500, dtype="float64").kernel_name == "bluestein"  # f64: M = 16384 does not fit LDS; even: half-length complex form (M = 8192)
    assert host_plan(12000, 3000, dtype="float32").kernel_name == "bluestein"  # likewise in f32 above 8192
    # round 5: what no on-chip kernel transforms in O(n log n) goes through global memory (bigfft.hip) — nothing above 2048 points is a sum
    assert host_plan(6001, 1500, dtype="float64").kernel_name == "big_chirpz"  # odd: no half-length form
    assert host_plan(4099, 1000, dtype="float64").kernel_name == "big_chirpz"
    assert host_plan(9001, 1000, dtype="float32").kernel_name == "big_chirpz"       # M = 32768
    assert host_plan(65536, 16384, dtype="float32").kernel_name == "big_four_step"
    assert host_plan(32768, 8192, dtype="float64").kernel_name == "big_four_step"  # (f64 frames above 16384 had no tile)
    assert host_plan(2039, 500, dtype="float64").kernel_name == "bluestein"         # (the LDS chirp-z where it fits)


def test_shard_range_partitions_batch():
    L = _ffi.lib()
    for batch in (1, 7, 8, 256, 8192, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = C.c_size_t(), C.c_size_t()
                assert L.sgx_shard_range(batch, world, r, C.byref(s), C.byref(c)) == 0
                seen.append((s.value, c.value))
            assert seen[0][0] == 0 and sum(c for _, c in seen) == batch
            for (s0, c0), (s1, _) in zip(seen, seen[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in seen) - min(c for _, c in seen) <= 1
    s, c = C.c_size_t(), C.c_size_t()
    assert L.sgx_shard_range(8, 0, 0, C.byref(s), C.byref(c)) == _ffi.SGX_INVALID_INPUT


def test_plan_cache_key_sees_custom_window_coefficients():
    """two custom windows of equal length must not share a cached plan (the key is built from every field by value)."""
    from spectrograms_amd.functions import _key
    a = sg.WindowType.custom(np.hamming(64))
    b = sg.WindowType.custom(np.hanning(64) + 0.1)
    pa = sg.SpectrogramParams(sg.StftParams(64, 16, a, True), 16000.0)
    pb = sg.SpectrogramParams(sg.StftParams(64, 16, b, True), 16000.0)
    assert _key(a) != _key(b) and _key(pa) != _key(pb)
    assert _key(pa) == _key(sg.SpectrogramParams(sg.StftParams(64, 16, sg.WindowType.custom(np.hamming(64)), True), 16000.0))
    assert _key(sg.WindowType.kaiser(5.0)) != _key(sg.WindowType.kaiser(6.0))
    assert _key(sg.MelParams(80, 0.0, 8000.0)) != _key(sg.MelParams(80, 0.0, 7999.0))
    hash(_key(pa))  # usable as a dict key
    with pytest.raises(TypeError):
        _key(object())


def test_dimension_mismatch_carries_expected_and_got():
    """DimensionMismatch { expected, got } (src/error.rs:19-21) reaches Python as numbers, not only as message text."""
    plan = host_plan(512, 256)
    x = np.zeros((1, 4000))
    with pytest.raises(sg.DimensionMismatchError) as ei:
        plan.compute_batch(x, out=np.empty((1, 257, 3)))
    nb, nf = plan.output_shape(4000)
    assert ei.value.expected == nb * nf and ei.value.got == 257 * 3 and "Dimension mismatch" in str(ei.value)
    assert plan.device == -2


def test_rccl_not_found_is_a_soft_failure():
    """A host without any RCCL: sgx_comm_unique_id / create / adopt report SGX_BACKEND with a message (round 2's code called
    dlerror() twice and built a std::string from NULL).  shard.hip is rebuilt with -DSGX_NO_RCCL (every library name fails to
    load) into build/libsgx_norccl.so and driven through ctypes in a child process, so a crash would only kill the child."""
    import subprocess
    import sys

    from spectrograms_amd import build as b

    so = b.variant("norccl", ["-DSGX_NO_RCCL"], ("shard.hip",))
    code = (
        "import ctypes as C, sys\n"
        f"L = C.CDLL({so!r})\n"
        "L.sgx_comm_last_error.restype = C.c_char_p\n"
        "L.sgx_comm_last_error.argtypes = [C.c_void_p]\n"
        "buf = (C.c_char * 128)()\n"
        "st = L.sgx_comm_unique_id(buf)\n"
        "msg = L.sgx_comm_last_error(None).decode()\n"
        "h = C.c_void_p()\n"
        "st2 = L.sgx_comm_create(buf, 2, 0, -1, C.byref(h))\n"
        "st3 = L.sgx_comm_adopt(C.c_void_p(0x1000), 2, 0, -1, C.byref(h))\n"
        "print(st, st2, st3, msg)\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    st, st2, st3, msg = r.stdout.strip().split(" ", 3)
    assert (int(st), int(st2), int(st3)) == (_ffi.SGX_BACKEND,) * 3
    assert "RCCL not found" in msg


def test_kernel_kind_is_resolved_with_the_band_schedule_known():
    """n_fft 512 filterbank plans take the tuned kernel's two-frames-per-transform mode only when the bank has a band schedule
    (built on the host before the kind is resolved); other banks must report the kernel that will really run them."""
    mel = host_plan(512, 128, mel=sg.MelParams(40, 0.0, 8000.0), dtype="float32")
    assert mel.kernel_name == "r32x16_f32"
    erb = sg.Plan(sg.SpectrogramParams(sg.StftParams(512, 128, sg.WindowType.hanning, True), 16000.0), _ffi.AMP_POWER,
                  sg.ErbParams(64, 50.0, 8000.0), None, "float32", device=HOST)
    assert erb.kernel_name == "reg_radix"  # 64 dense rows of 257 bins: too many words for the LDS schedule
    lin = host_plan(512, 128, dtype="float32")
    assert lin.kernel_name == "r32x16_f32"
    assert host_plan(512, 256, mel=sg.MelParams(40, 0.0, 8000.0), dtype="float32").kernel_name == "r32x16_f32"  # (round 4: larger LDS halves at hop 256)
    assert host_plan(512, 200, mel=sg.MelParams(40, 0.0, 8000.0), dtype="float32").kernel_name == "reg_radix"


def test_host_plans_over_the_tuned_shapes_and_banks():
    """Plan creation (host tables, band schedules with 4- and 8-byte weights, split-path decisions) for every shape-specific kernel with every kind
    of bank, without a device: no crash, and the kinds a plan asks for."""
    want = {("float32", 512, 160): "r32x16_f32", ("float32", 1024, 256): "r32x16_f32", ("float32", 2048, 512): "r32x32_f32", ("float32", 4096, 1024): "r64x32_f32",
            ("float64", 512, 160): "d512_f64", ("float64", 1024, 256): "d32x16_f64", ("float64", 2048, 512): "d32x32_f64", ("float64", 4096, 1024): "reg_radix",
            ("float64", 512, 262): "reg_radix", ("float32", 8192, 2048): "reg_radix"}
    for (dtype, n_fft, hop), kernel in want.items():
        p = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
        assert sg.Plan(p, _ffi.AMP_POWER, None, None, dtype, device=HOST).kernel_name == kernel
        assert sg.Plan(p, _ffi.AMP_COMPLEX, None, None, dtype, device=HOST).kernel_name == kernel
        for mel in (sg.MelParams(80, 0.0, 8000.0), sg.MelParams(128, 0.0, 8000.0), sg.MelParams(400, 0.0, 8000.0), sg.MelParams(3, 0.0, 100.0)):
            for amp, db in ((_ffi.AMP_POWER, None), (_ffi.AMP_MAGNITUDE, None), (_ffi.AMP_DECIBELS, sg.LogParams(-80.0))):
                assert sg.Plan(p, amp, mel, db, dtype, device=HOST).output_shape(16000)[0] == mel.n_mels
        for bank in (sg.ErbParams(64, 50.0, 8000.0), sg.LogHzParams(64, 50.0, 8000.0)):
            assert sg.Plan(p, _ffi.AMP_POWER, bank, None, dtype, device=HOST).output_shape(16000)[0] == 64
