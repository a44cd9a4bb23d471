#!/usr/bin/env python3
"""Repeat-run parity at full size: BATCH x 10 s through the tuned kernels, every element against the oracle, REPS launches each (an intermittent
fault — e.g. the 16-byte store hazard of DESIGN.md §3.5 — shows as a non-zero count of bad elements in some launch)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spectrograms_amd as sg
from spectrograms_amd import _ffi
from oracle import oracle as orc
from tests import helpers as H

B = int(os.environ.get("BATCH", 256)); REPS = int(os.environ.get("REPS", 6)); FORWARD_ONLY = os.environ.get("FORWARD_ONLY", "0") == "1"
base = H.cfg2_batch(B)
CASES = [("float32", 1024, 256, "complex", 0), ("float32", 1024, 256, "power", 0), ("float32", 1024, 256, "power", 80), ("float32", 1024, 512, "complex", 0),
         ("float32", 2048, 512, "complex", 0), ("float32", 2048, 512, "power", 80), ("float32", 512, 128, "complex", 0), ("float32", 512, 160, "power", 80),
         ("float64", 1024, 256, "complex", 0), ("float64", 1024, 256, "power", 0), ("float64", 1024, 256, "power", 80), ("float64", 1024, 512, "complex", 0),
         ("float64", 512, 128, "complex", 0), ("float64", 512, 256, "power", 80), ("float64", 512, 256, "power", 0), ("float32", 512, 256, "power", 80),
         ("float32", 4096, 1024, "complex", 0), ("float64", 2048, 512, "complex", 0), ("float64", 2048, 1024, "power", 0), ("float32", 4096, 1024, "power", 0), ("float32", 400, 160, "power", 80)]
if os.environ.get("ROUND5", "1") == "1":  # round 5: the fused MFCC epilogue, odd hops on the lane-pair kernels, a frame length on the global-memory transforms
    CASES += [("float32", 1024, 256, "mfcc", 80), ("float32", 1024, 441, "mfcc", 80), ("float64", 1024, 255, "complex", 0), ("float64", 1024, 257, "power", 80),
              ("float32", 2048, 511, "complex", 0), ("float64", 512, 159, "power", 0), ("float32", 4096, 1023, "power", 0), ("float64", 2048, 513, "complex", 0),
              ("float32", 9001, 2250, "power", 0), ("float32", 512, 100, "power", 0), ("float32", 512, 384, "complex", 0)]  # (the last two: n_fft 512 at hops without a staged variant — the packed form)
bad_total = 0
for dtype, n_fft, hop, amp, nm in CASES:
    x = base.astype(np.float64 if dtype == "float64" else np.float32)
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    mel = sg.MelParams(nm, 0.0, 8000.0) if nm else None
    x64 = base.astype(np.float64)
    if amp == "mfcc":  # Mel-80 dB(-80) -> DCT-II (13) + lifter 22, fused into the launch on the tuned f32 kernel.  Two checks: the Mel-dB launch
        # against the oracle in the power domain (a pure tone leaves most bands at rounding level, where 1 ulp of f32 power is tenths of a dB:
        # the tolerance of the power cases, carried through the log), and the fused output against the reference's f32 fold
        # (src/mfcc.rs:278-292) of that same Mel-dB tensor, within 4 ulp of the largest coefficient (tests/test_mfcc.py)
        assert dtype == "float32"
        plan = sg.SpectrogramPlanner().mfcc_plan(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0, nm, sg.MfccParams(13), dtype=dtype)
        mplan = sg.SpectrogramPlanner().mel_db_plan(params, mel, sg.LogParams(-80.0), dtype=dtype)
        op = orc.Params(n_fft=n_fft, hop=hop, n_mels=nm, f_min=0.0, f_max=8000.0, amp="db", floor_db=-80.0)
        ref_db = orc.spectrogram_batch(op, x64, nthreads=orc.max_threads())
        xd = torch.from_numpy(x).cuda()
        basis = np.cos(np.pi * np.arange(13)[:, None] * (np.arange(nm)[None, :] + 0.5) / nm).astype(np.float32).astype(np.float64)
        w = (1.0 + 11.0 * np.sin(np.pi * np.arange(13) / 22.0)).astype(np.float32)
        tolp = 2e-4 * max(1.0, float((10.0 ** (ref_db / 10.0)).max()))
        counts = []
        for r in range(REPS):
            mdb = mplan.compute_batch(xd).cpu().numpy()
            got = plan.compute_batch(xd).cpu().numpy()
            bad_mel = int(((np.abs(mdb - ref_db) > 1e-3) & (np.abs(10.0 ** (mdb.astype(np.float64) / 10.0) - 10.0 ** (ref_db / 10.0)) > tolp)).sum())
            acc = np.zeros((B, 13, mdb.shape[2]), np.float32)
            for i in range(nm):
                acc = (mdb[:, i, None, :].astype(np.float64) * basis[None, :, i, None] + acc.astype(np.float64)).astype(np.float32)
            acc = acc * w[None, :, None]
            counts.append(bad_mel + int((np.abs(got - acc) > 4 * np.spacing(np.float32(np.abs(acc).max()))).sum()))
        bad_total += sum(counts)
        print(f"{dtype} {n_fft}/{hop} mfcc-mel{nm} {plan.kernel_name}: bad elements per launch {counts} (bit-equal to the f32 fold: {np.mean(got == acc):.5f})", flush=True)
        continue
    else:
        plan = sg.Plan(params, _ffi.AMP_COMPLEX if amp == "complex" else _ffi.AMP_POWER, mel, None, dtype)
        op = orc.Params(n_fft=n_fft, hop=hop, n_mels=nm, f_min=0.0, f_max=8000.0)
        ref = orc.stft_batch(op, x64, nthreads=orc.max_threads()) if amp == "complex" else orc.spectrogram_batch(op, x64, nthreads=orc.max_threads())
        tol = (1e-10 if dtype == "float64" else 2e-4) * max(1.0, float(np.abs(ref).max()))
    xd = torch.from_numpy(x).cuda()
    counts = []
    for r in range(REPS):
        got = plan.compute_batch(xd).cpu().numpy()
        if amp == "complex" and not np.iscomplexobj(got):
            got = got[..., 0] + 1j * got[..., 1]
        counts.append(int((np.abs(got - ref) > tol).sum()))
    bad_total += sum(counts)
    print(f"{dtype} {n_fft}/{hop} {amp}{'-mel%d' % nm if nm else ''} {plan.kernel_name}: bad elements per launch {counts}", flush=True)
# inverse STFT: the fused kernels (f32 1024 / 2048, f64 1024) and the register-tiled path, 64 signals, every sample against the oracle
for dtype, n_fft, hop in (() if FORWARD_ONLY else (("float32", 1024, 256), ("float32", 2048, 512), ("float64", 1024, 256), ("float32", 1024, 100), ("float64", 1024, 512), ("float32", 512, 128), ("float64", 512, 160), ("float64", 512, 128))):
    x = base[:64].astype(np.float64 if dtype == "float64" else np.float32)
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hamming, True), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, dtype)
    S = np.ascontiguousarray(plan.compute_batch(x))
    ref = np.stack([orc.istft(S[b].astype(np.complex128), n_fft, hop, "hamming", True) for b in range(64)])
    tol = (1e-10 if dtype == "float64" else 2e-5) * max(1.0, float(np.abs(ref).max()))
    Sd = torch.from_numpy(S).cuda()
    counts = []
    for r in range(REPS):
        y = plan.istft_batch(Sd).cpu().numpy()
        counts.append(int((np.abs(y - ref) > tol).sum()))
    bad_total += sum(counts)
    print(f"istft {dtype} {n_fft}/{hop}: bad samples per launch {counts}", flush=True)
print("TOTAL BAD", bad_total)
sys.exit(1 if bad_total else 0)
