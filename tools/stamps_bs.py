#!/usr/bin/env python3
"""Diagnostic: cycles per stage of the fused chirp-z kernel (k_bs_fused) on a library built with -DSGX_BS_STAMPS
(python -m spectrograms_amd.build --variant bsstamps --src bluestein.hip -DSGX_BS_STAMPS).  64 x 10 s, hop n/4, linear power.
A stamped build forbids overlap across the stamps, so read the numbers as shares."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SGX_LIB_PATH"] = os.environ.get("SGX_STAMPS_LIB", os.path.join(ROOT, "build", "libsgx_bsstamps.so"))
import torch
import spectrograms_amd as sg
from spectrograms_amd import _ffi

NAMES = ["sequence table + barrier", "load, window x chirp, P1, T1", "barrier", "P2 + T2 (two-pass: P2, product, P2^-1)", "barrier",
         "P3, product, P3^-1", "barrier", "T2, P2 (inverse)", "barrier", "T1, P1 (inverse), chirp", "barrier", "two-frame split + stores"]
B, N = 64, 160000
for arg in sys.argv[1:] or ["1009:float32"]:
    n_fft, dtype = arg.split(":")
    n_fft = int(n_fft)
    tdt = torch.float32 if dtype == "float32" else torch.float64
    x = torch.randn((B, N), dtype=tdt, device="cuda")
    params = sg.SpectrogramParams(sg.StftParams(n_fft, max(1, n_fft // 4), sg.WindowType.hanning, True), 16000.0)
    plan = sg.SpectrogramPlanner().linear_power_plan(params, dtype=dtype)
    nb, nf = plan.output_shape(N)
    out = torch.empty((B, nb, nf), dtype=tdt, device="cuda")
    plan.time_batch_torch(x, out, 1)
    L = _ffi.lib()
    buf = (C.c_ulonglong * 16)()
    L.sgx_debug_read_bs_stamps(buf, 1)
    ms = plan.time_batch_torch(x, out, 3)
    L.sgx_debug_read_bs_stamps(buf, 1)
    waves = max(buf[12], 1)
    tot = sum(buf[i] for i in range(12))
    print(f"n_fft={n_fft} {dtype} kernel={plan.kernel_name} ms(stamped)={ms:.4f} waves={waves}")
    for i, nm in enumerate(NAMES):
        print(f"  {nm:42s} {buf[i] / waves:9.0f} cyc/wave  {100.0 * buf[i] / max(tot, 1):5.1f} %")
    print(f"  total {tot / waves:.0f} cycles per wave (one tile)")
