#!/usr/bin/env python3
"""Diagnostic: phase shares of the register-tiled generic kernel (k_reg_radix) on a library built with -DSGX_RR_STAMPS
    python -m spectrograms_amd.build --variant rrstamps --src kernels_generic.hip -DSGX_RR_STAMPS
    N_FFT=512 HOP=128 DTYPE=float32 WL=linear python tools/stamps_generic.py
A stamped build forbids overlaps across the stamps, so read the numbers as shares."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SGX_LIB_PATH"] = os.environ.get("SGX_STAMPS_LIB", os.path.join(ROOT, "build", "libsgx_rrstamps.so"))
import numpy as np
import torch
import spectrograms_amd as sg
from spectrograms_amd import _ffi
from tests import helpers as H

n_fft, hop, dt, wl = int(os.environ.get("N_FFT", 512)), int(os.environ.get("HOP", 128)), os.environ.get("DTYPE", "float32"), os.environ.get("WL", "linear")
params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
pl = sg.SpectrogramPlanner()
plan = pl.linear_power_plan(params, dtype=dt) if wl == "linear" else pl.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype=dt)
x = torch.from_numpy(H.cfg2_batch(256)).to(torch.float32 if dt == "float32" else torch.float64).cuda()
out = plan.compute_batch(x)
torch.cuda.synchronize()
L = C.CDLL(os.environ["SGX_LIB_PATH"])
buf = (C.c_ulonglong * 32)()
L.sgx_debug_read_rr_stamps(buf, 1)
ms = plan.time_batch_torch(x, out, 5)
L.sgx_debug_read_rr_stamps(buf, 1)
names = ["staging writes", "barrier", "pass 1: reads + transform", "barrier", "pass 1: twiddles + tile writes", "load issue", "barrier", "pass 2", "barrier",
         "pass 3", "barrier", "wait for next samples", "split + stores (filterbank outputs: bank stage)", "barrier",
         "filterbank outputs: split + |X|^2 rows + barrier"]
waves, rounds = buf[16], buf[15]
tot = sum(buf[i] for i in range(15))
print(f"n_fft={n_fft} hop={hop} {dt} {wl} kernel={plan.kernel_name} kernel_ms(stamped)={ms:.4f} waves={waves} wave-tiles={rounds}")
for i, n in enumerate(names):
    print(f"  {n:34s} {buf[i] / max(rounds, 1):9.0f} cyc/wave/tile  {100.0 * buf[i] / max(tot, 1):5.1f} %")
print(f"  total {tot / max(rounds, 1):.0f} cycles per wave per tile")
