#!/usr/bin/env python3
"""Diagnostic: cycles per phase of the fused inverse STFT kernel (k_istft1024b) on a library built with -DSGX_IS_STAMPS
(python -m spectrograms_amd.build --variant isstamps --src kernels_c2c1024.hip -DSGX_IS_STAMPS); 256 x [513, 626] f32."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SGX_LIB_PATH"] = os.environ.get("SGX_STAMPS_LIB", os.path.join(ROOT, "build", "libsgx_isstamps.so"))
import torch
import spectrograms_amd as sg
from spectrograms_amd import _ffi
from tests import helpers as H

NAMES = ["wait for the pairs + fold", "next tile's requests (issue)", "16-point transforms + exchange writes", "barrier", "column reads, twiddles, 32-point transform",
         "barrier", "scale, window, frame writes", "barrier", "overlap-add, normalise, stores", "barrier"]
x = torch.from_numpy(H.cfg2_batch(256)).cuda()
params = sg.SpectrogramParams(sg.StftParams(1024, int(os.environ.get("HOP", 256)), sg.WindowType.hanning, True), 16000.0)
plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
S = plan.compute_batch(x).contiguous()
y = plan.istft_batch(S)
torch.cuda.synchronize()
L = _ffi.lib()
buf = (C.c_ulonglong * 16)()
L.sgx_debug_read_is_stamps(buf, 1)
for _ in range(3):
    plan.istft_batch(S, out=y)
torch.cuda.synchronize()
L.sgx_debug_read_is_stamps(buf, 1)
waves, rounds = max(buf[11], 1), max(buf[10], 1)
tot = sum(buf[i] for i in range(10))
print(f"waves={waves} wave-tiles={rounds}")
for i, nm in enumerate(NAMES):
    print(f"  {nm:44s} {buf[i] / rounds:9.0f} cyc/wave/tile  {100.0 * buf[i] / max(tot, 1):5.1f} %")
print(f"  total {tot / rounds:.0f} cycles per wave per tile")
