#!/usr/bin/env python3
"""Diagnostic: build the library with -DSGX_STAMPS (s_memtime stamps around the phases of the tuned kernel), run the
BASELINE workload and print each phase's share of a wave's cycles.  Shares only — a stamped build forbids overlaps."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.environ.get("SGX_STAMPS_LIB", "/tmp/libsgx_stamps.so")  # prebuilt (tools/mkvariant.sh stamps -DSGX_STAMPS) or built here
flags = sys.argv[2:] if len(sys.argv) > 2 else []
if "SGX_STAMPS_LIB" not in os.environ:
  subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DSGX_STAMPS", *flags,
                "-I" + ROOT + "/include", "-I" + ROOT + "/spectrograms_amd/csrc", "-o", lib] +
               [ROOT + "/spectrograms_amd/csrc/" + f for f in ("plan.hip", "fft2d.hip", "kernels_generic.hip", "kernels_r32x16.hip", "kernels_fft2d.hip", "kernels_c2c1024.hip", "kernels_reg2d.hip", "kernels_q16x32.hip")], check=True)
os.environ["SGX_LIB_PATH"] = lib
import numpy as np
import torch
import bench
import spectrograms_amd as sg
from spectrograms_amd import _ffi

wl = sys.argv[1] if len(sys.argv) > 1 else "linear_power"
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
pl = sg.SpectrogramPlanner()
plan = {"linear_power": lambda: pl.linear_power_plan(params, dtype="float32"),
        "mel_db": lambda: pl.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32"),
        "stft": lambda: pl.stft_plan(params, dtype="float32")}[wl]()
x = torch.from_numpy(np.stack([bench.cfg_signal(b) for b in range(256)])).cuda()
out = plan.compute_batch(x)
torch.cuda.synchronize()
L = _ffi.lib()
buf = (C.c_ulonglong * 32)()
L.sgx_debug_read_stamps(buf, 1)
iters = 5
o = torch.view_as_real(out) if out.is_complex() else out
ms = plan.time_batch_torch(x, o, iters)
L.sgx_debug_read_stamps(buf, 1)
ws = os.environ.get("SGX_KERNEL", "") == "ws"
if ws:
    names = ["phase R (LDS reads)", "barrier M wait", "phase C (compute/store)", "barrier E wait", "  C: stage xs (+vmcnt wait)", "  C: fetch issue", "-"]
    roles = (("producer", 0), ("consumer", 8))
else:
    names = ["window reads + x wait", "pass 1 (FFT32, tw, ex writes)", "prefetch issue", "barrier 1", "ex reads", "barrier 2", "pass 2 (+stores, Mel)"]
    roles = (("all waves", 0),)
print(f"workload={wl} flags={flags} kernel_ms(stamped)={ms:.4f}")
for role, base in roles:
    waves = buf[base + 7]
    tot = sum(buf[base + i] for i in range(7))
    ticks = 41.0 if ws else 20.0
    print(f" {role}: waves={waves}")
    for i, n in enumerate(names):
        print(f"  {n:30s} {buf[base + i] / max(waves, 1) / ticks:10.0f} cyc/wave/tile  {100.0 * buf[base + i] / max(tot, 1):5.1f} %")
    print(f"  total {tot / max(waves, 1) / ticks:.0f} cycles per wave per tile")
    if not ws:
        print(f"  (separately) sample wait + stage writes {buf[16] / max(waves, 1) / ticks:10.0f} cyc/wave/tile — the first row then holds barrier + column / window reads + barrier only")
