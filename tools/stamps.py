#!/usr/bin/env python3
"""Diagnostic: run the BASELINE workload on a library built with -DSGX_STAMPS (s_memtime stamps around the phases of the tuned
kernel; python -m spectrograms_amd.build --variant stamps -DSGX_STAMPS) and print each phase's cycles per wave per round.
A stamped build forbids overlaps across the stamps, so read the numbers as shares."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SGX_LIB_PATH"] = os.environ.get("SGX_STAMPS_LIB", os.path.join(ROOT, "build", "libsgx_stamps.so"))
import numpy as np
import torch
import bench
import spectrograms_amd as sg
from spectrograms_amd import _ffi

wl = sys.argv[1] if len(sys.argv) > 1 else "linear_power"
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
pl = sg.SpectrogramPlanner()
plan = {"linear_power": lambda: pl.linear_power_plan(params, dtype="float32"),
        "mel_power": lambda: pl.mel_power_plan(params, sg.MelParams(80, 0.0, 8000.0), dtype="float32"),
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
names = ["wait samples + stage writes", "barrier 1", "col/window reads + FFT32", "barrier 2", "twiddle + ex writes", "load issue",
         "barrier 3", "row reads", "barrier 4", "FFT16 x2 (+ job-0 fixup)", "real split + stores / LDS writes", "barrier (|X|^2 tile complete)", "band stage: prologue", "band stage: loops",
         "band stage: epilogues + stores"]
waves, rounds = buf[16], buf[15]
tot = sum(buf[i] for i in range(15))
print(f"workload={wl} kernel_ms(stamped)={ms:.4f} waves={waves} wave-rounds={rounds}")
for i, n in enumerate(names):
    if n == '-':
        continue
    print(f"  {n:30s} {buf[i] / max(rounds, 1):9.0f} cyc/wave/round  {100.0 * buf[i] / max(tot, 1):5.1f} %")
print(f"  total {tot / max(rounds, 1):.0f} cycles per wave per round")
