#!/usr/bin/env python3
"""Frames of lengths outside the register-tiled lists (64 x 10 s, hop n/4): primes and 2 x prime on the chirp-z path against their
power-of-two neighbours (no row more than ~4x off its neighbour) and against the two-factor kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spectrograms_amd as sg

B, N = 64, 160000
LENGTHS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 251, 512, 509, 1006, 1009, 1023, 1024, 2003, 2048, 4093, 4096, 5003, 8192]
DTYPES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["float32", "float64"]
for dtype in DTYPES:
    tdt = torch.float32 if dtype == "float32" else torch.float64
    x = torch.randn((B, N), dtype=tdt, device="cuda")
    for n_fft in LENGTHS:
        if dtype == "float64" and n_fft > 4096:
            continue
        hop = max(1, n_fft // 4)
        params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
        for name, plan in (("linear", sg.SpectrogramPlanner().linear_power_plan(params, dtype=dtype)),
                           ("mel80db", sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype=dtype))):
            nb, nf = plan.output_shape(N)
            out = torch.empty((B, nb, nf), dtype=tdt, device="cuda")
            plan.time_batch_torch(x, out, 1)
            ms = plan.time_batch_torch(x, out, 3)
            print(f"{dtype} n_fft={n_fft:5d} hop={hop:5d} {name:8s} {plan.kernel_name:14s} {ms * 1e3:10.1f} us {B * nf / ms / 1e3:9.1f} M frames/s", flush=True)
