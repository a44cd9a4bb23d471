#!/usr/bin/env python3
"""Odd hops at n_fft 1024 on the tuned kernel (variant build -DSGX_ODDHOP=1): parity against the oracle for linear power, Mel dB and
complex outputs at hops 255, 257, 441 (44.1 kHz / 10 ms), 1 and 1023, centred or not, and the launch time against the even neighbours."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import spectrograms_amd as sg
from oracle import oracle as orc
from tests.test_gpu_parity import run_case

ok = True
for hop in (255, 257, 441, 1, 1023, 101, 271, 273):
    for amp, kw in (("power", {}), ("complex", {}), ("db", {"n_mels": 80, "floor": -80.0})):
        for centre in (True, False):
            try:
                plan, _ = run_case(n=20011 if hop > 1 else 3000, batch=3, n_fft=1024, hop=hop, amp=amp, centre=centre, **kw)
                print(f"hop {hop:4d} {amp:8s} centre={centre}: ok on {plan.kernel_name}", flush=True)
            except AssertionError as e:
                ok = False
                print(f"hop {hop:4d} {amp:8s} centre={centre}: FAIL {str(e)[:120]}", flush=True)
x = torch.randn((64, 160000), dtype=torch.float32, device="cuda")
for hop in (254, 255, 256, 257, 258, 440, 441, 442):
    params = sg.SpectrogramParams(sg.StftParams(1024, hop, sg.WindowType.hanning, True), 16000.0)
    for name, plan in (("linear", sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32")),
                       ("mel80db", sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32"))):
        nb, nf = plan.output_shape(160000)
        out = torch.empty((64, nb, nf), dtype=torch.float32, device="cuda")
        plan.time_batch_torch(x, out, 3)
        ms = plan.time_batch_torch(x, out, 20)
        print(f"hop {hop:4d} {name:8s} {plan.kernel_name:12s} {ms * 1e3:8.1f} us {64 * nf / ms / 1e3:8.1f} M frames/s", flush=True)
sys.exit(0 if ok else 1)
