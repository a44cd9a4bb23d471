#!/usr/bin/env python3
"""One pass over the kernels added or reworked late in round 2, for a rocprofv3 --kernel-trace --stats summary
(profiles/r02_round_end_kernel_stats.csv): the n_fft 512 mode of the tuned kernel at hops 64 / 128 / 160, the fused generic
inverse STFT, the split filterbank (k_bank_rows), the register-tiled 1000-point column pass and the two-factor LDS fallback."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import spectrograms_amd as sg
from spectrograms_amd import _ffi
from tests import helpers as H

x32 = torch.from_numpy(H.cfg2_batch(256)).cuda()
REPS = 6
for hop in (64, 128, 160):
    params = sg.SpectrogramParams(sg.StftParams(512, hop, sg.WindowType.hanning, True), 16000.0)
    for plan in (sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32"),
                 sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32")):
        nb, nf = plan.output_shape(x32.shape[1])
        out = torch.empty((256, nb, nf), dtype=torch.float32, device="cuda")
        for _ in range(REPS):
            plan.compute_batch(x32, out=out) if hasattr(plan, "compute_batch_into") else plan.time_batch_torch(x32, out, 1)
for n_fft, hop in ((512, 128), (400, 160), (256, 64)):
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
    S = plan.compute_batch(x32).contiguous()
    y = plan.istft_batch(S)
    for _ in range(REPS):
        plan.istft_batch(S, out=y)
    del S, y
x64 = x32[:64].double()
for n_fft, hop in ((2048, 512), (4096, 1024)):
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    plan = sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float64")
    nb, nf = plan.output_shape(x64.shape[1])
    out = torch.empty((64, nb, nf), dtype=torch.float64, device="cuda")
    for _ in range(REPS):
        plan.time_batch_torch(x64, out, 1)
for R, C, B in ((1000, 1000, 128), (1023, 1023, 32)):
    img = torch.randn((B, R, C), device="cuda")
    plan = sg.Fft2dPlan(R, C, "float32")
    spec = plan.forward_torch(img)
    for _ in range(REPS):
        plan.forward_torch(img, spec)
torch.cuda.synchronize()
print("done")
