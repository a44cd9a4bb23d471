#!/usr/bin/env python3
"""The reference's own default shapes (SpectrogramParams::speech_default 512 / 160, ::music_default 2048 / 512, src/spectrogram.rs:4215-4248) in both
Sample types: B x 10 s of 16 kHz audio, device-resident, HIP events."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import spectrograms_amd as sg
from tests import helpers as H
B = int(os.environ.get("B", 64))
x32 = torch.from_numpy(H.cfg2_batch(B)).cuda()
P = sg.SpectrogramPlanner()
mel = sg.MelParams(80, 0.0, 8000.0)
for name, n_fft, hop in (("speech_default", 512, 160), ("music_default", 2048, 512)):
    for dtype in ("float32", "float64"):
        x = x32 if dtype == "float32" else x32.double()
        params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
        for mode, plan in (("linear_power", P.linear_power_plan(params, dtype=dtype)), ("mel80_db", P.mel_db_plan(params, mel, sg.LogParams(-80.0), dtype=dtype)),
                           ("stft", P.stft_plan(params, dtype=dtype))):
            nb, nf = plan.output_shape(x.shape[1])
            out = torch.empty((B, nb, nf) + ((2,) if mode == "stft" else ()), dtype=x.dtype, device="cuda")
            plan.time_batch_torch(x, out, 3)
            ms = plan.time_batch_torch(x, out, 10)
            print(f"{name:14s} {n_fft}/{hop} {dtype} {mode:12s} {plan.kernel_name:11s} {ms*1e3:8.1f} us {B*nf/ms/1e3:8.1f} M frames/s", flush=True)
