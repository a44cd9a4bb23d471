#!/usr/bin/env python3
"""Batches of short signals (tiles that continue into the next signal): frames/s of the tuned kernel for 1 ... 40 frames
per signal at a fixed total of ~262 k frames, linear power and Mel-80 dB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spectrograms_amd as sg


def run(n_fft, hop, frames, n_mels=None):
    n = (frames - 1) * hop + 8  # centred: (n + n_fft - n_fft) / hop + 1 frames
    batch = max(1, 262144 // frames)
    x = torch.randn((batch, n), dtype=torch.float32, device="cuda")
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    pl = sg.SpectrogramPlanner()
    plan = pl.linear_power_plan(params, dtype="float32") if n_mels is None else pl.mel_db_plan(params, sg.MelParams(n_mels, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32")
    nb, nf = plan.output_shape(n)
    assert nf == frames, (nf, frames)
    out = torch.empty((batch, nb, nf), dtype=torch.float32, device="cuda")
    plan.time_batch_torch(x, out, 3)
    ms = plan.time_batch_torch(x, out, 20)
    print(f"n_fft={n_fft:5d} hop={hop:4d} {'linear' if n_mels is None else 'Mel-80 dB':9s} frames/signal={frames:3d} B={batch:6d} {plan.kernel_name:12s} {ms * 1e3:8.1f} us {batch * nf / ms / 1e3:8.1f} M frames/s", flush=True)


for mel in (None, 80):
    for fr in (1, 2, 4, 5, 8, 12, 15, 16, 17, 20, 31, 40, 63):
        run(1024, 256, fr, mel)
for fr in (1, 2, 4, 5, 8, 16, 24, 31, 32, 33, 40):
    run(512, 128, fr)
