#!/usr/bin/env python3
"""Odd corners of the forward path (slow-corner hunt): batch sizes, signal lengths, hops and band counts around the headline shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import spectrograms_amd as sg


def run(tag, n_fft, hop, batch, n, dtype="float32", n_mels=None, centre=True):
    tdt = torch.float32 if dtype == "float32" else torch.float64
    x = torch.randn((batch, n), dtype=tdt, device="cuda")
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, centre), 16000.0)
    pl = sg.SpectrogramPlanner()
    plan = pl.linear_power_plan(params, dtype=dtype) if n_mels is None else pl.mel_db_plan(params, sg.MelParams(n_mels, 0.0, 8000.0), sg.LogParams(-80.0), dtype=dtype)
    nb, nf = plan.output_shape(n)
    out = torch.empty((batch, nb, nf), dtype=tdt, device="cuda")
    plan.time_batch_torch(x, out, 2)
    ms = plan.time_batch_torch(x, out, 10)
    es = 4 if dtype == "float32" else 8
    print(f"{tag:34s} n_fft={n_fft:5d} hop={hop:5d} B={batch:5d} n={n:9d} {plan.kernel_name:14s} {ms * 1e3:10.1f} us {batch * nf / ms / 1e3:9.1f} M frames/s {(x.numel() + out.numel()) * es / ms / 1e6:7.0f} GB/s", flush=True)
    del x, out, plan
    torch.cuda.empty_cache()


for b in (1, 8, 256, 2048):
    run("batch sweep, linear", 1024, 256, b, 160000)
for b in (1, 8, 256):
    run("batch sweep, Mel-80 dB", 1024, 256, b, 160000, n_mels=80)
for n in (1000, 16000, 1600000, 57600000):
    run("one signal, length sweep", 1024, 256, 1 if n > 2000000 else 64, n)
run("one hour f64", 1024, 256, 1, 57600000, dtype="float64")
for hop in (255, 257, 1024):
    run("hop sweep, linear", 1024, hop, 64, 160000)
for hop in (1, 101, 400):
    run("hop sweep, n_fft 400 Mel-80", 400, hop, 64, 160000, n_mels=80)
for nm in (1, 8, 40, 128, 256, 513):
    run("band-count sweep", 1024, 256, 64, 160000, n_mels=nm)
for nm in (1, 201):
    run("band-count sweep, 400", 400, 160, 64, 160000, n_mels=nm)
run("not centred", 1024, 256, 64, 160000, centre=False)
run("short signals, many", 512, 128, 65536, 400)
run("short signals, many (1024)", 1024, 256, 65536, 1000)
