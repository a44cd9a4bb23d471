#!/usr/bin/env python3
"""Shape sweep (looking for slow corners, not for headline numbers): forward linear power / Mel-80 dB and the inverse STFT over
n_fft x hop x dtype, B x 10 s of 16 kHz audio, device-resident, HIP events.  Prints time per batch, frames/s and the algorithmic
GB/s (samples in + outputs out; the inverse: spectrum in + samples out)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import spectrograms_amd as sg
from spectrograms_amd import _ffi
from tests import helpers as H

B = int(os.environ.get("B", 64))
x32 = H.cfg2_batch(B)
which = os.environ.get("WHICH", "forward,inverse").split(",")
for dtype in os.environ.get("DTYPES", "float32,float64").split(","):
    tdt = torch.float32 if dtype == "float32" else torch.float64
    es = 4 if dtype == "float32" else 8
    x = torch.from_numpy(x32).to(tdt).cuda()
    for n_fft in [int(v) for v in os.environ.get("NFFTS", "64,128,256,400,512,1024,2048,4096").split(",")]:
        for div in (8, 4, 2, 1):
            hop = n_fft // div
            params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
            if "forward" in which:
                for name, plan in (("linear", sg.SpectrogramPlanner().linear_power_plan(params, dtype=dtype)),
                                   ("mel80db", sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(min(80, n_fft // 4), 0.0, 8000.0), sg.LogParams(-80.0), dtype=dtype))):
                    nb, nf = plan.output_shape(x.shape[1])
                    out = torch.empty((B, nb, nf), dtype=tdt, device="cuda")
                    plan.time_batch_torch(x, out, 2)
                    ms = plan.time_batch_torch(x, out, 5)
                    gb = (x.numel() + out.numel()) * es / ms / 1e6
                    print(f"{dtype} n_fft={n_fft:5d} hop={hop:5d} {name:8s} {plan.kernel_name:12s} {ms * 1e3:9.1f} us {B * nf / ms / 1e3:8.1f} M frames/s {gb:7.0f} GB/s", flush=True)
            if "inverse" in which:
                plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, dtype)
                S = plan.compute_batch(x).contiguous()
                y = plan.istft_batch(S)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    plan.istft_batch(S, out=y)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                gb = (S.numel() * 2 + y.numel()) * es / ms / 1e6
                print(f"{dtype} n_fft={n_fft:5d} hop={hop:5d} inverse  {'':12s} {ms * 1e3:9.1f} us {B * S.shape[2] / ms / 1e3:8.1f} M frames/s {gb:7.0f} GB/s", flush=True)
