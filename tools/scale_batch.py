"""Kernel time of the tuned linear-power launch versus batch size: separates the fixed cost (launch, prologue, tail) from the
per-round cost of the persistent workgroups."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectrograms_amd as sg
from tests import helpers as H
wl = sys.argv[1] if len(sys.argv) > 1 else "linear_power"
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
pl = sg.SpectrogramPlanner()
plan = pl.linear_power_plan(params, dtype="float32") if wl == "linear_power" else pl.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32")
full = torch.from_numpy(H.cfg2_batch(256)).cuda()
for b in (1, 6, 13, 26, 52, 128, 256):
    x = full[:b].contiguous()
    nb, nf = plan.output_shape(x.shape[1])
    out = torch.empty((b, nb, nf), dtype=torch.float32, device="cuda")
    plan.time_batch_torch(x, out, 3)
    ms = plan.time_batch_torch(x, out, 20)
    tiles = b * 40
    print(f"batch {b:4d} tiles {tiles:6d} rounds/CU {tiles / 512:6.2f}  {ms * 1e3:8.1f} us")
