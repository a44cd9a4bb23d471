"""is the bin rows' cache-line straddling what limits the headline?  The unchanged tuned kernel on 256 signals whose
frame count makes the output row pitch a whole number of 128-byte lines (640 frames: 2560 B = 20 lines; 608: 2432 B = 19 lines) against
BASELINE's 626 frames (2504 B) and 624 (2496 B, 64-byte aligned only).  Interleaved, REPS rounds; reports ns per frame."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spectrograms_amd as sg

B = int(os.environ.get("B", 256)); REPS = int(os.environ.get("REPS", 5)); ITERS = int(os.environ.get("ITERS", 200))
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
plans = {"linear": sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32"),
         "mel80": sg.SpectrogramPlanner().mel_power_plan(params, sg.MelParams(80, 0.0, 8000.0), dtype="float32"),
         "stft": sg.SpectrogramPlanner().stft_plan(params, dtype="float32")}
rng = np.random.default_rng(5)
cases = {}
for nf in [int(v) for v in os.environ.get("FRAMES", "626,640,624,608,632").split(",")]:
    ns = (nf - 1) * 256
    x = torch.from_numpy((0.1 * rng.standard_normal((B, ns))).astype(np.float32)).cuda()
    for name, plan in plans.items():
        nb, f = plan.output_shape(ns)
        assert f == nf
        out = torch.empty((B, nb, nf, 2) if name == "stft" else (B, nb, nf), dtype=torch.float32, device="cuda")
        cases[(name, nf)] = (plan, x, out)
res = {k: [] for k in cases}
for k, (plan, x, out) in cases.items():
    plan.time_batch_torch(x, out, 50)
for r in range(REPS):
    for k, (plan, x, out) in cases.items():
        plan.time_batch_torch(x, out, 20)
        res[k].append(plan.time_batch_torch(x, out, ITERS))
for (name, nf), v in res.items():
    ms = float(np.median(v))
    base = float(np.median(res[(name, 626)])) / (B * 626)
    print(f"{name:7s} frames={nf:4d} pitch={nf * 4:5d}B  {ms * 1e3:8.1f} us  {ms * 1e6 / (B * nf):7.4f} ns/frame  vs626 {ms / (B * nf) / base:6.3f}  min {min(v) * 1e3:8.1f} max {max(v) * 1e3:8.1f}", flush=True)
