"""Device time per 256 x 10 s batch of the shape-generic kernels (f64, and f32 at sizes the tuned kernel does not cover)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectrograms_amd as sg
from tests import helpers as H
B = int(os.environ.get("B", 64))
x32 = H.cfg2_batch(B)
ONLY = os.environ.get("ONLY")
for dtype, n_fft, hop in (("float32", 256, 64), ("float32", 512, 128), ("float32", 2048, 512), ("float32", 4096, 1024), ("float32", 400, 160), ("float64", 1024, 256), ("float64", 512, 128), ("float64", 2048, 512), ("float64", 400, 160)):
    if ONLY and ONLY != f"{dtype}:{n_fft}":
        continue
    tdt = torch.float32 if dtype == "float32" else torch.float64
    x = torch.from_numpy(x32).to(tdt).cuda()
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    for name, plan in (("linear", sg.SpectrogramPlanner().linear_power_plan(params, dtype=dtype)),
                       ("mel80db", sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype=dtype))):
        nb, nf = plan.output_shape(x.shape[1])
        out = torch.empty((B, nb, nf), dtype=tdt, device="cuda")
        plan.time_batch_torch(x, out, 2)
        ms = plan.time_batch_torch(x, out, int(os.environ.get("ITERS", 5)))
        print(f"{dtype} n_fft={n_fft:5d} hop={hop:4d} {name:8s} {plan.kernel_name:12s} {ms * 1e3:9.1f} us  {B * nf / ms / 1e3:8.1f} M frames/s", flush=True)
