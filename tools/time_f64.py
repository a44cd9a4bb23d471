import os, sys; sys.path.insert(0, os.getcwd())
import numpy as np, torch
import spectrograms_amd as sg
from oracle import oracle as orc
from tests import helpers as H
B = 256
x32 = H.cfg2_batch(B)
for dtype, n_fft, hop in (("float64", 1024, 256), ("float64", 512, 128), ("float64", 2048, 512), ("float64", 256, 64)):
    x = torch.from_numpy(x32).to(torch.float64).cuda()
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    for name, plan in (("linear", sg.SpectrogramPlanner().linear_power_plan(params, dtype=dtype)),
                       ("mel80db", sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype=dtype))):
        nb, nf = plan.output_shape(x.shape[1])
        out = torch.empty((B, nb, nf), dtype=torch.float64, device="cuda")
        plan.time_batch_torch(x, out, 2)
        ms = plan.time_batch_torch(x, out, 5)
        err = ""
        if name == "linear":
            ref = orc.spectrogram_batch(orc.Params(n_fft=n_fft, hop=hop), x32[:2].astype(np.float64))
            got = plan.compute_batch(x[:2]).cpu().numpy()
            err = "relerr=%.2e" % (np.abs(got - ref).max() / ref.max())
        print(f"{dtype} n_fft={n_fft:5d} {name:8s} {ms * 1e3:9.1f} us {err}", flush=True)
