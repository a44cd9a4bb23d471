import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectrograms_amd as sg
for B in (64, 128, 192, 512):
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((B, 1024, 1024), generator=g, device="cuda", dtype=torch.float32)
    plan = sg.Fft2dPlan(1024, 1024, "float32")
    for rep in range(2):
        z = plan.convolve_torch(x, np.ones((1, 1), np.float32))
        torch.cuda.synchronize()
        err = (z - x).abs().amax(dim=(1, 2)).cpu().numpy()
        bad = np.nonzero(err > 1e-4)[0]
        print("B", B, "rep", rep, "bad images", bad[:20], len(bad), flush=True)
        if len(bad):
            i = int(bad[0])
            e = (z[i] - x[i]).abs().cpu().numpy()
            rows = np.nonzero(e.max(axis=1) > 1e-4)[0]; cols = np.nonzero(e.max(axis=0) > 1e-4)[0]
            print("   image", i, "bad rows", rows[:10], len(rows), "bad cols", cols[:10], len(cols), flush=True)
