#!/usr/bin/env python3
"""Debug aid: n_fft 512 packed tiles vs the oracle — which signals / frames / bins differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spectrograms_amd as sg
from oracle import oracle as orc

rng = np.random.default_rng(1)
for (B, N, hop) in ((3, 6000, 256), (3, 6000, 128), (5, 400, 128), (2, 1000, 128)):
    x = (0.2 * rng.standard_normal((B, N))).astype(np.float32)
    params = sg.SpectrogramParams(sg.StftParams(512, hop, sg.WindowType.hanning, True), 16000.0)
    plan = sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32")
    P = plan.compute_batch(x)
    ref = orc.spectrogram_batch(orc.Params(n_fft=512, hop=hop), x.astype(np.float64))
    err = np.abs(P - ref) / ref.max()
    print(f"B={B} N={N} hop={hop} shape={P.shape} kernel={plan.kernel_name} max rel err {err.max():.3e}")
    if err.max() > 1e-4:
        bad = err > 1e-4
        for b in range(B):
            fr = np.nonzero(bad[b].any(axis=0))[0]
            print(f"  signal {b}: bad frames {fr[:40].tolist()} ({len(fr)} of {P.shape[2]}); bad bins in first bad frame: {np.nonzero(bad[b][:, fr[0]])[0][:20].tolist() if len(fr) else []}")
        # is a bad frame equal to some OTHER frame's reference?
        b, f = np.argwhere(bad.any(axis=1))[0]
        d = np.abs(ref - P[b, :, f][None, :, None]).max(axis=1) / ref.max()
        bb, ff = np.unravel_index(np.argmin(d), d.shape)
        print(f"  output (signal {b}, frame {f}) is closest to reference (signal {bb}, frame {ff}), distance {d.min():.3e}")
