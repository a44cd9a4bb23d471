#!/usr/bin/env python3
"""Debug aid: compares the tuned kernel's outputs with the CPU oracle and prints where they differ (frames / bins)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectrograms_amd as sg
from oracle import oracle as orc

rng = np.random.default_rng(3)
for (B, N, hop) in ((3, 20000, 256), (2, 9000, 128), (1, 30000, 340)):
    x = (0.2 * rng.standard_normal((B, N))).astype(np.float32)
    x[0] += 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(N) / 16000.0).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    params = sg.SpectrogramParams(sg.StftParams(1024, hop, sg.WindowType.hanning, True), 16000.0)
    pl = sg.SpectrogramPlanner()
    for name, plan, op in (
        ("linear", pl.linear_power_plan(params, dtype="float32"), orc.Params(n_fft=1024, hop=hop)),
        ("mel", pl.mel_power_plan(params, sg.MelParams(80, 0.0, 8000.0), dtype="float32"), orc.Params(n_fft=1024, hop=hop, n_mels=80)),
    ):
        P = plan.compute_batch(xd).cpu().numpy()
        ref = orc.spectrogram_batch(op, x.astype(np.float64))
        err = np.abs(P - ref) / ref.max()
        print(f"{name} B={B} N={N} hop={hop} shape={P.shape} max rel err {err.max():.3e}")
        if err.max() > 1e-4:
            bad = err > 1e-4
            print("  bad frames per signal:", [np.nonzero(bad[b].any(axis=0))[0][:40].tolist() for b in range(B)])
            print("  bad bins (first signal with errors):", np.nonzero(bad.any(axis=(0, 2)))[0][:60].tolist(), "count", int(bad.any(axis=(0, 2)).sum()))
    S = pl.stft_plan(params, dtype="float32").compute_batch(xd).cpu().numpy()
    refs = orc.stft_batch(orc.Params(n_fft=1024, hop=hop), x.astype(np.float64))
    e = np.abs(S - refs) / np.abs(refs).max()
    print(f"stft B={B} N={N} hop={hop} max rel err {e.max():.3e}")
    if e.max() > 1e-4:
        bad = e > 1e-4
        print("  bad frames:", [np.nonzero(bad[b].any(axis=0))[0][:40].tolist() for b in range(B)])
        print("  bad bins:", np.nonzero(bad.any(axis=(0, 2)))[0][:60].tolist(), "count", int(bad.any(axis=(0, 2)).sum()))
