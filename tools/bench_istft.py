"""Device time of the batched inverse STFT on the config-2 shape (256 x [513, 626] complex f32 -> 256 x 160000 f32)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
import torch
import spectrograms_amd as sg
from spectrograms_amd import _ffi
from tests import helpers as H

B = int(os.environ.get("B", 256))
NS = int(os.environ.get("NS", 0))   # samples per signal other than config 2's 160000 (e.g. 159488 -> 624 frames: bin rows 128-B aligned)
x = torch.from_numpy(H.cfg2_batch(B) if not NS else np.random.default_rng(3).standard_normal((B, NS)).astype(np.float32)).cuda()
if os.environ.get("DTYPE", "float32") == "float64":
    x = x.double()
N_FFT = int(os.environ.get("N_FFT", 1024)); HOP = int(os.environ.get("HOP", N_FFT // 4)); DT = os.environ.get("DTYPE", "float32")
params = sg.SpectrogramParams(sg.StftParams(N_FFT, HOP, sg.WindowType.hanning, True), 16000.0)
plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, DT)
S = plan.compute_batch(x).contiguous()
y = plan.istft_batch(S)
torch.cuda.synchronize()
m = min(y.shape[1], x.shape[1]) - N_FFT
err = float((y[:, N_FFT:m] - x[:, N_FFT:m]).abs().max())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 20
for _ in range(3):
    plan.istft_batch(S, out=y)
e0.record()
for _ in range(iters):
    plan.istft_batch(S, out=y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
frames = B * S.shape[2]
esz = 4 if DT == "float32" else 8
alg = frames * (N_FFT // 2 + 1) * 2 * esz + y.numel() * esz
print(json.dumps({"op": "istft", "n_fft": N_FFT, "hop": HOP, "dtype": DT, "batch": B, "frames_per_signal": int(S.shape[2]), "ms": ms, "frames_per_s": frames / ms * 1e3, "algorithmic_GBps": alg / ms / 1e6,
                  "hbm_frac": alg / ms / 1e6 / 8000, "roundtrip_max_err": err}))
