"""Device time of the batched inverse STFT on the config-2 shape (256 x [513, 626] complex f32 -> 256 x 160000 f32)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
import torch
import spectrograms_amd as sg
from spectrograms_amd import _ffi
from tests import helpers as H

B = int(os.environ.get("B", 256))
x = torch.from_numpy(H.cfg2_batch(B)).cuda()
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
S = plan.compute_batch(x).contiguous()
y = plan.istft_batch(S)
torch.cuda.synchronize()
err = float((y[:, 512:159000] - x[:, 512:159000]).abs().max())
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
alg = frames * 513 * 8 + y.numel() * 4
print(json.dumps({"op": "istft", "batch": B, "ms": ms, "frames_per_s": frames / ms * 1e3, "algorithmic_GBps": alg / ms / 1e6,
                  "hbm_frac": alg / ms / 1e6 / 8000, "roundtrip_max_err": err}))
