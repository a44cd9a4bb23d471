"""Device time per 256 x 10 s batch of the tuned kernel for each frequency mapping."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import spectrograms_amd as sg
from spectrograms_amd import _ffi
from tests import helpers as H
x = torch.from_numpy(H.cfg2_batch(256)).cuda()
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
for name, mp in [("erb64", sg.ErbParams(64, 0.0, 8000.0)), ("erb40", sg.ErbParams(40, 0.0, 8000.0)), ("loghz128", sg.LogHzParams(128, 20.0, 8000.0)), ("mel80", sg.MelParams(80, 0.0, 8000.0))]:
    plan = sg.Plan(params, _ffi.AMP_POWER, mp, None, "float32")
    nb, nf = plan.output_shape(x.shape[1])
    out = torch.empty((x.shape[0], nb, nf), dtype=torch.float32, device='cuda')
    plan.time_batch_torch(x, out, 2)
    ms = plan.time_batch_torch(x, out, 10)
    print(name, plan.kernel_name, "%.1f us" % (ms * 1e3), flush=True)
