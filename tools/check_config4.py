"""BASELINE config 4 per-GPU shard on one device: 1024 x 10 s utterances, Mel-80 power (and linear power), timed and checked
against the oracle on a few utterances."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
import spectrograms_amd as sg
from oracle import oracle as orc
from tests import helpers as H
B = 1024
x = torch.from_numpy(np.stack([H.cfg2_signal(b) for b in range(B)])).cuda()
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
plan = sg.SpectrogramPlanner().mel_power_plan(params, sg.MelParams(80, 0.0, 8000.0), dtype="float32")
out = plan.compute_batch(x)
torch.cuda.synchronize()
ms = plan.time_batch_torch(x, out, 10)
print("config-4 shard (1024 x 10 s, Mel-80 power): %.3f ms = %.1f M frames/s" % (ms, B * out.shape[2] / ms / 1e3))
idx = [0, 1, 511, 512, 1023]
ref = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=256, n_mels=80), x[idx].cpu().numpy().astype(np.float64))
got = out[idx].cpu().numpy()
print("rel err vs oracle:", H.rel_err(got, ref))
lin = sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32")
o2 = lin.compute_batch(x)
ms2 = lin.time_batch_torch(x, o2, 5)
print("linear power 1024 x 10 s: %.3f ms = %.1f M frames/s" % (ms2, B * o2.shape[2] / ms2 / 1e3))
ref2 = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=256), x[[1023]].cpu().numpy().astype(np.float64))
print("rel err vs oracle:", H.rel_err(o2[[1023]].cpu().numpy(), ref2))
