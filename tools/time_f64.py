import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import spectrograms_amd as sg
from tests import helpers as H
B = int(os.environ.get("B", 64))
x = torch.from_numpy(H.cfg2_batch(B)).double().cuda()
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
P = sg.SpectrogramPlanner()
mel = sg.MelParams(80, 0.0, 8000.0)
for name, plan in (("linear_power", P.linear_power_plan(params, dtype="float64")), ("linear_db", P.linear_db_plan(params, sg.LogParams(-80.0), dtype="float64")),
                   ("mel_power", P.mel_power_plan(params, mel, dtype="float64")), ("mel_db", P.mel_db_plan(params, mel, sg.LogParams(-80.0), dtype="float64")),
                   ("mel128_db", P.mel_db_plan(params, sg.MelParams(128, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float64")), ("stft", P.stft_plan(params, dtype="float64"))):
    nb, nf = plan.output_shape(x.shape[1])
    out = torch.empty((B, nb, nf) + ((2,) if name == "stft" else ()), dtype=torch.float64, device="cuda")
    plan.time_batch_torch(x, out, 3)
    ms = plan.time_batch_torch(x, out, 10)
    print(f"f64 1024/256 B={B} {name:12s} {plan.kernel_name:11s} {ms*1e3:8.1f} us {B*nf/ms/1e3:8.1f} M frames/s", flush=True)
