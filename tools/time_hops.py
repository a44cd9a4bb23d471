"""Device time per B x 10 s batch (B=256, DTYPE=float32 by default) of the tuned kernels over hops (linear power, Mel-80 dB); N_FFT=512 HOPS=64,128,160 for the
two-frames-per-transform mode."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrograms_amd as sg
from tests import helpers as H
DT = os.environ.get("DTYPE", "float32"); B = int(os.environ.get("B", 256))
x = torch.from_numpy(H.cfg2_batch(B)).cuda()
if DT == "float64":
    x = x.double()
N_FFT = int(os.environ.get("N_FFT", 1024))
for hop in [int(h) for h in os.environ.get("HOPS", "64,128,160,200,256,270,320,512").split(",")]:
    params = sg.SpectrogramParams(sg.StftParams(N_FFT, hop, sg.WindowType.hanning, True), 16000.0)
    for name, plan in (("linear", sg.SpectrogramPlanner().linear_power_plan(params, dtype=DT)),
                       ("mel80db", sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype=DT))):
        nb, nf = plan.output_shape(x.shape[1])
        out = torch.empty((B, nb, nf), dtype=x.dtype, device="cuda")
        plan.time_batch_torch(x, out, 3)
        ms = plan.time_batch_torch(x, out, int(os.environ.get("ITERS", 20)))
        print(f"n_fft={N_FFT} hop={hop:4d} {name:8s} {plan.kernel_name:12s} {ms * 1e3:9.1f} us  {B * nf / ms / 1e3:8.1f} M frames/s", flush=True)
