"""Device time per 256 x 10 s batch of the tuned f32 kernel over hops (linear power, Mel-80 dB); N_FFT=512 HOPS=64,128,160 for the
two-frames-per-transform mode."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrograms_amd as sg
from tests import helpers as H
x = torch.from_numpy(H.cfg2_batch(256)).cuda()
N_FFT = int(os.environ.get("N_FFT", 1024))
for hop in [int(h) for h in os.environ.get("HOPS", "64,128,160,200,256,270,320,512").split(",")]:
    params = sg.SpectrogramParams(sg.StftParams(N_FFT, hop, sg.WindowType.hanning, True), 16000.0)
    for name, plan in (("linear", sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32")),
                       ("mel80db", sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32"))):
        nb, nf = plan.output_shape(x.shape[1])
        out = torch.empty((256, nb, nf), dtype=torch.float32, device="cuda")
        plan.time_batch_torch(x, out, 3)
        ms = plan.time_batch_torch(x, out, int(os.environ.get("ITERS", 20)))
        print(f"n_fft={N_FFT} hop={hop:4d} {name:8s} {plan.kernel_name:12s} {ms * 1e3:9.1f} us  {256 * nf / ms / 1e3:8.1f} M frames/s", flush=True)
