"""Device time per 256 x 10 s batch: Mel-dB alone vs MFCC (fused into the Mel-dB launch at n_fft 1024 f32; elsewhere Mel-dB + the DCT-II / lifter
epilogue kernel) vs chromagram.  N_FFT / HOP from the environment (default 1024 / 256; 512 / 160 is the reference's speech default)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrograms_amd as sg
from tests import helpers as H
x = torch.from_numpy(H.cfg2_batch(256)).cuda()
st = sg.StftParams(int(os.environ.get("N_FFT", 1024)), int(os.environ.get("HOP", 256)), sg.WindowType.hanning, True)
params = sg.SpectrogramParams(st, 16000.0)
pl = sg.SpectrogramPlanner()
plans = {"mel_db_80": pl.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32"),
         "mfcc_13_of_80": pl.mfcc_plan(st, 16000.0, 80, sg.MfccParams(13), dtype="float32"),
         "mfcc_13_of_40": pl.mfcc_plan(st, 16000.0, 40, sg.MfccParams(13), dtype="float32"),
         "chroma_l2": pl.chroma_plan(st, 16000.0, sg.ChromaParams(), dtype="float32")}
for name, plan in plans.items():
    nb, nf = plan.output_shape(x.shape[1])
    out = torch.empty((x.shape[0], nb, nf), dtype=torch.float32, device="cuda")
    plan.time_batch_torch(x, out, 3)
    ms = plan.time_batch_torch(x, out, 20)
    print(f"{name:16s} {plan.kernel_name:12s} {ms * 1e3:8.1f} us", flush=True)
