"""Device time of the global-memory transforms (bigfft.hip): forward linear power and the inverse STFT at lengths past the on-chip
kernels, 64 x 10 s at 16 kHz with hop = n_fft / 4 (one-shot lengths: 64 frames of n_fft samples)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spectrograms_amd as sg
from spectrograms_amd import _ffi

rng = np.random.default_rng(0)
CASES = [(12000, "float64"), (12000, "float32"), (9001, "float32"), (9001, "float64"), (16385, "float32"), (20000, "float32"), (44100, "float32"),
         (65536, "float32"), (65536, "float64"), (100003, "float32"), (100003, "float64"), (1 << 20, "float32")]
if os.environ.get("CASES"):  # e.g. CASES=998:float32,514:float64 — any length: forward and inverse beside each other (twice-a-prime lengths)
    CASES = [(int(c.split(":")[0]), c.split(":")[1]) for c in os.environ["CASES"].split(",")]
for n_fft, dt in CASES:
    hop = n_fft // 4
    B, N = (64, 160000) if n_fft <= 65536 else (64, n_fft)
    tdt = torch.float32 if dt == "float32" else torch.float64
    x = torch.from_numpy(rng.standard_normal((B, N))).to(tdt).cuda()
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
    plan = sg.SpectrogramPlanner().linear_power_plan(params, dtype=dt)
    nb, nf = plan.output_shape(N)
    out = torch.empty((B, nb, nf), dtype=tdt, device="cuda")
    plan.time_batch_torch(x, out, 2)
    ms = plan.time_batch_torch(x, out, 10)
    esz = 4 if dt == "float32" else 8
    alg = (B * N + B * nb * nf) * esz
    line = f"n_fft={n_fft:8d} {dt:8s} {plan.kernel_name:14s} B={B} frames/signal={nf:3d}  forward {ms * 1e3:9.1f} us  {B * nf / ms / 1e3:9.2f} M frames/s  {alg / ms / 1e6:7.0f} GB/s algorithmic"
    if n_fft <= 65536:
        cplan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, dt)
        S = cplan.compute_batch(x).contiguous()
        y = cplan.istft_batch(S)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cplan.istft_batch(S, out=y)
        e0.record()
        for _ in range(5):
            cplan.istft_batch(S, out=y)
        e1.record(); torch.cuda.synchronize()
        m = min(y.shape[1], N) - n_fft
        err = float((y[:, n_fft:m] - x[:, n_fft:m]).abs().max())
        line += f"  inverse {e0.elapsed_time(e1) / 5 * 1e3:9.1f} us (roundtrip err {err:.1e})"
    print(line, flush=True)
