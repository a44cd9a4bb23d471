#!/usr/bin/env python3
"""2-D shape sweep (slow corners, not headline numbers): fft2d / ifft2d / convolve_fft over image shapes, ~128 Mpixel per batch,
device-resident, HIP events; prints ms per batch and algorithmic GB/s (image in + half spectrum out; convolve: image in + out)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import spectrograms_amd as sg


def timed(fn, iters=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for dtype in os.environ.get("DTYPES", "float32,float64").split(","):
    tdt = torch.float32 if dtype == "float32" else torch.float64
    es = 4 if dtype == "float32" else 8
    shapes = ((64, 64), (100, 50), (256, 256), (512, 512), (1000, 1000), (1024, 1000), (1000, 1024), (1023, 1023), (1024, 1024), (2048, 2048), (4096, 4096), (37, 1024),
              (1024, 37), (4096, 64))
    if os.environ.get("SHAPES"):  # e.g. SHAPES=1023x1023,509x509
        shapes = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(","))
    for R, C in shapes:
        batch = max(1, (1 << 27) // (R * C) // (es // 4))
        x = torch.randn((batch, R, C), dtype=tdt, device="cuda")
        plan = sg.Fft2dPlan(R, C, dtype)
        k = sg.gaussian_kernel_2d(5, 1.5, dtype=dtype)
        spec = plan.forward_torch(x)
        y = torch.empty_like(x)
        tf = timed(lambda: plan.forward_torch(x, spec))
        ti = timed(lambda: plan.inverse_torch(spec, y))
        tc = timed(lambda: plan.convolve_torch(x, k, y))
        bf = (x.numel() * es + spec.numel() * 2 * es) / 1e6
        bc = 2 * x.numel() * es / 1e6
        print(f"{dtype} {R:5d} x {C:5d} batch {batch:6d}: fft2d {tf:8.3f} ms {bf / tf:6.0f} GB/s   ifft2d {ti:8.3f} ms {bf / ti:6.0f} GB/s   convolve {tc:8.3f} ms {bc / tc:6.0f} GB/s", flush=True)
        del x, spec, y, plan
        torch.cuda.empty_cache()
