#!/usr/bin/env python3
"""BASELINE config 5: batch of 512 x 1024x1024 f32 images — fft2d alone and convolve_fft with gaussian_kernel_2d(9, 2.0).
Device-resident inputs/outputs, HIP-event timing on the launch stream; prints images/s and algorithmic GB/s
(fft2d: 4 MiB read + 1024*513*8 B written per image; convolve: 4 MiB read + 4 MiB written per image — BASELINE.md §3)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import spectrograms_amd as sg


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    R = C = int(os.environ.get("SIDE", 1024))
    rng = np.random.default_rng(7)
    r, c = np.meshgrid(np.arange(R), np.arange(C), indexing="ij")
    base = (np.sin(0.01 * r) + np.cos(0.02 * c)).astype(np.float32)
    # inputs are built on the host and uploaded with one copy, so a profile of this script holds the engine's kernels only
    nz = min(batch, 32)
    host = (base[None] + 0.05 * rng.standard_normal((nz, R, C), dtype=np.float32))
    host = np.concatenate([host] * ((batch + nz - 1) // nz))[:batch]
    x = torch.from_numpy(np.ascontiguousarray(host)).cuda()
    plan = sg.Fft2dPlan(R, C, "float32")
    k = sg.gaussian_kernel_2d(9, 2.0, dtype="float32")
    spec = plan.forward_torch(x)
    y = plan.convolve_torch(x, k)
    torch.cuda.synchronize()
    res = {}
    for name, fn, bytes_per_img in (("fft2d", lambda: plan.forward_torch(x, spec), R * C * 4 + R * (C // 2 + 1) * 8),
                                    ("convolve_fft", lambda: plan.convolve_torch(x, k, y), 2 * R * C * 4)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res[name] = {"ms_per_batch": ms, "images_per_s": batch / (ms * 1e-3), "algorithmic_GBps": batch * bytes_per_img / (ms * 1e-3) / 1e9,
                     "frac_of_8TBps": batch * bytes_per_img / (ms * 1e-3) / 8e12}
    print(json.dumps({"workload": f"configs[4]-like: {batch} x {R}x{C} f32", **res}))


if __name__ == "__main__":
    main()
