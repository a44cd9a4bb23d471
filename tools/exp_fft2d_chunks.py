#!/usr/bin/env python3
"""Experiment: does walking the 512-image batch in small chunks on TWO streams — so that chunk c's column pass
overlaps chunk c + 1's row pass and no launch leaves the chip under-filled — keep the intermediate half spectrum in the 256 MiB
Infinity Cache?  fft2d and convolve_fft, 512 x 1024 x 1024 f32, chunk sizes 8 ... 128 against the single call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import spectrograms_amd as sg

B, R, C = 512, 1024, 1024
x = torch.randn((B, R, C), dtype=torch.float32, device="cuda")
k = sg.gaussian_kernel_2d(9, 2.0, dtype="float32")
plans = [sg.Fft2dPlan(R, C, "float32") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
spec = torch.empty((B, R, C // 2 + 1, 2), dtype=torch.float32, device="cuda")
y = torch.empty_like(x)


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def whole(op):
    if op == "fft2d":
        plans[0].forward_torch(x, spec)
    else:
        plans[0].convolve_torch(x, k, y)


def chunked(op, n, nstreams):
    cur = torch.cuda.current_stream()
    for s in streams[:nstreams]:
        s.wait_stream(cur)
    for i, c in enumerate(range(0, B, n)):
        s = streams[i % nstreams]
        with torch.cuda.stream(s):
            if op == "fft2d":
                plans[i % nstreams].forward_torch(x[c:c + n], spec[c:c + n])
            else:
                plans[i % nstreams].convolve_torch(x[c:c + n], k, y[c:c + n])
    for s in streams[:nstreams]:
        cur.wait_stream(s)


for op in ("fft2d", "convolve_fft"):
    print(f"{op}: single call {timed(lambda: whole(op)):.3f} ms", flush=True)
    for n in (8, 16, 24, 32, 64, 128):
        for ns in (1, 2):
            print(f"{op}: chunks of {n:3d} on {ns} stream(s): {timed(lambda: chunked(op, n, ns)):.3f} ms", flush=True)
