#!/usr/bin/env python3
"""Runs the hot path a few times on device-resident synthetic data so rocprofv3 can attach:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_driver.py linear_power 5
Same workload as bench.py (BASELINE configs[1]/[2])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import bench
import spectrograms_amd as sg


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "linear_power"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    # SGX_PROF_NFFT / SGX_PROF_HOP / SGX_PROF_DTYPE: profile the shape-generic kernels on the same signals
    n_fft = int(os.environ.get("SGX_PROF_NFFT", bench.N_FFT))
    hop = int(os.environ.get("SGX_PROF_HOP", n_fft // 4))
    dtype = os.environ.get("SGX_PROF_DTYPE", "float32")
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), bench.SR)
    if workload in ("fft2d", "convolve_fft"):  # BASELINE configs[4]: every step is the leg's whole chain of launches
        dev = torch.device("cuda", 0)
        x = bench.make_images(torch, dev)
        plan = sg.Fft2dPlan(bench.IMG_SIDE, bench.IMG_SIDE, "float32")
        k = sg.gaussian_kernel_2d(9, 2.0, dtype="float32")
        buf = plan.forward_torch(x) if workload == "fft2d" else plan.convolve_torch(x, k)  # (iteration 1: also builds the cached kernel spectrum)
        for _ in range(iters - 1):
            plan.forward_torch(x, buf) if workload == "fft2d" else plan.convolve_torch(x, k, buf)
        torch.cuda.synchronize()
        print(workload, tuple(buf.shape), "iters", iters)
        return
    if workload == "istft":
        from spectrograms_amd import _ffi
        plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
        x = torch.from_numpy(np.stack([bench.cfg_signal(b) for b in range(batch)])).cuda()
        S = plan.compute_batch(x).contiguous()
        y = None
        for _ in range(iters):
            y = plan.istft_batch(S, out=y)
        torch.cuda.synchronize()
        print(workload, tuple(y.shape), "iters", iters)
        return
    pl = sg.SpectrogramPlanner()
    if workload == "config4":  # BASELINE configs[3]'s per-GPU shard: 1024 utterances, Mel-80 power
        workload, batch = "mel_power", 1024
    if workload.endswith("_f64"):  # bench.py's f64 legs: configs[1] / [2] in the reference's other Sample type
        workload, dtype = workload[:-4], "float64"
    if workload == "linear_power":
        plan = pl.linear_power_plan(params, dtype=dtype)
    elif workload == "mel_power":
        plan = pl.mel_power_plan(params, sg.MelParams(80, 0.0, 8000.0), dtype="float32")
    elif workload == "stft":
        plan = pl.stft_plan(params, dtype="float32")
    elif workload == "mfcc":  # Mel-80 dB -> DCT-II (13) + lifter 22 in the same launch (kernels_r32x16.hip mfcc_tile)
        plan = pl.mfcc_plan(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), bench.SR, 80, sg.MfccParams(13), dtype="float32")
    elif workload == "linear_db":
        plan = pl.linear_db_plan(params, sg.LogParams(-80.0), dtype=dtype)
    elif workload == "erb_power":
        plan = pl.erb_power_plan(params, sg.ErbParams(64, 0.0, 8000.0), dtype="float32")
    else:
        plan = pl.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype=dtype)
    host = np.stack([bench.cfg_signal(b) for b in range(batch)]).astype(np.float32 if dtype == "float32" else np.float64)
    x = torch.from_numpy(host).cuda()
    out = None
    for _ in range(iters):
        out = plan.compute_batch(x, out=None if out is None else (torch.view_as_real(out) if out.is_complex() else out))
    torch.cuda.synchronize()
    print(workload, plan.kernel_name, tuple(out.shape))


if __name__ == "__main__":
    main()
