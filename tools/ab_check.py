#!/usr/bin/env python3
"""A/B correctness gate for kernel variants: runs Mel-80 power, Mel-80 dB, linear power and the complex STFT (f32, n_fft 1024, hop 256,
6 x 1 s signals + one 10 s signal) on the library named by SGX_LIB_PATH and compares with the f64 oracle at the tolerances of
tests/test_gpu_parity.py; with a second argument, also saves the outputs so that two variants can be compared bit for bit:
    SGX_LIB_PATH=build/libsgx_x.so python tools/ab_check.py x        -> gpurun_out/ab_x.npz
    python tools/ab_check.py --diff a b"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    if sys.argv[1] == "--diff":
        a, b = (np.load(os.path.join(ROOT, "gpurun_out", f"ab_{n}.npz")) for n in sys.argv[2:4])
        for k in a.files:
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            print(f"{k}: bit-equal={np.array_equal(a[k], b[k])} max|diff|/max|a|={d.max() / max(np.abs(a[k]).max(), 1e-300):.3e}")
        return
    import torch

    import bench
    import spectrograms_amd as sg
    from oracle import oracle as orc

    name = sys.argv[1]
    x1 = np.stack([bench.cfg_signal(b)[:16000] for b in range(6)])
    x10 = np.stack([bench.cfg_signal(1), bench.cfg_signal(2)])
    out = {}
    ok = True
    for tag, x in (("1s", x1), ("10s", x10)):
        xd = torch.from_numpy(x).cuda()
        for wl in ("mel_power", "mel_db", "linear_power", "stft"):
            plan = bench.make_plan(sg, wl)
            y = plan.compute_batch(xd)
            torch.cuda.synchronize()
            y = y.cpu().numpy()
            out[f"{wl}_{tag}"] = y
            if wl == "stft":
                ref = orc.stft_batch(orc.Params(n_fft=1024, hop=256), x.astype(np.float64))
                err = np.abs(y - ref).max() / np.abs(ref).max()
                good = err < 2e-5
            else:
                op = {"mel_power": orc.Params(n_fft=1024, hop=256, n_mels=80), "linear_power": orc.Params(n_fft=1024, hop=256),
                      "mel_db": orc.Params(n_fft=1024, hop=256, n_mels=80, amp="db", floor_db=-80.0)}[wl]
                ref = orc.spectrogram_batch(op, x.astype(np.float64))
                if wl == "mel_db":
                    pw = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=256, n_mels=80), x.astype(np.float64))
                    m = pw > 1e-4 * pw.max()
                    err = np.abs(y[m] - ref[m]).max()
                    good = err < 1e-3
                else:
                    m = ref > 1e-4 * ref.max()
                    err = (np.abs(y[m] - ref[m]) / ref[m]).max()
                    good = err < 1e-4
            ok = ok and good
            print(f"{name} {wl} {tag}: kernel={plan.kernel_name} err={err:.3e} {'ok' if good else 'FAIL'}")
    np.savez(os.path.join(ROOT, "gpurun_out", f"ab_{name}.npz"), **out)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
