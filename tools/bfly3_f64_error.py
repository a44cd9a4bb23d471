#!/usr/bin/env python3
"""f64 accuracy of the three-instruction twiddled butterfly (fft_inreg.h, SGX_BFLY3: x1 = 2 e - x0 carries x0's rounding into x1)
against the four-instruction form, over the sizes the f64 paths serve: complex STFT of seeded noise + a loud tone (bins
next to a strong one are where e ~ w o), max and RMS error against the f64 oracle, relative to max|X|.
    SGX_LIB_PATH=build/libsgx_nobfly3.so python tools/bfly3_f64_error.py four     # -DSGX_BFLY3=0 in every transform source
    python tools/bfly3_f64_error.py three                                            # the product library"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    import spectrograms_amd as sg
    from oracle import oracle as orc

    tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
    rng = np.random.default_rng(4242)
    rows = []
    for n_fft in (64, 256, 400, 512, 1000, 1024, 1440, 2048, 4096, 8192, 251, 1009, 2003):
        hop = n_fft // 4
        n = 6 * n_fft if n_fft <= 2048 else 3 * n_fft
        t = np.arange(n)
        x = np.stack([0.1 * rng.standard_normal(n) + np.sin(2 * np.pi * (0.1234 + 0.05 * b) * t) for b in range(2)])
        params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), 16000.0)
        plan = sg.SpectrogramPlanner().stft_plan(params, dtype="float64")
        got = np.asarray(plan.compute_batch(x))
        ref = orc.stft_batch(orc.Params(n_fft=n_fft, hop=hop), x)
        d = np.abs(got - ref)
        s = np.abs(ref).max()
        # relative error of the small bins on their own scale: where the carried rounding would show
        small = np.abs(ref) < 1e-3 * s
        rel_small = float((d[small] / np.maximum(np.abs(ref[small]), 1e-300)).max()) if small.any() else 0.0
        rows.append((n_fft, plan.kernel_name, d.max() / s, np.sqrt((d ** 2).mean()) / s, rel_small))
        print(f"{tag} n_fft={n_fft:5d} kernel={plan.kernel_name:12s} max|err|/max|X|={rows[-1][2]:.3e} rms/max|X|={rows[-1][3]:.3e} "
              f"max rel err of bins below 1e-3 max|X|={rel_small:.3e}", flush=True)
    print(f"{tag} worst: max {max(r[2] for r in rows):.3e} rms {max(r[3] for r in rows):.3e} (test bound 1e-10)")


if __name__ == "__main__":
    main()
