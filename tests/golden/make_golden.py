"""Generate golden vectors from the REFERENCE's own Python restatement.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports /root/reference/python/examples/numpy_impls.py — whose `stft`
(:6-31), `hann_window` (:34-36), `power_spectrogram` / `magnitude_spectrogram`
(:39-44) have the same semantics as the Rust hot path (zero padding of n_fft//2,
frame count, symmetric Hann, unnormalised rfft, (bins, frames) layout; SURVEY.md
§8c) — evaluates it on the BASELINE workloads and stores INPUT DESCRIPTIONS +
EXPECTED OUTPUTS as small .npz fixtures next to this script.  No reference source
text is stored.  The fixtures are what travels to the GPU box; the reference
does not.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/python/examples/numpy_impls.py"


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_numpy_impls", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cfg2_signal(b: int, n: int = 160000, sr: float = 16000.0) -> np.ndarray:
    """BASELINE.md §2 config-2 generator (row b of the batch), f32."""
    if b % 2 == 0:
        f = 110.0 * 2.0 ** ((b % 48) / 12.0)
        i = np.arange(n, dtype=np.float64)
        return (0.5 * np.sin(2.0 * np.pi * f * i / sr)).astype(np.float32)
    rng = np.random.default_rng(1234 + b)
    return (0.1 * rng.standard_normal(n)).astype(np.float32)


def frame_subset(n_frames: int) -> np.ndarray:
    idx = sorted(set(list(range(0, 3)) + list(range(0, n_frames, 25)) + list(range(n_frames - 3, n_frames))))
    return np.asarray(idx, dtype=np.int64)


def main():
    ref = load_ref()
    out = {}

    # --- config 1: 1 s 16 kHz 440 Hz sine, f64, Hanning, centre (examples/basic_linear.rs:25-30 uses 512/256;
    #     BASELINE.json says 256/128) -------------------------------------------------------------------------
    sr = 16000
    x1 = np.sin(2.0 * np.pi * 440.0 * np.arange(16000, dtype=np.float64) / sr)
    for n_fft, hop in ((512, 256), (256, 128)):
        w = ref.hann_window(n_fft)
        S, freqs, _times = ref.stft(x1, sr, n_fft, hop, w, centre=True)
        out[f"c1_{n_fft}_{hop}_stft"] = S.astype(np.complex128)
        out[f"c1_{n_fft}_{hop}_power"] = ref.power_spectrogram(S)
        out[f"c1_{n_fft}_{hop}_magnitude"] = ref.magnitude_spectrogram(S)
        out[f"c1_{n_fft}_{hop}_freqs"] = freqs
        out[f"hann_{n_fft}"] = w
        # centre=False variant (no padding, trailing samples dropped)
        S2, _, _ = ref.stft(x1, sr, n_fft, hop, w, centre=False)
        out[f"c1_{n_fft}_{hop}_nocentre_power"] = ref.power_spectrogram(S2)
    out["hann_8"] = ref.hann_window(8)
    out["hann_1024"] = ref.hann_window(1024)
    np.savez_compressed(os.path.join(HERE, "config1_ref.npz"), **out)

    # --- config 2: rows 0 (sine 110 Hz) and 1 (noise seed 1235) of the 256 x 10 s f32 batch, 1024/256 --------
    out = {}
    w = ref.hann_window(1024)
    for b in (0, 1):
        x = cfg2_signal(b)
        S, _, _ = ref.stft(x, sr, 1024, 256, w, centre=True)  # reference computes in f64
        assert S.shape == (513, 626), S.shape
        sub = frame_subset(S.shape[1])
        out[f"c2_b{b}_frames"] = sub
        out[f"c2_b{b}_stft"] = S[:, sub].astype(np.complex128)
        out[f"c2_b{b}_power_rowsum"] = ref.power_spectrogram(S).sum(axis=1)  # all frames, per-bin checksum
        out[f"c2_b{b}_x_head"] = x[:64].copy()
        out[f"c2_b{b}_x_sum"] = np.float64(x.astype(np.float64).sum())
    np.savez_compressed(os.path.join(HERE, "config2_ref.npz"), **out)

    # --- short / ragged inputs (tests/spectrogram_tests.rs:112-121 — 5 samples -> 1 frame) ------------------
    out = {}
    rng = np.random.default_rng(99)
    for n in (5, 300, 511, 512, 513, 1000):
        x = rng.standard_normal(n)
        w = ref.hann_window(512)
        if n + 512 >= 512:
            S, _, _ = ref.stft(x, sr, 512, 256, w, centre=True)
            out[f"short_{n}_x"] = x
            out[f"short_{n}_stft"] = S.astype(np.complex128)
    np.savez_compressed(os.path.join(HERE, "short_ref.npz"), **out)
    for f in ("config1_ref.npz", "config2_ref.npz", "short_ref.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
