"""Generate golden vectors from the REFERENCE's own Python restatement.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports /root/reference/python/examples/numpy_impls.py — whose `stft`
(:6-31), `hann_window` (:34-36), `power_spectrogram` / `magnitude_spectrogram`
(:39-44) have the same semantics as the Rust hot path (zero padding of n_fft//2,
frame count, symmetric Hann, unnormalised rfft, (bins, frames) layout; SURVEY.md
§8c), and whose `log_frequency_matrix` / `logfreq_spectrogram` (:94-123) and `erb_centers` /
`gammatone_response` / `erb_spectrogram` (:126-159) have those of build_loghz_matrix (src/spectrogram.rs:2438-2508) and of the
linear-spaced ErbFilterbank (src/erb.rs:266-403) — evaluates it on the BASELINE workloads and stores INPUT DESCRIPTIONS +
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

    # --- frequency mappings from the reference's own NumPy restatement --------------------------------
    # log_frequency_matrix / logfreq_spectrogram (:94-123) have the semantics of build_loghz_matrix (src/spectrogram.rs:2438-2508;
    # f_max below Nyquist, so the `.min(out_len - 1)` clamp at :2488 is not reached); erb / erb_to_hz / erb_centers /
    # gammatone_response / erb_spectrogram (:126-159) those of ErbFilterbank::generate with ErbSpacing::Linear and
    # apply_to_power_spectrum (src/erb.rs:206-210, 266-335, 374-403): sum_k |G(f_k) X_k|^2 = sum_k |H(f_k)|^2 |X_k|^2.
    # NOT taken: numpy_impls.chroma (:198-215) — hard nearest-pitch-class assignment and per-frame sum normalisation, where
    # build_chroma_filterbank (src/chroma.rs:262-345) spreads every bin over the 12 classes with a Gaussian of one semitone and
    # normalises the filterbank rows: different semantics, like its mel_filterbank / db_spectrogram / mfcc.
    w = ref.hann_window(1024)
    lh = dict(n_bins=96, f_min=30.0, f_max=7900.0)
    eb = dict(n_filters=64, f_min=50.0, f_max=7800.0)
    out_l, out_e = {}, {}
    M = ref.log_frequency_matrix(sr, 1024, lh["n_bins"], lh["f_min"], lh["f_max"])
    out_l["matrix"] = M
    out_l["params"] = np.asarray([lh["n_bins"], lh["f_min"], lh["f_max"]], dtype=np.float64)
    cf = ref.erb_centers(eb["f_min"], eb["f_max"], eb["n_filters"])
    out_e["centres"] = cf
    out_e["params"] = np.asarray([eb["n_filters"], eb["f_min"], eb["f_max"]], dtype=np.float64)
    for b in (0, 1):
        x = cfg2_signal(b)
        S, freqs, _ = ref.stft(x, sr, 1024, 256, w, centre=True)
        sub = frame_subset(S.shape[1])
        P = ref.power_spectrogram(S)
        L = ref.logfreq_spectrogram(P, M)
        out_l[f"c2_b{b}_frames"] = sub
        out_l[f"c2_b{b}_loghz_power"] = L[:, sub]
        out_l[f"c2_b{b}_loghz_rowsum"] = L.sum(axis=1)
        E = ref.erb_spectrogram(S, freqs, cf)
        out_e[f"c2_b{b}_frames"] = sub
        out_e[f"c2_b{b}_erb_power"] = E[:, sub]
        out_e[f"c2_b{b}_erb_rowsum"] = E.sum(axis=1)
    # the |H(f)|^2 matrix itself, as the reference's gammatone_response gives it
    out_e["matrix_cols"] = np.arange(0, 513, 4)
    out_e["matrix"] = np.stack([np.abs(ref.gammatone_response(freqs, fc)) ** 2 for fc in cf])[:, ::4]  # every 4th bin: < 200 KB
    np.savez_compressed(os.path.join(HERE, "loghz_ref.npz"), **out_l)
    np.savez_compressed(os.path.join(HERE, "erb_ref.npz"), **out_e)
    for f in ("config1_ref.npz", "config2_ref.npz", "short_ref.npz", "loghz_ref.npz", "erb_ref.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
