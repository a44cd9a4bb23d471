"""The tuned kernel's experiment switches (SGX_KERNEL / SGX_WIDE / SGX_SKEW / SGX_LOADS) select other code paths for the same
f32 n_fft = 1024 shapes; each is read once per process, so every variant runs in its own interpreter and is checked against
the CPU oracle there (complex STFT, linear power, Mel-dB; interior, edge and ragged last tiles; odd tile counts)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import spectrograms_amd as sg
from oracle import oracle as orc
rng = np.random.default_rng(5)
worst = 0.0
for batch, n, hop in ((3, 16000, 256), (5, 41234, 256), (2, 9000, 128), (1, 30000, 340)):
    x = (0.2 * rng.standard_normal((batch, n))).astype(np.float32)
    x[0] += 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(n) / 16000.0).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    params = sg.SpectrogramParams(sg.StftParams(1024, hop, sg.WindowType.hanning, True), 16000.0)
    pl = sg.SpectrogramPlanner()
    S = pl.stft_plan(params, dtype="float32").compute_batch(xd).cpu().numpy()
    ref = orc.stft_batch(orc.Params(n_fft=1024, hop=hop), x.astype(np.float64))
    e = np.max(np.abs(S - ref)) / np.max(np.abs(ref))
    assert S.shape == ref.shape and e < 2e-5, ("stft", batch, n, hop, e)
    P = pl.linear_power_plan(params, dtype="float32").compute_batch(xd).cpu().numpy()
    refp = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=hop), x.astype(np.float64))
    ep = np.max(np.abs(P - refp)) / np.max(refp)
    assert ep < 1e-5, ("power", batch, n, hop, ep)
    D = pl.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32").compute_batch(xd).cpu().numpy()
    refd = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=hop, n_mels=80, amp="db", floor_db=-80.0), x.astype(np.float64))
    pw = orc.spectrogram_batch(orc.Params(n_fft=1024, hop=hop, n_mels=80), x.astype(np.float64))
    msk = pw > 1e-4 * pw.max()
    ed = np.max(np.abs(D[msk] - refd[msk]))
    assert ed < 1e-3, ("mel_db", batch, n, hop, ed)
    worst = max(worst, e, ep)
    # inverse STFT of the oracle-checked spectrum: the fused kernels against the CPU oracle's istft
    from spectrograms_amd import _ffi
    iplan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
    y = iplan.istft_batch(torch.from_numpy(S).cuda()).cpu().numpy()
    yref = np.stack([orc.istft(S[i].astype(np.complex128), 1024, hop, "hanning", True) for i in range(batch)])
    ei = np.max(np.abs(y - yref)) / max(np.max(np.abs(yref)), 1e-30)
    assert y.shape == yref.shape and ei < 2e-5, ("istft", batch, n, hop, ei)
torch.cuda.synchronize()
print("ok", worst)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {},                                   # defaults: wide pass 2, staged loads
    {"SGX_WIDE": "0"},                    # per-half pass 2
    {"SGX_WIDE": "0", "SGX_SKEW": "1"},   # halves one barrier apart
    {"SGX_LOADS": "direct"},              # per-lane sample loads
    {"SGX_KERNEL": "single"},             # two independent 256-thread workgroups per CU
    {"SGX_KERNEL": "q"},                  # 16 values per lane, radix-2 / real split across lanes by DPP
    {"SGX_ISTFT": "a"},                   # first fused inverse-STFT kernel (every spectrum pair loaded by two lanes)
    {"SGX_ISTFT_GENERIC": "1"},           # C2R rows + gather overlap-add instead of the fused kernel
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "default")
def test_switch_variant_matches_oracle(env):
    full = dict(os.environ)
    full.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=full, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:]
