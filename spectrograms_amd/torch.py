"""PyTorch convenience layer with the reference's names (python/spectrograms/torch.py): `to_torch`, `TorchSpectrogram`,
`batch`, `batch_with_metadata` — plus the device-resident fast path the HIP engine makes possible.

Reference behaviour kept: `batch([...])` converts every spectrogram through DLPack, optionally pads to a common shape, stacks
and moves to `device` (torch.py:200-285).  Added: a `SpectrogramBatch` (from `Plan.compute_batch_resident`) is already one
[B, n_bins, n_frames] device tensor, so `batch(sb)` is a zero-copy view, and `batch_signals(plan, signals)` computes the
whole batch on the GPU in one launch instead of per-signal host transforms.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .planner import Mfcc, Plan, Spectrogram, SpectrogramBatch, StftResult


@dataclass
class TorchSpectrogram:
    """torch.py:44-87: tensor + preserved metadata."""
    tensor: torch.Tensor
    frequencies: Optional[np.ndarray] = None
    times: Optional[np.ndarray] = None
    params: Optional[object] = None
    shape: Optional[tuple] = None
    db_range: Optional[tuple] = None

    def to(self, device) -> "TorchSpectrogram":
        return TorchSpectrogram(self.tensor.to(device), self.frequencies, self.times, self.params, self.shape, self.db_range)

    def cpu(self) -> "TorchSpectrogram":
        return self.to("cpu")

    def cuda(self, device: Optional[int] = None) -> "TorchSpectrogram":
        return self.to(f"cuda:{device}" if device is not None else "cuda")


def _spectrogram_to_torch(self, device="cpu", with_metadata: bool = False, dtype: Optional[torch.dtype] = None):
    """torch.py:127-197."""
    tensor = torch.from_dlpack(self)
    if str(device) != "cpu" or dtype is not None:
        tensor = tensor.to(device=device, dtype=dtype if dtype is not None else tensor.dtype)
    if not with_metadata:
        return tensor
    dbr = self.db_range() if hasattr(self, "db_range") and callable(self.db_range) else None
    return TorchSpectrogram(tensor, np.array(self.frequencies), np.array(self.times), self.params, tuple(self.shape), dbr)


Spectrogram.to_torch = _spectrogram_to_torch  # `import spectrograms_amd.torch` adds the method, as the reference module does


def batch(spectrograms, device="cpu", dtype: Optional[torch.dtype] = None, pad: bool = False) -> torch.Tensor:
    """torch.py:200-285.  A `SpectrogramBatch` is returned as its own device tensor (zero-copy) when device/dtype agree."""
    if isinstance(spectrograms, SpectrogramBatch):
        t = torch.from_dlpack(spectrograms)
        want = torch.device(device) if str(device) != "cpu" else None
        if want is not None and (t.device.type != want.type or (want.index is not None and want.index != t.device.index)):
            t = t.to(want)
        elif want is None and t.device.type != "cpu":
            t = t.cpu()
        return t if dtype is None else t.to(dtype=dtype)
    if not spectrograms:
        raise ValueError("Cannot batch empty list of spectrograms")
    tensors = []
    for spec in spectrograms:
        t = spec.to_torch(device="cpu", dtype=dtype) if hasattr(spec, "to_torch") else torch.from_dlpack(spec)
        if dtype is not None:
            t = t.to(dtype=dtype)
        tensors.append(t)
    if pad:
        max_frames = max(t.shape[1] for t in tensors)
        max_bins = max(t.shape[0] for t in tensors)
        tensors = [t if tuple(t.shape) == (max_bins, max_frames)
                   else torch.nn.functional.pad(t, (0, max_frames - t.shape[1], 0, max_bins - t.shape[0]), value=0) for t in tensors]
    else:
        shape = tensors[0].shape
        if not all(t.shape == shape for t in tensors):
            raise ValueError(f"All spectrograms must have the same shape. Got shapes: {[t.shape for t in tensors]}. "
                             f"Use pad=True to pad to the same size.")
    out = torch.stack(tensors)
    return out.to(device) if str(device) != "cpu" else out


def batch_with_metadata(spectrograms, device="cpu", dtype: Optional[torch.dtype] = None, pad: bool = False):
    """torch.py:288-330."""
    if isinstance(spectrograms, SpectrogramBatch):
        sb = spectrograms
        meta = [{"shape": (sb.n_bins, sb.n_frames), "frequencies": np.array(sb.frequencies), "times": np.array(sb.times),
                 "params": sb.params} for _ in range(len(sb))]
        return batch(sb, device=device, dtype=dtype), meta
    metadata = []
    for spec in spectrograms:
        meta = {"shape": getattr(spec, "shape", None),
                "frequencies": np.array(spec.frequencies) if hasattr(spec, "frequencies") else None,
                "times": np.array(spec.times) if hasattr(spec, "times") else None,
                "params": getattr(spec, "params", None)}
        if hasattr(spec, "db_range") and callable(spec.db_range):
            meta["db_range"] = spec.db_range()
        metadata.append(meta)
    return batch(spectrograms, device=device, dtype=dtype, pad=pad), metadata


def batch_signals(plan: Plan, signals, device=None) -> SpectrogramBatch:
    """The whole `[spec(s) for s in signals]` + `batch(...)` pipeline as ONE launch: signals is [B, N] (numpy or a torch
    device tensor); the result stays on the GPU."""
    if device is not None and not type(signals).__module__.startswith("torch"):
        with torch.cuda.device(torch.device(device)):
            return plan.compute_batch_resident(signals)
    return plan.compute_batch_resident(signals)
