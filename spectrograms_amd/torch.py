"""Device-resident interop of the batched engine with PyTorch.

The reference's `python/spectrograms/torch.py` builds a batch on the host — one DLPack conversion per spectrogram, `torch.stack`,
then `.to(device)` — and, per BASELINE's north_star, stays as it is on the Rust/PyO3 side.  This module is only what the HIP
engine adds next to it: a `[B, n_bins, n_frames]` result that the kernel wrote into device memory is ALREADY the stacked batch,
so handing it to torch (or any DLPack consumer, device type kDLROCM) is a view, not a copy.

    sb = batch_signals(plan, signals)      # one launch for the whole batch, nothing returns to the host
    t  = batch(sb)                         # torch view of the same memory
    t, axes = batch_with_axes(sb)          # + the frequency / time axes all items share
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np
import torch

from .planner import Plan, SpectrogramBatch


class BatchAxes(NamedTuple):
    """Axes and parameters common to every item of a `SpectrogramBatch` (one plan produced them all)."""
    frequencies: np.ndarray
    times: np.ndarray
    params: object
    item_shape: tuple


def _same_place(t: torch.Tensor, device: Optional[torch.device]) -> bool:
    return device is None or (t.device.type == device.type and (device.index is None or device.index == t.device.index))


def batch(sb: SpectrogramBatch, device=None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """The batch as a torch tensor.  With `device` / `dtype` left alone (or equal to where and what the data already is) the
    result shares the kernel's output memory; anything else is an explicit torch conversion of that view."""
    if not isinstance(sb, SpectrogramBatch):
        raise TypeError("batch() takes the device-resident SpectrogramBatch of Plan.compute_batch_resident / batch_signals; "
                        "lists of host spectrograms are what the reference's own torch module stacks")
    t = torch.from_dlpack(sb)
    want = None if device is None else torch.device(device)
    if not _same_place(t, want):
        t = t.to(want)
    return t if dtype is None or dtype == t.dtype else t.to(dtype)


def batch_with_axes(sb: SpectrogramBatch, device=None, dtype: Optional[torch.dtype] = None):
    """`batch(sb, ...)` plus the axes its items share."""
    return batch(sb, device, dtype), BatchAxes(np.asarray(sb.frequencies), np.asarray(sb.times), sb.params, (sb.n_bins, sb.n_frames))


def batch_signals(plan: Plan, signals, device=None) -> SpectrogramBatch:
    """The whole `[plan.compute(s) for s in signals]` + stack pipeline as ONE launch: `signals` is [B, N] (a torch tensor on
    the plan's device, or a numpy array that is uploaded once); the result stays on the GPU."""
    if device is not None and not isinstance(signals, torch.Tensor):
        with torch.cuda.device(torch.device(device)):
            return plan.compute_batch_resident(signals)
    return plan.compute_batch_resident(signals)
