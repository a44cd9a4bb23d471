"""One process per GPU: utterances shard across ranks (contiguous blocks, remainder to the low ranks — SURVEY.md
§8e); every rank runs the single-GPU plan on its shard; the only collective is the optional all-gather that
reassembles the batched output on every rank (RCCL over xGMI with backend "nccl"; gloo on CPU tensors in tests).

The path has no exchange step inside the transform, so there is no data-path collective: `gather_outputs` exists for
callers that need the whole [B, n_bins, n_frames] tensor on every rank (BASELINE.json configs[3])."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

from . import _ffi


def shard_range(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(start, count) of rank's utterances; sgx_shard_range in include/spectro_hip.h."""
    s, c = C.c_size_t(), C.c_size_t()
    _ffi.raise_status(_ffi.lib().sgx_shard_range(batch, world_size, rank, C.byref(s), C.byref(c)))
    return s.value, c.value


def gather_outputs(local, batch_total: int, group=None):
    """All-gather per-rank output shards [count_r, ...] into [batch_total, ...] on every rank.

    Shards are the contiguous blocks of `shard_range`; when batch_total does not divide evenly the shorter shards are
    padded to the longest for one `all_gather_into_tensor` and the padding rows are dropped afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return local
    counts = [shard_range(batch_total, world, r)[1] for r in range(world)]
    cmax = max(counts)
    if local.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError(f"local shard has {local.shape[0]} rows, expected {counts[dist.get_rank(group)]}")
    tail = tuple(local.shape[1:])
    send = local
    if local.shape[0] != cmax:
        send = torch.zeros((cmax,) + tail, dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    buf = torch.empty((world * cmax,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, send.contiguous(), group=group)
    if all(c == cmax for c in counts):
        return buf
    parts = [buf[r * cmax: r * cmax + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


class OverlappedGather:
    """All-gather of equal-sized output shards that overlaps the next launch.

    `depth` result buffers rotate: `submit(i, shard)` starts an asynchronous `all_gather_into_tensor` of step i's shard (on
    RCCL's own stream — it waits for the work already queued on the caller's stream, i.e. for the kernel that produced the
    shard) and returns at once, so step i + 1's kernel runs while step i's shards travel over xGMI.  Before a shard buffer
    or a gathered buffer is reused, `submit` waits for the collective that last used that slot.  `finish()` waits for all."""

    def __init__(self, shard_shape, dtype, device, depth: int = 2, group=None):
        import torch
        import torch.distributed as dist

        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.depth = depth
        shard_shape = tuple(shard_shape)
        # concatenated along dim 0 ([world * n0, ...]): the form both RCCL and gloo accept; `gathered[i].view(world, n0, ...)`
        self._flat = [torch.empty((self.world * shard_shape[0],) + shard_shape[1:], dtype=dtype, device=device) for _ in range(depth)]
        self.gathered = [t.view((self.world,) + shard_shape) for t in self._flat]
        self._work = [None] * depth

    def wait_slot(self, i: int) -> None:
        w = self._work[i % self.depth]
        if w is not None:
            w.wait()  # makes the caller's stream (CPU thread for gloo) wait for that collective
            self._work[i % self.depth] = None

    def submit(self, i: int, shard):
        """Start gathering step i's shard; returns the [world, ...] tensor that will hold it once `wait_slot(i)` returns."""
        import torch.distributed as dist

        self.wait_slot(i)
        dst = self.gathered[i % self.depth]
        if self.world == 1:
            dst[0].copy_(shard)
        else:
            self._work[i % self.depth] = dist.all_gather_into_tensor(self._flat[i % self.depth], shard.contiguous(), group=self.group,
                                                                     async_op=True)
        return dst

    def finish(self) -> None:
        for i in range(self.depth):
            self.wait_slot(i)


class ShardedPlan:
    """Wraps a single-GPU `Plan`: `compute(x_full_or_local)` on this rank's utterances (+ optional gather)."""

    def __init__(self, plan, group=None):
        import torch.distributed as dist

        self.plan, self.group = plan, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def local_slice(self, batch_total: int) -> slice:
        s, c = shard_range(batch_total, self.world, self.rank)
        return slice(s, s + c)

    def compute_local(self, x_local, out=None):
        return self.plan.compute_batch(x_local, out=out)

    def compute(self, x_local, batch_total: int, gather: bool = False, out=None):
        y = self.compute_local(x_local, out=out)
        if gather and self.world > 1:
            import torch
            y = gather_outputs(torch.view_as_real(y) if y.is_complex() else y, batch_total, self.group)
        return y


class ShardComm:
    """The C ABI's own RCCL communicator (`sgx_comm_*`, include/spectro_hip.h) for this rank: rank 0 makes the 128-byte id
    (ncclGetUniqueId), `torch.distributed` carries it to the other ranks (any backend — it is 128 bytes of host data), every rank
    runs ncclCommInitRank on its own device.  `execute(plan, x_local, batch_total, gathered, chunks)` is `sgx_shard_execute` /
    `sgx_shard_execute_chunked`: the rank's shard computed straight into its slice of `gathered` and the shards exchanged inside the
    same call, with `chunks > 1` on the communicator's second stream while the next chunk computes."""

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist

        self._lib = _ffi.lib()
        self._h = C.c_void_p()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            st = self._lib.sgx_comm_unique_id(ident)
            if st != 0:
                raise _ffi.FFTBackendError((self._lib.sgx_comm_last_error(None) or b"").decode() or "sgx_comm_unique_id failed")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0, group=group)
        dev = torch.device(device)
        st = self._lib.sgx_comm_create(box[0], self.world, self.rank, dev.index if dev.index is not None else -1, C.byref(self._h))
        if st != 0:
            raise _ffi.FFTBackendError((self._lib.sgx_comm_last_error(None) or b"").decode() or "sgx_comm_create failed")

    def execute(self, plan, x_local, batch_total: int, gathered, chunks: int = 1, shard_out=None, stream: int = 0,
                n_samples: Optional[int] = None):
        """x_local: this rank's [count, N] device tensor (None, or zero rows, on a rank whose shard is empty — then `n_samples` says
        N); gathered: contiguous [batch_total, n_bins, n_frames(, 2)] device tensor of the plan's dtype.  Everything the C side
        writes through raw pointers is checked here first: row count against `sgx_shard_range`, dtype, device, contiguity."""
        import torch

        tdt = torch.float32 if plan._dt == _ffi.F32 else torch.float64
        _, count = shard_range(batch_total, self.world, self.rank)
        rows = 0 if x_local is None else int(x_local.shape[0])
        if rows != count:
            raise ValueError(f"rank {self.rank} of {self.world}: local shard has {rows} rows, sgx_shard_range gives {count} of {batch_total}")
        if rows:
            if x_local.dim() != 2 or not x_local.is_cuda or x_local.dtype != tdt or x_local.stride(1) != 1:
                raise ValueError("x_local must be a 2-D CUDA tensor of the plan's dtype with unit inner stride")
            if n_samples is not None and n_samples != x_local.shape[1]:
                raise ValueError(f"n_samples {n_samples} != x_local.shape[1] {x_local.shape[1]}")
            n, stride, xp = int(x_local.shape[1]), int(x_local.stride(0)), x_local.data_ptr()
            dev = x_local.device
        else:
            if n_samples is None and (x_local is None or x_local.dim() != 2):
                raise ValueError("a rank with an empty shard passes n_samples (or a [0, N] tensor)")
            n = int(n_samples if n_samples is not None else x_local.shape[1])
            stride, xp, dev = n, None, gathered.device
        for name, t in (("gathered", gathered), ("shard_out", shard_out)):
            if t is None:
                continue
            if not t.is_cuda or t.device != dev or t.dtype != tdt or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous CUDA tensor of the plan's dtype on the samples' device")
        nb, nf = plan.output_shape(n)
        per = nb * nf * (2 if plan.is_complex else 1)
        if gathered.numel() != batch_total * per:
            raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {batch_total * per}, got {gathered.numel()}", batch_total * per, gathered.numel())
        if shard_out is not None and shard_out.numel() != count * per:
            raise _ffi.DimensionMismatchError(f"Dimension mismatch: expected {count * per}, got {shard_out.numel()}", count * per, shard_out.numel())
        s = stream or torch.cuda.current_stream(gathered.device).cuda_stream
        st = self._lib.sgx_shard_execute_chunked(plan._h, self._h, xp, batch_total, n, stride,
                                                 shard_out.data_ptr() if shard_out is not None else None,
                                                 gathered.data_ptr(), int(chunks), C.c_void_p(s))
        if st != 0:
            msg = (self._lib.sgx_comm_last_error(self._h) or b"").decode() or f"sgx_shard_execute_chunked failed (status {st})"
            raise {_ffi.SGX_INVALID_INPUT: _ffi.InvalidInputError, _ffi.SGX_DIM_MISMATCH: _ffi.DimensionMismatchError,
                   _ffi.SGX_BACKEND: _ffi.FFTBackendError}.get(st, _ffi.InternalError)(msg)
        return gathered

    def close(self) -> None:
        if self._h:
            self._lib.sgx_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
