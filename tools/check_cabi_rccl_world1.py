"""The C ABI's communicator against the REAL RCCL on one GPU: a one-rank torch process group carries the unique id, sgx_comm_create
runs ncclCommInitRank, sgx_shard_execute / sgx_shard_execute_chunked (1, 4 and 7 chunks: compute on the caller's stream, the grouped
ncclBroadcast of each chunk on the communicator's stream, ordered by events) gather into a poisoned buffer, and the result must equal a
plain launch bit for bit.  (tests/c_abi/shard_ranks.c runs 2 and 3 ranks against a stand-in RCCL; real RCCL refuses two ranks on one
device, so one rank is what a one-GPU box can do.)  Run: python tools/check_cabi_rccl_world1.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
import numpy as np
import torch
import torch.distributed as dist

import bench
import spectrograms_amd as sg
from spectrograms_amd.distributed import ShardComm

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.from_numpy(np.stack([bench.cfg_signal(b)[:40000] for b in range(13)])).to(dev)
comm = ShardComm(dev)
for wl in ("mel_db", "linear_power", "stft"):
    plan = bench.make_plan(sg, wl)
    ref = plan.compute_batch(x)
    ref = torch.view_as_real(ref) if ref.is_complex() else ref
    for chunks in (1, 4, 7):
        g = torch.full_like(ref, float("nan"))
        comm.execute(plan, x, x.shape[0], g, chunks=chunks)
        torch.cuda.synchronize()
        assert torch.equal(g, ref), (wl, chunks)
        shard = torch.empty_like(ref)  # separate shard buffer, then gathered
        g.fill_(float("nan"))
        comm.execute(plan, x, x.shape[0], g, chunks=chunks, shard_out=shard)
        torch.cuda.synchronize()
        assert torch.equal(g, ref) and torch.equal(shard, ref), (wl, chunks, "separate")
    print(wl, "ok")
comm.close()
dist.barrier()
dist.destroy_process_group()
print("c-abi communicator on real RCCL, one rank: passed")
