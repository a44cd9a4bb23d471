"""One-rank RCCL sanity check for a GPU box (init_process_group("nccl"), all_reduce, all_gather_into_tensor, barrier) — the
collectives bench.py uses for N > 1; run as `python tools/check_rccl_world1.py`."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dev=torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t=torch.ones(4,device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
ids=torch.zeros(1,dtype=torch.int64,device=dev); dist.all_gather_into_tensor(ids, torch.tensor([7],dtype=torch.int64,device=dev))
dist.barrier(); torch.cuda.synchronize()
print("rccl world=1 ok", t.tolist(), ids.tolist(), dist.get_world_size())
dist.destroy_process_group()
