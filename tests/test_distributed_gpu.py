"""N > 1 code paths against the REAL RCCL on the one GPU of the test box (world size 1: RCCL refuses two ranks on one device): the C
ABI's communicator (sgx_comm_create = ncclCommInitRank from the RCCL copy already in the process, sgx_shard_execute_chunked with its
second stream and events) and bench.py's multi-rank legs.  Each runs in a child process with its own process group."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    e.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return e


@pytest.mark.gpu
def test_c_abi_communicator_on_real_rccl_one_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_cabi_rccl_world1.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, env=_env(29547), timeout=300)
    assert r.returncode == 0 and "one rank: passed" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_bench_multi_rank_legs_rehearsal():
    """`bench.py --rehearse-multi-rank`: the legs the driver's `--gpus 8` run executes (config 4 compute, all-gather behind every launch,
    the all-gather alone, gather overlapped with the next launch, the C ABI's chunked path), here with a one-rank RCCL group."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--rehearse-multi-rank",
                        "--no-cpu-baseline", "--preheat-s", "0.05", "--experimental-legs", "cabi"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env(29549), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    w = d["workloads"]
    assert set(w) == {"config4", "config4_gather_sync", "config4_gather_alone", "config4_gather_overlap", "config4_strong", "config4_cabi_chunked4"}
    assert w["config4_strong"]["utterances_this_rank"] == 8192 and w["config4_strong"]["value"] > 5e8
    for k, v in w.items():
        assert "error" not in v, (k, v)
    assert w["config4"]["value"] > 5e8 and w["config4_cabi_chunked4"]["value"] > 5e8
