"""bench.py's own multi-rank launcher (no GPU): `python bench.py --gpus 2 --dry-run` must start two fresh ranks, rendezvous over
gloo, time K steps between barriers, take the max over ranks and print ONE JSON line from rank 0; a WORLD_SIZE that disagrees
with --gpus is an error."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, env=e, timeout=300)


def test_direct_multi_rank_run_spawns_its_ranks():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--workload", "config4", "--gather", "overlap", "--dry-run", "--experimental-legs", "cabi"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]  # gloo may print a connection notice of its own
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["config"]["batch_per_gpu"] == 1024
    assert d["config"]["gather"] == "overlap" and d["ms_per_step"] > 0
    # what the multi-rank line must carry so that the driver's scaling records can be audited: how many ranks the process group
    # really has, and the spread of the per-rank launch times
    assert d["rccl_ranks"] == 2 and 0 < d["kernel_ms_min"] <= d["kernel_ms_max"]
    # N > 1 carries BASELINE configs[3] with and without the exchange (bench.multi_rank_legs, here on a stand-in plan over gloo): compute
    # only, all-gather behind every launch, gather overlapped with the next launch, the all-gather alone with its per-rank rate, and
    # the C ABI's chunked path — which has no device here and must report its error without taking the line down
    w = d["workloads"]
    assert set(w) == {"config4", "config4_gather_sync", "config4_gather_alone", "config4_gather_overlap", "config4_strong", "config4_cabi_chunked4"}
    # the strong-scaling leg: ONE fixed job (here 11 utterances) cut with sgx_shard_range — 6 + 5 rows over the two ranks
    assert w["config4_strong"]["scaling"] == "strong" and w["config4_strong"]["utterances_total"] == 11 and w["config4_strong"]["utterances_this_rank"] == 6
    assert w["config4_strong"]["value"] > 0
    for k in ("config4", "config4_gather_sync", "config4_gather_overlap"):
        assert w[k]["value"] > 0 and w[k]["ms_per_step"] > 0 and w[k]["steps"] == 3, w[k]
    for k in ("config4_gather_sync", "config4_gather_overlap"):
        assert w[k]["GBps_per_rank"] > 0 and w[k]["shard_MB"] == 4 * 8 * 5 * 4 / 1e6
    assert w["config4_gather_alone"]["GBps_per_rank"] > 0 and w["config4_gather_alone"]["GBps_per_link"] > 0
    assert "no HIP device" in w["config4_cabi_chunked4"]["error"]
    # the C ABI's chunked leg is opt-in (experimental until it has run on more than one real GPU): the default line leaves it out
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr
    w = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["workloads"]
    assert "config4_cabi_chunked4" not in w and "config4_strong" in w


def test_two_ranks_on_one_device_are_refused():
    r = _run(["--gpus", "2", "--steps", "2", "--dry-run"], env={"SGX_DRY_DEVICE": "0"})  # both ranks claim device 0
    assert r.returncode == 3 and "device ordinal repeats" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_rank_dry_run_and_world_size_mismatch():
    r = _run(["--gpus", "1", "--steps", "3", "--dry-run"])
    assert r.returncode == 0 and json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1
    r = _run(["--gpus", "4", "--dry-run"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                 "MASTER_PORT": "29999"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
