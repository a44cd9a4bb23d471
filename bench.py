#!/usr/bin/env python3
"""bench.py — STFT frames/s (f32, n_fft=1024, hop=256) on N MI355X, with roofline and CPU-baseline objects.

A "step" is one pass of the hot path (one batched kernel launch through the C ABI) over one batch of synthetic
signals that is already resident in HBM.  Workloads (BASELINE.json configs):
    linear_power   configs[1]: 256 x 10 s 16 kHz f32 per GPU, linear-power STFT, Hanning, centre   (default: BASELINE's metric)
    mel_db         configs[2]: the same batch, Mel-80 power + log-dB
    mel_power      the same batch, Mel-80 power (north_star's target sentence)
    stft           the same batch, complex STFT
    config4        configs[3]: 1024 utterances per GPU (8192 over 8 GPUs), Mel-80 power; `--gather` adds the RCCL all-gather

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  Utterances shard by rank with no data-path
collective (weak scaling: every rank owns a full batch); `--gather sync|overlap` adds the all-gather that reassembles the
batched output on every rank inside the timed region.  `python bench.py --gpus N` run directly (no RANK/WORLD_SIZE in the
environment) starts the N ranks itself — fresh child processes, created before this process touches the GPU — and relays rank
0's JSON line; under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak is ~6290 GB/s
VALU_PEAK_TFLOPS = 157.3   # FP32 vector peak (same guide)
FLOPS_PER_FRAME = {"linear_power": 28.1e3, "stft": 26.6e3, "mel_power": 30.1e3, "mel_db": 30.1e3}  # FFT 25.6 k + window 1 k (+ |.|^2 1.5 k, Mel 2 k)
SR, N_FFT, HOP, N_SAMPLES = 16000.0, 1024, 256, 160000
WORKLOADS = {  # name -> (kernel workload, utterances per GPU, BASELINE config index)
    "linear_power": ("linear_power", 256, 1), "mel_db": ("mel_db", 256, 2), "mel_power": ("mel_power", 256, 2),
    "stft": ("stft", 256, 1), "config4": ("mel_power", 1024, 3),
}


def cfg_signal(b: int) -> np.ndarray:
    """BASELINE.md §2 generator: even rows sine 0.5*sin(2*pi*f_b*i/16000), odd rows N(0, 0.1^2), seed 1234+b."""
    if b % 2 == 0:
        f = 110.0 * 2.0 ** ((b % 48) / 12.0)
        return (0.5 * np.sin(2.0 * np.pi * f * np.arange(N_SAMPLES, dtype=np.float64) / SR)).astype(np.float32)
    return (0.1 * np.random.default_rng(1234 + b).standard_normal(N_SAMPLES)).astype(np.float32)


def bytes_per_frame(kernel_wl: str, n_frames: int):
    """Algorithmic HBM bytes per frame: every input sample read once, every output written once."""
    read = N_SAMPLES * 4.0 / n_frames
    write = {"linear_power": 513 * 4.0, "mel_db": 80 * 4.0, "mel_power": 80 * 4.0, "stft": 513 * 8.0}[kernel_wl]
    return read, write


def kernel_source_stamp() -> str:
    """Identity of the kernel sources a measurement belongs to (profiles/traffic_latest.json carries the stamp it was taken at)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spectrograms_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith("kernels_r32x16") or f in ("fft_inreg.h", "r32x16_layout.h"):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(kernel_wl: str):
    """PMC-measured HBM bytes per launch, or None when the committed measurement predates the current kernel sources."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    except Exception:
        return None
    if t.get("kernel_source_stamp") != kernel_source_stamp():
        return None
    return t.get(kernel_wl)


def cpu_baseline(kernel_wl: str, budget_s: float = 12.0):
    """Times the CPU restatement of the reference algorithm (oracle/, kind 'port': per-frame window -> real FFT ->
    |.|^2 -> [sparse Mel -> dB], one plan per thread over utterances — the reference's batch idiom, src/lib.rs:228-236)
    on this host's cores, into a preallocated output (the reference allocates per call; that is not charged here)."""
    from oracle import oracle as orc

    cores = orc.max_threads()
    nsig = 256
    x = np.stack([cfg_signal(b) for b in range(nsig)])
    if kernel_wl in ("linear_power", "stft"):
        op = orc.Params(n_fft=N_FFT, hop=HOP)
    elif kernel_wl == "mel_power":
        op = orc.Params(n_fft=N_FFT, hop=HOP, n_mels=80)
    else:
        op = orc.Params(n_fft=N_FFT, hop=HOP, n_mels=80, amp="db", floor_db=-80.0)
    out = orc.spectrogram_batch(op, x, nthreads=cores)  # warm-up, allocates the output once
    frames_per_pass = out.shape[0] * out.shape[2]
    # the box may expose more hardware threads than this process can really use (cgroup quota, SMT): keep the thread count
    # that is fastest on one pass, so the baseline is the best this host does, not an oversubscribed one
    cands = set()
    nt = cores
    while nt >= 4:
        cands.add(nt)
        nt //= 2
    try:  # cgroup v2 CPU quota: "max" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cands.add(max(1, min(cores, -(-int(q) // int(per)))))
    except Exception:
        pass
    best = (0.0, cores)
    for nt in sorted(cands, reverse=True):
        t0 = time.perf_counter()
        n_pass = 0
        while time.perf_counter() - t0 < 0.6:  # several scheduler periods: a CFS quota throttles in 100 ms slices
            orc.spectrogram_batch(op, x, nthreads=nt, out=out)
            n_pass += 1
        rate = n_pass * frames_per_pass / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, nt)
    cores = best[1]
    frames, reps = 0, 0
    t0 = time.perf_counter()
    while True:
        orc.spectrogram_batch(op, x, nthreads=cores, out=out)
        frames += frames_per_pass
        reps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    o1 = np.empty((8,) + out.shape[1:], out.dtype)
    orc.spectrogram_batch(op, x[:8], nthreads=1, out=o1)
    t1 = time.perf_counter()
    orc.spectrogram_batch(op, x[:8], nthreads=1, out=o1)
    dt1 = time.perf_counter() - t1
    single = 8 * out.shape[2] / dt1
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes over a {nsig}-utterance batch ({frames} frames, {dt:.1f} s wall, one plan per "
                      f"thread, {cores} threads = the fastest of the thread counts tried); one thread alone: {single:.0f} frames/s",
            "single_thread_value": single}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preheat-s", type=float, default=0.3,
                    help="seconds of untimed steps before the warmup steps, so that the timed steps run at steady clocks (0: none)")
    ap.add_argument("--workload", default="linear_power", choices=sorted(WORKLOADS))
    ap.add_argument("--gather", nargs="?", const="sync", default=None, choices=["sync", "overlap"],
                    help="RCCL all-gather of the output shards inside the timed region: 'sync' (default when given) gathers "
                         "after every launch on the launch stream; 'overlap' gathers step i asynchronously while step i+1 computes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks rendezvous over gloo and time an empty step — exercises the launcher, the barrier / "
                         "max-over-ranks timing and the JSON line (tests/test_bench_launcher.py)")
    return ap.parse_args(argv)


def spawn_ranks(args) -> int:
    """`--gpus N` without a launcher: start the N ranks as fresh child processes (this process has not touched the GPU and
    never does), relay rank 0's stdout (the JSON line), return the worst exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    return rc


def main() -> int:
    args = parse_args()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU", file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist

    kernel_wl, batch, cfg_idx = WORKLOADS[args.workload]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the product path has no CPU fallback", file=sys.stderr)
        return 2
    import spectrograms_amd as sg

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- plan + synthetic device-resident batch (weak scaling: every rank owns `batch` utterances)
    params = sg.SpectrogramParams(sg.StftParams(N_FFT, HOP, sg.WindowType.hanning, True), SR)
    planner = sg.SpectrogramPlanner()
    if kernel_wl == "linear_power":
        plan = planner.linear_power_plan(params, dtype="float32")
    elif kernel_wl == "mel_power":
        plan = planner.mel_power_plan(params, sg.MelParams(80, 0.0, 8000.0), dtype="float32")
    elif kernel_wl == "stft":
        plan = planner.stft_plan(params, dtype="float32")
    else:
        plan = planner.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32")
    n_bins, n_frames = plan.output_shape(N_SAMPLES)
    # the generator has 96 distinct rows (48 pitches + 48 noise seeds would repeat the sines); build 256 and tile for config 4
    base = np.stack([cfg_signal(rank * batch + b) for b in range(min(batch, 256))])
    host = base if batch <= 256 else np.concatenate([base] * (batch // 256))
    nsets = 2  # rotate buffer sets so a step never re-reads its own input out of the 256 MiB Infinity Cache
    xs = [torch.from_numpy(host).to(dev) for _ in range(nsets)]
    oshape = (batch, n_bins, n_frames, 2) if kernel_wl == "stft" else (batch, n_bins, n_frames)
    outs = [torch.empty(oshape, dtype=torch.float32, device=dev) for _ in range(nsets)]
    gathered = None
    overlap = None
    if args.gather and world > 1:
        if args.gather == "overlap":
            from spectrograms_amd.distributed import OverlappedGather
            overlap = OverlappedGather(oshape, torch.float32, dev, depth=nsets)
        else:
            gathered = torch.empty((world * oshape[0],) + tuple(oshape[1:]), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step(i: int) -> None:
        if overlap is not None:
            overlap.wait_slot(i)  # the gather that last read outs[i % nsets] is done before the kernel overwrites it
        plan.compute_batch(xs[i % nsets], out=outs[i % nsets])
        if overlap is not None:
            overlap.submit(i, outs[i % nsets])  # travels while step i + 1 computes
        elif gathered is not None:
            dist.all_gather_into_tensor(gathered, outs[i % nsets])

    def fence() -> None:
        if overlap is not None:
            overlap.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Clock settling, before the W warmup steps: after the idle seconds of start-up (imports, plan creation, uploads) the GPU's power
    # management needs about 300 launches (~35 ms) of this workload to reach its steady clocks — measured per block of 25 launches from
    # idle: 132 152 137 131 126 124 121 121 121 118 116 ... 115 us — so without it the K timed steps sit on the ramp.  Same steps, same
    # buffers, every rank the same count (from the slowest rank's step time); reported as `preheat_steps`.
    preheat_steps = 0
    if args.preheat_s > 0:
        t_probe = time.perf_counter()
        for i in range(10):
            step(i)
        fence()
        tp = torch.tensor([(time.perf_counter() - t_probe) / 10], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        preheat_steps = int(min(20000, max(0.0, args.preheat_s / max(float(tp.item()), 1e-6))))
        for i in range(preheat_steps):
            step(i)
        fence()
        preheat_steps += 10
    for i in range(args.warmup):
        step(i)
    fence()
    # with a gather in the step, the kernel alone is timed here (the C ABI's own hipEvent pair on the launch stream, same
    # buffers); without one, the timed region below is back-to-back launches and its own events give the launch duration
    kernel_ms = plan.time_batch_torch(xs[0], outs[0], max(1, min(args.steps, 50)))
    gather_ms = None
    if gathered is not None:  # the gather alone, for the per-link rate
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        e0.record(stream)
        for _ in range(5):
            dist.all_gather_into_tensor(gathered, outs[0])
        e1.record(stream)
        fence()
        gather_ms = e0.elapsed_time(e1) / 5
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)  # the stream the kernels are launched on (the plan launches on torch's current stream)
    for i in range(args.steps):
        step(i)
    ev1.record(stream)
    fence()
    dt = time.perf_counter() - t0
    if gathered is None and overlap is None:
        kernel_ms = ev0.elapsed_time(ev1) / args.steps  # mean launch duration over the timed region itself

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    frames_per_launch = batch * n_frames
    frames_per_step = frames_per_launch * world
    value = frames_per_step * args.steps / dt
    rd, wr = bytes_per_frame(kernel_wl, n_frames)
    kernel_fps = frames_per_launch / (kernel_ms * 1e-3)
    achieved = (rd + wr) * kernel_fps / 1e9
    valu_frac = FLOPS_PER_FRAME[kernel_wl] * kernel_fps / (VALU_PEAK_TFLOPS * 1e12)
    if rank == 0:
        line = {
            "metric": "STFT frames/sec (f32, n_fft=1024 hop=256)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "preheat_steps": preheat_steps,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[{cfg_idx}]: {batch} x 10 s 16 kHz f32 per GPU, {kernel_wl} n_fft=1024 hop=256 Hanning centre"
                                   + (" (8192 utterances over 8 GPUs)" if args.workload == "config4" else ""),
                       "batch_per_gpu": batch, "n_samples": N_SAMPLES, "frames_per_step": frames_per_step, "kernel": plan.kernel_name,
                       "gather": (args.gather if (gathered is not None or overlap is not None) else False),
                       "parallelism": f"utterance-shard x{world}"},
            # the dominant kernel, measured live (hipEvents around back-to-back launches on the launch stream, no gather): `achieved`
            # is algorithmic bytes / kernel time.  The kernel is bounded by its arithmetic + LDS work next to the HBM stream, not by
            # HBM alone: `valu_frac` is its share of the FP32 vector peak, `hbm_read_frac` north_star's read-only line.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(kernel_wl) if batch == 256 else None, "kernel_ms": kernel_ms,
                         "kernel_ms_scope": "HIP events over the timed region" if (gathered is None and overlap is None) else "compute only (separate back-to-back launches, no gather)", "algorithmic_bytes_per_frame": rd + wr,
                         "frames_per_launch": frames_per_launch, "hbm_read_frac": rd * kernel_fps / 1e9 / HBM_PEAK_GBS,
                         "valu_frac": valu_frac, "flops_per_frame": FLOPS_PER_FRAME[kernel_wl],
                         "limiter": "valu+lds" if valu_frac > achieved / HBM_PEAK_GBS else "hbm"},
        }
        if gathered is not None or overlap is not None:
            shard_bytes = float(np.prod(oshape)) * 4.0
            line["gather"] = {"mode": args.gather, "compute_only_value": kernel_fps * world, "shard_MB": shard_bytes / 1e6,
                              "gather_ms": gather_ms,
                              # ring all-gather: every rank sends and receives (world - 1) shards over its links
                              "GBps_per_rank": None if not gather_ms else shard_bytes * (world - 1) / (gather_ms * 1e-3) / 1e9}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kernel_wl)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def dry_run(args, rank: int, world: int) -> int:
    """The launcher / timing skeleton without a GPU: gloo rendezvous, barrier, K empty steps, max over ranks, one JSON line."""
    import torch
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-4)
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    _, batch, cfg_idx = WORKLOADS[args.workload]
    if rank == 0:
        print(json.dumps({"metric": "STFT frames/sec (f32, n_fft=1024 hop=256)", "value": 0.0, "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(dt.item()) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "dry-run (no GPU work)",
                          "config": {"workload": f"configs[{cfg_idx}] dry run", "batch_per_gpu": batch, "gather": args.gather or False,
                                     "parallelism": f"utterance-shard x{world}"}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
