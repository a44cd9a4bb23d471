#!/usr/bin/env python3
"""bench.py — STFT frames/s (f32, n_fft=1024, hop=256) on N MI355X, with roofline and CPU-baseline objects.

A "step" is one pass of the hot path (one batched kernel launch through the C ABI) over one batch of synthetic
signals that is already resident in HBM.  Workloads (BASELINE.json configs):
    linear_power   configs[1]: 256 x 10 s 16 kHz f32 per GPU, linear-power STFT, Hanning, centre   (default: BASELINE's metric)
    mel_db         configs[2]: the same batch, Mel-80 power + log-dB
    mel_power      the same batch, Mel-80 power (north_star's target sentence)
    stft           the same batch, complex STFT
    config4        configs[3]: 1024 utterances per GPU (8192 over 8 GPUs), Mel-80 power; `--gather` adds the RCCL all-gather

The default run (what the driver records) carries, next to the headline line of `--workload` (default linear_power), a
`workloads` object with short legs of the other BASELINE configurations — mel_power (north_star's target sentence), mel_db
(configs[2]), config4's per-GPU shard (configs[3]), configs[4]: fft2d / convolve_fft over 512 x 1024 x 1024 images, and chirpz_1009
(a prime frame length, 64 utterances: the reference plans every length), the inverse STFT, and linear_power_f64 / mel_db_f64
(configs[1] / configs[2] in the reference's other `Sample` type, the Python API's default dtype) — each with its own ms_per_step, value and roofline; `cold_ms_per_step` (the W + K steps from idle clocks, measured before anything
else has run); `roofline.peak_measured` (sgx_membench: copy / read / write rates of this very device, in-process); and a
NumPy / pocketfft datapoint inside `cpu_baseline`.  `--no-legs` switches the extra legs off.

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
FLOPS_PER_FRAME = {"linear_power": 28.1e3, "stft": 26.6e3, "mel_power": 30.1e3, "mel_db": 30.1e3, "mfcc": 32.2e3}  # FFT 25.6 k + window 1 k (+ |.|^2 1.5 k, Mel 2 k)
LEGS = ("mel_power", "mel_db", "mfcc", "config4", "istft", "fft2d", "convolve_fft", "chirpz_1009", "linear_power_f64", "mel_db_f64", "linear_db_f64")  # the default run's extra legs (besides the headline workload)
IMG_SIDE, IMG_BATCH = 1024, 512  # BASELINE configs[4]
SR, N_FFT, HOP, N_SAMPLES = 16000.0, 1024, 256, 160000
WORKLOADS = {  # name -> (kernel workload, utterances per GPU, BASELINE config index)
    "linear_power": ("linear_power", 256, 1), "mel_db": ("mel_db", 256, 2), "mel_power": ("mel_power", 256, 2),
    "stft": ("stft", 256, 1), "config4": ("mel_power", 1024, 3),
    # MFCC-13 (Mel-80 dB(-80) -> DCT-II + lifter 22, src/mfcc.rs:224-316) fused into the Mel-dB launch
    "mfcc": ("mfcc", 256, 2),
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
    write = {"linear_power": 513 * 4.0, "mel_db": 80 * 4.0, "mel_power": 80 * 4.0, "stft": 513 * 8.0, "mfcc": 13 * 4.0}[kernel_wl]
    return read, write


# the kernel sources a measured traffic figure belongs to, per workload family
STAMP_FILES = {
    "stft": ("kernels_r32x16.hip", "fft_inreg.h", "r32x16_layout.h"),
    "2d": ("kernels_r32x16.hip", "kernels_c2c1024.hip", "fft2d.hip", "fft_inreg.h", "r32x16_layout.h"),
    "istft": ("kernels_c2c1024.hip", "fft_inreg.h"),
    "f64": ("kernels_d32x16.hip", "d32x16_layout.h", "fft_inreg.h", "db_f64.h", "lane_pair.h"),
}
STAMP_FAMILY = {"linear_power": "stft", "mel_power": "stft", "mel_db": "stft", "mfcc": "stft", "config4": "stft", "stft": "stft", "fft2d": "2d", "convolve_fft": "2d", "istft": "istft",
                "linear_power_f64": "f64", "mel_db_f64": "f64", "linear_db_f64": "f64"}


def kernel_source_stamp(family: str = "stft") -> str:
    """Identity of the kernel sources a measurement belongs to (profiles/traffic_latest.json carries the stamp it was taken at)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spectrograms_amd", "csrc")
    for f in sorted(STAMP_FILES[family]):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(workload: str, detail: bool = False):
    """PMC-measured HBM bytes per step (tools/profile.sh -> tools/summarize_prof.py --traffic), or None when the committed
    measurement predates the current sources of the kernels it was taken on.  detail: the per-kernel table next to the total."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    except Exception:
        return None
    e = (t.get("entries") or {}).get(workload)
    if not e or e.get("stamp") != kernel_source_stamp(STAMP_FAMILY[workload]):
        return None
    return e if detail else e.get("bytes")


def usable_cores() -> int:
    """CPUs this process may really use: its affinity mask, capped by a cgroup v2 CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # "max" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(kernel_wl: str, budget_s: float = 10.0):
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
            "single_thread_value": single, "pocketfft": cpu_pocketfft(kernel_wl, x[:16]),
            "pocketfft_all_cores": cpu_pocketfft_all_cores(kernel_wl, x[:16])}


def _pocketfft_worker(job):
    kernel_wl, x, budget_s = job
    return cpu_pocketfft(kernel_wl, x, budget_s)["value"] * budget_s  # ~frames done (each worker measures its own wall)


def cpu_pocketfft_all_cores(kernel_wl: str, x: np.ndarray, budget_s: float = 3.0):
    """The optimised-FFT datapoint on every core this process may use: one worker PROCESS per core (fork, started before this
    process has touched the GPU), each running cpu_pocketfft's single-thread loop over the same utterances for the same wall time —
    the reference's one-plan-per-thread batch idiom with pocketfft's SIMD butterflies in place of the scalar port's."""
    import multiprocessing as mp

    cores = usable_cores()
    try:
        ctx = mp.get_context("fork")
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            per = pool.map(_pocketfft_worker, [(kernel_wl, x, budget_s)] * cores)
        wall = time.perf_counter() - t0
    except Exception as e:  # a host that refuses the pool: the single-thread figure stands
        return {"error": repr(e)}
    return {"value": sum(per) / budget_s, "unit": "frames/s", "cores": cores, "kind": "numpy-pocketfft",
            "sample": f"{cores} worker processes x {budget_s:.0f} s of the single-thread pocketfft loop each (sum of the workers' own rates; "
                      f"{wall:.1f} s wall with pool start-up)"}


def cpu_pocketfft(kernel_wl: str, x: np.ndarray, budget_s: float = 3.0):
    """Secondary CPU datapoint: the same pipeline written the NumPy way — zero-padded strided frames x window ->
    numpy.fft.rfft (pocketfft, SIMD, f32 in / complex64 out) over all frames of an utterance at once -> |.|^2 [-> dense Mel
    matmul -> 10 log10] — as python/examples/numpy_impls.py does it, one utterance per call, on ONE thread of this host.  An
    optimised CPU FFT next to the scalar port above; neither is RustFFT (no Rust toolchain on either box)."""
    from numpy.lib.stride_tricks import sliding_window_view

    from oracle import oracle as orc

    n = np.arange(N_FFT, dtype=np.float64)
    w = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / (N_FFT - 1))).astype(np.float32)  # symmetric Hann (S3)
    melT = None
    if kernel_wl in ("mel_power", "mel_db"):
        ptr, col, val, _ = orc.mel_filterbank(SR, N_FFT, 80, 0.0, 8000.0, None)
        m = np.zeros((80, N_FFT // 2 + 1), np.float32)
        for r in range(80):
            m[r, col[ptr[r]:ptr[r + 1]]] = val[ptr[r]:ptr[r + 1]]
        melT = np.ascontiguousarray(m.T)
    eps = np.float32(10.0 ** (-80.0 / 10.0))

    def one(sig):
        xp = np.pad(sig, N_FFT // 2)
        fr = sliding_window_view(xp, N_FFT)[::HOP] * w
        spec = np.fft.rfft(fr, axis=-1)
        if kernel_wl == "stft":
            return np.ascontiguousarray(spec.T)
        p = spec.real * spec.real + spec.imag * spec.imag
        if melT is not None:
            p = p @ melT
        if kernel_wl == "mel_db":
            p = 10.0 * np.log10(np.maximum(p, eps))
        return np.ascontiguousarray(p.T)  # (bins, frames), the reference's layout (S9)

    try:  # one thread for the Mel matmul too (OpenBLAS' thread pool costs more than it gains on a 626 x 513 x 80 product)
        import threadpoolctl
        limit = threadpoolctl.threadpool_limits(1)
    except Exception:
        limit = None
    o = one(x[0])
    frames, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for sig in x:
            one(sig)
            frames += o.shape[1]
    dt = time.perf_counter() - t0
    if limit is not None:
        limit.restore_original_limits()
    return {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "numpy-pocketfft",
            "sample": f"{frames} frames in {dt:.1f} s: {x.shape[0]} utterances per pass, one numpy.fft.rfft call over the 626 frames of an utterance"}


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
    ap.add_argument("--no-legs", action="store_true",
                    help="headline workload only: skip the `workloads` legs, the cold-clock run and the measured HBM peak (A/B timing runs)")
    ap.add_argument("--legs", default=",".join(LEGS), help="comma-separated legs of the default run (subset of %s)" % ",".join(LEGS))
    ap.add_argument("--sustained-s", type=float, default=3.0,
                    help="seconds of back-to-back headline launches for the `sustained` object (frames/s + shader clock); 0: none")
    ap.add_argument("--legs-timeout", type=float, default=240.0,
                    help="seconds the extra legs may take before the line is printed without the unfinished ones (0: no limit)")
    ap.add_argument("--experimental-legs", default="",
                    help="comma-separated opt-in legs of the N > 1 run: 'cabi' = configs[3] through the C ABI's own communicator "
                         "(sgx_shard_execute_chunked) - not yet run on more than one real GPU, so not in the default line")
    ap.add_argument("--rehearse-multi-rank", action="store_true",
                    help="with --gpus 1: open a ONE-rank RCCL process group and run the N > 1 legs (config 4 compute / gather / C-ABI chunked) "
                         "instead of the single-GPU ones — every call the driver's 8-GPU run makes, against the real RCCL, on one GPU")
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


def make_plan(sg, kernel_wl: str):
    params = sg.SpectrogramParams(sg.StftParams(N_FFT, HOP, sg.WindowType.hanning, True), SR)
    planner = sg.SpectrogramPlanner()
    if kernel_wl == "linear_power":
        return planner.linear_power_plan(params, dtype="float32")
    if kernel_wl == "mel_power":
        return planner.mel_power_plan(params, sg.MelParams(80, 0.0, 8000.0), dtype="float32")
    if kernel_wl == "stft":
        return planner.stft_plan(params, dtype="float32")
    if kernel_wl == "mfcc":
        return planner.mfcc_plan(sg.StftParams(N_FFT, HOP, sg.WindowType.hanning, True), SR, 80, sg.MfccParams(13), dtype="float32")
    return planner.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32")


def stft_roofline(kernel_wl: str, batch: int, n_frames: int, kernel_ms: float, peak_measured=None, scope="HIP events over the timed region", full_peak=False):
    """The roofline object of one STFT-type launch: algorithmic bytes / launch time against the 8 TB/s line; `valu_frac` is its
    share of the FP32 vector peak (the Mel workloads are bound by arithmetic + LDS, not by HBM), `hbm_read_frac`
    north_star's read-only line."""
    rd, wr = bytes_per_frame(kernel_wl, n_frames)
    frames = batch * n_frames
    fps = frames / (kernel_ms * 1e-3)
    achieved = (rd + wr) * fps / 1e9
    valu = FLOPS_PER_FRAME[kernel_wl] * fps / (VALU_PEAK_TFLOPS * 1e12)
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": measured_traffic(kernel_wl) if batch == 256 else measured_traffic("config4") if (batch == 1024 and kernel_wl == "mel_power") else None, "kernel_ms": kernel_ms, "kernel_ms_scope": scope,
         "algorithmic_bytes_per_frame": rd + wr, "frames_per_launch": frames, "hbm_read_frac": rd * fps / 1e9 / HBM_PEAK_GBS,
         "valu_frac": valu, "flops_per_frame": FLOPS_PER_FRAME[kernel_wl], "limiter": "valu+lds" if valu > achieved / HBM_PEAK_GBS else "hbm"}
    if peak_measured:
        if full_peak:
            r["peak_measured"] = peak_measured
        if peak_measured.get("copy"):
            r["frac_of_measured_copy"] = achieved / peak_measured["copy"]
        if kernel_wl == "linear_power" and peak_measured.get("mix_1r2w"):
            r["frac_of_measured_mix"] = achieved / peak_measured["mix_1r2w"]  # against a plain kernel moving the same read : write mix
    return r


def timed_steps(torch, stream, step, steps: int, fence):
    """K back-to-back steps between two events on the launch stream and two fences: (wall seconds, device ms per step)."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(steps):
        step(i)
    ev1.record(stream)
    fence()
    return time.perf_counter() - t0, ev0.elapsed_time(ev1) / steps


def preheat(step, fence, seconds: float, reduce_max=None) -> int:
    """Clock settling: after idle seconds (imports, plan creation, uploads) the GPU's power management needs about 300 launches
    (~35 ms) of the STFT workload to reach its steady clocks — measured per block of 25 launches from idle: 132 152 137 131 126
    124 121 121 121 118 116 ... 115 us — so without it the K timed steps sit on the ramp.  Same steps, same buffers, every rank
    the same count (from the slowest rank's step time); the count is reported as `preheat_steps`."""
    if seconds <= 0:
        return 0
    t_probe = time.perf_counter()
    for i in range(10):
        step(i)
    fence()
    per = (time.perf_counter() - t_probe) / 10
    if reduce_max is not None:
        per = reduce_max(per)
    n = int(min(20000, max(0.0, seconds / max(per, 1e-6))))
    for i in range(n):
        step(i)
    fence()
    return n + 10


def measured_peak(lib, device: int):
    """HBM rates of this device from the library's own streaming kernels (sgx_membench: 1 GiB buffers, 16 B per lane, 5 passes)."""
    import ctypes as C

    out = {}
    for mode, name in ((0, "copy"), (1, "read"), (2, "write"), (3, "mix_1r2w")):
        g = C.c_double()
        st = lib.sgx_membench(device, 0, mode, 5, C.byref(g))
        out[name] = float(g.value) if st == 0 else None
    out["unit"] = "GB/s"
    out["how"] = "sgx_membench in this process: 1 GiB per buffer (4x the Infinity Cache), 16 B per lane, mean of 5 passes, best of four grids; copy counts bytes read + written, mix_1r2w reads one buffer and writes two (the linear-power STFT's read : write ratio)"
    return out


def stft_leg(torch, sg, dev, name: str, xs256, args, peak):
    """One extra leg of the default run: BASELINE configs[2] (mel_db), north_star's Mel-power sentence, or configs[3]'s per-GPU
    shard (1024 utterances, Mel-80 power; input = the 256-utterance batch four times)."""
    kernel_wl, batch, cfg_idx = WORKLOADS[name]
    plan = make_plan(sg, kernel_wl)
    n_bins, n_frames = plan.output_shape(N_SAMPLES)
    xs = xs256 if batch == 256 else [torch.cat([x] * (batch // 256)) for x in xs256]
    outs = [torch.empty((batch, n_bins, n_frames), dtype=torch.float32, device=dev) for _ in xs]
    stream = torch.cuda.current_stream(dev)

    def step(i):
        plan.compute_batch(xs[i % len(xs)], out=outs[i % len(xs)])

    def fence():
        torch.cuda.synchronize(dev)

    ph = preheat(step, fence, args.preheat_s)
    for i in range(args.warmup):
        step(i)
    fence()
    dt, kernel_ms = timed_steps(torch, stream, step, args.steps, fence)
    frames = batch * n_frames
    return {"config": f"configs[{cfg_idx}]: {batch} x 10 s 16 kHz f32, {kernel_wl} n_fft=1024 hop=256 Hanning centre",
            "kernel": plan.kernel_name, "steps": args.steps, "warmup": args.warmup, "preheat_steps": ph,
            "ms_per_step": dt / args.steps * 1e3, "value": frames * args.steps / dt, "unit": "frames/s",
            "roofline": stft_roofline(kernel_wl, batch, n_frames, kernel_ms, peak)}


def f64_leg(torch, sg, dev, name: str, xs256, args, peak):
    """BASELINE configs[1] / configs[2] in the reference's other `Sample` type (f64: the Python API's default dtype, src/sample.rs:23-86):
    256 x 10 s, n_fft 1024 / hop 256 on the tuned f64 kernel k_d32x16.  Algorithmic bytes: 8-byte samples in, 8-byte outputs out."""
    kernel_wl = {"linear_power_f64": "linear_power", "mel_db_f64": "mel_db", "linear_db_f64": "linear_db"}[name]
    params = sg.SpectrogramParams(sg.StftParams(N_FFT, HOP, sg.WindowType.hanning, True), SR)
    planner = sg.SpectrogramPlanner()
    plan = (planner.linear_power_plan(params, dtype="float64") if kernel_wl == "linear_power" else
            planner.linear_db_plan(params, sg.LogParams(-80.0), dtype="float64") if kernel_wl == "linear_db" else  # 82 M dB values per launch: db_f64.h
            planner.mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float64"))
    n_bins, n_frames = plan.output_shape(N_SAMPLES)
    xs = [x.double() for x in xs256]
    outs = [torch.empty((256, n_bins, n_frames), dtype=torch.float64, device=dev) for _ in xs]
    stream = torch.cuda.current_stream(dev)

    def step(i):
        plan.compute_batch(xs[i % len(xs)], out=outs[i % len(xs)])

    def fence():
        torch.cuda.synchronize(dev)

    ph = preheat(step, fence, args.preheat_s)
    for i in range(args.warmup):
        step(i)
    fence()
    dt, kernel_ms = timed_steps(torch, stream, step, args.steps, fence)
    frames = 256 * n_frames
    rd, wr = N_SAMPLES * 8.0 / n_frames, n_bins * 8.0
    fps = frames / (kernel_ms * 1e-3)
    roof = {"bound": "hbm", "achieved": (rd + wr) * fps / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (rd + wr) * fps / 1e9 / HBM_PEAK_GBS,
            "traffic": measured_traffic(name), "kernel_ms": kernel_ms, "kernel_ms_scope": "HIP events over the timed region", "algorithmic_bytes_per_frame": rd + wr,
            "frames_per_launch": frames}
    if peak and peak.get("copy"):
        roof["frac_of_measured_copy"] = roof["achieved"] / peak["copy"]
    return {"config": f"configs[{2 if kernel_wl == 'mel_db' else 1}] in f64: 256 x 10 s 16 kHz, {kernel_wl} n_fft=1024 hop=256 Hanning centre",
            "dtype": "f64", "kernel": plan.kernel_name, "steps": args.steps, "warmup": args.warmup, "preheat_steps": ph,
            "ms_per_step": dt / args.steps * 1e3, "value": frames * args.steps / dt, "unit": "frames/s", "roofline": roof}


def istft_leg(torch, sg, dev, xs256, args, peak):
    """Inverse STFT of the configs[1] batch (src/spectrogram.rs:4860-4946): 256 x [513, 626] complex f32 spectra ->
    256 x 160 000 f32 samples, one launch of the fused tuned kernel.  Algorithmic bytes: every spectrum value read once, every sample
    written once."""
    from spectrograms_amd import _ffi

    params = sg.SpectrogramParams(sg.StftParams(N_FFT, HOP, sg.WindowType.hanning, True), SR)
    plan = sg.Plan(params, _ffi.AMP_COMPLEX, None, None, "float32")
    S = plan.compute_batch(xs256[0]).contiguous()
    y = plan.istft_batch(S)
    stream = torch.cuda.current_stream(dev)

    def step(i):
        plan.istft_batch(S, out=y)

    def fence():
        torch.cuda.synchronize(dev)

    fence()
    err = float((y[:, 1024:159000] - xs256[0][:, 1024:159000]).abs().max())
    ph = preheat(step, fence, min(args.preheat_s, 0.2))
    for i in range(args.warmup):
        step(i)
    fence()
    dt, ms = timed_steps(torch, stream, step, args.steps, fence)
    frames = S.shape[0] * S.shape[2]
    alg = float(S.numel() * 8 + y.numel() * 4)
    achieved = alg / (ms * 1e-3) / 1e9
    tr = measured_traffic("istft", detail=True)
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": tr["bytes"] if tr else None, "kernel_ms": ms, "kernel_ms_scope": "HIP events over the timed region",
            "algorithmic_bytes_per_frame": alg / frames, "frames_per_launch": frames}
    if tr:
        roof["traffic_over_algorithmic"] = tr["bytes"] / alg
    if peak and peak.get("copy"):
        roof["frac_of_measured_copy"] = achieved / peak["copy"]
    return {"config": "inverse STFT of configs[1]: 256 x [513, 626] complex f32 -> 256 x 160000 f32, n_fft=1024 hop=256 Hanning centre",
            "kernel": "istft1024", "steps": args.steps, "warmup": args.warmup, "preheat_steps": ph, "ms_per_step": dt / args.steps * 1e3,
            "value": frames * args.steps / dt, "unit": "frames/s", "roundtrip_max_err": err, "roofline": roof}


def sustained_leg(torch, lib, dev, step, fence, frames_per_step: int, seconds: float):
    """What a caller gets who streams for seconds: `seconds` of back-to-back headline launches (same buffers, same stream as the
    timed region), in blocks of ~0.25 s with the shader clock probed behind each block (sgx_clock_probe: s_memtime against the 100 MHz
    counter on every CU, MI355X_MICROARCH.md DVFS item 6).  frames/s over the whole run, probes included."""
    import ctypes as C

    stream = torch.cuda.current_stream(dev)
    t_probe = time.perf_counter()
    for i in range(50):
        step(i)
    fence()
    per = (time.perf_counter() - t_probe) / 50
    block = max(50, int(0.25 / max(per, 1e-6)))
    clocks, n = [], 0
    mhz = C.c_double()
    fence()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for i in range(block):
            step(i)
        n += block
        if lib.sgx_clock_probe(dev.index, C.c_void_p(stream.cuda_stream), C.byref(mhz)) == 0:  # waits for the stream
            clocks.append(float(mhz.value))
        else:
            fence()
    fence()
    dt = time.perf_counter() - t0
    return {"seconds": dt, "steps": n, "ms_per_step": dt / n * 1e3, "value": frames_per_step * n / dt, "unit": "frames/s",
            "shader_clock_mhz_mean": sum(clocks) / len(clocks) if clocks else None,
            "shader_clock_mhz_min": min(clocks) if clocks else None, "shader_clock_mhz_max": max(clocks) if clocks else None,
            "clock_probes": len(clocks),
            "how": f"blocks of {block} launches, sgx_clock_probe (one wave per CU, ~20 us) behind each block"}


def multi_rank_legs(torch, dist, dev, world, plan, xs, steps, warm, reduce_max, sync, make_comm, label, cabi=False, strong_total=0):
    """BASELINE configs[3] when the job has more than one rank: every rank's shard of utterances through
    the plan — compute only; with the all-gather of the output shards behind every launch (one all_gather on the launch stream); with
    the gather of step i overlapped with the compute of step i + 1 (OverlappedGather); and through the C ABI's own communicator
    (sgx_shard_execute_chunked, 4 chunks: the exchange of chunk k under the compute of chunk k + 1).  Every leg: barrier + synchronize
    on both sides, max over ranks; `value` = frames of all ranks per second.  A leg that fails reports its error instead of a number.
    (`--dry-run` drives the same function with CPU tensors over gloo and a stand-in plan: tests/test_bench_launcher.py.)"""
    from spectrograms_amd.distributed import OverlappedGather

    batch = xs[0].shape[0]
    n_bins, n_frames = plan.output_shape(xs[0].shape[1])
    oshape = (batch, n_bins, n_frames)
    outs = [torch.empty(oshape, dtype=torch.float32, device=dev) for _ in xs]
    shard_bytes = float(np.prod(oshape)) * 4.0
    frames_all = batch * n_frames * world
    res = {}

    def fence():
        dist.barrier()
        sync()

    def agree(err):
        """Every rank learns whether ANY rank failed (max of a flag) before the next fence, so that a failure on one rank ends the leg
        on all of them instead of leaving the others in a barrier."""
        bad = reduce_max(0.0 if err is None else 1.0) > 0.0
        if bad:
            return {"error": repr(err)[:300] if err is not None else "failed on another rank"}
        return None

    def phase(n, step, finish):
        try:
            for i in range(n):
                step(i)
            if finish:
                finish()
            return None
        except Exception as e:  # the leg's error, not the whole line's
            return e

    def run(name, step, finish=None, extra=None, frames=frames_all):
        # (a rank whose phase failed still joins the fence — phase() has caught the error — and the ranks agree on the failure BEHIND
        # it, so the flag's all-reduce and its host read-back stay outside the timed region)
        err = phase(warm, step, finish)
        fence()
        bad = agree(err)
        if bad:
            res[name] = bad
            return
        t0 = time.perf_counter()
        err = phase(steps, step, finish)
        fence()
        dt = time.perf_counter() - t0
        bad = agree(err)
        if bad:
            res[name] = bad
            return
        dt = reduce_max(dt)
        ms = dt / steps * 1e3
        r = {"config": label, "steps": steps, "warmup": warm, "ms_per_step": ms, "value": frames * steps / dt, "unit": "frames/s"}
        if extra:
            r.update(extra(ms))
        res[name] = r

    run("config4", lambda i: plan.compute_batch(xs[i % len(xs)], out=outs[i % len(xs)]))
    compute_ms = res["config4"].get("ms_per_step")

    def gather_extra(ms):
        g = max(ms - (compute_ms or 0.0), 1e-9)
        return {"shard_MB": shard_bytes / 1e6, "gather_exposed_ms": ms - (compute_ms or 0.0),
                # every rank sends its shard to, and receives one from, each of the other ranks
                "GBps_per_rank": shard_bytes * (world - 1) / (g * 1e-3) / 1e9}

    gathered = torch.empty((world * batch, n_bins, n_frames), dtype=torch.float32, device=dev)

    def step_sync(i):
        plan.compute_batch(xs[i % len(xs)], out=outs[i % len(xs)])
        dist.all_gather_into_tensor(gathered, outs[i % len(xs)])

    run("config4_gather_sync", step_sync, extra=gather_extra)
    def gather_alone():
        for _ in range(5):
            dist.all_gather_into_tensor(gathered, outs[0])

    fence()
    t0 = time.perf_counter()
    err = phase(1, lambda i: gather_alone(), None)
    fence()
    gdt = time.perf_counter() - t0
    bad = agree(err)
    if bad:
        res["config4_gather_alone"] = bad
    else:
        gms = reduce_max(gdt) / 5 * 1e3
        res["config4_gather_alone"] = {"ms": gms, "shard_MB": shard_bytes / 1e6, "GBps_per_rank": shard_bytes * (world - 1) / (gms * 1e-3) / 1e9,
                                       "GBps_per_link": shard_bytes / (gms * 1e-3) / 1e9,
                                       "note": "per link: one shard per peer over that peer's xGMI link (fully connected node); wall time of 5 back-to-back all-gathers between fences"}
    del gathered
    ov = OverlappedGather(oshape, torch.float32, dev, depth=len(xs))

    def step_ov(i):
        ov.wait_slot(i)
        plan.compute_batch(xs[i % len(xs)], out=outs[i % len(xs)])
        ov.submit(i, outs[i % len(xs)])

    run("config4_gather_overlap", step_ov, finish=ov.finish, extra=gather_extra)
    del ov
    # Strong scaling: ONE fixed job of `strong_total` utterances cut with sgx_shard_range — this rank computes only its
    # block, so the step time should fall as 1 / world; `value` = the job's frames per second (north_star: ">= 6x at 8 GPUs").
    if strong_total:
        from spectrograms_amd.distributed import shard_range
        _, cnt = shard_range(strong_total, world, dist.get_rank())
        reps = -(-cnt // batch)
        xsh = torch.cat([xs[0]] * reps)[:cnt].contiguous() if cnt else None
        osh = torch.empty((cnt, n_bins, n_frames), dtype=torch.float32, device=dev) if cnt else None
        run("config4_strong", (lambda i: plan.compute_batch(xsh, out=osh)) if cnt else (lambda i: None), frames=strong_total * n_frames,
            extra=lambda ms: {"scaling": "strong", "utterances_total": strong_total, "utterances_this_rank": cnt})
        del xsh, osh
    if cabi:
        # EXPERIMENTAL: sgx_shard_execute_chunked has run on a stand-in RCCL at 2-3 ranks and on the real RCCL at one rank only;
        # its numbers are not part of the default line until it has run on >= 2 real GPUs (`--experimental-legs cabi`).
        comm, err = None, None
        try:
            comm = make_comm()
            g2 = torch.empty((world * batch, n_bins, n_frames), dtype=torch.float32, device=dev)
        except Exception as e:
            err = e
        bad = agree(err)
        if bad:
            res["config4_cabi_chunked4"] = bad
        else:
            run("config4_cabi_chunked4", lambda i: comm.execute(plan, xs[i % len(xs)], world * batch, g2, chunks=4), extra=gather_extra)
            res["config4_cabi_chunked4"]["experimental"] = True
            fence()
        if comm is not None:
            comm.close()
    return res


def chirpz_leg(torch, sg, dev, xs256, args):
    """A frame length outside the power-of-two / listed sizes (the reference plans every length through RustFFT, src/fft_backend.rs:376-385):
    n_fft 1009 (prime) / hop 252, linear power, the first 64 utterances of the batch — the chirp-z kernel."""
    n_fft, hop, batch = 1009, 252, 64
    params = sg.SpectrogramParams(sg.StftParams(n_fft, hop, sg.WindowType.hanning, True), SR)
    plan = sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32")
    n_bins, n_frames = plan.output_shape(N_SAMPLES)
    xs = [x[:batch].contiguous() for x in xs256]
    outs = [torch.empty((batch, n_bins, n_frames), dtype=torch.float32, device=dev) for _ in xs]
    stream = torch.cuda.current_stream(dev)

    def step(i):
        plan.compute_batch(xs[i % len(xs)], out=outs[i % len(xs)])

    def fence():
        torch.cuda.synchronize(dev)

    ph = preheat(step, fence, min(args.preheat_s, 0.1))
    for i in range(args.warmup):
        step(i)
    fence()
    dt, kernel_ms = timed_steps(torch, stream, step, args.steps, fence)
    frames = batch * n_frames
    fps = frames / (kernel_ms * 1e-3)
    bpf = 4.0 * (N_SAMPLES / n_frames) + 4.0 * n_bins  # every sample once + the bins
    # two length-2048 complex transforms per frame PAIR (5 N log2 N each) + chirps, product and split
    flops = (2 * 5 * 2048 * 11 + 8 * 2048 + 6 * 2 * n_fft + 10 * n_bins) / 2.0
    return {"config": f"{batch} x 10 s 16 kHz f32, linear_power n_fft={n_fft} (prime) hop={hop} Hanning centre",
            "kernel": plan.kernel_name, "steps": args.steps, "warmup": args.warmup, "preheat_steps": ph,
            "ms_per_step": dt / args.steps * 1e3, "value": frames * args.steps / dt, "unit": "frames/s",
            "roofline": {"bound": "valu", "achieved": flops * fps / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops * fps / (VALU_PEAK_TFLOPS * 1e12), "kernel_ms": kernel_ms, "hbm_frac": bpf * fps / 1e9 / HBM_PEAK_GBS,
                         "flops_per_frame": flops, "algorithmic_bytes_per_frame": bpf, "traffic": None,
                         "note": "bound by its instruction count (VALU issue 71 % of the kernel's cycles, profiles/r03_chirpz_1009_rocprof_summary.txt); peak = FP32 vector"}}


def make_images(torch, dev, batch: int = IMG_BATCH):
    """BASELINE configs[4] input: img[r, c] = sin(0.01 r) + cos(0.02 c) + N(0, 0.05^2), 32 distinct noise fields tiled over the batch."""
    R = C = IMG_SIDE
    rng = np.random.default_rng(7)
    r, c = np.meshgrid(np.arange(R), np.arange(C), indexing="ij")
    base = (np.sin(0.01 * r) + np.cos(0.02 * c)).astype(np.float32)
    nz = 32
    host = base[None] + 0.05 * rng.standard_normal((nz, R, C), dtype=np.float32)
    return torch.from_numpy(np.ascontiguousarray(host)).to(dev).repeat(batch // nz, 1, 1)


def fft2d_legs(torch, sg, dev, which, args, peak):
    """BASELINE configs[4]: 512 x 1024 x 1024 f32 images, img[r, c] = sin(0.01 r) + cos(0.02 c) + N(0, 0.05^2) (32 distinct noise
    fields, tiled): `fft2d` alone and `convolve_fft` with gaussian_kernel_2d(9, 2.0).  Algorithmic bytes per image: fft2d 4 MiB
    read + 1024 * 513 * 8 B written; convolve_fft 4 MiB read + 4 MiB written (kernel spectrum cached by the plan; the Gaussian is an outer
    product, which the plan runs as two separable passes over pairs of real rows: DESIGN.md §4)."""
    R = C = IMG_SIDE
    x = make_images(torch, dev)
    plan = sg.Fft2dPlan(R, C, "float32")
    k = sg.gaussian_kernel_2d(9, 2.0, dtype="float32")
    stream = torch.cuda.current_stream(dev)
    res = {}
    flops = {"fft2d": 2.5 * R * C * 20.0, "convolve_fft": 2 * 2.5 * R * C * 20.0 + 6.0 * R * (C // 2 + 1)}
    for name in which:
        if name == "fft2d":
            buf = plan.forward_torch(x)
            fn = lambda i: plan.forward_torch(x, buf)
            bpi = R * C * 4 + R * (C // 2 + 1) * 8
        else:
            buf = plan.convolve_torch(x, k)
            fn = lambda i: plan.convolve_torch(x, k, buf)
            bpi = 2 * R * C * 4

        def fence():
            torch.cuda.synchronize(dev)

        fence()
        ph = preheat(fn, fence, args.preheat_s)
        steps, warm = max(3, min(args.steps, 20)), max(1, min(args.warmup, 5))
        for i in range(warm):
            fn(i)
        fence()
        dt, ms = timed_steps(torch, stream, fn, steps, fence)
        ips = IMG_BATCH / (ms * 1e-3)
        achieved = ips * bpi / 1e9
        tr = measured_traffic(name, detail=True)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": tr["bytes"] if tr else None,
                "kernel_ms": ms, "kernel_ms_scope": "HIP events over the timed region (all launches of a step)",
                "algorithmic_bytes_per_image": bpi, "images_per_step": IMG_BATCH, "valu_frac": flops[name] * ips / (VALU_PEAK_TFLOPS * 1e12),
                "limiter": "hbm"}
        if tr:  # the passes' real HBM bytes (PMC): what the multi-pass structure moves, against the algorithmic bytes
            roof["traffic_per_image"] = tr["bytes"] / IMG_BATCH
            roof["traffic_over_algorithmic"] = tr["bytes"] / (bpi * IMG_BATCH)
            roof["traffic_rate_GBps"] = tr["bytes"] / (ms * 1e-3) / 1e9
            roof["traffic_kernels"] = tr.get("kernels")
            if peak and peak.get("copy"):
                roof["traffic_rate_frac_of_measured_copy"] = roof["traffic_rate_GBps"] / peak["copy"]
        if peak and peak.get("copy"):
            roof["frac_of_measured_copy"] = achieved / peak["copy"]
        res[name] = {"config": f"configs[4]: {IMG_BATCH} x {R}x{C} f32 images, {name}" + (" with gaussian_kernel_2d(9, 2.0)" if name != "fft2d" else ""),
                     "steps": steps, "warmup": warm, "preheat_steps": ph, "ms_per_step": dt / steps * 1e3, "value": IMG_BATCH * steps / dt,
                     "unit": "images/s", "roofline": roof}
        del buf
    return res


def device_identity(torch, dev):
    """(number, strong): the number is the same for two ranks when they use the same device ordinal of the same physical GPU.  The
    ordinal alone repeats legitimately when every rank is confined to one visible device (HIP_VISIBLE_DEVICES per rank, ordinal 0), so
    the rank's visible-device lists are part of the identity; `strong` says the runtime reported a real uuid or PCI address — only then
    does a repeat prove that two ranks share a GPU (a build that reports placeholders for every GPU must not make a correct launch
    look wrong: such a repeat is recorded in the JSON line, not fatal)."""
    p = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(p, "uuid", "") or "")
    pci = "%s:%s:%s" % (getattr(p, "pci_domain_id", ""), getattr(p, "pci_bus_id", ""), getattr(p, "pci_device_id", ""))
    strong = bool(uuid.strip("0-")) or pci not in ("::", "0:0:0")
    vis = "|".join(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    ident = "%d|%s|%s|%s" % (dev.index, uuid, pci, vis)
    return int.from_bytes(hashlib.sha256(ident.encode()).digest()[:7], "little"), strong


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
    legs = [l for l in args.legs.split(",") if l]
    if any(l not in LEGS for l in legs):
        print(f"bench.py: --legs takes a subset of {','.join(LEGS)}", file=sys.stderr)
        return 2

    kernel_wl, batch, cfg_idx = WORKLOADS[args.workload]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_run:
        return dry_run(args, rank, world)
    # the CPU baseline runs FIRST, before this process has touched the GPU: its all-core leg forks worker processes, and a process
    # that has initialised HIP must not be forked.  (rank 0 at N = 1 only; a bounded sample: ~10 s port + 2 x 3 s pocketfft)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(kernel_wl)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the product path has no CPU fallback", file=sys.stderr)
        return 2
    import spectrograms_amd as sg
    from spectrograms_amd import _ffi

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    device_note = None
    rehearse = args.rehearse_multi_rank and world == 1
    if rehearse:
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        rccl_ranks = dist.get_world_size()
        # one rank per GPU: two ranks on one device would report a scaling curve of a time-shared GPU
        ident, strong = device_identity(torch, dev)
        ids = torch.empty(2 * world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(ids, torch.tensor([ident, int(strong)], dtype=torch.int64, device=dev))
        ids = ids.view(world, 2).tolist()
        distinct = len({i for i, _ in ids})
        if distinct != world:
            if any(st for _, st in ids):
                if rank == 0:
                    print(f"bench.py: {world} ranks but only {distinct} distinct GPUs — a device ordinal repeats", file=sys.stderr)
                dist.destroy_process_group()
                return 3
            device_note = f"{world} ranks, {distinct} distinct device identities, but the runtime reports no uuid / PCI address: not checked"

    # ---- plan + synthetic device-resident batch (weak scaling: every rank owns `batch` utterances)
    plan = make_plan(sg, kernel_wl)
    n_bins, n_frames = plan.output_shape(N_SAMPLES)
    # the generator has 96 distinct rows (48 pitches + 48 noise seeds would repeat the sines); build 256 and tile for config 4
    base = np.stack([cfg_signal(rank * batch + b) for b in range(min(batch, 256))])
    nsets = 2  # rotate buffer sets so a step never re-reads its own input out of the 256 MiB Infinity Cache
    xs256 = [torch.from_numpy(base).to(dev) for _ in range(nsets)]
    xs = xs256 if batch <= 256 else [torch.cat([x] * (batch // 256)) for x in xs256]
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

    def reduce_max(v: float) -> float:
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- cold clocks first: the W + K steps a caller gets who launches from an idle GPU (no preheat; nothing has run yet but the
    # one launch that loads the code object)
    step(0)
    fence()
    cold_dt, _ = timed_steps(torch, stream, step, args.warmup + args.steps, fence)
    cold_ms = reduce_max(cold_dt) / (args.warmup + args.steps) * 1e3

    preheat_steps = preheat(step, fence, args.preheat_s, reduce_max)
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
    dt, ev_ms = timed_steps(torch, stream, step, args.steps, fence)  # the stream the kernels are launched on (torch's current stream)
    if gathered is None and overlap is None:
        kernel_ms = ev_ms  # mean launch duration over the timed region itself
    dt = reduce_max(dt)
    kmin = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
    kmax = kmin.clone()
    if world > 1:
        dist.all_reduce(kmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)

    extra = world == 1 and not args.no_legs and not rehearse
    peak = measured_peak(_ffi.lib(), local_rank) if extra else None  # right behind the timed steps: steady clocks
    sustained = None
    if extra and args.sustained_s > 0:
        sustained = sustained_leg(torch, _ffi.lib(), dev, step, fence, batch * n_frames, args.sustained_s)
    frames_per_launch = batch * n_frames
    frames_per_step = frames_per_launch * world
    value = frames_per_step * args.steps / dt
    kernel_fps = frames_per_launch / (kernel_ms * 1e-3)
    line = None
    if rank == 0:
        scope = "HIP events over the timed region" if (gathered is None and overlap is None) else "compute only (separate back-to-back launches, no gather)"
        line = {
            "metric": "STFT frames/sec (f32, n_fft=1024 hop=256)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "preheat_steps": preheat_steps,
            "ms_per_step": dt / args.steps * 1e3,
            # the same W + K steps launched from idle clocks, before anything else ran on the device (no preheat)
            "cold_ms_per_step": cold_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[{cfg_idx}]: {batch} x 10 s 16 kHz f32 per GPU, {kernel_wl} n_fft=1024 hop=256 Hanning centre"
                                   + (" (8192 utterances over 8 GPUs)" if args.workload == "config4" else ""),
                       "batch_per_gpu": batch, "n_samples": N_SAMPLES, "frames_per_step": frames_per_step, "kernel": plan.kernel_name,
                       "gather": (args.gather if (gathered is not None or overlap is not None) else False),
                       "parallelism": f"utterance-shard x{world}"},
            "rccl_ranks": rccl_ranks, "kernel_ms_min": float(kmin.item()), "kernel_ms_max": float(kmax.item()),
            **({"device_check": device_note} if device_note else {}),
            # the dominant kernel, measured live (hipEvents around back-to-back launches on the launch stream, no gather)
            "roofline": stft_roofline(kernel_wl, batch, n_frames, kernel_ms, peak, scope, full_peak=True),
        }
        if gathered is not None or overlap is not None:
            shard_bytes = float(np.prod(oshape)) * 4.0
            line["gather"] = {"mode": args.gather, "compute_only_value": kernel_fps * world, "shard_MB": shard_bytes / 1e6,
                              "gather_ms": gather_ms,
                              # ring all-gather: every rank sends and receives (world - 1) shards over its links
                              "GBps_per_rank": None if not gather_ms else shard_bytes * (world - 1) / (gather_ms * 1e-3) / 1e9}
    emitted = []

    def emit():
        if rank == 0 and not emitted:
            emitted.append(1)
            if cpu is not None:
                line["cpu_baseline"] = cpu
            print(json.dumps(line), flush=True)

    # A leg that hangs (a first-run RCCL path, say) must not cost the headline line: past `--legs-timeout` seconds the line is
    # printed with the legs finished so far and the process leaves.
    watchdog = None
    if (extra or rehearse or (world > 1 and not args.no_legs)) and args.legs_timeout > 0:
        import threading

        def bail():
            if rank == 0:
                line["workloads_error"] = f"legs not finished after {args.legs_timeout:.0f} s: line printed without the rest"
            emit()
            # non-zero: a hung leg is a failure the driver / CI must see; the line above still carries the headline and the
            # finished legs.  No in-process restart, no exec: the process just leaves (a rank stuck in a collective cannot be joined).
            os._exit(4)

        watchdog = threading.Timer(args.legs_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
    if sustained is not None:
        line["sustained"] = sustained
    if extra:
        del outs, xs
        torch.cuda.empty_cache()
        wl = line.setdefault("workloads", {})
        for name in legs:
            if name in ("fft2d", "convolve_fft") or name == args.workload:
                continue
            if name == "chirpz_1009":
                wl[name] = chirpz_leg(torch, sg, dev, xs256, args)
            elif name == "istft":
                wl[name] = istft_leg(torch, sg, dev, xs256, args, peak)
            elif name.endswith("_f64"):
                wl[name] = f64_leg(torch, sg, dev, name, xs256, args, peak)
            else:
                wl[name] = stft_leg(torch, sg, dev, name, xs256, args, peak)
            torch.cuda.empty_cache()
        two_d = [l for l in legs if l in ("fft2d", "convolve_fft")]
        if two_d:
            del xs256
            torch.cuda.empty_cache()
            wl.update(fft2d_legs(torch, sg, dev, two_d, args, peak))
    elif (world > 1 and not args.no_legs) or rehearse:
        # N > 1: BASELINE configs[3] with and without the exchange (every rank takes part; rank 0 reports)
        del outs, xs
        torch.cuda.empty_cache()
        from spectrograms_amd.distributed import ShardComm
        wl4, b4, _ = WORKLOADS["config4"]
        mr = multi_rank_legs(torch, dist, dev, world, make_plan(sg, wl4), [torch.cat([x] * (b4 // 256)) for x in xs256],
                             max(3, min(args.steps, 20)), max(1, min(args.warmup, 3)), reduce_max, lambda: torch.cuda.synchronize(dev),
                             lambda: ShardComm(dev), f"configs[3]: {b4} x 10 s per GPU x {world} GPUs, mel_power n_fft=1024 hop=256",
                             cabi="cabi" in args.experimental_legs.split(","), strong_total=8192)
        if rank == 0:
            line["workloads"] = mr
    if watchdog is not None:
        watchdog.cancel()
    emit()
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def dry_run(args, rank: int, world: int) -> int:
    """The launcher / timing skeleton without a GPU: gloo rendezvous, barrier, K empty steps, max over ranks, the per-rank
    min / max reduction, the one-rank-per-device check (here: the rank number stands in for the device), one JSON line."""
    import torch
    import torch.distributed as dist

    ranks = 1
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ranks = dist.get_world_size()
        ids = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        fake = int(os.environ.get("SGX_DRY_DEVICE", rank))  # tests: two ranks claiming one device must be refused
        dist.all_gather(ids, torch.tensor([fake], dtype=torch.int64))
        if len({int(t.item()) for t in ids}) != world:
            if rank == 0:
                print(f"bench.py: {world} ranks but only {len({int(t.item()) for t in ids})} distinct GPUs — a device ordinal repeats", file=sys.stderr)
            dist.destroy_process_group()
            return 3
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-4)
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    dt = torch.tensor([el], dtype=torch.float64)
    kmin, kmax = dt.clone() / args.steps * 1e3, dt.clone() / args.steps * 1e3
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    _, batch, cfg_idx = WORKLOADS[args.workload]
    workloads = None
    if world > 1 and not args.no_legs:
        class StandInPlan:  # the launcher's legs without a GPU: a "launch" fills its output
            def output_shape(self, n):
                return 8, 5

            def compute_batch(self, x, out=None):
                out.fill_(float(x[0, 0]))
                return out

        def no_comm():
            raise RuntimeError("dry run: no HIP device for the C ABI's communicator")

        def rmax(v):
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        workloads = multi_rank_legs(torch, dist, "cpu", world, StandInPlan(), [torch.full((4, 64), float(rank + 1)) for _ in range(2)], 3, 1, rmax,
                                    lambda: None, no_comm, "dry run: stand-in plan, gloo", cabi="cabi" in args.experimental_legs.split(","), strong_total=11)
    if rank == 0:
        print(json.dumps({"metric": "STFT frames/sec (f32, n_fft=1024 hop=256)", "value": 0.0, "unit": "frames/s", "n_gpus": world,
                          **({"workloads": workloads} if workloads else {}),
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(dt.item()) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "dry-run (no GPU work)",
                          "rccl_ranks": ranks, "kernel_ms_min": float(kmin.item()), "kernel_ms_max": float(kmax.item()),
                          "config": {"workload": f"configs[{cfg_idx}] dry run", "batch_per_gpu": batch, "gather": args.gather or False,
                                     "parallelism": f"utterance-shard x{world}"}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
