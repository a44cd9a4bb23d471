import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import bench, spectrograms_amd as sg
params = sg.SpectrogramParams(sg.StftParams(1024, 256, sg.WindowType.hanning, True), 16000.0)
plan = sg.SpectrogramPlanner().linear_power_plan(params, dtype="float32")
x = np.stack([bench.cfg_signal(b) for b in range(256)])
out = plan.compute_batch(x)
t0 = time.perf_counter()
for _ in range(5): plan.compute_batch(x, out=out)
dt = (time.perf_counter() - t0) / 5
print(f"host-pointer path (H2D + kernel + D2H, pageable numpy): {dt*1e3:.2f} ms per batch = {160256/dt/1e6:.1f} Mframes/s")
mel = sg.SpectrogramPlanner().mel_db_plan(params, sg.MelParams(80, 0.0, 8000.0), sg.LogParams(-80.0), dtype="float32")
o2 = mel.compute_batch(x)
t0 = time.perf_counter()
for _ in range(5): mel.compute_batch(x, out=o2)
dt = (time.perf_counter() - t0) / 5
print(f"mel_db host-pointer path: {dt*1e3:.2f} ms per batch = {160256/dt/1e6:.1f} Mframes/s")
