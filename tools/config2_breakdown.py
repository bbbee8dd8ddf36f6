"""configs[2] per-model breakdown: wall time of each model's plain (B=S, 1 pass) and guided (B=3S, 2 passes) sampler loops
run alone on an idle GPU, their FLOP rate, and the launch plan of one forward.  DT_C2_SIZES=0.01,0.5 restricts the sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from distillation_trajectories_amd import engine
from distillation_trajectories_amd.analysis.trajectory_engine import sample_grid_groups
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model, noise_table

MACS = {0.01: 3.43e6, 0.1: 3.43e6, 0.2: 8.27e6, 0.3: 18.9e6, 0.4: 34.0e6, 0.5: 53.3e6, 0.6: 75.1e6, 0.7: 102.8e6, 0.8: 134.9e6,
        0.9: 171.4e6, 1.0: 212.2e6}
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
S, T, H, C = int(os.environ.get("DT_C2_S", "64")), 50, 16, 3
cfg = Config(); cfg.image_size, cfg.timesteps = H, T
table = noise_table(42, S + T - 1, (1, C, H, H)).reshape(S + T - 1, -1).to(dev)
sizes = [float(v) for v in os.environ.get("DT_C2_SIZES", ",".join(map(str, bench.SIZES))).split(",")]
tot = [0.0, 0.0, 0.0]
for sf in sizes:
    m = make_model(DiffusionUNet, cfg, sf).to(dev)
    h = engine.UNetHandle.for_module(m)
    row = []
    for k, scales in enumerate(([1.0], [3.0, 7.0, 20.0], [1.0, 3.0, 7.0, 20.0])):
        fn = lambda: sample_grid_groups(h, table, 0, S, T, scales, H, H)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        rows = (S, 6 * S, 7 * S)[k]
        tf = 2 * MACS[sf] * rows * (T - 1) / (ms * 1e-3) / 1e12
        row.append((ms, tf))
        tot[k] += ms
    print(f"sf {sf:4}: plain B={S} {row[0][0]:7.2f} ms ({row[0][0] / 49 * 1e3:6.0f} us/fwd, {row[0][1]:6.1f} TF/s)   guided B={3 * S}x2 {row[1][0]:7.2f} ms "
          f"({row[1][0] / 49 * 1e3:6.0f} us/fwd, {row[1][1]:6.1f} TF/s)   mixed {7 * S} rows {row[2][0]:7.2f} ms ({row[2][0] / 49 * 1e3:6.0f} us/fwd, {row[2][1]:6.1f} TF/s)", flush=True)
    if os.environ.get("DT_C2_PLAN"):
        for bt in (7 * S,):
            print("   ", bt, [(c[0], c[1], c[2], c[3], c[4], c[5]) for c in h.conv_choices(bt, H, H)])
print(f"sum plain {tot[0]:.1f} ms, guided {tot[1]:.1f} ms, mixed {tot[2]:.1f} ms (the teacher = sf 1.0 runs twice in the grid)")
