"""configs[2] step: how much of it is the serial host tail (the one D2H of all students' reductions + the vectorised scalar
post-transform), measured by wrapping engine.batch_scalar_metrics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from distillation_trajectories_amd import engine
spec = bench.CONFIGS[2]
torch.cuda.set_device(0)
wl = bench.GridWorkload(spec, torch.device("cuda:0"), 0, spec["batch"])
acc = {"t": 0.0, "n": 0}
orig = engine.batch_scalar_metrics
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); acc["t"] += time.perf_counter() - t0; acc["n"] += 1; return r
engine.batch_scalar_metrics = timed
for _ in range(2):
    wl.step(1, [spec["batch"]])
torch.cuda.synchronize()
acc["t"] = 0.0; acc["n"] = 0
t0 = time.perf_counter()
K = 5
for _ in range(K):
    wl.step(1, [spec["batch"]])
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / K
print(f"step {el * 1e3:.1f} ms, host post-transform {acc['t'] / K * 1e3:.2f} ms per step ({acc['n'] // K} call per step)")
