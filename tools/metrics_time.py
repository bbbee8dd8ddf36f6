"""Wall time of the metric reductions at the configs[1] shape (256 pairs x 51 states x 768 coordinates): the one-pass kernel
(dt_traj_pair_metrics) against the two separate kernels, 50 back-to-back launches each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distillation_trajectories_amd import engine
n, B, E = 51, int(os.environ.get("DT_B", "256")), int(os.environ.get("DT_E", "768"))
g = torch.Generator().manual_seed(6)
X = (torch.randn(n, B, E, generator=g).cumsum(0) * 0.05).cuda()
Y = X + 0.02 * torch.randn(n, B, E, generator=g).cuda()
for name, fn in (("pair_metrics (one pass)", lambda: engine.device_pair_metrics(X, Y)),
                 ("traj_metrics", lambda: engine.device_metric_sums(X, Y)), ("wasserstein", lambda: engine.device_wasserstein(X, Y))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    print(f"{name:26s} {us:8.1f} us   {8.0 * n * B * E / us / 1e6:7.2f} TB/s of algorithmic bytes (one read of both trajectories)")
