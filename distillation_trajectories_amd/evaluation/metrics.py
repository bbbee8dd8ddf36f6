"""Trajectory divergence (reference evaluation/metrics.py:118-183) on the HIP metric kernels.

Only ``compute_trajectory_divergence`` is provided: LPIPS and FID in the reference's module need
pretrained networks that cannot be fetched offline.  Distances, cosine
similarities and path lengths come from two device reductions (dt_pair_stats, dt_traj_metrics) over the
stacked trajectories instead of 3 x len python loops of ``torch.norm(...).item()`` / sklearn calls.
"""
import numpy as np
import torch

from .. import engine
from ..analysis.metrics.trajectory_metrics import _metrics_device, _stack_on_device


def compute_trajectory_divergence(trajectory1, trajectory2):
    """Same keys as the reference: distances, similarities, avg/max distance, avg/min similarity, length_ratio.
    Trajectories are lists of (image, timestep) pairs (the reference indexes ``item[0]`` unconditionally)."""
    im1 = [item[0] for item in trajectory1]
    im2 = [item[0] for item in trajectory2]
    device = im1[0].device if im1[0].is_cuda else _metrics_device()
    X, Y = _stack_on_device(im1, device), _stack_on_device(im2, device)
    n = min(len(im1), len(im2))
    f32 = np.float32
    stats = engine.device_pair_stats(X[:n].contiguous(), Y[:n].contiguous())[0].cpu().numpy().astype(f32)   # [n,5]
    sums = engine.device_metric_sums(X, Y)[0].cpu().numpy().astype(f32)
    distances = [float(np.sqrt(stats[i, 0])) for i in range(n)]
    sims = []
    for i in range(n):
        nx, ny = np.sqrt(stats[i, 3]), np.sqrt(stats[i, 4])
        nx = nx if nx > 0 else f32(1.0)            # sklearn normalises zero rows with a unit scale
        ny = ny if ny > 0 else f32(1.0)
        sims.append(f32(stats[i, 2] / (nx * ny)))
    length1 = 0
    for i in range(1, len(im1)):
        length1 += float(np.sqrt(sums[i, 1]))
    length2 = 0
    for i in range(1, len(im2)):
        length2 += float(np.sqrt(sums[i, 2]))
    return {
        "distances": distances,
        "similarities": sims,
        "avg_distance": np.mean(distances),
        "max_distance": np.max(distances),
        "avg_similarity": np.mean(sims),
        "min_similarity": np.min(sims),
        "length_ratio": length2 / length1 if length1 > 0 else float("inf"),
    }
