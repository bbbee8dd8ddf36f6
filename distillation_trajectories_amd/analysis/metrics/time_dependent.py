"""Consecutive-state distances per timestep (reference analysis/metrics/time_dependent.py:10-120).

The per-trajectory L2 distances |x_i - x_{i-1}| come from the same device reduction as the
velocity terms of compute_trajectory_metrics (dt_traj_metrics, slots 1/2); the averaging over
trajectories and the population std are the reference's python arithmetic.  Plotting (:122-150) is
out of scope.
"""
import numpy as np
import torch

from ... import engine
from .trajectory_metrics import _images, _metrics_device


def _distances(trajectories):
    """List (per trajectory) of python-float distance lists, computed on the device in one launch per length."""
    out = [None] * len(trajectories)
    by_len = {}
    for k, tr in enumerate(trajectories):
        by_len.setdefault(len(tr), []).append(k)
    for n, ks in by_len.items():
        if n < 2:
            for k in ks:
                out[k] = []
            continue
        ims = [_images(trajectories[k]) for k in ks]
        device = ims[0][0].device if ims[0][0].is_cuda else _metrics_device()
        X = torch.stack([torch.stack([im.detach().reshape(-1) for im in tr]) for tr in ims], dim=1)
        X = X.to(device=device, dtype=torch.float32).contiguous()             # [n, len(ks), E]
        sums = engine.device_metric_sums(X, X).cpu().numpy().astype(np.float32)
        for j, k in enumerate(ks):
            out[k] = [float(np.sqrt(sums[j, i, 1])) for i in range(1, n)]
    return [d for d in out if d]


def _per_timestep(all_d):
    if not all_d:
        return [], 0, 0
    L = min(len(d) for d in all_d)
    avg = [sum(d[t] for d in all_d) / len(all_d) for t in range(L)]
    mean = sum(avg) / len(avg) if avg else 0
    std = (sum((a - mean) ** 2 for a in avg) / len(avg)) ** 0.5 if avg else 0
    return avg, mean, std


def analyze_time_dependent_distances(teacher_trajectories, student_trajectories, config, size_factor=None, save_dir=None):
    """Same result dict as the reference (numeric keys); ``save_dir`` plotting is not provided."""
    results = {"teacher_distances": [], "student_distances": [], "teacher_avg_distance": 0, "student_avg_distance": 0,
               "teacher_std_distance": 0, "student_std_distance": 0, "size_factor": size_factor}
    if not teacher_trajectories or not student_trajectories:
        return results
    td, sd = _distances(teacher_trajectories), _distances(student_trajectories)
    results["teacher_distances"], results["student_distances"] = td, sd
    ta, tm, ts = _per_timestep(td) if td and sd else ([], 0, 0)
    sa, sm, ss = _per_timestep(sd) if td and sd else ([], 0, 0)
    results.update(teacher_avg_per_timestep=ta, student_avg_per_timestep=sa, teacher_avg_distance=tm,
                   student_avg_distance=sm, teacher_std_distance=ts, student_std_distance=ss)
    return results
