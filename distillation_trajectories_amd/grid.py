"""The (size factor x guidance scale x sample) grid, sharded by sample across the GPUs of one node.

The reference walks this grid with three nested python loops of B=1 trajectories
(analysis/trajectory_engine.py:142-164 inside scripts/analysis/analyze_trajectory_metrics.py:476-505).
Cells are independent; only the final per-(sf, gs) mean over samples couples them (:171-175).  Here:
  * one process per GPU owns a contiguous slice of the sample axis (load-balanced: every rank runs
    every model, so teacher/tiny-student cost differences do not skew the shards);
  * within a rank the teacher runs once per CFG plan with batch = samples x guidance scales and is
    reused for every student;
  * the only exchange is one all-gather (RCCL over xGMI) of the per-sample scalar metrics
    [n_sf, n_gs, S_local, 19] float64; every rank then averages in sample order in float64, which
    reproduces the reference's ``sum(...) / len(...)`` exactly.
"""
import numpy as np
import torch

from . import engine
from .analysis.trajectory_engine import sample_grid
from .synthetic import noise_table

K = len(engine.SCALAR_KEYS)


def shard_range(n, rank, world):
    """Contiguous balanced slice [start, start+count) of n samples for ``rank`` of ``world``."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def all_gather_rows(local, counts, dim):
    """Concatenate every rank's ``local`` along ``dim`` (row counts may differ by one). No-op without a
    process group.  Uses the default group's backend: nccl (= RCCL) on GPUs, gloo in the CPU tests."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    if dist.get_backend() != "nccl" and local.is_cuda:
        # CPU rehearsal backends (gloo): exchange on the host, hand the result back on the device
        return all_gather_rows(local.cpu(), counts, dim).to(local.device)
    pad = max(counts)
    shape = list(local.shape)
    shape[dim] = pad
    buf = local.new_zeros(shape)
    buf.narrow(dim, 0, local.shape[dim]).copy_(local)
    parts = [torch.empty_like(buf) for _ in counts]
    dist.all_gather(parts, buf.contiguous())
    return torch.cat([p.narrow(dim, 0, c) for p, c in zip(parts, counts)], dim=dim)


def hip_cell_metrics(teacher, students, cfg, guidance_scales, first_sample, count, device):
    """float64 tensor [n_sf, n_gs, count, K] on ``device`` for samples first_sample .. +count-1 (seed 42+s)."""
    C, H, T = cfg.channels, cfg.image_size, cfg.timesteps
    state = torch.get_rng_state()
    table = noise_table(42 + first_sample, count + T - 1, (1, C, H, H)).reshape(count + T - 1, -1).to(device)
    torch.set_rng_state(state)
    th = engine.UNetHandle.for_module(teacher)
    t_traj = sample_grid(th, table, 0, count, T, guidance_scales, H, H)
    out = torch.empty(len(students), len(guidance_scales), count, K, dtype=torch.float64, device=device)
    for i, student in enumerate(students):
        s_traj = sample_grid(engine.UNetHandle.for_module(student), table, 0, count, T, guidance_scales, H, H)
        for j, gs in enumerate(guidance_scales):
            X, Y = t_traj[gs].contiguous(), s_traj[gs].contiguous()
            sums = engine.device_metric_sums(X, Y).cpu().numpy()
            if X.shape[2] > 1000:
                from .analysis.metrics.trajectory_metrics import wasserstein_index_tables
                tables, rows = wasserstein_index_tables([42 + first_sample + s for s in range(count)], T + 1, X.shape[2])
                w1 = engine.device_wasserstein(X, Y, tables.to(device), rows.to(device)).cpu().numpy()
            else:
                w1 = engine.device_wasserstein(X, Y).cpu().numpy()
            vals = engine.batch_scalar_metrics(sums, w1, H * H, X.shape[2])
            out[i, j] = torch.from_numpy(np.stack([vals[k] for k in engine.SCALAR_KEYS], axis=1)).to(device)
    return out


def grid_metrics(teacher, students, cfg, guidance_scales, num_samples, rank=0, world=1, device=None, cell_fn=None):
    """Averaged scalar metrics of the whole grid on every rank: ``result[i_student][gs][key] -> float``.

    ``cell_fn(first_sample, count) -> tensor [n_sf, n_gs, count, K]`` computes one shard (default: the
    HIP path); the CPU/gloo tests inject an oracle-backed function to exercise sharding + collective.
    """
    start, count = shard_range(num_samples, rank, world)
    if cell_fn is None:
        device = device or next(teacher.parameters()).device
        local = hip_cell_metrics(teacher, students, cfg, list(guidance_scales), start, count, device)
    else:
        local = cell_fn(start, count)
    counts = [shard_range(num_samples, r, world)[1] for r in range(world)]
    full = all_gather_rows(local, counts, dim=2).cpu().numpy()        # [n_sf, n_gs, S, K]
    result = []
    for i in range(full.shape[0]):
        per_gs = {}
        for j, gs in enumerate(guidance_scales):
            cell = full[i, j]
            # python's sum(): sequential float64 accumulation in sample order (trajectory_engine.py:174)
            per_gs[gs] = {k: float(np.cumsum(cell[:, c])[-1] / cell.shape[0]) for c, k in enumerate(engine.SCALAR_KEYS)}
        result.append(per_gs)
    return result
