"""Drop-in for ``compute_trajectory_metrics`` (reference analysis/metrics/trajectory_metrics.py:12-325).

The ~300 ``torch.norm(...).item()`` calls and 51 scipy sorts per pair of the reference are replaced
by three device reductions over the stacked trajectories (csrc/dt_metrics.hip); the scalar
post-transforms (log1p / exp / min-max ratios, NaN behaviour) are replayed on the host exactly as
the reference writes them (engine.metrics_from_sums).  Plotting (reference :327-715) is out of scope.
"""
import numpy as np
import torch

from ... import engine


def _images(traj):
    """Entries may be tensors or (tensor, timestep) tuples (reference :29-37)."""
    return [e[0] for e in traj] if isinstance(traj[0], tuple) else list(traj)


def _stack_on_device(images, device):
    """[n, 1, E] fp32 on ``device``: every list entry [B,C,H,W] is ONE point of the trajectory (the
    reference's norms run over the whole entry)."""
    t = torch.stack([im.detach().reshape(-1) for im in images]).to(device=device, dtype=torch.float32)
    return t.unsqueeze(1).contiguous()


def _metrics_device():
    if not torch.cuda.is_available():
        raise engine.HipLibraryError("compute_trajectory_metrics needs a HIP device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def wasserstein_index_tables(seeds, n, E, sample_size=1000):
    """Coordinate tables of the reference's Wasserstein sub-sampling for the engine's seeded pairs.

    reference :303 draws ``np.random.choice(E, 1000, replace=False)`` from the GLOBAL numpy generator
    once per step; inside compare_trajectories that generator was last seeded with ``seed + 1`` by the
    student's generate_trajectory (trajectory_engine.py:93, t = 1), so the tables depend on the sample
    seed only.  Returns (int32 [n_unique, n, sample_size], int32 [len(seeds)] row of each pair).
    """
    uniq = sorted(set(seeds))
    tables = np.empty((len(uniq), n, sample_size), dtype=np.int32)
    for r, seed in enumerate(uniq):
        rs = np.random.RandomState(seed + 1 if n > 2 else seed)
        for i in range(n):
            tables[r, i] = rs.choice(E, sample_size, replace=False)
    row = torch.tensor([uniq.index(s) for s in seeds], dtype=torch.int32)
    return torch.from_numpy(tables), row


def _resize_like(Y, size, device):
    """reference :40-52: every student entry resized to the teacher's H x W (bilinear, align_corners=True) -- one
    dt_resize_bilinear launch over all entries."""
    rows = [y.shape[0] for y in Y]
    out = engine.resize_bilinear(torch.cat([y.detach().to(device=device, dtype=torch.float32) for y in Y]), size)
    return list(torch.split(out, rows))


def _prepare(teacher_trajectory, student_trajectory):
    X, Y = _images(teacher_trajectory), _images(student_trajectory)
    device = X[0].device if X[0].is_cuda else _metrics_device()
    if X[-1].shape != Y[-1].shape and X[-1].shape[2:] != Y[-1].shape[2:]:
        Y = _resize_like(Y, X[0].shape[2:], device)
    return X, Y, device


def compute_trajectory_metrics(teacher_trajectory, student_trajectory, config=None):
    """25-key metric dict for one (teacher, student) pair; same keys and types as the reference."""
    return compute_trajectory_metrics_many([(teacher_trajectory, student_trajectory)], config)[0]


def compute_trajectory_metrics_many(pairs, config=None):
    """``[compute_trajectory_metrics(t, s, config) for t, s in pairs]`` with the device work batched: pairs whose
    trajectories have the same lengths and entry shape are stacked and reduced by ONE dt_traj_metrics, ONE
    dt_traj_wasserstein (and one dt_traj_resampled_distance for unequal lengths) launch per group, followed by one
    device-to-host copy; the scalar post-transforms then run per pair on the host.

    The Wasserstein coordinates are drawn from the GLOBAL numpy generator pair by pair, one ``np.random.choice`` per zipped
    step, exactly as a loop over the reference's function would (:303) -- also when E <= 1000, where the draw is a full
    permutation that cannot change the result but does advance the generator."""
    prepared, groups = [], {}
    for k, (tt, st) in enumerate(pairs):
        X, Y, device = _prepare(tt, st)
        shape = tuple(X[0].shape)
        E = int(np.prod(shape))
        n = min(len(X), len(Y))
        cnt = min(1000, E)
        draws = [np.random.choice(E, cnt, replace=False) for _ in range(n)]
        prepared.append((X, Y, device, shape, E, draws))
        groups.setdefault((len(X), len(Y), shape, str(device)), []).append(k)
    out = [None] * len(pairs)
    for (nT, nS, shape, _), members in groups.items():
        device, E = prepared[members[0]][2], prepared[members[0]][4]
        P, n = len(members), min(nT, nS)
        # [n, P, E]: every list entry [B,C,H,W] is ONE point of a trajectory (the reference's norms run over the whole entry)
        Xd = torch.stack([_stack_on_device(prepared[k][0], device)[:, 0] for k in members], dim=1).contiguous()
        Yd = torch.stack([_stack_on_device(prepared[k][1], device)[:, 0] for k in members], dim=1).contiguous()
        index = index_row = None
        if E > 1000:
            index = torch.from_numpy(np.stack([np.stack(prepared[k][5]) for k in members]).astype(np.int32)).to(device)
            index_row = torch.arange(P, dtype=torch.int32, device=device)
        sums_d, w1_d = engine.device_pair_metrics(Xd, Yd, index, index_row)   # one launch for equal lengths and E <= 1000
        parts = [sums_d.reshape(P, -1)]
        if nT != nS:
            longer, shorter = (Xd, Yd) if nT > nS else (Yd, Xd)
            parts.append(engine.device_resampled_distance(longer, shorter))
        parts.append(w1_d)
        host = torch.cat(parts, dim=1).cpu().numpy()                 # the one device-to-host copy of the group
        n_max = max(nT, nS)
        pixels = shape[2] * shape[3]
        for row, k in enumerate(members):
            sums = host[row, : 4 * n_max].reshape(n_max, 4)
            off = 4 * n_max
            resampled = None
            if nT != nS:
                resampled = host[row, off: off + n]
                off += n
            out[k] = engine.metrics_from_sums(sums, host[row, off: off + n], nT, nS, pixels, E, resampled)
    return out
