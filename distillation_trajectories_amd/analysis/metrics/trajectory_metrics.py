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


def compute_trajectory_metrics(teacher_trajectory, student_trajectory, config=None):
    """25-key metric dict for one (teacher, student) pair; same keys and types as the reference."""
    X, Y = _images(teacher_trajectory), _images(student_trajectory)
    device = X[0].device if X[0].is_cuda else _metrics_device()
    if X[-1].shape != Y[-1].shape and X[-1].shape[2:] != Y[-1].shape[2:]:
        # reference :40-52; never reached by the models of this repo (all preserve H, W)
        Y = [torch.nn.functional.interpolate(y.to(device), size=X[0].shape[2:], mode="bilinear", align_corners=True)
             for y in Y]
    nT, nS = len(X), len(Y)
    n = min(nT, nS)
    shape = X[0].shape
    pixels = shape[2] * shape[3]
    elems = shape[1] * shape[2] * shape[3]          # reference's total_elements (:68) -- only documents E
    Xd, Yd = _stack_on_device(X, device), _stack_on_device(Y, device)
    E = Xd.shape[2]
    sums = engine.device_metric_sums(Xd, Yd)[0].cpu().numpy()
    # torch.mean divides by the FULL entry size (batch included)
    resampled = None
    if nT != nS:
        longer, shorter = (Xd, Yd) if nT > nS else (Yd, Xd)
        resampled = engine.device_resampled_distance(longer, shorter)[0].cpu().numpy()
    # Wasserstein coordinates: one global-generator draw per zipped step, like the reference (:303)
    cnt = min(1000, E)
    draws = [np.random.choice(E, cnt, replace=False) for _ in range(n)]
    index = None
    if E > 1000:
        index = torch.from_numpy(np.stack(draws).astype(np.int32)).unsqueeze(0).to(device)
    w1 = engine.device_wasserstein(Xd, Yd, index)[0].cpu().numpy()
    del elems
    return engine.metrics_from_sums(sums, w1, nT, nS, pixels, E, resampled)
