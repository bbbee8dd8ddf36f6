"""Drop-in for the reference's ``analysis/trajectory_engine.py`` on the HIP path.

``generate_trajectory`` keeps the reference signature and list-of-CPU-tensors result
(reference :24-115).  ``compare_trajectories`` (reference :117-180) returns the same dict, but instead
of 2 x num_samples x len(guidance_scales) sequential B=1 loops it runs each model ONCE per CFG plan
with batch = samples x guidance scales, keeps all trajectories in HBM, reduces the metrics there
and caches the teacher's trajectories across calls (the reference recomputes them for every
student although they do not depend on it, :156).

RNG contract: sample s uses seed 42+s; its start noise is the CPU-generator draw
right after ``manual_seed(42+s)`` and the step noise at timestep t the draw after ``manual_seed(42+s+t)``
-- i.e. row (s+t) of one table N[k] = randn(1,C,H,W | seed 42+k), which is uploaded once.
"""
import numpy as np
import torch

from .. import engine
from .._hip import COND_NONE, COND_ONE, COND_ZERO, RULE_ENGINE
from ..synthetic import noise_table
from ..utils.diffusion import get_diffusion_params


def extract(a, t, x_shape):
    """reference analysis/trajectory_engine.py:14-22 (duplicate of utils.diffusion.extract)."""
    t = torch.clamp(t, 0, a.shape[0] - 1)
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def engine_coefficients(timesteps):
    """[(c1, c2, sigma)] for t = 0..T-1 from the per-step alphas = 1 - betas (reference :49, :97-110),
    formed with the same fp32 torch ops; row 0 is unused (x is recorded unchanged at t == 0)."""
    alphas = 1.0 - get_diffusion_params(timesteps)["betas"].cpu()
    rows = [(1.0, 0.0, 0.0)]
    for t in range(1, timesteps):
        a_t, a_p = alphas[t], alphas[t - 1]
        c1 = torch.sqrt(a_p) / torch.sqrt(a_t)
        c2 = torch.sqrt(1 - a_p) - torch.sqrt(a_p / a_t) * torch.sqrt(1 - a_t)
        sigma = torch.sqrt(1 - a_p) * torch.sqrt(1 - a_t / a_p)
        rows.append((float(c1), float(c2), float(sigma)))
    return rows


def uses_cfg(guidance_scale):
    """The reference only takes the two-pass branch for guidance_scale > 1.0 (:65)."""
    return guidance_scale is not None and guidance_scale > 1.0


def _plan(timesteps, cfg):
    """(time-bias row spec, n_pass) for t = T-1..0: CFG rows are (cond=0, cond=1) per step (:73)."""
    ts = list(range(timesteps - 1, -1, -1))
    if cfg:
        return [t for t in ts for _ in (0, 1)], [COND_ZERO, COND_ONE] * timesteps, 2
    return ts, [COND_NONE] * timesteps, 1


def generate_trajectory(model, noise, timesteps, device, seed=None, guidance_scale=None):
    """T+1 CPU tensors [B,C,H,W]: entry 0 is ``noise``, entry T repeats entry T-1 (reference :24-115)."""
    model.eval()
    h = engine.UNetHandle.for_module(model)
    device = torch.device(device)
    B, C, H, W = noise.shape
    E = C * H * W
    coefs = engine_coefficients(timesteps)
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
    # step noise in the reference's order: one CPU-generator draw per t = T-1..1, re-seeded per step
    zs = []
    for t in range(timesteps - 1, 0, -1):
        if seed is not None:
            torch.manual_seed(seed + t)
            np.random.seed(seed + t)
        zs.append(torch.randn(B, C, H, W))
    traj = torch.empty(timesteps + 1, B, E, dtype=torch.float32, device=device)
    traj[0].copy_(noise.detach().reshape(B, E).float())
    z = torch.stack(zs).reshape(-1, E).to(device) if zs else None
    cfg = uses_cfg(guidance_scale)
    t_rows, modes, n_pass = _plan(timesteps, cfg)
    tb = h.time_bias(t_rows, modes)
    order = list(range(timesteps - 1, -1, -1))
    h.sample(RULE_ENGINE, traj, H, W, tb, n_pass, [coefs[t] for t in order], [t > 0 for t in order], z=z,
             z_shift=[i * B for i in range(timesteps)], w_scalar=float(guidance_scale) if cfg else 1.0)
    host = traj.cpu().reshape(timesteps + 1, B, C, H, W)
    return [host[i].clone() for i in range(timesteps + 1)]


# ------------------------------------------------------------------------------------ batched grid
_TEACHER_CACHE = {}
_ROW_CONSTS = {}


def _grid_rows(device, S, G, first_row, group, cfg):
    """(z_row int32 [G*S], w float32 [G*S] or None) on ``device`` for one CFG plan, uploaded once per distinct plan: a
    pageable host-to-device copy blocks the host until the stream has drained, so per-call uploads would keep a worker from
    queueing the next sampler loop behind the running one."""
    key = (str(device), S, G, first_row, tuple(group), cfg)
    hit = _ROW_CONSTS.get(key)
    if hit is None:
        if len(_ROW_CONSTS) > 64:
            _ROW_CONSTS.clear()
        z_row = (torch.arange(S, dtype=torch.int32).repeat(G) + first_row).to(device)
        w = torch.tensor([float(gs) for gs in group], dtype=torch.float32).repeat_interleave(S).to(device) if cfg else None
        hit = _ROW_CONSTS[key] = (z_row, w)
    return hit


class GridGroup(tuple):
    """(scales, traj, block_of): one launch sequence of ``sample_grid_groups``.  ``traj`` is [T+1, n_blocks*S, E];
    rows [b*S, (b+1)*S) hold the S samples of row block b, and scales[i] lives in block block_of[i]."""
    __slots__ = ()

    def __new__(cls, scales, traj, block_of):
        return tuple.__new__(cls, (list(scales), traj, list(block_of)))

    scales = property(lambda self: self[0])
    traj = property(lambda self: self[1])
    block_of = property(lambda self: self[2])


def _merge_plans():
    import os
    return os.environ.get("DT_GRID_MERGE", "1") != "0"


def sample_grid_groups(handle, table, first_row, num_samples, timesteps, guidance_scales, H, W, throttle=False):
    """Trajectories of one model for every guidance scale: a list of ``GridGroup`` (one per launch sequence).

    ``table`` is the device noise table [rows, E]; sample s starts from row first_row+s and takes
    row first_row+s+t at timestep t.  The scales that do not take the reference's two-pass branch (gs <= 1, None:
    trajectory_engine.py:65,83) all give the same trajectory and share ONE row block; every CFG scale has a row block
    of its own with a per-row guidance scale.  When both kinds are present they run as ONE mixed launch sequence
    (dt_sample_trajectory_mixed: the single-pass samples ride in the CFG samples' launches -- block 0 single-pass,
    blocks 1..G the CFG scales); DT_GRID_MERGE=0 keeps one sequence per kind (two GridGroups).

    ``throttle``: return only once every launch sequence but the last has finished on the device (the host waits on
    an event recorded between them).  The grid's worker threads use it so that a thread holds at most two sampler
    loops in its stream when it picks its next model: models are then handed out by GPU progress (dynamic load
    balance across the streams) instead of by how fast the host can queue launches.
    """
    S, T = num_samples, timesteps
    E = table.shape[1]
    coefs = engine_coefficients(T)
    order = list(range(T - 1, -1, -1))
    plain = [gs for gs in guidance_scales if not uses_cfg(gs)]
    guided = [gs for gs in guidance_scales if uses_cfg(gs)]
    dev = table.device
    groups, marks = [], []
    if plain and guided and _merge_plans():
        G = len(guided)
        key = (str(dev), "mixed", S, first_row, tuple(guided))
        hit = _ROW_CONSTS.get(key)
        if hit is None:
            z_row = (torch.arange(S, dtype=torch.int32).repeat(1 + G) + first_row).to(dev)
            w = torch.cat([torch.zeros(S), torch.tensor([float(gs) for gs in guided], dtype=torch.float32).repeat_interleave(S)]).to(dev)
            hit = _ROW_CONSTS[key] = (z_row, w)
        z_row, w = hit
        traj = torch.empty(T + 1, (1 + G) * S, E, dtype=torch.float32, device=dev)
        traj[0].copy_(table[first_row: first_row + S].repeat(1 + G, 1))
        # time-bias rows of a step, one per S rows of the forward: [cond None | cond 0 x G | cond 1 x G]
        modes = ([COND_NONE] + [COND_ZERO] * G + [COND_ONE] * G) * T
        t_rows = [t for t in order for _ in range(1 + 2 * G)]
        tb = handle.time_bias(t_rows, modes)
        handle.sample_mixed(RULE_ENGINE, traj, H, W, tb, S, S, [coefs[t] for t in order], [t > 0 for t in order],
                            z=table, z_row=z_row, z_shift=list(order), w=w)
        return [GridGroup(plain + guided, traj, [0] * len(plain) + list(range(1, 1 + G)))]
    for cfg, group in ((False, plain), (True, guided)):
        if not group:
            continue
        G = len(group) if cfg else 1            # without CFG every scale gives the same trajectory
        z_row, w = _grid_rows(dev, S, G, first_row, group, cfg)
        traj = torch.empty(T + 1, G * S, E, dtype=torch.float32, device=dev)
        traj[0].copy_(table[first_row: first_row + S].repeat(G, 1))
        t_rows, modes, n_pass = _plan(T, cfg)
        tb = handle.time_bias(t_rows, modes)
        handle.sample(RULE_ENGINE, traj, H, W, tb, n_pass, [coefs[t] for t in order], [t > 0 for t in order],
                      z=table, z_row=z_row, z_shift=list(order), w=w)
        groups.append(GridGroup(group, traj, list(range(G)) if cfg else [0] * len(group)))
        if throttle:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            marks.append(ev)
    for ev in marks[:-1]:
        ev.synchronize()
    return groups


def sample_grid(handle, table, first_row, num_samples, timesteps, guidance_scales, H, W):
    """{gs: device tensor [T+1, S, E]} over ``sample_grid_groups`` (views of the groups' tensors: not contiguous when a
    group holds several row blocks)."""
    S = num_samples
    out = {}
    for scales, traj, block_of in sample_grid_groups(handle, table, first_row, S, timesteps, guidance_scales, H, W):
        for gs, b in zip(scales, block_of):
            out[gs] = traj[:, b * S: (b + 1) * S]
    return out


def pair_metrics_device(X, Y, pixels, seeds=None):
    """Per-pair metric dicts for device trajectories X, Y [n, B, E] (B independent pairs)."""
    from .metrics.trajectory_metrics import wasserstein_index_tables
    n, B, E = X.shape
    Xc, Yc = X.contiguous(), Y.contiguous()
    index = index_row = None
    if E > 1000:
        tables, index_row = wasserstein_index_tables(seeds, n, E)
        index, index_row = tables.to(X.device), index_row.to(X.device)
    sums, w1 = engine.device_pair_metrics(Xc, Yc, index, index_row)      # one launch when all coordinates are sampled
    sums, w1 = sums.cpu().numpy(), w1.cpu().numpy()
    return [engine.metrics_from_sums(sums[b], w1[b], n, n, pixels, E) for b in range(B)]


def average_sample_trajectories(teacher_model, student_model, config, guidance_scales, num_samples, base_seed=42):
    """Per-guidance-scale trajectories averaged over samples, as scripts/analysis/analyze_trajectories.py
    :437-486 builds them before its PCA plots: sample i uses seed base_seed+i; returns
    (teacher, student) dicts {gs: [T+1 CPU tensors [1,C,H,W]]}.  One batched launch sequence per model
    and CFG plan, then a device mean over the sample axis (dt_traj_sample_mean)."""
    device = next(teacher_model.parameters()).device
    teacher_model.eval(); student_model.eval()
    C, H, T, S = config.channels, config.image_size, config.timesteps, num_samples
    state = torch.get_rng_state()
    table = noise_table(base_seed, S + T - 1, (1, C, H, H)).reshape(S + T - 1, -1).to(device)
    torch.set_rng_state(state)
    out = []
    for model in (teacher_model, student_model):
        grid = sample_grid(engine.UNetHandle.for_module(model), table, 0, S, T, list(guidance_scales), H, H)
        avg = {}
        for gs in guidance_scales:
            mean = engine.device_sample_mean(grid[gs].contiguous()).cpu().reshape(T + 1, 1, C, H, H)
            avg[gs] = [mean[i].clone() for i in range(T + 1)]
        out.append(avg)
    # leave the global generators as the reference's last generate_trajectory call does
    last = base_seed + S - 1
    torch.manual_seed(last + 1 if T > 1 else last)
    np.random.seed(last + 1 if T > 1 else last)
    if T > 1:
        torch.randn(1, C, H, H)
    return out[0], out[1]


def compare_trajectories(teacher_model, student_model, config, guidance_scales=[1.0, 3.0, 5.0], size_factor=1.0,
                         num_samples=3):
    """Teacher-vs-student metric averages per guidance scale (reference :117-180).

    Returns {'teacher_metrics': {gs: {key: float}}, 'student_metrics': {...}}; both hold the same
    numbers, as in the reference (:163-164).
    """
    device = next(teacher_model.parameters()).device
    teacher_model.eval(); student_model.eval()
    th = engine.UNetHandle.for_module(teacher_model)
    sh = engine.UNetHandle.for_module(student_model)
    C, H, T, S = config.channels, config.image_size, config.timesteps, num_samples
    scales = list(guidance_scales)
    rng_state = torch.get_rng_state()
    table = noise_table(42, S + T - 1, (1, C, H, H)).reshape(S + T - 1, -1).to(device)
    torch.set_rng_state(rng_state)

    key = (id(th), T, S, C, H, tuple(scales))
    teacher = _TEACHER_CACHE.get(key)
    if teacher is None:
        if len(_TEACHER_CACHE) >= 2:
            _TEACHER_CACHE.clear()
        teacher = _TEACHER_CACHE[key] = (th, sample_grid(th, table, 0, S, T, scales, H, H))
    student = sample_grid(sh, table, 0, S, T, scales, H, H)

    per_gs = {}
    for gs in scales:
        per_gs[gs] = pair_metrics_device(teacher[1][gs], student[gs], H * H, seeds=[42 + s for s in range(S)])

    # leave the global generators where the reference's last generate_trajectory + metrics leave them
    last = 42 + S - 1
    torch.manual_seed(last + 1 if T > 1 else last)
    np.random.seed(last + 1 if T > 1 else last)
    if T > 1:
        torch.randn(1, C, H, H)
    E = C * H * H
    for _ in range(T + 1):
        np.random.choice(E, min(1000, E), replace=False)

    avg = {gs: {} for gs in scales}
    for gs in scales:
        runs = per_gs[gs]
        for k, v in runs[0].items():
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                avg[gs][k] = sum(m[k] for m in runs) / len(runs)
    return {"teacher_metrics": avg, "student_metrics": {gs: dict(v) for gs, v in avg.items()}}
