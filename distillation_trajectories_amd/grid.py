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


def all_gather_rows(local, counts, dim, force=False):
    """Concatenate every rank's ``local`` along ``dim`` (row counts may differ by one). No-op without a
    process group.  Uses the default group's backend: nccl (= RCCL) on GPUs, gloo in the CPU tests.
    ``force``: run the collective also in a group of ONE rank (bench.py DT_BENCH_NCCL1=1: the RCCL code path on a
    single-GPU box)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
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


_STREAMS = {}


def _side_streams(device, n):
    """n HIP streams of ``device``, created once (workspaces are cached per (handle, shape, stream))."""
    have = _STREAMS.setdefault(str(device), [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


def _upload(host):
    """Host tensor -> current device through pinned memory, asynchronously on the current stream."""
    return host.pin_memory().to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)


def hip_cell_metrics(teacher, students, cfg, guidance_scales, first_sample, count, device, table=None, streams=None):
    """float64 tensor [n_sf, n_gs, count, K] on ``device`` for samples first_sample .. +count-1 (seed 42+s).

    The teacher runs once per CFG plan on the current stream, alone (see the note at its launch).  The students are
    independent of each other, so they then share the chip on ``streams`` side streams (default DT_GRID_STREAMS or 3),
    one host thread each, assigned longest-first by a static cost model: the small students, whose launches cannot fill
    256 CUs, overlap the bigger ones.  Every device result stays in HBM until ONE device-to-host copy at the end; the scalar
    post-transforms then run once, vectorised over every (student, scale, sample) row.
    ``table``: the device noise table [count+T-1, E] when the caller already holds it (bench.py).
    """
    import os
    import threading
    from .analysis.metrics.trajectory_metrics import wasserstein_index_tables
    from .analysis.trajectory_engine import sample_grid_groups
    C, H, T = cfg.channels, cfg.image_size, cfg.timesteps
    scales = list(guidance_scales)
    n = T + 1
    with torch.cuda.device(device):
        if table is None:
            table = _upload(noise_table(42 + first_sample, count + T - 1, (1, C, H, H)).reshape(count + T - 1, -1))
        E = table.shape[1]
        index = index_row = None
        if E > 1000:
            tables, rows = wasserstein_index_tables([42 + first_sample + s for s in range(count)], n, E)
            index, index_row = _upload(tables), rows
        handles = [engine.UNetHandle.for_module(m) for m in [teacher] + list(students)]
        from .analysis.trajectory_engine import _merge_plans, uses_cfg
        n_cfg = sum(1 for gs in scales if uses_cfg(gs))
        merged = n_cfg and n_cfg < len(scales) and _merge_plans()
        plans = [] if merged else [(1, count)] + ([(2, count * n_cfg)] if n_cfg else [])
        for h in handles:             # settle every launch plan (table lookup, or timing with DT_AUTOTUNE=1) before streams share the chip
            if merged and ((1 + 2 * n_cfg) * count, H, H, (1 + n_cfg) * count, count) not in h._plans:
                B = (1 + n_cfg) * count
                tb = h.time_bias([T - 1] * (1 + 2 * n_cfg), [0] * (1 + 2 * n_cfg))
                x = table[torch.arange(B, device=device) % table.shape[0]]            # real noise: zeros clock differently
                h.forward_mixed(x.reshape(B, C, H, H), tb, count, count)
            for n_pass, B in plans:
                if B and (n_pass * B, H, H, B, 0) not in h._plans:
                    tb = h.time_bias([T - 1] * n_pass, [0] * n_pass)
                    x = table[torch.arange(B, device=device) % table.shape[0]]        # real noise: zeros clock differently
                    h.forward(x.reshape(B, C, H, H), tb, n_pass, B)
        main = torch.cuda.current_stream()
        n_streams = max(1, min(len(students), streams or int(os.environ.get("DT_GRID_STREAMS", "3"))))
        side = _side_streams(device, n_streams)          # kept across calls: a handle keeps one workspace per stream
        results = [None] * len(students)
        errors = []
        # Static longest-first assignment of the students to the side streams (cost model from the per-model timings in
        # tools/config2_breakdown.py: ~10 ms of launch-bound work per loop + time proportional to the parameter count).
        # Static, because a handle keeps one workspace per stream it has run on: with models handed out by arrival the
        # (model, stream) pairs changed from step to step and every new pair allocated a workspace mid-step (a 20 ms
        # stall of all streams in the kernel trace).
        cost = [10.0 + 4.5e-6 * sum(p.numel() for p in m.parameters()) for m in students]
        queues, load = [[] for _ in range(n_streams)], [0.0] * n_streams
        for i in sorted(range(len(students)), key=lambda i: -cost[i]):
            k = min(range(n_streams), key=lambda q: load[q])
            queues[k].append(i)
            load[k] += cost[i]
        # The teacher runs FIRST, alone: measured on MI355X (configs[2], same box, alternating runs) its loop sharing the
        # chip with the students' small launches costs 13 % of the step (469-482 vs 418-428 ms) -- its big tiles are tuned
        # for an idle chip and lose more to the interference than the students gain from the overlap.  The students then
        # share the chip among themselves on the side streams.
        t_groups = sample_grid_groups(handles[0], table, 0, count, T, scales, H, H)
        # Wasserstein coordinate table of every row (E > 1000 only): the sample's table, repeated per row block
        row_sets = [None if index_row is None else index_row.repeat(g.traj.shape[1] // count).to(device) for g in t_groups]
        ready = torch.cuda.Event()       # the teacher's trajectories (and every upload above) are complete
        ready.record(main)

        def worker(k):
            try:
                with torch.cuda.device(device), torch.cuda.stream(side[k]):
                    side[k].wait_event(ready)
                    for i in queues[k]:
                        s_groups = sample_grid_groups(handles[1 + i], table, 0, count, T, scales, H, H, throttle=len(side) > 1)
                        parts = []
                        for (_, X, _), (_, Y, _), rows in zip(t_groups, s_groups, row_sets):
                            sums, w1 = engine.device_pair_metrics(X, Y, index, rows)     # [G*S, n, 4], [G*S, n]: one launch at E <= 1000
                            parts.append(torch.cat([sums.reshape(X.shape[1], -1), w1], dim=1))
                        results[i] = torch.cat(parts, dim=0)
            except Exception as e:       # surfaced on the calling thread
                errors.append(e)
        if streams == 1:                 # serial mode (per-kernel profiling): everything on the caller's stream, in order
            side = [main]
            queues = [sorted(range(len(students)), key=lambda i: -cost[i])]
            worker(0)
            threads = []
        else:
            threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        for st in side:
            if st is not main:
                main.wait_stream(st)
        host = torch.stack(results).cpu().numpy()                   # the one D2H: [n_sf, rows, 5n] float64
    n_sf, n_rows = host.shape[0], host.shape[1]
    flat = host.reshape(n_sf * n_rows, 5 * n)
    vals = engine.batch_scalar_metrics(flat[:, : 4 * n].reshape(-1, n, 4), flat[:, 4 * n:], H * H, E)
    per_row = np.stack([vals[k] for k in engine.SCALAR_KEYS], axis=1).reshape(n_sf, n_rows, K)
    out = np.empty((n_sf, len(scales), count, K))
    base = 0
    for group, X, block_of in t_groups:
        for gs, blk in zip(group, block_of):
            lo = base + blk * count
            for j, s in enumerate(scales):
                if s == gs or (s is None and gs is None):
                    out[:, j] = per_row[:, lo: lo + count]
        base += X.shape[1]
    return torch.from_numpy(out).to(device)


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
