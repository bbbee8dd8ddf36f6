"""The N > 1 path on CPU: two gloo ranks shard the sample axis, exchange one all-gather of per-sample
metric rows and must reproduce the single-process averages exactly (sample order preserved).
The per-shard compute is injected (oracle-backed, tiny model) because this container has no GPU; on
the GPU box the same driver runs with the HIP cell function and the nccl (= RCCL) backend."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from distillation_trajectories_amd import engine
from distillation_trajectories_amd.grid import all_gather_rows, grid_metrics, shard_range

GS = [1.0, 3.0]
SAMPLES, T = 5, 6
K = len(engine.SCALAR_KEYS)


def oracle_cell_fn():
    """cell_fn(first, count) -> [1 student, len(GS), count, K] from the CPU oracle (sf 0.2 vs 0.01, 16x16, T=6)."""
    from distillation_trajectories_amd.config import Config
    from distillation_trajectories_amd.models import DiffusionUNet
    from distillation_trajectories_amd.synthetic import make_model, seeded_noise
    from oracle import metrics_ref, sampler_ref, unet_ref
    cfg = Config()
    cfg.image_size = 16
    sds = [make_model(DiffusionUNet, cfg, sf).state_dict() for sf in (0.2, 0.01)]
    fns = [lambda x, t, c, sd=sd: unet_ref.unet_forward(sd, x, t, c) for sd in sds]

    def cell(first, count):
        out = torch.zeros(1, len(GS), count, K, dtype=torch.float64)
        with torch.no_grad():
            for s in range(count):
                seed = 42 + first + s
                noise = seeded_noise(seed, (1, 3, 16, 16))
                for j, gs in enumerate(GS):
                    a = sampler_ref.generate_trajectory(fns[0], noise, T, seed=seed, guidance_scale=gs)
                    b = sampler_ref.generate_trajectory(fns[1], noise, T, seed=seed, guidance_scale=gs)
                    m = metrics_ref.compute_trajectory_metrics(a, b)
                    out[0, j, s] = torch.tensor([float(m[k]) for k in engine.SCALAR_KEYS], dtype=torch.float64)
        return out
    return cell


# configs[3] shape: 11 size factors x 8 guidance scales (scripts/analysis/analyze_trajectory_metrics.py:38-42), 13 samples
# (ragged over two ranks: 7 + 6).  The cell values are a synthetic function of (student, scale, sample, metric) -- this leg
# checks the shard / all-gather / sample-order-mean plumbing at the grid's real extent, the oracle-backed leg checks values.
GS8 = [1.0, 2.0, 3.0, 5.0, 7.5, 10.0, 15.0, 20.0]
N_SF, SAMPLES8 = 11, 13


def synthetic_cell_fn(first, count):
    i, j, s, k = np.meshgrid(np.arange(N_SF), np.arange(len(GS8)), first + np.arange(count), np.arange(K), indexing="ij")
    v = np.sin(0.37 * i + 1.3 * j + 0.11 * k) * (1.0 + 0.01 * s) + 1e-3 * s * s
    v[:, :, :, 2] = np.where((s[:, :, :, 2] + j[:, :, :, 2]) % 5 == 0, np.nan, v[:, :, :, 2])     # trajectory_mse goes NaN at high guidance
    return torch.from_numpy(v)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res8 = grid_metrics(None, [None] * N_SF, None, GS8, SAMPLES8, rank=rank, world=world, cell_fn=synthetic_cell_fn)
        q.put((rank, "grid8", res8, None))
        res = grid_metrics(None, [None], None, GS, SAMPLES, rank=rank, world=world, cell_fn=oracle_cell_fn())
        # ragged all-gather on its own: rank r contributes r+2 rows
        rows = torch.full((rank + 2, 3), float(rank), dtype=torch.float64)
        cat = all_gather_rows(rows, [r + 2 for r in range(world)], dim=0)
        q.put((rank, "cells", res, cat.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_grid_matches_single_process():
    threads = torch.get_num_threads()
    torch.set_num_threads(1)          # same reduction order as the single-threaded workers
    try:
        single = grid_metrics(None, [None], None, GS, SAMPLES, rank=0, world=1, cell_fn=oracle_cell_fn())
    finally:
        torch.set_num_threads(threads)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    single8 = grid_metrics(None, [None] * N_SF, None, GS8, SAMPLES8, rank=0, world=1, cell_fn=synthetic_cell_fn)
    got = [q.get() for _ in range(2 * len(procs))]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, kind, res8, _ in (g for g in got if g[1] == "grid8"):
        assert len(res8) == N_SF and all(set(cell) == set(GS8) for cell in res8)
        for i in range(N_SF):
            for gs in GS8:
                for k in engine.SCALAR_KEYS:
                    a, b = res8[i][gs][k], single8[i][gs][k]
                    assert a == b or (np.isnan(a) and np.isnan(b)), (rank, i, gs, k, a, b)
    for rank, kind, res, cat in (g for g in got if g[1] == "cells"):
        assert cat == [[0.0] * 3] * 2 + [[1.0] * 3] * 3
        for gs in GS:
            for k in engine.SCALAR_KEYS:
                a, b = res[0][gs][k], single[0][gs][k]
                assert a == b or (np.isnan(a) and np.isnan(b)), (rank, gs, k, a, b)   # identical, not just close


def test_sample_shards_are_the_reference_sample_order():
    spans = [shard_range(SAMPLES, r, 2) for r in range(2)]
    assert spans == [(0, 3), (3, 2)]
