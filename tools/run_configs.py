#!/usr/bin/env python3
"""Run every BASELINE.json config once on one MI355X and print one JSON line per config (throughput in
trajectory-timesteps/s, API-level = including host RNG, H2D and D2H where the reference API implies them).
Not the benchmark of record (that is bench.py); this documents that every config runs end to end."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # noqa: E402

from distillation_trajectories_amd.analysis.trajectory_engine import compare_trajectories   # noqa: E402
from distillation_trajectories_amd.config import Config   # noqa: E402
from distillation_trajectories_amd.grid import grid_metrics   # noqa: E402
from distillation_trajectories_amd.models import DiffusionUNet   # noqa: E402
from distillation_trajectories_amd.synthetic import make_model   # noqa: E402
from distillation_trajectories_amd.utils.diffusion import get_diffusion_params, p_sample_loop   # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def model(cfg, sf):
    return make_model(DiffusionUNet, cfg, sf).to(DEV)


def main():
    which = set(sys.argv[1:]) or {"0", "1", "2", "3", "4"}
    cfg = Config()
    cfg.image_size, cfg.timesteps, cfg.sample_steps = 16, 50, 50
    teacher = model(cfg, 1.0)
    if "0" in which:     # configs[0]: teacher, 16x16, T=50, batch 8 (reference's CPU-runnable case; API level)
        f = lambda: p_sample_loop(teacher, (8, 3, 16, 16), 50, get_diffusion_params(50, cfg), device=DEV, config=cfg,
                                  track_trajectory=True, guidance_scale=1.0)
        timed(f)
        (_, tr), dt = timed(f)
        print(json.dumps({"config": 0, "what": "p_sample_loop API teacher B=8 T=50 (host noise + D2H included)",
                          "traj_steps_per_s": round(8 * 50 / dt, 1), "seconds": round(dt, 4), "entries": len(tr)}), flush=True)
    if "1" in which:     # configs[1] at API level: p_sample_loop B=256 for teacher and student incl. host noise + D2H
        student = model(cfg, 0.5)
        def f():
            out = []
            for m in (teacher, student):
                torch.manual_seed(1234)
                out.append(p_sample_loop(m, (256, 3, 16, 16), 50, get_diffusion_params(50, cfg), device=DEV, config=cfg,
                                         track_trajectory=True, guidance_scale=1.0))
            return out
        timed(f)
        _, dt = timed(f)
        print(json.dumps({"config": 1, "what": "p_sample_loop API teacher+student B=256 T=50, PCIe-inclusive "
                          "(CPU-generator noise drawn + uploaded, 51 CPU tensors returned per model)",
                          "traj_steps_per_s": round(2 * 256 * 50 / dt, 1), "seconds": round(dt, 4)}), flush=True)
    if "2" in which or "3" in which:   # configs[2]/[3]: size sweep x CFG grid, sample axis as the batch
        sizes = [0.01, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]
        students = [model(cfg, sf) for sf in sizes]
        if "2" in which:
            scales, S = [1.0, 3.0, 7.0, 20.0], 64
            f = lambda: grid_metrics(teacher, students, cfg, scales, S)
            timed(lambda: grid_metrics(teacher, students[:1], cfg, scales, S))
            res, dt = timed(f)
            units = (len(sizes) + 1) * len(scales) * S * 50
            print(json.dumps({"config": 2, "what": f"11 sizes x CFG {scales} x {S} samples, T=50 (teacher once per CFG plan)",
                              "traj_steps_per_s": round(units / dt, 1), "seconds": round(dt, 3),
                              "path_length_similarity_sf0.5_gs3": res[5][3.0]["path_length_similarity"]}), flush=True)
        if "3" in which:
            scales, S = [1.0, 2.0, 3.0, 5.0, 7.5, 10.0, 15.0, 20.0], 64     # one GPU's shard of the 512-sample grid
            f = lambda: grid_metrics(teacher, students, cfg, scales, S)
            res, dt = timed(f)
            units = (len(sizes) + 1) * len(scales) * S * 50
            print(json.dumps({"config": 3, "what": f"one rank's shard (64 of 512 samples) of 11 sizes x 8 CFG scales, T=50",
                              "traj_steps_per_s": round(units / dt, 1), "seconds": round(dt, 3)}), flush=True)
    if "4" in which:     # configs[4]: CIFAR shape 32x32, T=1000, teacher + sf=1.0 student, batch 128, CFG 7
        c5 = Config()
        c5.image_size, c5.timesteps, c5.sample_steps = 32, 1000, 1000
        student = model(c5, 1.0)
        from distillation_trajectories_amd import engine
        from distillation_trajectories_amd._hip import COND_NONE, COND_ONE, RULE_PSAMPLE
        from distillation_trajectories_amd.utils.diffusion import psample_coefficients, timestep_indices
        idx = timestep_indices(1000, 1000)
        coef = psample_coefficients(get_diffusion_params(1000, c5), idx)
        B, E = 128, 3072
        g = torch.Generator().manual_seed(3)
        x_T = torch.randn(B, E, generator=g).to(DEV)
        z = torch.randn(64 * B, E, generator=g).to(DEV)      # 64 distinct noise slabs, cycled (device-resident run)
        trajs = []
        def f():
            trajs.clear()
            for m in (teacher, student):
                h = engine.UNetHandle.for_module(m)
                traj = torch.empty(1001, B, E, device=DEV)
                traj[0].copy_(x_T)
                tb = h.time_bias([i for i in idx for _ in (0, 1)], [COND_NONE, COND_ONE] * 1000)
                h.sample(RULE_PSAMPLE, traj, 32, 32, tb, 2, coef, [i > 0 for i in idx], z=z,
                         z_shift=[(s % 64) * B for s in range(1000)], w_scalar=7.0)
                trajs.append(traj)
            return engine.device_metric_sums(trajs[0], trajs[1])
        (sums), dt = timed(f)
        print(json.dumps({"config": 4, "what": "32x32x3, T=1000, teacher + sf=1.0 student, batch 128, CFG 7, device resident "
                          "(FID-input dump = the [1001,128,3072] trajectory tensors), + metric sums",
                          "traj_steps_per_s": round(2 * B * 1000 / dt, 1), "seconds": round(dt, 3),
                          "finite": bool(torch.isfinite(sums).all())}), flush=True)


if __name__ == "__main__":
    main()
