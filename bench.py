#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: trajectory-timesteps/sec (BASELINE.json metric).

Default workload (BASELINE.json configs[1]): teacher (size_factor 1.0) vs student (size_factor 0.5), 16x16x3,
T=50, batch 256, guidance scale 1.0 through the CFG sampler of utils/diffusion.py (p_sample_loop: two
U-Net passes per step, cond=ones and cond=None, then the DDPM update), followed by the trajectory
metrics of the 256 (teacher, student) pairs.  One "step" of this benchmark is one such pass:
2 models x 256 samples x 50 timesteps = 25,600 trajectory-timesteps per GPU.  The time/condition embedding
tower of every (timestep, cond) row is evaluated inside the step, like every reference forward does.

``--config 4`` runs the same pass at configs[4]'s shape (32x32x3, T=1000, batch 128, CFG 7, teacher + a second
size-factor-1.0 model); ``--config 2`` runs configs[2] (11 size factors x CFG {1,3,7,20} x 64 samples per GPU
through the grid driver, teacher once per CFG plan).

Inputs (start noise x_T and the per-step Gaussian noise, shared by teacher and student) are synthetic,
seeded, and resident in HBM before the timed region.

``--gpus N`` (N > 1) without a launcher: this process spawns N rank processes (one per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in their environment) BEFORE touching the GPU, relays rank 0's JSON line and exits with the
ranks' status.  Under ``torch.distributed.run`` (WORLD_SIZE already set) it is one of the ranks.  Samples are
sharded across ranks, no data-path collective; the only exchange is one RCCL all-gather of the per-sample metric
tensor at the end of each step (weak scaling: the same batch per GPU).  ``DT_BENCH_BACKEND=gloo`` lets N ranks
share fewer GPUs to rehearse that path (tools/rehearse_ranks.sh).

Prints ONE JSON line on rank 0 (contract in the task description), with a ``roofline`` object for the
dominant kernel (HIP-event timed inside this process), a ``cpu_baseline`` object (the oracle's restatement of the
same loop timed on the host cores, rank 0, N=1 only) and ``metric_delta_vs_cpu`` (GPU metrics of the first pairs
against the oracle's full-length run on the same inputs).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense matrix peaks.  The split-bf16 kernels evaluate every fp32
# product as 6 bf16 plane products, so their fp32-equivalent ceiling is the dense bf16 peak / 6.
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PLANE_PRODUCTS = 6
C = 3
SIZES = [0.01, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]          # configs[2] size sweep (README.md:23)
CONFIGS = {
    1: dict(name="configs[1]", H=16, T=50, batch=256, guidance=1.0, sf=(1.0, 0.5), seeds=(None, None)),
    4: dict(name="configs[4]", H=32, T=1000, batch=128, guidance=7.0, sf=(1.0, 1.0), seeds=(None, 4242)),
    2: dict(name="configs[2]", H=16, T=50, batch=64, scales=[1.0, 3.0, 7.0, 20.0], sizes=SIZES),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="BASELINE.json configs[] index")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the config's batch)")
    ap.add_argument("--serial", action="store_true", help="run the teacher and student loops back to back on one stream")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------- rank launcher
def launch_ranks(args):
    """Parent of a ``--gpus N`` run: N children of this script, one per GPU.  Nothing here (or imported so far)
    initialises HIP -- the parent never touches the GPU and never execs."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()))
    reader.start()
    code = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and code == 0:
                code = rc
                sys.stderr.write(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks\n")
                for q in pending:
                    procs[q].terminate()          # exactly the PIDs started above
        time.sleep(0.05)
    reader.join()
    lines = [ln for ln in out0 if ln.strip().startswith("{")]
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    elif code == 0:
        code = 1
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
    return code


# ------------------------------------------------------------------------------------------- workloads
class PairWorkload:
    """Device-resident state of one rank's share of a (teacher, student) p_sample_loop + metrics pass."""

    def __init__(self, spec, device, rank, batch, concurrent=True):
        import torch
        from distillation_trajectories_amd import engine
        from distillation_trajectories_amd.config import Config
        from distillation_trajectories_amd.models import DiffusionUNet
        from distillation_trajectories_amd.synthetic import make_model
        from distillation_trajectories_amd.utils.diffusion import (get_diffusion_params, psample_coefficients,
                                                                   timestep_indices)
        self.torch, self.engine, self.device, self.B, self.spec = torch, engine, device, batch, spec
        self.collective = False
        H, T = spec["H"], spec["T"]
        self.H, self.T, self.E = H, T, C * H * H
        cfg = Config()
        cfg.image_size, cfg.timesteps, cfg.sample_steps = H, T, T
        self.models = [make_model(DiffusionUNet, cfg, sf, seed=sd).to(device) for sf, sd in zip(spec["sf"], spec["seeds"])]
        self.handles = [engine.UNetHandle.for_module(m) for m in self.models]
        # Samples are independent, so a model's batch may run as several contiguous sub-batches, each with its own
        # handle (workspace), HIP stream and host thread: DT_BENCH_PARTS="teacher_parts,student_parts".  The default
        # stays 1,1: the teacher as two halves buys a few % of wall time by overlap but halves every teacher launch,
        # so the per-launch (roofline) figures then describe smaller, less efficient launches.
        parts = [int(v) for v in os.environ.get("DT_BENCH_PARTS", "1,1").split(",")]
        self.parts = []
        for i, n in enumerate(parts):
            n = max(1, min(n, batch))
            bounds = [(batch * k // n, batch * (k + 1) // n) for k in range(n)]
            hs = [self.handles[i]] + [engine.UNetHandle(self.models[i].state_dict(), device) for _ in range(n - 1)]
            self.parts.append(list(zip(hs, bounds)))
        idx = timestep_indices(T, T)
        self.coef = psample_coefficients(get_diffusion_params(T, cfg), idx)
        self.has_noise = [i > 0 for i in idx]
        g = torch.Generator().manual_seed(1234 + rank)          # torch CPU generator, like the reference's CPU mode
        x_T = torch.randn(batch, self.E, generator=g)
        n_z = sum(self.has_noise)
        z = torch.empty(n_z * batch, self.E)
        for s in range(n_z):                                     # slab by slab: bounded host memory at T=1000
            z[s * batch:(s + 1) * batch] = torch.randn(batch, self.E, generator=g)
        keep = min(2, batch)                                     # host copies of the first pairs' inputs (CPU legs)
        self.host_inputs = (x_T[:keep].clone(), z.reshape(n_z, batch, self.E)[:, :keep].clone(), idx)
        self.x_T, self.z = x_T.to(device), z.to(device)
        del z
        self.z_shift, k = [], 0
        for flag in self.has_noise:
            self.z_shift.append(k * batch)
            k += int(flag)
        # (t, cond) rows of the loop: pass 0 = cond None, pass 1 = cond ones (utils/diffusion.py:122-123); the
        # embedding tower itself runs inside step()
        self.tb_t = torch.tensor([i for i in idx for _ in (0, 1)], dtype=torch.int32).to(device)
        self.tb_cond = torch.tensor([0.0, 1.0] * T, dtype=torch.float32).to(device)
        self.tb_present = torch.tensor([0, 1] * T, dtype=torch.uint8).to(device)
        self.traj = [torch.empty(T + 1, batch, self.E, device=device) for _ in self.handles]
        self.part_traj = [[self.traj[i] if len(ps) == 1 else torch.empty(T + 1, hi - lo, self.E, device=device)
                           for (_, (lo, hi)) in ps] for i, ps in enumerate(self.parts)]
        # Wasserstein sub-sampling tables when E > 1000 (trajectory_metrics.py:295-315; seed-dependent only)
        self.w_index = self.w_rows = None
        if self.E > 1000:
            from distillation_trajectories_amd.analysis.metrics.trajectory_metrics import wasserstein_index_tables
            tables, rows = wasserstein_index_tables([42 + rank * batch + s for s in range(batch)], T + 1, self.E)
            self.w_index, self.w_rows = tables.to(device), rows.to(device)
        for i, ps in enumerate(self.parts):
            for h, (lo, hi) in ps:
                # the launch plan of the shape is settled here (committed table; timing only with DT_AUTOTUNE=1), on an idle GPU
                tb = h.time_bias_general(self.tb_t[:2], self.tb_cond[:2], self.tb_present[:2], 2)
                h.forward(self.x_T[lo:hi].reshape(hi - lo, C, H, H), tb, 2, hi - lo)
        torch.cuda.synchronize()
        self.choices = None
        self.concurrent = concurrent
        # DT_BENCH_PRIO=t,s: HIP stream priorities of the teacher / student loops (lower = more urgent; default equal)
        prio = [int(v) for v in os.environ.get("DT_BENCH_PRIO", "0,0").split(",")]
        self.streams = [[torch.cuda.Stream(device=device, priority=prio[i]) for _ in ps] for i, ps in enumerate(self.parts)]
        self.units_per_step = 2 * batch * T

    def describe(self, world):
        s = self.spec
        return {"workload": f"{s['name']}: teacher sf={s['sf'][0]} vs student sf={s['sf'][1]}, {self.H}x{self.H}x{C}, "
                            f"T={self.T}, batch {self.B}/GPU, p_sample_loop CFG (2 U-Net passes/step, w={s['guidance']}) "
                            f"+ time/cond embedding rows + trajectory metrics of the {self.B} pairs",
                "batch_per_gpu": self.B, "timesteps": self.T, "image": [C, self.H, self.H],
                "guidance_scale": s["guidance"], "unet_passes_per_step": 2,
                # both passes are computed every step; enc1 (whose input x the passes share) runs once and emits both passes'
                # outputs from one accumulator tile unless DT_NO_SHARED_ENC1=1
                "enc1_shared_by_cfg_passes": os.environ.get("DT_NO_SHARED_ENC1") is None,
                "parallelism": f"sample-sharded x{world}, RCCL all-gather of metrics",
                "streams_per_gpu": {"teacher_sub_batches": len(self.parts[0]), "student_sub_batches": len(self.parts[1])}}

    def step(self, world, counts):
        """One pass of the hot path: embedding rows, both samplers, the metric reductions, the metric all-gather."""
        from distillation_trajectories_amd._hip import RULE_PSAMPLE
        from distillation_trajectories_amd.grid import all_gather_rows
        torch, eng, H, T = self.torch, self.engine, self.H, self.T

        def run(i, k=0):
            (h, (lo, hi)), traj = self.parts[i][k], self.part_traj[i][k]
            tb = h.time_bias_general(self.tb_t, self.tb_cond, self.tb_present, 2 * T)     # A1, every (t, cond) row
            traj[0].copy_(self.x_T[lo:hi])
            h.sample(RULE_PSAMPLE, traj, H, H, tb, 2, self.coef, self.has_noise, z=self.z,
                     z_shift=[v + lo for v in self.z_shift], w_scalar=self.spec["guidance"])
        if self.concurrent:
            # teacher and student loops are independent: one HIP stream + one host thread each (the C call
            # releases the GIL), so the small spatial levels of one model overlap the other's work
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)

            def worker(i, k):
                with torch.cuda.device(self.device), torch.cuda.stream(self.streams[i][k]):
                    self.streams[i][k].wait_event(ready)
                    run(i, k)
            threads = [threading.Thread(target=worker, args=(i, k)) for i, ps in enumerate(self.parts) for k in range(len(ps))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            for sts in self.streams:
                for st in sts:
                    main.wait_stream(st)
        else:
            for i, ps in enumerate(self.parts):
                for k in range(len(ps)):
                    run(i, k)
        for i, ps in enumerate(self.parts):
            if len(ps) > 1:                      # reassemble [T+1, B, E] for the pairwise metrics
                torch.cat(self.part_traj[i], dim=1, out=self.traj[i])
        t_gpu0 = time.perf_counter() if os.environ.get("DT_BENCH_TIMING") else 0.0
        # [B, T+1, 4] and [B, T+1] float64: one launch (dt_traj_pair_metrics) when all E <= 1000 coordinates are sampled
        sums, w1 = eng.device_pair_metrics(self.traj[0], self.traj[1], self.w_index, self.w_rows)
        local = torch.cat([sums.reshape(self.B, -1), w1], dim=1)
        full = all_gather_rows(local, counts, dim=0, force=True) if (world > 1 or self.collective) else local
        host = full.cpu().numpy()                                          # syncs the stream
        n = T + 1
        t_host0 = time.perf_counter() if t_gpu0 else 0.0
        vals = eng.batch_scalar_metrics(host[:, : 4 * n].reshape(-1, n, 4), host[:, 4 * n:], H * H, self.E)
        if t_gpu0:
            sys.stderr.write(f"[bench timing] launch threads joined -> results on host {1e3 * (t_host0 - t_gpu0):.2f} ms, "
                             f"host post-transform {1e3 * (time.perf_counter() - t_host0):.2f} ms\n")
        if self.choices is None:
            self.choices = {f"sf={sf}": [list(c) for c in ps[0][0].conv_choices(2 * (ps[0][1][1] - ps[0][1][0]), H, H)]
                            for sf, ps in zip(self.spec["sf"], self.parts)}
        return vals

    def check(self, vals):
        import numpy as np
        return {"mean_endpoint_distance": float(np.mean(vals["endpoint_distance"])),
                "mean_wasserstein": float(np.mean(vals["mean_wasserstein"])), "pairs": int(len(vals["mse"]))}


class GridWorkload:
    """configs[2]: (size factor x guidance scale x sample) grid on one rank's sample shard (grid.hip_cell_metrics)."""

    def __init__(self, spec, device, rank, batch, concurrent=True):
        import torch
        from distillation_trajectories_amd.config import Config
        from distillation_trajectories_amd.models import DiffusionUNet
        from distillation_trajectories_amd.synthetic import make_model, noise_table
        self.torch, self.device, self.B, self.spec, self.rank = torch, device, batch, spec, rank
        self.collective = False
        H, T = spec["H"], spec["T"]
        self.H, self.T = H, T
        self.cfg = Config()
        self.cfg.image_size, self.cfg.timesteps, self.cfg.sample_steps = H, T, T
        self.teacher = make_model(DiffusionUNet, self.cfg, 1.0).to(device)
        self.students = [make_model(DiffusionUNet, self.cfg, sf).to(device) for sf in spec["sizes"]]
        first = rank * batch
        self.table = noise_table(42 + first, batch + T - 1, (1, C, H, H)).reshape(batch + T - 1, -1).to(device)
        self.first = first
        self.choices = None
        self.concurrent = concurrent
        self.units_per_step = (len(spec["sizes"]) + 1) * len(spec["scales"]) * batch * T
        self.parts = [[None], [None]]

    def describe(self, world):
        s = self.spec
        return {"workload": f"{s['name']}: {len(s['sizes'])} size factors {s['sizes']} x CFG {s['scales']} x {self.B} samples/GPU, "
                            f"{self.H}x{self.H}x{C}, T={self.T}, generate_trajectory rule (CFG concat for gs > 1), teacher once per "
                            "CFG plan, metrics of every (student, gs, sample) cell",
                "batch_per_gpu": self.B, "timesteps": self.T, "image": [C, self.H, self.H], "guidance_scales": s["scales"],
                "parallelism": f"sample-sharded x{world}, RCCL all-gather of metrics"}

    def step(self, world, counts):
        from distillation_trajectories_amd import grid
        local = grid.hip_cell_metrics(self.teacher, self.students, self.cfg, self.spec["scales"], self.first, self.B,
                                      self.device, table=self.table, streams=None if self.concurrent else 1)
        full = grid.all_gather_rows(local, counts, dim=2, force=True) if (world > 1 or self.collective) else local
        return full.cpu().numpy()

    def check(self, vals):
        from distillation_trajectories_amd import engine
        k = engine.SCALAR_KEYS.index("path_length_similarity")
        return {"path_length_similarity_sf0.5_gs3": float(vals[self.spec["sizes"].index(0.5), self.spec["scales"].index(3.0), :, k].mean()),
                "cells": int(vals.shape[0] * vals.shape[1] * vals.shape[2])}


# ------------------------------------------------------------------------------------------- CPU legs (oracle)
def usable_cores():
    """(threads to use, host logical cores, cgroup CPU quota or None).  The GPU box shows all of the host's cores (256:
    2 x EPYC 9575F) but grants this job a CPU quota (/sys/fs/cgroup/cpu.max, 16 per GPU): running more threads than the
    quota only thrashes, so the CPU legs use min(affinity, quota) threads and the line states all three numbers."""
    host = os.cpu_count() or 1
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // period)
        except (OSError, ValueError):
            pass
    return (min(n, quota) if quota else n), host, quota



def _oracle_models(spec):
    from distillation_trajectories_amd.config import Config
    from distillation_trajectories_amd.models import DiffusionUNet
    from distillation_trajectories_amd.synthetic import make_model
    from oracle import unet_ref
    cfg = Config()
    cfg.image_size = spec["H"]
    fns = []
    for sf, sd in zip(spec["sf"], spec["seeds"]):
        state = make_model(DiffusionUNet, cfg, sf, seed=sd).state_dict()
        fns.append(lambda x, t, c, state=state: unet_ref.unet_forward(state, x, t, c))
    return fns


def cpu_baseline(spec, batch=256, steps=8, pairs=8):
    """The oracle's restatement of the same loop on the host cores: p_sample_loop semantics (2 passes per
    step) for teacher and student at ``batch`` samples over the first ``steps`` timesteps, plus
    the metric function on ``pairs`` pairs scaled to the batch.  Reported in the benchmark's unit."""
    import torch
    from oracle import metrics_ref, sampler_ref
    H, T = spec["H"], spec["T"]
    prev_threads = torch.get_num_threads()
    threads, host_cores, quota = usable_cores()     # every core this job may use, as the reference's CPU mode does (scripts/run_on_cpu.py:27)
    torch.set_num_threads(threads)
    params = sampler_ref.diffusion_params(T)
    g = torch.Generator().manual_seed(1234)
    x0 = torch.randn(batch, C, H, H, generator=g)
    trajs, t_sample = [], 0.0
    with torch.no_grad():
        for fn in _oracle_models(spec):
            x = x0.clone()
            traj = [x]
            fn(x[:2], torch.full((2,), T - 1), None)                        # warm the allocator / oneDNN primitives
            t0 = time.perf_counter()
            for i in range(T - 1, T - 1 - steps, -1):
                x = sampler_ref.p_sample(fn, x, torch.full((batch,), i, dtype=torch.long), i, params, spec["guidance"],
                                         noise=torch.randn(x.shape, generator=g))
                traj.append(x)
            t_sample += time.perf_counter() - t0
            trajs.append(traj)
    t0 = time.perf_counter()
    for b in range(pairs):
        metrics_ref.compute_trajectory_metrics([s[b:b + 1] for s in trajs[0]], [s[b:b + 1] for s in trajs[1]])
    t_metric_pair = (time.perf_counter() - t0) / pairs * (T + 1) / (steps + 1)   # scaled to T+1-state trajectories
    units = 2 * batch * steps
    # per unit: sampler time + the pair's metric time spread over its 2*T trajectory-timesteps
    sec_per_unit = t_sample / units + t_metric_pair / (2 * T)
    torch.set_num_threads(prev_threads)
    return {"value": round(1.0 / sec_per_unit, 2), "unit": "trajectory-timesteps/s", "cores": threads, "host_cores": host_cores, "cpu_quota": quota, "kind": "port",
            "sample": f"oracle (torch-CPU restatement, {threads} threads): teacher+student p_sample loop, batch {batch}, "
                      f"{steps} of {T} timesteps, CFG 2 passes, + metrics on {pairs} pairs scaled to T+1 states; "
                      f"sampler {t_sample:.2f}s, metrics {t_metric_pair * 1e3:.1f} ms/pair"}


def cpu_full_length(spec, host_inputs, gpu_vals, gpu_traj, max_seconds=40.0, fns=None):
    """Reference-faithful B=1 leg: the oracle's FULL-length loop (all T timesteps, 2 passes per step) for the first
    pairs of rank 0's batch on the very inputs the GPU used, plus the oracle's metric function on each pair.
    Returns (b1 timing object, metric_delta_vs_cpu object)."""
    import numpy as np
    import torch
    from distillation_trajectories_amd import engine
    from oracle import metrics_ref, sampler_ref
    H, T = spec["H"], spec["T"]
    x_T, z, idx = host_inputs
    threads, host_cores, quota = usable_cores()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    params = sampler_ref.diffusion_params(T)
    fns = fns or _oracle_models(spec)
    per_key, traj_err, done, t_loop, t_met = {}, 0.0, 0, 0.0, 0.0
    nan_mismatch = 0
    t_start = time.perf_counter()
    with torch.no_grad():
        for b in range(x_T.shape[0]):
            if done and time.perf_counter() - t_start > max_seconds / 2:
                break
            pair = []
            for m, fn in enumerate(fns):
                x = x_T[b:b + 1].reshape(1, C, H, H)
                traj, s = [x], 0
                t0 = time.perf_counter()
                for i in idx:
                    noise = None
                    if i > 0:
                        noise = z[s, b:b + 1].reshape(1, C, H, H)
                        s += 1
                    x = sampler_ref.p_sample(fn, x, torch.full((1,), i, dtype=torch.long), i, params, spec["guidance"], noise=noise)
                    traj.append(x)
                t_loop += time.perf_counter() - t0
                pair.append(traj)
                want = torch.stack(traj).reshape(T + 1, -1)
                traj_err = max(traj_err, float((gpu_traj[m][:, b] - want).abs().max() / want.abs().max()))
            t0 = time.perf_counter()
            if C * H * H > 1000:
                np.random.seed(42 + b + 1)              # the engine's per-sample sub-sampling stream (trajectory_engine.py:93)
            want = metrics_ref.compute_trajectory_metrics(pair[0], pair[1])
            t_met += time.perf_counter() - t0
            for k in engine.SCALAR_KEYS:
                a, w = float(gpu_vals[k][b]), float(want[k])
                if np.isnan(a) or np.isnan(w):
                    nan_mismatch += int(np.isnan(a) != np.isnan(w))
                    continue
                if k == "trajectory_mse":               # compared before the ill-conditioned log1p(1 - 1000 mse)
                    a, w = 1.0 - np.expm1(a), 1.0 - np.expm1(w)
                per_key[k] = max(per_key.get(k, 0.0), abs(a - w) / max(abs(w), 1e-12))
            done += 1
    torch.set_num_threads(prev_threads)
    units = 2 * done * T
    b1 = {"value": round(units / (t_loop + t_met), 2), "unit": "trajectory-timesteps/s", "cores": threads, "host_cores": host_cores, "cpu_quota": quota, "kind": "port",
          "sample": f"oracle, B=1 like the reference's loops: {done} pair(s), all {T} timesteps, 2 passes per step, + metric "
                    f"function per pair; sampler {t_loop:.2f}s, metrics {t_met * 1e3 / max(done, 1):.1f} ms/pair"}
    delta = {"pairs": done, "timesteps": T, "max_rel": max(per_key.values()) if per_key else None,
             "worst_key": max(per_key, key=per_key.get) if per_key else None, "nan_position_mismatches": nan_mismatch,
             "trajectory_max_err_rel_to_max": traj_err, "tolerance": 1e-4,
             "per_key": {k: float(f"{v:.3g}") for k, v in per_key.items()},
             "note": "GPU metrics of rank 0's first pairs vs the oracle's full-length CPU loop on the same x_T / z / weights; "
                     "trajectory_mse compared as 1000 * mean step-MSE (pre-transform)"}
    return b1, delta


def tame_pair_delta(spec, wl, device, scale=0.01):
    """``metric_delta_vs_cpu`` in the INFORMATIVE regime.  The benchmark's random-init models diverge under the sampler
    (endpoint distance in the hundreds), where four of the 19 scalar metrics sit at their floor or at NaN; here the same
    two models with their final 1x1 layer scaled by ``scale`` (states stay O(1), teacher and student close) run the same
    loop on rank 0's first pairs -- GPU sampler + metric kernels against the oracle's loop + metric function, all 19
    metrics compared.  Outside the timed region."""
    import copy
    import torch
    from distillation_trajectories_amd import engine
    from distillation_trajectories_amd._hip import RULE_PSAMPLE
    from oracle import unet_ref
    x_T, z, idx = wl.host_inputs
    B, T, H = x_T.shape[0], wl.T, wl.H
    fns, trajs = [], []
    for m in wl.models:
        m2 = copy.deepcopy(m)
        with torch.no_grad():
            m2.final.weight.mul_(scale)
            m2.final.bias.mul_(scale)
        sd = {k: v.detach().cpu().clone() for k, v in m2.state_dict().items()}
        fns.append(lambda x, t, c, sd=sd: unet_ref.unet_forward(sd, x, t, c))
        h = engine.UNetHandle(m2.state_dict(), device)
        tb = h.time_bias_general(wl.tb_t, wl.tb_cond, wl.tb_present, 2 * T)
        traj = torch.empty(T + 1, B, wl.E, device=device)
        traj[0].copy_(x_T.to(device))
        zz = z.reshape(-1, wl.E).to(device)
        shift, k = [], 0
        for flag in wl.has_noise:
            shift.append(k * B)
            k += int(flag)
        h.sample(RULE_PSAMPLE, traj, H, H, tb, 2, wl.coef, wl.has_noise, z=zz, z_shift=shift, w_scalar=spec["guidance"])
        trajs.append(traj)
    if wl.E > 1000:
        return None
    sums, w1 = engine.device_pair_metrics(trajs[0], trajs[1])
    host = torch.cat([sums.reshape(B, -1), w1], dim=1).cpu().numpy()
    n = T + 1
    vals = engine.batch_scalar_metrics(host[:, : 4 * n].reshape(-1, n, 4), host[:, 4 * n:], H * H, wl.E)
    _, delta = cpu_full_length(spec, wl.host_inputs, vals, [t.cpu() for t in trajs], fns=fns)
    delta["final_layer_scale"] = scale
    delta["gpu_values_pair0"] = {k: float(f"{float(vals[k][0]):.6g}") for k in engine.SCALAR_KEYS}
    delta["note"] = ("same models with final.weight / final.bias scaled by %g so that the states stay O(1) and teacher / student stay "
                     "close: every one of the 19 scalar metrics is in its informative range (see gpu_values_pair0); GPU sampler + "
                     "metric kernels vs the oracle's full-length CPU loop + metric function on the same x_T / z" % scale)
    return delta


def traffic_from_profiles(kernel_name, config=1):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 note), or None if not recorded."""
    names = ("r03c2_hbm_traffic.json",) if config == 2 else ("r03_hbm_traffic.json", "r02_hbm_traffic.json", "hbm_traffic.json")
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                hit = json.load(f).get(kernel_name)
            if hit:
                return hit
        except (OSError, ValueError):
            pass
    return None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))            # before anything below touches the GPU
    import numpy as np   # noqa: F401
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DT_BENCH_DRYRUN=1 (tests/test_bench_launcher.py, no GPU): rendezvous + rank gather over gloo, then stop
    dry = os.environ.get("DT_BENCH_DRYRUN") == "1"
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # one process per GPU; DT_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path
    backend = "gloo" if dry else os.environ.get("DT_BENCH_BACKEND", "nccl")
    n_dev = 1 if dry else torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, this node shows {n_dev} (set DT_BENCH_BACKEND=gloo "
                         "to rehearse the multi-rank path on fewer GPUs)")
    dev_index = local_rank % n_dev
    device = None
    if not dry:
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    ranks = [{"rank": 0, "local_rank": 0, "device": dev_index, "pid": os.getpid()}]
    # DT_BENCH_NCCL1=1: a process group of ONE rank over nccl (= RCCL), so that the collective code path (barrier, the metric
    # all-gather on device tensors, the max-over-ranks all-reduce) runs on a single-GPU box too
    solo_group = world == 1 and not dry and os.environ.get("DT_BENCH_NCCL1") == "1"
    if solo_group:
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or solo_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)      # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
        me = torch.tensor([rank, local_rank, dev_index, os.getpid()], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        everyone = [torch.empty_like(me) for _ in range(world)]
        dist.all_gather(everyone, me)                               # the collective itself shows who is in the job
        ranks = [{"rank": int(t[0]), "local_rank": int(t[1]), "device": int(t[2]), "pid": int(t[3])} for t in everyone]
    else:
        dist = None
    if dry:
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": ranks, "backend": backend}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from distillation_trajectories_amd import _hip
    _hip.load()
    spec = CONFIGS[args.config]
    batch = args.batch or spec["batch"]
    wl = (GridWorkload if args.config == 2 else PairWorkload)(spec, device, rank, batch, concurrent=not args.serial)
    wl.collective = solo_group
    counts = [batch] * world

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        vals = wl.step(world, counts)
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, no instrumentation
    barrier()
    if os.environ.get("DT_BENCH_MARKERS") == "1":          # rocprofv3 runs: bracket the timed steps (profiles/collect_r02.sh)
        _hip.profile_marker(1)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vals = wl.step(world, counts)
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("DT_BENCH_MARKERS") == "1":
        _hip.profile_marker(2)
        torch.cuda.synchronize()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # ---- the same K steps again with every launch bracketed by HIP events on the launch stream
    # (per-kernel durations for the roofline line; the ~2 event records per launch cost wall time,
    # so this pass is kept out of `value` and its own wall time is reported beside it)
    kernels, profiled_elapsed = {}, None
    if not args.no_profile and rank == 0:
        wl.concurrent = False      # one stream: per-kernel durations free of cross-stream contention
        _hip.profile_begin()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl.step(1, [batch])
        torch.cuda.synchronize()
        profiled_elapsed = time.perf_counter() - t1
        kernels = _hip.profile_end()
    if dist is not None:
        dist.barrier()

    units_per_step = wl.units_per_step * world
    value = units_per_step * args.steps / elapsed
    out = {
        "metric": "trajectory-timesteps/sec (B×T U-Net fwd) at 16×16 T=50",
        "value": round(value, 1), "unit": "trajectory-timesteps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (fp32-accurate; convolutions: exact 3-plane bf16 operand split on bf16 MFMA with fp32 accumulate, or "
                 "native fp32 MFMA, chosen per layer)", "data": "synthetic",
        "config": wl.describe(world),
        "per_gpu": round(value / world, 1),
        "ranks": ranks, "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if (world > 1 or solo_group) else None,
        "collective_path": ("one-rank nccl group: barrier + metric all-gather on device tensors + all-reduce of the step time went through RCCL"
                            if solo_group else None),
        "metric_check": wl.check(vals),
    }
    # launch plans in use (engine._Plans): "table:<sha1>" = the committed per-arch table, "heuristic", "tuned:<sha1>" only with DT_AUTOTUNE=1
    from distillation_trajectories_amd import engine as _engine
    if args.config == 2:
        mods = [("teacher", wl.teacher)] + [(f"sf={sf}", m) for sf, m in zip(spec["sizes"], wl.students)]
        out["launch_plans"] = {name: _engine.UNetHandle.for_module(m).plan_ids() for name, m in mods}
    else:
        out["launch_plans"] = {f"sf={sf}": ps[0][0].plan_ids() for sf, ps in zip(spec["sf"], wl.parts)}
    if kernels:
        # A kernel = one __global__ template; its tile instantiations <BM,BN> are the same code on other tile sizes and
        # which of them the autotuner picks per layer varies run to run, so the roofline entry is per template.
        groups = {}
        for n, k in kernels.items():
            g = groups.setdefault(n.split("<")[0], {"ms": 0.0, "flops": 0.0, "launches": 0, "members": {}})
            g["ms"] += k["ms"]; g["flops"] += k["flops"]; g["launches"] += k["launches"]; g["members"][n] = k
        dom_name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
        total_ms = sum(k["ms"] for k in kernels.values())
        conv = {n: k for n, k in kernels.items() if n.startswith(("conv_gemm", "conv_strip", "lowres_"))}
        conv_ms, conv_fl = sum(k["ms"] for k in conv.values()), sum(k["flops"] for k in conv.values())
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        split = "bf16x6" in dom_name
        peak = PEAK_BF16_MFMA_TFLOPS / PLANE_PRODUCTS if split else PEAK_FP32_MFMA_TFLOPS
        traffic = None
        per = {n: traffic_from_profiles(n, args.config) for n in dom["members"]}
        if all(per.values()):
            traffic = {"hbm_bytes_per_launch": int(sum(per[n]["hbm_bytes_per_launch"] * k["launches"] for n, k in dom["members"].items())
                                                   / dom["launches"]),
                       "per_instantiation": {n: per[n]["hbm_bytes_per_launch"] for n in per}, "method": next(iter(per.values()))["method"]}
        out["roofline"] = {"bound": "mfma", "kernel": dom_name + "<BM,BN>", "achieved": round(achieved, 2),
                           "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                           "peak_basis": ("dense bf16 MFMA 2500 TF/s / 6 plane products per fp32-accurate product; "
                                          f"issued bf16 MFMA rate {achieved * PLANE_PRODUCTS:.0f} TF/s; native fp32 MFMA "
                                          f"peak is {PEAK_FP32_MFMA_TFLOPS} TF/s") if split else "dense fp32 MFMA",
                           "traffic": traffic["hbm_bytes_per_launch"] if traffic else None, "traffic_detail": traffic,
                           "launches": dom["launches"],
                           "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2),
                           "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
                           "instantiations": {n: {"launches": k["launches"], "avg_launch_us": round(k["ms"] / k["launches"] * 1e3, 2),
                                                  "tflops": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2)}
                                              for n, k in sorted(dom["members"].items(), key=lambda kv: -kv[1]["ms"])},
                           "share_of_kernel_time": round(dom["ms"] / total_ms, 3),
                           "all_conv_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                           "all_conv_vs_native_fp32_peak": round(conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                           "timed_with": f"hipEventRecord pairs around every launch, second pass of the same {args.steps} "
                                         f"steps on ONE stream ({profiled_elapsed / args.steps * 1e3:.1f} ms/step with events vs "
                                         f"{elapsed / args.steps * 1e3:.1f} ms/step in the timed region, where the teacher and "
                                         "student loops (and sub-batches, if any) each run on their own stream)"}
        out["tile_choices"] = wl.choices
        out["kernels"] = {n: {"launches": k["launches"], "ms": round(k["ms"], 3),
                              "tflops": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2) if k["flops"] else None,
                              "gbps": round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 1) if k["bytes"] else None}
                          for n, k in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config != 2:
        gpu_traj = [t[:, :2].cpu() for t in wl.traj]
        if args.config == 1:
            out["cpu_baseline"] = cpu_baseline(spec)
            b1, delta = cpu_full_length(spec, wl.host_inputs, vals, gpu_traj)
            out["cpu_baseline"]["b1"] = b1
            out["metric_delta_vs_cpu"] = delta
            out["metric_delta_vs_cpu_tame"] = tame_pair_delta(spec, wl, device)
        else:       # configs[4]: one full-length CPU pair would take minutes; a bounded batched sample only
            out["cpu_baseline"] = cpu_baseline(spec, batch=16, steps=2, pairs=1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
