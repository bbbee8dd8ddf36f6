#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: trajectory-timesteps/sec (BASELINE.json metric).

Workload (BASELINE.json configs[1]): teacher (size_factor 1.0) vs student (size_factor 0.5), 16x16x3,
T=50, batch 256, guidance scale 1.0 through the CFG sampler of utils/diffusion.py (p_sample_loop: two
U-Net passes per step, cond=ones and cond=None, then the DDPM update), followed by the trajectory
metrics of the 256 (teacher, student) pairs.  One "step" of this benchmark is one such pass:
2 models x 256 samples x 50 timesteps = 25,600 trajectory-timesteps per GPU.

Inputs (start noise x_T and the per-step Gaussian noise, shared by teacher and student) are synthetic,
seeded, and resident in HBM before the timed region.  N > 1: one process per GPU (torchrun), samples
sharded across ranks, no data-path collective; the only exchange is one RCCL all-gather of the
per-sample metric tensor at the end of each step (weak scaling: 256 samples per GPU).

Prints ONE JSON line on rank 0 (contract in the task description), with a ``roofline`` object for the
dominant kernel (HIP-event timed inside this process) and a ``cpu_baseline`` object (the oracle's
restatement of the same loop timed on the host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md: dense matrix peaks.  The split-bf16 kernels evaluate every fp32
# product as 6 bf16 plane products, so their fp32-equivalent ceiling is the dense bf16 peak / 6.
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PLANE_PRODUCTS = 6
T, BATCH, H, C = 50, 256, 16, 3
TEACHER_SF, STUDENT_SF = 1.0, 0.5
GUIDANCE = 1.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--batch", type=int, default=BATCH, help="samples per GPU (default: the configs[1] batch)")
    ap.add_argument("--serial", action="store_true", help="run the teacher and student loops back to back on one stream")
    return ap.parse_args()


class Workload:
    """Device-resident state of one rank's share of the benchmark."""

    def __init__(self, device, rank, batch, concurrent=True):
        from distillation_trajectories_amd import engine
        from distillation_trajectories_amd._hip import COND_NONE, COND_ONE
        from distillation_trajectories_amd.config import Config
        from distillation_trajectories_amd.models import DiffusionUNet
        from distillation_trajectories_amd.synthetic import make_model
        from distillation_trajectories_amd.utils.diffusion import (get_diffusion_params, psample_coefficients,
                                                                   timestep_indices)
        self.engine, self.device, self.B = engine, device, batch
        cfg = Config()
        cfg.image_size, cfg.timesteps, cfg.sample_steps = H, T, T
        self.models = [make_model(DiffusionUNet, cfg, sf).to(device) for sf in (TEACHER_SF, STUDENT_SF)]
        self.handles = [engine.UNetHandle.for_module(m) for m in self.models]
        # Samples are independent, so a model's batch may run as several contiguous sub-batches, each with its own
        # handle (workspace), HIP stream and host thread: DT_BENCH_PARTS="teacher_parts,student_parts".  Measured
        # 1,1: 259-262 k  2,1: 270-274 k  2,2: 242 k  3,1: 225 k trajectory-timesteps/s.  The default stays 1,1:
        # the teacher as two halves buys 4 % of wall time by overlap but halves every teacher launch, so the
        # per-launch (roofline) figures then describe smaller, less efficient launches (all convs 126 vs 145 TF/s).
        parts = [int(v) for v in os.environ.get("DT_BENCH_PARTS", "1,1").split(",")]   # --serial: same sub-batches, one stream
        self.parts = []
        for i, n in enumerate(parts):
            n = max(1, min(n, batch))
            bounds = [(batch * k // n, batch * (k + 1) // n) for k in range(n)]
            hs = [self.handles[i]] + [engine.UNetHandle(self.models[i].state_dict(), device) for _ in range(n - 1)]
            self.parts.append(list(zip(hs, bounds)))
        self.E = C * H * H
        idx = timestep_indices(T, T)
        self.coef = psample_coefficients(get_diffusion_params(T, cfg), idx)
        self.has_noise = [i > 0 for i in idx]
        g = torch.Generator().manual_seed(1234 + rank)          # torch CPU generator, like the reference's CPU mode
        x_T = torch.randn(batch, self.E, generator=g)
        z = torch.randn(sum(self.has_noise) * batch, self.E, generator=g)
        self.x_T, self.z = x_T.to(device), z.to(device)
        self.z_shift, k = [], 0
        for flag in self.has_noise:
            self.z_shift.append(k * batch)
            k += int(flag)
        self.tb = [h.time_bias([i for i in idx for _ in (0, 1)], [COND_NONE, COND_ONE] * T) for h in self.handles]
        self.traj = [torch.empty(T + 1, batch, self.E, device=device) for _ in self.handles]
        self.part_traj = [[self.traj[i] if len(ps) == 1 else torch.empty(T + 1, hi - lo, self.E, device=device)
                           for (_, (lo, hi)) in ps] for i, ps in enumerate(self.parts)]
        for i, ps in enumerate(self.parts):
            for h, (lo, hi) in ps:
                # tile / split / arithmetic autotuning happens here, one launch shape at a time on an idle GPU
                h.forward(self.x_T[lo:hi].reshape(hi - lo, C, H, H), self.tb[i][:2].contiguous(), 2, hi - lo, tune=True)
        torch.cuda.synchronize()
        self.choices = None
        self.concurrent = concurrent
        self.streams = [[torch.cuda.Stream(device=device) for _ in ps] for ps in self.parts]

    def step(self, world, counts):
        """One pass of the hot path: both samplers, the metric reductions, the metric all-gather."""
        from distillation_trajectories_amd._hip import RULE_PSAMPLE
        from distillation_trajectories_amd.grid import all_gather_rows
        eng = self.engine

        def run(i, k=0):
            (h, (lo, hi)), tb, traj = self.parts[i][k], self.tb[i], self.part_traj[i][k]
            traj[0].copy_(self.x_T[lo:hi])
            h.sample(RULE_PSAMPLE, traj, H, H, tb, 2, self.coef, self.has_noise, z=self.z,
                     z_shift=[v + lo for v in self.z_shift], w_scalar=GUIDANCE)
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
            for i, ps in enumerate(self.parts):
                if len(ps) > 1:                      # reassemble [T+1, B, E] for the pairwise metrics
                    torch.cat(self.part_traj[i], dim=1, out=self.traj[i])
        else:
            for i, ps in enumerate(self.parts):
                for k in range(len(ps)):
                    run(i, k)
                if len(ps) > 1:
                    torch.cat(self.part_traj[i], dim=1, out=self.traj[i])
        sums = eng.device_metric_sums(self.traj[0], self.traj[1])          # [B, T+1, 4] float64
        w1 = eng.device_wasserstein(self.traj[0], self.traj[1])            # [B, T+1]   float64
        local = torch.cat([sums.reshape(self.B, -1), w1], dim=1)
        full = all_gather_rows(local, counts, dim=0) if world > 1 else local
        host = full.cpu().numpy()                                          # syncs the stream
        n = T + 1
        vals = eng.batch_scalar_metrics(host[:, : 4 * n].reshape(-1, n, 4), host[:, 4 * n:], H * H, self.E)
        if self.choices is None:
            self.choices = {f"sf={sf}": [list(c) for c in ps[0][0].conv_choices(2 * (ps[0][1][1] - ps[0][1][0]), H, H)]
                            for sf, ps in zip((TEACHER_SF, STUDENT_SF), self.parts)}
        return vals


def cpu_baseline(batch=256, steps=8, pairs=8):
    """The oracle's restatement of the same loop on the host cores: p_sample_loop semantics (2 passes per
    step) for teacher and student at ``batch`` samples over the first ``steps`` of the 50 timesteps, plus
    the metric function on ``pairs`` pairs scaled to the batch.  Reported in the benchmark's unit."""
    from distillation_trajectories_amd.config import Config
    from distillation_trajectories_amd.models import DiffusionUNet
    from distillation_trajectories_amd.synthetic import make_model
    from oracle import metrics_ref, sampler_ref, unet_ref
    cfg = Config()
    cfg.image_size = H
    prev_threads = torch.get_num_threads()
    threads = min(16, os.cpu_count() or 1)          # the GPU box's CPU share per GPU
    torch.set_num_threads(threads)
    params = sampler_ref.diffusion_params(T)
    g = torch.Generator().manual_seed(1234)
    x0 = torch.randn(batch, C, H, H, generator=g)
    trajs, t_sample = [], 0.0
    with torch.no_grad():
        for sf in (TEACHER_SF, STUDENT_SF):
            sd = make_model(DiffusionUNet, cfg, sf).state_dict()
            fn = lambda x, t, c, sd=sd: unet_ref.unet_forward(sd, x, t, c)   # noqa: E731
            x = x0.clone()
            traj = [x]
            fn(x[:2], torch.full((2,), T - 1), None)                        # warm the allocator / oneDNN primitives
            t0 = time.perf_counter()
            for i in range(T - 1, T - 1 - steps, -1):
                x = sampler_ref.p_sample(fn, x, torch.full((batch,), i, dtype=torch.long), i, params, GUIDANCE,
                                         noise=torch.randn(x.shape, generator=g))
                traj.append(x)
            t_sample += time.perf_counter() - t0
            trajs.append(traj)
    t0 = time.perf_counter()
    for b in range(pairs):
        metrics_ref.compute_trajectory_metrics([s[b:b + 1] for s in trajs[0]], [s[b:b + 1] for s in trajs[1]])
    t_metric_pair = (time.perf_counter() - t0) / pairs * (T + 1) / (steps + 1)   # scaled to 51-state trajectories
    units = 2 * batch * steps
    # per unit: sampler time + the pair's metric time spread over its 2*T trajectory-timesteps
    sec_per_unit = t_sample / units + t_metric_pair / (2 * T)
    torch.set_num_threads(prev_threads)
    return {"value": round(1.0 / sec_per_unit, 2), "unit": "trajectory-timesteps/s", "cores": threads, "kind": "port",
            "sample": f"oracle (torch-CPU restatement, {threads} threads): teacher+student p_sample loop, batch {batch}, "
                      f"{steps} of {T} timesteps, CFG 2 passes, + metrics on {pairs} pairs scaled to T+1 states; "
                      f"sampler {t_sample:.2f}s, metrics {t_metric_pair * 1e3:.1f} ms/pair"}


def traffic_from_profiles(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 note), or None if not recorded."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel_name)
    except (OSError, ValueError):
        return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # one process per GPU; DT_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path
    backend = os.environ.get("DT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)      # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    else:
        dist = None

    from distillation_trajectories_amd import _hip
    _hip.load()
    wl = Workload(device, rank, args.batch, concurrent=not args.serial)
    counts = [args.batch] * world

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        vals = wl.step(world, counts)
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, no instrumentation
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vals = wl.step(world, counts)
    barrier()
    elapsed = time.perf_counter() - t0
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
            wl.step(1, [args.batch])
        torch.cuda.synchronize()
        profiled_elapsed = time.perf_counter() - t1
        kernels = _hip.profile_end()
    if dist is not None:
        dist.barrier()

    units_per_step = 2 * args.batch * T * world
    value = units_per_step * args.steps / elapsed
    out = {
        "metric": "trajectory-timesteps/sec (B×T U-Net fwd) at 16×16 T=50",
        "value": round(value, 1), "unit": "trajectory-timesteps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (fp32-accurate; convolutions: exact 3-plane bf16 operand split on bf16 MFMA with fp32 accumulate, or "
                 "native fp32 MFMA, chosen per layer)", "data": "synthetic",
        "config": {"workload": "configs[1]: teacher sf=1.0 vs student sf=0.5, 16x16x3, T=50, batch 256/GPU, "
                               "p_sample_loop CFG (2 U-Net passes/step, w=1.0) + trajectory metrics of the 256 pairs",
                   "batch_per_gpu": args.batch, "timesteps": T, "image": [C, H, H], "guidance_scale": GUIDANCE,
                   "unet_passes_per_step": 2, "parallelism": f"sample-sharded x{world}, RCCL all-gather of metrics",
                   "streams_per_gpu": {"teacher_sub_batches": len(wl.parts[0]), "student_sub_batches": len(wl.parts[1])}},
        "per_gpu": round(value / world, 1),
        "metric_check": {"mean_endpoint_distance": float(np.mean(vals["endpoint_distance"])),
                         "mean_wasserstein": float(np.mean(vals["mean_wasserstein"])),
                         "pairs": int(len(vals["mse"]))},
    }
    if kernels:
        # A kernel = one __global__ template; its tile instantiations <BM,BN> are the same code on other tile sizes and
        # which of them the autotuner picks per layer varies run to run, so the roofline entry is per template.
        groups = {}
        for n, k in kernels.items():
            g = groups.setdefault(n.split("<")[0], {"ms": 0.0, "flops": 0.0, "launches": 0, "members": {}})
            g["ms"] += k["ms"]; g["flops"] += k["flops"]; g["launches"] += k["launches"]; g["members"][n] = k
        dom_name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
        total_ms = sum(k["ms"] for k in kernels.values())
        conv = {n: k for n, k in kernels.items() if n.startswith(("conv_gemm", "conv_strip"))}
        conv_ms, conv_fl = sum(k["ms"] for k in conv.values()), sum(k["flops"] for k in conv.values())
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        split = "bf16x6" in dom_name
        peak = PEAK_BF16_MFMA_TFLOPS / PLANE_PRODUCTS if split else PEAK_FP32_MFMA_TFLOPS
        traffic = None
        per = {n: traffic_from_profiles(n) for n in dom["members"]}
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
