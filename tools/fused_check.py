"""Fused small-model kernel (dt_fused.hip) against the oracle and against the layered kernels: forwards (plain, CFG pair,
mixed batch), then the three sampler rules, plus the wall time of a 448-row sampler loop either way."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distillation_trajectories_amd import engine
from distillation_trajectories_amd._hip import COND_NONE, COND_ONE, RULE_ENGINE, RULE_MANAGER, RULE_PSAMPLE
from distillation_trajectories_amd.analysis.trajectory_engine import sample_grid_groups
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model, noise_table, seeded_noise
from oracle import unet_ref

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = Config(); cfg.image_size, cfg.timesteps = 16, 50
sizes = [float(v) for v in os.environ.get("DT_SIZES", "0.01,0.1,0.2").split(",")]
for sf in sizes:
    m = make_model(DiffusionUNet, cfg, sf)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev)
    h = engine.UNetHandle.for_module(m)
    lib = h.lib
    print(f"sf {sf}: dims {h.dims}, fused active {lib.dt_unet_fused_active(h.h, 16, 16)}")
    for B in (1, 2, 5, 8):
        x = seeded_noise(5 + B, (B, 3, 16, 16))
        t = torch.randint(0, 50, (B,))
        cond = torch.rand(B, 1)
        with torch.no_grad():
            want = unet_ref.unet_forward(sd, x, t, cond)
        lib.dt_unet_set_fused(h.h, 1)
        got = m(x.to(dev), t.to(dev), cond.to(dev)).cpu()
        lib.dt_unet_set_fused(h.h, 0)
        lay = m(x.to(dev), t.to(dev), cond.to(dev)).cpu()
        lib.dt_unet_set_fused(h.h, 1)
        sc = want.abs().max().item()
        print(f"  forward B={B}: fused-oracle {(got - want).abs().max().item() / sc:.2e}  layered-oracle {(lay - want).abs().max().item() / sc:.2e} (rel. to max {sc:.2f})")
    # CFG pair and mixed batch through the handle
    B = 6
    x = seeded_noise(99, (B, 3, 16, 16)).to(dev)
    tb = h.time_bias([7, 7], [COND_NONE, COND_ONE])
    outs = []
    for on in (1, 0):
        lib.dt_unet_set_fused(h.h, on)
        # mixed: images 0..1 single pass, 2..5 two passes: 10 rows, tb_div 2 -> 5 time-bias rows
        tbm = h.time_bias([7] * 5, [COND_NONE, COND_NONE, COND_NONE, COND_ONE, COND_ONE])
        outs.append((h.forward(x, tb, 2, B).cpu(), h.forward_mixed(x, tbm, 2, 2).cpu()))
    lib.dt_unet_set_fused(h.h, 1)
    print(f"  cfg pair fused-layered {(outs[0][0] - outs[1][0]).abs().max().item():.2e}, mixed fused-layered {(outs[0][1] - outs[1][1]).abs().max().item():.2e} "
          f"(max {outs[1][0].abs().max().item():.2f})")
    # sampler loops: grid driver (ENGINE rule, mixed batch)
    S, T = int(os.environ.get("DT_S", "64")), 50
    table = noise_table(42, S + T - 1, (1, 3, 16, 16)).reshape(S + T - 1, -1).to(dev)
    res = []
    for on in (1, 0):
        lib.dt_unet_set_fused(h.h, on)
        fn = lambda: sample_grid_groups(h, table, 0, S, T, [1.0, 3.0, 7.0, 20.0], 16, 16)
        out = fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        res.append((ms, out))
    lib.dt_unet_set_fused(h.h, 1)
    a, b = res[0][1], res[1][1]
    def flat(o):
        if isinstance(o, torch.Tensor):
            return [o]
        if isinstance(o, dict):
            return [v for k in sorted(o, key=str) for v in flat(o[k])]
        if isinstance(o, (list, tuple)):
            return [v for e in o for v in flat(e)]
        return []
    fa, fb = flat(a), flat(b)
    err = max((u.float() - v.float()).abs().max().item() for u, v in zip(fa, fb))
    mx = max(v.float().abs().max().item() for v in fb)
    print(f"  grid loop {7 * S} rows x {T - 1} forwards: fused {res[0][0]:.2f} ms, layered {res[1][0]:.2f} ms; trajectories differ by {err:.2e} (max {mx:.2f}, {len(fa)} tensors)")
