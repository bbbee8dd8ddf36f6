"""Randomised parity of the fused small-model kernel: size factors with padded dims 16/32, 32/48, 32/64 (odd real channel counts),
1-3 image channels, batches 1..40, the three update rules, plain / CFG / mixed batches, 1..80 timesteps (loops past 64 steps
chain launches) -- the fused loop against the layered per-timestep launches, and single forwards against the oracle.
Usage: fuzz_fused.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distillation_trajectories_amd import engine
from distillation_trajectories_amd._hip import COND_NONE, COND_ONE, COND_ZERO, RULE_ENGINE, RULE_MANAGER, RULE_PSAMPLE
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model
from oracle import unet_ref

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
worst_f = worst_l = 0.0
for case in range(n_cases):
    sf = float(rng.choice([0.01, 0.08, 0.13, 0.15, 0.17, 0.2, 0.22, 0.25]))
    C = int(rng.choice([1, 2, 3, 3, 3]))
    cfg = Config(); cfg.image_size, cfg.channels = 16, C
    m = make_model(DiffusionUNet, cfg, sf, seed=2000 + case)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev)
    h = engine.UNetHandle.for_module(m)
    assert h.fused_active(16, 16), (sf, h.dims)
    E = C * 256
    B = int(rng.integers(1, 41))
    # ---- forward vs oracle
    x = torch.randn(B, C, 16, 16)
    t = torch.randint(0, 50, (B,))
    cond = torch.rand(B, 1)
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x, t, cond)
    got = m(x.to(dev), t.to(dev), cond.to(dev)).cpu()
    ef = float((got - want).abs().max() / (want.abs().max() + 1e-12))
    # ---- sampler loop: fused vs layered
    rule = int(rng.choice([RULE_ENGINE, RULE_PSAMPLE, RULE_MANAGER]))
    n_steps = int(rng.choice([1, 2, 7, 30, 64, 65, 80]))
    kind = str(rng.choice(["plain", "cfg", "mixed"])) if B >= 2 else str(rng.choice(["plain", "cfg"]))
    ts = [int(v) for v in rng.integers(0, 50, n_steps)]
    has_noise = [bool(v) for v in (rng.random(n_steps) > 0.15)]
    coef = [(float(0.97 + 0.03 * rng.random()), float(0.02 + 0.05 * rng.random()), float(0.05 * rng.random())) for _ in range(n_steps)]
    if rule == RULE_MANAGER:
        coef = [(0.1, 0.9486833, 0.03)] * n_steps
    z = torch.randn(n_steps * B + 3, E).to(dev)
    z_row = torch.from_numpy(rng.permutation(B).astype(np.int32)).to(dev)
    shift = [i * B for i in range(n_steps)]
    x_T = torch.randn(B, E)
    trajs = []
    for fused in (True, False):
        h.set_fused(fused)
        traj = torch.empty(n_steps + 1, B, E, device=dev)
        traj[0] = x_T.to(dev)
        if kind == "mixed":
            single = int(rng.integers(1, B)) if fused else single
            G = B - single
            # rows [single | G cond 0 | G cond 1]; tb_div must divide `single` and the row count: 1
            modes = ([COND_NONE] * single + [COND_ZERO] * G + [COND_ONE] * G)
            tb = h.time_bias([tt for tt in ts for _ in range(single + 2 * G)], modes * n_steps)
            w = torch.cat([torch.zeros(single), torch.from_numpy(rng.uniform(1.0, 9.0, G).astype(np.float32))]) if fused else w
            h.sample_mixed(rule, traj, 16, 16, tb, 1, single, coef, has_noise, z, z_row, shift, w.to(dev))
        else:
            n_pass = 2 if kind == "cfg" else 1
            modes = [COND_NONE] * n_steps if n_pass == 1 else [COND_NONE, COND_ONE] * n_steps
            tb = h.time_bias([tt for tt in ts for _ in range(n_pass)], modes)
            w = (torch.from_numpy(rng.uniform(0.0, 9.0, B).astype(np.float32)).to(dev) if fused else w) if n_pass == 2 else None
            h.sample(rule, traj, 16, 16, tb, n_pass, coef, has_noise, z=z, z_row=z_row, z_shift=shift, w=w, w_scalar=2.0)
        trajs.append(traj.cpu())
    h.set_fused(True)
    fin = torch.isfinite(trajs[1])
    same_fin = torch.equal(fin, torch.isfinite(trajs[0]))
    scale = trajs[1][fin].abs().max().item() + 1e-12
    el = float((trajs[0][fin] - trajs[1][fin]).abs().max().item() / scale) if same_fin else float("inf")
    worst_f, worst_l = max(worst_f, ef), max(worst_l, el)
    ok = ef < 1e-4 and el < 1e-4
    print(f"case {case:3d}: sf={sf:4.2f} dims={h.dims[:2]} C={C} B={B:2d} rule={rule} {kind:5s} steps={n_steps:2d}: forward vs oracle {ef:.1e}, "
          f"loop fused vs layered {el:.1e}{'' if ok else '   <-- FAIL'}", flush=True)
    assert ok
print("worst forward", worst_f, "worst loop", worst_l)
