"""Randomised forward parity: batch, size factor, image height / width (multiples of 16) and guidance batch against
the oracle.  Usage: fuzz_forward.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model
from oracle import unet_ref
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n_cases):
    sf = float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5, 0.75, 1.0]))
    H, W = int(rng.choice([16, 32, 48])), int(rng.choice([16, 32, 48]))
    B = int(rng.integers(1, 40)) if H * W <= 1024 else int(rng.integers(1, 12))
    cfg = Config(); cfg.image_size = H
    torch.manual_seed(1000 + case)
    m = make_model(DiffusionUNet, cfg, sf)
    x = torch.randn(B, 3, H, W)
    t = torch.randint(0, 50, (B,))
    cond = (torch.rand(B, 1) > 0.5).float()
    with torch.no_grad():
        want = unet_ref.unet_forward(m.state_dict(), x, t, cond).numpy()
    got = m.to("cuda:0")(x.to("cuda:0"), t.to("cuda:0"), cond.to("cuda:0")).cpu().numpy()
    err = float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))
    worst = max(worst, err)
    flag = "" if err < 1e-4 else "   <-- FAIL"
    print(f"case {case:3d}: sf={sf:4.2f} B={B:3d} {H}x{W}  max rel err {err:.2e}{flag}", flush=True)
    assert err < 1e-4
print("worst", worst)
