import time, sys, os
sys.path.insert(0, os.getcwd())
import torch
from distillation_trajectories_amd import _hip, engine
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model
cfg = Config(); cfg.image_size = 16
for sf in (1.0, 0.5, 0.2):
    m = make_model(DiffusionUNet, cfg, sf).to("cuda:0")
    h = engine.UNetHandle.for_module(m)
    x = torch.randn(256, 3, 16, 16, device="cuda:0")
    tb = h.time_bias([10, 10], [_hip.COND_NONE, _hip.COND_ONE])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.forward(x, tb, 2, 256, tune=True)
    torch.cuda.synchronize(); print("sf", sf, "autotune + forward", round(time.perf_counter() - t0, 2), "s", flush=True)
