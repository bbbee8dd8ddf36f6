"""One teacher forward at the bench shape (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distillation_trajectories_amd import _hip, engine
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model
cfg = Config(); cfg.image_size = 16
m = make_model(DiffusionUNet, cfg, 1.0).to("cuda:0")
h = engine.UNetHandle.for_module(m)
x = torch.randn(256, 3, 16, 16, device="cuda:0")
tb = h.time_bias([10, 10], [_hip.COND_NONE, _hip.COND_ONE])
h.forward(x, tb, 2, 256, tune=False)
for (j, slot) in ((1, 2), (7, 1), (2, 1)):
    h.set_conv_choice(512, 16, 16, j, slot, 128, 128, 1, int(os.environ.get("DT_ONE_PREC", "4")), 0)
for _ in range(3):
    h.forward(x, tb, 2, 256, tune=False)
torch.cuda.synchronize()
