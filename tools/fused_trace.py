"""Per-layer cycle counts of one step of the fused small-model kernel (tools build: python -m distillation_trajectories_amd.csrc.build --tools;
DT_FUSED_TRACE=1 makes launch_unet_fused print them)."""
import os, sys
os.environ["DT_FUSED_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distillation_trajectories_amd import engine
from distillation_trajectories_amd.analysis.trajectory_engine import sample_grid_groups
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model, noise_table

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = Config(); cfg.image_size, cfg.timesteps = 16, 50
S, T = 64, 50
table = noise_table(42, S + T - 1, (1, 3, 16, 16)).reshape(S + T - 1, -1).to(dev)
for sf in [float(v) for v in os.environ.get("DT_SIZES", "0.01,0.2").split(",")]:
    h = engine.UNetHandle.for_module(make_model(DiffusionUNet, cfg, sf).to(dev))
    print("sf", sf, file=sys.stderr)
    sample_grid_groups(h, table, 0, S, T, [1.0, 3.0, 7.0, 20.0], 16, 16)
    torch.cuda.synchronize()
