"""A/B of the alternating wave priority between co-resident workgroups (DT_ABLATE=7 switches it off), strip kernel."""
import ctypes, os, sys
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
ws = h.workspace(512, 16, 16)
lib = _hip.load()
for prec in (3, 4):
    for rnd in range(3):
        for ab in ("7", "0"):
            os.environ["DT_ABLATE"] = ab
            row = f"prec {prec} {'no prio' if ab == '7' else 'alternating prio':18s}"
            for j, slot, name, sp in ((0, 2, "enc1.conv2", 1), (1, 1, "enc2.conv1", 1), (1, 2, "enc2.conv2", 1), (6, 1, "dec2.conv1", 4), (7, 1, "dec1.conv1", 2), (2, 1, "enc3.conv1", 2)):
                ms, fl = ctypes.c_float(), ctypes.c_double()
                lib.dt_unet_time_conv(h.h, 512, 16, 16, j, slot, 128, 128, sp, prec, 0, 20, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(), ctypes.byref(ms), ctypes.byref(fl))
                row += f" {name} {ms.value*1e3:6.1f}us ({fl.value/ms.value/1e9:4.0f})"
            print(row, flush=True)
