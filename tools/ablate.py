#!/usr/bin/env python3
"""Timing ablations of the split-bf16 conv kernel on a few big layers (results are wrong by design)."""
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
layers = [(0, 2, "enc1.conv2"), (1, 2, "enc2.conv2"), (7, 1, "dec1.conv1"), (2, 1, "enc3.conv1")]
PREC = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # 1: plain split-bf16 kernel, 3: strip kernel
SPLIT = int(sys.argv[2]) if len(sys.argv) > 2 else 0         # 0: the usual per-layer splits, else this split everywhere
NAMES = {1: ((0, "full"), (1, "no barrier"), (2, "no LDS fragment reads"), (3, "no MFMA"), (4, "no global loads"), (5, "no split VALU")),
         3: ((0, "full"), (1, "no per-tap barrier"), (2, "no weight staging"), (3, "no MFMA"), (4, "no fragment reads"), (5, "no strip re-staging"), (6, "cache-hot weight loads"))}
for ab, name in ((0, "(warm-up pass)"),) + NAMES[PREC] + ((0, "full again"),):
    os.environ["DT_ABLATE"] = str(ab)
    row = f"{name:28s}"
    for j, slot, lname in layers:
        ms, fl = ctypes.c_float(), ctypes.c_double()
        sp = SPLIT or ((3 if PREC == 1 else 2) if lname in ("dec1.conv1", "enc3.conv1") else 1)
        st = lib.dt_unet_time_conv(h.h, 512, 16, 16, j, slot, 128, 128, sp, PREC, 0, 10, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(),
                                   ctypes.byref(ms), ctypes.byref(fl))
        row += f" {lname} {ms.value*1e3:7.1f}us" if st == 0 else f" {lname} ERR{st}"
    print(row, flush=True)
