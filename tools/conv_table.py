#!/usr/bin/env python3
"""Per-layer convolution timing table on one MI355X: every conv launch of a forward shape under chosen
(tile, tap split, arithmetic) settings, via dt_unet_time_conv.  Usage: conv_table.py [sf] [batch_total]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # noqa: E402

from distillation_trajectories_amd import _hip, engine   # noqa: E402
from distillation_trajectories_amd.config import Config   # noqa: E402
from distillation_trajectories_amd.models import DiffusionUNet   # noqa: E402
from distillation_trajectories_amd.synthetic import make_model   # noqa: E402

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = 16
cfg = Config(); cfg.image_size = H
m = make_model(DiffusionUNet, cfg, sf).to("cuda:0")
h = engine.UNetHandle.for_module(m)
x = torch.randn(Bt // 2, 3, H, H, device="cuda:0")
tb = h.time_bias([10, 10], [_hip.COND_NONE, _hip.COND_ONE])
h.forward(x, tb, 2, Bt // 2, tune=False)       # real activations in the workspace
ws = h.workspace(Bt, H, H)
lib = _hip.load()
names = engine.BLOCK_NAMES
print(f"sf={sf} batch_total={Bt}: us (TF/s fp32-equivalent) per launch; prec/tile/splits")
configs = [(p, bm, bn, sp) for p in (0, 1) for (bm, bn) in ((128, 128), (128, 64), (64, 128), (64, 64)) for sp in (1, 2, 3, 4, 8, 9)]
configs += [(pr, bm, bn, sp) for pr in (3, 4) for (bm, bn) in ((128, 128), (128, 64), (64, 128), (64, 64)) for sp in (1, 2, 4, 8)]
for j in range(8):
    for slot in range(3):
        best = []
        for (prec, bm, bn, sp) in configs:
            ms, fl = ctypes.c_float(), ctypes.c_double()
            st = lib.dt_unet_time_conv(h.h, Bt, H, H, j, slot, bm, bn, sp, prec, 0, 10, _hip.ptr(ws), ws.numel(),
                                       _hip.stream_ptr(), ctypes.byref(ms), ctypes.byref(fl))
            if st != 0 or fl.value == 0:
                continue
            best.append((ms.value, prec, bm, bn, sp, fl.value))
        if not best:
            continue
        best.sort()
        line = f"{names[j]:10s} {('skip','conv1','conv2')[slot]:5s} GF={best[0][5]/1e9:6.2f} | "
        for ms, prec, bm, bn, sp, fl in best[:4]:
            line += f"{('f32','b6','dma','strip','strip32','pipe')[prec]}/{bm}x{bn}/s{sp}: {ms*1e3:6.1f}us ({fl/ms/1e9:4.0f}) | "
        for prec in (0, 1, 3, 4):      # best of each arithmetic
            cand = [b for b in best if b[1] == prec]
            if cand:
                ms, _, bm, bn, sp, fl = cand[0]
                line += f" {('f32','b6','dma','strip','strip32','pipe')[prec]}={fl/ms/1e9:4.0f}"
        print(line, flush=True)
