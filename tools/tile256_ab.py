"""(Round-3 experiment, kernel variant since removed: the measurements are profiles/r03_tile256_direct_weights_*.txt.  Launch kind 6 --
the 256 x 64 tile with K = 16 steps and its weight fragments straight from global memory, no weight LDS, no barrier per tap -- tied
kind 3 (same K = 16 steps, weight tile in LDS) within +-5 % on every layer, and both are 25-35 % slower than kind 4 (K = 32 steps);
a K = 32 form needs two register sets of 2 x 2 x 3 fragments (96 VGPRs) on top of a kernel that already sits at the 256-VGPR cap.)

A/B of the 256 x 64 strip tile's variants (prec 3: K=16 steps, weight tile in LDS; 4: K=32 steps; 6: K=16 steps with the weight
fragments straight from global memory) per layer, interleaved rounds in one process.  Usage: tile256_ab.py [sf] [batch_total]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distillation_trajectories_amd import _hip, engine
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = Config(); cfg.image_size = 16
m = make_model(DiffusionUNet, cfg, sf).to("cuda:0")
h = engine.UNetHandle.for_module(m)
x = torch.randn(Bt // 2, 3, 16, 16, device="cuda:0")
tb = h.time_bias([10, 10], [_hip.COND_NONE, _hip.COND_ONE])
h.forward(x, tb, 2, Bt // 2, tune=False)
ws = h.workspace(Bt, 16, 16)
lib = _hip.load()
def t(j, slot, prec, fuse, reps=20):
    ms, fl = ctypes.c_float(), ctypes.c_double()
    st = lib.dt_unet_time_conv(h.h, Bt, 16, 16, j, slot, 256, 64, 1, prec, fuse, reps, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(), ctypes.byref(ms), ctypes.byref(fl))
    return (ms.value * 1e3, fl.value) if st == 0 and fl.value else (None, 0)
for j in range(8):
    for slot in (1, 2):
        for fuse in ((0, 1) if slot == 2 else (0,)):
            res = {p: [] for p in (3, 4, 6)}
            for rnd in range(5):
                for p in (3, 4, 6):
                    us, fl = t(j, slot, p, fuse)
                    if us: res[p].append((us, fl))
            if not res[3]: continue
            line = f"{engine.BLOCK_NAMES[j]:10s} {('', 'conv1', 'conv2')[slot]}{'+skip' if fuse else '     '} "
            for p in (3, 4, 6):
                if res[p]:
                    us = sorted(u for u, _ in res[p])
                    line += f"| prec {p}: min {us[0]:6.1f} med {us[len(us) // 2]:6.1f} us ({res[p][0][1] / us[len(us) // 2] / 1e6:4.0f} TF/s) "
            print(line, flush=True)
