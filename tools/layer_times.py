"""Quick A/B table: the big strip-kernel layers under explicit (tile, split, arithmetic) choices, 20 back-to-back launches each."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distillation_trajectories_amd import _hip, engine
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cfg = Config(); cfg.image_size = 16
m = make_model(DiffusionUNet, cfg, sf).to("cuda:0")
h = engine.UNetHandle.for_module(m)
x = torch.randn(256, 3, 16, 16, device="cuda:0")
tb = h.time_bias([10, 10], [_hip.COND_NONE, _hip.COND_ONE])
h.forward(x, tb, 2, 256, tune=False)
ws = h.workspace(512, 16, 16)
lib = _hip.load()
LAYERS = ((0, 2, "enc1.conv2", 1), (1, 1, "enc2.conv1", 1), (1, 2, "enc2.conv2", 1), (2, 1, "enc3.conv1", 2), (6, 1, "dec2.conv1", 4), (6, 2, "dec2.conv2", 2), (7, 1, "dec1.conv1", 2), (7, 2, "dec1.conv2", 2), (3, 1, "enc4.conv1", 8), (5, 1, "dec3.conv1", 8))
for rnd in range(2):
    # (arithmetic code, tile, split override or None = the layer's usual split)
    for prec, bm, bn, spo in ((4, 128, 128, None), (4, 128, 64, None), (4, 64, 64, None), (4, 64, 64, 1), (4, 256, 64, None),
                              (5, 128, 64, None), (5, 128, 64, 1), (5, 64, 64, None), (5, 64, 64, 1), (5, 64, 64, 2), (5, 64, 64, 4)):
        row = f"prec {prec} {bm:3d}x{bn:<3d} s{spo if spo else '*'}"
        for j, slot, name, sp in LAYERS:
            sp = spo or sp
            ms, fl = ctypes.c_float(), ctypes.c_double()
            st = lib.dt_unet_time_conv(h.h, 512, 16, 16, j, slot, bm, bn, sp, prec, 0, 20, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(), ctypes.byref(ms), ctypes.byref(fl))
            row += f" {name} {ms.value*1e3:6.1f}us ({fl.value/ms.value/1e9:4.0f})" if st == 0 and ms.value > 0 else f" {name}   n/a      "
        print(row, flush=True)
