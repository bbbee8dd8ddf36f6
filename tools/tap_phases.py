"""Where a strip-kernel step spends its cycles (DT_ABLATE=9 diagnostic build: s_memtime stamps, per-wave sums in the
split-K slab).  Phases per step: fragment reads + MFMA issue | wait for the next tap's weight loads | their LDS
writes | barrier wait.  Read the SHARES, not the total (the stamps forbid overlaps the real kernel has)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
PREC = int(sys.argv[1]) if len(sys.argv) > 1 else 3     # 3: strip kernel
off, cp, oh, ow = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
lib.dt_unet_debug_activation(h.h, 512, 16, 16, 8, ctypes.byref(off), ctypes.byref(cp), ctypes.byref(oh), ctypes.byref(ow))
for j, slot, name, grid in ((7, 1, "dec1.conv1 (1 WG/CU)", 256), (1, 2, "enc2.conv2 (2 WG/CU)", 512), (0, 2, "enc1.conv2 (4 WG/CU)", 1024)):
    for ab in ("0", "9"):
        os.environ["DT_ABLATE"] = ab
        ms, fl = ctypes.c_float(), ctypes.c_double()
        st = lib.dt_unet_time_conv(h.h, 512, 16, 16, j, slot, 128, 128, 1, PREC, 0, 5, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(), ctypes.byref(ms), ctypes.byref(fl))
        torch.cuda.synchronize()
        if ab == "0":
            base = ms.value
    rec = ws.view(torch.float32)[off.value: off.value + grid * 4 * 16].view(torch.int64).cpu().numpy().reshape(grid * 4, 8)
    steps = rec[:, 4].astype(np.float64)
    per = rec[:, :4] / steps[:, None]
    tot = per.sum(1)
    print(f"{name}: launch {base*1e3:.1f} us plain, {ms.value*1e3:.1f} us stamped; {int(steps[0])} steps per wave; cycles per step (median over waves):")
    for k, lab in enumerate(("reads + MFMA issue", "weight-load wait", "weight LDS writes", "barrier wait")):
        print(f"   {lab:20s} {np.median(per[:, k]):7.0f}   (p10 {np.percentile(per[:, k],10):6.0f}  p90 {np.percentile(per[:, k],90):6.0f})")
    print(f"   {'total':20s} {np.median(tot):7.0f}   (MFMA alone: {24*32})", flush=True)
