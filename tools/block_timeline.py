"""Per-workgroup timeline of strip-kernel launches (DT_ABLATE=8 records start / prologue end / loop end /
end per workgroup into the split-K slab): prologue, main loop, epilogue and tail against the event-timed launch.
Usage: block_timeline.py [sf=1.0] [batch_total=512]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distillation_trajectories_amd import _hip, engine
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 512
PREC = int(sys.argv[3]) if len(sys.argv) > 3 else 3      # 3: strip kernel, 4: strip kernel with K = 32 per step
cfg = Config(); cfg.image_size = 16
m = make_model(DiffusionUNet, cfg, sf).to("cuda:0")
h = engine.UNetHandle.for_module(m)
x = torch.randn(Bt // 2, 3, 16, 16, device="cuda:0")
tb = h.time_bias([10, 10], [_hip.COND_NONE, _hip.COND_ONE])
h.forward(x, tb, 2, Bt // 2, tune=False)
ws = h.workspace(Bt, 16, 16)
lib = _hip.load()
off, cp, oh, ow = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
lib.dt_unet_debug_activation(h.h, Bt, 16, 16, 8, ctypes.byref(off), ctypes.byref(cp), ctypes.byref(oh), ctypes.byref(ow))
d = h.dims
LAYERS = [(0, 2, "enc1.conv2", 16, d[0]), (1, 1, "enc2.conv1", 8, d[1]), (1, 2, "enc2.conv2", 8, d[1]), (2, 1, "enc3.conv1", 4, d[2]),
          (3, 1, "enc4.conv1", 2, d[3]), (6, 1, "dec2.conv1", 4, d[1]), (7, 1, "dec1.conv1", 8, d[0]), (7, 2, "dec1.conv2", 8, d[0])]
print(f"sf={sf} batch_total={Bt}; per workgroup, microseconds (100 MHz wall clock): median [min .. max]")
for j, slot, name, hw, cout in LAYERS:
    M = (Bt // 2 if j == 0 else Bt) * hw * hw      # enc1.conv2 covers the B images once for both CFG passes
    npad = (cout + 63) // 64 * 64
    for bm, bn in ((256, 64), (128, 128), (64, 64)):
        if npad % bn:
            continue
        grid = (M + bm - 1) // bm * (npad // bn)
        os.environ["DT_ABLATE"] = "0"
        ms0, fl = ctypes.c_float(), ctypes.c_double()
        lib.dt_unet_time_conv(h.h, Bt, 16, 16, j, slot, bm, bn, 1, PREC, 0, 5, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(), ctypes.byref(ms0), ctypes.byref(fl))
        os.environ["DT_ABLATE"] = "8"
        ms = ctypes.c_float()
        st = lib.dt_unet_time_conv(h.h, Bt, 16, 16, j, slot, bm, bn, 1, PREC, 0, 5, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(), ctypes.byref(ms), ctypes.byref(fl))
        torch.cuda.synchronize()
        if st != 0:
            print(f"{name} {bm}x{bn}: status {st}")
            continue
        rec = ws.view(torch.float32)[off.value: off.value + grid * 8].view(torch.int64).cpu().numpy().reshape(grid, 4) * 0.01
        t0 = rec[:, 0].min()
        pro, loop, epi = rec[:, 1] - rec[:, 0], rec[:, 2] - rec[:, 1], rec[:, 3] - rec[:, 2]
        f = lambda a: f"{np.median(a):6.1f} [{a.min():5.1f} .. {a.max():5.1f}]"
        print(f"{name:11s} {bm:3d}x{bn:<3d} {grid:5d} WGs ({grid / 256:4.1f}/CU)  launch {ms0.value * 1e3:6.1f} us ({fl.value / ms0.value / 1e9:4.0f} TF/s) | prologue {f(pro)} | loop {f(loop)} | "
              f"epilogue {f(epi)} | first start->last end {rec[:, 3].max() - t0:6.1f}, starts spread {rec[:, 0].max() - t0:5.1f}", flush=True)
os.environ["DT_ABLATE"] = "0"
