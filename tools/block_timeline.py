"""Per-workgroup timeline of one strip-kernel launch (DT_ABLATE=8 records start / loop end / end / hw id per
workgroup into the split-K slab): dispatch ramp, steady state, epilogue and tail against the event-timed launch."""
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
off, cp, oh, ow = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
lib.dt_unet_debug_activation(h.h, 512, 16, 16, 8, ctypes.byref(off), ctypes.byref(cp), ctypes.byref(oh), ctypes.byref(ow))
os.environ["DT_ABLATE"] = "8"
for j, slot, name, grid in ((1, 2, "enc2.conv2", 512), (7, 1, "dec1.conv1 (s1)", 256), (0, 2, "enc1.conv2", 1024)):
    ms, fl = ctypes.c_float(), ctypes.c_double()
    st = lib.dt_unet_time_conv(h.h, 512, 16, 16, j, slot, 128, 128, 1, 3, 0, 5, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(), ctypes.byref(ms), ctypes.byref(fl))
    torch.cuda.synchronize()
    rec = ws.view(torch.float32)[off.value: off.value + grid * 8].view(torch.int64).cpu().numpy().reshape(grid, 4)
    t0 = rec[:, 0].min()
    start, loop_end, end = (rec[:, 0] - t0) * 0.01, (rec[:, 1] - t0) * 0.01, (rec[:, 2] - t0) * 0.01    # us (100 MHz clock)
    hw = rec[:, 3] & 0xffffffff
    xcc = rec[:, 3] >> 32
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7   # gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    key = xcc * 1000 + se * 100 + ((hw >> 12) & 1) * 16 + cu
    per_cu = np.unique(key, return_counts=True)[1]
    print(f"{name}: event-timed launch {ms.value*1e3:.1f} us, {grid} workgroups on {len(per_cu)} CUs "
          f"(per CU: min {per_cu.min()} max {per_cu.max()})")
    print(f"  start  : median {np.median(start):6.1f}  p95 {np.percentile(start,95):6.1f}  max {start.max():6.1f} us after the first")
    print(f"  loop   : median {np.median(loop_end-start):6.1f}  min {np.min(loop_end-start):6.1f}  max {np.max(loop_end-start):6.1f} us")
    print(f"  epilog : median {np.median(end-loop_end):6.1f}  max {np.max(end-loop_end):6.1f} us")
    print(f"  last workgroup ends {end.max():6.1f} us after the first one started; median end {np.median(end):6.1f}", flush=True)
    loop = loop_end - start
    print("  loop time by XCD (median / max us):", "  ".join(f"{int(k)}: {np.median(loop[xcc == k]):.0f}/{loop[xcc == k].max():.0f}" for k in np.unique(xcc)))
    order = np.arange(grid)
    print("  loop time by block-id quarter (median):", "  ".join(f"{np.median(loop[(order >= a) & (order < a + grid // 4)]):.0f}" for a in range(0, grid, grid // 4)))
