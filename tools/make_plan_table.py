"""Measure the launch plans of the BASELINE.json shapes on this MI355X and write the committed per-arch table
distillation_trajectories_amd/plans/gfx950.json (see engine._Plans: by default every process takes a shape's plan from this
table, so results do not depend on which process ran the autotuner).  Run on the GPU box from the repo root:
    python tools/make_plan_table.py [out.json]
Shapes: configs[1] (teacher / student, 2 x 256 rows), configs[2] (11 sizes + teacher, mixed batch 64 + 3 x 64 images),
configs[3] (8 guidance scales: 64 + 7 x 64 images), configs[4] (32 x 32, 2 x 128 rows)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from distillation_trajectories_amd import _hip, engine
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model

out = sys.argv[1] if len(sys.argv) > 1 else engine.PLAN_TABLE
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
plans, t_all = {}, time.perf_counter()


def tune(h, H, B, n_pass, single=0):
    rows = 2 * B - single if single else n_pass * B
    x = torch.randn(B, 3, H, H, device=dev)
    if single:
        G = (B - single) // single
        tb = h.time_bias([10] * (1 + 2 * G), [_hip.COND_NONE] + [_hip.COND_ZERO] * G + [_hip.COND_ONE] * G)
        h.forward_mixed(x, tb, single, single, tune=True)
    else:
        tb = h.time_bias([10] * n_pass, [_hip.COND_NONE, _hip.COND_ONE][:n_pass])
        h.forward(x, tb, n_pass, B, tune=True)
    key = h.plan_key(rows, H, H, B, single)
    plans[key] = h._read_plan(rows, H, H)
    print(f"{key}: {h._plans[(rows, H, H, B, single)]}", flush=True)


for H in (16, 32):
    cfg = Config(); cfg.image_size = H
    for sf in ([1.0, 0.5] + [s for s in bench.SIZES if s not in (1.0, 0.5)]) if H == 16 else [1.0]:
        m = make_model(DiffusionUNet, cfg, sf).to(dev)
        h = engine.UNetHandle.for_module(m)
        if H == 32:
            tune(h, 32, 128, 2)                       # configs[4]
            continue
        if sf in (1.0, 0.5):
            tune(h, 16, 256, 2)                       # configs[1]
        tune(h, 16, 4 * 64, 2, single=64)             # configs[2]: CFG {1, 3, 7, 20} x 64 samples
        tune(h, 16, 8 * 64, 2, single=64)             # configs[3]: 8 scales x 64 samples per GPU
        del h, m
meta = {"arch": "gfx950", "abi": _hip.ABI_VERSION, "device": torch.cuda.get_device_name(0), "seconds": round(time.perf_counter() - t_all, 1),
        "made_by": "tools/make_plan_table.py", "entry": "[block, slot (0 skip, 1 conv1, 2 conv2), bm, bn, splits, launch kind, skip fused]"}
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    json.dump({"meta": meta, "plans": plans}, f, indent=0, sort_keys=True)
print("wrote", out, len(plans), "plans in", meta["seconds"], "s")
