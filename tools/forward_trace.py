"""Launch-by-launch timeline of one mixed-batch forward (configs[2] shape: 64 single-pass + 192 CFG images = 448 rows).

  run:   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ft -o p -- python3 tools/forward_trace.py run 0.5
  read:  python3 tools/forward_trace.py read gpurun_out/ft
The run brackets its last forward with profile markers; `read` prints every dispatch between them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(sf, S=64, G=3):
    import torch
    from distillation_trajectories_amd import _hip, engine
    from distillation_trajectories_amd.config import Config
    from distillation_trajectories_amd.models import DiffusionUNet
    from distillation_trajectories_amd.synthetic import make_model
    cfg = Config(); cfg.image_size = 16
    m = make_model(DiffusionUNet, cfg, sf).to("cuda:0")
    h = engine.UNetHandle.for_module(m)
    B = (1 + G) * S
    x = torch.randn(B, 3, 16, 16, device="cuda:0")
    tb = h.time_bias([10] * (1 + 2 * G), [_hip.COND_NONE] + [_hip.COND_ZERO] * G + [_hip.COND_ONE] * G)
    h.forward_mixed(x, tb, S, S)                      # tunes (unless DT_AUTOTUNE=0)
    for _ in range(20):
        h.forward_mixed(x, tb, S, S, tune=False)
    torch.cuda.synchronize()
    _hip.profile_marker(1)
    h.forward_mixed(x, tb, S, S, tune=False)
    _hip.profile_marker(2)
    torch.cuda.synchronize()
    print([(c[0], c[1], c[2], c[3], c[4], c[5]) for c in h.conv_choices(2 * B - S, 16, 16)])


def read(d):
    import csv, glob
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"),
                   r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?")) for r in csv.DictReader(open(f)))
    marks = [i for i, r in enumerate(rows) if "profile_marker" in r[2]]
    lo, hi = marks[-2], marks[-1]
    t0 = rows[lo][1]
    prev_end = t0
    tot = 0
    for s, e, name, grid, wg, lds, vgpr in rows[lo + 1: hi]:
        short = name.split("(")[0].replace("void dt::", "").replace("dt::", "")[:70]
        try:
            nwg = int(grid) // int(wg)
        except ValueError:
            nwg = "?"
        print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f} us  gap {(s - prev_end) / 1e3:5.1f}  wgs {nwg:>6}  lds {lds:>6} vgpr {vgpr:>4}  {short}")
        prev_end = e
        tot += e - s
    print(f"span {(rows[hi][0] - t0) / 1e3:.1f} us, sum of kernel durations {tot / 1e3:.1f} us, {hi - lo - 1} launches")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(float(sys.argv[2]))
    else:
        read(sys.argv[2])
