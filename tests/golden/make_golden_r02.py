#!/usr/bin/env python3
"""Round-2 additions to the reference vectors: tests/golden/reference_vectors_r02.{npz,json}.

Run from the repo root:  python tests/golden/make_golden_r02.py   (container only: imports /root/reference
through make_golden.py's stub modules; nothing of the reference is copied, only inputs / seeds / outputs).

Cases (VERDICT r01 "close the parity gaps at real model sizes"):
  * U-Net forwards for the size factors the first file does not cover (0.05, 0.1, 0.3, 0.4, 0.6, 0.7, 0.75,
    0.8, 0.9: odd channel counts 25 ... 230 of reference models.py:101-110), B=2, 16x16, all three cond modes;
  * configs[0]: the teacher's ``p_sample_loop`` at B=8, T=50, guidance 1.0, whole trajectory (utils/diffusion.py:160-212);
  * configs[4]: teacher ``p_sample_loop`` at 32x32, T=1000, guidance 7, B=2 -- the state after 950 steps and the
    50-step tail (entries 950..1000 of the 1001-entry trajectory), under a fixed global seed;
  * configs[2]: ``compare_trajectories`` of the teacher against the students of two mid sizes (0.6, 0.9) at
    guidance scales {1, 7} (2 samples, T=50), the grid cell values the heat maps are built from
    (scripts/analysis/analyze_trajectory_metrics.py:476-505).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg   # noqa: E402  (registers the torchvision / umap stubs, imports the reference)

import numpy as np   # noqa: E402
import torch         # noqa: E402

from distillation_trajectories_amd.synthetic import make_model, seeded_noise, state_dict_digest  # noqa: E402

NEW_SIZES = (0.05, 0.1, 0.3, 0.4, 0.6, 0.7, 0.75, 0.8, 0.9)


def main():
    out_npz, out_json = {}, {"torch": torch.__version__, "numpy": np.__version__}
    mdl = {sf: make_model(mg.ref_models.DiffusionUNet, mg.cfg(), sf) for sf in NEW_SIZES + (1.0,)}
    out_json["state_dict_sha256"] = {str(sf): state_dict_digest(m.state_dict()) for sf, m in mdl.items()}

    # ---------------------------------------------------------------- forwards at the remaining size factors
    cases = []
    for sf in NEW_SIZES:
        for cm in ("none", "zero", "one"):
            seed = 9600 + len(cases)
            x = seeded_noise(seed, (2, 3, 16, 16))
            t = torch.full((2,), (11 * len(cases)) % 50, dtype=torch.long)
            cond = None if cm == "none" else torch.full((2, 1), 0.0 if cm == "zero" else 1.0)
            with torch.no_grad():
                y = mdl[sf](x, t, cond)
            key = f"fwd_r02_{len(cases)}"
            out_npz[key] = y.numpy()
            cases.append(dict(key=key, sf=sf, h=16, b=2, cond=cm, seed=seed, t=int(t[0])))
    out_json["forward_cases"] = cases

    # ---------------------------------------------------------------- configs[0]: teacher p_sample_loop B=8 T=50
    c = mg.cfg(16, 50)
    torch.manual_seed(4321)
    with mg.quiet():
        img, tr = mg.ref_diff.p_sample_loop(mdl[1.0], (8, 3, 16, 16), 50, mg.ref_diff.get_diffusion_params(50, c),
                                            device=torch.device("cpu"), config=c, track_trajectory=True, guidance_scale=1.0)
    out_npz["config0_teacher"] = mg.stack(tr)
    out_json["config0_case"] = dict(key="config0_teacher", sf=1.0, w=1.0, sample_steps=50, timesteps=50, global_seed=4321, b=8)

    # ---------------------------------------------------------------- configs[4]: 32x32, T=1000, w=7, tail
    c = mg.cfg(32, 1000)
    torch.manual_seed(8642)
    with mg.quiet():
        img, tr = mg.ref_diff.p_sample_loop(mdl[1.0], (2, 3, 32, 32), 1000, mg.ref_diff.get_diffusion_params(1000, c),
                                            device=torch.device("cpu"), config=c, track_trajectory=True, guidance_scale=7.0)
    full = mg.stack(tr)
    assert full.shape[0] == 1001
    out_npz["config4_tail"] = full[950:]
    out_json["config4_case"] = dict(key="config4_tail", sf=1.0, w=7.0, sample_steps=1000, timesteps=1000, global_seed=8642,
                                    b=2, h=32, first_entry=950, finite=bool(np.isfinite(full).all()),
                                    max_abs=float(np.abs(full).max()))

    # ---------------------------------------------------------------- configs[2]: two mid-size grid cells
    c = mg.cfg(16, 50)
    cells = []
    for sf in (0.6, 0.9):
        with mg.quiet():
            res = mg.ref_engine.compare_trajectories(mdl[1.0], mdl[sf], c, guidance_scales=[1.0, 7.0], size_factor=sf, num_samples=2)
        cells.append(dict(student_sf=sf, teacher_sf=1.0, T=50, guidance_scales=[1.0, 7.0], num_samples=2,
                          result={k: {str(gs): mg.jsonable(v) for gs, v in d.items()} for k, d in res.items()}))
    out_json["grid_cells"] = cells

    np.savez_compressed(os.path.join(HERE, "reference_vectors_r02.npz"), **out_npz)
    with open(os.path.join(HERE, "reference_vectors_r02.json"), "w") as f:
        json.dump(out_json, f, indent=1)
    print("wrote", len(out_npz), "arrays;", os.path.getsize(os.path.join(HERE, "reference_vectors_r02.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
