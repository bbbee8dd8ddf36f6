#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by RUNNING THE REFERENCE in the build container.

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (read-only) and is therefore never run on the GPU box; only its
outputs travel.  The reference imports torchvision and umap at module top although the hot
path never touches them (config/config.py:2-3, analysis/__init__.py:5-13); both are absent
from this image, so empty stub modules are registered first.  Nothing from the reference is
copied: the fixtures are inputs, seeds and the reference's numeric outputs.

Everything is seeded; models are ``make_model`` random initialisations (synthetic.py), whose
state_dict digest is stored so the consumer can assert it rebuilt the very same weights.
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("MPLBACKEND", "Agg")
os.environ["CUDA_VISIBLE_DEVICES"] = ""          # the reference's CPU mode (scripts/run_on_cpu.py:27)
sys.dont_write_bytecode = True

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


tv = _stub("torchvision")
tv.transforms = _stub("torchvision.transforms")
tv.datasets = _stub("torchvision.datasets")
tv.models = _stub("torchvision.models", inception_v3=None)
tv.utils = _stub("torchvision.utils", make_grid=None)
_stub("umap", UMAP=None)
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np   # noqa: E402
import torch         # noqa: E402

torch.set_num_threads(1)      # fixtures are taken single-threaded so reduction order is fixed

import models as ref_models                                         # noqa: E402
from config.config import Config as RefConfig                       # noqa: E402
import utils.diffusion as ref_diff                                  # noqa: E402
import analysis.trajectory_engine as ref_engine                     # noqa: E402
from analysis.metrics.trajectory_metrics import compute_trajectory_metrics as ref_metrics   # noqa: E402
from analysis.metrics.time_dependent import analyze_time_dependent_distances as ref_timedep  # noqa: E402
from utils.trajectory_manager import TrajectoryManager as RefManager  # noqa: E402
from utils.metric_transformations import transform_metrics as ref_transform  # noqa: E402

from evaluation.metrics import compute_trajectory_divergence as ref_divergence      # noqa: E402
from analysis.noise_prediction.noise_analysis import calculate_noise_metrics as ref_noise_metrics  # noqa: E402

import analysis.metrics.fid_score as ref_fid                                         # noqa: E402

from distillation_trajectories_amd.synthetic import make_model, state_dict_digest, seeded_noise  # noqa: E402

quiet = lambda: contextlib.redirect_stdout(io.StringIO())   # noqa: E731


def cfg(image_size=16, timesteps=50):
    c = RefConfig()
    c.image_size = image_size
    c.timesteps = timesteps
    c.sample_steps = timesteps
    c.teacher_steps = timesteps
    c.student_steps = timesteps
    c.trajectory_dir = tempfile.mkdtemp(prefix="dt_golden_")
    return c


def jsonable(v):
    if isinstance(v, dict):
        return {str(k): jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, torch.Tensor):
        return v.tolist()
    return v


def stack(traj):
    ims = [e[0] if isinstance(e, tuple) else e for e in traj]
    return torch.stack(ims).numpy()


def main():
    out_npz = {}
    out_json = {"torch": torch.__version__, "numpy": np.__version__}

    # ---------------------------------------------------------------- models
    mdl = {}
    digests = {}
    for sf in (0.01, 0.2, 0.5, 1.0):
        mdl[sf] = make_model(ref_models.DiffusionUNet, cfg(), sf)
        digests[str(sf)] = state_dict_digest(mdl[sf].state_dict())
    out_json["state_dict_sha256"] = digests

    # ---------------------------------------------------------------- (1) U-Net forward
    cases = []
    for sf, hs, bs in ((0.01, (16, 32), (1, 2, 3)), (0.2, (16, 32), (1, 2, 3)), (0.5, (16,), (2,)), (1.0, (16,), (2,))):
        for h in hs:
            for b in bs:
                for cm in ("none", "zero", "one"):
                    seed = 9000 + len(cases)
                    x = seeded_noise(seed, (b, 3, h, h))
                    t = torch.full((b,), (7 * len(cases)) % 50, dtype=torch.long)
                    cond = None if cm == "none" else torch.full((b, 1), 0.0 if cm == "zero" else 1.0)
                    with torch.no_grad():
                        y = mdl[sf](x, t, cond)
                    key = f"fwd{len(cases)}"
                    out_npz[key] = y.numpy()
                    cases.append(dict(key=key, sf=sf, h=h, b=b, cond=cm, seed=seed, t=int(t[0])))
    out_json["forward_cases"] = cases

    # per-layer activations of the tiny model, B=2, cond=one, via forward hooks
    acts = {}
    hooks = []
    m = mdl[0.01]
    for name in ("enc1", "enc2", "enc3", "enc4", "bottleneck", "dec3", "dec2", "dec1"):
        hooks.append(getattr(m, name).register_forward_hook(lambda mod, i, o, n=name: acts.__setitem__(n, o.detach().numpy().copy())))
    x = seeded_noise(9500, (2, 3, 16, 16))
    with torch.no_grad():
        y = m(x, torch.tensor([13, 13]), torch.ones(2, 1))
    for hk in hooks:
        hk.remove()
    for n, a in acts.items():
        out_npz["act_" + n] = a
    out_npz["act_out"] = y.numpy()
    out_json["activation_case"] = dict(sf=0.01, seed=9500, b=2, h=16, t=13, cond="one")

    # ---------------------------------------------------------------- (2) schedules, (3) index sets
    for n in (20, 50, 100, 1000):
        p = ref_diff.get_diffusion_params(n)
        for k, v in p.items():
            out_npz[f"sched{n}_{k}"] = v.numpy()
    idx = {}
    for ss, nt in ((50, 50), (100, 50), (4000, 50), (50, 100), (100, 20), (100, 5)):
        # p_sample_loop has no standalone index function: capture the t values a recording model sees
        seen = []

        class Rec(torch.nn.Module):
            def __init__(s):
                super().__init__()
                s.p = torch.nn.Parameter(torch.zeros(1))

            def forward(s, x, t, cond=None):
                if cond is None:
                    seen.append(int(t[0]))
                return torch.zeros_like(x)
        c = cfg()
        c.timesteps = nt
        params = ref_diff.get_diffusion_params(ss)
        with quiet():
            ref_diff.p_sample_loop(Rec(), (1, 3, 4, 4), ss, params, device=torch.device("cpu"), config=c)
        idx[f"psample_{ss}_{nt}"] = seen
    out_json["index_sets"] = idx

    # ---------------------------------------------------------------- (4) generate_trajectory
    T = 50
    eng = []
    for sf, gs, seed in ((0.01, None, 42), (0.01, 1.0, 42), (0.01, 3.0, 43), (0.01, 20.0, 44), (0.2, 7.5, 45)):
        noise = seeded_noise(seed, (1, 3, 16, 16))
        with quiet():
            tr = ref_engine.generate_trajectory(mdl[sf], noise, T, torch.device("cpu"), seed=seed, guidance_scale=gs)
        key = f"engine{len(eng)}"
        out_npz[key] = stack(tr)
        eng.append(dict(key=key, sf=sf, gs=gs, seed=seed, T=T))
    # unseeded case: the loop consumes the global generator sequentially
    torch.manual_seed(777)
    noise = torch.randn(1, 3, 16, 16)
    with quiet():
        tr = ref_engine.generate_trajectory(mdl[0.01], noise, 20, torch.device("cpu"), seed=None, guidance_scale=3.0)
    out_npz["engine_unseeded"] = stack(tr)
    eng.append(dict(key="engine_unseeded", sf=0.01, gs=3.0, seed=None, T=20, global_seed=777))
    out_json["engine_cases"] = eng

    # ---------------------------------------------------------------- (5) p_sample_loop B=8
    ps = []
    for sf, w, ss, nt in ((0.01, 1.0, 50, 50), (0.01, 3.0, 50, 50), (0.2, 3.0, 100, 50)):
        c = cfg()
        c.timesteps = nt
        torch.manual_seed(1234)
        with quiet():
            img, tr = ref_diff.p_sample_loop(mdl[sf], (8, 3, 16, 16), ss, ref_diff.get_diffusion_params(ss, c),
                                             device=torch.device("cpu"), config=c, track_trajectory=True, guidance_scale=w)
        key = f"psample{len(ps)}"
        out_npz[key] = stack(tr)
        ps.append(dict(key=key, sf=sf, w=w, sample_steps=ss, timesteps=nt, global_seed=1234, b=8))
    out_json["psample_cases"] = ps

    # ---------------------------------------------------------------- (6) TrajectoryManager
    mg = []
    for (tsteps, ssteps, seed) in ((20, 20, 3), (20, 5, 4)):
        c = cfg(timesteps=20)
        c.sample_steps = 100
        c.teacher_steps, c.student_steps = tsteps, ssteps
        with quiet():
            man = RefManager(mdl[0.2], mdl[0.01], c, size_factor=0.01)
            tt, st = man.generate_trajectory(seed=seed)
        key = f"manager{len(mg)}"
        out_npz[key + "_teacher"] = stack(tt)
        out_npz[key + "_student"] = stack(st)
        mcase = dict(key=key, teacher_sf=0.2, student_sf=0.01, sample_steps=100, teacher_steps=tsteps,
                     student_steps=ssteps, seed=seed, teacher_t=[t for _, t in tt], student_t=[t for _, t in st])
        # (7b) metrics of these pairs, including the unequal-length interp1d path
        np.random.seed(100 + seed)
        mcase["metrics"] = jsonable(ref_metrics(tt, st, c))
        mcase["np_seed"] = 100 + seed
        mg.append(mcase)
    out_json["manager_cases"] = mg

    # ---------------------------------------------------------------- (7) metrics on engine pairs
    met = []
    for sf, gs, seed in ((0.01, 1.0, 42), (0.01, 3.0, 43), (0.01, 20.0, 44), (0.5, 3.0, 46)):
        noise = seeded_noise(seed, (1, 3, 16, 16))
        with quiet():
            a = ref_engine.generate_trajectory(mdl[1.0] if sf == 0.5 else mdl[0.2], noise, T, torch.device("cpu"), seed=seed, guidance_scale=gs)
            b = ref_engine.generate_trajectory(mdl[sf], noise, T, torch.device("cpu"), seed=seed, guidance_scale=gs)
        key = f"pair{len(met)}"
        out_npz[key + "_teacher"] = stack(a)
        out_npz[key + "_student"] = stack(b)
        met.append(dict(key=key, teacher_sf=1.0 if sf == 0.5 else 0.2, student_sf=sf, gs=gs, seed=seed, T=T,
                        metrics=jsonable(ref_metrics(a, b, cfg()))))
    # a forced-NaN trajectory_mse case: synthetic trajectories far apart (1000*mean step-MSE > 2)
    g = torch.Generator().manual_seed(5)
    a = [torch.randn(1, 3, 16, 16, generator=g) for _ in range(11)]
    b = [x + 0.2 * torch.randn(1, 3, 16, 16, generator=g) for x in a]
    out_npz["nan_teacher"], out_npz["nan_student"] = stack(a), stack(b)
    met.append(dict(key="nan", metrics=jsonable(ref_metrics(a, b, None))))
    # identical trajectories (zero distances: the 1.0 / 0 guards) with a repeated last state
    a = [torch.randn(1, 3, 16, 16, generator=g) for _ in range(6)]
    a.append(a[-1].clone())
    out_npz["same_teacher"] = stack(a)
    met.append(dict(key="same", metrics=jsonable(ref_metrics(a, [x.clone() for x in a], None))))
    # 32x32: E=3072 > 1000 exercises the np.random.choice sub-sampling of the Wasserstein term
    a = [torch.randn(1, 3, 32, 32, generator=g) for _ in range(9)]
    b = [x + 0.05 * torch.randn(1, 3, 32, 32, generator=g) for x in a]
    out_npz["big_teacher"], out_npz["big_student"] = stack(a), stack(b)
    np.random.seed(2024)
    met.append(dict(key="big", np_seed=2024, metrics=jsonable(ref_metrics(a, b, None))))
    # batched entries [B,C,H,W] with B=4 (what p_sample_loop trajectories look like)
    a = [torch.randn(4, 3, 16, 16, generator=g) for _ in range(8)]
    b = [x + 0.01 * torch.randn(4, 3, 16, 16, generator=g) for x in a]
    out_npz["batched_teacher"], out_npz["batched_student"] = stack(a), stack(b)
    np.random.seed(11)
    met.append(dict(key="batched", np_seed=11, metrics=jsonable(ref_metrics(a, b, None))))
    out_json["metric_cases"] = met

    # time-dependent distances over the four engine pairs
    tts = [[torch.from_numpy(x) for x in out_npz[f"pair{i}_teacher"]] for i in range(4)]
    sts = [[torch.from_numpy(x) for x in out_npz[f"pair{i}_student"]] for i in range(4)]
    with quiet():
        td = ref_timedep(tts, sts, cfg(), size_factor=0.5, save_dir=None)
    out_json["time_dependent"] = jsonable({k: v for k, v in td.items()})

    # ---------------------------------------------------------------- (8) compare_trajectories
    c = cfg(timesteps=20)
    with quiet():
        res = ref_engine.compare_trajectories(mdl[0.2], mdl[0.01], c, guidance_scales=[1.0, 3.0, 7.5], size_factor=0.01, num_samples=2)
    out_json["compare_case"] = dict(teacher_sf=0.2, student_sf=0.01, T=20, guidance_scales=[1.0, 3.0, 7.5], num_samples=2,
                                    result={k: {str(gs): jsonable(v) for gs, v in d.items()} for k, d in res.items()})

    # ---------------------------------------------------------------- (9) transform_metrics
    tcases = [(0.5, 0.3, -0.7, 0.6), (0.69, -1.5, 0.2, 0.9), (0.1, float("nan"), 1.0, 0.0), (0.0, 5.0, -1.0, 3.0)]
    out_json["transform_cases"] = [dict(args=list(a), result=jsonable(ref_transform(*a))) for a in tcases]

    # ---------------------------------------------------------------- (10) SURVEY §8f next rows
    # trajectory divergence on the stored manager pairs (equal and unequal lengths)
    div = []
    for c in mg:
        tt = [(torch.from_numpy(x), t) for x, t in zip(out_npz[c["key"] + "_teacher"], c["teacher_t"])]
        st = [(torch.from_numpy(x), t) for x, t in zip(out_npz[c["key"] + "_student"], c["student_t"])]
        div.append(dict(key=c["key"], result=jsonable(ref_divergence(tt, st))))
    out_json["divergence_cases"] = div
    # noise-prediction metrics on random prediction batches
    g = torch.Generator().manual_seed(77)
    tn = torch.randn(5, 3, 16, 16, generator=g)
    sn = tn + 0.3 * torch.randn(5, 3, 16, 16, generator=g)
    out_npz["noise_teacher"], out_npz["noise_student"] = tn.numpy(), sn.numpy()
    out_json["noise_metric_case"] = jsonable(ref_noise_metrics(tn, sn))
    # sample-averaged trajectories exactly as scripts/analysis/analyze_trajectories.py:437-486 builds them
    c = cfg(timesteps=12)
    scales, S, base_seed = [1.0, 3.0], 3, 42
    per = [{gs: [] for gs in scales} for _ in range(2)]
    for i in range(S):
        seed = base_seed + i
        torch.manual_seed(seed)
        np.random.seed(seed)
        noise = torch.randn(1, c.channels, c.image_size, c.image_size)
        for gs in scales:
            for k, model in enumerate((mdl[0.2], mdl[0.01])):
                with quiet():
                    per[k][gs].append(ref_engine.generate_trajectory(model, noise, c.timesteps, torch.device("cpu"), seed=seed, guidance_scale=gs))
    for k, who in enumerate(("teacher", "student")):
        for gs in scales:
            avg = [torch.mean(torch.stack([tr[t] for tr in per[k][gs]]), dim=0) for t in range(len(per[k][gs][0]))]
            out_npz[f"avg_{who}_{gs}"] = torch.stack(avg).numpy()
    out_json["average_case"] = dict(teacher_sf=0.2, student_sf=0.01, T=12, guidance_scales=scales, num_samples=S, base_seed=base_seed)

    # FID-input sampler (fid_score.py:199-318) and the FID formula on synthetic features
    c = cfg(timesteps=20)
    torch.manual_seed(2468)
    with quiet():
        out_npz["fid_samples"] = ref_fid.generate_samples(mdl[0.2], c, 3, torch.device("cpu")).numpy()
    x = seeded_noise(321, (2, 3, 16, 16))
    torch.manual_seed(1357)
    with quiet():
        out_npz["fid_loop"] = ref_fid.p_sample_loop(mdl[0.01], x.clone(), c).numpy()
    g = torch.Generator().manual_seed(99)
    f1 = torch.randn(40, 12, generator=g).double().numpy()
    f2 = (0.3 + 1.2 * torch.randn(36, 12, generator=g)).double().numpy()
    out_npz["fid_feat1"], out_npz["fid_feat2"] = f1, f2
    with quiet():
        out_json["fid_case"] = dict(sf_samples=0.2, sf_loop=0.01, T=20, seed_samples=2468, seed_loop=1357, x_seed=321,
                                    fid=float(ref_fid.calculate_fid(f1, f2)), fid_too_few=float(ref_fid.calculate_fid(f1[:1], f2)))

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out_npz)
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump(out_json, f, indent=1)
    print("wrote", len(out_npz), "arrays;", os.path.getsize(os.path.join(HERE, "reference_vectors.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
