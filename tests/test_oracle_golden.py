"""The oracle (oracle/*.py) against the vectors captured from the reference itself.

Pins SURVEY.md §8c items (1)-(9).  Bit-exact where the oracle issues the same torch-CPU
kernels on the same shapes (single-threaded, like the capture); a few-ulp tolerance where a
result passes through float64 python arithmetic only.
"""
import math

import numpy as np
import pytest
import torch

from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.synthetic import seeded_noise, state_dict_digest
from oracle import metrics_ref, sampler_ref, unet_ref


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def eps_fn(model):
    sd = model.state_dict()
    return lambda x, t, c: unet_ref.unet_forward(sd, x, t, c)


def cond_of(mode, b):
    return None if mode == "none" else torch.full((b, 1), 0.0 if mode == "zero" else 1.0)


def test_same_weights_as_reference(golden, models):
    _, meta = golden
    for sf, digest in meta["state_dict_sha256"].items():
        assert state_dict_digest(models(float(sf)).state_dict()) == digest


def test_unet_forward_bit_exact(golden, models):
    arrays, meta = golden
    for c in meta["forward_cases"]:
        x = seeded_noise(c["seed"], (c["b"], 3, c["h"], c["h"]))
        t = torch.full((c["b"],), c["t"], dtype=torch.long)
        with torch.no_grad():
            y = eps_fn(models(c["sf"]))(x, t, cond_of(c["cond"], c["b"]))
        assert np.array_equal(y.numpy(), arrays[c["key"]]), c


def test_unet_activations(golden, models):
    arrays, meta = golden
    c = meta["activation_case"]
    x = seeded_noise(c["seed"], (c["b"], 3, c["h"], c["h"]))
    with torch.no_grad():
        y, acts = unet_ref.unet_forward(models(c["sf"]).state_dict(), x, torch.tensor([c["t"]] * c["b"]),
                                        torch.ones(c["b"], 1), return_activations=True)
    for name in unet_ref.BLOCKS:
        assert np.array_equal(acts[name].numpy(), arrays["act_" + name]), name
    assert np.array_equal(y.numpy(), arrays["act_out"])


def test_schedules_and_indices(golden):
    arrays, meta = golden
    for n in (20, 50, 100, 1000):
        p = sampler_ref.diffusion_params(n)
        for k, v in p.items():
            assert np.array_equal(v.numpy(), arrays[f"sched{n}_{k}"]), (n, k)
    for key, seen in meta["index_sets"].items():
        _, ss, nt = key.split("_")
        assert sampler_ref.psample_indices(int(ss), int(nt)) == seen
    for c in meta["manager_cases"]:
        assert sampler_ref.manager_indices(c["sample_steps"], c["teacher_steps"]) == c["teacher_t"]
        assert sampler_ref.manager_indices(c["sample_steps"], c["student_steps"]) == c["student_t"]


def test_engine_trajectories(golden, models):
    arrays, meta = golden
    for c in meta["engine_cases"]:
        if c["seed"] is None:
            torch.manual_seed(c["global_seed"])
            noise = torch.randn(1, 3, 16, 16)
        else:
            noise = seeded_noise(c["seed"], (1, 3, 16, 16))
        with torch.no_grad():
            tr = sampler_ref.generate_trajectory(eps_fn(models(c["sf"])), noise, c["T"], seed=c["seed"], guidance_scale=c["gs"])
        got = torch.stack(tr).numpy()
        assert got.shape == arrays[c["key"]].shape
        assert np.array_equal(got, arrays[c["key"]]), c
        assert np.array_equal(got[0], noise.numpy()) and np.array_equal(got[-1], got[-2])


def test_psample_loops(golden, models):
    arrays, meta = golden
    for c in meta["psample_cases"]:
        torch.manual_seed(c["global_seed"])
        params = sampler_ref.diffusion_params(c["sample_steps"])
        with torch.no_grad():
            _, tr = sampler_ref.p_sample_loop(eps_fn(models(c["sf"])), (c["b"], 3, 16, 16), c["sample_steps"], params,
                                              num_timesteps=c["timesteps"], guidance_scale=c["w"])
        assert np.array_equal(torch.stack(tr).numpy(), arrays[c["key"]]), c


def _close(a, b, rel=1e-12):
    if isinstance(b, list):
        assert len(a) == len(b)
        return all(_close(x, y, rel) for x, y in zip(a, b))
    a, b = float(a), float(b)
    if math.isnan(b):
        return math.isnan(a)
    return a == b or abs(a - b) <= rel * max(abs(a), abs(b))


def _check_metrics(got, want, rel=1e-12):
    assert set(got) == set(want)
    for k in want:
        assert _close(got[k], want[k], rel), (k, got[k], want[k])


def test_manager_trajectories_and_metrics(golden, models):
    arrays, meta = golden
    for c in meta["manager_cases"]:
        cfg = Config()
        cfg.image_size, cfg.sample_steps = 16, c["sample_steps"]
        cfg.teacher_steps, cfg.student_steps = c["teacher_steps"], c["student_steps"]
        with torch.no_grad():
            tt, st = sampler_ref.manager_generate(eps_fn(models(c["teacher_sf"])), eps_fn(models(c["student_sf"])), cfg, seed=c["seed"])
        assert np.array_equal(torch.stack([x for x, _ in tt]).numpy(), arrays[c["key"] + "_teacher"])
        assert np.array_equal(torch.stack([x for x, _ in st]).numpy(), arrays[c["key"] + "_student"])
        np.random.seed(c["np_seed"])
        _check_metrics(metrics_ref.compute_trajectory_metrics(tt, st), c["metrics"])


def test_metrics_on_stored_pairs(golden):
    arrays, meta = golden
    for c in meta["metric_cases"]:
        a = [torch.from_numpy(x) for x in arrays[c["key"] + "_teacher"]]
        b = a if c["key"] == "same" else [torch.from_numpy(x) for x in arrays[c["key"] + "_student"]]
        if "np_seed" in c:
            np.random.seed(c["np_seed"])
        got = metrics_ref.compute_trajectory_metrics(a, [x.clone() for x in b])
        _check_metrics(got, c["metrics"])
    nan_case = next(c for c in meta["metric_cases"] if c["key"] == "nan")
    assert math.isnan(nan_case["metrics"]["trajectory_mse"])      # the fixture really exercises the NaN branch


def test_time_dependent(golden):
    arrays, meta = golden
    tts = [[torch.from_numpy(x) for x in arrays[f"pair{i}_teacher"]] for i in range(4)]
    sts = [[torch.from_numpy(x) for x in arrays[f"pair{i}_student"]] for i in range(4)]
    got = metrics_ref.time_dependent_distances(tts, sts)
    want = meta["time_dependent"]
    for k in got:
        assert _close(got[k], want[k]), k


def test_compare_trajectories(golden, models):
    _, meta = golden
    c = meta["compare_case"]
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, c["T"]
    with torch.no_grad():
        res = sampler_ref.compare_trajectories(eps_fn(models(c["teacher_sf"])), eps_fn(models(c["student_sf"])), cfg,
                                               guidance_scales=c["guidance_scales"], num_samples=c["num_samples"])
    for side in ("teacher_metrics", "student_metrics"):
        for gs in c["guidance_scales"]:
            _check_metrics(res[side][gs], c["result"][side][str(gs)])


def test_transform_metrics(golden):
    _, meta = golden
    for c in meta["transform_cases"]:
        got = metrics_ref.transform_metrics(*c["args"])
        for k, v in c["result"].items():
            assert _close(got[k], v), (c, k)


# ------------------------------------------------------------------ SURVEY §8f next rows
def test_divergence_noise_metrics_and_sample_average(golden, models):
    arrays, meta = golden
    for c, d in zip(meta["manager_cases"], meta["divergence_cases"]):
        tt = [(torch.from_numpy(x), t) for x, t in zip(arrays[c["key"] + "_teacher"], c["teacher_t"])]
        st = [(torch.from_numpy(x), t) for x, t in zip(arrays[c["key"] + "_student"], c["student_t"])]
        got = metrics_ref.trajectory_divergence(tt, st)
        for k, v in d["result"].items():
            assert _close(got[k], v, 1e-12), (c["key"], k)
    got = metrics_ref.noise_metrics(torch.from_numpy(arrays["noise_teacher"]), torch.from_numpy(arrays["noise_student"]))
    for k, v in meta["noise_metric_case"].items():
        assert _close(got[k], v, 1e-12), k
    c = meta["average_case"]
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, c["T"]
    with torch.no_grad():
        ta, sa = sampler_ref.average_sample_trajectories(eps_fn(models(c["teacher_sf"])), eps_fn(models(c["student_sf"])), cfg,
                                                         c["guidance_scales"], c["num_samples"], c["base_seed"])
    for gs in c["guidance_scales"]:
        assert np.array_equal(torch.stack(ta[gs]).numpy(), arrays[f"avg_teacher_{gs}"])
        assert np.array_equal(torch.stack(sa[gs]).numpy(), arrays[f"avg_student_{gs}"])


def test_fid_sampler_and_formula(golden, models):
    arrays, meta = golden
    c = meta["fid_case"]
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, c["T"]
    fn_samples, fn_loop = eps_fn(models(c["sf_samples"])), eps_fn(models(c["sf_loop"]))   # (model construction re-seeds)
    x = seeded_noise(c["x_seed"], (2, 3, 16, 16))
    torch.manual_seed(c["seed_samples"])
    with torch.no_grad():
        got = sampler_ref.fid_generate_samples(fn_samples, cfg, 3)
    assert np.array_equal(got.numpy(), arrays["fid_samples"])
    torch.manual_seed(c["seed_loop"])
    with torch.no_grad():
        got = sampler_ref.fid_p_sample_loop(fn_loop, x, cfg)
    assert np.array_equal(got.numpy(), arrays["fid_loop"])
    assert _close(metrics_ref.calculate_fid(arrays["fid_feat1"], arrays["fid_feat2"]), c["fid"], 1e-12)
    assert metrics_ref.calculate_fid(arrays["fid_feat1"][:1], arrays["fid_feat2"]) == c["fid_too_few"] == 999.0


# ------------------------------------------------------------------ round-2 vectors: real model sizes
def test_r02_same_weights_as_reference(golden_r02, models):
    _, meta = golden_r02
    for sf, digest in meta["state_dict_sha256"].items():
        assert state_dict_digest(models(float(sf)).state_dict()) == digest


def test_r02_unet_forward_all_size_factors_bit_exact(golden_r02, models):
    arrays, meta = golden_r02
    for c in meta["forward_cases"]:
        x = seeded_noise(c["seed"], (c["b"], 3, c["h"], c["h"]))
        t = torch.full((c["b"],), c["t"], dtype=torch.long)
        with torch.no_grad():
            y = eps_fn(models(c["sf"]))(x, t, cond_of(c["cond"], c["b"]))
        assert np.array_equal(y.numpy(), arrays[c["key"]]), c


def test_r02_config0_teacher_loop_bit_exact(golden_r02, models):
    """configs[0]: teacher p_sample_loop, B=8, T=50, two passes per step (utils/diffusion.py:160-212)."""
    arrays, meta = golden_r02
    c = meta["config0_case"]
    torch.manual_seed(c["global_seed"])
    with torch.no_grad():
        _, tr = sampler_ref.p_sample_loop(eps_fn(models(c["sf"])), (c["b"], 3, 16, 16), c["sample_steps"],
                                          sampler_ref.diffusion_params(c["sample_steps"]), c["timesteps"], c["w"])
    assert np.array_equal(torch.stack(tr).numpy(), arrays[c["key"]])


def test_r02_config4_tail_from_stored_state(golden_r02, models):
    """configs[4]: the last 50 of 1000 steps at 32x32, w=7, restarted from the reference's state after 950 steps with
    the noise the reference drew (global generator stream: x_T, then one z per step)."""
    arrays, meta = golden_r02
    c = meta["config4_case"]
    assert c["finite"]
    tail = arrays[c["key"]]
    shape = (c["b"], 3, c["h"], c["h"])
    torch.manual_seed(c["global_seed"])
    torch.randn(shape)                                              # x_T
    zs = [torch.randn(shape) for _ in range(c["sample_steps"] - 1)]  # steps i = 999..1 draw; i = 0 does not
    params = sampler_ref.diffusion_params(c["sample_steps"])
    fn = eps_fn(models(c["sf"]))
    x = torch.from_numpy(tail[0])
    idx = sampler_ref.psample_indices(c["sample_steps"], c["timesteps"])
    with torch.no_grad():
        for k in range(c["first_entry"], c["sample_steps"]):
            i = idx[k]
            x = sampler_ref.p_sample(fn, x, torch.full((c["b"],), i, dtype=torch.long), i, params, c["w"],
                                     noise=zs[k] if i > 0 else None)
            assert np.array_equal(x.numpy(), tail[k - c["first_entry"] + 1]), k
            if k - c["first_entry"] >= 5:      # the oracle is the same torch code path: six steps pin it, the GPU test runs all
                break


# ------------------------------------------------------------------ round 3: batch aggregates, resize branches, 8 scales
def _check_value(got, want, what):
    if isinstance(want, list):
        assert len(got) == len(want), what
        for i, (g, w) in enumerate(zip(got, want)):
            _check_value(g, w, f"{what}[{i}]")
    else:
        assert _close(got, want), (what, got, want)


def test_r03_metrics_batch_aggregates(golden_r03, models):
    """utils/trajectory_manager.py:434-548 on three stored pairs (equal lengths, and 21 vs 6 states: interp1d path)."""
    _, meta = golden_r03
    for c in meta["batch_metric_cases"]:
        cfg = Config()
        cfg.image_size, cfg.sample_steps = 16, c["sample_steps"]
        cfg.teacher_steps, cfg.student_steps = c["teacher_steps"], c["student_steps"]
        with torch.no_grad():
            pairs = [sampler_ref.manager_generate(eps_fn(models(c["teacher_sf"])), eps_fn(models(c["student_sf"])), cfg, seed=i)
                     for i in range(c["num_samples"])]
        np.random.seed(c["np_seed"])
        got = sampler_ref.manager_metrics_batch(pairs, metrics_ref.compute_trajectory_metrics)
        want = c["result"]
        assert set(got) == set(want), set(got) ^ set(want)
        for k, w in want.items():
            _check_value(got[k], w, f"{c['student_steps']} student steps: {k}")


def test_r03_student_resize_branches(golden_r03, models):
    """A student with image_size 32 under config.image_size 16 (utils/trajectory_manager.py:120-122,153-163) and the metric
    function's own resize of an unresized 32 x 32 student list (analysis/metrics/trajectory_metrics.py:40-52)."""
    arrays, meta = golden_r03
    c = meta["resize_case"]
    cfg = Config()
    cfg.image_size, cfg.sample_steps = 16, c["sample_steps"]
    cfg.teacher_steps, cfg.student_steps = c["teacher_steps"], c["student_steps"]
    with torch.no_grad():
        tt, st = sampler_ref.manager_generate(eps_fn(models(c["teacher_sf"])), eps_fn(models(c["student_sf"])), cfg, seed=c["seed"],
                                              student_image_size=c["student_image_size"])
    assert [t for _, t in tt] == c["teacher_t"] and [t for _, t in st] == c["student_t"]
    assert np.array_equal(torch.stack([x for x, _ in tt]).numpy(), arrays["resize_teacher"])
    assert np.array_equal(torch.stack([x for x, _ in st]).numpy(), arrays["resize_student"])
    a = [torch.from_numpy(x) for x in arrays["resize_metric_teacher"]]
    b = [torch.from_numpy(x) for x in arrays["resize_metric_student32"]]
    np.random.seed(c["metric_np_seed"])
    _check_metrics(metrics_ref.compute_trajectory_metrics(a, b), c["metrics"])


def test_r03_eight_scale_grid_cell(golden_r03, models):
    """configs[3]: the eight guidance scales of scripts/analysis/analyze_trajectory_metrics.py:40-42 in one compare_trajectories call."""
    _, meta = golden_r03
    c = meta["grid8_cell"]
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, c["T"]
    with torch.no_grad():
        res = sampler_ref.compare_trajectories(eps_fn(models(c["teacher_sf"])), eps_fn(models(c["student_sf"])), cfg,
                                               guidance_scales=c["guidance_scales"], num_samples=c["num_samples"])
    for side in ("teacher_metrics", "student_metrics"):
        for gs in c["guidance_scales"]:
            _check_metrics(res[side][gs], c["result"][side][str(gs)])
