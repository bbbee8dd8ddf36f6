"""GPU checks at BASELINE.json's full sizes, through size-independent properties plus oracle spot checks.

configs[1]: teacher vs sf=0.5, 16x16, T=50, batch 256, p_sample_loop CFG.
configs[4]: 32x32 (CIFAR shape), teacher, CFG 7 -- forward parity and a short loop.
configs[2]/[3]: size sweep x CFG grid through grid.grid_metrics on one rank.
"""
import copy

import numpy as np
import pytest
import torch

from distillation_trajectories_amd import _hip, engine
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model
from oracle import metrics_ref, sampler_ref, unet_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _loop(h, x_T, z, idx, coef, w, B, E, H):
    from distillation_trajectories_amd._hip import COND_NONE, COND_ONE, RULE_PSAMPLE
    n = len(idx)
    traj = torch.empty(n + 1, B, E, device=DEV)
    traj[0].copy_(x_T)
    has_noise = [i > 0 for i in idx]
    shift, k = [], 0
    for f in has_noise:
        shift.append(k * B)
        k += int(f)
    tb = h.time_bias([i for i in idx for _ in (0, 1)], [COND_NONE, COND_ONE] * n)
    h.sample(RULE_PSAMPLE, traj, H, H, tb, 2, coef, has_noise, z=z, z_shift=shift, w_scalar=w)
    return traj


@pytest.fixture(scope="module")
def bench_pair(models):
    return [copy.deepcopy(models(sf)).to(DEV) for sf in (1.0, 0.5)]


def test_config1_full_batch_loop_properties(bench_pair, models):
    """B=256, T=50, two passes per step: permutation equivariance of whole trajectories (bit-exact) and
    the first steps of four samples against the CPU oracle."""
    from distillation_trajectories_amd.utils.diffusion import get_diffusion_params, psample_coefficients, timestep_indices
    B, T, H, E = 256, 50, 16, 768
    idx = timestep_indices(T, T)
    coef = psample_coefficients(get_diffusion_params(T), idx)
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn(B, E, generator=g)
    z = torch.randn((T - 1) * B, E, generator=g)
    perm = torch.randperm(B, generator=g)
    for m, sf in zip(bench_pair, (1.0, 0.5)):
        h = engine.UNetHandle.for_module(m)
        traj = _loop(h, x_T.to(DEV), z.to(DEV), idx, coef, 1.0, B, E, H)
        assert torch.isfinite(traj).all()
        zp = z.reshape(T - 1, B, E)[:, perm].reshape(-1, E).contiguous()
        traj_p = _loop(h, x_T[perm].to(DEV), zp.to(DEV), idx, coef, 1.0, B, E, H)
        assert torch.equal(traj_p, traj[:, perm.to(DEV)]), f"sf={sf}: rows of the batch must be independent"
        # oracle: 3 steps of samples 0..3
        sd = models(sf).state_dict()
        fn = lambda x, t, c: unet_ref.unet_forward(sd, x, t, c)    # noqa: E731
        params = sampler_ref.diffusion_params(T)
        x = x_T[:4].reshape(4, 3, H, H)
        with torch.no_grad():
            for s, i in enumerate(idx[:3]):
                x = sampler_ref.p_sample(fn, x, torch.full((4,), i, dtype=torch.long), i, params, 1.0,
                                         noise=z.reshape(T - 1, B, E)[s, :4].reshape(4, 3, H, H))
                got = traj[s + 1, :4].cpu().reshape(4, 3, H, H)
                assert torch.allclose(got, x, rtol=1e-4, atol=1e-4), (sf, s, (got - x).abs().max())


def test_config1_full_batch_metrics(bench_pair):
    """256 pairs x 51 states through the metric kernels: oracle spot checks + structural properties."""
    g = torch.Generator().manual_seed(6)
    n, B, E = 51, 256, 768
    X = torch.randn(n, B, E, generator=g).cumsum(0) * 0.05
    Y = X + 0.02 * torch.randn(n, B, E, generator=g)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    sums = engine.device_metric_sums(Xd, Yd).cpu().numpy()
    w1 = engine.device_wasserstein(Xd, Yd).cpu().numpy()
    swapped = engine.device_metric_sums(Yd, Xd).cpu().numpy()
    assert np.array_equal(swapped[..., 0], sums[..., 0]) and np.array_equal(swapped[..., 1], sums[..., 2])
    same = engine.device_metric_sums(Xd, Xd).cpu().numpy()
    assert not same[..., 0].any() and np.array_equal(same[..., 1], same[..., 2]) and not same[:, 0, 3].any()
    assert not engine.device_wasserstein(Xd, Xd).cpu().numpy().any()
    vec = engine.batch_scalar_metrics(sums, w1, 256, E)
    for b in (0, 97, 255):
        a = [X[i, b].reshape(1, 3, 16, 16) for i in range(n)]
        c = [Y[i, b].reshape(1, 3, 16, 16) for i in range(n)]
        want = metrics_ref.compute_trajectory_metrics(a, c)
        for k in engine.SCALAR_KEYS:
            if k == "trajectory_mse":
                continue
            assert abs(vec[k][b] - float(want[k])) <= 1e-5 * max(abs(float(want[k])), 1e-9), (b, k)


def test_config4_cifar_shape_forward_and_loop(models):
    """32x32x3, teacher, CFG w=7: forward vs oracle at B=2 and a 4-step loop at B=128 (finite, row-independent)."""
    from distillation_trajectories_amd.utils.diffusion import get_diffusion_params, psample_coefficients, timestep_indices
    m = copy.deepcopy(models(1.0)).to(DEV)
    sd = models(1.0).state_dict()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 3, 32, 32, generator=g)
    t = torch.tensor([999, 999])
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x, t, torch.ones(2, 1))
    got = m(x.to(DEV), t.to(DEV), torch.ones(2, 1, device=DEV))
    assert torch.allclose(got.cpu(), want, rtol=1e-4, atol=2e-5), (got.cpu() - want).abs().max()
    B, H, E, T = 128, 32, 3072, 1000
    idx = timestep_indices(T, T)[:4]
    coef = psample_coefficients(get_diffusion_params(T), idx)
    x_T = torch.randn(B, E, generator=g)
    z = torch.randn(4 * B, E, generator=g)
    h = engine.UNetHandle.for_module(m)
    traj = _loop(h, x_T.to(DEV), z.to(DEV), idx, coef, 7.0, B, E, H)
    assert torch.isfinite(traj).all()
    fn = lambda a, b, c: unet_ref.unet_forward(sd, a, b, c)    # noqa: E731
    with torch.no_grad():
        ref = sampler_ref.p_sample(fn, x_T[:2].reshape(2, 3, H, H), torch.full((2,), idx[0], dtype=torch.long), idx[0],
                                   sampler_ref.diffusion_params(T), 7.0, noise=z[:2].reshape(2, 3, H, H))
    assert torch.allclose(traj[1, :2].cpu().reshape(2, 3, H, H), ref, rtol=1e-4, atol=1e-4)


def test_config3_grid_one_rank_matches_compare_trajectories(models):
    """Size sweep x CFG grid through grid_metrics equals per-student compare_trajectories calls."""
    from distillation_trajectories_amd.analysis.trajectory_engine import compare_trajectories
    from distillation_trajectories_amd.grid import grid_metrics
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, 12
    teacher = copy.deepcopy(models(0.5)).to(DEV)
    students = [copy.deepcopy(models(sf)).to(DEV) for sf in (0.01, 0.2)]
    scales = [1.0, 3.0, 20.0]
    grid = grid_metrics(teacher, students, cfg, scales, num_samples=6, rank=0, world=1)
    for i, s in enumerate(students):
        res = compare_trajectories(teacher, s, cfg, guidance_scales=scales, num_samples=6)["student_metrics"]
        for gs in scales:
            for k, v in res[gs].items():
                a, b = grid[i][gs][k], v
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-9 * max(abs(b), 1e-12), (i, gs, k, a, b)
