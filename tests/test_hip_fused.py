"""GPU parity of the fused small-model kernel (csrc/dt_fused.hip: the whole forward, and the whole sampler loop, of a U-Net with
padded dims <= 32 / 64 at 16x16 as ONE launch, exact fp32 MFMA) against the CPU oracle, against the layered kernels, and
the properties the reference's seeded workflow relies on (row independence, run-to-run identical bits).

The reference vectors themselves (tests/golden: forwards, generate_trajectory, p_sample_loop, TrajectoryManager,
compare_trajectories -- almost all on size factors 0.01 and 0.2) run through this kernel by default and through the layered
kernels via the ``conv_path`` fixture of tests/test_hip_parity.py.  Tolerances: rtol 1e-4 (+ small atol) as there.
"""
import copy

import numpy as np
import pytest
import torch

from distillation_trajectories_amd import engine
from distillation_trajectories_amd._hip import COND_NONE, COND_ONE, COND_ZERO, RULE_ENGINE, RULE_MANAGER, RULE_PSAMPLE
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.models import DiffusionUNet
from distillation_trajectories_amd.synthetic import make_model, seeded_noise
from oracle import unet_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
E = 768


def assert_close(got, want, rtol=1e-4, atol=2e-5, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.all(err <= tol), f"{what}: max err {err.max():.3e} at tol {tol.flat[err.argmax()]:.3e}"


def small_model(sf, channels=3):
    cfg = Config()
    cfg.image_size, cfg.channels = 16, channels
    return make_model(DiffusionUNet, cfg, sf)


# size factor -> padded dims: 0.01 / 0.1 -> 16/32, 0.15 -> [19, 38..] = 32/48, 0.2 -> [25, 50..] = 32/64, 0.25 -> 32/64 exactly
@pytest.mark.parametrize("sf", [0.01, 0.1, 0.15, 0.2, 0.25])
def test_fused_forward_matches_oracle_and_layered(sf):
    m = small_model(sf)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    h = engine.UNetHandle.for_module(m)
    assert h.fused_active(16, 16) and not h.fused_active(32, 32)
    for B in (1, 2, 3, 8):                                  # odd row counts leave half a workgroup empty
        x = seeded_noise(600 + B, (B, 3, 16, 16))
        t = torch.randint(0, 50, (B,), generator=torch.Generator().manual_seed(B))
        cond = torch.rand(B, 1, generator=torch.Generator().manual_seed(10 + B))
        with torch.no_grad():
            want = unet_ref.unet_forward(sd, x, t, cond)
        got = m(x.to(DEV), t.to(DEV), cond.to(DEV)).cpu()
        assert_close(got.numpy(), want.numpy(), what=f"sf={sf} B={B} fused vs oracle")
        h.set_fused(False)
        lay = m(x.to(DEV), t.to(DEV), cond.to(DEV)).cpu()
        h.set_fused(True)
        assert_close(got.numpy(), lay.numpy(), rtol=2e-5, atol=5e-6, what=f"sf={sf} B={B} fused vs layered")


@pytest.mark.parametrize("channels", [1, 2])
def test_fused_forward_one_and_two_channel_images(channels):
    m = small_model(0.1, channels)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    assert engine.UNetHandle.for_module(m).fused_active(16, 16)
    x = seeded_noise(71, (3, channels, 16, 16))
    t = torch.tensor([0, 20, 49])
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x, t, None)
    assert_close(m(x.to(DEV), t.to(DEV)).cpu().numpy(), want.numpy(), what=f"C={channels}")


def test_fused_cfg_pair_and_mixed_batch_rows():
    """The row layouts of dt_unet_forward (n_pass = 2) and dt_unet_forward_mixed: [pass 0 of all images | pass 1 of the CFG
    images], one time-bias row per tb_div rows, against the oracle per row."""
    m = small_model(0.2)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    h = engine.UNetHandle.for_module(m)
    B, single = 6, 2
    x = seeded_noise(99, (B, 3, 16, 16))
    tb = h.time_bias([7, 7], [COND_NONE, COND_ONE])
    got = h.forward(x.to(DEV), tb, 2, B).cpu()
    with torch.no_grad():
        t = torch.full((B,), 7)
        want = torch.cat([unet_ref.unet_forward(sd, x, t, None), unet_ref.unet_forward(sd, x, t, torch.ones(B, 1))])
    assert_close(got.numpy(), want.numpy(), what="cfg pair")
    # mixed: images 0..1 one pass (cond None), images 2..5 two passes (cond 0, then cond 1): 10 rows, tb_div 2 -> 5 rows
    tbm = h.time_bias([3] * 5, [COND_NONE, COND_ZERO, COND_ZERO, COND_ONE, COND_ONE])
    gotm = h.forward_mixed(x.to(DEV), tbm, single, 2).cpu()
    with torch.no_grad():
        t = torch.full((B,), 3)
        wantm = torch.cat([unet_ref.unet_forward(sd, x[:single], t[:single], None),
                           unet_ref.unet_forward(sd, x[single:], t[single:], torch.zeros(B - single, 1)),
                           unet_ref.unet_forward(sd, x[single:], t[single:], torch.ones(B - single, 1))])
    assert_close(gotm.numpy(), wantm.numpy(), what="mixed batch")


def _loop_inputs(B, n_steps, seed):
    g = torch.Generator().manual_seed(seed)
    x_T = torch.randn(B, E, generator=g)
    z = torch.randn(n_steps * B + 7, E, generator=g)
    return x_T, z


@pytest.mark.parametrize("rule", [RULE_ENGINE, RULE_PSAMPLE, RULE_MANAGER])
@pytest.mark.parametrize("n_pass", [1, 2])
def test_fused_sampler_loop_matches_layered_loop(rule, n_pass):
    """dt_sample_trajectory through the ONE-launch loop against the per-timestep launch sequence of the layered kernels (which
    the golden loops pin): same coefficients, noise rows, per-row guidance scales; a dead ENGINE step (t = 0) included; more
    than 64 timesteps, so that the loop spans two launches."""
    m = small_model(0.1).to(DEV)
    h = engine.UNetHandle.for_module(m)
    B, n_steps = 5, 70
    x_T, z = _loop_inputs(B, n_steps, 31 + rule)
    ts = [n_steps - 1 - i for i in range(n_steps)]
    coef = [(0.98 + 0.0002 * i, 0.05 + 0.001 * i, 0.01 * (i % 5)) for i in range(n_steps)]
    if rule == RULE_MANAGER:
        coef = [(0.1, 0.95, 0.02)] * n_steps
    has_noise = [t > 0 for t in ts]
    modes = [COND_NONE] * n_steps if n_pass == 1 else [COND_NONE, COND_ONE] * n_steps
    tb = h.time_bias([t % 50 for t in ts for _ in range(n_pass)], modes)
    z_row = torch.tensor([3, 0, 4, 1, 2], dtype=torch.int32).to(DEV)
    w = torch.tensor([1.0, 3.0, 7.5, 0.0, 20.0]).to(DEV) if n_pass == 2 else None
    shift = [i * B for i in range(n_steps)]
    trajs = []
    for fused in (True, False):
        h.set_fused(fused)
        traj = torch.empty(n_steps + 1, B, E, device=DEV)
        traj[0] = x_T.to(DEV)
        h.sample(rule, traj, 16, 16, tb, n_pass, coef, has_noise, z=z.to(DEV), z_row=z_row, z_shift=shift, w=w, w_scalar=2.0)
        trajs.append(traj.cpu())
    h.set_fused(True)
    assert torch.isfinite(trajs[0]).all()
    scale = trajs[1].abs().max().item()
    assert (trajs[0] - trajs[1]).abs().max().item() <= 2e-5 * scale, "fused loop vs layered loop"
    if rule == RULE_ENGINE:
        assert torch.equal(trajs[0][-1], trajs[0][-2])          # t = 0: x is recorded unchanged (trajectory_engine.py:111-113)


def test_fused_mixed_sampler_loop_matches_layered_loop():
    """dt_sample_trajectory_mixed (the grid's launch sequence: single-pass block + CFG blocks with per-row scales)."""
    m = small_model(0.2).to(DEV)
    h = engine.UNetHandle.for_module(m)
    S, G, n_steps = 3, 2, 12
    B, rows = (1 + G) * S, (1 + 2 * G) * S
    x_T, z = _loop_inputs(B, n_steps, 77)
    ts = [n_steps - 1 - i for i in range(n_steps)]
    tb = h.time_bias([t for t in ts for _ in range(1 + 2 * G)], ([COND_NONE] + [COND_ZERO] * G + [COND_ONE] * G) * n_steps)
    coef = [(0.99, 0.04 + 0.002 * i, 0.03) for i in range(n_steps)]
    has_noise = [t > 0 for t in ts]
    z_row = (torch.arange(S, dtype=torch.int32).repeat(1 + G)).to(DEV)
    w = torch.cat([torch.zeros(S), torch.tensor([3.0, 7.0]).repeat_interleave(S)]).to(DEV)
    trajs = []
    for fused in (True, False):
        h.set_fused(fused)
        traj = torch.empty(n_steps + 1, B, E, device=DEV)
        traj[0] = x_T.to(DEV)
        h.sample_mixed(RULE_ENGINE, traj, 16, 16, tb, S, S, coef, has_noise, z.to(DEV), z_row, list(ts), w)
        trajs.append(traj.cpu())
    h.set_fused(True)
    assert rows == 2 * B - S
    scale = trajs[1].abs().max().item()
    assert (trajs[0] - trajs[1]).abs().max().item() <= 2e-5 * scale


def test_fused_rows_are_independent_and_runs_repeat_bit_for_bit():
    """A row's result must not depend on which rows share its workgroup or on the run (the library's determinism contract:
    tests/test_hip_fullsize.py checks the same for the layered kernels at batch 256)."""
    m = small_model(0.01).to(DEV)
    h = engine.UNetHandle.for_module(m)
    B, n_steps = 9, 10
    x_T, z = _loop_inputs(B, n_steps, 5)
    ts = list(range(n_steps - 1, -1, -1))
    tb = h.time_bias([t for t in ts for _ in (0, 1)], [COND_NONE, COND_ONE] * n_steps)
    coef = [(1.01, 0.03, 0.02)] * n_steps

    def run(order):
        traj = torch.empty(n_steps + 1, B, E, device=DEV)
        traj[0] = x_T[order].to(DEV)
        z_row = order.to(torch.int32).to(DEV)
        h.sample(RULE_PSAMPLE, traj, 16, 16, tb, 2, coef, [True] * n_steps, z=z.to(DEV), z_row=z_row,
                 z_shift=[i * B for i in range(n_steps)], w_scalar=3.0)
        return traj.cpu()
    ident = torch.arange(B)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    a, b, c = run(ident), run(ident), run(perm)
    assert torch.equal(a, b), "two runs of the same loop differ"
    assert torch.equal(c, a[:, perm]), "a row depends on its neighbours"
    # and a single row alone (half-empty workgroup) gives the same bits as inside the batch
    one = torch.empty(n_steps + 1, 1, E, device=DEV)
    one[0] = x_T[4:5].to(DEV)
    h.sample(RULE_PSAMPLE, one, 16, 16, tb, 2, coef, [True] * n_steps, z=z.to(DEV), z_row=torch.tensor([4], dtype=torch.int32).to(DEV),
             z_shift=[i * B for i in range(n_steps)], w_scalar=3.0)
    assert torch.equal(one.cpu()[:, 0], a[:, 4])


def test_fused_forward_equals_first_loop_step():
    """forward + dt_cfg_update (two launches) against the first step of the fused loop: the update arithmetic is shared
    (dt_update_math.h), so equal predictions give equal bits."""
    m = small_model(0.2).to(DEV)
    h = engine.UNetHandle.for_module(m)
    B = 4
    x_T, z = _loop_inputs(B, 1, 8)
    tb = h.time_bias([30, 30], [COND_NONE, COND_ONE])
    coef = (1.02, 0.07, 0.05)
    traj = torch.empty(2, B, E, device=DEV)
    traj[0] = x_T.to(DEV)
    h.sample(RULE_PSAMPLE, traj, 16, 16, tb, 2, [coef], [True], z=z.to(DEV), w_scalar=2.5)
    eps = h.forward(x_T.reshape(B, 3, 16, 16).to(DEV), tb, 2, B)
    want = engine.cfg_update(RULE_PSAMPLE, x_T.to(DEV), eps[:B].reshape(B, E), eps[B:].reshape(B, E), z.to(DEV), coef, True, w_scalar=2.5)
    assert torch.equal(traj[1], want)


def test_fused_path_switches():
    """set_fused / set_head_fusion(False) / set_conv_choice select the layered kernels; models that do not qualify never fuse."""
    m = small_model(0.1).to(DEV)
    h = engine.UNetHandle.for_module(m)
    assert h.fused_active(16, 16)
    h.set_head_fusion(False)
    assert not h.fused_active(16, 16)
    h.set_head_fusion(True)
    h.set_conv_choice(4, 16, 16, 1, 1, 64, 64, 1, 3, 0)
    assert not h.fused_active(16, 16)
    h.set_fused(True)
    assert h.fused_active(16, 16)
    big = engine.UNetHandle.for_module(small_model(0.5).to(DEV))
    big.set_fused(True)
    assert not big.fused_active(16, 16)


def test_fused_randomised_models_batches_rules():
    """tools/fuzz_fused.py: random small models (padded dims 16/32, 32/48, 32/64 from odd real channel counts, 1-3 image
    channels), batches, update rules, plain / CFG / mixed batches, loops of 1..80 timesteps: forwards against the oracle, fused
    loops against the layered per-timestep launches (a child process: the fuzzer is also a command-line tool)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_fused.py"), "14", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "worst forward" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("sf", [0.1, 0.2])
def test_fused_non_finite_image_stays_in_its_rows(sf):
    """An Inf pixel in one image: that image's rows are non-finite, every other row -- also the one sharing its workgroup, whose
    pixels sit in the same MFMA tiles at the 2x2 / 1x1 levels -- is bit-identical to a clean run (forward and sampler loop)."""
    m = small_model(sf).to(DEV)
    h = engine.UNetHandle.for_module(m)
    B = 6
    x = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(8)).to(DEV)
    tb = h.time_bias([30], [COND_NONE])
    clean = h.forward(x, tb, 1, B).clone()
    bad = x.clone()
    bad[2, 1, 7, 9] = float("inf")                              # rows 2 and 3 share a workgroup (G = 2 rows)
    got = h.forward(bad, tb, 1, B)
    keep = [0, 1, 3, 4, 5]
    assert torch.equal(got[keep], clean[keep]) and not torch.isfinite(got[2]).all() and torch.isfinite(clean).all()
    n_steps = 3
    ztab = torch.randn(n_steps * B, E, generator=torch.Generator().manual_seed(9)).to(DEV)
    tbl = h.time_bias([30, 30, 20, 20, 10, 10], [COND_NONE, COND_ONE] * n_steps)
    out = []
    for x0 in (x, bad):
        traj = torch.empty(n_steps + 1, B, E, device=DEV)
        traj[0] = x0.reshape(B, E)
        h.sample(RULE_PSAMPLE, traj, 16, 16, tbl, 2, [(1.0, 0.05, 0.02)] * n_steps, [True] * n_steps, z=ztab,
                 z_shift=[i * B for i in range(n_steps)], w_scalar=3.0)
        out.append(traj.clone())
    assert torch.equal(out[1][:, keep], out[0][:, keep]) and not torch.isfinite(out[1][-1, 2]).all()
