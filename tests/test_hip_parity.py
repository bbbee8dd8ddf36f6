"""GPU parity: the hand-written HIP path (through the C ABI) against the oracle and the golden vectors.

Tolerances: integer/index work bit-exact; the fused update bit-exact given equal inputs; U-Net
activations and trajectories within fp32 re-association noise (rtol 1e-4 with a small atol);
metric values within 1e-4 relative (BASELINE.json north star), `trajectory_mse` checked on its
well-conditioned pre-transform quantity plus identical NaN positions.
"""
import math

import numpy as np
import pytest
import torch

from distillation_trajectories_amd import _hip, engine
from distillation_trajectories_amd.config import Config
from distillation_trajectories_amd.synthetic import seeded_noise
from oracle import metrics_ref, sampler_ref, unet_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cond_of(mode, b):
    return None if mode == "none" else torch.full((b, 1), 0.0 if mode == "zero" else 1.0)


@pytest.fixture(scope="module")
def gpu_models(models):
    cache = {}

    def get(sf):
        if sf not in cache:
            import copy
            cache[sf] = copy.deepcopy(models(sf)).to(DEV)
        return cache[sf]
    return get


@pytest.fixture(params=["fused", "layered"])
def conv_path(request):
    """The small models of the golden vectors (sf 0.01 / 0.2 at 16x16) run the fused whole-forward kernel by default; the
    tests that carry this fixture also run them through the layered kernels, which serve every other shape."""
    engine.set_fused_default(request.param == "fused")
    yield request.param
    engine.set_fused_default(True)


def assert_close(got, want, rtol=1e-4, atol=2e-5, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.all(err <= tol), f"{what}: max err {err.max():.3e} at tol {tol.flat[err.argmax()]:.3e}"


# ------------------------------------------------------------------ the library is the thing under test
def test_library_loaded_and_on_torch_runtime():
    lib = _hip.load()
    assert lib.dt_abi_version() == _hip.ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libdt_hip.so" in maps
    hips = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(hips) == 1, f"two HIP runtimes in one process: {hips}"


def test_no_cpu_fallback(models):
    m = models(0.01)
    with pytest.raises(_hip.HipLibraryError):
        m(torch.zeros(1, 3, 16, 16), torch.tensor([3]))


# ------------------------------------------------------------------ U-Net forward
def test_unet_forward_golden(golden, gpu_models, conv_path):
    arrays, meta = golden
    for c in meta["forward_cases"]:
        x = seeded_noise(c["seed"], (c["b"], 3, c["h"], c["h"]))
        t = torch.full((c["b"],), c["t"], dtype=torch.long)
        cond = cond_of(c["cond"], c["b"])
        y = gpu_models(c["sf"])(x.to(DEV), t.to(DEV), None if cond is None else cond.to(DEV))
        assert_close(y.cpu().numpy(), arrays[c["key"]], what=str(c))


def test_unet_block_activations_golden(golden, gpu_models):
    arrays, meta = golden
    c = meta["activation_case"]
    m = gpu_models(c["sf"])
    x = seeded_noise(c["seed"], (c["b"], 3, c["h"], c["h"])).to(DEV)
    y = m(x, torch.tensor([c["t"]] * c["b"], device=DEV), torch.ones(c["b"], 1, device=DEV))
    h = engine.UNetHandle.for_module(m)
    h.set_head_fusion(False)                               # dec1's output is only materialised without the fused head
    y_unfused = m(x.to(DEV), torch.tensor([c["t"]] * c["b"]).to(DEV), torch.ones(c["b"], 1, device=DEV))
    for j, name in enumerate(unet_ref.BLOCKS):
        want = arrays["act_" + name]                       # NCHW
        got = h.debug_activation(c["b"], c["h"], c["h"], j).cpu().numpy()   # NHWC, padded channels
        assert_close(got[..., : want.shape[1]].transpose(0, 3, 1, 2), want, what=name)
        assert not got[..., want.shape[1]:].any(), f"{name}: channel padding must stay zero"
    h.set_head_fusion(True)
    assert_close(y.cpu().numpy(), arrays["act_out"], what="eps (head in dec1.conv2's epilogue)")
    assert_close(y_unfused.cpu().numpy(), arrays["act_out"], what="eps (separate head launch)")


@pytest.mark.parametrize("sf", [0.05, 0.3, 0.4, 0.75])
def test_unet_forward_odd_channel_counts(models, sf):
    """Size factors whose channel counts (38/76, 51/102, 96/192 ...) are not tile multiples."""
    import copy
    m = models(sf)
    sd = m.state_dict()
    x = seeded_noise(77, (5, 3, 16, 16))
    t = torch.tensor([0, 7, 19, 33, 49])
    cond = torch.tensor([[0.0], [1.0], [0.5], [1.0], [0.0]])
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x, t, cond)
    got = copy.deepcopy(m).to(DEV)(x.to(DEV), t.to(DEV), cond.to(DEV))
    assert_close(got.cpu().numpy(), want.numpy(), what=f"sf={sf}")


@pytest.mark.parametrize("hw", [(16, 32), (32, 16), (48, 16)])
def test_unet_forward_non_square_images(models, hw):
    """H != W (the reference only ever uses squares, the kernels index rows and columns separately): strip rows,
    tap masks, the in-epilogue pool (W in {8, 16}) and the low-resolution head against the oracle."""
    import copy
    m = models(0.5)
    sd = m.state_dict()
    x = seeded_noise(78, (3, 3, hw[0], hw[1]))
    t = torch.tensor([2, 31, 49])
    cond = torch.tensor([[1.0], [0.0], [1.0]])
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x, t, cond)
    got = copy.deepcopy(m).to(DEV)(x.to(DEV), t.to(DEV), cond.to(DEV))
    assert_close(got.cpu().numpy(), want.numpy(), what=f"H x W = {hw}")


def test_unet_forward_random_shapes():
    """Seeded random (size factor, batch, H, W, t, cond) draws against the oracle (tools/fuzz_forward.py runs more)."""
    import numpy as np
    from distillation_trajectories_amd.config import Config
    from distillation_trajectories_amd.models import DiffusionUNet
    from distillation_trajectories_amd.synthetic import make_model
    rng = np.random.default_rng(5)
    for case in range(8):
        sf = float(rng.choice([0.05, 0.2, 0.3, 0.5, 0.75, 1.0]))
        H, W = int(rng.choice([16, 32, 48])), int(rng.choice([16, 32, 48]))
        B = int(rng.integers(1, 24)) if H * W <= 1024 else int(rng.integers(1, 8))
        cfg = Config(); cfg.image_size = H
        torch.manual_seed(2000 + case)
        m = make_model(DiffusionUNet, cfg, sf)
        x, t = torch.randn(B, 3, H, W), torch.randint(0, 50, (B,))
        cond = (torch.rand(B, 1) > 0.5).float()
        with torch.no_grad():
            want = unet_ref.unet_forward(m.state_dict(), x, t, cond)
        got = m.to(DEV)(x.to(DEV), t.to(DEV), cond.to(DEV))
        assert_close(got.cpu().numpy(), want.numpy(), what=f"case {case}: sf={sf} B={B} {H}x{W}")


def test_unet_forward_large_batch_properties(gpu_models):
    """Full bench batch (2 passes x 256) through the 128-row tiles: rows are independent (bit-exact under
    a batch permutation) and the CFG batch agrees with two separate single-pass calls (those take other
    tile / split-K shapes, so only to fp32 re-association noise)."""
    m = gpu_models(0.5)
    h = engine.UNetHandle.for_module(m)
    x = torch.randn(256, 3, 16, 16, generator=torch.Generator().manual_seed(3)).to(DEV)
    tb = h.time_bias([17, 17], [_hip.COND_NONE, _hip.COND_ONE])
    both = h.forward(x, tb, 2, 256)
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(4)).to(DEV)
    again = h.forward(x[perm].contiguous(), tb, 2, 256)
    assert torch.equal(again[:256], both[:256][perm]) and torch.equal(again[256:], both[256:][perm])
    solo_u = h.forward(x, tb[0:1].contiguous(), 1, 256)
    solo_c = h.forward(x, tb[1:2].contiguous(), 1, 256)
    assert_close(solo_u.cpu().numpy(), both[:256].cpu().numpy(), rtol=1e-5, atol=1e-6, what='solo uncond')
    assert_close(solo_c.cpu().numpy(), both[256:].cpu().numpy(), rtol=1e-5, atol=1e-6, what='solo cond')
    # oracle spot check on 4 rows, for the autotuned plan and for the untuned heuristic plan in both arithmetics
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x[:4].cpu(), torch.full((4,), 17), torch.ones(4, 1))
    assert_close(both[256:260].cpu().numpy(), want.numpy(), what="cond rows")
    for mode in (_hip.PREC_FP32, _hip.PREC_SPLIT_BF16):
        h.set_precision(mode)                        # also drops the tuned plan
        plain = h.forward(x, tb, 2, 256, tune=False)
        kinds = {c[5] for c in h.conv_choices(512, 16, 16)}
        assert any(k.endswith("+skip") for k in kinds), kinds     # the fused-skip walk is exercised
        assert_close(plain[256:260].cpu().numpy(), want.numpy(), what=f"heuristic plan, mode {mode}")
    h.set_precision(_hip.PREC_AUTO)


@pytest.mark.parametrize("channels", [1, 2])
def test_one_and_two_channel_images(channels):
    """MNIST-shaped (1-channel) and 2-channel inputs: the first-layer kernel, enc1's image skip in conv2's epilogue,
    the 1x1 head and the fused update all take the channel count at run time (reference models.py:138, 218)."""
    from distillation_trajectories_amd.config import Config
    from distillation_trajectories_amd.models import DiffusionUNet
    from distillation_trajectories_amd.synthetic import make_model
    cfg = Config(); cfg.image_size = 16; cfg.channels = channels
    m = make_model(DiffusionUNet, cfg, 0.2, seed=900 + channels).to(DEV)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    h = engine.UNetHandle.for_module(m)
    B = 5
    x = torch.randn(B, channels, 16, 16, generator=torch.Generator().manual_seed(channels)).to(DEV)
    tb = h.time_bias([31, 31], [_hip.COND_NONE, _hip.COND_ONE])
    got = h.forward(x, tb, 2, B, tune=False).cpu().numpy()
    assert got.shape == (2 * B, channels, 16, 16)
    with torch.no_grad():
        want_u = unet_ref.unet_forward(sd, x.cpu(), torch.full((B,), 31), None).numpy()
        want_c = unet_ref.unet_forward(sd, x.cpu(), torch.full((B,), 31), torch.ones(B, 1)).numpy()
    assert_close(got[:B], want_u, what=f"{channels}-channel, unconditional")
    assert_close(got[B:], want_c, what=f"{channels}-channel, conditional")


def test_shared_enc1_matches_per_pass_form(gpu_models, monkeypatch):
    """The CFG passes share x, so enc1 runs once for both (class-bias rows in conv2's epilogue);
    DT_NO_SHARED_ENC1=1 at create time keeps one enc1 per pass.  Both forms against each other and the oracle."""
    m = gpu_models(0.5)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    B = 6
    x = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(11)).to(DEV)
    h_shared = engine.UNetHandle(m.state_dict(), DEV)
    monkeypatch.setenv("DT_NO_SHARED_ENC1", "1")
    h_plain = engine.UNetHandle(m.state_dict(), DEV)
    monkeypatch.delenv("DT_NO_SHARED_ENC1")
    assert h_shared.tb_stride > h_plain.tb_stride                 # only the shared form carries the class-bias columns
    outs = []
    for h in (h_shared, h_plain):
        tb = h.time_bias([23, 23], [_hip.COND_NONE, _hip.COND_ONE])
        outs.append(h.forward(x, tb, 2, B, tune=False).cpu().numpy())
    assert_close(outs[0], outs[1], rtol=2e-5, atol=2e-5, what="shared vs per-pass enc1")
    with torch.no_grad():
        want_u = unet_ref.unet_forward(sd, x.cpu(), torch.full((B,), 23), None).numpy()
        want_c = unet_ref.unet_forward(sd, x.cpu(), torch.full((B,), 23), torch.ones(B, 1)).numpy()
    assert_close(outs[0][:B], want_u, what="shared enc1, unconditional pass")
    assert_close(outs[0][B:], want_c, what="shared enc1, conditional pass")


def test_conv_tile_variants_match_oracle(gpu_models):
    """Every launch shape of the conv kernels (tile, wave layout, tap split, arithmetic, fused skip), pinned one
    layer at a time through dt_unet_set_conv_choice, against the oracle forward."""
    m = gpu_models(1.0)
    h = engine.UNetHandle.for_module(m)
    B = 41                                           # 41 images: ragged against the 128-row tile below 4x4
    x = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(5)).to(DEV)
    tb = h.time_bias([9, 9], [_hip.COND_NONE, _hip.COND_ONE])
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x[:3].cpu(), torch.full((3,), 9), torch.ones(3, 1)).numpy()
    tried = 0
    for block in range(8):
        for slot in (1, 2):
            for prec, bm, bn, sp, fuse in [(0, 128, 128, 1, 0), (0, 64, 64, 3, 0), (1, 128, 128, 1, 0), (1, 64, 128, 9, 0),
                                           (1, 128, 64, 1, 1), (1, 64, 64, 1, 0), (0, 128, 64, 1, 1), (0, 64, 128, 9, 0),
                                           (1, 128, 128, 3, 0), (3, 128, 128, 1, 0), (3, 128, 64, 2, 0), (3, 64, 128, 4, 0),
                                           (3, 64, 64, 1, 1), (3, 128, 128, 1, 1), (1, 64, 64, 4, 0), (0, 64, 64, 2, 0), (1, 128, 64, 8, 0), (4, 128, 128, 1, 0), (4, 64, 64, 2, 0),
                                           (4, 128, 64, 1, 1), (4, 64, 128, 4, 0), (3, 256, 64, 1, 0), (4, 256, 64, 2, 0), (4, 256, 64, 1, 1),
                                           (5, 64, 64, 1, 1), (5, 64, 64, 2, 0), (5, 128, 64, 1, 1), (5, 128, 64, 4, 0), (5, 64, 128, 1, 1), (5, 64, 128, 2, 0)]:
                h.set_precision(_hip.PREC_AUTO)       # back to the heuristic plan
                try:
                    h.set_conv_choice(2 * B, 16, 16, block, slot, bm, bn, sp, prec, fuse if slot == 2 else 0)
                except _hip.HipLibraryError:
                    continue                           # tile wider than the layer's padded channel count
                got = h.forward(x, tb, 2, B, tune=False)
                assert_close(got[B:B + 3].cpu().numpy(), want, what=f"block {block} slot {slot} {prec}/{bm}x{bn}/s{sp}/f{fuse}")
                tried += 1
    # the blocks' 1x1 skip convolutions as launches of their own (conv2 pinned unfused), with channel-chunk splits
    skips = 0
    for block in range(1, 8):
        for prec, bm, bn, sp in [(1, 64, 64, 1), (1, 64, 64, 2), (1, 128, 64, 4), (0, 64, 64, 8), (0, 64, 128, 1)]:
            h.set_precision(_hip.PREC_AUTO)
            try:
                h.set_conv_choice(2 * B, 16, 16, block, 2, 64, 64, 1, 1, 0)
                h.set_conv_choice(2 * B, 16, 16, block, 0, bm, bn, sp, prec, 0)
            except _hip.HipLibraryError:
                continue
            names = {(c[0], c[1]) for c in h.conv_choices(2 * B, 16, 16)}
            if (engine.BLOCK_NAMES[block], "skip") not in names:
                continue                               # identity skip: no launch
            got = h.forward(x, tb, 2, B, tune=False)
            assert_close(got[B:B + 3].cpu().numpy(), want, what=f"block {block} skip {prec}/{bm}x{bn}/s{sp}")
            skips += 1
    h.set_precision(_hip.PREC_AUTO)
    assert tried >= 310 and skips >= 10
    # the 256 x 64 tile (four waves stacked along M) where the autotuner uses it: 64-channel layers of the half-size
    # student, including the epilogue's pool / image-skip / 1x1-head modes on a 128-row stage
    ms = gpu_models(0.5)
    hs = engine.UNetHandle.for_module(ms)
    tbs = hs.time_bias([9, 9], [_hip.COND_NONE, _hip.COND_ONE])
    sds = {k: v.cpu() for k, v in ms.state_dict().items()}
    with torch.no_grad():
        want_s = unet_ref.unet_forward(sds, x[:3].cpu(), torch.full((3,), 9), torch.ones(3, 1)).numpy()
    tall = 0
    for block in range(8):
        for slot in (1, 2):
            for prec, bm, sp, fuse in [(3, 256, 1, 0), (4, 256, 1, 1), (4, 256, 2, 0), (3, 256, 4, 0), (5, 64, 1, 1), (5, 128, 1, 1), (5, 64, 2, 0), (5, 128, 1, 0)]:
                hs.set_precision(_hip.PREC_AUTO)
                try:
                    hs.set_conv_choice(2 * B, 16, 16, block, slot, bm, 64, sp, prec, fuse if slot == 2 else 0)
                except _hip.HipLibraryError:
                    continue
                got = hs.forward(x, tbs, 2, B, tune=False)
                assert_close(got[B:B + 3].cpu().numpy(), want_s, what=f"sf 0.5 block {block} slot {slot} {prec}/{bm}x64/s{sp}/f{fuse}")
                tall += 1
    hs.set_precision(_hip.PREC_AUTO)
    assert tall >= 70


def test_k_split_kernels_are_deterministic_and_row_independent(gpu_models):
    """The K-split strip kernels (partial tiles of four / two waves summed in wave order, a mid-kernel reduction before
    the fused skip walk) must give bit-identical rows whatever the row's place in the batch and from run to run."""
    m = gpu_models(1.0)
    h = engine.UNetHandle.for_module(m)
    B = 48
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 3, 16, 16, generator=g).to(DEV)
    perm = torch.randperm(B, generator=g).to(DEV)
    tb = h.time_bias([5, 5], [_hip.COND_NONE, _hip.COND_ONE])
    pinned = 0
    for block, slot, bm, bn, fuse in [(6, 2, 128, 64, 1), (6, 2, 64, 64, 1), (7, 2, 64, 128, 1), (5, 2, 64, 64, 1), (2, 1, 64, 64, 0),
                                      (2, 2, 128, 64, 0), (1, 2, 128, 64, 1), (7, 1, 128, 64, 0)]:
        h.set_precision(_hip.PREC_AUTO)
        h.set_conv_choice(2 * B, 16, 16, block, slot, bm, bn, 1, 5, fuse)
        kinds = [c for c in h.conv_choices(2 * B, 16, 16) if c[0] == engine.BLOCK_NAMES[block] and c[1] == ("conv1", "conv2")[slot - 1]]
        assert kinds and kinds[0][5].startswith("split-bf16-stripk"), kinds
        a = h.forward(x, tb, 2, B, tune=False).clone()
        b = h.forward(x, tb, 2, B, tune=False).clone()
        c = h.forward(x[perm].contiguous(), tb, 2, B, tune=False)
        assert torch.equal(a, b), f"block {block} slot {slot} {bm}x{bn}: run-to-run"
        assert torch.equal(c[:B], a[:B][perm]) and torch.equal(c[B:], a[B:][perm]), f"block {block} slot {slot} {bm}x{bn}: row placement"
        pinned += 1
    h.set_precision(_hip.PREC_AUTO)
    assert pinned == 8


def test_sampler_graph_replay_is_bit_exact(gpu_models, monkeypatch):
    """DT_GRAPH=1: dt_sample_trajectory captures the loop into a hipGraph on a side stream and replays it on the
    next call with the same arguments; both must equal the plain launch sequence bit for bit."""
    from distillation_trajectories_amd.utils.diffusion import get_diffusion_params, psample_coefficients, timestep_indices
    from distillation_trajectories_amd.config import Config
    m = gpu_models(0.5)
    h = engine.UNetHandle.for_module(m)
    cfg = Config(); cfg.image_size, cfg.timesteps = 16, 50
    idx = timestep_indices(50, 10)
    coef = psample_coefficients(get_diffusion_params(50, cfg), idx)
    noise = [i > 0 for i in idx]
    B, E = 6, 3 * 16 * 16
    g = torch.Generator().manual_seed(11)
    x_T, z = torch.randn(B, E, generator=g).to(DEV), torch.randn(len(idx) * B, E, generator=g).to(DEV)
    tb = h.time_bias([i for i in idx for _ in (0, 1)], [_hip.COND_NONE, _hip.COND_ONE] * len(idx))
    shift = [k * B for k in range(len(idx))]

    def run(traj):
        traj[0].copy_(x_T)
        h.sample(_hip.RULE_PSAMPLE, traj, 16, 16, tb, 2, coef, noise, z=z, z_shift=shift, w_scalar=3.0)
        return traj
    plain = run(torch.empty(len(idx) + 1, B, E, device=DEV)).clone()
    monkeypatch.setenv("DT_GRAPH", "1")
    side = torch.cuda.Stream()            # (the handle keeps one workspace per stream: no synchronisation with the plain run needed)
    side.wait_stream(torch.cuda.current_stream())     # x_T, z, tb were produced on the current stream
    traj = torch.empty(len(idx) + 1, B, E, device=DEV)
    with torch.cuda.stream(side):
        first = run(traj).clone()         # capture + first launch
        traj.zero_()
        second = run(traj).clone()        # replay
    side.synchronize()
    assert torch.equal(first, plain) and torch.equal(second, plain)


def test_time_bias_rows(gpu_models):
    m = gpu_models(0.2)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    h = engine.UNetHandle.for_module(m)
    ts, modes = [0, 1, 49, 999, 3920], [_hip.COND_NONE, _hip.COND_ZERO, _hip.COND_ONE, _hip.COND_ONE, _hip.COND_NONE]
    got = h.time_bias(ts, modes).cpu()
    off = 0
    for name in unet_ref.BLOCKS:
        cout = sd[f"{name}.time_mlp.weight"].shape[0]
        for r, (t, mode) in enumerate(zip(ts, modes)):
            temb = unet_ref.time_embedding(sd, torch.tensor([t]), cond_of({0: "none", 1: "zero", 2: "one"}[mode], 1))
            want = torch.relu(torch.nn.functional.linear(temb, sd[f"{name}.time_mlp.weight"], sd[f"{name}.time_mlp.bias"]))[0]
            assert_close(got[r, off: off + cout].numpy(), want.numpy(), rtol=2e-5, atol=2e-6, what=f"{name} t={t}")
        pad = (cout + 15) // 16 * 16
        assert not got[:, off + cout: off + pad].any()
        off += pad
    # the rest of a row: enc1.conv2's nine class-bias vectors (the spatially constant time bias under a zero-padded 3x3
    # conv: sum over the taps inside the picture of W2[n][ci][tap] b[ci]; corner / edge / interior of the picture)
    c0 = sd["enc1.conv2.weight"].shape[0]
    c0p = (c0 + 15) // 16 * 16
    assert h.tb_stride == off + 9 * c0p
    w2 = sd["enc1.conv2.weight"].double()                                  # [n][ci][3][3]
    for r in range(len(ts)):
        b = got[r, :c0].double()
        per_tap = torch.einsum("nckl,c->nkl", w2, b)                       # [n][dy+1][dx+1]
        for cy in range(3):
            for cx in range(3):
                rows = [d for d in range(3) if not (cy == 0 and d == 0) and not (cy == 2 and d == 2)]
                cols = [d for d in range(3) if not (cx == 0 and d == 0) and not (cx == 2 and d == 2)]
                want = per_tap[:, rows][:, :, cols].sum(dim=(1, 2))
                lo = off + (cy * 3 + cx) * c0p
                assert_close(got[r, lo: lo + c0].numpy(), want.float().numpy(), rtol=2e-5, atol=2e-5, what=f"class bias {cy}{cx} row {r}")
                assert not got[r, lo + c0: lo + c0p].any()


# ------------------------------------------------------------------ fused update: bit exact
@pytest.mark.parametrize("rule", ["engine", "psample", "manager"])
def test_cfg_update_bit_exact(rule):
    g = torch.Generator().manual_seed(11)
    B, shape = 7, (7, 3, 16, 16)
    x, eu, ec, z = (torch.randn(shape, generator=g) for _ in range(4))
    w = 3.7
    if rule == "engine":
        co = sampler_ref.engine_coefficients(50)[23]
        eps = eu + w * (ec - eu)
        want = co[0] * x - co[1] * eps
        want = want + co[2] * z
        got = engine.cfg_update(_hip.RULE_ENGINE, x.to(DEV), eu.to(DEV), ec.to(DEV), z.reshape(B, -1).to(DEV),
                                [float(v) for v in co], True, w_scalar=w)
        assert torch.equal(got.cpu(), want)
        keep = engine.cfg_update(_hip.RULE_ENGINE, x.to(DEV), eu.to(DEV), ec.to(DEV), None, [1, 0, 0], False, w_scalar=w)
        assert torch.equal(keep.cpu(), x)
    elif rule == "psample":
        p = sampler_ref.diffusion_params(50)
        t = torch.full((B,), 31, dtype=torch.long)
        for t_index in (31, 0):
            want = sampler_ref.p_sample(lambda xx, tt, c: ec if c is not None else eu, x, t, t_index, p, w,
                                        noise=z if t_index > 0 else None)
            co = (float(p["sqrt_recip_alphas"][31]), float(1.0 - p["sqrt_one_minus_alphas_cumprod"][31]), float(p["betas"][31]))
            got = engine.cfg_update(_hip.RULE_PSAMPLE, x.to(DEV), eu.to(DEV), ec.to(DEV),
                                    z.reshape(B, -1).to(DEV) if t_index > 0 else None, co, t_index > 0, w_scalar=w)
            assert torch.equal(got.cpu(), want), t_index
    else:
        want = sampler_ref.manager_update(x, eu, 60, 20, z)
        co = (1 - 0.9, float(torch.sqrt(torch.tensor(0.9))), 0.1 * (60.0 / 20.0))
        got = engine.cfg_update(_hip.RULE_MANAGER, x.to(DEV), eu.to(DEV), None, z.reshape(B, -1).to(DEV), co, True)
        assert torch.equal(got.cpu(), want)


def test_cfg_update_per_row_scale_and_noise_rows():
    g = torch.Generator().manual_seed(12)
    B, E = 6, 768
    x, eu, ec = (torch.randn(B, E, generator=g) for _ in range(3))
    table = torch.randn(20, E, generator=g)
    w = torch.tensor([1.5, 2.0, 3.0, 5.0, 7.5, 20.0])
    rows = torch.tensor([3, 4, 5, 3, 4, 5], dtype=torch.int32)
    co = sampler_ref.engine_coefficients(50)[9]
    lib = _hip.load()
    out = torch.empty(B, E, device=DEV)
    args = [t.to(DEV) for t in (x, eu, ec, table, rows, w)]
    import ctypes
    coef = (ctypes.c_float * 4)(float(co[0]), float(co[1]), float(co[2]), 0.0)
    # z row = rows[r] + shift is exercised through dt_sample_trajectory; here shift 0
    _hip.check(lib.dt_cfg_update(_hip.RULE_ENGINE, _hip.ptr(args[0]), _hip.ptr(args[1]), _hip.ptr(args[2]), _hip.ptr(args[3]),
                                 _hip.ptr(args[4]), coef, 1, _hip.ptr(args[5]), 0.0, _hip.ptr(out), B, E, _hip.stream_ptr()), "upd")
    eps = eu + w[:, None] * (ec - eu)
    want = co[0] * x - co[1] * eps
    want = want + co[2] * table[rows.long()]
    assert torch.equal(out.cpu(), want)


# ------------------------------------------------------------------ whole loops vs golden
def test_engine_trajectories_golden(golden, gpu_models, conv_path):
    from distillation_trajectories_amd.analysis.trajectory_engine import generate_trajectory
    arrays, meta = golden
    for c in meta["engine_cases"]:
        if c["seed"] is None:
            torch.manual_seed(c["global_seed"])
            noise = torch.randn(1, 3, 16, 16)
        else:
            noise = seeded_noise(c["seed"], (1, 3, 16, 16))
        tr = generate_trajectory(gpu_models(c["sf"]), noise, c["T"], torch.device(DEV), seed=c["seed"], guidance_scale=c["gs"])
        got = torch.stack(tr).numpy()
        assert len(tr) == c["T"] + 1 and not tr[0].is_cuda
        assert np.array_equal(got[0], noise.numpy()) and np.array_equal(got[-1], got[-2])
        assert_close(got, arrays[c["key"]], rtol=1e-4, atol=1e-4, what=str(c))


def test_psample_loops_golden(golden, gpu_models, conv_path):
    from distillation_trajectories_amd.utils.diffusion import get_diffusion_params, p_sample_loop
    arrays, meta = golden
    for c in meta["psample_cases"]:
        cfg = Config()
        cfg.timesteps = c["timesteps"]
        torch.manual_seed(c["global_seed"])
        img, tr = p_sample_loop(gpu_models(c["sf"]), (c["b"], 3, 16, 16), c["sample_steps"],
                                get_diffusion_params(c["sample_steps"], cfg), device=torch.device(DEV), config=cfg,
                                track_trajectory=True, guidance_scale=c["w"])
        got = torch.stack(tr).numpy()
        assert got.shape == arrays[c["key"]].shape
        assert_close(got, arrays[c["key"]], rtol=1e-4, atol=1e-4, what=str(c))
        assert img.is_cuda and np.array_equal(img.cpu().numpy(), got[-1])


def test_psample_single_step_matches_loop_entry(gpu_models):
    from distillation_trajectories_amd.utils.diffusion import get_diffusion_params, p_sample
    m = gpu_models(0.01)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    params = get_diffusion_params(50)
    x = seeded_noise(5, (3, 3, 16, 16))
    t = torch.full((3,), 20, dtype=torch.long)
    torch.manual_seed(9)
    got = p_sample(m, x.to(DEV), t.to(DEV), 20, params, guidance_scale=2.5)
    torch.manual_seed(9)
    with torch.no_grad():
        want = sampler_ref.p_sample(lambda a, b, c: unet_ref.unet_forward(sd, a, b, c), x, t, 20,
                                    sampler_ref.diffusion_params(50), 2.5)
    assert_close(got.cpu().numpy(), want.numpy(), what="p_sample")


def _check_metrics(got, want, rel=1e-4, skip=()):
    assert set(got) == set(want)
    for k, w in want.items():
        if k in skip:
            continue
        g = got[k]
        if isinstance(w, list):
            assert len(g) == len(w), k
            assert_close(np.array(g, dtype=np.float64), np.array(w, dtype=np.float64), rtol=rel, atol=1e-7, what=k)
        elif isinstance(w, float) and math.isnan(w):
            assert math.isnan(float(g)), k
        else:
            assert_close(float(g), float(w), rtol=rel, atol=1e-9, what=k)


def test_manager_golden(golden, gpu_models, tmp_path, conv_path):
    from distillation_trajectories_amd.analysis.metrics.trajectory_metrics import compute_trajectory_metrics
    from distillation_trajectories_amd.utils.trajectory_manager import TrajectoryManager
    arrays, meta = golden
    for c in meta["manager_cases"]:
        cfg = Config()
        cfg.image_size, cfg.sample_steps = 16, c["sample_steps"]
        cfg.teacher_steps, cfg.student_steps = c["teacher_steps"], c["student_steps"]
        cfg.trajectory_dir = str(tmp_path / "traj")
        man = TrajectoryManager(gpu_models(c["teacher_sf"]), gpu_models(c["student_sf"]), cfg, size_factor=c["student_sf"])
        tt, st = man.generate_trajectory(seed=c["seed"])
        assert [t for _, t in tt] == c["teacher_t"] and [t for _, t in st] == c["student_t"]     # bit-exact indexing
        assert_close(torch.stack([x.cpu() for x, _ in tt]).numpy(), arrays[c["key"] + "_teacher"], rtol=1e-4, atol=1e-4)
        assert_close(torch.stack([x.cpu() for x, _ in st]).numpy(), arrays[c["key"] + "_student"], rtol=1e-4, atol=1e-4)
        # metrics on the reference's own trajectories isolate the metric kernels (incl. interp1d resampling)
        a = [(torch.from_numpy(x), t) for x, t in zip(arrays[c["key"] + "_teacher"], c["teacher_t"])]
        b = [(torch.from_numpy(x), t) for x, t in zip(arrays[c["key"] + "_student"], c["student_t"])]
        np.random.seed(c["np_seed"])
        _check_metrics(compute_trajectory_metrics(a, b, cfg), c["metrics"], rel=2e-5)


def test_manager_disk_roundtrip(gpu_models, tmp_path):
    from distillation_trajectories_amd.utils.trajectory_manager import generate_trajectories_with_disk_storage
    cfg = Config()
    cfg.image_size, cfg.sample_steps, cfg.teacher_steps, cfg.student_steps = 16, 20, 10, 5
    cfg.trajectory_dir = str(tmp_path / "traj")
    man = generate_trajectories_with_disk_storage(gpu_models(0.2), gpu_models(0.01), cfg, size_factor=0.01, num_samples=3)
    tts, sts = man.load_trajectories()
    assert len(tts) == 3 and len(tts[0]) == 11 and len(sts[0]) == 6 and tts[0][0][0].shape == (1, 3, 16, 16)
    agg = man.compute_trajectory_metrics_batch()
    assert len(agg["endpoint_distances"]) == 3 and "distribution_similarity_avg" in agg
    from distillation_trajectories_amd.analysis.metrics.time_dependent import analyze_time_dependent_distances
    td = analyze_time_dependent_distances(tts, sts, cfg, size_factor=0.01)
    cpu = metrics_ref.time_dependent_distances([[(x.cpu(), t) for x, t in tr] for tr in tts],
                                               [[(x.cpu(), t) for x, t in tr] for tr in sts])
    assert_close(td["teacher_avg_per_timestep"], cpu["teacher_avg_per_timestep"], rtol=1e-5, atol=1e-7)
    assert_close(td["student_std_distance"], cpu["student_std_distance"], rtol=1e-4, atol=1e-8)


def test_metrics_golden(golden):
    from distillation_trajectories_amd.analysis.metrics.trajectory_metrics import compute_trajectory_metrics
    arrays, meta = golden
    for c in meta["metric_cases"]:
        a = [torch.from_numpy(x) for x in arrays[c["key"] + "_teacher"]]
        b = a if c["key"] == "same" else [torch.from_numpy(x) for x in arrays[c["key"] + "_student"]]
        if "np_seed" in c:
            np.random.seed(c["np_seed"])
        got = compute_trajectory_metrics(a, [x.clone() for x in b])
        # log1p(1 - 1000*mse) is ill-conditioned near its pole: check it scaled by its condition number
        want = c["metrics"]
        _check_metrics(got, want, rel=2e-5, skip=("trajectory_mse",))
        w = want["trajectory_mse"]
        if math.isnan(w):
            assert math.isnan(float(got["trajectory_mse"]))
        else:
            arg = math.expm1(w)                    # = 1 - 1000*mean step mse
            cond = max(1.0, abs(1.0 - arg) / max(abs(1.0 + arg), 1e-12))
            assert abs(float(got["trajectory_mse"]) - w) <= 2e-5 * cond * max(1.0, abs(w)), c["key"]


def test_metrics_device_inputs_and_tuples(golden):
    from distillation_trajectories_amd.analysis.metrics.trajectory_metrics import compute_trajectory_metrics
    arrays, meta = golden
    c = meta["metric_cases"][1]
    a = [(torch.from_numpy(x).to(DEV), i) for i, x in enumerate(arrays[c["key"] + "_teacher"])]
    b = [(torch.from_numpy(x).to(DEV), i) for i, x in enumerate(arrays[c["key"] + "_student"])]
    _check_metrics(compute_trajectory_metrics(a, b), c["metrics"], rel=2e-5, skip=("trajectory_mse",))


def test_compare_trajectories_golden(golden, gpu_models, conv_path):
    from distillation_trajectories_amd.analysis.trajectory_engine import compare_trajectories
    _, meta = golden
    c = meta["compare_case"]
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, c["T"]
    res = compare_trajectories(gpu_models(c["teacher_sf"]), gpu_models(c["student_sf"]), cfg,
                               guidance_scales=c["guidance_scales"], size_factor=c["student_sf"], num_samples=c["num_samples"])
    assert set(res) == {"teacher_metrics", "student_metrics"}
    for side in res:
        for gs in c["guidance_scales"]:
            _check_metrics(res[side][gs], c["result"][side][str(gs)], rel=1e-4)
    # second call hits the cached teacher trajectories and must not change anything
    res2 = compare_trajectories(gpu_models(c["teacher_sf"]), gpu_models(c["student_sf"]), cfg,
                                guidance_scales=c["guidance_scales"], size_factor=c["student_sf"], num_samples=c["num_samples"])
    assert res2 == res


def test_wasserstein_subsampled_tables():
    """E = 3072 > 1000: per-pair coordinate tables (seed dependent) through dt_traj_wasserstein."""
    from distillation_trajectories_amd.analysis.metrics.trajectory_metrics import wasserstein_index_tables
    from scipy.stats import wasserstein_distance
    g = torch.Generator().manual_seed(21)
    n, B, E = 5, 3, 3072
    X = torch.randn(n, B, E, generator=g)
    Y = X + 0.1 * torch.randn(n, B, E, generator=g)
    seeds = [42, 43, 42]
    tables, rows = wasserstein_index_tables(seeds, n, E)
    got = engine.device_wasserstein(X.to(DEV), Y.to(DEV), tables.to(DEV), rows.to(DEV)).cpu().numpy()
    for b, seed in enumerate(seeds):
        np.random.seed(seed + 1)
        for i in range(n):
            idx = np.random.choice(E, 1000, replace=False)
            want = wasserstein_distance(X[i, b].numpy()[idx], Y[i, b].numpy()[idx])
            assert abs(got[b, i] - want) <= 1e-12 + 1e-9 * want


# ------------------------------------------------------------------ both convolution arithmetics are fp32-accurate
def test_conv_arithmetic_accuracy_vs_float64(models):
    """Exact-fp32 MFMA and split-bf16 (3 planes, 6 products) against a float64 evaluation of the same
    U-Net: both must sit at fp32 rounding level, and the split path must not be worse than 2x the native one."""
    import copy
    import torch.nn.functional as F
    sd64 = {k: v.double() for k, v in models(0.5).state_dict().items() if v.dtype.is_floating_point}
    g = torch.Generator().manual_seed(31)
    x = torch.randn(16, 3, 16, 16, generator=g)
    t = torch.full((16,), 23, dtype=torch.long)
    up = lambda a: F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True)   # noqa: E731
    with torch.no_grad():
        sd32 = models(0.5).state_dict()
        temb = unet_ref.time_embedding(sd32, t, torch.ones(16, 1)).double()   # embedding tower: fp32 oracle (tiny)
        x1 = unet_ref.block_forward(sd64, "enc1", x.double(), temb)
        x2 = unet_ref.block_forward(sd64, "enc2", F.max_pool2d(x1, 2), temb)
        x3 = unet_ref.block_forward(sd64, "enc3", F.max_pool2d(x2, 2), temb)
        x4 = unet_ref.block_forward(sd64, "enc4", F.max_pool2d(x3, 2), temb)
        xb = unet_ref.block_forward(sd64, "bottleneck", F.max_pool2d(x4, 2), temb)
        d3 = unet_ref.block_forward(sd64, "dec3", torch.cat([up(xb), x4], 1), temb)
        d2 = unet_ref.block_forward(sd64, "dec2", torch.cat([up(d3), x3], 1), temb)
        d1 = unet_ref.block_forward(sd64, "dec1", torch.cat([up(d2), x2], 1), temb)
        want = F.conv2d(up(d1), sd64["final.weight"], sd64["final.bias"])
    m = copy.deepcopy(models(0.5)).to(DEV)
    h = engine.UNetHandle.for_module(m)
    errs = {}
    for name, mode in (("fp32", _hip.PREC_FP32), ("split-bf16", _hip.PREC_SPLIT_BF16)):
        h.set_precision(mode)
        got = m(x.to(DEV), t.to(DEV), torch.ones(16, 1, device=DEV)).double().cpu()
        errs[name] = ((got - want).norm() / want.norm()).item()
    h.set_precision(_hip.PREC_AUTO)
    print("relative L2 error vs float64:", errs)
    assert max(errs.values()) < 2e-6, errs
    assert errs["split-bf16"] <= 2.0 * errs["fp32"] + 1e-7, errs


# ------------------------------------------------------------------ remaining API corners and SURVEY §8f next rows
def test_p_sample_per_sample_timesteps(gpu_models):
    """p_sample with a different t per batch row (the general form of utils/diffusion.py:102)."""
    from distillation_trajectories_amd.utils.diffusion import get_diffusion_params, p_sample
    m = gpu_models(0.01)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    x = seeded_noise(15, (4, 3, 16, 16))
    t = torch.tensor([3, 40, 3, 17])
    torch.manual_seed(21)
    got = p_sample(m, x.to(DEV), t.to(DEV), 5, get_diffusion_params(50), guidance_scale=2.0)
    torch.manual_seed(21)
    with torch.no_grad():
        want = sampler_ref.p_sample(lambda a, b, c: unet_ref.unet_forward(sd, a, b, c), x, t, 5, sampler_ref.diffusion_params(50), 2.0)
    assert_close(got.cpu().numpy(), want.numpy(), what="per-sample t")


def test_generate_trajectory_batched_without_cfg(gpu_models):
    """B > 1 is legal in the reference when no CFG branch is taken (trajectory_engine.py:83)."""
    from distillation_trajectories_amd.analysis.trajectory_engine import generate_trajectory
    m = gpu_models(0.2)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    noise = seeded_noise(3, (3, 3, 16, 16))
    got = torch.stack(generate_trajectory(m, noise, 10, torch.device(DEV), seed=9, guidance_scale=1.0))
    with torch.no_grad():
        want = torch.stack(sampler_ref.generate_trajectory(lambda a, b, c: unet_ref.unet_forward(sd, a, b, c), noise, 10, seed=9,
                                                           guidance_scale=1.0))
    assert_close(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-4, what="B=3 engine")


def test_manager_fixed_samples_and_batched_generation(gpu_models, tmp_path):
    from distillation_trajectories_amd.utils.trajectory_manager import TrajectoryManager
    cfg = Config()
    cfg.image_size, cfg.sample_steps, cfg.teacher_steps, cfg.student_steps = 16, 40, 8, 4
    cfg.trajectory_dir = str(tmp_path / "t")
    teacher, student = gpu_models(0.2), gpu_models(0.01)
    sds = [{k: v.cpu() for k, v in mm.state_dict().items()} for mm in (teacher, student)]
    fns = [lambda a, b, c, sd=sd: unet_ref.unet_forward(sd, a, b, c) for sd in sds]
    man = TrajectoryManager(teacher, student, cfg, size_factor=0.01)
    # batched generation over seeds 0..3 equals the reference's one-by-one loop (oracle)
    pairs = man.generate_trajectories_batched([0, 1, 2, 3])
    state_after = torch.get_rng_state()
    for seed, (tt, st) in enumerate(pairs):
        with torch.no_grad():
            wt, ws = sampler_ref.manager_generate(fns[0], fns[1], cfg, seed=seed)
        assert [t for _, t in tt] == [t for _, t in wt] and [t for _, t in st] == [t for _, t in ws]
        assert_close(torch.stack([x.cpu() for x, _ in tt]).numpy(), torch.stack([x for x, _ in wt]).numpy(), rtol=1e-4, atol=1e-4)
        assert_close(torch.stack([x.cpu() for x, _ in st]).numpy(), torch.stack([x for x, _ in ws]).numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(state_after, torch.get_rng_state())          # same generator state as the per-seed loop leaves
    # fixed start samples (reference :265-387): noise continues from the seeded generator, student not re-seeded
    fixed = seeded_noise(99, (2, 1, 3, 16, 16))
    man2 = TrajectoryManager(teacher, student, cfg, size_factor=0.01, fixed_samples=fixed)
    tt, st = man2.generate_trajectory_from_sample(fixed[1], seed=1)
    torch.manual_seed(1)
    with torch.no_grad():
        wt = sampler_ref.manager_trajectory(fns[0], fixed[1], sampler_ref.manager_indices(40, 8), 8)
        ws = sampler_ref.manager_trajectory(fns[1], fixed[1], sampler_ref.manager_indices(40, 4), 8)
    assert_close(torch.stack([x.cpu() for x, _ in tt]).numpy(), torch.stack([x for x, _ in wt]).numpy(), rtol=1e-4, atol=1e-4)
    assert_close(torch.stack([x.cpu() for x, _ in st]).numpy(), torch.stack([x for x, _ in ws]).numpy(), rtol=1e-4, atol=1e-4)
    assert len(man2.generate_and_save_trajectories(2)) == 2


def test_next_rows_divergence_noise_metrics_average(golden, gpu_models):
    from distillation_trajectories_amd.analysis.noise_prediction.noise_analysis import calculate_noise_metrics, predict_noise
    from distillation_trajectories_amd.analysis.trajectory_engine import average_sample_trajectories
    from distillation_trajectories_amd.evaluation.metrics import compute_trajectory_divergence
    arrays, meta = golden
    for c, d in zip(meta["manager_cases"], meta["divergence_cases"]):
        tt = [(torch.from_numpy(x), t) for x, t in zip(arrays[c["key"] + "_teacher"], c["teacher_t"])]
        st = [(torch.from_numpy(x), t) for x, t in zip(arrays[c["key"] + "_student"], c["student_t"])]
        _check_metrics(compute_trajectory_divergence(tt, st), d["result"], rel=2e-5)
    got = calculate_noise_metrics(torch.from_numpy(arrays["noise_teacher"]), torch.from_numpy(arrays["noise_student"]))
    _check_metrics(got, meta["noise_metric_case"], rel=2e-5)
    assert all(isinstance(v, float) for v in got.values())
    m = gpu_models(0.2)
    x = seeded_noise(4, (3, 3, 16, 16)).to(DEV)
    eps = predict_noise(m, x, torch.tensor([7, 7, 7], device=DEV), torch.device(DEV))
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        assert_close(eps.cpu().numpy(), unet_ref.unet_forward(sd, x.cpu(), torch.tensor([7, 7, 7])).numpy(), what="predict_noise")
    c = meta["average_case"]
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, c["T"]
    ta, sa = average_sample_trajectories(gpu_models(c["teacher_sf"]), gpu_models(c["student_sf"]), cfg, c["guidance_scales"],
                                         c["num_samples"], base_seed=c["base_seed"])
    for gs in c["guidance_scales"]:
        assert ta[gs][0].shape == (1, 3, 16, 16) and len(ta[gs]) == c["T"] + 1
        assert_close(torch.stack(ta[gs]).numpy(), arrays[f"avg_teacher_{gs}"], rtol=1e-4, atol=1e-4, what=f"avg teacher {gs}")
        assert_close(torch.stack(sa[gs]).numpy(), arrays[f"avg_student_{gs}"], rtol=1e-4, atol=1e-4, what=f"avg student {gs}")


def test_next_row_fid_sampler(golden, gpu_models):
    from distillation_trajectories_amd.analysis.metrics.fid_score import calculate_fid, generate_samples, p_sample_loop
    arrays, meta = golden
    c = meta["fid_case"]
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, c["T"]
    m_samples, m_loop = gpu_models(c["sf_samples"]), gpu_models(c["sf_loop"])      # (model construction re-seeds)
    x = seeded_noise(c["x_seed"], (2, 3, 16, 16)).to(DEV)
    torch.manual_seed(c["seed_samples"])
    got = generate_samples(m_samples, cfg, 3, torch.device(DEV))
    assert got.shape == (3, 3, 16, 16) and got.is_cuda
    assert_close(got.cpu().numpy(), arrays["fid_samples"], rtol=1e-4, atol=1e-4, what="generate_samples")
    torch.manual_seed(c["seed_loop"])
    got = p_sample_loop(m_loop, x, cfg)
    assert_close(got.cpu().numpy(), arrays["fid_loop"], rtol=1e-4, atol=1e-4, what="fid p_sample_loop")
    assert abs(calculate_fid(arrays["fid_feat1"], arrays["fid_feat2"]) - c["fid"]) <= 1e-12 * abs(c["fid"])
    assert calculate_fid(arrays["fid_feat1"][:1], arrays["fid_feat2"]) == 999.0


def test_split_k_is_deterministic_under_load(gpu_models):
    """Every 3x3 layer split along K as deep as it goes: the slabs are summed in z order, so 150 repeated forwards must
    agree bit for bit with the first one -- also while another model keeps the chip unevenly busy on a second stream;
    the first result is checked against the oracle."""
    import threading
    m, other = gpu_models(1.0), gpu_models(0.5)
    h, ho = engine.UNetHandle.for_module(m), engine.UNetHandle.for_module(other)
    B = 37
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, 3, 16, 16, generator=g).to(DEV)
    tb = h.time_bias([21, 21], [_hip.COND_NONE, _hip.COND_ONE])
    tbo = ho.time_bias([21, 21], [_hip.COND_NONE, _hip.COND_ONE])
    h.set_precision(_hip.PREC_AUTO)
    splits = {}
    for block in range(1, 8):                       # every 3x3 layer split as deep as its channel count allows
        for slot in (1, 2):
            for sp in (8, 4, 2):
                try:
                    h.set_conv_choice(2 * B, 16, 16, block, slot, 64 if block in (3, 4, 5) else 128, 64, sp, 3 if block != 4 else 1, 0)
                    splits[(block, slot)] = sp
                    break
                except _hip.HipLibraryError:
                    continue
    assert len(splits) >= 10, splits
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x[:3].cpu(), torch.full((3,), 21), torch.ones(3, 1)).numpy()
    first = h.forward(x, tb, 2, B, tune=False).clone()
    assert_close(first[B:B + 3].cpu().numpy(), want, what="split everywhere")
    stop = threading.Event()
    side = torch.cuda.Stream()

    def noise_maker():                              # uneven load: bursts of another model's forwards
        xo = torch.randn(64, 3, 16, 16, device=DEV)
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(3):
                    ho.forward(xo, tbo, 2, 64, tune=False)
                side.synchronize()
    t = threading.Thread(target=noise_maker)
    t.start()
    try:
        for it in range(150):
            again = h.forward(x, tb, 2, B, tune=False)
            assert torch.equal(again, first), f"forward {it} differs from the first one"
    finally:
        stop.set()
        t.join()
    h.set_precision(_hip.PREC_AUTO)


# ------------------------------------------------------------------ round 3: odd chunk counts, mixed batches
@pytest.mark.parametrize("sf", [0.3, 0.8])
def test_multi_chunk_kernels_on_odd_chunk_counts(gpu_models, sf):
    """Layers whose padded channel count is an ODD number of 16-channel chunks (48 / 80: size factor 0.3; 112 / 208: 0.8)
    through the kernels that multiply 2 or 4 chunks per step (prec 4, K split across waves: prec 5) and through channel
    splits: the weight packs carry zero chunks up to a multiple of four and the activation loads of the chunks past the
    layer's own are replaced by zeros (ConvParams::ccw)."""
    m = gpu_models(sf)
    h = engine.UNetHandle.for_module(m)
    B = 23
    x = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(11)).to(DEV)
    tb = h.time_bias([17, 17], [_hip.COND_ZERO, _hip.COND_ONE])
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = unet_ref.unet_forward(sd, x[:3].cpu(), torch.full((3,), 17), torch.ones(3, 1)).numpy()
    tried = 0
    for block in range(8):
        for slot in (1, 2):
            for prec, bm, bn, sp, fuse in [(4, 128, 64, 1, 0), (4, 64, 64, 1, 1), (4, 256, 64, 1, 0), (4, 128, 128, 1, 1), (4, 64, 64, 2, 0),
                                           (5, 64, 64, 1, 0), (5, 64, 64, 1, 1), (5, 128, 64, 1, 1), (5, 128, 64, 2, 0), (5, 64, 128, 1, 1),
                                           (3, 64, 64, 2, 0), (3, 128, 64, 4, 0)]:
                h.set_precision(_hip.PREC_AUTO)
                try:
                    h.set_conv_choice(2 * B, 16, 16, block, slot, bm, bn, sp, prec, fuse if slot == 2 else 0)
                except _hip.HipLibraryError:
                    continue
                kinds = [c for c in h.conv_choices(2 * B, 16, 16) if c[0] == engine.BLOCK_NAMES[block] and c[1] == ("conv1", "conv2")[slot - 1]]
                got = h.forward(x, tb, 2, B, tune=False)
                assert_close(got[B:B + 3].cpu().numpy(), want, what=f"sf {sf} block {block} slot {slot} {prec}/{bm}x{bn}/s{sp}/f{fuse} -> {kinds}")
                tried += 1
    h.set_precision(_hip.PREC_AUTO)
    assert tried >= 100, tried


@pytest.mark.parametrize("sf", [0.2, 1.0])
def test_mixed_batch_forward_matches_oracle(gpu_models, sf):
    """dt_unet_forward_mixed: images [0, b_single) take one pass (cond None), the others the two CFG passes (cond 0 / 1);
    every row against the oracle forward of its (image, t, cond)."""
    m = gpu_models(sf)
    h = engine.UNetHandle.for_module(m)
    S, G = 2, 3                                   # row blocks of S images: 1 single-pass block, G CFG blocks
    B, b_single = (1 + G) * S, S
    x = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(3)).to(DEV)
    t = 23
    tb = h.time_bias([t] * (1 + 2 * G), [_hip.COND_NONE] + [_hip.COND_ZERO] * G + [_hip.COND_ONE] * G)
    got = h.forward_mixed(x, tb, b_single, S, tune=False).cpu().numpy()
    assert got.shape[0] == 2 * B - b_single
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    xc = x.cpu()
    with torch.no_grad():
        tt = lambda n: torch.full((n,), t)
        want = torch.cat([unet_ref.unet_forward(sd, xc[:S], tt(S), None),
                          unet_ref.unet_forward(sd, xc[S:], tt(B - S), torch.zeros(B - S, 1)),
                          unet_ref.unet_forward(sd, xc[S:], tt(B - S), torch.ones(B - S, 1))]).numpy()
    assert_close(got, want, what=f"mixed forward sf {sf}")


def test_mixed_batch_sampler_matches_separate_plans(gpu_models, monkeypatch):
    """One mixed launch sequence (single-pass samples riding in the CFG samples' launches) gives the trajectories of the
    two separate launch sequences: same kernels, same arithmetic per row, other launch shapes (fp32 re-association only)."""
    from distillation_trajectories_amd.analysis.trajectory_engine import sample_grid
    from distillation_trajectories_amd.synthetic import noise_table
    m = gpu_models(0.5)
    h = engine.UNetHandle.for_module(m)
    S, T = 5, 12
    table = noise_table(42, S + T - 1, (1, 3, 16, 16)).reshape(S + T - 1, -1).to(DEV)
    scales = [1.0, None, 3.0, 7.5]
    monkeypatch.setenv("DT_GRID_MERGE", "0")
    sep = {gs: v.clone() for gs, v in sample_grid(h, table, 0, S, T, scales, 16, 16).items()}
    monkeypatch.setenv("DT_GRID_MERGE", "1")
    mix = sample_grid(h, table, 0, S, T, scales, 16, 16)
    assert set(mix) == set(sep)
    for gs in scales:
        assert mix[gs].shape == sep[gs].shape == (T + 1, S, 768)
        scale = float(sep[gs].abs().max())
        err = float((mix[gs] - sep[gs]).abs().max()) / scale
        assert err < 2e-5, (gs, err)
    assert torch.equal(mix[1.0], mix[None])           # one row block serves every scale without the CFG branch


def test_one_handle_on_two_streams_is_race_free(gpu_models):
    """One handle, two HIP streams, launches queued concurrently from two host threads: every forward equals the serial
    result bit for bit (a handle keeps one workspace per stream)."""
    import threading
    m = gpu_models(0.5)
    h = engine.UNetHandle.for_module(m)
    B = 32
    g = torch.Generator().manual_seed(21)
    xs = [torch.randn(B, 3, 16, 16, generator=g).to(DEV) for _ in range(2)]
    tb = h.time_bias([12, 12], [_hip.COND_NONE, _hip.COND_ONE])
    serial = [h.forward(x, tb, 2, B, tune=False).clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs, errs = [[], []], []

    def worker(k):
        try:
            with torch.cuda.stream(streams[k]):
                for _ in range(40):
                    outs[k].append(h.forward(xs[k], tb, 2, B, tune=False))
        except Exception as e:
            errs.append(e)
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    torch.cuda.synchronize()
    assert not errs, errs
    for k in range(2):
        assert all(torch.equal(o, serial[k]) for o in outs[k]), f"stream {k}"


_DIGEST_SCRIPT = r"""
import hashlib, json, sys
sys.path.insert(0, sys.argv[1])
import torch
import bench
from distillation_trajectories_amd._hip import RULE_PSAMPLE
torch.cuda.set_device(0)
spec = bench.CONFIGS[1]
wl = bench.PairWorkload(spec, torch.device("cuda:0"), 0, 256, concurrent=False)
out = {}
for i, name in enumerate(("teacher", "student")):
    h, traj = wl.handles[i], wl.traj[i]
    tb = h.time_bias_general(wl.tb_t, wl.tb_cond, wl.tb_present, 2 * wl.T)
    traj[0].copy_(wl.x_T)
    h.sample(RULE_PSAMPLE, traj, wl.H, wl.H, tb, 2, wl.coef, wl.has_noise, z=wl.z, z_shift=wl.z_shift, w_scalar=spec["guidance"])
    out[name] = {"sha256": hashlib.sha256(traj.cpu().numpy().tobytes()).hexdigest(), "plans": h.plan_ids()}
print("DIGEST " + json.dumps(out))
"""


def test_two_fresh_processes_give_identical_bits():
    """configs[1] (teacher and student, batch 256, T = 50, two passes per step) in two fresh processes: the trajectories are
    bit-identical, because a shape's launch plan comes from the committed table / the deterministic heuristic and nothing is
    timed in the process (the reference's contract is seeded reproducibility, trajectory_engine.py:55-57)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("DT_AUTOTUNE", "DT_TUNE_CACHE")}
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", _DIGEST_SCRIPT, root], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")][-1]
        runs.append(json.loads(line[7:]))
    assert runs[0] == runs[1], runs
    assert all(not v.startswith("tuned") for m in runs[0].values() for v in m["plans"].values()), runs[0]


def test_plan_cache_respects_arithmetic_mode(gpu_models, tmp_path, monkeypatch):
    """A plan measured under the auto mode must not be replayed by a handle in exact-fp32 mode (the cache key
    lacked the mode): record a cache under auto, then PREC_FP32 still reports only fp32 kernels."""
    cache = tmp_path / "tune.json"
    monkeypatch.setenv("DT_TUNE_CACHE", str(cache))
    m = gpu_models(0.5)
    h = engine.UNetHandle.for_module(m)
    h.set_precision(_hip.PREC_AUTO)
    B = 40
    x = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(2)).to(DEV)
    tb = h.time_bias([7, 7], [_hip.COND_NONE, _hip.COND_ONE])
    auto = h.forward(x, tb, 2, B, tune=True)
    assert cache.exists() and h._plans[(2 * B, 16, 16, B, 0)].startswith("tuned:")
    kinds = {c[5] for c in h.conv_choices(2 * B, 16, 16)}
    assert any(k.startswith("split-bf16") for k in kinds), kinds
    # a second handle of the same model replays the recorded plan (no timing) ...
    h2 = engine.UNetHandle(m.state_dict(), torch.device(DEV))
    again = h2.forward(x, tb, 2, B)
    assert h2._plans[(2 * B, 16, 16, B, 0)].startswith("cache:") and torch.equal(again, auto)
    # ... and one in exact-fp32 mode does not: other key, fp32 kernels only
    h2.set_precision(_hip.PREC_FP32)
    exact = h2.forward(x, tb, 2, B)
    assert {c[5].replace("+skip", "") for c in h2.conv_choices(2 * B, 16, 16)} == {"fp32"}
    assert_close(exact.cpu().numpy(), auto.cpu().numpy(), rtol=1e-5, atol=1e-6, what="fp32 vs auto")
    h.set_precision(_hip.PREC_AUTO)


# ------------------------------------------------------------------ round 3: batch aggregates, resize, eight scales
def _check_aggregate(got, want, rel):
    assert set(got) == set(want), set(got) ^ set(want)
    for k, w in want.items():
        g = got[k]
        if isinstance(w, list):
            assert len(g) == len(w), k
            if w:
                assert_close(np.array(g, dtype=np.float64), np.array(w, dtype=np.float64), rtol=rel, atol=1e-7, what=k)
        else:
            assert_close(float(g), float(w), rtol=rel, atol=1e-9, what=k)


def test_r03_metrics_batch_aggregates_golden(golden_r03, gpu_models, tmp_path):
    """TrajectoryManager.compute_trajectory_metrics_batch (utils/trajectory_manager.py:434-548) against the reference's own
    aggregate over three generated-and-stored pairs: every list, every ``_avg`` key, wasserstein_distances_per_timestep --
    equal lengths (21 / 21 states) and the interp1d path (21 / 6).  The pairs of a batch are reduced by one launch of each
    metric kernel (compute_trajectory_metrics_many)."""
    from distillation_trajectories_amd.utils.trajectory_manager import TrajectoryManager
    _, meta = golden_r03
    for n, c in enumerate(meta["batch_metric_cases"]):
        cfg = Config()
        cfg.image_size, cfg.sample_steps = 16, c["sample_steps"]
        cfg.teacher_steps, cfg.student_steps = c["teacher_steps"], c["student_steps"]
        cfg.trajectory_dir = str(tmp_path / f"traj{n}")
        man = TrajectoryManager(gpu_models(c["teacher_sf"]), gpu_models(c["student_sf"]), cfg, size_factor=c["student_sf"])
        assert len(man.generate_and_save_trajectories(num_samples=c["num_samples"])) == c["num_samples"]
        np.random.seed(c["np_seed"])
        got = man.compute_trajectory_metrics_batch()
        _check_aggregate(got, c["result"], rel=1e-4)
        # batch_size = 1 walks the same pairs one launch group at a time: identical numbers
        np.random.seed(c["np_seed"])
        one = man.compute_trajectory_metrics_batch(batch_size=1)
        assert all(np.array_equal(np.asarray(one[k], dtype=np.float64), np.asarray(got[k], dtype=np.float64)) for k in got if k != "architecture_type")


def test_r03_student_resize_branches_golden(golden_r03, gpu_models, tmp_path):
    """The student at another resolution: dt_resize_bilinear in the manager (utils/trajectory_manager.py:120-122,153-163) and
    in compute_trajectory_metrics (analysis/metrics/trajectory_metrics.py:40-52), against the reference's outputs."""
    import copy
    from distillation_trajectories_amd.analysis.metrics.trajectory_metrics import compute_trajectory_metrics
    from distillation_trajectories_amd.utils.trajectory_manager import TrajectoryManager
    arrays, meta = golden_r03
    c = meta["resize_case"]
    cfg = Config()
    cfg.image_size, cfg.sample_steps = 16, c["sample_steps"]
    cfg.teacher_steps, cfg.student_steps = c["teacher_steps"], c["student_steps"]
    cfg.trajectory_dir = str(tmp_path / "traj")
    student = copy.deepcopy(gpu_models(c["student_sf"]))
    student.image_size = c["student_image_size"]
    man = TrajectoryManager(gpu_models(c["teacher_sf"]), student, cfg, size_factor=c["student_sf"])
    tt, st = man.generate_trajectory(seed=c["seed"])
    assert [t for _, t in tt] == c["teacher_t"] and [t for _, t in st] == c["student_t"]
    assert st[0][0].shape == (1, 3, 16, 16)
    assert_close(torch.stack([x.cpu() for x, _ in tt]).numpy(), arrays["resize_teacher"], rtol=1e-4, atol=1e-4)
    assert_close(torch.stack([x.cpu() for x, _ in st]).numpy(), arrays["resize_student"], rtol=1e-4, atol=1e-4)
    # the first stored student state is the resized start noise: isolates the resize kernel (no model in between)
    assert_close(st[0][0].cpu().numpy(), arrays["resize_student"][0], rtol=1e-6, atol=1e-6, what="resized x_T")
    a = [torch.from_numpy(x) for x in arrays["resize_metric_teacher"]]
    b = [torch.from_numpy(x) for x in arrays["resize_metric_student32"]]
    np.random.seed(c["metric_np_seed"])
    _check_metrics(compute_trajectory_metrics(a, b, cfg), c["metrics"], rel=2e-5)


@pytest.mark.parametrize("shape", [((2, 3, 32, 32), (16, 16)), ((1, 3, 16, 16), (32, 32)), ((3, 1, 10, 7), (13, 22)), ((1, 2, 5, 5), (5, 5)),
                                   ((2, 3, 8, 8), (1, 1))])
def test_resize_bilinear_matches_torch(shape):
    """dt_resize_bilinear against torch.nn.functional.interpolate(mode='bilinear', align_corners=True) on the CPU: down, up,
    odd non-square sizes, identity and a single output pixel."""
    (N, C, h, w), size = shape
    x = torch.randn(N, C, h, w, generator=torch.Generator().manual_seed(h * 100 + w))
    want = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=True)
    got = engine.resize_bilinear(x.to(DEV), size)
    assert got.shape == want.shape
    assert_close(got.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-6, what=str(shape))


def test_r03_eight_scale_grid_cell_golden(golden_r03, gpu_models):
    """configs[3]: the eight guidance scales {1, 2, 3, 5, 7.5, 10, 15, 20} (scripts/analysis/analyze_trajectory_metrics.py:40-42)
    in ONE mixed launch sequence per model (1 single-pass + 7 CFG row blocks: 15 x S rows per forward), through
    compare_trajectories and through the grid driver, against the reference's cell."""
    from distillation_trajectories_amd.analysis.trajectory_engine import compare_trajectories
    from distillation_trajectories_amd.grid import grid_metrics
    _, meta = golden_r03
    c = meta["grid8_cell"]
    cfg = Config()
    cfg.image_size, cfg.timesteps = 16, c["T"]
    teacher, student = gpu_models(c["teacher_sf"]), gpu_models(c["student_sf"])
    res = compare_trajectories(teacher, student, cfg, guidance_scales=c["guidance_scales"], size_factor=c["student_sf"],
                               num_samples=c["num_samples"])
    for side in ("teacher_metrics", "student_metrics"):
        for gs in c["guidance_scales"]:
            _check_metrics(res[side][gs], c["result"][side][str(gs)], rel=1e-4)
    grid = grid_metrics(teacher, [student], cfg, c["guidance_scales"], num_samples=c["num_samples"], rank=0, world=1)
    for gs in c["guidance_scales"]:
        want = c["result"]["teacher_metrics"][str(gs)]
        assert set(want) <= set(grid[0][gs])      # (the reference's average skips np.float32-valued keys such as path_alignment)
        for k, w in want.items():
            v = grid[0][gs][k]
            if math.isnan(w):
                assert math.isnan(v), (gs, k)
            else:
                assert_close(v, w, rtol=1e-4, atol=1e-9, what=f"grid gs={gs} {k}")


def test_non_finite_input_stays_in_its_own_image(gpu_models):
    """An Inf pixel: the reference's fp32 convolution carries Inf / NaN through that image; here the three-way bf16 split turns
    Inf into NaN one layer earlier (Inf - Inf) -- either way the image's prediction is non-finite, which is what is pinned,
    together with the property that matters for a batched sampler: the OTHER images of the batch are bit-identical to a
    clean run (rows are independent; zero-padded channels of the poisoned rows never leak)."""
    m = gpu_models(0.5)
    h = engine.UNetHandle.for_module(m)
    B = 5
    x = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(8)).to(DEV)
    tb = h.time_bias([30, 30], [_hip.COND_NONE, _hip.COND_ONE])
    clean = h.forward(x, tb, 2, B, tune=False).clone()
    bad = x.clone()
    bad[2, 1, 7, 9] = float("inf")
    got = h.forward(bad, tb, 2, B, tune=False)
    keep = [0, 1, 3, 4]
    for p in (0, 1):
        assert torch.equal(got[p * B:(p + 1) * B][keep], clean[p * B:(p + 1) * B][keep])
        assert not torch.isfinite(got[p * B + 2]).all()
    assert torch.isfinite(clean).all()


def test_four_channel_images_are_rejected_cleanly():
    """The library supports 1-3 image channels (the reference's datasets: MNIST 1, CIFAR-10 3); a 4-channel model fails at
    dt_unet_create with DT_E_SHAPE -- raised, not a silent fallback."""
    from distillation_trajectories_amd.models import DiffusionUNet
    from distillation_trajectories_amd.synthetic import make_model
    cfg = Config()
    cfg.image_size, cfg.channels = 16, 4
    m = make_model(DiffusionUNet, cfg, 0.2).to(DEV)
    with pytest.raises(_hip.HipLibraryError, match="unsupported or inconsistent shape"):
        m(torch.zeros(1, 4, 16, 16, device=DEV), torch.tensor([3], device=DEV))
