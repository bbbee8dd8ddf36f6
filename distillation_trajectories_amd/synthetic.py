"""Synthetic, seed-defined inputs for tests and the benchmark.

The reference ships no checkpoints, trajectories or image data, so every model is
a seeded random initialisation:  ``torch.manual_seed(1000 + round(100*sf))`` then
``DiffusionUNet(cfg, sf).eval()``, with the BatchNorm statistics (and affine terms)
randomised from ``Generator(7)`` -- default-initialised BN is the identity and would
hide folding bugs.  Start noise of sample ``s`` is ``randn(1,C,H,W)`` drawn from the
torch **CPU** generator right after ``manual_seed(42+s)``; the step noise of
(seed, t) is the same draw after ``manual_seed(seed+t)`` (trajectory_engine.py:86-95).
"""
import hashlib

import torch
import torch.nn as nn


def randomize_bn(model, seed=7):
    """mean~N(0,0.1^2), var~U(0.75,1.25), gamma~U(0.8,1.2), beta~N(0,0.05^2); works on any nn.Module."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.normal_(0.0, 0.1, generator=g)
                mod.running_var.uniform_(0.75, 1.25, generator=g)
                mod.weight.uniform_(0.8, 1.2, generator=g)
                mod.bias.normal_(0.0, 0.05, generator=g)
    return model


def model_seed(size_factor):
    return 1000 + round(100 * size_factor)


def make_model(model_cls, cfg, size_factor, bn_seed=7, quiet=True, seed=None):
    """Seeded random-init model in eval mode with randomised BN (see module docstring); ``seed`` overrides
    the size-factor-derived seed (two different models of one size factor)."""
    import contextlib, io
    torch.manual_seed(model_seed(size_factor) if seed is None else seed)
    with (contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()):
        m = model_cls(cfg, size_factor)
    return randomize_bn(m.eval(), bn_seed)


def state_dict_digest(sd):
    """sha256 over the raw bytes of every tensor in key order -- pins 'same weights' across machines."""
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def seeded_noise(seed, shape):
    """``randn(shape)`` of the torch CPU generator right after ``manual_seed(seed)``.

    Leaves the global generator in the same state the reference would (it re-seeds the
    global generator too), which callers of the numpy/torch global RNG rely on.
    """
    torch.manual_seed(seed)
    return torch.randn(*shape)


def noise_table(first_seed, count, shape):
    """[count, *shape] table N[k] = seeded_noise(first_seed + k): start noise of sample seed s is
    N[s-first_seed]; step noise of (seed s, step t) is N[s+t-first_seed]."""
    state = torch.get_rng_state()
    out = torch.stack([seeded_noise(first_seed + k, shape) for k in range(count)])
    torch.set_rng_state(state)
    return out
