"""FID-input sampling and the FID formula (reference analysis/metrics/fid_score.py:61-93, 199-318).

The reference's fourth sampler (``p_sample_loop(model, x, config)``, :261-318) is the textbook DDPM posterior
with beta_t = beta_start + (beta_end-beta_start)*t/T and a running-product alpha_bar; its step
``x = (x - c0*eps)/sqrt(alpha_t) [+ sqrt(beta_t)*z]`` has exactly the shape of the fused update kernel's
MANAGER rule, so it runs device-resident with per-step coefficients computed on the host the way the
reference computes them (python floats -> fp32 0-dim tensors).  ``generate_samples`` batches all samples into
one loop (the reference runs them one by one); noise is drawn from the CPU generator in the reference's
order.  ``calculate_fid`` is the reference's numpy/scipy formula.  The InceptionV3 feature extractor needs
pretrained weights that cannot be fetched offline and is not provided.
"""
import numpy as np
import torch
from scipy import linalg

from ... import engine
from ..._hip import COND_NONE, RULE_MANAGER


def calculate_fid(features_1, features_2):
    """Fréchet distance between two feature sets (reference :61-93), 999.0 placeholder below 2 samples."""
    if len(features_1) < 2 or len(features_2) < 2:
        print("  Warning: Not enough samples for a proper FID calculation.")
        print(f"  Number of samples in set 1: {len(features_1)}")
        print(f"  Number of samples in set 2: {len(features_2)}")
        print("  Returning a placeholder FID score of 999.0")
        return 999.0
    mu1, sigma1 = features_1.mean(axis=0), np.cov(features_1, rowvar=False)
    mu2, sigma2 = features_2.mean(axis=0), np.cov(features_2, rowvar=False)
    ssdiff = np.sum((mu1 - mu2) ** 2.0)
    covmean = linalg.sqrtm(sigma1.dot(sigma2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return ssdiff + np.trace(sigma1 + sigma2 - 2.0 * covmean)


def posterior_coefficients(config):
    """[(c0, sqrt(alpha_t), sqrt(beta_t))] for t = T-1..0 with the reference's arithmetic (:284-305)."""
    T = config.timesteps
    rows = []
    for t in range(T - 1, -1, -1):
        beta_t = config.beta_start + (config.beta_end - config.beta_start) * t / T
        alpha_t = 1.0 - beta_t
        alpha_bar_t = 1.0
        for i in range(t + 1):
            alpha_bar_t *= 1.0 - (config.beta_start + (config.beta_end - config.beta_start) * i / T)
        b, a, ab = torch.as_tensor(beta_t), torch.as_tensor(alpha_t), torch.as_tensor(alpha_bar_t)
        rows.append((float((1 - a) / torch.sqrt(1 - ab)), float(torch.sqrt(a)), float(torch.sqrt(b))))
    return rows


def _run(model, x0, config, z):
    """Final samples [S,C,H,W] (device) for start images x0[S,C,H,W] and step noise z[T-1,S,E]."""
    h = engine.UNetHandle.for_module(model)
    S, C, H, W = x0.shape
    E, T = C * H * W, config.timesteps
    device = next(model.parameters()).device
    traj = torch.empty(T + 1, S, E, dtype=torch.float32, device=device)
    traj[0].copy_(x0.reshape(S, E))
    order = list(range(T - 1, -1, -1))
    tb = h.time_bias(order, [COND_NONE] * T)
    h.sample(RULE_MANAGER, traj, H, W, tb, 1, posterior_coefficients(config), [t > 0 for t in order],
             z=None if z is None else z.reshape(-1, E).to(device), z_shift=[k * S for k in range(T)])
    return traj[T].reshape(S, C, H, W).clone()


def p_sample_loop(model, x, config):
    """Reference :261-318: denoise ``x`` through all config.timesteps steps; returns the final sample."""
    model.eval()
    T = config.timesteps
    zs = [torch.randn(x.shape) for _ in range(T - 1)]            # one CPU-generator draw per step with t > 0
    z = torch.stack(zs).reshape(T - 1, x.shape[0], -1) if zs else None
    return _run(model, x.detach().float(), config, z)


def generate_samples(model, config, num_samples, device, fixed_samples=None):
    """Reference :199-259: ``num_samples`` final samples [N,C,H,W]; all of them advance in one batched loop."""
    model.eval()
    image_size = getattr(model, "image_size", config.image_size)
    T = config.timesteps
    starts, zs = [], []
    if fixed_samples is not None:
        print(f"    Using {min(num_samples, len(fixed_samples))} fixed samples as starting points")
        n = len(fixed_samples[:num_samples])
    else:
        n = num_samples
    for i in range(n):           # reference draw order: start noise of sample i, then its T-1 step draws
        if fixed_samples is not None:
            x = fixed_samples[i:i + 1].clone().cpu().float()
            if x.shape[2] != image_size or x.shape[3] != image_size:
                x = engine.resize_bilinear(x, (image_size, image_size)).cpu()
        else:
            x = torch.randn(1, config.channels, image_size, image_size)
        starts.append(x)
        zs.append(torch.stack([torch.randn(x.shape) for _ in range(T - 1)]) if T > 1 else None)
    x0 = torch.cat(starts)
    z = torch.stack(zs, dim=1).reshape(T - 1, n, -1) if T > 1 else None
    return _run(model, x0, config, z)
