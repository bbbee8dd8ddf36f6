"""Oracle: the diffusion U-Net forward on torch-CPU ops, driven by a state_dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates reference
``models.py:15-39`` (sinusoidal embedding), ``models.py:59-83`` (Block.forward)
and ``models.py:159-224`` (DiffusionUNet.forward) as free functions over the
reference's ``state_dict`` layout, with the same op order so that on the same
torch build the result is bit-identical to the reference module in eval mode.
"""
import math

import torch
import torch.nn.functional as F

BLOCKS = ("enc1", "enc2", "enc3", "enc4", "bottleneck", "dec3", "dec2", "dec1")


def sinusoidal_embedding(t, dim):
    """reference models.py:15-39.  t: integer/float tensor [B] or [B,1] -> [B, dim] fp32."""
    dim = max(dim, 2)
    half = max(dim // 2, 1)
    scale = math.log(10000) / (half - 1 + 1e-8)
    freqs = torch.exp(torch.arange(half) * -scale)
    if t.dim() > 1:
        t = t.squeeze(-1)
    arg = t[:, None] * freqs[None, :]
    emb = torch.cat((arg.sin(), arg.cos()), dim=-1)
    if emb.shape[-1] < dim:            # odd dim: zero-pad (models.py:33-36)
        emb = torch.cat((emb, torch.zeros(emb.shape[0], dim - emb.shape[-1])), dim=-1)
    elif emb.shape[-1] > dim:
        emb = emb[:, :dim]
    return emb


def time_embedding(sd, t, cond=None):
    """reference models.py:175-185: time MLP plus optional condition embedding."""
    t = t.unsqueeze(-1) if t.dim() == 1 else t
    d = sd["time_mlp.1.weight"].shape[0]
    temb = F.relu(F.linear(sinusoidal_embedding(t, d), sd["time_mlp.1.weight"], sd["time_mlp.1.bias"]))
    if cond is not None:
        c = F.relu(F.linear(cond, sd["cond_emb.0.weight"], sd["cond_emb.0.bias"]))
        c = F.linear(c, sd["cond_emb.2.weight"], sd["cond_emb.2.bias"])
        temb = temb + c
    return temb


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training=False, eps=1e-5)


def block_forward(sd, name, x, temb):
    """reference models.py:59-83 (eval mode)."""
    rkey = name + ".residual_conv.weight"
    res = F.conv2d(x, sd[rkey], sd[name + ".residual_conv.bias"]) if rkey in sd else x
    h = F.relu(_bn(sd, name + ".norm1", F.conv2d(x, sd[name + ".conv1.weight"], sd[name + ".conv1.bias"], padding=1)))
    tb = F.relu(F.linear(temb, sd[name + ".time_mlp.weight"], sd[name + ".time_mlp.bias"]))
    h = h + tb[:, :, None, None].expand(-1, -1, h.size(2), h.size(3))
    h = F.relu(_bn(sd, name + ".norm2", F.conv2d(h, sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], padding=1)))
    return h + res


def _up(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def unet_forward(sd, x, t, cond=None, return_activations=False):
    """reference models.py:159-224 in eval mode (dropout is identity)."""
    temb = time_embedding(sd, t, cond)
    acts = {}
    x1 = block_forward(sd, "enc1", x, temb)
    x2 = block_forward(sd, "enc2", F.max_pool2d(x1, 2), temb)
    x3 = block_forward(sd, "enc3", F.max_pool2d(x2, 2), temb)
    x4 = block_forward(sd, "enc4", F.max_pool2d(x3, 2), temb)
    xb = block_forward(sd, "bottleneck", F.max_pool2d(x4, 2), temb)
    d3 = block_forward(sd, "dec3", torch.cat([_up(xb), x4], dim=1), temb)
    d2 = block_forward(sd, "dec2", torch.cat([_up(d3), x3], dim=1), temb)
    d1 = block_forward(sd, "dec1", torch.cat([_up(d2), x2], dim=1), temb)
    out = F.conv2d(_up(d1), sd["final.weight"], sd["final.bias"])
    if return_activations:
        acts.update(temb=temb, enc1=x1, enc2=x2, enc3=x3, enc4=x4, bottleneck=xb, dec3=d3, dec2=d2, dec1=d1)
        return out, acts
    return out


def macs_per_forward(sd, h, w):
    """Multiply-accumulates of one forward per sample (convs + linears), for roofline bookkeeping."""
    total = 0
    res = {"enc1": 1, "enc2": 2, "enc3": 4, "enc4": 8, "bottleneck": 16, "dec3": 8, "dec2": 4, "dec1": 2}
    for name in BLOCKS:
        hh, ww = h // res[name], w // res[name]
        for conv in ("conv1", "conv2", "residual_conv"):
            k = f"{name}.{conv}.weight"
            if k in sd:
                total += sd[k].numel() * hh * ww
        total += sd[f"{name}.time_mlp.weight"].numel()
    total += sd["final.weight"].numel() * h * w
    total += sd["time_mlp.1.weight"].numel() + sd["cond_emb.0.weight"].numel() + sd["cond_emb.2.weight"].numel()
    return total
