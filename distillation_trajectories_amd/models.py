"""Weight container + forward entry for the size-scalable diffusion U-Net.

The reference model (``models.py:85-224``) is the *weight interchange format*
of the hot path: checkpoints are plain ``state_dict`` files
(``scripts/train_teacher.py:86``) with 146 tensors whose names are fixed by the
module attribute names.  This module declares the same parameter tree -- same
attribute names, same construction order, so ``torch.manual_seed(k)`` followed
by ``DiffusionUNet(cfg, sf)`` yields bit-identical tensors to the reference --
but it owns **no** ATen forward: ``forward`` hands the packed weights to the
hand-written gfx950 kernels in ``csrc/`` (see ``engine.py``).  There is no CPU
fallback; calling ``forward`` without the HIP library or with host tensors
raises.

Architecture facts restated from the reference (not code):
  * ``time_emb_dim = max(int(256*sf), 16)``, ``base = max(int(128*sf), 16)``,
    ``dims = [max(16, int(base*m)) for m in (1, 2, 2, 2)]``  (models.py:101-110)
  * nine residual blocks enc1..enc4, bottleneck, dec3..dec1; each block is
    conv3x3 -> BN -> ReLU -> (+ReLU(Linear(temb))) -> conv3x3 -> BN -> ReLU,
    plus a 1x1 (or identity) skip of the block input (models.py:59-83)
  * MaxPool2d(2) between encoder levels, bilinear x2 (align_corners=True)
    before each decoder block and before the 1x1 head (models.py:134-135,188-224)
"""
import torch
import torch.nn as nn

BLOCK_NAMES = ("enc1", "enc2", "enc3", "enc4", "bottleneck", "dec3", "dec2", "dec1")


def unet_dims(size_factor):
    """(time_emb_dim, dims[4]) for a size factor -- reference models.py:101-110."""
    d = max(int(256 * size_factor), 16)
    base = max(int(128 * size_factor), 16)
    return d, [max(16, int(base * m)) for m in (1, 2, 2, 2)]


def block_channels(channels, dims):
    """(in_ch, out_ch) of the eight named blocks in construction order (models.py:138-154)."""
    d0, d1, d2, d3 = dims
    return [
        (channels, d0), (d0, d1), (d1, d2), (d2, d3), (d3, d3),
        (d3 + d3, d2), (d2 + d2, d1), (d1 + d1, d0),
    ]


class SinusoidalPositionEmbeddings(nn.Module):
    """Parameter-free placeholder at index 0 of ``time_mlp`` (keeps key ``time_mlp.1.*``).

    The embedding itself (reference models.py:15-39) is evaluated on the device by
    ``time_bias_kernel`` (csrc/dt_layers.hip, C entry ``dt_unet_time_bias``) together with the time /
    condition MLPs; ``frequencies`` reproduces the host-side frequency vector so that the device only
    does ``sin/cos(t * f_k)``.
    """

    def __init__(self, dim):
        super().__init__()
        self.dim = max(dim, 2)

    def frequencies(self):
        import math
        half = max(self.dim // 2, 1)
        scale = math.log(10000) / (half - 1 + 1e-8)
        # int64 arange times a python float -> float32 product, then float32 exp,
        # exactly the dtype path torch takes in the reference (models.py:20-21)
        return torch.exp(torch.arange(half) * -scale)


class Block(nn.Module):
    """Parameter holder of one residual block; same keys as reference models.py:45-57."""

    def __init__(self, in_ch, out_ch, time_emb_dim=None):
        super().__init__()
        self.time_mlp = nn.Linear(time_emb_dim, out_ch) if time_emb_dim else None
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.norm1 = nn.BatchNorm2d(out_ch)
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.norm2 = nn.BatchNorm2d(out_ch)
        self.relu = nn.ReLU()
        self.residual_conv = nn.Conv2d(in_ch, out_ch, 1) if in_ch != out_ch else nn.Identity()


class DiffusionUNet(nn.Module):
    """state_dict-compatible U-Net whose forward runs on hand-written HIP kernels."""

    def __init__(self, config, size_factor=1.0):
        super().__init__()
        self.channels = config.channels
        self.size_factor = size_factor
        self.time_emb_dim, self.dims = unet_dims(size_factor)
        self.base_channels = max(int(128 * size_factor), 16)
        self.channel_multipliers = [1, 2, 2, 2]
        # the reference announces every construction on stdout (models.py:113-114)
        print(f"Model size factor: {size_factor}")
        print(f"Model dimensions: {self.dims}")

        d = self.time_emb_dim
        self.dropout = nn.Dropout(config.dropout)   # identity in eval; the sampler always evals
        self.time_mlp = nn.Sequential(SinusoidalPositionEmbeddings(d), nn.Linear(d, d), nn.ReLU())
        self.cond_emb = nn.Sequential(nn.Linear(1, d), nn.ReLU(), nn.Linear(d, d))
        self.pool = nn.MaxPool2d(2)
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        for name, (cin, cout) in zip(BLOCK_NAMES, block_channels(self.channels, self.dims)):
            setattr(self, name, Block(cin, cout, d))
        self.final = nn.Conv2d(self.dims[0], self.channels, 1)

    def forward(self, x, t, cond=None):
        """eps = U-Net(x[B,C,H,W], t[B] or [1], cond[B,1] or None) on the HIP path."""
        from . import engine
        return engine.unet_forward_module(self, x, t, cond)


class SimpleUNet(DiffusionUNet):
    """Teacher alias (reference models.py:227-232)."""

    def __init__(self, config):
        super().__init__(config, size_factor=1.0)


class StudentUNet(DiffusionUNet):
    """Student alias; ``architecture_type`` is accepted and ignored (models.py:234-243)."""

    def __init__(self, config, size_factor=1.0, architecture_type=None):
        if architecture_type is not None:
            print(f"Warning: architecture_type '{architecture_type}' is ignored in the new unified model architecture")
        super().__init__(config, size_factor=size_factor)
