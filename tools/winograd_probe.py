#!/usr/bin/env python3
"""Would Winograd F(2x2, 3x3) pay for the big 3x3 layers?  

F(2x2, 3x3) turns a 3x3 convolution over M pixels into 16 independent GEMMs [M/4 tiles x Cin] x [Cin x Cout] (2.25x fewer
multiplications) between an input transform (16 values per 4 pixels and channel) and an output transform (16 -> 4 values).
Whatever the transforms cost, the matrix stage alone has to beat the direct strip kernel.  This probe times, with the
library's own kernels through dt_unet_time_conv:
  direct : the layer itself under its best strip-kernel plan (rows = pixels)
  matrix : ONE 1x1 convolution with 16 * M/4 = 4 M rows, the same Cin -> Cout -- the 16 batched GEMMs' arithmetic and
           operand traffic exactly (their weights differ per GEMM, which changes no byte count: each weight matrix is read
           once per row tile either way), under every tile of the split-bf16 GEMM kernel
for the teacher's enc2.conv2 (256 -> 256 at 8x8) and enc1.conv2 / dec1.conv2 (128 -> 128), batch 512 rows of 16x16.
A 1x1 skip convolution with the wanted channel counts exists in a model whose decoder concat has Cin channels:
dims [c, Cout, Cin / 2, Cin / 2] -> dec2's skip conv maps 2 * dims[2] = Cin -> dims[1] = Cout at 4x4.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # noqa: E402

from distillation_trajectories_amd import _hip, engine   # noqa: E402
from distillation_trajectories_amd.config import Config   # noqa: E402
from distillation_trajectories_amd.models import DiffusionUNet   # noqa: E402
from distillation_trajectories_amd.synthetic import make_model   # noqa: E402

dev = torch.device("cuda:0")
lib = _hip.load()


def time_conv(h, Bt, block, slot, bm, bn, sp, prec, fuse=0, reps=10):
    ws = h.workspace(Bt, 16, 16)
    ms, fl = ctypes.c_float(), ctypes.c_double()
    st = lib.dt_unet_time_conv(h.h, Bt, 16, 16, block, slot, bm, bn, sp, prec, fuse, reps, _hip.ptr(ws), ws.numel(), _hip.stream_ptr(),
                               ctypes.byref(ms), ctypes.byref(fl))
    return (ms.value * 1e3, fl.value) if st == 0 and fl.value > 0 else None


def custom_handle(dims, D=64):
    """A handle for arbitrary dims (the C ABI takes any dims[4]; the reference's constructor ties them to one size factor)."""
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.05)
    C = 3
    cin = [C, dims[0], dims[1], dims[2], dims[3], 2 * dims[3], 2 * dims[2], 2 * dims[1]]
    cout = [dims[0], dims[1], dims[2], dims[3], dims[3], dims[2], dims[1], dims[0]]
    sd = {"time_mlp.1.weight": r(D, D), "time_mlp.1.bias": r(D), "cond_emb.0.weight": r(D, 1), "cond_emb.0.bias": r(D),
          "cond_emb.2.weight": r(D, D), "cond_emb.2.bias": r(D), "final.weight": r(C, dims[0], 1, 1), "final.bias": r(C)}
    for j, name in enumerate(engine.BLOCK_NAMES):
        sd[f"{name}.time_mlp.weight"], sd[f"{name}.time_mlp.bias"] = r(cout[j], D), r(cout[j])
        sd[f"{name}.conv1.weight"], sd[f"{name}.conv1.bias"] = r(cout[j], cin[j], 3, 3), r(cout[j])
        sd[f"{name}.conv2.weight"], sd[f"{name}.conv2.bias"] = r(cout[j], cout[j], 3, 3), r(cout[j])
        for k in (1, 2):
            sd[f"{name}.norm{k}.weight"], sd[f"{name}.norm{k}.bias"] = torch.ones(cout[j]), torch.zeros(cout[j])
            sd[f"{name}.norm{k}.running_mean"], sd[f"{name}.norm{k}.running_var"] = torch.zeros(cout[j]), torch.ones(cout[j])
        if cin[j] != cout[j]:
            sd[f"{name}.residual_conv.weight"], sd[f"{name}.residual_conv.bias"] = r(cout[j], cin[j], 1, 1), r(cout[j])
    return engine.UNetHandle(sd, dev)


def best(rows):
    rows = [r for r in rows if r[0] is not None]
    return min(rows, key=lambda r: r[0][0]) if rows else None


cfg = Config(); cfg.image_size = 16
teacher = engine.UNetHandle.for_module(make_model(DiffusionUNet, cfg, 1.0).to(dev))
x = torch.randn(256, 3, 16, 16, device=dev)
tb = teacher.time_bias([10, 10], [_hip.COND_NONE, _hip.COND_ONE])
teacher.forward(x, tb, 2, 256)                       # table plan + real activations in the workspace
cases = (("enc2.conv2  256->256 @8x8 ", 1, 2, 256, 256, 512 * 64), ("dec1.conv2  128->128 @8x8 ", 7, 2, 128, 128, 512 * 64))
strip = [(bm, bn, sp, pr) for pr in (3, 4, 5) for (bm, bn) in ((256, 64), (128, 128), (128, 64), (64, 128), (64, 64)) for sp in (1, 2)]
gemm = [(bm, bn, sp, 1) for (bm, bn) in ((128, 128), (128, 64), (64, 128), (64, 64)) for sp in (1, 2)]
for name, blk, slot, ci, co, M in cases:
    d = best([(time_conv(teacher, 512, blk, slot, bm, bn, sp, pr), (bm, bn, sp, pr)) for bm, bn, sp, pr in strip])
    us_direct, fl = d[0]
    # the matrix stage: 4 M rows of a 1x1 conv ci -> co, as dec2's skip conv of a model with dims [64, co, ci/2, ci/2] at batch
    # 4 M / 16 rows (dec2 runs at 4x4 = 16 pixels per image)
    # (a residual_conv exists only where in_ch != out_ch, models.py:53-56: the probe model has co - 16 output channels, which the
    # library pads back to co columns -- the launch does the work of the co-column GEMM; its FLOPs are counted for co)
    hw = custom_handle([64, co - 16, ci // 2, ci // 2])
    Bt = 4 * M // 16
    xs = torch.randn(Bt, 3, 16, 16, device=dev)
    hw.forward(xs, hw.time_bias([10], [_hip.COND_NONE]), 1, Bt, tune=False)
    g = best([(time_conv(hw, Bt, 6, 0, bm, bn, sp, pr), (bm, bn, sp, pr)) for bm, bn, sp, pr in gemm])
    us_mat, flm = g[0]
    flm *= co / (co - 16.0)
    in_bytes, v_bytes = 4.0 * M * ci, 4.0 * 4 * M * ci           # the input tensor; its 16-per-4-pixels transform in fp32
    print(f"{name} direct {us_direct:7.1f} us ({fl / us_direct / 1e6:5.0f} TF/s, plan {d[1]})   matrix stage of F(2x2,3x3) {us_mat:7.1f} us "
          f"({flm / us_mat / 1e6:5.0f} TF/s on {flm / 1e9:5.2f} GF = direct / {fl / flm:4.2f}, tile {g[1][:3]})   "
          f"transformed input {v_bytes / 1e6:5.1f} MB (x4 of {in_bytes / 1e6:5.1f} MB): {2 * v_bytes / 4e6:5.1f} us to write + re-read at 4 TB/s", flush=True)
    del hw
    torch.cuda.empty_cache()
