"""Host-side driver of the HIP path: weight hand-off, time-bias tables, device-resident
reverse loops and metric post-processing.  PyTorch is used for device memory, streams and
host<->device copies only; all arithmetic on trajectories happens in csrc/ kernels.

Everything here raises if the HIP library is missing or a tensor is not on a CUDA(HIP)
device -- the product has no CPU path (the CPU restatement lives in oracle/ and is test-only).
"""
import ctypes
import math
import os
import sys
from ctypes import c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import numpy as np
import torch

from . import _hip
from ._hip import (COND_NONE, COND_ONE, COND_ZERO, RULE_ENGINE, RULE_MANAGER, RULE_PSAMPLE, HipLibraryError, check,
                   ptr, stream_ptr)

AUTOTUNE_MIN_ROWS = 16384       # batch_total*H*W from which a forward shape is worth tuning once
BLOCK_NAMES = ("enc1", "enc2", "enc3", "enc4", "bottleneck", "dec3", "dec2", "dec1")
_BLOCK_KEYS = ("time_mlp.weight", "time_mlp.bias",
               "conv1.weight", "conv1.bias", "norm1.weight", "norm1.bias", "norm1.running_mean", "norm1.running_var",
               "conv2.weight", "conv2.bias", "norm2.weight", "norm2.bias", "norm2.running_mean", "norm2.running_var",
               "residual_conv.weight", "residual_conv.bias")
_GLOBAL_KEYS = ("time_mlp.1.weight", "time_mlp.1.bias", "cond_emb.0.weight", "cond_emb.0.bias",
                "cond_emb.2.weight", "cond_emb.2.bias", "final.weight", "final.bias")


def _require_cuda(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipLibraryError(f"{what} must be a CUDA(HIP) tensor: the MI355X path has no CPU fallback "
                              "(the CPU restatement lives in oracle/ and is for tests only)")


def sinusoid_frequencies(dim):
    """models.py:17-21: exp(arange(half) * -(ln 1e4 / (half-1+1e-8))) with torch's fp32 dtype path."""
    half = max(max(dim, 2) // 2, 1)
    return torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1 + 1e-8)))


# (DT_PLAN_TABLE=file: another table, e.g. a candidate from tools/plan_search.py; DT_PLAN_TABLE= (empty): no table)
PLAN_TABLE = os.environ.get("DT_PLAN_TABLE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans", "gfx950.json"))


class _Plans:
    """Launch plans (tile / split / kernel family per convolution of a forward shape).

    Results must not depend on which process runs them: by default a shape's plan is a pure function of the model and the
    shape -- the entry of the COMMITTED per-arch table ``plans/gfx950.json`` (measured once with
    ``tools/make_plan_table.py`` for the BASELINE shapes) or, for shapes the table does not hold, the library's
    deterministic heuristic.  Timing candidates in the process (``dt_unet_autotune``: its picks between near-equal
    candidates differ from run to run, which changes results at fp32-rounding level) is opt-in: ``DT_AUTOTUNE=1``, or
    ``tune=True`` on a call; ``DT_TUNE_CACHE=file`` then records what was measured so that later processes replay it.
    """
    _table = None
    _lock = None

    @classmethod
    def table(cls):
        if cls._table is None:
            import json
            try:
                with open(PLAN_TABLE) as f:
                    cls._table = json.load(f).get("plans", {})
            except (OSError, ValueError):
                cls._table = {}
        return cls._table

    @staticmethod
    def cache_path():
        return os.environ.get("DT_TUNE_CACHE") or None

    @staticmethod
    def cache_load():
        import json
        p = _Plans.cache_path()
        if not p:
            return None
        try:
            with open(p) as f:
                return json.load(f)
        except (OSError, ValueError):
            return {}

    @staticmethod
    def cache_store(key, plan):
        """Read-modify-replace under an exclusive lock file (several ranks may tune at once)."""
        import fcntl
        import json
        p = _Plans.cache_path()
        with open(p + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            cache = _Plans.cache_load() or {}
            cache[key] = plan
            tmp = f"{p}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump(cache, f)
            os.replace(tmp, p)

    @staticmethod
    def digest(plan):
        import hashlib
        import json
        return hashlib.sha1(json.dumps(plan, sort_keys=True).encode()).hexdigest()[:10]


import weakref

_HANDLES = weakref.WeakKeyDictionary()      # nn.Module -> (weights key, UNetHandle)
_TRAIN_WARNED = weakref.WeakSet()
_FUSED_DEFAULT = True


def set_fused_default(on):
    """Whether small models (padded dims <= 32 / 64) at 16x16 take the fused whole-forward kernel (dt_fused.hip): applies to
    every handle cached on a module and to handles created later.  ``DT_NO_FUSED=1`` in the environment is the same switch
    at library level (read when a handle is created)."""
    global _FUSED_DEFAULT
    _FUSED_DEFAULT = bool(on)
    for _, h in list(_HANDLES.values()):
        h.set_fused(_FUSED_DEFAULT)


class UNetHandle:
    """Owns one ``dt_unet`` (packed weights in HBM) built from a module's state_dict."""

    def __init__(self, state_dict, device):
        self.lib = _hip.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise HipLibraryError("models must live on a CUDA(HIP) device; there is no CPU fallback")
        sd = {k: v.detach().to(device=self.device, dtype=torch.float32).contiguous()
              for k, v in state_dict.items() if v.dtype.is_floating_point}
        self.channels = sd["final.weight"].shape[0]
        self.temb_dim = sd["time_mlp.1.weight"].shape[0]
        self.dims = [sd[f"{n}.conv1.weight"].shape[0] for n in ("enc1", "enc2", "enc3", "enc4")]
        desc = _hip.UNetDesc(self.channels, (c_int32 * 4)(*self.dims), self.temb_dim)
        bt = (c_void_p * (_hip.N_BLOCKS * _hip.BT_COUNT))()
        for j, name in enumerate(BLOCK_NAMES):
            for i, key in enumerate(_BLOCK_KEYS):
                t = sd.get(f"{name}.{key}")
                bt[j * _hip.BT_COUNT + i] = t.data_ptr() if t is not None else None
        freqs = sinusoid_frequencies(self.temb_dim).to(self.device)
        gt = (c_void_p * _hip.GT_COUNT)(*[sd[k].data_ptr() for k in _GLOBAL_KEYS], freqs.data_ptr())
        h = c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.dt_unet_create(ctypes.byref(desc), bt, gt, stream_ptr(), ctypes.byref(h)), "dt_unet_create")
            torch.cuda.current_stream().synchronize()   # sd / freqs may be freed after this
        self.h = h
        self.tb_stride = self.lib.dt_unet_time_bias_stride(self.h)
        mode = os.environ.get("DT_PRECISION", "auto")
        check(self.lib.dt_unet_set_precision(self.h, {"fp32": 0, "split-bf16": 1, "auto": 2}[mode]), "dt_unet_set_precision")
        self._mode = mode
        self._shared_enc1 = os.environ.get("DT_NO_SHARED_ENC1") is None      # read by dt_unet_create
        self._ws = {}
        self._consts = {}
        self._plans = {}              # (rows, H, W, images, single-pass images) -> plan id ("table:<sha1>", "tuned:<sha1>", "heuristic", "pinned")
        if not _FUSED_DEFAULT:
            self.set_fused(False)

    def __del__(self):
        try:
            h, self.h = getattr(self, "h", None), None
            if h and not sys.is_finalizing():    # at interpreter exit the HIP runtime may already be gone
                self.lib.dt_unet_destroy(h)
        except Exception:
            pass

    # ------------------------------------------------------------------ caching per nn.Module
    @staticmethod
    def for_module(module):
        """Handle cached on the module; rebuilt when any weight tensor was replaced or mutated.

        The HIP path is inference-only (BatchNorm running statistics, no dropout, no autograd): a module left in
        train mode -- the reference's samplers never call ``eval()`` themselves (utils/diffusion.py:102-212) --
        would give different numbers in the reference, so that is reported once per module."""
        tensors = list(module.parameters()) + list(module.buffers())
        _require_cuda(tensors[0], "model parameters")
        # (the cache lives beside the module, not in its __dict__: a module that has run on the HIP path can still be
        # deep-copied, pickled and torch.save()d like any other nn.Module)
        if module.training and module not in _TRAIN_WARNED:
            import warnings
            _TRAIN_WARNED.add(module)
            warnings.warn("distillation_trajectories_amd: the model is in train mode, but the HIP path is inference-only "
                          "(BatchNorm running statistics, dropout off); call model.eval() as the reference's callers do",
                          RuntimeWarning, stacklevel=3)
        key = tuple((v.data_ptr(), v._version) for v in tensors)
        cached = _HANDLES.get(module)
        if cached is None or cached[0] != key:
            cached = (key, UNetHandle(module.state_dict(), tensors[0].device))
            _HANDLES[module] = cached
        return cached[1]

    # ------------------------------------------------------------------ primitives
    def workspace(self, batch_total, H, W):
        """Scratch of one forward shape, one buffer PER STREAM: two streams may run the same handle concurrently
        (activations live in the workspace, so a shared buffer would be a silent race)."""
        key = (batch_total, H, W, torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            n = self.lib.dt_unet_workspace_bytes(self.h, batch_total, H, W)
            if n == 0:
                raise HipLibraryError(f"unsupported shape batch={batch_total} H={H} W={W} (H, W must be multiples of 16)")
            if len(self._ws) > 8:
                torch.cuda.synchronize(self.device)     # a buffer may still be in use on another stream
                self._ws.clear()
            ws = self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        return ws

    def set_precision(self, mode):
        """PREC_FP32 (exact fp32 MFMA), PREC_SPLIT_BF16 (3-plane bf16 split, 6 products) or PREC_AUTO."""
        check(self.lib.dt_unet_set_precision(self.h, int(mode)), "dt_unet_set_precision")
        self._mode = {0: "fp32", 1: "split-bf16", 2: "auto"}[int(mode)]
        self._plans.clear()               # the library drops its tuned shapes too: choices are per arithmetic mode

    def set_fused(self, on):
        """Small models at 16x16: whole forward / whole sampler loop as one launch (dt_unet_set_fused); no-op for models
        that do not qualify."""
        check(self.lib.dt_unet_set_fused(self.h, int(bool(on))), "dt_unet_set_fused")

    def fused_active(self, H, W):
        return bool(self.lib.dt_unet_fused_active(self.h, H, W))

    def set_head_fusion(self, on):
        """Test hook (dt_unet_set_head_fusion): off = separate head launch, dec1's output is materialised."""
        check(self.lib.dt_unet_set_head_fusion(self.h, int(bool(on))), "dt_unet_set_head_fusion")

    @property
    def _tuned(self):
        """{(rows, H, W)} of the shapes that run a table / measured / pinned plan (not the bare heuristic)."""
        return {k[:3] for k, v in self._plans.items() if v != "heuristic"}

    # version of the launch-plan vocabulary (tile sizes, launch kinds, split semantics of dt_unet_set_conv_choice): the committed
    # table and recorded caches stay valid across ABI revisions that only ADD entry points (ABI 4 added the fused-path switches)
    PLAN_VERSION = 3

    def plan_key(self, rows, H, W, imgs, single):
        return (f"abi{self.PLAN_VERSION}|gfx950|prec={self._mode}|enc1={'shared' if self._shared_enc1 else 'per-pass'}|C{self.channels}"
                f"|D{self.temb_dim}|{','.join(map(str, self.dims))}|{rows}x{H}x{W}|{imgs}/{single}")

    def _read_plan(self, rows, H, W):
        plan = []
        for j in range(8):
            for slot in range(3):
                bm, bn, sp, pr, tu = c_int(), c_int(), c_int(), c_int(), c_int()
                check(self.lib.dt_unet_conv_choice(self.h, rows, H, W, j, slot, ctypes.byref(bm), ctypes.byref(bn),
                                                   ctypes.byref(sp), ctypes.byref(pr), ctypes.byref(tu)), "dt_unet_conv_choice")
                if bm.value:
                    plan.append([j, slot, bm.value, bn.value, sp.value, pr.value & 7, 1 if pr.value & 8 else 0])
        return plan

    def _apply_plan(self, rows, H, W, plan):
        """Pin every launch of a recorded plan; an entry this build no longer admits (or that contradicts the handle's
        arithmetic mode) is skipped, which leaves that slot on the heuristic."""
        ok = True
        for block, slot, bm, bn, sp, prec, fuse in plan:
            if (self._mode == "fp32") != (prec == 0) and self._mode != "auto":
                ok = False
                continue
            if self.lib.dt_unet_set_conv_choice(self.h, rows, H, W, block, slot, bm, bn, sp, prec, fuse) != 0:
                ok = False
        return ok

    def ensure_plan(self, rows, H, W, imgs, single, tune=None, warm=None):
        """Settle the launch plan of a forward shape once (see ``_Plans``).  ``warm`` runs one forward of the shape
        (real activations in the workspace) before candidates are timed.  Returns the plan id."""
        key = (rows, H, W, imgs, single)
        have = self._plans.get(key)
        if have is not None and not (tune is True and not have.startswith(("tuned", "pinned"))):
            return have
        check(self.lib.dt_unet_declare_shape(self.h, rows, H, W, imgs, single), "dt_unet_declare_shape")
        if tune is False:
            self._plans[key] = "heuristic"
            return "heuristic"
        pkey = self.plan_key(rows, H, W, imgs, single)
        if tune is not True:
            for source, entries in (("table", _Plans.table()), ("cache", _Plans.cache_load() or {})):
                plan = entries.get(pkey)
                if plan and self._apply_plan(rows, H, W, plan):
                    self._plans[key] = f"{source}:{_Plans.digest(plan)}"
                    return self._plans[key]
        measure = tune is True or (os.environ.get("DT_AUTOTUNE", "0") == "1" and rows * H * W >= AUTOTUNE_MIN_ROWS)
        if not measure:
            self._plans[key] = "heuristic"
            return "heuristic"
        if warm is not None:
            warm()
        ws = self.workspace(rows, H, W)
        with torch.cuda.device(self.device):
            check(self.lib.dt_unet_autotune(self.h, rows, H, W, ptr(ws), c_size_t(ws.numel()), stream_ptr()), "dt_unet_autotune")
        plan = self._read_plan(rows, H, W)
        self._plans[key] = f"tuned:{_Plans.digest(plan)}"
        if _Plans.cache_path():
            _Plans.cache_store(pkey, plan)
        return self._plans[key]

    def plan_ids(self):
        """{"rowsxHxW imgs/single": plan id} of every shape this handle has run (bench.py prints it)."""
        return {f"{k[0]}x{k[1]}x{k[2]} {k[3]}/{k[4]}": v for k, v in sorted(self._plans.items())}

    def set_conv_choice(self, batch_total, H, W, block, slot, bm, bn, splits=1, prec=1, fuse=0):
        """Pin one convolution's launch choice for this forward shape (dt_unet_set_conv_choice).  A launch pinned by hand asks
        for the layered kernels, so the handle leaves the fused small-model path (``set_fused(True)`` returns to it)."""
        check(self.lib.dt_unet_set_conv_choice(self.h, batch_total, H, W, block, slot, bm, bn, splits, prec, fuse),
              "dt_unet_set_conv_choice")
        self.set_fused(False)
        for k in [k for k in self._plans if k[:3] == (batch_total, H, W)]:
            self._plans[k] = "pinned"
        self._pinned = getattr(self, "_pinned", set()) | {(batch_total, H, W)}

    def conv_choices(self, batch_total, H, W):
        """[(block, slot, bm, bn, splits, tuned)] for reporting."""
        out = []
        for j in range(8):
            for slot in range(3):
                bm, bn, sp, pr, tu = c_int(), c_int(), c_int(), c_int(), c_int()
                check(self.lib.dt_unet_conv_choice(self.h, batch_total, H, W, j, slot, ctypes.byref(bm), ctypes.byref(bn),
                                                   ctypes.byref(sp), ctypes.byref(pr), ctypes.byref(tu)), "dt_unet_conv_choice")
                if bm.value:
                    out.append((BLOCK_NAMES[j], ("skip", "conv1", "conv2")[slot], bm.value, bn.value, sp.value,
                                ("fp32", "split-bf16", "-", "split-bf16-strip", "split-bf16-strip32", "split-bf16-stripk", "-", "-")[pr.value & 7]
                                + ("+skip" if pr.value & 8 else ""), bool(tu.value)))
        return out

    def time_bias(self, t_values, cond_modes):
        """[rows, tb_stride] table for rows (t_values[i], cond_modes[i]); cond mode in {NONE, ZERO, ONE}.

        The three small index tensors are uploaded once per distinct row list and kept with the handle: a pageable
        host-to-device copy blocks the host until the stream has drained, which would stop a caller from queueing one
        sampler loop behind another."""
        rows = len(t_values)
        key = ("tb_in", tuple(int(v) for v in t_values), tuple(int(m) for m in cond_modes))
        cached = self._consts.get(key)
        if cached is None:
            if len(self._consts) > 16:
                self._consts.clear()
            t = torch.tensor(list(t_values), dtype=torch.int32).to(self.device)
            cond = torch.tensor([1.0 if m == COND_ONE else 0.0 for m in cond_modes], dtype=torch.float32).to(self.device)
            present = torch.tensor([0 if m == COND_NONE else 1 for m in cond_modes], dtype=torch.uint8).to(self.device)
            cached = self._consts[key] = (t, cond, present)
        return self.time_bias_general(cached[0], cached[1], cached[2], rows)

    def time_bias_general(self, t_i32, cond_f32, present_u8, rows):
        out = torch.empty(rows, self.tb_stride, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.dt_unet_time_bias(self.h, ptr(t_i32), ptr(cond_f32), ptr(present_u8), rows, ptr(out),
                                             stream_ptr()), "dt_unet_time_bias")
        return out

    def forward(self, x, tb, n_pass, tb_div, tune=None):
        """eps[n_pass*B,C,H,W] for x[B,C,H,W]; tb rows as documented in dt_hip.h."""
        _require_cuda(x, "x")
        x = x.contiguous().float()
        B, C, H, W = x.shape
        if C != self.channels:
            raise HipLibraryError(f"x has {C} channels, the model expects {self.channels}")
        eps = torch.empty(n_pass * B, C, H, W, dtype=torch.float32, device=self.device)
        ws = self.workspace(n_pass * B, H, W)

        def run():
            with torch.cuda.device(self.device):
                check(self.lib.dt_unet_forward(self.h, ptr(x), B, n_pass, H, W, ptr(tb), tb_div, ptr(eps), ptr(ws),
                                               c_size_t(ws.numel()), stream_ptr()), "dt_unet_forward")
        self._settle(n_pass * B, H, W, B, 0, tune, run)
        run()
        return eps

    def _settle(self, rows, H, W, imgs, single, tune, warm):
        if (rows, H, W) in getattr(self, "_pinned", ()):      # launches pinned by hand (tests, tools): leave them alone
            self._plans.setdefault((rows, H, W, imgs, single), "pinned")
            return
        self.ensure_plan(rows, H, W, imgs, single, tune, warm)

    def forward_mixed(self, x, tb, b_single, tb_div, tune=None):
        """eps[2B - b_single, C, H, W] of a mixed batch (dt_unet_forward_mixed): images [0, b_single) take one pass,
        the others two; rows [pass 0 of all images | pass 1 of images b_single..]; row r uses tb[r // tb_div]."""
        _require_cuda(x, "x")
        x = x.contiguous().float()
        B, C, H, W = x.shape
        rows = 2 * B - b_single
        eps = torch.empty(rows, C, H, W, dtype=torch.float32, device=self.device)
        ws = self.workspace(rows, H, W)

        def run():
            with torch.cuda.device(self.device):
                check(self.lib.dt_unet_forward_mixed(self.h, ptr(x), B, b_single, H, W, ptr(tb), tb_div, ptr(eps), ptr(ws),
                                                     c_size_t(ws.numel()), stream_ptr()), "dt_unet_forward_mixed")
        self._settle(rows, H, W, B, b_single, tune, run)
        run()
        return eps

    def sample_mixed(self, rule, traj, H, W, tb, tb_div, b_single, coef, has_noise, z, z_row, z_shift, w):
        """len(coef) reverse steps in place on traj[(n_steps+1), B, E] of a mixed batch (dt_sample_trajectory_mixed)."""
        _require_cuda(traj, "traj")
        n_steps = len(coef)
        B = traj.shape[1]
        rows = 2 * B - b_single
        assert traj.shape[0] == n_steps + 1 and traj.is_contiguous() and traj.dtype == torch.float32
        coef_c = (c_float * (4 * n_steps))(*[float(v) for row in coef for v in (list(row) + [0.0] * 4)[:4]])
        noise_c = (c_int32 * n_steps)(*[int(bool(v)) for v in has_noise])
        shift_c = (c_int64 * n_steps)(*[int(v) for v in (z_shift if z_shift is not None else [0] * n_steps)])
        ws = self.workspace(rows, H, W)
        if n_steps and (rows, H, W, B, b_single) not in self._plans:
            per_step = rows // tb_div
            self.forward_mixed(traj[0].reshape(B, self.channels, H, W), tb[:per_step].contiguous(), b_single, tb_div)
        with torch.cuda.device(self.device):
            check(self.lib.dt_sample_trajectory_mixed(self.h, rule, B, b_single, H, W, n_steps, ptr(tb), tb_div, coef_c, noise_c,
                                                      ptr(z), ptr(z_row), shift_c, ptr(w), ptr(traj), ptr(ws),
                                                      c_size_t(ws.numel()), stream_ptr()), "dt_sample_trajectory_mixed")
        return traj

    def debug_activation(self, batch_total, H, W, which):
        """NHWC view [Bt,h,w,cp] of block ``which``'s output inside the workspace of the last forward."""
        off, cp, oh, ow = c_size_t(), c_int(), c_int(), c_int()
        check(self.lib.dt_unet_debug_activation(self.h, batch_total, H, W, which, ctypes.byref(off), ctypes.byref(cp),
                                                ctypes.byref(oh), ctypes.byref(ow)), "dt_unet_debug_activation")
        ws = self.workspace(batch_total, H, W).view(torch.float32)
        n = batch_total * oh.value * ow.value * cp.value
        return ws[off.value: off.value + n].view(batch_total, oh.value, ow.value, cp.value)

    def sample(self, rule, traj, H, W, tb, n_pass, coef, has_noise, z=None, z_row=None, z_shift=None, w=None,
               w_scalar=1.0):
        """Run len(coef) reverse steps in place on traj[(n_steps+1), B, E] (slot 0 = x_T)."""
        _require_cuda(traj, "traj")
        n_steps = len(coef)
        B = traj.shape[1]
        assert traj.shape[0] == n_steps + 1 and traj.is_contiguous() and traj.dtype == torch.float32
        coef_c = (c_float * (4 * n_steps))(*[float(v) for row in coef for v in (list(row) + [0.0] * 4)[:4]])
        noise_c = (c_int32 * n_steps)(*[int(bool(v)) for v in has_noise])
        shift_c = (c_int64 * n_steps)(*[int(v) for v in (z_shift if z_shift is not None else [0] * n_steps)])
        ws = self.workspace(n_pass * B, H, W)
        if n_steps and (n_pass * B, H, W, B, 0) not in self._plans:
            self.forward(traj[0].reshape(B, self.channels, H, W), tb[:n_pass].contiguous(), n_pass, B)
        with torch.cuda.device(self.device):
            check(self.lib.dt_sample_trajectory(self.h, rule, B, n_pass, H, W, n_steps, ptr(tb), coef_c, noise_c,
                                                ptr(z), ptr(z_row), shift_c, ptr(w), c_float(w_scalar), ptr(traj),
                                                None, ptr(ws), c_size_t(ws.numel()), stream_ptr()),
                  "dt_sample_trajectory")
        return traj


def cfg_update(rule, x, eps_u, eps_c, z, coef, has_noise, w=None, w_scalar=1.0, z_row=None):
    """One fused CFG-mix + update step (dt_cfg_update); returns the new x."""
    lib = _hip.load()
    _require_cuda(x, "x")
    B, E = x.shape[0], x[0].numel()
    out = torch.empty_like(x)
    coef_c = (c_float * 4)(*(list(coef) + [0.0] * 4)[:4])
    with torch.cuda.device(x.device):
        check(lib.dt_cfg_update(rule, ptr(x), ptr(eps_u), ptr(eps_c), ptr(z), ptr(z_row), coef_c, int(bool(has_noise)),
                                ptr(w), c_float(w_scalar), ptr(out), B, E, stream_ptr()), "dt_cfg_update")
    return out


# ---------------------------------------------------------------------- module-level forward
def unet_forward_module(module, x, t, cond=None):
    """General ``model(x, t, cond)`` (reference models.py:159-224): per-row t / cond with broadcasting."""
    _require_cuda(x, "x")
    h = UNetHandle.for_module(module)
    B = x.shape[0]
    t = t.reshape(t.shape[0], -1)[:, 0] if t.dim() > 1 else t
    rows = t.shape[0]
    c = None
    if cond is not None:
        c = cond.reshape(cond.shape[0], -1)[:, 0].float()
        rows = max(rows, c.shape[0])
    if rows not in (1, B):
        raise RuntimeError(f"time/condition rows ({rows}) cannot broadcast onto batch {B}")
    t_i = t.to(device=h.device, dtype=torch.int32).expand(rows).contiguous()
    c_f = c.to(h.device).expand(rows).contiguous() if c is not None else None
    tb = h.time_bias_general(t_i, c_f, None, rows)
    return h.forward(x, tb, 1, B if rows == 1 else 1)


# ---------------------------------------------------------------------- metrics
def device_metric_sums(X, Y):
    """float64 [B, n_max, 4] sums of dt_traj_metrics for trajectories X[nT,B,E], Y[nS,B,E] on device."""
    lib = _hip.load()
    _require_cuda(X, "teacher trajectory"); _require_cuda(Y, "student trajectory")
    nT, B, E = X.shape
    nS = Y.shape[0]
    out = torch.empty(B, max(nT, nS), 4, dtype=torch.float64, device=X.device)
    with torch.cuda.device(X.device):
        check(lib.dt_traj_metrics(ptr(X), ptr(Y), nT, nS, B, E, ptr(out), stream_ptr()), "dt_traj_metrics")
    return out


def device_wasserstein(X, Y, index=None, index_row=None):
    """float64 [B, n] per-step W1 over the first n = min(len) states; index int32 [tables, n, n_idx] or
    None (all coordinates); index_row int32 [B] picks each pair's table (None: table 0)."""
    lib = _hip.load()
    n = min(X.shape[0], Y.shape[0])
    _, B, E = X.shape
    out = torch.empty(B, n, dtype=torch.float64, device=X.device)
    n_idx = 0 if index is None else index.shape[-1]
    with torch.cuda.device(X.device):
        check(lib.dt_traj_wasserstein(ptr(X), ptr(Y), n, B, E, ptr(index), ptr(index_row), n_idx, ptr(out), stream_ptr()),
              "dt_traj_wasserstein")
    return out


def device_pair_metrics(X, Y, index=None, index_row=None):
    """(sums float64 [B, n_max, 4], w1 float64 [B, min(n)]) of ``device_metric_sums`` and ``device_wasserstein`` -- in ONE
    launch (dt_traj_pair_metrics: each state read from HBM once, register / cross-lane sort) when the trajectories have equal
    lengths and the Wasserstein term uses all E <= 4096 coordinates, otherwise through the two separate kernels."""
    lib = _hip.load()
    _require_cuda(X, "teacher trajectory"); _require_cuda(Y, "student trajectory")
    n, B, E = X.shape
    if index is not None or Y.shape[0] != n or E > 4096 or E % 4 or B > 65535:
        return device_metric_sums(X, Y), device_wasserstein(X, Y, index, index_row)
    sums = torch.empty(B, n, 4, dtype=torch.float64, device=X.device)
    w1 = torch.empty(B, n, dtype=torch.float64, device=X.device)
    with torch.cuda.device(X.device):
        check(lib.dt_traj_pair_metrics(ptr(X), ptr(Y), n, B, E, ptr(sums), ptr(w1), stream_ptr()), "dt_traj_pair_metrics")
    return sums, w1


def device_pair_stats(X, Y):
    """float64 [B, n, 5] = {sum (x-y)^2, sum |x-y|, sum xy, sum x^2, sum y^2} for X, Y [n, B, E] (dt_pair_stats)."""
    lib = _hip.load()
    _require_cuda(X, "X"); _require_cuda(Y, "Y")
    n, B, E = X.shape
    out = torch.empty(B, n, 5, dtype=torch.float64, device=X.device)
    with torch.cuda.device(X.device):
        check(lib.dt_pair_stats(ptr(X), ptr(Y), n, B, E, ptr(out), stream_ptr()), "dt_pair_stats")
    return out


def device_sample_mean(traj):
    """fp32 [n, E]: mean over the B samples of traj [n, B, E] (dt_traj_sample_mean)."""
    lib = _hip.load()
    _require_cuda(traj, "traj")
    n, B, E = traj.shape
    out = torch.empty(n, E, dtype=torch.float32, device=traj.device)
    with torch.cuda.device(traj.device):
        check(lib.dt_traj_sample_mean(ptr(traj), n, B, E, ptr(out), stream_ptr()), "dt_traj_sample_mean")
    return out


def resize_bilinear(images, size):
    """``torch.nn.functional.interpolate(images, size=size, mode='bilinear', align_corners=True)`` for an NCHW fp32 tensor,
    on the device (dt_resize_bilinear); a host tensor is uploaded and the result stays on the device."""
    lib = _hip.load()
    if not images.is_cuda:
        if not torch.cuda.is_available():
            raise HipLibraryError("resize_bilinear needs a HIP device; there is no CPU fallback")
        images = images.to(torch.device("cuda", torch.cuda.current_device()))
    x = images.detach().contiguous().float()
    N, C, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.dt_resize_bilinear(ptr(x), ptr(out), N * C, h, w, H, W, stream_ptr()), "dt_resize_bilinear")
    return out


def device_resampled_distance(longer, shorter):
    lib = _hip.load()
    nl, B, E = longer.shape
    ns = shorter.shape[0]
    out = torch.empty(B, ns, dtype=torch.float64, device=longer.device)
    with torch.cuda.device(longer.device):
        check(lib.dt_traj_resampled_distance(ptr(longer), ptr(shorter), nl, ns, B, E, ptr(out), stream_ptr()),
              "dt_traj_resampled_distance")
    return out


def metrics_from_sums(sums, w1, nT, nS, pixels, elems, resampled=None):
    """The 25-key dict of trajectory_metrics.py:12-325 from one pair's device reductions.

    sums: float64 [n_max,4] (dt_traj_metrics layout), w1: float64 [min(nT,nS)], resampled: float64
    [min] distances of the interp1d path (unequal lengths) or None.  The host part reproduces the
    reference's scalar arithmetic: fp32 where torch returned fp32 scalars, python floats elsewhere.
    """
    f32 = np.float32
    n = min(nT, nS)
    s32 = sums.astype(f32)                       # torch reduces in fp32 before .item()
    D = [float(np.sqrt(s32[i, 0])) for i in range(n)]
    Vt = [float(np.sqrt(s32[i, 1])) for i in range(1, nT)]
    Vs = [float(np.sqrt(s32[i, 2])) for i in range(1, nS)]
    m = {}
    # endpoint / final-image terms use each trajectory's OWN last state (row 0, slot 3)
    m["endpoint_distance"] = float(np.sqrt(s32[0, 3]))
    mse = float(s32[0, 3] / f32(elems))
    m["mse"] = mse
    acc = 0.0
    for i in range(n):
        acc += float(s32[i, 0] / f32(elems))
    m["trajectory_mse"] = np.log1p(1.0 - (acc / n) * 1000)
    m["point_by_point_similarity"] = np.exp(-5.0 * (np.mean(D) if D else float("inf")))
    m["log_mse_similarity"] = max(0, 1.0 - np.log1p(mse * 5000) / np.log1p(5000))
    tl = sl = 0
    for i in range(1, n):
        tl += Vt[i - 1] / pixels
        sl += Vs[i - 1] / pixels
    tl /= (n - 1)
    sl /= (n - 1)
    m["teacher_path_length"], m["student_path_length"] = tl, sl
    m["path_length_similarity"] = np.log1p(min(tl, sl) / max(tl, sl) if max(tl, sl) > 0 else 1.0)
    te = float(np.sqrt(s32[0, 1])) / tl if tl > 0 else 0
    se = float(np.sqrt(s32[0, 2])) / sl if sl > 0 else 0
    m["teacher_efficiency"], m["student_efficiency"] = te, se
    m["efficiency_similarity"] = np.log1p(min(te, se) / max(te, se) if max(te, se) > 0 else 1.0)
    m["teacher_velocities"], m["student_velocities"] = Vt, Vs
    vsim = [(min(a, b) / max(a, b) if max(a, b) > 0 else 1.0) for a, b in zip(Vt, Vs)]
    m["velocity_similarities"] = vsim
    m["mean_velocity_similarity"] = np.mean(vsim) if vsim else 0.0
    m["position_differences"] = list(D)
    m["mean_position_difference"] = np.mean(D) if D else 0.0
    m["max_position_difference"] = np.max(D) if D else 0.0
    cos, wcos = [], []
    for i in range(n - 1):
        vt, vs = f32(np.sqrt(s32[i + 1, 1])), f32(np.sqrt(s32[i + 1, 2]))
        if vt > 0 and vs > 0:
            c = float(s32[i + 1, 3] / (vt * vs))
            cos.append(c)
            wcos.append(c * ((float(vt) + float(vs)) / 2))
    m["directional_consistency"] = cos
    m["mean_directional_consistency"] = np.mean(cos) if cos else 0.0
    if wcos:
        total_w = sum((Vt[i] + Vs[i]) / 2 for i in range(min(len(Vt), len(Vs))))
        m["weighted_directional_consistency"] = (sum(wcos) / total_w if total_w > 0 else 0) ** 2
    else:
        m["weighted_directional_consistency"] = 0.0
    if resampled is None:
        pd = [f32(d) for d in D]                 # numpy norms of fp32 arrays stay fp32
    else:
        pd = [np.float64(d) for d in resampled]
    m["path_alignment"] = np.exp(-10.0 * np.sum(pd) / len(pd))
    W = [np.float64(v) for v in w1]
    m["wasserstein_distances"] = W
    m["mean_wasserstein"] = np.mean(W)
    m["distribution_similarity"] = np.log1p(np.exp(-m["mean_wasserstein"]))
    return m


SCALAR_KEYS = ("endpoint_distance", "mse", "trajectory_mse", "point_by_point_similarity", "log_mse_similarity",
               "teacher_path_length", "student_path_length", "path_length_similarity", "teacher_efficiency",
               "student_efficiency", "efficiency_similarity", "mean_velocity_similarity", "mean_position_difference",
               "max_position_difference", "mean_directional_consistency", "weighted_directional_consistency",
               "path_alignment", "mean_wasserstein", "distribution_similarity")


def batch_scalar_metrics(sums, w1, pixels, elems):
    """Vectorised form of ``metrics_from_sums`` for B equal-length pairs: {key: float64 array [B]} for the 19
    scalar metrics (the per-step lists stay on the caller's side as sums / w1).

    sums float64 [B,n,4], w1 float64 [B,n].  Follows the same dtype path (fp32 where torch returned fp32
    scalars, float64 python arithmetic elsewhere, sequential accumulation where the reference loops)."""
    f32 = np.float32
    B, n, _ = sums.shape
    with np.errstate(invalid="ignore", divide="ignore"):
        s32 = sums.astype(f32)
        D = np.sqrt(s32[:, :, 0]).astype(np.float64)              # [B,n]
        Vt = np.sqrt(s32[:, 1:, 1]).astype(np.float64)            # [B,n-1]
        Vs = np.sqrt(s32[:, 1:, 2]).astype(np.float64)
        out = {}
        out["endpoint_distance"] = np.sqrt(s32[:, 0, 3]).astype(np.float64)
        mse = (s32[:, 0, 3] / f32(elems)).astype(np.float64)
        out["mse"] = mse
        step_mse = (s32[:, :, 0] / f32(elems)).astype(np.float64)
        acc = np.zeros(B)
        for i in range(n):
            acc = acc + step_mse[:, i]
        out["trajectory_mse"] = np.log1p(1.0 - (acc / n) * 1000)
        meanD = D.mean(axis=1)
        out["point_by_point_similarity"] = np.exp(-5.0 * meanD)
        lms = 1.0 - np.log1p(mse * 5000) / np.log1p(5000)
        out["log_mse_similarity"] = np.where(lms > 0, lms, 0.0)      # python max(0, x): NaN -> 0
        tl, sl = np.zeros(B), np.zeros(B)
        for i in range(n - 1):
            tl = tl + Vt[:, i] / pixels
            sl = sl + Vs[:, i] / pixels
        tl, sl = tl / (n - 1), sl / (n - 1)
        out["teacher_path_length"], out["student_path_length"] = tl, sl
        hi = np.maximum(tl, sl)
        out["path_length_similarity"] = np.log1p(np.where(hi > 0, np.minimum(tl, sl) / np.where(hi > 0, hi, 1), 1.0))
        te = np.where(tl > 0, np.sqrt(s32[:, 0, 1]).astype(np.float64) / np.where(tl > 0, tl, 1), 0.0)
        se = np.where(sl > 0, np.sqrt(s32[:, 0, 2]).astype(np.float64) / np.where(sl > 0, sl, 1), 0.0)
        out["teacher_efficiency"], out["student_efficiency"] = te, se
        hi = np.maximum(te, se)
        out["efficiency_similarity"] = np.log1p(np.where(hi > 0, np.minimum(te, se) / np.where(hi > 0, hi, 1), 1.0))
        vhi = np.maximum(Vt, Vs)
        vsim = np.where(vhi > 0, np.minimum(Vt, Vs) / np.where(vhi > 0, vhi, 1), 1.0)
        out["mean_velocity_similarity"] = vsim.mean(axis=1) if n > 1 else np.zeros(B)
        out["mean_position_difference"] = meanD
        out["max_position_difference"] = D.max(axis=1)
        vt32, vs32 = np.sqrt(s32[:, 1:, 1]), np.sqrt(s32[:, 1:, 2])
        ok = (vt32 > 0) & (vs32 > 0)
        cos = np.where(ok, (s32[:, 1:, 3] / np.where(ok, vt32 * vs32, f32(1))).astype(np.float64), 0.0)
        cnt = ok.sum(axis=1)
        csum = np.zeros(B)
        wsum = np.zeros(B)
        for i in range(n - 1):
            csum = csum + cos[:, i]
            wsum = wsum + cos[:, i] * ((Vt[:, i] + Vs[:, i]) / 2) * ok[:, i]
        out["mean_directional_consistency"] = np.where(cnt > 0, csum / np.where(cnt > 0, cnt, 1), 0.0)
        tw = np.zeros(B)
        for i in range(n - 1):
            tw = tw + (Vt[:, i] + Vs[:, i]) / 2
        out["weighted_directional_consistency"] = np.where(cnt > 0, np.where(tw > 0, wsum / np.where(tw > 0, tw, 1), 0.0) ** 2, 0.0)
        pd32 = np.sqrt(s32[:, :, 0])                                # numpy norms of fp32 arrays stay fp32
        out["path_alignment"] = np.exp(-10.0 * pd32.sum(axis=1, dtype=f32) / n).astype(np.float64)
        mw = w1.mean(axis=1)
        out["mean_wasserstein"] = mw
        out["distribution_similarity"] = np.log1p(np.exp(-mw))
    return out
