"""Training-step harness around the rasterizer (reference: train.py:120-306 and
scene/gaussian_model.py:187-223): parameters, gradients and Adam moments of all eight optimizer groups live in FOUR flat
float32 buffers with one layout, so that

  * the rasterizer backward writes its gradients straight into the gradient buffer (gradient sink),
  * the multi-GPU exchange is ONE all-reduce of that buffer (gsr_dist.FlatGrads),
  * the optimizer is ONE fused Adam launch over the buffers (gsr_adam_step, include/gsr_hip.h).

The reference keeps `_features_dc` (P,1,3) and `_features_rest` (P,15,3) as separate parameters and concatenates them
for every render (scene/gaussian_model.py:135-139); here `shs` (P,16,3) is stored concatenated and the two learning
rates (feature_lr and feature_lr / 20, scene/gaussian_model.py:198-199) are applied by position inside each 48-float row.
"""
import ctypes
import math

import torch

import _gsr
from _gsr import AdamSegment, check, lib, stream_ptr
from gsr_dist import FlatGrads

# arguments/__init__.py:82-102 of the reference (OptimizationParams defaults)
DEFAULT_LRS = dict(position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30_000,
                   feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, refl_lr=0.006, envmap_cubemap_lr=0.05)


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """utils/general_utils.py:29-62 of the reference: log-linear interpolation with an optional warm-up delay."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        else:
            delay_rate = 1.0
        t = min(max(step / max_steps, 0.0), 1.0)
        log_lerp = math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
        return delay_rate * log_lerp
    return helper


class FlatParams:
    """Leaf tensors that are views of one flat float32 buffer (order of `tensors` = order in the buffer; every slice
    starts on a 16-byte boundary so the fused Adam kernel can use float4 accesses across group borders)."""

    def __init__(self, tensors, device, shapes=None, shards=1):
        """tensors: dict name -> initial value; or None with `shapes` (dict name -> shape) for an uninitialised store.
        shards > 1: the buffer length is rounded up to a multiple of 4 * shards floats so that it splits into `shards` equal,
        16-byte-aligned chunks (reduce-scatter / all-gather over ranks, gsr_dist.ShardedStep); the padding rides with the last group."""
        if tensors is not None:
            shapes = {k: tuple(v.shape) for k, v in tensors.items()}
        self.names = list(shapes.keys())
        self.shapes = {k: tuple(v) for k, v in shapes.items()}
        self.slices = {}
        off = 0
        for k in self.names:
            n = 1
            for d in self.shapes[k]:
                n *= int(d)
            self.slices[k] = (off, off + n)
            off += (n + 3) // 4 * 4
        q = 4 * max(1, int(shards))
        self.total = (off + q - 1) // q * q
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.p = {}
        for k in self.names:
            a, b = self.slices[k]
            if tensors is not None:
                self.flat[a:b].copy_(tensors[k].reshape(-1).to(device=device, dtype=torch.float32))
            self.p[k] = self.flat[a:b].view(self.shapes[k]).requires_grad_(True)

    def like(self):
        return torch.zeros_like(self.flat)

    def view_of(self, flat, name):
        a, b = self.slices[name]
        return flat[a:b].view(self.shapes[name])


class FlatAdam:
    """torch.optim.Adam(l, lr=0.0, eps=1e-15) of scene/gaussian_model.py:196-209 over FlatParams, one kernel launch
    per step.  `groups`: name -> lr or (lr, lr2, period, split) (see gsr_adam_segment in include/gsr_hip.h)."""

    def __init__(self, params, grads_flat, groups, betas=(0.9, 0.999), eps=1e-15, owned=None):
        """owned = (begin, end) (multiples of 4): this optimizer only ever steps that range of the flat buffer (one rank's shard of a
        sharded step) and keeps Adam moments for it alone — 1/N of the moment memory."""
        self.params, self.grad = params, grads_flat
        if grads_flat.numel() != params.total or grads_flat.data_ptr() % 16 or params.flat.data_ptr() % 16:
            raise ValueError("gradient buffer must mirror the parameter buffer (same length, 16-byte aligned)")
        self.owned = (0, params.total) if owned is None else (int(owned[0]), int(owned[1]))
        a, b = self.owned
        if not (0 <= a <= b <= params.total) or a % 4 or (b % 4 and b != params.total):
            raise ValueError("owned range must lie inside the buffer with multiples of 4 as bounds")
        self.exp_avg = torch.zeros(b - a, dtype=torch.float32, device=params.flat.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.betas, self.eps, self.step_count = betas, eps, 0
        self.groups = {k: (groups[k] if isinstance(groups[k], tuple) else (float(groups[k]), 0.0, 0, 0)) for k in params.names}

    def set_lr(self, name, lr):
        g = self.groups[name]
        self.groups[name] = (float(lr),) + tuple(g[1:])

    def _segments(self):
        names = self.params.names
        segs = (AdamSegment * len(names))()
        for i, k in enumerate(names):
            a, _ = self.params.slices[k]
            end = self.params.slices[names[i + 1]][0] if i + 1 < len(names) else self.params.total   # padding rides with the group
            lr, lr2, period, split = self.groups[k]
            segs[i] = AdamSegment(a, end, lr, lr2, period, split)
        return segs

    def step(self):
        """One Adam step over the owned range (the whole buffer unless `owned` was given)."""
        self.step_count += 1
        segs = self._segments()
        dev = self.params.flat.device
        _gsr.side_join(dev)      # (the cubemap gradient may still be in flight on the library's side stream)
        a, b = self.owned
        # the kernel indexes all four buffers with the GLOBAL element index; the moment buffers only exist for [a, b), so their base
        # pointers are moved back by `a` elements (a is a multiple of 4: still 16-byte aligned; nothing outside [a, b) is touched)
        with torch.cuda.device(dev):
            check(lib.gsr_adam_step_range(self.params.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr() - 4 * a,
                                          self.exp_avg_sq.data_ptr() - 4 * a, self.params.total, segs, len(segs), self.betas[0], self.betas[1],
                                          self.eps, self.step_count, a, b, stream_ptr(dev)), "gsr_adam_step")


class GaussianTrainState:
    """The optimizer side of the reference's GaussianModel.training_setup (scene/gaussian_model.py:187-223) for the
    surfel + reflection model: groups xyz, f_dc / f_rest (inside shs), opacity, scaling, rotation, refl, env."""

    ORDER = ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths", "cubemap", "fail")

    def __init__(self, tensors, device, spatial_lr_scale=1.0, lrs=None, _params=None, shard=None):
        """shard = (rank, world_size): this rank owns chunk `rank` of `world_size` equal chunks of the flat buffers — its optimizer steps only
        that chunk and keeps moments only for it (gsr_dist.ShardedStep moves gradients in and parameters out)."""
        lr = dict(DEFAULT_LRS)
        lr.update(lrs or {})
        self.lrs, self.spatial_lr_scale = lr, spatial_lr_scale
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        self.params = _params if _params is not None else FlatParams({k: tensors[k] for k in self.ORDER if k in tensors}, device,
                                                                     shards=1 if shard is None else self.shard[1])
        self.p = self.params.p
        self.grads = FlatGrads.mirroring(self.params)
        M = self.params.shapes["shs"][1]
        groups = dict(means3D=lr["position_lr_init"] * spatial_lr_scale,
                      shs=(lr["feature_lr"], lr["feature_lr"] / 20.0, 3 * M, 3),
                      opacities=lr["opacity_lr"], scales=lr["scaling_lr"], rotations=lr["rotation_lr"], refl_strengths=lr["refl_lr"],
                      cubemap=lr["envmap_cubemap_lr"], fail=lr["envmap_cubemap_lr"])
        owned = None
        if self.shard is not None:
            n = self.params.total // self.shard[1]
            owned = (self.shard[0] * n, (self.shard[0] + 1) * n)
        self.optimizer = FlatAdam(self.params, self.grads.flat, {k: groups[k] for k in self.params.names}, owned=owned)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=lr["position_lr_init"] * spatial_lr_scale,
                                                    lr_final=lr["position_lr_final"] * spatial_lr_scale,
                                                    lr_delay_mult=lr["position_lr_delay_mult"], max_steps=lr["position_lr_max_steps"])

    def update_learning_rate(self, iteration):
        # scene/gaussian_model.py:215-221
        v = self.xyz_scheduler_args(iteration)
        self.optimizer.set_lr("means3D", v)
        return v
