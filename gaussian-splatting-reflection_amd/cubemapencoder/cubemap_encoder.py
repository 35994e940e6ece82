"""Cubemap environment-map encoder: same classes and autograd function as
submodules/cubemapencoder/cubemapencoder/cubemap_encoder.py, with `_backend.cubemap_encode_forward /
cubemap_encode_backward` (CME src/bindings.cpp:5-8) provided by libgsr_hip.so.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gsr  # noqa: E402
from _gsr import check, lib, ptr, stream_ptr  # noqa: E402


class _Backend:
    """Signature-compatible with the reference's `_cubemapencoder` pybind module: outputs are
    caller-allocated, functions return None (CME src/cubemapencoder.h:6-17)."""

    @staticmethod
    def _chk(t, name):
        # CHECK_CUDA / CHECK_CONTIGUOUS / CHECK_IS_FLOATING of CME cubemapencoder.cu:23-26
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous tensor")
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be a float32 tensor (the HIP backend is fp32-only, as the reference's backward is)")

    @staticmethod
    def cubemap_encode_forward(inputs, cubemap, fail_value, outputs, interp, seamless, B, C, L):
        for n, t in (("inputs", inputs), ("cubemap", cubemap), ("fail_value", fail_value), ("outputs", outputs)):
            _Backend._chk(t, n)
        with torch.cuda.device(cubemap.device):
            check(lib.gsr_cubemap_forward(ptr(inputs), ptr(cubemap), ptr(fail_value), ptr(outputs), int(interp), int(seamless), int(B), int(C),
                                          int(L), stream_ptr(cubemap.device)), "gsr_cubemap_forward")

    @staticmethod
    def cubemap_encode_backward(grad_outputs, inputs, cubemap, grad_cubemap, grad_inputs, grad_fail, interp, seamless, B, C, L):
        for n, t in (("grad_outputs", grad_outputs), ("inputs", inputs), ("cubemap", cubemap), ("grad_cubemap", grad_cubemap),
                     ("grad_inputs", grad_inputs), ("grad_fail", grad_fail)):
            _Backend._chk(t, n)
        with torch.cuda.device(cubemap.device):
            check(lib.gsr_cubemap_backward(ptr(grad_outputs), ptr(inputs), ptr(cubemap), ptr(grad_cubemap), ptr(grad_inputs), ptr(grad_fail),
                                           int(interp), int(seamless), int(B), int(C), int(L), stream_ptr(cubemap.device)),
                  "gsr_cubemap_backward")


_backend = _Backend()
_interp_to_id = dict(nearest=0, linear=1)


class _cubemap_encode(torch.autograd.Function):
    """(directions [B,3], texture [6,C,L,L], fail_value [C]) -> features [C,B]; gradients for all three.
    Inputs are brought to contiguous fp32 first (the reference wraps this in custom_fwd(cast_inputs=float32),
    cubemap_encoder.py:19-63)."""

    @staticmethod
    def forward(ctx, inputs, embeddings, fail_value, interpolation, enable_seamless):
        dirs, tex, fail = (x.float().contiguous() for x in (inputs, embeddings, fail_value))
        B, (C, L) = dirs.shape[0], tex.shape[1:3]
        out = tex.new_empty((C, B))
        _backend.cubemap_encode_forward(dirs, tex, fail, out, interpolation, enable_seamless, B, C, L)
        ctx.mode = (int(interpolation), int(enable_seamless))
        ctx.save_for_backward(dirs, tex)
        return out

    @staticmethod
    def backward(ctx, grad_outputs):
        dirs, tex = ctx.saved_tensors
        B, (C, L) = dirs.shape[0], tex.shape[1:3]
        g_out = grad_outputs.float().contiguous()
        g_tex, g_dirs, g_fail = torch.zeros_like(tex), torch.empty_like(dirs), tex.new_zeros((C,))
        _backend.cubemap_encode_backward(g_out, dirs, tex, g_tex, g_dirs, g_fail, ctx.mode[0], ctx.mode[1], B, C, L)
        return g_dirs, g_tex, g_fail, None, None


cubemap_encode = _cubemap_encode.apply


def _adjust_sharpness(img, factor):
    """torchvision.transforms.functional.adjust_sharpness restated in plain torch (torchvision is not a
    dependency here): blend with a 3x3 smoothed copy (kernel [[1,1,1],[1,5,1],[1,1,1]]/13), borders kept."""
    if min(img.shape[-2:]) <= 2:
        return img
    chans = img.shape[-3]
    kernel = torch.ones(3, 3, dtype=img.dtype, device=img.device)
    kernel[1, 1] = 5.0
    kernel = (kernel / kernel.sum()).expand(chans, 1, 3, 3)
    batch = img.reshape(-1, chans, *img.shape[-2:])
    smooth = batch.clone()
    smooth[..., 1:-1, 1:-1] = nn.functional.conv2d(batch, kernel, groups=chans)
    return (factor * batch + (1.0 - factor) * smooth).clamp(0, 1).reshape(img.shape)


class _CubeLookup(nn.Module):
    """What both encoders share: direction input, interpolation mode, seamless edge handling."""
    input_dim = 3
    seamless = 1

    def _set_mode(self, interpolation):
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]

    def _lookup(self, dirs, texture, fail_value):
        return cubemap_encode(dirs, texture, fail_value, self.interp_id, self.seamless)


class CubemapEncoder(_CubeLookup):
    """One 6 x C x L x L texture + a C-vector returned for the zero direction (reference cubemap_encoder.py:81-123).
    `params` keeps the reference's ParameterDict keys because checkpoints (`.map` files) are its state dict."""

    def __init__(self, output_dim=6, resolution=256, interpolation='linear'):
        super().__init__()
        self.output_dim, self.resolution = output_dim, resolution
        self._set_mode(interpolation)
        self.params = nn.ParameterDict(dict(Cubemap_texture=nn.Parameter(torch.rand(6, output_dim, resolution, resolution) - 0.5),
                                            Cubemap_failv=nn.Parameter(torch.zeros(output_dim))))

    @property
    def n_elems(self):
        return 6 * self.output_dim * self.resolution ** 2 + self.output_dim

    def __repr__(self):
        return (f"CubemapEncoder: input_dim={self.input_dim} output_dim={self.output_dim} resolution={self.resolution} -> {self.n_elems} "
                f"interpolation={self.interpolation} seamless={self.seamless}")

    def set_textures(self, textures):
        self.resolution = textures.shape[2]
        self.params['Cubemap_texture'] = nn.Parameter(textures)

    def resize(self, new_resolution):
        up = nn.functional.interpolate(self.params['Cubemap_texture'], size=(new_resolution,) * 2, mode='bicubic', align_corners=True)
        self.resolution = new_resolution
        self.params['Cubemap_texture'] = up

    def filter(self, activation, inverse_activation, factor=2.0):
        sharpened = _adjust_sharpness(activation(self.params['Cubemap_texture']), factor)
        self.params['Cubemap_texture'] = inverse_activation(sharpened.clamp(min=1e-3, max=1 - 1e-3))

    def forward(self, inputs):
        return self._lookup(inputs, self.params['Cubemap_texture'], self.params['Cubemap_failv']).permute(1, 0)   # CxN -> NxC


class MipCubemapEncoder(_CubeLookup):
    """A pyramid of cubemaps whose features are concatenated or summed (reference cubemap_encoder.py:126-178; the
    reference never instantiates it, it is here for API completeness)."""

    def __init__(self, num_levels=4, level_dim=6, per_level_scale=4, base_resolution=4, interpolation='linear', concat=True):
        super().__init__()
        self.num_levels, self.level_dim, self.per_level_scale = num_levels, level_dim, per_level_scale
        self.base_resolution, self.concat = base_resolution, concat
        self.output_dim = num_levels * level_dim if concat else level_dim
        self._set_mode(interpolation)
        sizes = [int(np.ceil(float(base_resolution) * per_level_scale ** lvl)) for lvl in range(num_levels)]
        self.params_list = nn.ParameterList([nn.Parameter(torch.empty(6, level_dim, n, n)) for n in sizes])
        self.fail_value = nn.Parameter(torch.zeros(level_dim))
        self.n_elems = sum(6 * level_dim * n * n for n in sizes) + level_dim
        self.reset_parameters()

    def reset_parameters(self, std=1e-4):
        for tex in self.params_list:
            tex.data.uniform_(-std, std)

    def forward(self, inputs):
        levels = [self._lookup(inputs, tex, self.fail_value) for tex in self.params_list]
        return (torch.cat(levels, dim=0) if self.concat else sum(levels)).permute(1, 0)
