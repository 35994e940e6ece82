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

_interp_to_id = {
    'nearest': 0,
    'linear': 1
}


class _cubemap_encode(torch.autograd.Function):
    # reference cubemap_encoder.py:19-63 (custom_fwd(cast_inputs=float32): inputs are cast to fp32)
    @staticmethod
    def forward(ctx, inputs, embeddings, fail_value, interpolation, enable_seamless):
        embeddings = embeddings.float().contiguous()
        inputs = inputs.float().contiguous()
        fail_value = fail_value.float().contiguous()
        C = embeddings.shape[1]
        L = embeddings.shape[2]
        B = inputs.shape[0]
        outputs = torch.empty([C, B], dtype=embeddings.dtype, device=embeddings.device)
        _backend.cubemap_encode_forward(inputs, embeddings, fail_value, outputs, interpolation, enable_seamless, B, C, L)
        ctx.params = (int(interpolation), int(enable_seamless))
        ctx.save_for_backward(inputs, embeddings)
        return outputs

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, embeddings = ctx.saved_tensors
        grad_outputs = grad_outputs.float().contiguous()
        C = embeddings.shape[1]
        L = embeddings.shape[2]
        B = inputs.shape[0]
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.empty_like(inputs)
        grad_fail = torch.zeros([C], dtype=embeddings.dtype, device=embeddings.device)
        _backend.cubemap_encode_backward(grad_outputs, inputs, embeddings, grad_embeddings, grad_inputs, grad_fail, ctx.params[0],
                                         ctx.params[1], B, C, L)
        return grad_inputs, grad_embeddings, grad_fail, None, None


cubemap_encode = _cubemap_encode.apply


def _adjust_sharpness(img, factor):
    """torchvision.transforms.functional.adjust_sharpness restated in plain torch (torchvision is not a
    dependency here): blend with a 3x3 smoothed copy (kernel [[1,1,1],[1,5,1],[1,1,1]]/13), borders kept."""
    if img.shape[-1] <= 2 or img.shape[-2] <= 2:
        return img
    k = torch.ones(3, 3, dtype=img.dtype, device=img.device)
    k[1, 1] = 5.0
    k = (k / k.sum()).expand(img.shape[-3], 1, 3, 3)
    x = img.reshape(-1, img.shape[-3], img.shape[-2], img.shape[-1])
    blurred = nn.functional.conv2d(x, k, groups=x.shape[1])
    degenerate = x.clone()
    degenerate[..., 1:-1, 1:-1] = blurred
    out = (factor * x + (1.0 - factor) * degenerate).clamp(0, 1)
    return out.reshape(img.shape)


class CubemapEncoder(nn.Module):
    # reference cubemap_encoder.py:81-123
    def __init__(self, output_dim=6, resolution=256, interpolation='linear'):
        super().__init__()
        self.input_dim = 3
        self.resolution = resolution
        self.output_dim = output_dim
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.seamless = 1
        self.params = nn.ParameterDict({
            'Cubemap_texture': nn.Parameter(torch.rand(6, self.output_dim, resolution, resolution) - 0.5),
            'Cubemap_failv': nn.Parameter(torch.zeros(self.output_dim))
        })
        self.n_elems = 6 * self.output_dim * resolution * resolution + self.output_dim

    def __repr__(self):
        return (f"CubemapEncoder: input_dim={self.input_dim} output_dim={self.output_dim} resolution={self.resolution} -> {self.n_elems} "
                f"interpolation={self.interpolation} seamless={self.seamless}")

    def resize(self, new_resolution):
        self.resolution = new_resolution
        self.params['Cubemap_texture'] = nn.functional.interpolate(self.params['Cubemap_texture'], size=(new_resolution, new_resolution),
                                                                   mode='bicubic', align_corners=True)
        self.n_elems = 6 * self.output_dim * self.resolution * self.resolution + self.output_dim

    def filter(self, activation, inverse_activation, factor=2.0):
        textures = self.params['Cubemap_texture']
        textures = activation(textures)
        textures = _adjust_sharpness(textures, factor)
        textures = torch.clamp(textures, min=1e-3, max=1 - 1e-3)
        textures = inverse_activation(textures)
        self.params['Cubemap_texture'] = textures

    def set_textures(self, textures):
        self.resolution = textures.shape[2]
        self.params['Cubemap_texture'] = nn.Parameter(textures)
        self.n_elems = 6 * self.output_dim * self.resolution * self.resolution + self.output_dim

    def forward(self, inputs):
        outputs = cubemap_encode(inputs, self.params['Cubemap_texture'], self.params['Cubemap_failv'], self.interp_id, self.seamless)
        return outputs.permute(1, 0)  # CxN -> NxC


class MipCubemapEncoder(nn.Module):
    # reference cubemap_encoder.py:126-178 (never instantiated by the reference; kept for API completeness)
    def __init__(self, num_levels=4, level_dim=6, per_level_scale=4, base_resolution=4, interpolation='linear', concat=True):
        super().__init__()
        self.input_dim = 3
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.base_resolution = base_resolution
        self.concat = concat
        self.output_dim = num_levels * level_dim if concat else level_dim
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.seamless = 1
        params_list = []
        L = float(base_resolution)
        n_elems = 0
        for _ in range(num_levels):
            iL = int(np.ceil(L))
            params_list.append(nn.Parameter(torch.empty(6, self.level_dim, iL, iL)))
            n_elems += 6 * self.level_dim * iL * iL
            L = L * per_level_scale
        self.params_list = nn.ParameterList(params_list)
        self.fail_value = nn.Parameter(torch.zeros(self.level_dim))
        self.n_elems = n_elems + self.level_dim
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        for ii in range(self.num_levels):
            self.params_list[ii].data.uniform_(-std, std)

    def forward(self, inputs):
        outputs = [cubemap_encode(inputs, self.params_list[ii], self.fail_value, self.interp_id, self.seamless) for ii in range(self.num_levels)]
        outputs = torch.cat(outputs, dim=0) if self.concat else sum(outputs)
        return outputs.permute(1, 0)
