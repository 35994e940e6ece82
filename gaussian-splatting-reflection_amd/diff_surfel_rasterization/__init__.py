"""MI355X drop-in for the reference's `diff_surfel_rasterization` package
(submodules/diff-surfel-rasterization/diff_surfel_rasterization/__init__.py): same
`GaussianRasterizationSettings` / `GaussianRasterizer` / `rasterize_gaussians` surface, same
argument meaning and error behaviour, backed by hand-written HIP kernels (libgsr_hip.so).

Returns of GaussianRasterizer.forward (reference __init__.py:106):
    (color[3,H,W], radii[P] int32, allmap[8,H,W], refl_strength_map[1,H,W], gaussian_weights[P])
allmap planes: 0 depth, 1 alpha, 2-4 view-space normal, 5 median depth, 6 distortion, 7 env-scope mask.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, refl_strengths, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, env_scope_mask):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, refl_strengths, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, env_scope_mask)


class _RasterizeGaussians(torch.autograd.Function):
    # reference __init__.py:50-166
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, refl_strengths, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, env_scope_mask):
        args = (raster_settings.bg, means3D, env_scope_mask, colors_precomp, refl_strengths, opacities, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
                raster_settings.tanfovx, raster_settings.tanfovy, raster_settings.image_height, raster_settings.image_width, sh,
                raster_settings.sh_degree, raster_settings.campos, raster_settings.prefiltered, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted
            try:
                num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer, refl_strength_map, gaussian_weights = \
                    _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer, refl_strength_map, gaussian_weights = \
                _C.rasterize_gaussians(*args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, refl_strengths, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii, gaussian_weights)
        return color, radii, depth, refl_strength_map, gaussian_weights

    @staticmethod
    def backward(ctx, grad_out_color, _, grad_depth, grad_out_strength_map, __):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        colors_precomp, refl_strengths, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
            ctx.saved_tensors
        args = (raster_settings.bg, means3D, radii, colors_precomp, refl_strengths, scales, rotations, raster_settings.scale_modifier,
                cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy,
                grad_out_color, grad_depth, grad_out_strength_map, sh, raster_settings.sh_degree, raster_settings.campos, geomBuffer,
                num_rendered, binningBuffer, imgBuffer, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                grads_ = _C.rasterize_gaussians_backward(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            grads_ = _C.rasterize_gaussians_backward(*args)
        grad_means2D, grad_colors_precomp, grad_refl_strengths, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, \
            grad_scales, grad_rotations = grads_
        # autograd insists on None for inputs that were passed as empty placeholders
        sink = _C.grad_sink or {}
        def opt(g, ref, name=None):
            if name is not None and name in sink:
                return None   # already written into the caller's sink tensor (see _C.set_grad_sink)
            return g if (ref is not None and ref.numel() != 0) else None
        return (opt(grad_means3D, means3D, "means3D"), grad_means2D, opt(grad_sh, sh, "shs"), opt(grad_colors_precomp, colors_precomp),
                opt(grad_refl_strengths, refl_strengths, "refl_strengths"), (None if "opacities" in sink else grad_opacities),
                opt(grad_scales, scales, "scales"), opt(grad_rotations, rotations, "rotations"),
                opt(grad_cov3Ds_precomp, cov3Ds_precomp), None, None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    @staticmethod
    def set_grad_sink(sink):
        """Extension (not in the reference): route this rasterizer's parameter gradients into preallocated tensors,
        see _C.set_grad_sink.  Pass None to restore plain autograd behaviour."""
        _C.set_grad_sink(sink)

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, refl_strengths=None, scales=None, rotations=None,
                cov3D_precomp=None, env_scope_mask=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        dev = means3D.device
        empty = lambda: torch.empty(0, dtype=torch.float32, device=dev)
        if shs is None:
            shs = empty()
        if colors_precomp is None:
            colors_precomp = empty()
        if scales is None:
            scales = empty()
        if rotations is None:
            rotations = empty()
        if cov3D_precomp is None:
            cov3D_precomp = empty()
        if env_scope_mask is None:
            env_scope_mask = torch.empty(0, dtype=torch.bool, device=dev)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, refl_strengths, opacities, scales, rotations, cov3D_precomp,
                                   raster_settings, env_scope_mask)
