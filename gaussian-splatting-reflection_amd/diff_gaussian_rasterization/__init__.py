"""MI355X drop-in for the reference's `diff_gaussian_rasterization` package
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py): the 3DGS rasterizer
with the fork's normal / reflection-strength / inverse-depth / anti-aliasing additions.

Returns of GaussianRasterizer.forward (reference __init__.py:98):
    (color[3,H,W], radii[P] int32, invdepths[1,H,W], normal_map[3,H,W], refl_strength_map[1,H,W])
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, normals, refl_strengths, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, normals, refl_strengths, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    # reference __init__.py:48-155
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, normals, refl_strengths, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        args = (raster_settings.bg, means3D, colors_precomp, normals, refl_strengths, opacities, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
                raster_settings.tanfovx, raster_settings.tanfovy, raster_settings.image_height, raster_settings.image_width, sh,
                raster_settings.sh_degree, raster_settings.campos, raster_settings.prefiltered, raster_settings.antialiasing,
                raster_settings.debug)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths, normal_map, refl_strength_map = \
            _C.rasterize_gaussians(*args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, normals, refl_strengths, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                              geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii, invdepths, normal_map, refl_strength_map

    @staticmethod
    def backward(ctx, grad_out_color, _, grad_out_depth, grad_out_normal_map, grad_out_strength_map):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        colors_precomp, normals, refl_strengths, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer, \
            binningBuffer, imgBuffer = ctx.saved_tensors
        args = (raster_settings.bg, means3D, radii, colors_precomp, normals, refl_strengths, opacities, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
                raster_settings.tanfovx, raster_settings.tanfovy, grad_out_color, grad_out_depth, grad_out_normal_map,
                grad_out_strength_map, sh, raster_settings.sh_degree, raster_settings.campos, geomBuffer, num_rendered, binningBuffer,
                imgBuffer, raster_settings.antialiasing, raster_settings.debug)
        grad_means2D, grad_colors_precomp, grad_normals, grad_refl_strengths, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, \
            grad_scales, grad_rotations = _C.rasterize_gaussians_backward(*args)

        def opt(g, ref):
            return g if (ref is not None and ref.numel() != 0) else None
        return (grad_means3D, grad_means2D, opt(grad_sh, sh), opt(grad_colors_precomp, colors_precomp), grad_normals, grad_refl_strengths,
                grad_opacities, opt(grad_scales, scales), opt(grad_rotations, rotations), opt(grad_cov3Ds_precomp, cov3Ds_precomp), None)


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
    antialiasing: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, normals=None, refl_strengths=None, scales=None,
                rotations=None, cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        # the reference uses CPU `torch.Tensor([])` placeholders here (reference __init__.py:198-208)
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, normals, refl_strengths, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings)
