"""MI355X drop-in for the reference's `diff_gaussian_rasterization` package (variant G: the 3DGS rasterizer with the
fork's normal / reflection-strength / inverse-depth / anti-aliasing additions).

GaussianRasterizer.forward(means3D, means2D, opacities, shs=None, colors_precomp=None, normals=None, refl_strengths=None,
scales=None, rotations=None, cov3D_precomp=None) returns
    (color[3,H,W], radii[P] int32, invdepths[1,H,W], normal_map[3,H,W], refl_strength_map[1,H,W])
Everything variant-independent lives in `_raster_api.py`; this file only describes variant G.
"""
import torch

from _raster_api import Variant, build_api, cpu_deep_copy_tuple  # noqa: F401
from . import _C


def _pack_forward(t, s):
    return (s.bg, t["means3D"], t["colors_precomp"], t["normals"], t["refl_strengths"], t["opacities"], t["scales"], t["rotations"],
            s.scale_modifier, t["cov3Ds_precomp"], s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, t["sh"],
            s.sh_degree, s.campos, s.prefiltered, s.antialiasing, s.debug)


def _split_forward(ret):
    num_rendered, color, radii, geom, binning, img, invdepths, normal_map, refl_map = ret
    return num_rendered, (color, radii, invdepths, normal_map, refl_map), (geom, binning, img), radii


def _pack_backward(saved, s, grads, num_rendered, buffers, radii):
    g_color, _, g_depth, g_normal, g_refl = grads
    geom, binning, img = buffers
    return (s.bg, saved["means3D"], radii, saved["colors_precomp"], saved["normals"], saved["refl_strengths"], saved["opacities"],
            saved["scales"], saved["rotations"], s.scale_modifier, saved["cov3Ds_precomp"], s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
            g_color, g_depth, g_normal, g_refl, saved["sh"], s.sh_degree, s.campos, geom, num_rendered, binning, img, s.antialiasing, s.debug)


def _grads_of(ret):
    means2D, colors, normals, refl, opacity, means3D, cov3D, sh, scales, rotations = ret
    return dict(means3D=means3D, means2D=means2D, sh=sh, colors_precomp=colors, normals=normals, refl_strengths=refl, opacities=opacity,
                scales=scales, rotations=rotations, cov3Ds_precomp=cov3D)


def _placeholder(name, device):
    # this variant's reference substitutes CPU `torch.Tensor([])` for omitted inputs (__init__.py:198-208)
    return torch.Tensor([])


_VARIANT = Variant(
    c_module=_C, extra_settings=("antialiasing",),
    tensors=("means3D", "means2D", "sh", "colors_precomp", "normals", "refl_strengths", "opacities", "scales", "rotations", "cov3Ds_precomp"),
    settings_pos=10,
    forward_kwargs=(("shs", None), ("colors_precomp", None), ("normals", None), ("refl_strengths", None), ("scales", None),
                    ("rotations", None), ("cov3D_precomp", None)),
    module_to_apply={"shs": "sh", "cov3D_precomp": "cov3Ds_precomp"},
    placeholder=_placeholder, pack_forward=_pack_forward, split_forward=_split_forward, nondiff_outputs=(1,),
    saved=("colors_precomp", "normals", "refl_strengths", "means3D", "scales", "rotations", "cov3Ds_precomp", "sh", "opacities"),
    pack_backward=_pack_backward, grads_of=_grads_of,
    optional_grads=("sh", "colors_precomp", "scales", "rotations", "cov3Ds_precomp"),
    sinkable={"means3D": "means3D", "sh": "shs", "opacities": "opacities", "scales": "scales", "rotations": "rotations",
              "refl_strengths": "refl_strengths", "normals": "normals"},
    skippable={"colors_precomp": "colors", "cov3Ds_precomp": "cov3D"})

GaussianRasterizationSettings, _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = build_api(_VARIANT)
