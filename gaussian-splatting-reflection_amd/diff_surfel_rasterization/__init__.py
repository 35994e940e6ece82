"""MI355X drop-in for the reference's `diff_surfel_rasterization` package (variant S, the 2D-surfel rasterizer the
reference's `gaussian_renderer` calls): same `GaussianRasterizationSettings` / `GaussianRasterizer` /
`rasterize_gaussians` surface, argument meaning and error behaviour, backed by hand-written HIP kernels (libgsr_hip.so).

GaussianRasterizer.forward(means3D, means2D, opacities, shs=None, colors_precomp=None, refl_strengths=None, scales=None,
rotations=None, cov3D_precomp=None, env_scope_mask=None) returns
    (color[3,H,W], radii[P] int32, allmap[8,H,W], refl_strength_map[1,H,W], gaussian_weights[P])
allmap planes: 0 depth, 1 alpha, 2-4 view-space normal, 5 median depth, 6 distortion, 7 env-scope mask.
Everything variant-independent lives in `_raster_api.py`; this file only describes variant S.
"""
import torch

from _raster_api import Variant, build_api, cpu_deep_copy_tuple  # noqa: F401
from . import _C


def _pack_forward(t, s):
    return (s.bg, t["means3D"], t["env_scope_mask"], t["colors_precomp"], t["refl_strengths"], t["opacities"], t["scales"], t["rotations"],
            s.scale_modifier, t["cov3Ds_precomp"], s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, t["sh"],
            s.sh_degree, s.campos, s.prefiltered, s.debug)


def _split_forward(ret):
    num_rendered, color, others, radii, geom, binning, img, refl_map, weights = ret
    return num_rendered, (color, radii, others, refl_map, weights), (geom, binning, img), radii


def _pack_backward(saved, s, grads, num_rendered, buffers, radii):
    g_color, _, g_others, g_refl, _ = grads
    geom, binning, img = buffers
    return (s.bg, saved["means3D"], radii, saved["colors_precomp"], saved["refl_strengths"], saved["scales"], saved["rotations"],
            s.scale_modifier, saved["cov3Ds_precomp"], s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, g_color, g_others, g_refl,
            saved["sh"], s.sh_degree, s.campos, geom, num_rendered, binning, img, s.debug)


def _grads_of(ret):
    means2D, colors, refl, opacity, means3D, trans, sh, scales, rotations = ret
    return dict(means3D=means3D, means2D=means2D, sh=sh, colors_precomp=colors, refl_strengths=refl, opacities=opacity, scales=scales,
                rotations=rotations, cov3Ds_precomp=trans)


def _placeholder(name, device):
    # the reference substitutes empty CUDA tensors for omitted inputs (__init__.py:210-223)
    return torch.empty(0, dtype=torch.bool if name == "env_scope_mask" else torch.float32, device=device)


_VARIANT = Variant(
    c_module=_C, extra_settings=(),
    tensors=("means3D", "means2D", "sh", "colors_precomp", "refl_strengths", "opacities", "scales", "rotations", "cov3Ds_precomp",
             "env_scope_mask"),
    settings_pos=9,
    forward_kwargs=(("shs", None), ("colors_precomp", None), ("refl_strengths", None), ("scales", None), ("rotations", None),
                    ("cov3D_precomp", None), ("env_scope_mask", None)),
    module_to_apply={"shs": "sh", "cov3D_precomp": "cov3Ds_precomp"},
    placeholder=_placeholder, pack_forward=_pack_forward, split_forward=_split_forward, nondiff_outputs=(1, 4),
    saved=("colors_precomp", "refl_strengths", "means3D", "scales", "rotations", "cov3Ds_precomp", "sh"),
    pack_backward=_pack_backward, grads_of=_grads_of,
    optional_grads=("sh", "colors_precomp", "refl_strengths", "scales", "rotations", "cov3Ds_precomp", "env_scope_mask"),
    sinkable={"means3D": "means3D", "sh": "shs", "opacities": "opacities", "scales": "scales", "rotations": "rotations",
              "refl_strengths": "refl_strengths"},
    skippable={"colors_precomp": "colors", "cov3Ds_precomp": "transMat"},
    taps={"normal_view": (2, 2, 5, "extra_normal_grad")},        # allmap[2:5], the reflection pass's input
    snapshot_on_debug=True)

GaussianRasterizationSettings, _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = build_api(_VARIANT)

