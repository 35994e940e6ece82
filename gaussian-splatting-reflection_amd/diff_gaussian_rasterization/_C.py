"""Stand-in for the reference's pybind module `diff_gaussian_rasterization._C`
(submodules/diff-gaussian-rasterization/ext.cpp:15-19) over the C ABI of libgsr_hip.so.

Tensor plumbing follows RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible of
submodules/diff-gaussian-rasterization/rasterize_points.cu:38-140, 142-264, 266-285.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gsr  # noqa: E402
from _gsr import check, f32c, lib, ptr, stream_ptr  # noqa: E402

NUM_CHANNELS = 3
# gradients a grad_sink may take (round 4: the extensions variant S has; normals are a per-Gaussian parameter of this variant)
SINKABLE = frozenset(("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths", "normals"))


def _dev(t, dev):
    """The reference's Python wrapper passes CPU `torch.Tensor([])` placeholders (DGR __init__.py:198-208);
    their data pointer is null, so only emptiness matters."""
    return t


def rasterize_gaussians(background, means3D, colors, normals, refl_strengths, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
                        antialiasing, debug):
    if _gsr.PYBIND is not None:      # GSR_BINDING=pybind: the compiled marshaling (csrc/gsr_torch_binding.cpp) instead of ctypes
        return _gsr.PYBIND.gauss_rasterize_gaussians(background, means3D, colors, normals, refl_strengths, opacity, scales, rotations,
                                                     float(scale_modifier), cov3D_precomp, viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy),
                                                     int(image_height), int(image_width), sh, int(degree), campos, bool(prefiltered),
                                                     bool(antialiasing), bool(debug))
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a CUDA tensor")
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    dev = means3D.device
    fopts = dict(dtype=torch.float32, device=dev)
    out_color = torch.empty((NUM_CHANNELS, H, W), **fopts)
    out_normal_map = torch.empty((3, H, W), **fopts)
    out_invdepth = torch.empty((1, H, W), **fopts)
    out_refl = torch.empty((1, H, W), **fopts)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    ws = _gsr.Workspace(dev)
    M = sh.size(1) if sh.numel() != 0 else 0
    keep = [f32c(background, "background"), f32c(means3D, "means3D"), f32c(sh, "sh"), f32c(colors, "colors"), f32c(normals, "normals"),
            f32c(refl_strengths, "refl_strengths"), f32c(opacity, "opacity"), f32c(scales, "scales"), f32c(rotations, "rotations"),
            f32c(cov3D_precomp, "cov3D_precomp"), f32c(viewmatrix, "viewmatrix"), f32c(projmatrix, "projmatrix"), f32c(campos, "campos")]
    bg, m3, shc, col, nrm, refl, opa, sca, rot, cov, vm, pm, cp = keep
    with torch.cuda.device(dev):
        rendered = check(lib.gsr_gauss_forward(ws.cb, None, P, int(degree), M, ptr(bg), W, H, ptr(m3), ptr(shc), ptr(col), ptr(nrm), ptr(refl),
                                               ptr(opa), ptr(sca), float(scale_modifier), ptr(rot), ptr(cov), ptr(vm), ptr(pm), ptr(cp),
                                               float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), ptr(out_color), ptr(out_normal_map),
                                               ptr(out_refl), ptr(out_invdepth), int(bool(antialiasing)), ptr(radii), int(bool(debug)),
                                               stream_ptr(dev)), "gsr_gauss_forward")
    if ws.error is not None:
        raise ws.error
    geomBuffer, binningBuffer, imgBuffer = ws.bufs
    return rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer, out_invdepth, out_normal_map, out_refl


def rasterize_gaussians_backward(background, means3D, radii, colors, normals, refl_strengths, opacities, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_invdepth,
                                 dL_dout_normal_map, dL_dout_refl_strength_map, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                 antialiasing, debug, *, grad_sink=None, accumulate=False, unused=()):
    """Same positional arguments and return tuple as the reference's `_C.rasterize_gaussians_backward` (DGR rasterize_points.cu:142-264).
    Keyword-only extensions, as in diff_surfel_rasterization._C: `grad_sink` maps any of means3D (P,3), shs (P,M,3), opacities (P,1),
    scales (P,3), rotations (P,4), refl_strengths (P,1), normals (P,3) to a preallocated contiguous float32 tensor (16-byte aligned; e.g.
    views of one flat all-reduce / optimizer buffer); the per-Gaussian backward kernel writes — or, with accumulate=True, ADDS — those
    gradients straight into them and the corresponding entries of the return tuple are those same tensors (gsr_gauss_backward_accum).
    `unused` (what the autograd wrapper passes): any of "colors", "cov3D" — gradients of inputs the caller did not supply (shs instead of
    colors_precomp, scales / rotations instead of cov3D_precomp): not computed, empty tensors in the tuple."""
    M_ = sh.size(1) if sh.numel() != 0 else 0
    if grad_sink:
        unknown = set(grad_sink) - SINKABLE
        if unknown:
            raise ValueError(f"grad sink: unknown gradient name(s) {sorted(unknown)}; expected a subset of {sorted(SINKABLE)}")
    if accumulate and (not grad_sink or not set(grad_sink) >= (SINKABLE - ({"shs"} if M_ == 0 else set()))):
        # the kernel has ONE accumulate switch for all parameter gradients: fresh (uninitialised) tensors cannot be added to
        raise ValueError("accumulate=True needs a sink for every parameter gradient: " + ", ".join(sorted(SINKABLE)))
    unused = frozenset(unused)
    if unused - {"colors", "cov3D"} or ("colors" in unused and sh.numel() == 0) or ("cov3D" in unused and scales.numel() == 0):
        raise ValueError("unused: 'colors' needs shs as the colour input, 'cov3D' needs scales / rotations; got %r" % (sorted(unused),))
    if _gsr.PYBIND is not None and not grad_sink:
        e = lambda t: t if t is not None else torch.empty(0, device=means3D.device)
        return _gsr.PYBIND.gauss_rasterize_gaussians_backward(
            background, means3D, radii, colors, normals, refl_strengths, opacities, scales, rotations, float(scale_modifier), cov3D_precomp,
            viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy), dL_dout_color, e(dL_dout_invdepth), dL_dout_normal_map,
            e(dL_dout_refl_strength_map), sh, int(degree), campos, geomBuffer, int(R), binningBuffer, imageBuffer, bool(antialiasing), bool(debug),
            (1 if "colors" in unused else 0) | (2 if "cov3D" in unused else 0))
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.numel() != 0 else 0
    dev = means3D.device
    o = dict(dtype=torch.float32, device=dev)
    mk0 = torch.empty if P != 0 else torch.zeros

    def mk(shape, sink_name=None, **kw):
        # gradient sink: the kernel writes this output straight into a caller-owned tensor
        t = grad_sink.get(sink_name) if (grad_sink and sink_name is not None) else None
        if t is not None:
            if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise ValueError(f"grad sink '{sink_name}': expected contiguous float32 {tuple(shape)} on {dev}, got {tuple(t.shape)} {t.dtype}")
            if t.data_ptr() % 16:
                # the kernel stores dL_dsh / dL_drot rows as float4 (include/gsr_hip.h, "alignment"); the C ABI refuses too
                raise ValueError(f"grad sink '{sink_name}': storage must be 16-byte aligned (got {t.data_ptr():#x}); pad the slices of a packed buffer "
                                 "to multiples of 4 floats as gsr_dist.FlatGrads does")
            return t
        return mk0(shape, **kw)
    # dL_dmean2D (the one that feeds the 3-D gradient) and dL_dconic are intermediates of the reference's backward, never returned: not materialised
    dL_dmeans3D, dL_dmeans2D, dL_dmeans2D_pixels = mk((P, 3), "means3D", **o), None, mk((P, 3), **o)
    dL_dcolors = torch.empty(0, **o) if "colors" in unused else mk((P, NUM_CHANNELS), **o)
    dL_dnormals, dL_dconic = mk((P, 3), "normals", **o), None
    dL_dopacity, dL_dsh = mk((P, 1), "opacities", **o), mk((P, M, 3), "shs", **o)
    dL_dcov3D = torch.empty(0, **o) if "cov3D" in unused else mk((P, 6), **o)
    dL_dscales, dL_drotations = mk((P, 3), "scales", **o), mk((P, 4), "rotations", **o)
    # depth / refl-strength backward are active whenever the incoming grad tensors are non-empty
    # (DGR rasterize_points.cu:196-216)
    has_inv = dL_dout_invdepth is not None and dL_dout_invdepth.numel() != 0
    dL_dinvdepths = mk((P, 1), **o) if has_inv else torch.zeros((0, 1), **o)
    has_refl = dL_dout_refl_strength_map is not None and dL_dout_refl_strength_map.numel() != 0
    if not has_refl:
        dL_dout_refl_strength_map = torch.zeros((1, H, W), **o)
    if accumulate and not has_refl:
        raise ValueError("accumulate=True needs the reflection-strength map's gradient (the reference drops dL_drefl without it)")
    dL_drefl = mk((P, 1), "refl_strengths", **o)
    if P != 0:
        keep = [f32c(background, "background"), f32c(means3D, "means3D"), f32c(sh, "sh"), f32c(colors, "colors"), f32c(normals, "normals"),
                f32c(refl_strengths, "refl_strengths"), f32c(opacities, "opacities"), f32c(scales, "scales"), f32c(rotations, "rotations"),
                f32c(cov3D_precomp, "cov3D_precomp"), f32c(viewmatrix, "viewmatrix"), f32c(projmatrix, "projmatrix"), f32c(campos, "campos"),
                f32c(dL_dout_color, "dL_dout_color"), f32c(dL_dout_normal_map, "dL_dout_normal_map"),
                f32c(dL_dout_refl_strength_map, "dL_dout_refl_strength_map"),
                f32c(dL_dout_invdepth, "dL_dout_invdepth") if has_inv else None, radii.contiguous()]
        bg, m3, shc, col, nrm, refl, opa, sca, rot, cov, vm, pm, cp, gcol, gnrm, grefl, ginv, rad = keep
        with torch.cuda.device(dev):
            check(lib.gsr_gauss_backward_accum(P, int(degree), M, int(R), ptr(bg), W, H, ptr(m3), ptr(shc), ptr(col), ptr(nrm), ptr(refl), ptr(opa),
                                         ptr(sca), float(scale_modifier), ptr(rot), ptr(cov), ptr(vm), ptr(pm), ptr(cp), float(tan_fovx),
                                         float(tan_fovy), ptr(rad), ptr(geomBuffer), ptr(binningBuffer), ptr(imageBuffer), ptr(gcol), ptr(gnrm),
                                         ptr(grefl), ptr(ginv), ptr(dL_dmeans2D), ptr(dL_dmeans2D_pixels), ptr(dL_dconic), ptr(dL_dopacity),
                                         ptr(dL_dcolors), ptr(dL_dnormals), ptr(dL_drefl), ptr(dL_dinvdepths), ptr(dL_dmeans3D),
                                         ptr(dL_dcov3D), ptr(dL_dsh), ptr(dL_dscales), ptr(dL_drotations), int(bool(antialiasing)),
                                         int(bool(accumulate)), int(bool(debug)), stream_ptr(dev)), "gsr_gauss_backward")
    if not has_refl:
        dL_drefl = torch.zeros((0, 1), **o)
    return (dL_dmeans2D_pixels, dL_dcolors, dL_dnormals, dL_drefl, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)


def mark_visible(means3D, viewmatrix, projmatrix):
    if _gsr.PYBIND is not None:
        return _gsr.PYBIND.mark_visible(means3D, viewmatrix, projmatrix)
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        m3, vm, pm = f32c(means3D, "means3D"), f32c(viewmatrix, "viewmatrix"), f32c(projmatrix, "projmatrix")
        with torch.cuda.device(means3D.device):
            check(lib.gsr_mark_visible(P, ptr(m3), ptr(vm), ptr(pm), ptr(present), stream_ptr(means3D.device)), "gsr_mark_visible")
    return present
