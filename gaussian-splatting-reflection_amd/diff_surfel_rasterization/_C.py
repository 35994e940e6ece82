"""Stand-in for the reference's pybind module `diff_surfel_rasterization._C`
(submodules/diff-surfel-rasterization/ext.cpp:15-19): the same three functions with the same
positional arguments and return tuples, implemented over the C ABI of libgsr_hip.so.

Tensor plumbing follows RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible of
submodules/diff-surfel-rasterization/rasterize_points.cu:39-151, 153-267, 269-288.
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gsr  # noqa: E402
from _gsr import check, f32c, lib, ptr, require_cuda, stream_ptr  # noqa: E402

NUM_CHANNELS = 3
SINKABLE = frozenset(("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths"))   # gradients a grad_sink may take


def rasterize_gaussians(background, means3D, env_scope_mask, colors, refl_strengths, opacity, scales, rotations, scale_modifier,
                        transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, *, refl=None):
    """Same positional arguments and return tuple as the reference's `_C.rasterize_gaussians` (DSR rasterize_points.cu:39-151).
    Keyword-only extension `refl` (round 4, the fused rasterize + reflect path, gsr_surfel_forward_refl): a dict with `cam` (the 33-float
    camera block of gaussian_renderer._cam_block), `cubemap` [6,3,L,L], `fail_value` [3] and `keys` (bool: also write the sort keys of the
    backward's footprint records).  The deferred-reflection pass then runs as the epilogue of the tile kernel and the return tuple grows
    by (final[3,H,W], refl_color[3,H,W], normal_world[3,H,W], cubemap_rgba, sort_keys | None, scratch | None); `early_sort`: the forward
    also sorts the keys on the library's side stream into `scratch`, which the reflection backward then takes (keys_sorted)."""
    if _gsr.PYBIND is not None and refl is None:      # GSR_BINDING=pybind: the compiled marshaling (csrc/gsr_torch_binding.cpp) instead of ctypes
        return _gsr.PYBIND.surfel_rasterize_gaussians(background, means3D, env_scope_mask, colors, refl_strengths, opacity, scales, rotations,
                                                      float(scale_modifier), transMat_precomp, viewmatrix, projmatrix, float(tan_fovx),
                                                      float(tan_fovy), int(image_height), int(image_width), sh, int(degree), campos,
                                                      bool(prefiltered), bool(debug))
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    for name, t in (("background", background), ("means3D", means3D), ("colors", colors), ("refl_strengths", refl_strengths),
                    ("opacity", opacity), ("scales", scales), ("rotations", rotations), ("transMat_precomp", transMat_precomp),
                    ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("sh", sh), ("campos", campos)):
        require_cuda(t, name)
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    dev = means3D.device
    fopts = dict(dtype=torch.float32, device=dev)
    out_color = torch.empty((NUM_CHANNELS, H, W), **fopts)
    out_others = torch.empty((3 + 3 + 1 + 1, H, W), **fopts)
    out_refl = torch.empty((1, H, W), **fopts)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    gaussian_weights = torch.empty((P,), **fopts)
    ws = _gsr.Workspace(dev)
    M = sh.size(1) if sh.numel() != 0 else 0
    mask = env_scope_mask
    if mask is not None and mask.numel() != 0:
        if mask.dtype != torch.bool:
            raise RuntimeError(f"expected scalar type Bool but found {mask.dtype} for env_scope_mask")
        mask = mask.contiguous()
    keep = [f32c(background, "background"), f32c(means3D, "means3D"), f32c(sh, "sh"), f32c(colors, "colors"),
            f32c(refl_strengths, "refl_strengths"), f32c(opacity, "opacity"), f32c(scales, "scales"), f32c(rotations, "rotations"),
            f32c(transMat_precomp, "transMat_precomp"), f32c(viewmatrix, "viewmatrix"), f32c(projmatrix, "projmatrix"),
            f32c(campos, "campos")]
    bg, m3, shc, col, rfl, opa, sca, rot, tmp, vm, pm, cp = keep
    desc, extra = None, ()
    if refl is not None:
        cm, fv, cam = f32c(refl["cubemap"], "cubemap"), f32c(refl["fail_value"], "fail_value"), f32c(refl["cam"], "cam")
        if cm.dim() != 4 or cm.shape[0] != 6 or cm.shape[1] != 3 or cm.shape[2] != cm.shape[3]:
            raise RuntimeError("rasterize + reflect: the cubemap must be (6, 3, L, L)")
        L = int(cm.shape[2])
        final, refl_color, normal_world = (torch.empty((3, H, W), **fopts) for _ in range(3))
        rgba = torch.empty(6 * L * L * 4, **fopts)
        keys = torch.empty(H * W, dtype=torch.int32, device=dev) if refl.get("keys") else None
        # early_sort: the forward also sorts the keys (on the side stream) into the scratch the reflection backward will use
        scratch = None
        if keys is not None and refl.get("early_sort"):
            scratch = torch.empty(int(lib.gsr_deferred_reflection_scratch_floats(L, W, H, 1)), **fopts)
        desc = _gsr.ReflForward(ptr(cam), ptr(cm), ptr(fv), L, ptr(rgba), ptr(final), ptr(refl_color), ptr(normal_world), ptr(keys), ptr(scratch),
                                scratch.numel() if scratch is not None else 0, 1 if scratch is not None else 0)
        keep += [cm, fv, cam]
        extra = (final, refl_color, normal_world, rgba, keys, scratch)
    with torch.cuda.device(dev):
        rendered = check(lib.gsr_surfel_forward_refl(ws.cb, None, P, int(degree), M, ptr(bg), W, H, ptr(m3), ptr(mask), ptr(shc), ptr(col),
                                                     ptr(rfl), ptr(opa), ptr(sca), float(scale_modifier), ptr(rot), ptr(tmp), ptr(vm), ptr(pm),
                                                     ptr(cp), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), ptr(out_color),
                                                     ptr(out_others), ptr(out_refl), ptr(radii), ptr(gaussian_weights),
                                                     ctypes.byref(desc) if desc is not None else None, int(bool(debug)), stream_ptr(dev)),
                         "gsr_surfel_forward")
    if ws.error is not None:
        raise ws.error
    geomBuffer, binningBuffer, imgBuffer = ws.bufs
    return (rendered, out_color, out_others, radii, geomBuffer, binningBuffer, imgBuffer, out_refl, gaussian_weights) + extra


def rasterize_gaussians_backward(background, means3D, radii, colors, refl_strengths, scales, rotations, scale_modifier, transMat_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_others, dL_dout_refl_strength_map, sh,
                                 degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, *, grad_sink=None, accumulate=False, unused=(),
                                 extra_normal_grad=None):
    """Same positional arguments and return tuple as the reference's `_C.rasterize_gaussians_backward`.  Keyword-only
    extension: `grad_sink` maps any of means3D (P,3), shs (P,M,3), opacities (P,1), scales (P,2), rotations (P,4),
    refl_strengths (P,1) to a preallocated contiguous float32 tensor (e.g. views of one flat all-reduce / optimizer buffer,
    gsr_dist.FlatGrads); the per-Gaussian backward kernel then writes — or, with accumulate=True, ADDS — those gradients
    straight into them and the corresponding entries of the return tuple are those same tensors.  The sink belongs to
    this call: there is no module-level state.  `unused` (keyword-only extension, what the autograd wrapper passes): any of "colors",
    "transMat" — gradients of inputs the caller did not supply (shs instead of colors_precomp, scales / rotations instead of
    transMat_precomp); the kernel does not write them and the tuple holds empty tensors in their place.
    `extra_normal_grad` (keyword-only extension): float32 (3,H,W), a second upstream gradient of planes 2..4 of the `others`
    output (the blended view-space normal) that the tile kernel adds to dL_dout_others[2:5] while loading it
    (gsr_surfel_backward_ex) — what the output tap `normal_view` of GaussianRasterizer receives."""
    M = sh.size(1) if sh.numel() != 0 else 0
    if grad_sink:
        unknown = set(grad_sink) - SINKABLE
        if unknown:
            raise ValueError(f"grad sink: unknown gradient name(s) {sorted(unknown)}; expected a subset of {sorted(SINKABLE)}")
    if accumulate and (not grad_sink or not set(grad_sink) >= (SINKABLE - ({"shs"} if M == 0 else set()))):
        # the kernel has ONE accumulate switch for all six parameter gradients: fresh (uninitialised) tensors cannot be added to
        raise ValueError("accumulate=True needs a sink for every parameter gradient: " + ", ".join(sorted(SINKABLE)))
    unused = frozenset(unused)
    if unused - {"colors", "transMat"} or ("colors" in unused and sh.numel() == 0) or ("transMat" in unused and scales.numel() == 0):
        raise ValueError("unused: 'colors' needs shs as the colour input, 'transMat' needs scales / rotations; got %r" % (sorted(unused),))
    if extra_normal_grad is not None:
        hw = tuple(dL_dout_color.shape[1:])
        if tuple(extra_normal_grad.shape) != (3,) + hw or extra_normal_grad.device != means3D.device:
            raise ValueError(f"extra_normal_grad: expected (3, H, W) = {(3,) + hw} on {means3D.device}, "
                             f"got {tuple(extra_normal_grad.shape)} on {extra_normal_grad.device}")
    if _gsr.PYBIND is not None and not grad_sink and extra_normal_grad is None:
        return _gsr.PYBIND.surfel_rasterize_gaussians_backward(
            background, means3D, radii, colors, refl_strengths, scales, rotations, float(scale_modifier), transMat_precomp, viewmatrix, projmatrix,
            float(tan_fovx), float(tan_fovy), dL_dout_color, dL_dout_others,
            dL_dout_refl_strength_map if dL_dout_refl_strength_map is not None else torch.empty(0, device=means3D.device), sh, int(degree), campos,
            geomBuffer, int(R), binningBuffer, imageBuffer, bool(debug), (1 if "colors" in unused else 0) | (2 if "transMat" in unused else 0))
    for name, t in (("background", background), ("means3D", means3D), ("radii", radii), ("colors", colors), ("scales", scales),
                    ("rotations", rotations), ("transMat_precomp", transMat_precomp), ("viewmatrix", viewmatrix),
                    ("projmatrix", projmatrix), ("sh", sh), ("campos", campos), ("binningBuffer", binningBuffer),
                    ("imageBuffer", imageBuffer), ("geomBuffer", geomBuffer)):
        require_cuda(t, name)
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    dev = means3D.device
    o = dict(dtype=torch.float32, device=dev)
    # the library writes every element, so no zero-fill is needed (the reference uses torch::zeros)
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
    # dL_dnormal3D is internal to the reference's backward (never returned): not materialised at all
    dL_dmeans3D, dL_dmeans2D, dL_dnormal = mk((P, 3), "means3D", **o), mk((P, 3), **o), None
    dL_dcolors = torch.empty(0, **o) if "colors" in unused else mk((P, NUM_CHANNELS), **o)
    dL_dtransMat = torch.empty(0, **o) if "transMat" in unused else mk((P, 9), **o)
    dL_dopacity, dL_dsh = mk((P, 1), "opacities", **o), mk((P, M, 3), "shs", **o)
    dL_dscales, dL_drotations, dL_drefl = mk((P, 2), "scales", **o), mk((P, 4), "rotations", **o), mk((P, 1), "refl_strengths", **o)
    if dL_dout_refl_strength_map is None or dL_dout_refl_strength_map.numel() == 0:
        dL_dout_refl_strength_map = torch.zeros((1, H, W), **o)
    if P != 0:
        keep = [f32c(background, "background"), f32c(means3D, "means3D"), f32c(sh, "sh"), f32c(colors, "colors"),
                f32c(refl_strengths, "refl_strengths"), f32c(scales, "scales"), f32c(rotations, "rotations"),
                f32c(transMat_precomp, "transMat_precomp"), f32c(viewmatrix, "viewmatrix"), f32c(projmatrix, "projmatrix"),
                f32c(campos, "campos"), f32c(dL_dout_color, "dL_dout_color"), f32c(dL_dout_others, "dL_dout_others"),
                f32c(dL_dout_refl_strength_map, "dL_dout_refl_strength_map"), radii.contiguous()]
        bg, m3, shc, col, rfl, sca, rot, tmp, vm, pm, cp, gcol, goth, grefl, rad = keep
        gnx = f32c(extra_normal_grad, "extra_normal_grad") if extra_normal_grad is not None else None
        with torch.cuda.device(dev):
            check(lib.gsr_surfel_backward_ex(P, int(degree), M, int(R), ptr(bg), W, H, ptr(m3), ptr(shc), ptr(col), ptr(rfl), ptr(sca),
                                          float(scale_modifier), ptr(rot), ptr(tmp), ptr(vm), ptr(pm), ptr(cp), float(tan_fovx),
                                          float(tan_fovy), ptr(rad), ptr(geomBuffer), ptr(binningBuffer), ptr(imageBuffer), ptr(gcol),
                                          ptr(goth), ptr(grefl), ptr(dL_dmeans2D), ptr(dL_dnormal), ptr(dL_dopacity), ptr(dL_dcolors),
                                          ptr(dL_drefl), ptr(dL_dmeans3D), ptr(dL_dtransMat), ptr(dL_dsh), ptr(dL_dscales),
                                          ptr(dL_drotations), int(bool(accumulate)), ptr(gnx), int(bool(debug)), stream_ptr(dev)), "gsr_surfel_backward")
    return dL_dmeans2D, dL_dcolors, dL_drefl, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations


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
