"""Shared host side of the two rasterizer packages.

`diff_surfel_rasterization` (variant S) and `diff_gaussian_rasterization` (variant G) expose the same three public
objects as the reference's packages of those names — `GaussianRasterizationSettings`, `GaussianRasterizer`,
`rasterize_gaussians` (+ the autograd Function `_RasterizeGaussians`) — and differ only in which per-Gaussian tensors
they take and which maps they return.  Instead of two hand-written copies, both are produced here from a small
description of the variant (`Variant`): the order of the positional tensors, which of them may be omitted, how the
arguments of the `_C` entry points are laid out, and which outputs are differentiable.

Reference surfaces reproduced (argument names, order, defaults, return tuples, exception texts):
  S: submodules/diff-surfel-rasterization/diff_surfel_rasterization/__init__.py:21-240
  G: submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py:20-225
"""
import inspect
from collections import namedtuple
from dataclasses import dataclass, field
from typing import Callable, Dict, Sequence, Tuple

import torch
import torch.nn as nn

SETTINGS_COMMON = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
                   "campos", "prefiltered", "debug")
MSG_COLOR = 'Please provide excatly one of either SHs or precomputed colors!'        # (sic) the reference's wording
MSG_COV = 'Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!'


def cpu_deep_copy_tuple(input_tuple):
    """Host copies of every tensor of an argument tuple (used by the debug snapshot path)."""
    return tuple(x.cpu().clone() if isinstance(x, torch.Tensor) else x for x in input_tuple)


class GradSink:
    """Gradient sink of one forward/backward pair (extension, not in the reference): `tensors` maps a gradient name
    (means3D, shs, opacities, scales, rotations, refl_strengths, ...) to a preallocated contiguous float32 tensor — e.g.
    views of one flat all-reduce / optimizer buffer (gsr_dist.FlatGrads).  The per-Gaussian backward kernel writes
    (accumulate=False) or adds (accumulate=True) those gradients straight into them and autograd receives None for the
    corresponding inputs: no zero-fill, no `grad += new` pass.  Not a tensor, so autograd passes it through untouched."""

    def __init__(self, tensors, accumulate=False, async_tail=False, taps=()):
        self.tensors = dict(tensors or {})
        self.accumulate = bool(accumulate)
        self.async_tail = bool(async_tail)     # deferred_reflection only: see its docstring
        self.taps = tuple(taps)                # rasterizer only: output taps of this forward (GaussianRasterizer.set_output_taps)


@dataclass
class Variant:
    c_module: object                                   # the package's _C module (ctypes-backed)
    extra_settings: Tuple[str, ...]                    # settings fields after the common twelve
    tensors: Tuple[str, ...]                           # positional tensors of Function.apply, in order (settings excluded)
    settings_pos: int                                  # where raster_settings sits among apply()'s arguments
    forward_kwargs: Tuple[Tuple[str, object], ...]     # GaussianRasterizer.forward signature after (means3D, means2D, opacities)
    module_to_apply: Dict[str, str]                    # forward() keyword -> apply() tensor name where they differ
    placeholder: Callable                              # (name, device) -> tensor standing in for an omitted optional input
    pack_forward: Callable                             # (tensors dict, settings) -> args of _C.rasterize_gaussians
    split_forward: Callable                            # _C return tuple -> (num_rendered, outputs, buffers, radii)
    nondiff_outputs: Tuple[int, ...]                   # indices of `outputs` marked non-differentiable
    saved: Tuple[str, ...]                             # tensors kept for the backward
    pack_backward: Callable                            # (saved dict, settings, grad_outputs, num_rendered, buffers, radii) -> _C args
    grads_of: Callable                                 # _C backward return tuple -> dict tensor name -> gradient
    optional_grads: Tuple[str, ...]                    # inputs whose gradient is None when they were passed as placeholders
    sinkable: Dict[str, str] = field(default_factory=dict)   # apply() tensor name -> gradient-sink key
    skippable: Dict[str, str] = field(default_factory=dict)  # apply() tensor name -> `unused` key of the _C backward: not computed when the input was a placeholder
    taps: Dict[str, Tuple[int, int, int, str]] = field(default_factory=dict)   # tap name -> (output index, first plane, last plane + 1, _C backward keyword)
    snapshot_on_debug: bool = False


def build_api(v: Variant):
    Settings = namedtuple("GaussianRasterizationSettings", SETTINGS_COMMON + v.extra_settings)
    n_args = len(v.tensors) + 1

    def split_args(args):
        args = list(args)
        sink = args.pop() if len(args) == n_args + 1 else None      # trailing GradSink (extension) or absent
        settings = args.pop(v.settings_pos)
        return dict(zip(v.tensors, args)), settings, sink

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *args):
            t, settings, sink = split_args(args)
            ctx.grad_sink, ctx.n_inputs = sink, len(args)
            c_args = v.pack_forward(t, settings)
            if v.snapshot_on_debug and settings.debug:
                host_copy = cpu_deep_copy_tuple(c_args)          # taken before anything can corrupt the inputs
                try:
                    ret = v.c_module.rasterize_gaussians(*c_args)
                except Exception as ex:
                    torch.save(host_copy, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                ret = v.c_module.rasterize_gaussians(*c_args)
            num_rendered, outputs, buffers, radii = v.split_forward(ret)
            ctx.raster_settings, ctx.num_rendered = settings, num_rendered
            ctx.save_for_backward(*[t[k] for k in v.saved], radii, *buffers)
            ctx.mark_non_differentiable(*[outputs[i] for i in v.nondiff_outputs])
            # autograd would otherwise hand backward a zero-FILLED tensor for every output without a gradient, including the
            # non-differentiable ones (radii, gaussian_weights: two P-sized fill dispatches per step that nobody reads)
            ctx.set_materialize_grads(False)
            ctx.out_meta = [(tuple(o.shape), o.dtype, o.device) for o in outputs]
            # output taps (extension): plane ranges of an output returned as further outputs that ALIAS it.  A consumer that reads the
            # tap sends its gradient to backward() as a separate argument, and the tile kernel adds it while loading the upstream
            # planes; through `output[a:b]` autograd would zero-fill a full-size gradient, copy the planes in and add the two images.
            ctx.taps = sink.taps if sink is not None else ()
            if ctx.taps:
                outputs = tuple(outputs) + tuple(outputs[v.taps[n][0]][v.taps[n][1]:v.taps[n][2]] for n in ctx.taps)
            return outputs

        @staticmethod
        def backward(ctx, *grad_outputs):
            settings = ctx.raster_settings
            kept = ctx.saved_tensors
            saved = dict(zip(v.saved, kept[:len(v.saved)]))
            radii, buffers = kept[len(v.saved)], kept[len(v.saved) + 1:]
            tap_grads = {v.taps[n][3]: g for n, g in zip(ctx.taps, grad_outputs[len(ctx.out_meta):]) if g is not None}
            grad_outputs = grad_outputs[:len(ctx.out_meta)]
            grad_outputs = [torch.zeros(m[0], dtype=m[1], device=m[2]) if (g is None and i not in v.nondiff_outputs) else g
                            for i, (g, m) in enumerate(zip(grad_outputs, ctx.out_meta))]
            c_args = v.pack_backward(saved, settings, grad_outputs, ctx.num_rendered, buffers, radii)
            sink_kw = {"grad_sink": ctx.grad_sink.tensors, "accumulate": ctx.grad_sink.accumulate} if (ctx.grad_sink and ctx.grad_sink.tensors) else {}
            sink_kw.update(tap_grads)
            # gradients of inputs that were passed as empty placeholders are dropped below anyway: tell the kernel not to write them
            unused = tuple(key for name, key in v.skippable.items() if saved.get(name) is None or saved[name].numel() == 0)
            if unused:
                sink_kw["unused"] = unused
            if v.snapshot_on_debug and settings.debug:
                host_copy = cpu_deep_copy_tuple(c_args)
                try:
                    ret = v.c_module.rasterize_gaussians_backward(*c_args, **sink_kw)
                except Exception as ex:
                    torch.save(host_copy, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise ex
            else:
                ret = v.c_module.rasterize_gaussians_backward(*c_args, **sink_kw)
            g = v.grads_of(ret)
            sink = ctx.grad_sink.tensors if ctx.grad_sink is not None else {}
            out = []
            for name in v.tensors:
                grad = g.get(name)
                if name in v.sinkable and v.sinkable[name] in sink:
                    grad = None            # already written (or added) into the caller's sink tensor by the backward kernel
                elif name in v.optional_grads and (saved.get(name) is None or saved[name].numel() == 0):
                    grad = None            # autograd wants None for inputs that were passed as empty placeholders
                out.append(grad)
            out.insert(v.settings_pos, None)
            out += [None] * (ctx.n_inputs - len(out))     # the trailing GradSink argument, when present
            return tuple(out)

    def rasterize_gaussians(*args, grad_sink=None):
        if len(args) != n_args:
            raise TypeError(f"rasterize_gaussians() takes {n_args} positional arguments but {len(args)} were given")
        if grad_sink is None:
            return _RasterizeGaussians.apply(*args)
        return _RasterizeGaussians.apply(*args, grad_sink)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings
            self._grad_sink = None
            self._taps = ()

        def set_grad_sink(self, sink, accumulate=False):
            """Extension (not in the reference): route THIS rasterizer's parameter gradients into caller-owned tensors.
            `sink`: dict gradient name -> preallocated contiguous float32 tensor (see GradSink), or None to restore plain
            autograd.  The sink is bound to each forward call made while it is set (it travels through the autograd
            context), so two rasterizers, or two backward passes in flight, never see each other's sinks.
            accumulate=False: the backward kernel overwrites the sink with this backward's gradient; True: it adds to it
            (several views per optimizer step: zero the buffer once, then every backward accumulates on the device)."""
            if sink and not v.sinkable:
                raise NotImplementedError("this rasterizer variant has no gradient sinks")
            self._grad_sink = GradSink(sink, accumulate) if sink else None

        def set_output_taps(self, names=()):
            """Extension (not in the reference): forward() returns, after the reference's tuple, one more tensor per name —
            a plane range of one of the outputs, aliasing it (variant S: "normal_view" = allmap[2:5], what the reference's
            render() hands to the reflection pass).  Use the tap instead of slicing the output yourself and the gradient it
            receives travels to the backward kernel as a separate pointer, which adds it to the upstream planes while loading
            them: no zero-filled full-size gradient, no slice copy, no image-sized add between the two autograd nodes
            (0.05 ms per 1080p view).  Same values either way."""
            unknown = [n for n in names if n not in v.taps]
            if unknown:
                raise NotImplementedError(f"this rasterizer variant has no output tap(s) {unknown}; available: {sorted(v.taps)}")
            self._taps = tuple(names)

        def markVisible(self, positions):
            """Boolean mask of the points in front of the camera's near plane (frustum test of the rasterizer)."""
            s = self.raster_settings
            with torch.no_grad():
                return v.c_module.mark_visible(positions, s.viewmatrix, s.projmatrix)

        def _collect(self, means3D, means2D, opacities, **kw):
            """The reference's argument checks and placeholder substitution; returns the tensors of Function.apply by name."""
            if (kw["shs"] is None) == (kw["colors_precomp"] is None):
                raise Exception(MSG_COLOR)
            have_sr = kw["scales"] is not None and kw["rotations"] is not None
            have_any_sr = kw["scales"] is not None or kw["rotations"] is not None
            if (not have_sr and kw["cov3D_precomp"] is None) or (have_any_sr and kw["cov3D_precomp"] is not None):
                raise Exception(MSG_COV)
            t = {"means3D": means3D, "means2D": means2D, "opacities": opacities}
            for key, value in kw.items():
                name = v.module_to_apply.get(key, key)
                t[name] = v.placeholder(name, means3D.device) if value is None and name in PLACEHOLDERS else value
            return t

        def _forward(self, means3D, means2D, opacities, **kw):
            t = self._collect(means3D, means2D, opacities, **kw)
            args = [t[name] for name in v.tensors]
            args.insert(v.settings_pos, self.raster_settings)
            ext = self._grad_sink
            if self._taps:
                ext = GradSink(ext.tensors if ext else None, ext.accumulate if ext else False, taps=self._taps)
            return rasterize_gaussians(*args, grad_sink=ext)

    PLACEHOLDERS = {"sh", "colors_precomp", "scales", "rotations", "cov3Ds_precomp", "env_scope_mask"}
    # give forward() the reference's explicit signature (keyword names and defaults are part of the API)
    params = [inspect.Parameter("self", inspect.Parameter.POSITIONAL_OR_KEYWORD)]
    params += [inspect.Parameter(n, inspect.Parameter.POSITIONAL_OR_KEYWORD) for n in ("means3D", "means2D", "opacities")]
    params += [inspect.Parameter(n, inspect.Parameter.POSITIONAL_OR_KEYWORD, default=d) for n, d in v.forward_kwargs]
    sig = inspect.Signature(params)

    def forward(*args, **kwargs):
        b = sig.bind(*args, **kwargs)
        b.apply_defaults()
        a = dict(b.arguments)
        self = a.pop("self")
        return self._forward(a.pop("means3D"), a.pop("means2D"), a.pop("opacities"), **a)
    forward.__signature__ = sig
    forward.__doc__ = "Same arguments and return tuple as the reference's GaussianRasterizer.forward."
    GaussianRasterizer.forward = forward
    GaussianRasterizer.variant = v
    return Settings, _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer
