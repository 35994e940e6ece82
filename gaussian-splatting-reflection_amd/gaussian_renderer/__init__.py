"""Caller side of the hot path: `render()`, `render_fast()` and `render_env_map()` with the arguments and output
dictionaries of the reference's gaussian_renderer/__init__.py (render :42-219, render_fast :221-325, render_env_map
:37-40), built on this package's rasterizer and three fused per-pixel HIP passes:

  deferred_reflection   shading normal -> camera ray -> reflect -> seamless cubemap lookup -> sigmoid -> lerp with the base
                        colour by the blended reflection strength (the reference: ~12 torch ops over [H,W,3],
                        :22-35,148,178-179,197-199 and utils/general_utils.py:177-197) as ONE kernel per direction
  shading_normal        the shading normal alone, for the initial stage that renders without the reflection chain
  surface_pass          depth select + pseudo-normal from depth (:151-176, utils/point_utils.py:9-37) as one kernel

There is no torch re-implementation of these chains in the product: the independently written float64 chains they are
checked against live under tests/ (tests/helpers_chain.py).
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gsr  # noqa: E402
from _gsr import check, lib, ptr, stream_ptr  # noqa: E402
from _raster_api import GradSink  # noqa: E402
import diff_surfel_rasterization as _dsr  # noqa: E402
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402


# ------------------------------------------------------------------------------------------- per-camera constants
class _Block:
    """A small device tensor derived from camera tensors, kept together with the tensors it was built from and their
    version counters.  Holding the sources keeps their storage alive, so a later camera can never be handed the same
    addresses and be mistaken for this one (the reference recomputes everything per call; this only saves the ~10 tiny
    host-side ops per step)."""

    def __init__(self, sources, extra, value):
        self.sources, self.versions, self.extra, self.value = sources, [t._version for t in sources], extra, value

    def matches(self, sources, extra):
        return (len(sources) == len(self.sources) and all(a is b for a, b in zip(sources, self.sources)) and
                all(t._version == v for t, v in zip(sources, self.versions)) and extra == self.extra)


_blocks = {}          # kind -> the most recently used _Blocks, newest last
_BLOCKS_KEPT = 64     # cameras whose constants are kept per kind: a batch of views per optimizer step (BASELINE C4: 8 views) cycles through its
                      # cameras every step, and rebuilding a block is ~10 tiny host-side ops plus a synchronous host-to-device copy (0.3-0.4 ms
                      # of idle GPU per view at C4 when only the last camera was remembered, round 2)


def _cached_block(kind, sources, extra, build):
    kept = _blocks.setdefault(kind, [])
    for k in range(len(kept) - 1, -1, -1):
        if kept[k].matches(sources, extra):
            hit = kept.pop(k)
            kept.append(hit)
            return hit.value
    value = build()
    kept.append(_Block(list(sources), extra, value))
    if len(kept) > _BLOCKS_KEPT:
        del kept[0]
    return value


def _cam_block(world_view_transform, HWK, R, T):
    """The 33 camera floats gsr_deferred_reflection_* / gsr_normal_world_* expect (layout in csrc/gsr_cubemap.hip):
    world-view rotation, K^-1, world-to-camera rotation, translation, camera centre."""
    Kb = np.asarray(HWK[2], dtype=np.float32)

    def build():
        dev = world_view_transform.device
        Kinv = torch.tensor(np.linalg.inv(Kb), dtype=torch.float32, device=dev)
        Rw = R.T.contiguous().float()
        Tf = T.float()
        centre = (-Rw.T @ Tf.unsqueeze(-1)).flatten()
        return torch.cat([world_view_transform[:3, :3].contiguous().float().reshape(-1), Kinv.reshape(-1), Rw.reshape(-1), Tf.reshape(-1),
                          centre.reshape(-1)]).contiguous()
    return _cached_block("cam", (world_view_transform, R, T), (int(HWK[0]), int(HWK[1]), Kb.tobytes()), build)


def _ray_block(view):
    """Device float[12] for gsr_surface_*: rays_d(x, y) = (x, y, 1) @ M with M = intrins^-1.T @ c2w[:3,:3].T (nine floats,
    row-major) followed by the camera centre, where intrins is the 3x3 pixel projection the reference derives from
    full_proj_transform and an (W/2, H/2)-centred NDC-to-pixel matrix (utils/point_utils.py:9-23)."""
    wvt, fpt = view.world_view_transform, view.full_proj_transform
    W, H = int(view.image_width), int(view.image_height)

    def build():
        c2w = (wvt.T).inverse()
        ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=torch.float32, device=wvt.device).T
        intrins = ((c2w.T @ fpt) @ ndc2pix)[:3, :3].T
        M = intrins.inverse().T @ c2w[:3, :3].T
        return torch.cat([M.reshape(-1), c2w[:3, 3].reshape(-1)]).float().contiguous()
    return _cached_block("ray", (wvt, fpt), (W, H), build)


# ------------------------------------------------------------------------------------------- fused pixel passes
# True: sorted-footprint path of the reflection backward (~44 bytes of scratch per pixel); False: float atomics
REFLECTION_BACKWARD_BINNED = True
# True: the forward writes the sort keys of the backward's footprint records (4 bytes per pixel), so that the backward's sort runs beside
# its pixel kernel instead of after it; False: the backward's pixel kernel writes them (what a plain C-ABI caller gets)
REFLECTION_FORWARD_KEYS = True


class _DeferredReflection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, normal_view, base_color, refl_strength, cubemap, fail_value, cam, sink):
        nv, bc, rs = normal_view.float().contiguous(), base_color.float().contiguous(), refl_strength.float().contiguous()
        cm, fv = cubemap.float().contiguous(), fail_value.float().contiguous()
        if cm.shape[1] != 3:
            raise RuntimeError("deferred_reflection: the cubemap must have 3 channels")
        H, W = nv.shape[1], nv.shape[2]
        final = torch.empty_like(bc)
        refl_color = torch.empty_like(bc)
        normal_world = torch.empty_like(nv)
        # texel-interleaved copy of the cubemap for the pixel kernels (one 16-byte gather per bilinear corner); the backward reuses it
        rgba = torch.empty(6 * cm.shape[2] * cm.shape[3] * 4, dtype=torch.float32, device=cm.device)
        # sort keys of the backward's footprint records (they depend on forward data only): written here when a backward can follow, so
        # that its sort does not have to wait for its pixel kernel
        keys = None
        if REFLECTION_BACKWARD_BINNED and REFLECTION_FORWARD_KEYS and any(ctx.needs_input_grad[:5]):
            keys = torch.empty(H * W, dtype=torch.int32, device=cm.device)
        with torch.cuda.device(nv.device):
            check(lib.gsr_deferred_reflection_forward_keys(ptr(nv), ptr(bc), ptr(rs), ptr(cam), ptr(cm), ptr(fv), cm.shape[2], W, H, ptr(final),
                                                         ptr(refl_color), ptr(normal_world), ptr(rgba), ptr(keys), stream_ptr(nv.device)),
                  "gsr_deferred_reflection_forward")
        ctx.save_for_backward(nv, bc, rs, cm, fv, cam, rgba)
        ctx.sort_keys = keys
        ctx.sink = sink
        ctx.set_materialize_grads(False)   # outputs nobody differentiates arrive as None instead of zero-filled [3,H,W] tensors
        return final, refl_color, normal_world

    @staticmethod
    def backward(ctx, g_final, g_refl_color, g_normal_world):
        nv, bc, rs, cm, fv, cam, rgba = ctx.saved_tensors
        H, W = nv.shape[1], nv.shape[2]
        g_final = torch.zeros_like(bc) if g_final is None else g_final.float().contiguous()
        g_refl_color = None if g_refl_color is None else g_refl_color.float().contiguous()
        g_normal_world = None if g_normal_world is None else g_normal_world.float().contiguous()
        g_nv, g_base, g_s = torch.empty_like(nv), torch.empty_like(bc), torch.empty_like(rs)
        # the library writes (or, accumulate mode, adds to) every element of both gradients.  With a sink they go straight
        # into caller-owned tensors and autograd gets None: no allocation, no `grad += new` pass
        sink = ctx.sink.tensors if ctx.sink is not None else {}
        accumulate = ctx.sink is not None and ctx.sink.accumulate
        g_cm, g_fail = sink.get("cubemap"), sink.get("fail")
        sunk_cm, sunk_fail = g_cm is not None, g_fail is not None
        if accumulate and not (sunk_cm and sunk_fail):
            raise ValueError("reflection grad sink: accumulate=True needs both 'cubemap' and 'fail' tensors")
        for t, like, name in ((g_cm, cm, "cubemap"), (g_fail, fv, "fail")):
            if t is not None and (tuple(t.shape) != tuple(like.shape) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != like.device):
                raise ValueError(f"reflection grad sink '{name}': expected contiguous float32 {tuple(like.shape)} on {like.device}")
        g_cm = torch.empty_like(cm) if g_cm is None else g_cm
        g_fail = torch.empty_like(fv) if g_fail is None else g_fail
        n_scratch = int(lib.gsr_deferred_reflection_scratch_floats(int(cm.shape[2]), W, H, 1 if REFLECTION_BACKWARD_BINNED else 0))
        scratch = torch.empty(n_scratch, dtype=torch.float32, device=cm.device)
        async_tail = ctx.sink is not None and ctx.sink.async_tail
        if async_tail and not (sunk_cm and sunk_fail):
            raise ValueError("reflection grad sink: async_tail=True needs both 'cubemap' and 'fail' tensors")
        with torch.cuda.device(nv.device):
            check(lib.gsr_deferred_reflection_backward_keys(ptr(nv), ptr(bc), ptr(rs), ptr(cam), ptr(cm), ptr(fv), cm.shape[2], W, H,
                                                          ptr(g_final), ptr(g_refl_color), ptr(g_normal_world), ptr(g_nv), ptr(g_base),
                                                          ptr(g_s), ptr(g_cm), ptr(g_fail), ptr(scratch), n_scratch, int(accumulate),
                                                          int(async_tail), ptr(rgba), ptr(ctx.sort_keys), 0, stream_ptr(nv.device)),
                  "gsr_deferred_reflection_backward")
        if async_tail:
            _gsr.side_hold(scratch, g_cm, g_fail, ctx.sort_keys)     # read / written on the side stream until side_join()
        return g_nv, g_base, g_s, (None if sunk_cm else g_cm), (None if sunk_fail else g_fail), None, None


def deferred_reflection(normal_view, base_color, refl_strength_map, env_map, world_view_transform, HWK, R, T, grad_sink=None,
                        accumulate=False, async_tail=False):
    """Fused pixel pass.  normal_view = allmap[2:5] (view space, un-normalised).  Returns
    (final_image[3,H,W], refl_color[3,H,W], render_normal_world[3,H,W] normalised).

    grad_sink (extension, the counterpart of GaussianRasterizer.set_grad_sink): {"cubemap": float32 [6,3,L,L],
    "fail": float32 [3]} — e.g. views of gsr_dist.FlatGrads.  The backward of THIS call then writes (accumulate=False) or
    adds (accumulate=True) the cubemap / fail-value gradient into them and returns None to autograd.

    async_tail (with a grad_sink only): the part of the backward that produces the cubemap / fail-value gradient (sort of the
    per-pixel footprint records, run combine, unpack — it feeds nothing but the sink) is enqueued on the library's side stream
    and overlaps the rasterizer backward; the per-pixel gradients autograd receives are in stream order as always.  The
    sink tensors are complete only after _gsr.side_join() — gsr_dist.FlatGrads.all_reduce / all_reduce_async / zero_ and
    gsr_train.FlatAdam.step call it; code that reads the sink itself must too."""
    if async_tail and not grad_sink:
        raise ValueError("deferred_reflection: async_tail=True needs a grad_sink (autograd would read the gradient at once)")
    cam = _cam_block(world_view_transform, HWK, R, T)
    sink = GradSink(grad_sink, accumulate, async_tail) if grad_sink else None
    return _DeferredReflection.apply(normal_view, base_color, refl_strength_map, env_map.params['Cubemap_texture'],
                                     env_map.params['Cubemap_failv'], cam, sink)


# ------------------------------------------------------------------------------------------- rasterizer + reflection in one pass
# True (round 4): render() / render_fast() run the deferred reflection's forward INSIDE the rasterizer's forward tile kernel and both
# backwards in one autograd node (rasterize_reflect below); False: rasterizer, then deferred_reflection() as two autograd nodes.  Same
# outputs (tests/test_gpu_fused.py).
FUSED_REFLECTION = True


class _RasterizeReflect(torch.autograd.Function):
    """Surfel rasterizer + deferred reflection as ONE autograd node.  Forward: gsr_surfel_forward_refl — the reflection's per-pixel code
    runs as the epilogue of the forward tile kernel (csrc/gsr_surfel.hip, csrc/gsr_refl.hpp) and the texel-interleaved cubemap copy is made
    by the per-Gaussian kernel: no pixel kernel, no interleave dispatch, no re-read of the seven planes the two passes exchanged.
    Backward: the reflection's pixel kernel and then the rasterizer's backward, back to back in one node (no autograd glue between them;
    the pixel gradients reach the tile backward as pointers).  The reflection backward is NOT folded into the tile backward: its
    texel-gradient tail (sort by texel, run combine) needs every pixel's record early so that it hides beside that kernel — as a prologue
    of the tile backward the records are complete only when it ends and the tail is exposed (built and measured in round 4).
    apply(*rasterizer tensors in diff_surfel_rasterization's order, cubemap, fail_value, cam_block, settings, raster_sink, refl_sink)
    -> (final, refl_color, normal_world, base_color, radii, allmap, refl_strength_map, gaussian_weights)."""
    V = _dsr._VARIANT
    probe = None        # tests only: a dict that receives clones of the pixel gradients the reflection backward hands the rasterizer backward

    @staticmethod
    def forward(ctx, *args):
        v = _RasterizeReflect.V
        n = len(v.tensors)
        t = dict(zip(v.tensors, args[:n]))
        cubemap, fail_value, cam, settings, raster_sink, refl_sink = args[n:]
        cm, fv = cubemap.float().contiguous(), fail_value.float().contiguous()
        if cm.dim() != 4 or cm.shape[1] != 3:
            raise RuntimeError("rasterize_reflect: the cubemap must be (6, 3, L, L)")
        want_keys = REFLECTION_BACKWARD_BINNED and REFLECTION_FORWARD_KEYS and any(ctx.needs_input_grad[:n + 2])
        # With an asynchronous tail the keys are sorted HERE, on the side stream behind the forward's tile kernel: the sort then runs beside
        # whatever follows the forward (the loss, the start of the backward) instead of racing the tile backward for CUs
        early = want_keys and refl_sink is not None and refl_sink.async_tail and bool(refl_sink.tensors)
        ret = _dsr._C.rasterize_gaussians(*v.pack_forward(t, settings), refl=dict(cam=cam, cubemap=cm, fail_value=fv, keys=want_keys, early_sort=early))
        (num_rendered, color, others, radii, geom, binning, img, refl_map, weights, final, refl_color, normal_world, rgba, keys, scratch) = ret
        if scratch is not None:
            _gsr.side_hold(scratch, keys)       # the side stream reads / writes them from now on (until side_join)
        ctx.raster_settings, ctx.num_rendered, ctx.n_tensors = settings, num_rendered, n
        ctx.raster_sink, ctx.refl_sink, ctx.sort_keys, ctx.scratch = raster_sink, refl_sink, keys, scratch
        ctx.save_for_backward(*[t[k] for k in v.saved], radii, geom, binning, img, color, others, refl_map, cm, fv, cam, rgba)
        ctx.mark_non_differentiable(radii, weights)
        ctx.set_materialize_grads(False)
        return final, refl_color, normal_world, color, radii, others, refl_map, weights

    @staticmethod
    def backward(ctx, g_final, g_refl_color, g_normal_world, g_color, _g_radii, g_others, g_refl_map, _g_weights):
        v = _RasterizeReflect.V
        kept = ctx.saved_tensors
        ns = len(v.saved)
        saved = dict(zip(v.saved, kept[:ns]))
        radii, geom, binning, img, color, others, refl_map, cm, fv, cam, rgba = kept[ns:]
        settings = ctx.raster_settings
        cont = lambda g: None if g is None else g.float().contiguous()
        H, W = color.shape[1], color.shape[2]
        dev = color.device
        # ---- 1. reflection backward (pixel kernel on this stream, texel-gradient tail beside what follows)
        g_final = torch.zeros_like(color) if g_final is None else cont(g_final)
        g_refl_color, g_normal_world = cont(g_refl_color), cont(g_normal_world)
        rs = ctx.refl_sink.tensors if ctx.refl_sink is not None else {}
        acc_refl = ctx.refl_sink is not None and ctx.refl_sink.accumulate
        async_tail = ctx.refl_sink is not None and ctx.refl_sink.async_tail
        g_cm, g_fail = rs.get("cubemap"), rs.get("fail")
        sunk_cm, sunk_fail = g_cm is not None, g_fail is not None
        if (acc_refl or async_tail) and not (sunk_cm and sunk_fail):
            raise ValueError("reflection grad sink: accumulate / async_tail need both 'cubemap' and 'fail' tensors")
        for tt, like, name in ((g_cm, cm, "cubemap"), (g_fail, fv, "fail")):
            if tt is not None and (tuple(tt.shape) != tuple(like.shape) or tt.dtype != torch.float32 or not tt.is_contiguous() or tt.device != like.device):
                raise ValueError(f"reflection grad sink '{name}': expected contiguous float32 {tuple(like.shape)} on {like.device}")
        g_cm = torch.empty_like(cm) if g_cm is None else g_cm
        g_fail = torch.empty_like(fv) if g_fail is None else g_fail
        keys_sorted = ctx.scratch is not None
        if keys_sorted:
            scratch, n_scratch = ctx.scratch, int(ctx.scratch.numel())
        else:
            n_scratch = int(lib.gsr_deferred_reflection_scratch_floats(int(cm.shape[2]), W, H, 1 if REFLECTION_BACKWARD_BINNED else 0))
            scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
        g_nv, g_base, g_s = torch.empty((3, H, W), dtype=torch.float32, device=dev), torch.empty_like(color), torch.empty_like(refl_map)
        nv = others[2:5]                          # (contiguous: planes 2..4 of the [8,H,W] output)
        with torch.cuda.device(dev):
            check(lib.gsr_deferred_reflection_backward_keys(ptr(nv), ptr(color), ptr(refl_map), ptr(cam), ptr(cm), ptr(fv), cm.shape[2], W, H,
                                                            ptr(g_final), ptr(g_refl_color), ptr(g_normal_world), ptr(g_nv), ptr(g_base), ptr(g_s),
                                                            ptr(g_cm), ptr(g_fail), ptr(scratch), n_scratch, int(acc_refl), int(async_tail),
                                                            ptr(rgba), ptr(ctx.sort_keys), int(keys_sorted), stream_ptr(dev)),
                  "gsr_deferred_reflection_backward")
        if async_tail:
            _gsr.side_hold(scratch, g_cm, g_fail, ctx.sort_keys)     # read / written on the side stream until side_join()
        # ---- 2. rasterizer backward: what reached its colour / reflection-strength outputs through the final image, plus the direct gradients
        # of those outputs (rare: nothing in the reference's losses reads them); the normal gradient travels as the tap's pointer
        g_color, g_refl_map, g_others = cont(g_color), cont(g_refl_map), cont(g_others)
        if g_color is not None:
            g_base.add_(g_color)
        if g_refl_map is not None:
            g_s.add_(g_refl_map)
        if g_others is None:
            g_others = torch.zeros_like(others)
        if _RasterizeReflect.probe is not None:
            _RasterizeReflect.probe.update(g_normal_view=g_nv.clone(), g_base=g_base.clone(), g_strength=g_s.clone())
        c_args = v.pack_backward(saved, settings, [g_base, None, g_others, g_s, None], ctx.num_rendered, (geom, binning, img), radii)
        kw = {"extra_normal_grad": g_nv}
        sink = ctx.raster_sink.tensors if (ctx.raster_sink is not None and ctx.raster_sink.tensors) else {}
        if sink:
            kw.update(grad_sink=sink, accumulate=ctx.raster_sink.accumulate)
        unused = tuple(key for name, key in v.skippable.items() if saved.get(name) is None or saved[name].numel() == 0)
        if unused:
            kw["unused"] = unused
        g = v.grads_of(_dsr._C.rasterize_gaussians_backward(*c_args, **kw))
        out = []
        for name in v.tensors:
            grad = g.get(name)
            if name in v.sinkable and v.sinkable[name] in sink:
                grad = None            # already written (or added) into the caller's sink tensor by the backward kernel
            elif name in v.optional_grads and (saved.get(name) is None or saved[name].numel() == 0):
                grad = None
            out.append(grad)
        return tuple(out) + ((None if sunk_cm else g_cm), (None if sunk_fail else g_fail), None, None, None, None)


def rasterize_reflect(rasterizer, env_map, world_view_transform, HWK, R, T, means3D, means2D, opacities, shs=None, colors_precomp=None,
                      refl_strengths=None, scales=None, rotations=None, cov3D_precomp=None, env_scope_mask=None, refl_grad_sink=None,
                      accumulate=False, async_tail=False):
    """Extension: `rasterizer(...)` followed by `deferred_reflection(allmap[2:5], base, refl_map, env_map, ...)` as ONE pass over the
    pixels (see _RasterizeReflect).  `rasterizer`: a diff_surfel_rasterization.GaussianRasterizer (its settings and its gradient sink —
    set_grad_sink — apply); the keyword arguments are those of its forward(); refl_grad_sink / accumulate / async_tail those of
    deferred_reflection().  Returns (final_image, refl_color, rend_normal_world, base_color, radii, allmap, refl_strength_map,
    gaussian_weights)."""
    if async_tail and not refl_grad_sink:
        raise ValueError("rasterize_reflect: async_tail=True needs a refl_grad_sink (autograd would read the gradient at once)")
    t = rasterizer._collect(means3D, means2D, opacities, shs=shs, colors_precomp=colors_precomp, refl_strengths=refl_strengths, scales=scales,
                            rotations=rotations, cov3D_precomp=cov3D_precomp, env_scope_mask=env_scope_mask)
    cam = _cam_block(world_view_transform, HWK, R, T)
    rsink = GradSink(refl_grad_sink, accumulate, async_tail) if refl_grad_sink else None
    return _RasterizeReflect.apply(*[t[name] for name in _dsr._VARIANT.tensors], env_map.params['Cubemap_texture'], env_map.params['Cubemap_failv'], cam,
                                   rasterizer.raster_settings, rasterizer._grad_sink, rsink)


class _ShadingNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, normal_view, cam):
        nv = normal_view.float().contiguous()
        out = torch.empty_like(nv)
        with torch.cuda.device(nv.device):
            check(lib.gsr_normal_world_forward(ptr(nv), ptr(cam), nv.shape[2], nv.shape[1], ptr(out), stream_ptr(nv.device)),
                  "gsr_normal_world_forward")
        ctx.save_for_backward(nv, cam)
        return out

    @staticmethod
    def backward(ctx, g):
        nv, cam = ctx.saved_tensors
        g_nv = torch.empty_like(nv)
        g = g.float().contiguous()
        with torch.cuda.device(nv.device):
            check(lib.gsr_normal_world_backward(ptr(nv), ptr(cam), nv.shape[2], nv.shape[1], ptr(g), ptr(g_nv), stream_ptr(nv.device)),
                  "gsr_normal_world_backward")
        return g_nv, None


def shading_normal(normal_view, world_view_transform, HWK, R, T):
    """rend_normal of the reference: the blended view-space normal allmap[2:5] rotated to world space and divided by
    (its length + 1e-6), as [3,H,W]."""
    return _ShadingNormal.apply(normal_view, _cam_block(world_view_transform, HWK, R, T))


class _SurfacePass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, raymat, depth_ratio):
        am = allmap.float().contiguous()
        if am.dim() != 3 or am.shape[0] != 8:
            raise RuntimeError("surface_pass: allmap must be (8,H,W)")
        H, W = am.shape[1], am.shape[2]
        sd = torch.empty((1, H, W), dtype=torch.float32, device=am.device)
        sn = torch.empty((3, H, W), dtype=torch.float32, device=am.device)
        with torch.cuda.device(am.device):
            check(lib.gsr_surface_forward(ptr(am), ptr(raymat), float(depth_ratio), H, W, ptr(sd), ptr(sn), stream_ptr(am.device)),
                  "gsr_surface_forward")
        ctx.save_for_backward(am, raymat, sd)
        ctx.depth_ratio = float(depth_ratio)
        ctx.set_materialize_grads(False)   # the backward takes NULL for an output without gradient
        return sd, sn

    @staticmethod
    def backward(ctx, g_sd, g_sn):
        am, raymat, sd = ctx.saved_tensors
        H, W = am.shape[1], am.shape[2]
        g_am = torch.empty_like(am)
        g_sd = None if g_sd is None else g_sd.float().contiguous()
        g_sn = None if g_sn is None else g_sn.float().contiguous()
        with torch.cuda.device(am.device):
            check(lib.gsr_surface_backward(ptr(am), ptr(raymat), ctx.depth_ratio, H, W, ptr(sd), ptr(g_sd), ptr(g_sn), ptr(g_am),
                                           stream_ptr(am.device)), "gsr_surface_backward")
        return g_am, None, None


def surface_pass(allmap, view, depth_ratio):
    """Returns (surf_depth[1,H,W], surf_normal[3,H,W]): the blend of expected and median depth selected by `depth_ratio`, and
    the pseudo-normal of that depth map (cross product of central differences of the unprojected points, zero on the
    border) scaled by the detached alpha — from the rasterizer's allmap in one kernel.  Gradients flow to allmap[0],
    allmap[1] and allmap[5]."""
    return _SurfacePass.apply(allmap, _ray_block(view), depth_ratio)


# ------------------------------------------------------------------------------------------- the reference's entry points
def _settings(viewpoint_camera, pc, bg_color, scaling_modifier):
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=False)


def _precomputed_transmats(viewpoint_camera, pc, scaling_modifier, dev):
    """pipe.compute_cov3D_python: the splat-to-pixel homographies formed on the host side and handed to the rasterizer as
    cov3D_precomp (P,9), column-major: splat2world[:, (u, v, origin)] @ world2pix[:, (x, y, w)]."""
    W, H = viewpoint_camera.image_width, viewpoint_camera.image_height
    near, far = viewpoint_camera.znear, viewpoint_camera.zfar
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near], [0, 0, 0, 1]],
                           dtype=torch.float32, device=dev).T
    world2pix = viewpoint_camera.full_proj_transform @ ndc2pix
    keep = [0, 1, 3]
    return (pc.get_covariance(scaling_modifier)[:, keep] @ world2pix[:, keep]).permute(0, 2, 1).reshape(-1, 9)


def _sinks(pipe):
    """Optional gradient sinks carried by the pipeline object (extension): pipe.gsr_grad_sink for the rasterizer,
    pipe.gsr_reflection_grad_sink for the fused reflection op, pipe.gsr_accumulate for accumulate mode,
    pipe.gsr_async_reflection_tail for deferred_reflection's async_tail."""
    return (getattr(pipe, "gsr_grad_sink", None), getattr(pipe, "gsr_reflection_grad_sink", None), bool(getattr(pipe, "gsr_accumulate", False)),
            bool(getattr(pipe, "gsr_async_reflection_tail", False)))


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, initial_stage=False,
           env_scope_center=[0.0, 0.0, 0.0], env_scope_radius=0.0):
    """Same contract as the reference's render().  Background tensor (bg_color) must be on GPU!"""
    xyz = pc.get_xyz
    dev = xyz.device
    # zero tensor whose only purpose is to receive the screen-space gradient (densification statistic)
    means2D = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=dev) + 0
    try:
        means2D.retain_grad()
    except Exception:
        pass
    raster_sink, refl_sink, accumulate, async_tail = _sinks(pipe)
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, scaling_modifier))
    rasterizer.set_grad_sink(raster_sink, accumulate)
    # allmap[2:5] as an output tap: the gradient of its consumer below reaches the tile backward as a pointer of its own instead
    # of being summed into a full-size allmap gradient by autograd (same values; GaussianRasterizer.set_output_taps)
    rasterizer.set_output_taps(("normal_view",))
    if env_scope_radius > 0.0:
        centre = torch.tensor([float(c) for c in env_scope_center], device=dev)
        env_scope_mask = ((xyz - centre[None]) ** 2).sum(dim=-1) < env_scope_radius ** 2
    else:
        env_scope_mask = torch.ones_like(xyz, device=dev) == 1.0     # (P,3) all-true; the kernel reads entry [id]
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = _precomputed_transmats(viewpoint_camera, pc, scaling_modifier, dev)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    shs = pc.get_features if override_color is None else None       # SH evaluation always happens in the rasterizer
    v = viewpoint_camera
    fused = FUSED_REFLECTION and not initial_stage
    if fused:
        rasterizer.set_output_taps(())
        (final_image, refl_color, rend_normal, base_color, radii, allmap, refl_strength_map, gaussian_weights) = rasterize_reflect(
            rasterizer, pc.get_envmap, v.world_view_transform, v.HWK, v.R, v.T, means3D=xyz, means2D=means2D, shs=shs,
            colors_precomp=override_color, refl_strengths=pc.get_refl, opacities=pc.get_opacity, scales=scales, rotations=rotations,
            cov3D_precomp=cov3D_precomp, env_scope_mask=env_scope_mask, refl_grad_sink=refl_sink, accumulate=accumulate,
            async_tail=async_tail and bool(refl_sink))
    else:
        base_color, radii, allmap, refl_strength_map, gaussian_weights, normal_view = rasterizer(
            means3D=xyz, means2D=means2D, shs=shs, colors_precomp=override_color, refl_strengths=pc.get_refl, opacities=pc.get_opacity,
            scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, env_scope_mask=env_scope_mask)

    surf_depth, surf_normal = surface_pass(allmap, viewpoint_camera, pipe.depth_ratio)
    out = {"viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii, "rend_alpha": allmap[1:2],
           "rend_dist": allmap[6:7], "surf_depth": surf_depth, "surf_normal": surf_normal, "gaussian_weights": gaussian_weights,
           "env_scope_mask": allmap[7:8]}
    if initial_stage:
        out["rend_normal"] = shading_normal(normal_view, v.world_view_transform, v.HWK, v.R, v.T)
        out["render"] = base_color
        return out
    if not fused:
        final_image, refl_color, rend_normal = deferred_reflection(normal_view, base_color, refl_strength_map, pc.get_envmap,
                                                                   v.world_view_transform, v.HWK, v.R, v.T, grad_sink=refl_sink,
                                                                   accumulate=accumulate, async_tail=async_tail and bool(refl_sink))
    out.update({"rend_normal": rend_normal, "render": final_image, "refl_strength_map": refl_strength_map, "refl_color_map": refl_color,
                "base_color_map": base_color})
    return out


def render_fast(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, initial_stage=False):
    """Inference path (the reference's render_fast, used by its eval_fps.py / render.py): all-true env-scope mask, no depth
    or normal-consistency outputs."""
    xyz = pc.get_xyz
    means2D = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, scaling_modifier))
    v = viewpoint_camera
    if FUSED_REFLECTION and not initial_stage:
        final_image, refl_color, rend_normal, base_color, _, allmap, refl_strength_map, _ = rasterize_reflect(
            rasterizer, pc.get_envmap, v.world_view_transform, v.HWK, v.R, v.T, means3D=xyz, means2D=means2D, shs=pc.get_features,
            refl_strengths=pc.get_refl, opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation,
            env_scope_mask=torch.ones_like(xyz).bool())
        return {"render": final_image, "rend_alpha": allmap[1:2], "rend_normal": rend_normal, "refl_strength_map": refl_strength_map,
                "refl_color_map": refl_color, "base_color_map": base_color}
    base_color, _, allmap, refl_strength_map, _ = rasterizer(
        means3D=xyz, means2D=means2D, shs=pc.get_features, colors_precomp=None, refl_strengths=pc.get_refl, opacities=pc.get_opacity,
        scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=None, env_scope_mask=torch.ones_like(xyz).bool())
    if initial_stage:
        return {"render": base_color, "rend_alpha": allmap[1:2], "refl_strength_map": refl_strength_map,
                "rend_normal": shading_normal(allmap[2:5], v.world_view_transform, v.HWK, v.R, v.T)}
    final_image, refl_color, rend_normal = deferred_reflection(allmap[2:5], base_color, refl_strength_map, pc.get_envmap,
                                                               v.world_view_transform, v.HWK, v.R, v.T)
    return {"render": final_image, "rend_alpha": allmap[1:2], "rend_normal": rend_normal, "refl_strength_map": refl_strength_map,
            "refl_color_map": refl_color, "base_color_map": base_color}


def _panorama_dirs(H, W, device):
    """The two latitude-longitude direction grids of the reference's environment-map visualisation: grid 1 is z-up with the
    azimuth running over [-pi, pi] and the polar angle over [0, pi], end points included; grid 2 is y-up, sampled at pixel
    centres, looking down -z at its centre column."""
    az = torch.linspace(-math.pi, math.pi, W, dtype=torch.float32, device=device)[None, :]
    po = torch.linspace(0.0, math.pi, H, dtype=torch.float32, device=device)[:, None]
    grid1 = torch.stack([torch.sin(po) * torch.cos(az), torch.sin(po) * torch.sin(az), torch.cos(po).expand(H, W)], dim=-1)
    th = math.pi * torch.linspace(1.0 / H, 1.0 - 1.0 / H, H, dtype=torch.float32, device=device)[:, None]
    ph = math.pi * torch.linspace(-1.0 + 1.0 / W, 1.0 - 1.0 / W, W, dtype=torch.float32, device=device)[None, :]
    grid2 = torch.stack([torch.sin(th) * torch.sin(ph), torch.cos(th).expand(H, W), -torch.sin(th) * torch.cos(ph)], dim=-1)
    return grid1, grid2


def render_env_map(pc, height=512, width=1024):
    """The reference's render_env_map(): two [3,H,W] panoramas of sigmoid(environment cubemap), keys env_cood1 / env_cood2."""
    env = pc.get_envmap
    dev = env.params['Cubemap_texture'].device
    out = {}
    for name, dirs in zip(("env_cood1", "env_cood2"), _panorama_dirs(height, width, dev)):
        rgb = torch.sigmoid(env(dirs.reshape(-1, 3)))
        out[name] = rgb.reshape(height, width, 3).permute(2, 0, 1)
    return out
