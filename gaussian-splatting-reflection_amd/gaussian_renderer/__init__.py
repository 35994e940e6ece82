"""Renderer glue above the rasterizer: the caller side of the hot path, mirroring
gaussian_renderer/__init__.py of the reference (render(), render_fast(), get_refl_color(), reflection(),
sample_cubemap_color()) with the same arguments and output dictionaries.

The deferred reflection chain of the reference (normal rotate + normalise, camera-ray generation, reflect,
cubemap lookup, sigmoid, lerp: ~12 torch ops over [H,W,3], gaussian_renderer/__init__.py:22-35,148,178-179,
197-199 and utils/general_utils.py:177-197) runs here as ONE fused HIP kernel per direction
(`deferred_reflection`); the un-fused composition through CubemapEncoder stays available
(`pipe.fused_reflection = False`) and both are parity-tested against each other.
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gsr  # noqa: E402
from _gsr import check, lib, ptr, stream_ptr  # noqa: E402
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402


def reflection(rayd, normal):
    refl = rayd - 2 * normal * torch.sum(rayd * normal, dim=-1, keepdim=True)
    return refl


def sample_cubemap_color(rays_d, env_map):
    H, W = rays_d.shape[:2]
    outcolor = torch.sigmoid(env_map(rays_d.reshape(-1, 3)))
    outcolor = outcolor.reshape(H, W, 3).permute(2, 0, 1)
    return outcolor


_pixel_camera = {}


def sample_camera_rays(HWK, R, T):
    """utils/general_utils.py:177-197 (R is stored transposed in 3DGS cameras)."""
    H, W, K = HWK
    R = R.T
    key = (int(H), int(W), tuple(np.asarray(K, dtype=np.float32).reshape(-1).tolist()), str(R.device))
    pc = _pixel_camera.get(key)
    if pc is None:
        K = np.asarray(K).astype(np.float32)
        i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
        xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
        pc = torch.tensor(np.dot(xy1, np.linalg.inv(K).T)).to(R.device)
        _pixel_camera.clear()
        _pixel_camera[key] = pc
    rays_o = (-R.T @ T.unsqueeze(-1)).flatten()
    pixel_world = (pc - T[None, None]).reshape(-1, 3) @ R
    rays_d = pixel_world - rays_o[None]
    rays_d = rays_d / torch.norm(rays_d, dim=1, keepdim=True)
    return rays_d.reshape(H, W, 3)


def get_refl_color(envmap, HWK, R, T, normal_map):  # RT W2C
    rays_d = sample_camera_rays(HWK, R, T)
    rays_d = reflection(rays_d, normal_map)
    return sample_cubemap_color(rays_d, envmap)


_cam_cache = {}


def _cam_block(world_view_transform, HWK, R, T):
    """Packs the 33 camera floats gsr_deferred_reflection_* expects (see csrc/gsr_cubemap.hip).  Cached per camera
    (keyed on the tensors' storage and version counters) so that a training loop does not rebuild it every step."""
    key = (world_view_transform.data_ptr(), world_view_transform._version, R.data_ptr(), R._version, T.data_ptr(), T._version,
           int(HWK[0]), int(HWK[1]), np.asarray(HWK[2], dtype=np.float32).tobytes())
    hit = _cam_cache.get(key)
    if hit is not None:
        return hit
    dev = world_view_transform.device
    K = np.asarray(HWK[2]).astype(np.float32)
    Kinv = torch.tensor(np.linalg.inv(K), dtype=torch.float32, device=dev)
    Rw = R.T.contiguous().float()
    Tf = T.float()
    rays_o = (-Rw.T @ Tf.unsqueeze(-1)).flatten()
    cam = torch.cat([world_view_transform[:3, :3].contiguous().float().reshape(-1), Kinv.reshape(-1), Rw.reshape(-1), Tf.reshape(-1),
                     rays_o.reshape(-1)]).contiguous()
    if len(_cam_cache) > 256:
        _cam_cache.clear()
    _cam_cache[key] = cam
    return cam


# True: sort-and-accumulate-in-LDS path of the reflection backward (needs ~84 bytes of scratch per pixel); False: float atomics
REFLECTION_BACKWARD_BINNED = True


class _DeferredReflection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, normal_view, base_color, refl_strength, cubemap, fail_value, cam):
        nv, bc, rs = normal_view.float().contiguous(), base_color.float().contiguous(), refl_strength.float().contiguous()
        cm, fv = cubemap.float().contiguous(), fail_value.float().contiguous()
        if cm.shape[1] != 3:
            raise RuntimeError("deferred_reflection: the cubemap must have 3 channels")
        H, W = nv.shape[1], nv.shape[2]
        final = torch.empty_like(bc)
        refl_color = torch.empty_like(bc)
        normal_world = torch.empty_like(nv)
        with torch.cuda.device(nv.device):
            check(lib.gsr_deferred_reflection_forward(ptr(nv), ptr(bc), ptr(rs), ptr(cam), ptr(cm), ptr(fv), cm.shape[2], W, H, ptr(final),
                                                      ptr(refl_color), ptr(normal_world), stream_ptr(nv.device)),
                  "gsr_deferred_reflection_forward")
        ctx.save_for_backward(nv, bc, rs, cm, fv, cam)
        ctx.set_materialize_grads(False)   # outputs nobody differentiates arrive as None instead of zero-filled [3,H,W] tensors
        return final, refl_color, normal_world

    @staticmethod
    def backward(ctx, g_final, g_refl_color, g_normal_world):
        nv, bc, rs, cm, fv, cam = ctx.saved_tensors
        H, W = nv.shape[1], nv.shape[2]
        g_final = torch.zeros_like(bc) if g_final is None else g_final.float().contiguous()
        g_refl_color = None if g_refl_color is None else g_refl_color.float().contiguous()
        g_normal_world = None if g_normal_world is None else g_normal_world.float().contiguous()
        g_nv, g_base, g_s = torch.empty_like(nv), torch.empty_like(bc), torch.empty_like(rs)
        # the library writes every element of both gradients.  With a gradient sink (set_reflection_grad_sink) they go
        # straight into caller-owned tensors and autograd gets None: no allocation, no `grad += new` pass
        sink = reflection_grad_sink or {}
        g_cm, g_fail = sink.get("cubemap"), sink.get("fail")
        sunk_cm, sunk_fail = g_cm is not None, g_fail is not None
        for t, like, name in ((g_cm, cm, "cubemap"), (g_fail, fv, "fail")):
            if t is not None and (tuple(t.shape) != tuple(like.shape) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != like.device):
                raise ValueError(f"reflection grad sink '{name}': expected contiguous float32 {tuple(like.shape)} on {like.device}")
        g_cm = torch.empty_like(cm) if g_cm is None else g_cm
        g_fail = torch.empty_like(fv) if g_fail is None else g_fail
        n_scratch = int(lib.gsr_deferred_reflection_scratch_floats(int(cm.shape[2]), W, H, 1 if REFLECTION_BACKWARD_BINNED else 0))
        scratch = torch.empty(n_scratch, dtype=torch.float32, device=cm.device)
        with torch.cuda.device(nv.device):
            check(lib.gsr_deferred_reflection_backward(ptr(nv), ptr(bc), ptr(rs), ptr(cam), ptr(cm), ptr(fv), cm.shape[2], W, H, ptr(g_final),
                                                       ptr(g_refl_color), ptr(g_normal_world), ptr(g_nv), ptr(g_base), ptr(g_s), ptr(g_cm),
                                                       ptr(g_fail), ptr(scratch), n_scratch, stream_ptr(nv.device)), "gsr_deferred_reflection_backward")
        return g_nv, g_base, g_s, (None if sunk_cm else g_cm), (None if sunk_fail else g_fail), None


# Optional gradient sink of the fused reflection op (not in the reference; the counterpart of GaussianRasterizer.set_grad_sink):
# {"cubemap": float32 [6,3,L,L], "fail": float32 [3]} — e.g. views of gsr_dist.FlatGrads.  While set, the backward writes
# THIS backward's cubemap / fail-value gradient into them (overwritten, not summed) and returns None to autograd.
reflection_grad_sink = None


def set_reflection_grad_sink(sink):
    global reflection_grad_sink
    reflection_grad_sink = dict(sink) if sink else None


def deferred_reflection(normal_view, base_color, refl_strength_map, env_map, world_view_transform, HWK, R, T):
    """Fused pixel pass.  normal_view = allmap[2:5] (view space, un-normalised).  Returns
    (final_image[3,H,W], refl_color[3,H,W], render_normal_world[3,H,W] normalised)."""
    cam = _cam_block(world_view_transform, HWK, R, T)
    return _DeferredReflection.apply(normal_view, base_color, refl_strength_map, env_map.params['Cubemap_texture'],
                                     env_map.params['Cubemap_failv'], cam)


def _ray_block(view):
    """Device float[12] for gsr_surface_*: rays_d(x, y) = (x, y, 1) @ M (M = intrins^-1.T @ c2w[:3,:3].T, nine floats
    row-major) and rays_o, built with the same tensor ops as utils/point_utils.py:9-23 and cached per camera."""
    wvt, fpt = view.world_view_transform, view.full_proj_transform
    W, H = int(view.image_width), int(view.image_height)
    key = ("ray", wvt.data_ptr(), wvt._version, fpt.data_ptr(), fpt._version, W, H)
    hit = _cam_cache.get(key)
    if hit is not None:
        return hit
    dev = wvt.device
    c2w = (wvt.T).inverse()
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W) / 2], [0, H / 2, 0, (H) / 2], [0, 0, 0, 1]]).float().to(dev).T
    intrins = ((c2w.T @ fpt) @ ndc2pix)[:3, :3].T
    M = intrins.inverse().T @ c2w[:3, :3].T
    blk = torch.cat([M.reshape(-1), c2w[:3, 3].reshape(-1)]).float().contiguous()
    if len(_cam_cache) > 256:
        _cam_cache.clear()
    _cam_cache[key] = blk
    return blk


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
    """Fused form of gaussian_renderer/__init__.py:151-176 of the reference: returns (surf_depth[1,H,W],
    surf_normal[3,H,W]) = (expected/median depth blend, depth_to_normal(surf_depth) * alpha.detach()) from the
    rasterizer's allmap in one kernel; gradients flow to allmap[0], allmap[1] and allmap[5]."""
    return _SurfacePass.apply(allmap, _ray_block(view), depth_ratio)


def depths_to_points(view, depthmap):
    """utils/point_utils.py:9-24"""
    dev = depthmap.device
    c2w = (view.world_view_transform.T).inverse()
    W, H = view.image_width, view.image_height
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W) / 2], [0, H / 2, 0, (H) / 2], [0, 0, 0, 1]]).float().to(dev).T
    projection_matrix = c2w.T @ view.full_proj_transform
    intrins = (projection_matrix @ ndc2pix)[:3, :3].T
    grid_x, grid_y = torch.meshgrid(torch.arange(W, device=dev).float(), torch.arange(H, device=dev).float(), indexing='xy')
    points = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)], dim=-1).reshape(-1, 3)
    rays_d = points @ intrins.inverse().T @ c2w[:3, :3].T
    rays_o = c2w[:3, 3]
    return depthmap.reshape(-1, 1) * rays_d + rays_o


def depth_to_normal(view, depth):
    """utils/point_utils.py:26-37"""
    points = depths_to_points(view, depth).reshape(*depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = torch.cat([points[2:, 1:-1] - points[:-2, 1:-1]], dim=0)
    dy = torch.cat([points[1:-1, 2:] - points[1:-1, :-2]], dim=1)
    normal_map = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    output[1:-1, 1:-1, :] = normal_map
    return output


def _settings(viewpoint_camera, pc, bg_color, scaling_modifier):
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width), tanfovx=tanfovx, tanfovy=tanfovy,
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=False)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, initial_stage=False,
           env_scope_center=[0.0, 0.0, 0.0], env_scope_radius=0.0):
    """gaussian_renderer/__init__.py:42-219 of the reference.  Background tensor (bg_color) must be on GPU!"""
    xyz = pc.get_xyz
    dev = xyz.device
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=dev) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, scaling_modifier))
    means3D = xyz
    if env_scope_radius > 0.0:
        center = torch.tensor([float(c) for c in env_scope_center], device=dev)
        env_scope_mask = torch.sum((xyz - center[None]) ** 2, dim=-1) < env_scope_radius ** 2
    else:
        env_scope_mask = torch.ones_like(xyz, device=dev) == 1.0
    means2D = screenspace_points
    opacity = pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        splat2world = pc.get_covariance(scaling_modifier)
        W, H = viewpoint_camera.image_width, viewpoint_camera.image_height
        near, far = viewpoint_camera.znear, viewpoint_camera.zfar
        ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near], [0, 0, 0, 1]]).float().to(dev).T
        world2pix = viewpoint_camera.full_proj_transform @ ndc2pix
        cov3D_precomp = (splat2world[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)  # column major
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        shs = pc.get_features  # convert_SHs_python is force-disabled in the reference (:113)
    else:
        colors_precomp = override_color
    refl_strengths = pc.get_refl

    base_color, radii, allmap, refl_strength_map, gaussian_weights = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, refl_strengths=refl_strengths, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, env_scope_mask=env_scope_mask)

    render_alpha = allmap[1:2]
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = allmap[0:1] / torch.clamp(render_alpha, min=1e-3)
    render_depth_expected = torch.nan_to_num(render_depth_expected, 0, 0)
    render_dist = allmap[6:7]
    mask = allmap[7:8]
    if getattr(pipe, "fused_surface", True):
        surf_depth, surf_normal = surface_pass(allmap, viewpoint_camera, pipe.depth_ratio)
    else:   # the reference's op chain (:151-176), kept for comparison
        surf_depth = render_depth_expected * (1 - pipe.depth_ratio) + (pipe.depth_ratio) * render_depth_median
        surf_normal = depth_to_normal(viewpoint_camera, surf_depth).permute(2, 0, 1)
        surf_normal = surf_normal * (render_alpha).detach()

    fused = getattr(pipe, "fused_reflection", True) and not initial_stage
    if fused:
        final_image, refl_color, rend_normal = deferred_reflection(allmap[2:5], base_color, refl_strength_map, pc.get_envmap,
                                                                   viewpoint_camera.world_view_transform, viewpoint_camera.HWK,
                                                                   viewpoint_camera.R, viewpoint_camera.T)
    else:
        render_normal = (allmap[2:5].permute(1, 2, 0) @ (viewpoint_camera.world_view_transform[:3, :3].T))
        render_normal = render_normal / (torch.norm(render_normal, dim=-1, keepdim=True) + 1e-6)
        rend_normal = render_normal.permute(2, 0, 1)
        if not initial_stage:
            refl_color = get_refl_color(pc.get_envmap, viewpoint_camera.HWK, viewpoint_camera.R, viewpoint_camera.T, render_normal)
            final_image = (1 - refl_strength_map) * base_color + refl_strength_map * refl_color

    out = {"viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii, 'rend_alpha': render_alpha,
           'rend_normal': rend_normal, 'rend_dist': render_dist, 'surf_depth': surf_depth, 'surf_normal': surf_normal,
           "gaussian_weights": gaussian_weights, 'env_scope_mask': mask}
    if initial_stage:
        out["render"] = base_color
    else:
        out.update({"render": final_image, "refl_strength_map": refl_strength_map, "refl_color_map": refl_color,
                    "base_color_map": base_color})
    return out


def render_fast(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, initial_stage=False):
    """gaussian_renderer/__init__.py:221-325 of the reference (inference path used by eval_fps.py / render.py)."""
    xyz = pc.get_xyz
    dev = xyz.device
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=dev) + 0
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, scaling_modifier))
    env_scope_mask = torch.ones_like(xyz, device=dev).bool()
    base_color, _, allmap, refl_strength_map, _ = rasterizer(
        means3D=xyz, means2D=screenspace_points, shs=pc.get_features, colors_precomp=None, refl_strengths=pc.get_refl,
        opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=None, env_scope_mask=env_scope_mask)
    render_alpha = allmap[1:2]
    if initial_stage or not getattr(pipe, "fused_reflection", True):
        render_normal = (allmap[2:5].permute(1, 2, 0) @ (viewpoint_camera.world_view_transform[:3, :3].T))
        render_normal = render_normal / (torch.norm(render_normal, dim=-1, keepdim=True) + 1e-6)
        if initial_stage:
            return {"render": base_color, 'rend_alpha': render_alpha, "rend_normal": render_normal.permute(2, 0, 1),
                    "refl_strength_map": refl_strength_map}
        refl_color = get_refl_color(pc.get_envmap, viewpoint_camera.HWK, viewpoint_camera.R, viewpoint_camera.T, render_normal)
        final_image = (1 - refl_strength_map) * base_color + refl_strength_map * refl_color
        rend_normal = render_normal.permute(2, 0, 1)
    else:
        final_image, refl_color, rend_normal = deferred_reflection(allmap[2:5], base_color, refl_strength_map, pc.get_envmap,
                                                                   viewpoint_camera.world_view_transform, viewpoint_camera.HWK,
                                                                   viewpoint_camera.R, viewpoint_camera.T)
    return {"render": final_image, 'rend_alpha': render_alpha, 'rend_normal': rend_normal, "refl_strength_map": refl_strength_map,
            "refl_color_map": refl_color, "base_color_map": base_color}
