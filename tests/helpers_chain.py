"""Checker-side torch chains (test infrastructure only; nothing under gaussian-splatting-reflection_amd/ imports this).

Plain torch restatements of what the fused HIP pixel passes compute, written from the maths of SURVEY.md Appendix A and
8(a) ("Cubemap + deferred reflection", F2) so that the kernels can be compared with differentiable float64 code:
  shading_normal_chain   n = (N_view rotated to world) / (|.| + 1e-6)            gaussian_renderer/__init__.py:148,178-179
  view_rays_chain        d = normalize((K^-1 [x,y,1] - T) Rw - o)                utils/general_utils.py:177-197
  reflect_chain          r = d - 2 n (d . n)                                      gaussian_renderer/__init__.py:22-24
  surface_chain          depth select + pseudo-normal of the depth map           gaussian_renderer/__init__.py:151-176,
                                                                                  utils/point_utils.py:9-37
Every function works in the dtype / on the device of its tensor arguments.
"""
import numpy as np
import torch


def shading_normal_chain(normal_view, world_view_transform):
    """normal_view [3,H,W] (view space, un-normalised) -> [H,W,3] unit world normals (with the reference's +1e-6)."""
    rot = world_view_transform[:3, :3].to(normal_view.dtype)
    n = torch.einsum("chw,dc->hwd", normal_view, rot)
    return n / (n.norm(dim=-1, keepdim=True) + 1e-6)


def view_rays_chain(H, W, K, R, T, dtype, device):
    """Unit world-space view ray of every pixel centre (integer pixel coordinates), [H,W,3].  R is the camera's stored
    rotation (camera-to-world, so R.T is world-to-camera), T the world-to-camera translation."""
    Kinv = torch.from_numpy(np.linalg.inv(np.asarray(K, dtype=np.float32))).to(dtype=dtype, device=device)
    xs = torch.arange(W, dtype=dtype, device=device)[None, :].expand(H, W)
    ys = torch.arange(H, dtype=dtype, device=device)[:, None].expand(H, W)
    pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1)
    cam_pts = torch.einsum("hwj,ij->hwi", pix, Kinv)
    Rw = R.to(dtype).T
    Tt = T.to(dtype)
    origin = -(Rw.T @ Tt)
    d = (cam_pts - Tt) @ Rw - origin
    return d / d.norm(dim=-1, keepdim=True)


def reflect_chain(d, n):
    return d - 2.0 * n * (d * n).sum(dim=-1, keepdim=True)


def reflection_chain(normal_view, base, strength, lookup, world_view_transform, H, W, K, R, T):
    """The whole deferred reflection pass with `lookup(dirs[B,3]) -> [B,3]` as the cubemap op: returns
    (final [3,H,W], reflected colour [3,H,W], shading normal [3,H,W])."""
    n = shading_normal_chain(normal_view, world_view_transform)
    d = view_rays_chain(H, W, K, R, T, normal_view.dtype, normal_view.device)
    r = reflect_chain(d, n)
    col = torch.sigmoid(lookup(r.reshape(-1, 3))).reshape(H, W, 3).permute(2, 0, 1)
    return (1 - strength) * base + strength * col, col, n.permute(2, 0, 1)


def surface_chain(allmap, world_view_transform, full_proj_transform, depth_ratio):
    """(surf_depth [1,H,W], surf_normal [3,H,W]) from the rasterizer's eight planes."""
    dt, dev = allmap.dtype, allmap.device
    H, W = allmap.shape[1:]
    alpha = allmap[1:2]
    expected = torch.nan_to_num(allmap[0:1] / torch.clamp(alpha, min=1e-3), 0, 0)
    median = torch.nan_to_num(allmap[5:6], 0, 0)
    depth = expected * (1 - depth_ratio) + depth_ratio * median
    c2w = torch.linalg.inv(world_view_transform.to(dt).T)
    to_pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=dt, device=dev).T
    intrins = ((c2w.T @ full_proj_transform.to(dt)) @ to_pix)[:3, :3].T
    xs = torch.arange(W, dtype=dt, device=dev)[None, :].expand(H, W)
    ys = torch.arange(H, dtype=dt, device=dev)[:, None].expand(H, W)
    pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1).reshape(-1, 3)
    dirs = pix @ torch.linalg.inv(intrins).T @ c2w[:3, :3].T
    pts = (depth.reshape(-1, 1) * dirs + c2w[:3, 3]).reshape(H, W, 3)
    ddy = pts[2:, 1:-1] - pts[:-2, 1:-1]
    ddx = pts[1:-1, 2:] - pts[1:-1, :-2]
    inner = torch.nn.functional.normalize(torch.cross(ddy, ddx, dim=-1), dim=-1)
    normal = torch.zeros_like(pts)
    normal[1:-1, 1:-1] = inner
    return depth, normal.permute(2, 0, 1) * alpha.detach()


# ---- the reference's op-by-op composition in float64 on the CPU, with the oracle's cubemap as the lookup (moved here from
# tests/test_gpu_cubemap.py so that the full-size tests can drive it too)
class OracleCubemap(torch.autograd.Function):
    """float64 CPU cubemap lookup through the oracle, as an autograd op (checker only)."""

    @staticmethod
    def forward(ctx, inputs, cubemap, fail):
        from oracle import oracle as orc
        out = orc.cubemap_forward(inputs.detach().numpy(), cubemap.detach().numpy(), fail.detach().numpy(), 1, 1, dtype=np.float64)
        ctx.save_for_backward(inputs, cubemap)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        from oracle import oracle as orc
        inputs, cubemap = ctx.saved_tensors
        gin, gcm, gf = orc.cubemap_backward(g.contiguous().numpy(), inputs.detach().numpy(), cubemap.detach().numpy(), 1, 1, dtype=np.float64)
        return torch.from_numpy(gin), torch.from_numpy(gcm), torch.from_numpy(gf)


def reference_chain(normal_view, base, strength, cubemap, fail, cam, W, H):
    """gaussian_renderer/__init__.py:22-35,148,178-179,197-199 + utils/general_utils.py:177-197, float64 on the CPU."""
    wvt = torch.from_numpy(cam["viewmatrix"]).double()
    R = torch.from_numpy(cam["R"]).double()
    T = torch.from_numpy(cam["T"]).double()
    K = cam["K"].astype(np.float32)
    rn = normal_view.permute(1, 2, 0) @ (wvt[:3, :3].T)
    rn = rn / (torch.norm(rn, dim=-1, keepdim=True) + 1e-6)
    Rw = R.T
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pc = torch.tensor(np.dot(xy1, np.linalg.inv(K).T)).double()
    rays_o = (-Rw.T @ T.unsqueeze(-1)).flatten()
    pw = (pc - T[None, None]).reshape(-1, 3) @ Rw
    rd = pw - rays_o[None]
    rd = (rd / torch.norm(rd, dim=1, keepdim=True)).reshape(H, W, 3)
    refl = rd - 2 * rn * torch.sum(rd * rn, dim=-1, keepdim=True)
    col = torch.sigmoid(OracleCubemap.apply(refl.reshape(-1, 3), cubemap, fail).permute(1, 0))
    col = col.reshape(H, W, 3).permute(2, 0, 1)
    final = (1 - strength) * base + strength * col
    return final, col, rn.permute(2, 0, 1)
