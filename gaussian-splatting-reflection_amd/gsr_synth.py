"""Synthetic cameras and Gaussian scenes for the BASELINE configs.

Everything is generated on the CPU with a seeded torch.Generator (seed = 1000 + config id) and
returned as float32 numpy arrays, so the oracle (numpy) and the HIP path (torch on the GPU) see the
same bytes.  Camera matrices follow scene/cameras.py:62-69 of the reference
(world_view_transform = getWorld2View2(R, T).T, full_proj_transform = wvt @ projection.T) and are
pinned against golden vectors produced by the reference's own utils/graphics_utils.py
(tests/golden/camera_golden.npz).
"""
import math
import numpy as np
import torch


def world2view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """utils/graphics_utils.py:38-49 — R is the camera-to-world rotation (3DGS stores W2C rotation transposed)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(C2W).astype(np.float32)


def projection_matrix(znear, zfar, fovX, fovY):
    """utils/graphics_utils.py:51-72 (float32 torch arithmetic there; float32 here)."""
    tanY, tanX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tanY * znear, tanX * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(W, H, fovy_deg=50.0, R=None, T=None, znear=0.01, zfar=100.0):
    """Camera dictionary with the tensors GaussianRasterizationSettings needs (scene/cameras.py:57-72)."""
    R = np.eye(3, dtype=np.float64) if R is None else np.asarray(R, dtype=np.float64)
    T = np.zeros(3, dtype=np.float64) if T is None else np.asarray(T, dtype=np.float64)
    fovy = math.radians(fovy_deg)
    fovx = 2.0 * math.atan(math.tan(fovy / 2) * W / H)
    wvt = world2view(R, T).T.copy()                                # world_view_transform
    proj = projection_matrix(znear, zfar, fovx, fovy).T.copy()     # projection_matrix (transposed)
    full = (wvt.astype(np.float32) @ proj.astype(np.float32)).astype(np.float32)
    campos = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    fx = W / (2.0 * math.tan(fovx / 2))
    fy = H / (2.0 * math.tan(fovy / 2))
    K = np.array([[fx, 0, W / 2.0], [0, fy, H / 2.0], [0, 0, 1]], dtype=np.float32)
    return dict(W=W, H=H, FoVx=fovx, FoVy=fovy, tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                viewmatrix=np.ascontiguousarray(wvt, dtype=np.float32), projmatrix=np.ascontiguousarray(full, dtype=np.float32),
                campos=campos, R=R.astype(np.float32), T=T.astype(np.float32), K=K, znear=znear, zfar=zfar)


def look_at_camera(W, H, eye, target=(0, 0, 0), up=(0, -1, 0), fovy_deg=50.0):
    """Camera at `eye` looking at `target` (+z forward, +y down as in COLMAP/3DGS)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64))  # x axis
    r /= np.linalg.norm(r)
    d = np.cross(f, r)                                  # y axis (down)
    c2w = np.stack([r, d, f], axis=1)                   # columns = camera axes in world
    w2c_R = c2w.T
    T = -w2c_R @ eye
    return make_camera(W, H, fovy_deg, R=c2w, T=T)


def yaw_camera(W, H, deg):
    """The C3 camera (R = I, T = 0) turned about the y axis by `deg` degrees.  View v of the C4 batch of bench.py looks 3 v degrees
    to the side (the C3 scene lives in a box in front of the C3 camera; the eight views see it with increasingly lopsided lists)."""
    a = math.radians(deg)
    c2w = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=np.float64)
    return make_camera(W, H, R=c2w, T=np.zeros(3))


def circle_cameras(W, H, n=8, radius=5.0, fovy_deg=50.0):
    """C4 views: n cameras on a circle of radius 5 in the xz-plane, height 0.5*sin(k), looking at the origin."""
    cams = []
    for k in range(n):
        ang = 2 * math.pi * k / n
        eye = (radius * math.sin(ang), 0.5 * math.sin(k), -radius * math.cos(ang))
        cams.append(look_at_camera(W, H, eye, fovy_deg=fovy_deg))
    return cams


# mean log-scale per config so that the mean tiles/Gaussian is about 6
CONFIGS = {
    1: dict(P=10_000, W=256, H=256, sh_degree=0, mu=-3.0),
    2: dict(P=100_000, W=800, H=800, sh_degree=3, mu=-3.6),
    3: dict(P=1_000_000, W=1920, H=1080, sh_degree=3, mu=-4.75),
    4: dict(P=1_000_000, W=1920, H=1080, sh_degree=3, mu=-4.75),
    5: dict(P=5_000_000, W=1920, H=1080, sh_degree=3, mu=-5.3),
}


def make_scene(P, variant="S", seed=1000, mu=-3.0, ball=False, mask_radius=0.0, cull_frac=0.02):
    """Random Gaussians.  variant 'S' -> scales (P,2); 'G' -> scales (P,3) + normals."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
    ru = lambda *s: torch.rand(*s, generator=g, dtype=torch.float32)
    if ball:
        d = rn(P, 3)
        d = d / d.norm(dim=1, keepdim=True)
        means = d * (2.0 * ru(P, 1) ** (1.0 / 3.0))
    else:
        lo = torch.tensor([-2.0, -1.2, 3.0])
        hi = torch.tensor([2.0, 1.2, 7.0])
        means = lo + (hi - lo) * ru(P, 3)
        ncull = int(P * cull_frac)
        if ncull > 0:
            means[:ncull, 2] = 0.3 * ru(ncull)   # z in [0, 0.3]: exercises the near cull
    ns = 2 if variant == "S" else 3
    scales = torch.exp(mu + 0.5 * rn(P, ns))
    rot = rn(P, 4)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opacity = torch.sigmoid(1.5 * rn(P, 1))
    shs = torch.cat([rn(P, 1, 3), 0.15 * rn(P, 15, 3)], dim=1)
    refl = torch.sigmoid(-2.0 + rn(P, 1))
    normals = rn(P, 3)
    normals = normals / normals.norm(dim=1, keepdim=True)
    if mask_radius > 0:
        mask = (means ** 2).sum(dim=1) < mask_radius ** 2
    else:
        mask = torch.ones(P, dtype=torch.bool)
    out = dict(means3D=means, scales=scales, rotations=rot, opacities=opacity, shs=shs, refl_strengths=refl, normals=normals,
               env_scope_mask=mask)
    return {k: np.ascontiguousarray(v.numpy()) for k, v in out.items()}


def make_cubemap(L=128, C=3, seed=1000):
    g = torch.Generator(device="cpu").manual_seed(int(seed) + 77)
    tex = torch.rand(6, C, L, L, generator=g, dtype=torch.float32) - 0.5   # cubemap_encoder.py:94
    fail = torch.zeros(C, dtype=torch.float32)
    return tex.numpy(), fail.numpy()


def make_upstream_grads(H, W, seed=1000, planes=8):
    """dL/d(outputs) = N(0,1)/HW."""
    g = torch.Generator(device="cpu").manual_seed(int(seed) + 99)
    rn = lambda *s: (torch.randn(*s, generator=g, dtype=torch.float32) / float(H * W)).numpy()
    return dict(dL_dcolor=rn(3, H, W), dL_dplanes=rn(planes, H, W), dL_drefl=rn(1, H, W), dL_dinvdepth=rn(1, H, W),
                dL_dnormal=rn(3, H, W))
