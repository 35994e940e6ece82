"""Generates tests/golden/*.npz from the parts of the reference that are importable on the CPU in the build
container: utils/sh_utils.py (eval_sh, RGB2SH, SH2RGB) and utils/graphics_utils.py
(getWorld2View2, getProjectionMatrix, getProjectionMatrixCorrect, fov2focal, focal2fov).

Run ONCE in the build container (the reference is not present on the GPU box); the outputs are committed:
    python tests/golden/make_golden.py
Only inputs and expected outputs are stored (data, no reference source text).
"""
import math
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from utils.sh_utils import eval_sh, RGB2SH, SH2RGB  # noqa: E402
from utils.graphics_utils import getWorld2View2, getProjectionMatrix, getProjectionMatrixCorrect, fov2focal, focal2fov  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sh_golden():
    g = torch.Generator().manual_seed(123)
    N = 257
    means = torch.randn(N, 3, generator=g) * 2.0
    campos = torch.tensor([0.3, -0.2, 0.5])
    shs = torch.cat([torch.randn(N, 1, 3, generator=g), 0.3 * torch.randn(N, 15, 3, generator=g)], dim=1)  # kernel layout (N,16,3)
    dirs = means - campos[None]
    dirs_n = dirs / dirs.norm(dim=1, keepdim=True)
    out = {"means": means.numpy(), "campos": campos.numpy(), "shs": shs.numpy()}
    for deg in range(4):
        # gaussian_renderer/__init__.py:118-122 of the reference: shs_view is (N,3,M); colour = clamp_min(eval_sh + 0.5, 0)
        raw = eval_sh(deg, shs.transpose(1, 2), dirs_n)
        out[f"raw_deg{deg}"] = raw.numpy()
        out[f"rgb_deg{deg}"] = torch.clamp_min(raw + 0.5, 0.0).numpy()
    rgb = torch.rand(64, 3, generator=g)
    out["rgb_in"] = rgb.numpy()
    out["rgb2sh"] = RGB2SH(rgb).numpy()
    out["sh2rgb"] = SH2RGB(RGB2SH(rgb)).numpy()
    np.savez(os.path.join(OUT, "sh_golden.npz"), **out)


def camera_golden():
    rs = np.random.RandomState(7)
    recs = {}
    cases = []
    for i in range(6):
        # random rotation (QR) and translation; 3DGS stores R = C2W rotation, T = W2C translation
        q, _ = np.linalg.qr(rs.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        if i == 0:
            q = np.eye(3)
        t = rs.randn(3) * (0.0 if i == 0 else 2.0)
        fovy = math.radians(30 + 10 * i)
        W, H = (1920, 1080) if i % 2 == 0 else (800, 800)
        fovx = 2 * math.atan(math.tan(fovy / 2) * W / H)
        wvt = torch.tensor(getWorld2View2(q, t)).transpose(0, 1)                       # scene/cameras.py:62
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)   # :64
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)                      # :68
        center = wvt.inverse()[3, :3]                                                    # :69
        K = np.array([[fov2focal(fovx, W), 0, W / 2.0], [0, fov2focal(fovy, H), H / 2.0], [0, 0, 1]], dtype=np.float64)
        projc = getProjectionMatrixCorrect(0.01, 100.0, H, W, K).transpose(0, 1)
        recs[f"R{i}"] = q
        recs[f"T{i}"] = t
        recs[f"fov{i}"] = np.array([fovx, fovy, W, H], dtype=np.float64)
        recs[f"wvt{i}"] = wvt.numpy()
        recs[f"proj{i}"] = proj.numpy()
        recs[f"full{i}"] = full.numpy()
        recs[f"center{i}"] = center.numpy()
        recs[f"projc{i}"] = projc.numpy()
        recs[f"focal{i}"] = np.array([fov2focal(fovx, W), fov2focal(fovy, H), focal2fov(fov2focal(fovx, W), W)], dtype=np.float64)
        cases.append(i)
    recs["n"] = np.array(len(cases))
    np.savez(os.path.join(OUT, "camera_golden.npz"), **recs)


if __name__ == "__main__":
    sh_golden()
    camera_golden()
    print("written", sorted(os.listdir(OUT)))
