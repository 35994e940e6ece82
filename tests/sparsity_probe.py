"""Development aid (GPU box): how sparse are the bench scenes?  Visible / instance / blended counts of C3 (variant S) and C5 (variant G)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-reflection_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsr_synth as S  # noqa: E402


def probe(variant, P, mu, seed):
    W, H = 1920, 1080
    sc = S.make_scene(P, variant, seed=seed, mu=mu)
    cam = S.make_camera(W, H)
    dev = "cuda"
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cam.items() if isinstance(v, np.ndarray)}
    t = {k: torch.from_numpy(sc[k]).to(dev) for k in sc if isinstance(sc[k], np.ndarray) and sc[k].dtype == np.float32}
    if variant == "S":
        from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                                           scale_modifier=1.0, viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"], sh_degree=3, campos=ct["campos"],
                                           prefiltered=False, debug=False)
        out = GaussianRasterizer(st)(means3D=t["means3D"], means2D=torch.zeros(P, 3, device=dev), opacities=t["opacities"], shs=t["shs"],
                                     refl_strengths=t["refl_strengths"], scales=t["scales"], rotations=t["rotations"],
                                     env_scope_mask=torch.from_numpy(sc["env_scope_mask"]).to(dev))
        radii, gw = out[1], out[4]
        print(variant, P, "visible %.3f" % (radii > 0).float().mean().item(), "blended (of all) %.3f" % (gw > 0).float().mean().item())
    else:
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                                           scale_modifier=1.0, viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"], sh_degree=3, campos=ct["campos"],
                                           prefiltered=False, antialiasing=True, debug=False)
        out = GaussianRasterizer(st)(means3D=t["means3D"], means2D=torch.zeros(P, 3, device=dev), opacities=t["opacities"], shs=t["shs"], normals=t["normals"],
                                     refl_strengths=t["refl_strengths"], scales=t["scales"], rotations=t["rotations"])
        radii = out[1]
        print(variant, P, "visible %.3f" % (radii > 0).float().mean().item())


probe("S", 1_000_000, -4.75, 1003)
probe("G", 5_000_000, -5.3, 1005)
