"""Helper of tests/test_gpu_binding.py (not a test module): runs a small scene of each variant through whichever binding
GSR_BINDING selects and writes every output and gradient to an .npz file."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import HipGauss, HipSurfel, S, scene_kwargs  # noqa: E402
import _gsr  # noqa: E402


def main(path):
    out = {"binding": np.array(0 if _gsr.PYBIND is None else 1)}
    g = S.make_upstream_grads(136, 200, 7)
    kw, _, _ = scene_kwargs("S", 5_000, 200, 136, 7, -3.0, 3, (1, 1, 1), mask_radius=4.5)
    hip = HipSurfel(kw)
    for k, v in hip.out().items():
        out["S_" + k] = np.asarray(v)
    out["S_n_contrib"] = hip.state("n_contrib")
    for k, v in hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"]).items():
        if v is not None:
            out["S_" + k] = v
    kw, _, _ = scene_kwargs("G", 5_000, 200, 136, 7, -3.0, 3, (1, 1, 1))
    hip = HipGauss(kw, antialiasing=True)
    for k, v in hip.out().items():
        out["G_" + k] = np.asarray(v)
    for k, v in hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"]).items():
        if v is not None:
            out["G_" + k] = v
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    import torch
    t = {k: torch.from_numpy(v).cuda() for k, v in kw.items() if isinstance(v, np.ndarray)}
    st = GaussianRasterizationSettings(image_height=136, image_width=200, tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"], bg=t["bg"], scale_modifier=1.0,
                                       viewmatrix=t["viewmatrix"], projmatrix=t["projmatrix"], sh_degree=3, campos=t["campos"], prefiltered=False,
                                       debug=False)
    out["visible"] = GaussianRasterizer(st).markVisible(t["means3D"]).cpu().numpy()
    np.savez(path, **out)


if __name__ == "__main__":
    main(sys.argv[1])
