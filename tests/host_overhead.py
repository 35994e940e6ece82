"""Development aid: host cost of one rasterizer call through the ctypes binding vs the compiled pybind binding at BASELINE config
C1 (10k surfels, 256x256, SH 0).  Run on the GPU box twice:
    python tests/host_overhead.py            and            GSR_BINDING=pybind python tests/host_overhead.py
forward: wall time per call (it contains the 4-byte num_rendered read-back, i.e. one GPU round trip); backward: host time to
ENQUEUE one call (no synchronisation inside), measured over a burst that is synchronised once at the end."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import S, scene_kwargs, to_cuda  # noqa: E402
from diff_surfel_rasterization import _C  # noqa: E402

kw, cam, sc = scene_kwargs("S", 10_000, 256, 256, 1001, -3.0, 0, (0, 0, 0))
t = to_cuda(kw)
e = torch.empty(0, device="cuda")
args = (t["bg"], t["means3D"], t["env_scope_mask"], e, t["refl_strengths"], t["opacities"], t["scales"], t["rotations"], 1.0, e, t["viewmatrix"],
        t["projmatrix"], kw["tanfovx"], kw["tanfovy"], 256, 256, t["shs"], 0, t["campos"], False, False)
for _ in range(20):
    out = _C.rasterize_gaussians(*args)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    out = _C.rasterize_gaussians(*args)
torch.cuda.synchronize()
fwd_us = (time.perf_counter() - t0) / N * 1e6
R, color, others, radii, geom, binning, img, refl_map, gw = out
g = S.make_upstream_grads(256, 256, 1)
gc, ga, gr = (torch.from_numpy(g[k]).cuda() for k in ("dL_dcolor", "dL_dplanes", "dL_drefl"))
bargs = (t["bg"], t["means3D"], radii, e, t["refl_strengths"], t["scales"], t["rotations"], 1.0, e, t["viewmatrix"], t["projmatrix"], kw["tanfovx"],
         kw["tanfovy"], gc, ga, gr, t["shs"], 0, t["campos"], geom, R, binning, img, False)
for _ in range(20):
    _C.rasterize_gaussians_backward(*bargs)
torch.cuda.synchronize()
# bursts of 8 calls from an idle stream: short enough for the launch queue, so the loop measures the HOST side only
host, total = [], []
for _ in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        _C.rasterize_gaussians_backward(*bargs)
    host.append((time.perf_counter() - t0) / 8 * 1e6)
    torch.cuda.synchronize()
    total.append((time.perf_counter() - t0) / 8 * 1e6)
host_us, total_us = float(np.median(host)), float(np.median(total))
print("binding=%s  forward %.1f us/call (wall, incl. read-back)  backward enqueue %.1f us/call (host), %.1f us/call (wall)" % (
    os.environ.get("GSR_BINDING", "ctypes"), fwd_us, host_us, total_us))
