"""Dev aid (CPU, uses the oracle): how many (pixel block, surfel) pairs survive footprint culling at several block
sizes for a synthetic configuration.  Sizes the wave decomposition of the tile kernels.  Not a test."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import scene_kwargs
from oracle import oracle as orc

P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kw, cam, sc = scene_kwargs("S", P, W, H, 1003, float(sys.argv[4]) if len(sys.argv) > 4 else -4.75, 3, (0, 0, 0))
o = orc.SurfelOracle(np.float32)
t = time.time(); o.forward(**kw); print("oracle fwd %.1fs" % (time.time() - t))
T = o.state("transMat").astype(np.float64).reshape(-1, 3, 3)
opa = o.state("normal_opacity")[:, 3].astype(np.float64)
m2d = o.state("means2D").astype(np.float64)
pl = o.state("point_list"); rg = o.state("ranges").astype(np.int64)
ncontrib = o.state("n_contrib")[0]
R = len(pl)
# ellipse per surfel
c2 = 2 * np.log(np.maximum(255 * opa, 1e-30)) * 1.05 + 0.1
A = T
adj = np.empty_like(A)
a = A
adj[:, 0, 0] = a[:, 1, 1] * a[:, 2, 2] - a[:, 1, 2] * a[:, 2, 1]; adj[:, 0, 1] = a[:, 0, 2] * a[:, 2, 1] - a[:, 0, 1] * a[:, 2, 2]; adj[:, 0, 2] = a[:, 0, 1] * a[:, 1, 2] - a[:, 0, 2] * a[:, 1, 1]
adj[:, 1, 0] = a[:, 1, 2] * a[:, 2, 0] - a[:, 1, 0] * a[:, 2, 2]; adj[:, 1, 1] = a[:, 0, 0] * a[:, 2, 2] - a[:, 0, 2] * a[:, 2, 0]; adj[:, 1, 2] = a[:, 0, 2] * a[:, 1, 0] - a[:, 0, 0] * a[:, 1, 2]
adj[:, 2, 0] = a[:, 1, 0] * a[:, 2, 1] - a[:, 1, 1] * a[:, 2, 0]; adj[:, 2, 1] = a[:, 0, 1] * a[:, 2, 0] - a[:, 0, 0] * a[:, 2, 1]; adj[:, 2, 2] = a[:, 0, 0] * a[:, 1, 1] - a[:, 0, 1] * a[:, 1, 0]
D = np.stack([np.ones_like(c2), np.ones_like(c2), -c2], 1)
C = np.einsum("pki,pk,pkj->pij", adj, D, adj)
det2 = C[:, 0, 0] * C[:, 1, 1] - C[:, 0, 1] ** 2
with np.errstate(all="ignore"):
    ex = -(C[:, 1, 1] * C[:, 0, 2] - C[:, 0, 1] * C[:, 1, 2]) / det2
    ey = -(C[:, 0, 0] * C[:, 1, 2] - C[:, 0, 1] * C[:, 0, 2]) / det2
    q0 = C[:, 2, 2] + C[:, 0, 2] * ex + C[:, 1, 2] * ey
    ea, eb, ec = C[:, 0, 0] / -q0, C[:, 0, 1] / -q0, C[:, 1, 1] / -q0
valid = (det2 > 0) & (C[:, 0, 0] > 0) & (q0 < 0) & (opa >= 1 / 255)
print("valid ellipses %.3f" % valid.mean())
r2 = 0.5 * c2

gx = (W + 15) // 16
tile_of = np.repeat(np.arange(len(rg)), rg[:, 1] - rg[:, 0])
# early termination: instances past the tile's max contributor are never touched
nc_tile = np.zeros(len(rg), np.int64)
for ty in range((H + 15) // 16):
    for tx in range(gx):
        nc_tile[ty * gx + tx] = ncontrib[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].max()
pos = np.arange(R) - rg[tile_of, 0]
live = pos < nc_tile[tile_of]
print("R %d, live (before tile termination) %d (%.3f)" % (R, live.sum(), live.mean()))
idx = pl[live]; tl = tile_of[live]
tx0 = (tl % gx) * 16.0; ty0 = (tl // gx) * 16.0

def hits(bs):
    n = 16 // bs
    tot = 0
    for by in range(n):
        for bx in range(n):
            x0 = tx0 + bx * bs - 0.5; x1 = x0 + bs; y0 = ty0 + by * bs - 0.5; y1 = y0 + bs
            cx, cy = ex[idx], ey[idx]; a_, b_, c_ = ea[idx], eb[idx], ec[idx]
            with np.errstate(all="ignore"):
                inside = (cx >= x0) & (cx <= x1) & (cy >= y0) & (cy <= y1)
                best = np.full(len(idx), np.inf)
                for e in range(2):
                    dx = (x1 if e else x0) - cx; dy = np.clip(-b_ / c_ * dx, y0 - cy, y1 - cy)
                    best = np.minimum(best, a_ * dx * dx + 2 * b_ * dx * dy + c_ * dy * dy)
                    dy = (y1 if e else y0) - cy; dx = np.clip(-b_ / a_ * dy, x0 - cx, x1 - cx)
                    best = np.minimum(best, a_ * dx * dx + 2 * b_ * dx * dy + c_ * dy * dy)
                h = inside | ~(best > 1) | ~valid[idx]
                mx, my = m2d[idx, 0], m2d[idx, 1]
                ddx = np.clip(mx, x0, x1) - mx; ddy = np.clip(my, y0, y1) - my
                h |= (ddx * ddx + ddy * ddy <= r2[idx])
                h &= opa[idx] >= 1 / 255
            tot += int(h.sum())
    return tot

for bs in (16, 8, 4, 2):
    n = hits(bs)
    print("block %2dx%-2d: surviving pairs %9d  -> wave iterations at 64 px/iter %9.0f   (pixel-pairs %d)" % (bs, bs, n, n * bs * bs / 64.0, n * bs * bs))
