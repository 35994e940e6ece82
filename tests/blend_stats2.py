"""Dev aid (CPU, uses the oracle): which lane-group granularity should the surfel backward walk its lists at?  Not a test.

    python tests/blend_stats2.py [scale_divisor=3]

EXACT per-pixel blend evaluation of the (down-scaled, statistics-preserving) C3 scene, folded to candidate decompositions of a wave's 8x8 pixel
block into lane groups that each walk their OWN list of blending entries: 4x4 (a 16-lane DPP row; what surfel_render_bwd_rows_kernel does),
4x2 / 2x4 (8 lanes), 2x2 (a quad).  For each: blending (group, entry) pairs, wave iterations when the groups of a wave are coupled per
window of 1, 2, 4 batches of 64 list entries or free, slab slots per batch, and the share of lanes that carry a blending pixel."""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import scene_kwargs  # noqa: E402
from oracle import oracle as orc  # noqa: E402

f = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P, W, H, mu = 1000000 // (f * f), 1920 // f, 1080 // f, -4.75 + math.log(f)
kw, cam, sc = scene_kwargs("S", P, W, H, 1003, mu, 3, (0, 0, 0))
o = orc.SurfelOracle(np.float32)
o.forward(**kw)
T = o.state("transMat").astype(np.float32).reshape(-1, 9)
opa = o.state("normal_opacity")[:, 3].astype(np.float32)
m2d = o.state("means2D").astype(np.float32)
pl = o.state("point_list")
rg = o.state("ranges").astype(np.int64)
last = o.state("n_contrib")[0].astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
print("P %d  %dx%d  R %d  tiles %d" % (P, W, H, len(pl), gx * gy), flush=True)

yy, xx = np.mgrid[0:16, 0:16]
quad = (yy // 8) * 2 + (xx // 8)                                    # the wave (8x8 block) of a pixel
ly, lx = yy % 8, xx % 8
SHAPES = {"4x4": (4, 4), "4x2": (4, 2), "2x4": (2, 4), "2x2": (2, 2), "8x8": (8, 8)}      # (width, height) of a lane group
group_of = {}
for name, (gw, gh) in SHAPES.items():
    per_row = 8 // gw
    group_of[name] = (quad * ((8 // gw) * (8 // gh)) + (ly // gh) * per_row + (lx // gw)).reshape(-1)
tot = {name: dict(pairs=0, w1=0, w2=0, w4=0, free=0, slots_max=0) for name in SHAPES}
pix_pairs = 0
t0 = time.time()
for tile in range(gx * gy):
    a, b = rg[tile]
    if b <= a:
        continue
    tx, ty = tile % gx, tile // gx
    px = (tx * 16 + xx).astype(np.float32)
    py = (ty * 16 + yy).astype(np.float32)
    lastp = np.where((px < W) & (py < H), last[np.minimum(ty * 16 + yy, H - 1), np.minimum(tx * 16 + xx, W - 1)], 0)
    n = int(lastp.max())
    if n == 0:
        continue
    ids = pl[a:a + n]
    Tm = T[ids]
    Tu, Tv, Tw = Tm[:, 0:3], Tm[:, 3:6], Tm[:, 6:9]
    PX, PY = px[None], py[None]
    k = PX[..., None] * Tw[:, None, None, :] - Tu[:, None, None, :]
    l = PY[..., None] * Tw[:, None, None, :] - Tv[:, None, None, :]
    p = np.cross(k, l)
    with np.errstate(all="ignore"):
        sx, sy = p[..., 0] / p[..., 2], p[..., 1] / p[..., 2]
        rho3 = sx * sx + sy * sy
        dx, dy = m2d[ids, 0][:, None, None] - PX, m2d[ids, 1][:, None, None] - PY
        rho2 = 2.0 * (dx * dx + dy * dy)
        rho = np.minimum(rho3, rho2)
        depth = np.where(rho3 <= rho2, sx * Tw[:, None, None, 0] + sy * Tw[:, None, None, 1] + Tw[:, None, None, 2], Tw[:, None, None, 2])
        alpha = np.minimum(0.99, opa[ids][:, None, None] * np.exp(-0.5 * rho))
        ok = (p[..., 2] != 0) & ~(depth < 0.2) & ~(-0.5 * rho > 0) & ~(alpha < 1.0 / 255.0)
    ok &= np.arange(n)[:, None, None] < lastp[None]
    okf = ok.reshape(n, 256)
    pix_pairs += int(okf.sum())
    nb = (n + 63) // 64
    for name, (gw, gh) in SHAPES.items():
        ng = 4 * (8 // gw) * (8 // gh)                                  # groups per tile
        gpw = ng // 4                                                   # groups per wave
        g_any = np.zeros((nb * 64, ng), bool)
        onehot = np.zeros((256, ng), np.float32)
        onehot[np.arange(256), group_of[name]] = 1.0
        g_any[:n] = (okf.astype(np.float32) @ onehot) > 0
        d = tot[name]
        d["pairs"] += int(g_any.sum())
        per_batch = g_any.reshape(nb, 64, 4, gpw).sum(1)               # batches x waves x groups
        d["w1"] += int(per_batch.max(2).sum())
        for wname, wlen in (("w2", 2), ("w4", 4)):
            nbw = (nb + wlen - 1) // wlen
            padded = np.zeros((nbw * wlen, 4, gpw), np.int64)
            padded[nbw * wlen - nb:] = per_batch                        # windows are formed from the END of the list (the backward walks back to front)
            d[wname] += int(padded.reshape(nbw, wlen, 4, gpw).sum(1).max(2).sum())
        d["free"] += int(per_batch.sum(0).max(1).sum())
        d["slots_max"] = max(d["slots_max"], int(per_batch.sum(2).max()))
        d["slots_sum"] = d.get("slots_sum", 0) + int(per_batch.sum())
        d["batches"] = d.get("batches", 0) + int((per_batch.sum(2) > 0).sum())
    if tile % 200 == 0:
        print("  tile %d / %d  %.0f s" % (tile, gx * gy, time.time() - t0), flush=True)

s = f * f
print("scaled to C3 (x %d): blending (pixel, entry) pairs %.2f M" % (s, pix_pairs * s / 1e6))
print("%-5s %10s %9s | iterations (M): %8s %8s %8s %8s | %s" % ("group", "pairs (M)", "lanes %", "window 1", "window 2", "window 4", "free", "slab slots per (wave, batch): mean / max"))
for name, (gw, gh) in SHAPES.items():
    d = tot[name]
    lanes = gw * gh
    print("%-5s %10.2f %9.1f | %25.2f %8.2f %8.2f %8.2f | %.1f / %d" % (name, d["pairs"] * s / 1e6, 100.0 * pix_pairs / (lanes * d["pairs"]), d["w1"] * s / 1e6,
                                                                        d["w2"] * s / 1e6, d["w4"] * s / 1e6, d["free"] * s / 1e6,
                                                                        d["slots_sum"] / max(1, d["batches"]), d["slots_max"]))
