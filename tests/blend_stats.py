"""Dev aid (CPU, uses the oracle): EXACT statistics of which (pixel, list entry) pairs blend in variant S, folded to the
work decompositions the backward tile kernel could use.  Not a test.

    python tests/blend_stats.py [scale_divisor=2]

Runs the C3 scene at 1 / f^2 of its size with the same per-tile statistics (Gaussians / f^2, image / f per side, scales x f)
and prints, per decomposition, the number of wave iterations (one iteration = one 64-lane pass of the pair code):
  shared 8x8     one list per 8x8 block: every entry that blends into >= 1 of its 64 pixels (what the kernel does today)
  rows, coupled  four 16-lane rows = four 4x4 sub-blocks with their own lists, all four walking the same batch of 64 list
                 entries (records staged per batch): sum over batches of the longest row
  rows, free     the same with rows free to run ahead into other batches: longest row per wave
  ideal          blending (4x4 sub-block, entry) pairs / 4
"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import scene_kwargs
from oracle import oracle as orc

f = int(sys.argv[1]) if len(sys.argv) > 1 else 2
P, W, H, mu = 1000000 // (f * f), 1920 // f, 1080 // f, -4.75 + math.log(f)
kw, cam, sc = scene_kwargs("S", P, W, H, 1003, mu, 3, (0, 0, 0))
o = orc.SurfelOracle(np.float32)
t = time.time()
o.forward(**kw)
print("oracle forward %.1f s" % (time.time() - t))
T = o.state("transMat").astype(np.float32).reshape(-1, 9)
opa = o.state("normal_opacity")[:, 3].astype(np.float32)
m2d = o.state("means2D").astype(np.float32)
pl = o.state("point_list")
rg = o.state("ranges").astype(np.int64)
last = o.state("n_contrib")[0].astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
print("P %d  %dx%d  R %d  tiles %d" % (P, W, H, len(pl), gx * gy))

tot = dict(shared=0, coupled=0, free=0, sub=0, pix=0, half_coupled=0, half_free=0, strip_free=0)
yy, xx = np.mgrid[0:16, 0:16]
# sub-block index of every pixel of a tile: quadrant (8x8) * 4 + 4x4 block inside the quadrant
quad = (yy // 8) * 2 + (xx // 8)
sub = quad * 4 + ((yy % 8) // 4) * 2 + ((xx % 8) // 4)
half = quad * 2 + ((yy % 8) // 4)          # 8 wide x 4 high halves
strip = quad * 4 + ((yy % 8) // 2)         # 8 wide x 2 high strips (16 lanes)
t = time.time()
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
    Tm = T[ids]                                            # n x 9 : Tu (0..2), Tv (3..5), Tw (6..8)
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
    tot["pix"] += int(ok.sum())
    okf = ok.reshape(n, 256)
    s_any = np.zeros((n, 16), bool)
    for s in range(16):
        s_any[:, s] = okf[:, (sub == s).reshape(-1)].any(1)
    h_any = np.zeros((n, 8), bool)
    for s in range(8):
        h_any[:, s] = okf[:, (half == s).reshape(-1)].any(1)
    st_any = np.zeros((n, 16), bool)
    for s in range(16):
        st_any[:, s] = okf[:, (strip == s).reshape(-1)].any(1)
    tot["sub"] += int(s_any.sum())
    nb = (n + 63) // 64
    pad = np.zeros((nb * 64 - n, 16), bool)
    sb = np.concatenate([s_any, pad]).reshape(nb, 64, 16).sum(1)          # batches x sub-blocks
    hb = np.concatenate([h_any, pad[:, :8]]).reshape(nb, 64, 8).sum(1)
    sp = np.concatenate([s_any, pad])
    for c in (32, 16, 8):
        sc = sp.reshape(nb * (64 // c), c, 16).sum(1)
        for q in range(4):
            tot["chunk%d" % c] = tot.get("chunk%d" % c, 0) + int(sc[:, q * 4:q * 4 + 4].max(1).sum())
    # adaptive chunks: consecutive entries of a batch while the (sub-block, entry) pairs of the quadrant fit CAP slab slots
    for q in range(4):
        sq = sp[:, q * 4:q * 4 + 4].reshape(nb, 64, 4)
        for cap in (32, 48, 64):
            it = 0
            for bi in range(nb):
                cnt = np.zeros(4, np.int64)
                for e in range(64):
                    row = sq[bi, e]
                    if cnt.sum() + row.sum() > cap:
                        it += int(cnt.max()); cnt[:] = 0
                    cnt += row
                it += int(cnt.max())
            tot["cap%d" % cap] = tot.get("cap%d" % cap, 0) + it
    for q in range(4):
        rows = sb[:, q * 4:q * 4 + 4]
        tot["shared"] += int(s_any[:, q * 4:q * 4 + 4].any(1).sum())
        tot["coupled"] += int(rows.max(1).sum())
        tot["free"] += int(rows.sum(0).max())
        hv = hb[:, q * 2:q * 2 + 2]
        tot["half_coupled"] += int(hv.max(1).sum())
        tot["half_free"] += int(hv.sum(0).max())
        tot["strip_free"] += int(st_any[:, q * 4:q * 4 + 4].sum(0).max())
    if tile % 500 == 0:
        print("  tile %d / %d  %.0f s" % (tile, gx * gy, time.time() - t), flush=True)

s = f * f
print("scaled to C3 (x %d):" % s)
print("  blending (pixel, entry) pairs      %10.2f M" % (tot["pix"] * s / 1e6))
print("  shared 8x8 list                     %10.2f M iterations   (%.0f %% useful lanes)" % (tot["shared"] * s / 1e6, 100.0 * tot["pix"] / (64.0 * tot["shared"])))
print("  4x4 sub-block pairs                 %10.2f M   -> ideal %.2f M iterations" % (tot["sub"] * s / 1e6, tot["sub"] * s / 4e6))
print("  four 4x4 rows, coupled per batch    %10.2f M iterations" % (tot["coupled"] * s / 1e6))
for c in (32, 16, 8):
    print("  four 4x4 rows, coupled per %2d entries %9.2f M iterations" % (c, tot["chunk%d" % c] * s / 1e6))
for c in (32, 48, 64):
    print("  four 4x4 rows, chunks of <= %d pairs %9.2f M iterations" % (c, tot["cap%d" % c] * s / 1e6))
print("  four 4x4 rows, free                 %10.2f M iterations" % (tot["free"] * s / 1e6))
print("  four 8x2 strips, free               %10.2f M iterations" % (tot["strip_free"] * s / 1e6))
print("  two 8x4 halves, coupled per batch   %10.2f M iterations" % (tot["half_coupled"] * s / 1e6))
print("  two 8x4 halves, free                %10.2f M iterations" % (tot["half_free"] * s / 1e6))
