"""Dev aid (CPU, uses the oracle): how many (8x8 block, list entry) pairs the forward tile kernel evaluates in variant S, split
by why a pair turns out empty.  Not a test.       python tests/vote_stats.py [scale_divisor=4]

Per quadrant block, in list order, until all 64 pixels have retired (T < 1e-4) or the list ends:
  voted       entries whose footprint (ellipse of the dual conic at c2 = 2 ln(255 opacity) * 1.05 + 0.1, united with the low-pass
              disc) meets the block's pixel-area box — the model of the kernel's vote
  voted, live the same against the bounding box of the pixels that have not retired yet (refreshed per batch of 64 entries)
  blended     entries that blend into at least one pixel
"""
import math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import scene_kwargs
from oracle import oracle as orc

f = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P, W, H, mu = 1000000 // (f * f), 1920 // f, 1080 // f, -4.75 + math.log(f)
kw, cam, sc = scene_kwargs("S", P, W, H, 1003, mu, 3, (0, 0, 0))
o = orc.SurfelOracle(np.float32)
o.forward(**kw)
T9 = o.state("transMat").astype(np.float64).reshape(-1, 3, 3)
opa = o.state("normal_opacity")[:, 3].astype(np.float64)
m2d = o.state("means2D").astype(np.float64)
pl = o.state("point_list"); rg = o.state("ranges").astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
# footprint ellipse per surfel (dual conic, as tests/cull_stats.py)
c2 = 2 * np.log(np.maximum(255 * opa, 1e-30)) * 1.05 + 0.1
a = T9
adj = np.empty_like(a)
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
valid = (det2 > 0) & (C[:, 0, 0] > 0) & (q0 < 0)
r2 = 0.5 * c2
# the same footprint without any margin (c2 = 2 ln(255 opacity)), to tell margin-caused votes from lattice misses
c2t = 2 * np.log(np.maximum(255 * opa, 1e-30))
Dt = np.stack([np.ones_like(c2t), np.ones_like(c2t), -np.maximum(c2t, 1e-9)], 1)
Ct = np.einsum("pki,pk,pkj->pij", adj, Dt, adj)
det2t = Ct[:, 0, 0] * Ct[:, 1, 1] - Ct[:, 0, 1] ** 2
with np.errstate(all="ignore"):
    ext = -(Ct[:, 1, 1] * Ct[:, 0, 2] - Ct[:, 0, 1] * Ct[:, 1, 2]) / det2t
    eyt = -(Ct[:, 0, 0] * Ct[:, 1, 2] - Ct[:, 0, 1] * Ct[:, 0, 2]) / det2t
    q0t = Ct[:, 2, 2] + Ct[:, 0, 2] * ext + Ct[:, 1, 2] * eyt
    eat, ebt, ect = Ct[:, 0, 0] / -q0t, Ct[:, 0, 1] / -q0t, Ct[:, 1, 1] / -q0t
validt = (det2t > 0) & (Ct[:, 0, 0] > 0) & (q0t < 0) & (c2t > 0)
r2t = 0.5 * np.maximum(c2t, 0)
TIGHT = False
# the kernel's record: E^-1 grown by 2 % in length and dilated by half a pixel
with np.errstate(all="ignore"):
    detE = ea * ec - eb * eb
    DIL = float(os.environ.get("DIL", "0.5")) ** 2
    Vxx, Vxy, Vyy = ec / detE * 1.0404 + DIL, -eb / detE * 1.0404, ea / detE * 1.0404 + DIL
    dV = Vxx * Vyy - Vxy * Vxy
    da, db, dc = Vyy / dV, -Vxy / dV, Vxx / dV
r2d = r2 * 1.02


def vote_px(ids, bx0, by0):
    """any pixel centre of the 8x8 block inside the dilated ellipse or the disc"""
    X = (bx0 + np.arange(8))[None, None, :]; Y = (by0 + np.arange(8))[None, :, None]
    with np.errstate(all="ignore"):
        dx = X - ex[ids][:, None, None]; dy = Y - ey[ids][:, None, None]
        q = da[ids][:, None, None] * dx * dx + 2 * db[ids][:, None, None] * dx * dy + dc[ids][:, None, None] * dy * dy
        h = (q <= 1).any(axis=(1, 2)) | ~valid[ids]
        ddx = X - m2d[ids, 0][:, None, None]; ddy = Y - m2d[ids, 1][:, None, None]
        h |= ((ddx * ddx + ddy * ddy) <= r2d[ids][:, None, None]).any(axis=(1, 2))
        h &= opa[ids] >= 1 / 255
    return h


def vote(ids, x0, x1, y0, y1):
    """footprint of surfels `ids` against the box [x0,x1] x [y0,y1] (arrays broadcast against ids)"""
    if TIGHT:
        cx, cy, a_, b_, c_ = ext[ids], eyt[ids], eat[ids], ebt[ids], ect[ids]
    else:
        cx, cy, a_, b_, c_ = ex[ids], ey[ids], ea[ids], eb[ids], ec[ids]
    with np.errstate(all="ignore"):
        inside = (cx >= x0) & (cx <= x1) & (cy >= y0) & (cy <= y1)
        best = np.full(len(ids), np.inf)
        for e in range(2):
            dx = (x1 if e else x0) - cx; dy = np.clip(-b_ / c_ * dx, y0 - cy, y1 - cy)
            best = np.minimum(best, a_ * dx * dx + 2 * b_ * dx * dy + c_ * dy * dy)
            dy = (y1 if e else y0) - cy; dx = np.clip(-b_ / a_ * dy, x0 - cx, x1 - cx)
            best = np.minimum(best, a_ * dx * dx + 2 * b_ * dx * dy + c_ * dy * dy)
        h = inside | ~(best > 1) | ~(validt if TIGHT else valid)[ids]
        mx, my = m2d[ids, 0], m2d[ids, 1]
        ddx = np.clip(mx, x0, x1) - mx; ddy = np.clip(my, y0, y1) - my
        h |= (ddx * ddx + ddy * ddy <= (r2t if TIGHT else r2)[ids])
        h &= opa[ids] >= 1 / 255
    return h


T32 = T9.astype(np.float32).reshape(-1, 9); opa32 = opa.astype(np.float32); m32 = m2d.astype(np.float32)
tot = dict(voted_px=0, blended_px_missed=0, voted=0, live=0, blended=0, walked=0, empty_done=0, empty_margin=0, empty_lattice=0)
t0 = time.time()
yy, xx = np.mgrid[0:8, 0:8]
for tile in range(gx * gy):
    a0, b0 = rg[tile]
    n = int(b0 - a0)
    if n == 0:
        continue
    tx, ty = tile % gx, tile // gx
    ids = pl[a0:b0]
    for q in range(4):
        bx0, by0 = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
        if bx0 >= W or by0 >= H:
            continue
        px = (bx0 + xx).astype(np.float32); py = (by0 + yy).astype(np.float32)
        inside = (px < W) & (py < H)
        Tm = T32[ids]; Tu, Tv, Tw = Tm[:, 0:3], Tm[:, 3:6], Tm[:, 6:9]
        k = px[None, ..., None] * Tw[:, None, None, :] - Tu[:, None, None, :]
        l = py[None, ..., None] * Tw[:, None, None, :] - Tv[:, None, None, :]
        p = np.cross(k, l)
        with np.errstate(all="ignore"):
            sx, sy = p[..., 0] / p[..., 2], p[..., 1] / p[..., 2]
            rho3 = sx * sx + sy * sy
            dx, dy = m32[ids, 0][:, None, None] - px[None], m32[ids, 1][:, None, None] - py[None]
            rho = np.minimum(rho3, 2.0 * (dx * dx + dy * dy))
            depth = np.where(rho3 <= 2.0 * (dx * dx + dy * dy), sx * Tw[:, None, None, 0] + sy * Tw[:, None, None, 1] + Tw[:, None, None, 2], Tw[:, None, None, 2])
            alpha = np.minimum(0.99, opa32[ids][:, None, None] * np.exp(-0.5 * rho))
            ok = (np.abs(p[..., 2]) >= 1e-4) & ~(depth < 0.2) & ~(alpha < 1.0 / 255.0) & inside[None]
        vfull = vote(ids, bx0 - 0.5, bx0 + 7.5, by0 - 0.5, by0 + 7.5)
        vpx = vote_px(ids, bx0, by0)
        Tpix = np.ones((8, 8), np.float32); done = ~inside
        lx0, lx1, ly0, ly1 = bx0 - 0.5, bx0 + 7.5, by0 - 0.5, by0 + 7.5
        for e in range(n):
            if done.all():
                break
            if e % 64 == 0:                      # live box refreshed per batch
                lv = ~done
                xs, ys = px[lv], py[lv]
                lx0, lx1, ly0, ly1 = xs.min() - 0.5, xs.max() + 0.5, ys.min() - 0.5, ys.max() + 0.5
            tot["walked"] += 1
            if not vfull[e]:
                continue
            tot["voted"] += 1
            if vpx[e]:
                tot["voted_px"] += 1
            if vote(ids[e:e + 1], lx0, lx1, ly0, ly1)[0]:
                tot["live"] += 1
            live = ok[e] & ~done
            if not live.any():
                if ok[e].any():
                    tot["empty_done"] += 1          # a pixel would blend, but it has retired
                else:
                    TIGHT = True
                    tight_hit = vote(ids[e:e + 1], bx0, bx0 + 7.0, by0, by0 + 7.0)[0]     # margin-free footprint against the box of pixel centres
                    TIGHT = False
                    tot["empty_lattice" if tight_hit else "empty_margin"] += 1
                continue
            tT = Tpix * (1 - alpha[e])
            sat = live & (tT < 1e-4)
            blend = live & ~sat
            done |= sat
            Tpix = np.where(blend, tT, Tpix)
            if blend.any():
                tot["blended"] += 1
                if not vpx[e]:
                    tot["blended_px_missed"] += 1
    if tile % 100 == 0:
        print("  tile %d / %d  %.0f s" % (tile, gx * gy, time.time() - t0), flush=True)
s = f * f
print("empty pairs: %.2f M because the pixels that would blend have retired, %.2f M where only the margins of the vote reach the block, %.2f M where the exact footprint meets the block but holds no pixel centre (or every pixel fails the depth / alpha tests)"
      % (tot["empty_done"] * s / 1e6, tot["empty_margin"] * s / 1e6, tot["empty_lattice"] * s / 1e6))
print("pixel-centre vote: %.2f M pairs (of %.2f M voted); blending pairs it would have dropped: %d" % (tot["voted_px"] * s / 1e6, tot["voted"] * s / 1e6, tot["blended_px_missed"]))
print("scaled to C3 (x %d):  walked %.2f M list entries, voted %.2f M pairs, voted against the live box %.2f M, blended %.2f M"
      % (s, tot["walked"] * s / 1e6, tot["voted"] * s / 1e6, tot["live"] * s / 1e6, tot["blended"] * s / 1e6))
