"""Dev aid: how coherent are the cubemap texels hit by neighbouring pixels of the C3 bench scene?
(decides whether run-combining before the LDS adds of the binned reflection backward can pay)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-reflection_amd"))
import bench
import gsr_synth as S
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda", 0)
P, W, H, L = 1000000, 1920, 1080, 128
scene = bench.Scene(S, P, -4.75, L, dev, seed=1003)
cam = bench.yaw_camera(S, W, H, 0.0)
ct = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cam.items() if isinstance(v, np.ndarray)}
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                                   scale_modifier=1.0, viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"], sh_degree=3,
                                   campos=ct["campos"], prefiltered=False, debug=False)
with torch.no_grad():
    base, radii, allmap, refl_map, gw = GaussianRasterizer(st)(means3D=scene.p["means3D"], means2D=torch.zeros(P, 3, device=dev),
        opacities=scene.p["opacities"], shs=scene.p["shs"], refl_strengths=scene.p["refl_strengths"], scales=scene.p["scales"],
        rotations=scene.p["rotations"], env_scope_mask=scene.mask)
n = allmap[2:5]
n = n / (n.norm(dim=0, keepdim=True) + 1e-6)
K = cam["K"]
ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
d = torch.stack([(xs - K[0][2]) / K[0][0], (ys - K[1][2]) / K[1][1], torch.ones_like(xs)])
d = d / d.norm(dim=0, keepdim=True)
r = d - 2 * (d * n).sum(0, keepdim=True) * n
a = r.abs()
face = a.argmax(0)
m = a.max(0).values.clamp_min(1e-9)
o1 = torch.where(face == 0, r[1], r[0]) / m
o2 = torch.where(face == 2, r[1], r[2]) / m
sign = torch.gather(r, 0, face[None])[0] < 0
tx = ((o1 * 0.5 + 0.5) * L - 0.5).floor().long().clamp(0, L - 1)
ty = ((o2 * 0.5 + 0.5) * L - 0.5).floor().long().clamp(0, L - 1)
t = ((face * 2 + sign.long()) * L + ty) * L + tx
print("pixels", t.numel(), "distinct texels", t.unique().numel())
print("same texel as x-neighbour:", (t[:, 1:] == t[:, :-1]).float().mean().item())
print("same texel as y-neighbour:", (t[1:] == t[:-1]).float().mean().item())
print("same band(1024) as x-neighbour:", (t[:, 1:] // 1024 == t[:, :-1] // 1024).float().mean().item())
for wdt in (4, 8, 16):
    tt = t[:, : W // wdt * wdt].reshape(H, -1, wdt)
    distinct = (tt.sort(dim=2).values.diff(dim=2) != 0).sum(2) + 1
    print(f"distinct texels per {wdt} consecutive pixels: {distinct.float().mean().item():.2f}")
print("|n| < 0.5 fraction:", (allmap[2:5].norm(dim=0) < 0.5).float().mean().item())
# ---- statistics of the sorted-footprint accumulation (refl_run_combine_kernel)
ts = t.flatten().sort().values
n = ts.numel()
for per_wg in (2048, 4096):
    m = n // per_wg * per_wg
    g = ts[:m].reshape(-1, per_wg)
    span = g[:, -1] - g[:, 0]
    print(f"records/WG {per_wg}: texel span median {span.median().item()}, mean {span.float().mean().item():.0f}, max {span.max().item()}, "
          f"WGs with span > 4096-130: {(span > 4096 - 130).float().mean().item():.3f}")
    over = (g - g[:, :1]) > (4096 - 130)
    print(f"   records beyond the window: {over.float().mean().item():.4f}")
for ch in (8, 16):
    m = n // ch * ch
    g = ts[:m].reshape(-1, ch)
    fl = (g[:, 1:] != g[:, :-1]).sum(1) + 1
    print(f"chunk {ch}: flushes per chunk {fl.float().mean().item():.2f} -> LDS adds per record {12 * fl.float().mean().item() / ch:.2f}")
cnt = torch.bincount(t.flatten(), minlength=6 * L * L)
print("records per texel: max", cnt.max().item(), " texels holding half of the records:", (cnt.sort(descending=True).values.cumsum(0) < n // 2).sum().item())
