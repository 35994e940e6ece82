"""Shared helpers for the parity tests: build a synthetic scene, run it through the
oracle (numpy, CPU) and through the HIP path (torch tensors on cuda:0, via the drop-in Python API)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting-reflection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import gsr_synth as S  # noqa: E402


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    if mse == 0:
        return 200.0
    return 10.0 * np.log10(peak * peak / mse)


def rel_maxnorm(a, b):
    """max-norm error of a against reference b, relative to max|b| (BASELINE.md §4 gradient gate)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = float(np.abs(b).max())
    if den == 0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max()) / den


def grad_gate(a, b, rtol=1e-4, floor=1e-6):
    """Elementwise gradient gate: fraction of elements with |a - b| > rtol * |b| + floor * max|b|.  Unlike the per-tensor
    max-norm (rel_maxnorm) it sees errors confined to small-magnitude rows; the floor term is the fp32 resolution of sums whose
    terms reach max|b| (atomics arrive in a different order on every run)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64).reshape(a.shape)
    if a.size == 0:
        return 0.0
    tol = rtol * np.abs(b) + floor * float(np.abs(b).max())
    return float((np.abs(a - b) > tol).mean())


# n_contrib (per-pixel last / median contributor) flips where alpha or T sits within an ulp of a threshold.  Observed on MI355X (round 3,
# tests/observed_errors.py): 0 pixels at C1, 1 of 640 000 (S) / 2 of 640 000 (G) at C2, 4 of 2 073 600 at C5, i.e. <= 3.1e-6 of the pixels.
# Budget = 10x that (the round-2 budget was 1e-4), never less than 2 pixels.
N_CONTRIB_BUDGET = 3e-5


def n_contrib_ok(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return int((a != b).sum()) <= max(2, int(N_CONTRIB_BUDGET * a.size))


def assert_image_close(out, ref, tol, what="color", cap=5e-3):
    """max-abs bound on an image with the same allowance as n_contrib: a pair whose alpha sits within an ulp of 1/255 (or whose T sits at the
    saturation threshold) is blended on one side and skipped on the other, which moves that pixel by up to alpha * T ~ 4e-3 — at most
    N_CONTRIB_BUDGET of the pixels (never fewer than 2) may exceed `tol`, none may exceed `cap`."""
    d = np.abs(np.asarray(out, np.float64) - np.asarray(ref, np.float64))
    d = d.reshape(-1, d.shape[-2] * d.shape[-1]).max(axis=0)
    over = int((d > tol).sum())
    assert over <= max(2, int(N_CONTRIB_BUDGET * d.size)), (what, "pixels over %g" % tol, over, float(d.max()))
    assert float(d.max()) <= cap, (what, float(d.max()))


GATE_BUDGET = 1e-5      # admissible failing fraction of grad_gate (threshold flips of a pixel's contributor list move a few rows)


def assert_planes_psnr(out, ref, min_db=50.0, what="allmap"):
    """Per-plane PSNR with the plane's own peak (at least 1): a depth-scale plane does not hide errors in unit-scale ones."""
    for plane in range(ref.shape[0]):
        peak = max(1.0, float(np.abs(ref[plane]).max()))
        v = psnr(out[plane], ref[plane], peak=peak)
        assert v >= min_db, (what, plane, v)


def scene_kwargs(variant, P, W, H, seed, mu, sh_degree=3, bg=(0.0, 0.0, 0.0), mask_radius=0.0, cam=None, ball=False):
    cam = cam or S.make_camera(W, H)
    sc = S.make_scene(P, variant, seed=seed, mu=mu, mask_radius=mask_radius, ball=ball)
    kw = dict(bg=np.asarray(bg, np.float32), means3D=sc["means3D"], opacities=sc["opacities"], viewmatrix=cam["viewmatrix"],
              projmatrix=cam["projmatrix"], campos=cam["campos"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], image_height=H,
              image_width=W, sh_degree=sh_degree, shs=sc["shs"], refl_strengths=sc["refl_strengths"], scales=sc["scales"],
              rotations=sc["rotations"])
    if variant == "G":
        kw["normals"] = sc["normals"]
    else:
        kw["env_scope_mask"] = sc["env_scope_mask"]
    return kw, cam, sc


def to_cuda(kw):
    import torch
    out = {}
    for k, v in kw.items():
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(v).cuda()
        else:
            out[k] = v
    return out


class HipSurfel:
    """Runs variant S through diff_surfel_rasterization on cuda:0 and exposes outputs, workspace
    arrays and gradients as numpy."""

    def __init__(self, kw, requires_grad=True, scale_modifier=1.0, debug=False, prefiltered=False, make_sink=None):
        """make_sink(self) -> (sink dict, accumulate): called once the leaf tensors exist, before the forward (the gradient
        sink of a rasterizer is bound to the forward calls made while it is set)."""
        import torch
        from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _RasterizeGaussians
        t = to_cuda(kw)
        self.t = t
        self.P = t["means3D"].shape[0]
        self.H, self.W = int(kw["image_height"]), int(kw["image_width"])
        leaf = lambda x: x.clone().requires_grad_(requires_grad) if x is not None else None
        self.means3D = leaf(t["means3D"])
        self.means2D = torch.zeros_like(t["means3D"]).requires_grad_(requires_grad)
        self.opac = leaf(t["opacities"])
        self.shs = leaf(t.get("shs")) if kw.get("colors_precomp") is None else None
        self.colors = leaf(t.get("colors_precomp")) if kw.get("colors_precomp") is not None else None
        self.refl = leaf(t["refl_strengths"])
        self.scales = leaf(t.get("scales")) if kw.get("cov3D_precomp") is None else None
        self.rots = leaf(t.get("rotations")) if kw.get("cov3D_precomp") is None else None
        self.cov = leaf(t.get("cov3D_precomp")) if kw.get("cov3D_precomp") is not None else None
        st = GaussianRasterizationSettings(image_height=self.H, image_width=self.W, tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"],
                                           bg=t["bg"], scale_modifier=scale_modifier, viewmatrix=t["viewmatrix"],
                                           projmatrix=t["projmatrix"], sh_degree=kw["sh_degree"], campos=t["campos"],
                                           prefiltered=prefiltered, debug=debug)
        self.settings = st
        rast = GaussianRasterizer(st)
        if make_sink is not None:
            rast.set_grad_sink(*make_sink(self))
        self.color, self.radii, self.allmap, self.refl_map, self.gw = rast(
            means3D=self.means3D, means2D=self.means2D, opacities=self.opac, shs=self.shs, colors_precomp=self.colors,
            refl_strengths=self.refl, scales=self.scales, rotations=self.rots, cov3D_precomp=self.cov,
            env_scope_mask=t.get("env_scope_mask"))
        fn = self.color.grad_fn
        self.ctx = fn
        self.R = fn.num_rendered if fn is not None else None

    def out(self):
        return dict(color=self.color.detach().cpu().numpy(), radii=self.radii.cpu().numpy(), allmap=self.allmap.detach().cpu().numpy(),
                    refl_strength_map=self.refl_map.detach().cpu().numpy(), gaussian_weights=self.gw.cpu().numpy(), num_rendered=self.R)

    def state(self, name):
        import torch
        import _gsr
        geom, binning, img = self.ctx.saved_tensors[-3:]
        P, R, W, H = self.P, self.R, self.W, self.H
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        spec = {"depths": (torch.float32, (P,)), "means2D": (torch.float32, (P, 2)), "tiles_touched": (torch.int32, (P,)),
                "point_offsets": (torch.int32, (P,)), "clamped": (torch.uint8, (P, 3)), "rgb": (torch.float32, (P, 3)),
                "geom4": (torch.float32, (P, 4)), "transMat": (torch.float32, (P, 9)), "point_list": (torch.int32, (R,)),
                "keys": (torch.int64, (R,)), "ranges": (torch.int32, (tiles, 2)), "final_T": (torch.float32, (3, H, W)),
                "n_contrib": (torch.int32, (2, H, W))}[name]
        return _gsr.debug_fetch(0, name, P, R, W, H, geom, binning, img, spec[0], spec[1]).cpu().numpy()

    def backward(self, dL_dcolor, dL_dallmap=None, dL_drefl=None):
        import torch
        loss = (self.color * torch.from_numpy(dL_dcolor).cuda()).sum()
        if dL_dallmap is not None:
            loss = loss + (self.allmap * torch.from_numpy(dL_dallmap).cuda()).sum()
        if dL_drefl is not None:
            loss = loss + (self.refl_map * torch.from_numpy(dL_drefl).cuda()).sum()
        loss.backward()
        g = lambda x: None if x is None or x.grad is None else x.grad.detach().cpu().numpy()
        return dict(dL_dmeans3D=g(self.means3D), dL_dmeans2D=g(self.means2D), dL_dopacity=g(self.opac), dL_dsh=g(self.shs),
                    dL_dcolors=g(self.colors), dL_drefl_strengths=g(self.refl), dL_dscales=g(self.scales), dL_drotations=g(self.rots),
                    dL_dtransMat=g(self.cov))


class HipGauss:
    """Runs variant G through diff_gaussian_rasterization on cuda:0."""

    def __init__(self, kw, requires_grad=True, scale_modifier=1.0, antialiasing=False, debug=False, prefiltered=False, make_sink=None):
        """make_sink(self) -> (sink dict, accumulate): as HipSurfel."""
        import torch
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        t = to_cuda(kw)
        self.t = t
        self.P = t["means3D"].shape[0]
        self.H, self.W = int(kw["image_height"]), int(kw["image_width"])
        leaf = lambda x: x.clone().requires_grad_(requires_grad) if x is not None else None
        self.means3D = leaf(t["means3D"])
        self.means2D = torch.zeros_like(t["means3D"]).requires_grad_(requires_grad)
        self.opac = leaf(t["opacities"])
        self.shs = leaf(t.get("shs")) if kw.get("colors_precomp") is None else None
        self.colors = leaf(t.get("colors_precomp")) if kw.get("colors_precomp") is not None else None
        self.normals = leaf(t["normals"])
        self.refl = leaf(t["refl_strengths"])
        self.scales = leaf(t.get("scales")) if kw.get("cov3D_precomp") is None else None
        self.rots = leaf(t.get("rotations")) if kw.get("cov3D_precomp") is None else None
        self.cov = leaf(t.get("cov3D_precomp")) if kw.get("cov3D_precomp") is not None else None
        st = GaussianRasterizationSettings(image_height=self.H, image_width=self.W, tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"],
                                           bg=t["bg"], scale_modifier=scale_modifier, viewmatrix=t["viewmatrix"],
                                           projmatrix=t["projmatrix"], sh_degree=kw["sh_degree"], campos=t["campos"],
                                           prefiltered=prefiltered, debug=debug, antialiasing=antialiasing)
        rast = GaussianRasterizer(st)
        if make_sink is not None:
            rast.set_grad_sink(*make_sink(self))
        self.color, self.radii, self.invdepth, self.normal_map, self.refl_map = rast(
            means3D=self.means3D, means2D=self.means2D, opacities=self.opac, shs=self.shs, colors_precomp=self.colors,
            normals=self.normals, refl_strengths=self.refl, scales=self.scales, rotations=self.rots, cov3D_precomp=self.cov)
        fn = self.color.grad_fn
        self.ctx = fn
        self.R = fn.num_rendered if fn is not None else None

    def out(self):
        return dict(color=self.color.detach().cpu().numpy(), radii=self.radii.cpu().numpy(), invdepth=self.invdepth.detach().cpu().numpy(),
                    normal_map=self.normal_map.detach().cpu().numpy(), refl_strength_map=self.refl_map.detach().cpu().numpy(),
                    num_rendered=self.R)

    def state(self, name):
        import torch
        import _gsr
        geom, binning, img = self.ctx.saved_tensors[-3:]
        P, R, W, H = self.P, self.R, self.W, self.H
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        spec = {"depths": (torch.float32, (P,)), "means2D": (torch.float32, (P, 2)), "tiles_touched": (torch.int32, (P,)),
                "point_offsets": (torch.int32, (P,)), "clamped": (torch.uint8, (P, 3)), "rgb": (torch.float32, (P, 3)),
                "geom4": (torch.float32, (P, 4)), "point_list": (torch.int32, (R,)),
                "keys": (torch.int64, (R,)), "ranges": (torch.int32, (tiles, 2)), "final_T": (torch.float32, (1, H, W)),
                "n_contrib": (torch.int32, (1, H, W))}[name]
        return _gsr.debug_fetch(1, name, P, R, W, H, geom, binning, img, spec[0], spec[1]).cpu().numpy()

    def backward(self, dL_dcolor, dL_dinvdepth=None, dL_dnormal=None, dL_drefl=None):
        import torch
        loss = (self.color * torch.from_numpy(dL_dcolor).cuda()).sum()
        if dL_dinvdepth is not None:
            loss = loss + (self.invdepth * torch.from_numpy(dL_dinvdepth).cuda()).sum()
        if dL_dnormal is not None:
            loss = loss + (self.normal_map * torch.from_numpy(dL_dnormal).cuda()).sum()
        if dL_drefl is not None:
            loss = loss + (self.refl_map * torch.from_numpy(dL_drefl).cuda()).sum()
        loss.backward()
        g = lambda x: None if x is None or x.grad is None else x.grad.detach().cpu().numpy()
        return dict(dL_dmeans3D=g(self.means3D), dL_dmeans2D=g(self.means2D), dL_dopacity=g(self.opac), dL_dsh=g(self.shs),
                    dL_dcolors=g(self.colors), dL_dnormals=g(self.normals), dL_drefl_strengths=g(self.refl), dL_dscales=g(self.scales),
                    dL_drotations=g(self.rots), dL_dcov3D=g(self.cov))
