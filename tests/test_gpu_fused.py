"""The fused rasterize + reflect path (round 4): gaussian_renderer.rasterize_reflect / gsr_surfel_forward_refl / gsr_surfel_backward_refl
run the deferred-reflection pixel code (gaussian_renderer/__init__.py:22-35,143-199 of the reference) inside the rasterizer's tile
kernels.  It must give what the two-node path — GaussianRasterizer, then deferred_reflection(), each checked against the oracle and the
float64 chain elsewhere — gives: the rasterizer's own outputs bit for bit (the same kernel code produced them), the reflection outputs and
every gradient to rounding (the arithmetic is the same text compiled into another kernel; atomics land in another order)."""
import numpy as np
import pytest
import torch

from helpers import S, rel_maxnorm

pytestmark = pytest.mark.gpu


def _scene(P, seed, mu, L, W, H, cam=None, bg=(0.1, 0.2, 0.3)):
    sc = S.make_scene(P, "S", seed=seed, mu=mu)
    tex, fail = S.make_cubemap(L, 3, seed)
    cam = cam or S.look_at_camera(W, H, eye=(0.3, -0.2, -0.8), target=(0, 0, 5))
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    names = ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths")
    src = {k: torch.from_numpy(sc[k]).cuda() for k in names}
    src["cubemap"] = torch.from_numpy(tex).cuda()
    src["fail"] = torch.from_numpy(fail).cuda() + 0.25
    mask = torch.from_numpy(sc["env_scope_mask"]).cuda()
    return src, mask, cam, ct, torch.tensor(bg, device="cuda")


class _Env:
    def __init__(self, tex, fail):
        self.params = {"Cubemap_texture": tex, "Cubemap_failv": fail}


def _rasterizer(cam, ct, W, H, bg):
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    return GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg,
                                                            scale_modifier=1.0, viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"], sh_degree=3,
                                                            campos=ct["campos"], prefiltered=False, debug=False))


def _run(fused, p, mask, cam, ct, W, H, bg, ups, raster_sink=None, refl_sink=None, accumulate=False, async_tail=False):
    """One forward + backward through either path with the same upstream gradients on every output; returns outputs and means2D.grad."""
    from gaussian_renderer import deferred_reflection, rasterize_reflect
    rast = _rasterizer(cam, ct, W, H, bg)
    rast.set_grad_sink(raster_sink, accumulate)
    means2D = torch.zeros_like(p["means3D"]).requires_grad_(True)
    env = _Env(p["cubemap"], p["fail"])
    HWK = (H, W, cam["K"])
    kw = dict(means3D=p["means3D"], means2D=means2D, opacities=p["opacities"], shs=p["shs"], refl_strengths=p["refl_strengths"], scales=p["scales"],
              rotations=p["rotations"], env_scope_mask=mask)
    if fused:
        final, refl_color, nworld, base, radii, allmap, refl_map, gw = rasterize_reflect(rast, env, ct["viewmatrix"], HWK, ct["R"], ct["T"], refl_grad_sink=refl_sink,
                                                                                         accumulate=accumulate, async_tail=async_tail, **kw)
    else:
        rast.set_output_taps(("normal_view",))
        base, radii, allmap, refl_map, gw, nview = rast(**kw)
        final, refl_color, nworld = deferred_reflection(nview, base, refl_map, env, ct["viewmatrix"], HWK, ct["R"], ct["T"], grad_sink=refl_sink,
                                                        accumulate=accumulate, async_tail=async_tail)
    out = dict(final=final, refl_color=refl_color, nworld=nworld, base=base, radii=radii, allmap=allmap, refl_map=refl_map, gw=gw)
    if ups is not None:
        loss = sum((out[k] * g).sum() for k, g in ups.items())
        loss.backward()
    return out, means2D.grad


def _upstream(H, W, seed, which=("final", "refl_color", "nworld", "base", "allmap", "refl_map")):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    shapes = dict(final=(3, H, W), refl_color=(3, H, W), nworld=(3, H, W), base=(3, H, W), allmap=(8, H, W), refl_map=(1, H, W))
    return {k: (torch.randn(shapes[k], generator=gen) / (H * W)).cuda() for k in which}


PARAMS = ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths", "cubemap", "fail")


@pytest.mark.parametrize("W,H,which", [(400, 240, None), (301, 203, ("final",)), (200, 120, ("final", "allmap"))])
def test_fused_equals_rasterizer_then_reflection(W, H, which):
    """Plain autograd, every output with its own upstream gradient (first case), only the final image (second: every direct gradient of the
    rasterizer's outputs is NULL at the C ABI; ragged tile edges: lanes without a pixel take part in the wave-cooperative rim loop) or the
    training loop's pair (third)."""
    src, mask, cam, ct, bg = _scene(30_000, 41, -3.3, 32, W, H)
    ups = _upstream(H, W, 3, which) if which else _upstream(H, W, 3)
    res = {}
    for fused in (False, True):
        p = {k: v.clone().requires_grad_(True) for k, v in src.items()}
        out, g2d = _run(fused, p, mask, cam, ct, W, H, bg, ups)
        res[fused] = (out, {k: p[k].grad for k in PARAMS}, g2d)
    (ou, gu, mu_), (of, gf, mf) = res[False], res[True]
    for k in ("base", "radii", "allmap", "refl_map", "gw"):
        assert torch.equal(ou[k], of[k]), k                                  # the rasterizer's own outputs: the same bits
    for k in ("final", "refl_color", "nworld"):
        assert float((ou[k] - of[k]).abs().max()) <= 5e-6, k      # (sigmoid outputs ~0.5: a few ulp between the two compilations of the same text)
    assert float(of["final"].abs().max()) > 0.1 and float(of["allmap"][1].max()) > 0.5
    for k in PARAMS:
        a, b = gf[k].cpu().numpy(), gu[k].cpu().numpy()
        assert np.isfinite(a).all() and (k == "fail" or np.abs(b).max() > 0), k       # (no pixel of these views has a zero reflection vector)
        assert rel_maxnorm(a, b) <= 5e-5, k
    assert rel_maxnorm(mf.cpu().numpy(), mu_.cpu().numpy()) <= 5e-5


def test_fused_without_autograd_and_with_an_empty_scene():
    """no_grad: no sort keys are written and nothing is kept; P = 0: the stand-alone pixel pass on zero planes."""
    W, H = 160, 96
    src, mask, cam, ct, bg = _scene(2000, 42, -2.8, 16, W, H)
    with torch.no_grad():
        of, _ = _run(True, src, mask, cam, ct, W, H, bg, None)
        ou, _ = _run(False, src, mask, cam, ct, W, H, bg, None)
    for k in ("base", "allmap", "refl_map"):
        assert torch.equal(ou[k], of[k]), k
    assert float((ou["final"] - of["final"]).abs().max()) <= 5e-6
    empty = {k: (v[:0] if k not in ("cubemap", "fail") else v) for k, v in src.items()}
    with torch.no_grad():
        oe, _ = _run(True, empty, mask[:0], cam, ct, W, H, bg, None)
        ue, _ = _run(False, empty, mask[:0], cam, ct, W, H, bg, None)
    assert float(oe["allmap"].abs().max()) == 0.0
    assert float((oe["final"] - ue["final"]).abs().max()) <= 5e-6


@pytest.mark.parametrize("async_tail", [False, True])
def test_fused_with_sinks_accumulates_two_views_like_the_two_node_path(async_tail):
    """As bench.py drives a batch: both sinks into one flat buffer, first view overwriting, second adding, the texel-gradient tail on the
    side stream; the fused path against the two-node path, and both against the sum of two plain-autograd views."""
    import _gsr
    from gsr_dist import FlatGrads
    W, H = 320, 200
    src, mask, cam, ct, bg = _scene(20_000, 43, -3.2, 32, W, H)
    cam2 = S.look_at_camera(W, H, eye=(-0.5, 0.3, -0.6), target=(0, 0, 5))
    ct2 = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam2.items() if isinstance(v, np.ndarray)}
    ups = _upstream(H, W, 4, ("final", "allmap"))
    plain = None
    for c, t in ((cam, ct), (cam2, ct2)):
        p = {k: v.clone().requires_grad_(True) for k, v in src.items()}
        _run(False, p, mask, c, t, W, H, bg, ups)
        g = {k: p[k].grad.clone() for k in PARAMS}
        plain = g if plain is None else {k: plain[k] + g[k] for k in PARAMS}
    flat = {}
    for fused in (False, True):
        p = {k: v.clone().requires_grad_(True) for k, v in src.items()}
        fg = FlatGrads(p)
        fg.flat.fill_(float("nan"))
        for i, (c, t) in enumerate(((cam, ct), (cam2, ct2))):
            _run(fused, p, mask, c, t, W, H, bg, ups, raster_sink=fg.sink(), refl_sink=fg.sink(names=("cubemap", "fail")), accumulate=i > 0,
                 async_tail=async_tail)
        _gsr.side_join()
        torch.cuda.synchronize()
        flat[fused] = {k: fg.view(k).clone() for k in PARAMS}
    for k in PARAMS:
        a, b, c = flat[True][k].cpu().numpy(), flat[False][k].cpu().numpy(), plain[k].cpu().numpy()
        assert np.isfinite(a).all(), k
        assert rel_maxnorm(a, b) <= 5e-5, k
        assert rel_maxnorm(a, c) <= 5e-5, k


def test_render_uses_the_fused_path_and_matches_the_two_node_render():
    """gaussian_renderer.render(): FUSED_REFLECTION on (default) against off, outputs and the gradients of a training-like loss."""
    import gaussian_renderer as gr
    from cubemapencoder import CubemapEncoder
    W, H, L = 320, 200, 32
    src, mask, cam, ct, bg = _scene(20_000, 44, -3.2, L, W, H)

    class View:
        FoVx, FoVy, image_width, image_height = cam["FoVx"], cam["FoVy"], W, H
        world_view_transform, full_proj_transform, camera_center = ct["viewmatrix"], ct["projmatrix"], ct["campos"]
        HWK, R, T, znear, zfar = (H, W, cam["K"]), ct["R"], ct["T"], cam["znear"], cam["zfar"]

    class Pipe:
        depth_ratio, compute_cov3D_python = 0.0, False
    res = {}
    for fused in (False, True):
        env = CubemapEncoder(output_dim=3, resolution=L).cuda()
        with torch.no_grad():
            env.params["Cubemap_texture"].copy_(src["cubemap"])
            env.params["Cubemap_failv"].copy_(src["fail"])
        t = {k: src[k].clone().requires_grad_(True) for k in PARAMS[:6]}

        class PC:
            get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_refl = (t["means3D"], t["opacities"], t["scales"], t["rotations"], t["shs"],
                                                                                       t["refl_strengths"])
            active_sh_degree, get_envmap = 3, env
        gr.FUSED_REFLECTION = fused
        try:
            pkg = gr.render(View, PC, Pipe, bg)
        finally:
            gr.FUSED_REFLECTION = True
        loss = pkg["render"].square().mean() + 0.05 * (1 - (pkg["rend_normal"] * pkg["surf_normal"]).sum(dim=0)).mean() + 0.1 * pkg["rend_dist"].mean()
        loss.backward()
        grads = {k: v.grad for k, v in t.items()}
        grads["cubemap"], grads["fail"] = env.params["Cubemap_texture"].grad, env.params["Cubemap_failv"].grad
        grads["viewspace"] = pkg["viewspace_points"].grad
        res[fused] = (pkg, grads)
    (pu, gu), (pf, gf) = res[False], res[True]
    for k in ("rend_alpha", "rend_dist", "surf_depth", "surf_normal", "radii", "gaussian_weights", "base_color_map", "refl_strength_map"):
        assert torch.equal(pu[k], pf[k]), k
    for k in ("render", "rend_normal", "refl_color_map"):
        assert float((pu[k] - pf[k]).abs().max()) <= 5e-6, k
    for k in gu:
        assert rel_maxnorm(gf[k].cpu().numpy(), gu[k].cpu().numpy()) <= 5e-5, k
