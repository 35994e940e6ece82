"""Full-size GPU parity of the fused rasterize + reflect node (gaussian_renderer.rasterize_reflect: the deferred reflection's forward as
the epilogue of the rasterizer's tile kernel, both backwards in one autograd node) exactly as bench.py drives it since round 4 — gradient
sinks into one flat buffer, asynchronous texel-gradient tail:

  * C3 (10^6 surfels, 1920x1080, L = 128): final image / reflection colour / shading normal against the reference's op-by-op chain in
    float64 on the oracle's cubemap (gaussian_renderer/__init__.py:22-35,148,178-199 of the reference; CME cubemapencoder.cu:298-334),
    cubemap gradient against the chain's, parameter gradients against the oracle's rasterizer backward fed with the chain's gradients;
  * C4 with the geometry SURVEY.md 8d fixes for it (10^6 surfels in the ball of radius 2, cameras on a circle of radius 5): two views of
    the batch, first overwriting the flat buffer, second adding to it, against the sum of the oracle's two backwards.
"""
import numpy as np
import pytest
import torch

import gaussian_renderer
from helpers import GATE_BUDGET, S, grad_gate, psnr, rel_maxnorm, scene_kwargs
from test_gpu_fullsize_chain import NAMES, ORACLE_NAMES, H, P, W, _Env, _chain_reference, _view

pytestmark = pytest.mark.gpu


class _Scene:
    def __init__(self, L, seed, mu, ball):
        from gsr_dist import FlatGrads
        self.kw, self.cam, sc = scene_kwargs("S", P, W, H, seed, mu, 3, (0, 0, 0), ball=ball)
        self.tex, self.fail = S.make_cubemap(L, 3, seed)
        src = {k: torch.from_numpy(sc[k]) for k in NAMES}
        src["cubemap"], src["fail"] = torch.from_numpy(self.tex), torch.from_numpy(self.fail)
        self.p = {k: v.cuda().requires_grad_(True) for k, v in src.items()}
        self.grads = FlatGrads(self.p)
        self.mask = torch.from_numpy(sc["env_scope_mask"]).cuda()
        self.means2D = torch.zeros(P, 3, device="cuda", requires_grad=True)
        self.env = _Env(self.p["cubemap"], self.p["fail"])


def _fused(sc, rast, ct, cam, accumulate):
    from gaussian_renderer import rasterize_reflect
    rast.set_grad_sink(sc.grads.sink(), accumulate=accumulate)
    sc.means2D.grad = None
    return rasterize_reflect(rast, sc.env, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"], means3D=sc.p["means3D"], means2D=sc.means2D,
                             opacities=sc.p["opacities"], shs=sc.p["shs"], refl_strengths=sc.p["refl_strengths"], scales=sc.p["scales"],
                             rotations=sc.p["rotations"], env_scope_mask=sc.mask, refl_grad_sink=sc.grads.sink(names=("cubemap", "fail")),
                             accumulate=accumulate, async_tail=True)


def _oracle_view(sc, cam, base, allmap, refl_map, g, L, probe):
    """float64 chain on the HIP rasterizer's outputs; the pixel gradients the node's reflection backward handed its rasterizer backward
    (`probe`) against the chain's (same tolerances as tests/test_gpu_fullsize_chain.py: the normal gradient is piecewise constant in the
    direction, a budget of pixels lands in the neighbouring texel cell); then the oracle's rasterizer backward with THOSE gradients as
    upstream, as the two-node test does through autograd hooks."""
    from oracle import oracle as orc
    npy = lambda t: t.detach().cpu().numpy()
    ref = _chain_reference(npy(allmap[2:5]), npy(base), npy(refl_map), sc.tex, sc.fail, cam, g["dL_dcolor"], None, None)
    g_nv, g_base, g_s = npy(probe["g_normal_view"]), npy(probe["g_base"]), npy(probe["g_strength"])
    bad = np.abs(g_nv - ref[3]).max(axis=0) > 1e-3 * np.abs(ref[3]).max()
    assert bad.mean() <= 2e-3, bad.mean()
    assert rel_maxnorm(g_base, ref[4]) <= 1e-5 and rel_maxnorm(g_s, ref[5]) <= 1e-4
    kw = dict(sc.kw)
    for k in ("viewmatrix", "projmatrix", "campos"):
        kw[k] = cam[k]
    o = orc.SurfelOracle(np.float32)
    fo = o.forward(**kw)
    planes = g["dL_dplanes"].copy()
    planes[2:5] += g_nv
    gr = o.backward(dL_dcolor=g_base, dL_dallmap=planes, dL_drefl_strength_map=g_s)
    return ref, fo, gr


def test_c3_fused_step_against_chain_and_oracle():
    import _gsr
    L = 128
    sc = _Scene(L, 1003, -4.75, False)
    rast, ct = _view(sc.cam)
    g = S.make_upstream_grads(H, W, 1003)
    to_c = lambda a: torch.from_numpy(a).cuda()
    sc.grads.flat.fill_(float("nan"))
    final, refl_color, nrm, base, radii, allmap, refl_map, gw = _fused(sc, rast, ct, sc.cam, False)
    probe = {}
    gaussian_renderer._RasterizeReflect.probe = probe
    try:
        torch.autograd.backward([final, allmap], [to_c(g["dL_dcolor"]), to_c(g["dL_dplanes"])])
    finally:
        gaussian_renderer._RasterizeReflect.probe = None
    _gsr.side_join()
    torch.cuda.synchronize()
    assert all(torch.isfinite(sc.grads.view(k)).all() for k in sc.grads.slices)
    ref, fo, gr = _oracle_view(sc, sc.cam, base, allmap, refl_map, g, L, probe)
    npy = lambda t: t.detach().cpu().numpy()
    np.testing.assert_allclose(npy(final), ref[0], atol=2e-5)
    np.testing.assert_allclose(npy(refl_color), ref[1], atol=2e-5)
    np.testing.assert_allclose(npy(nrm), ref[2], atol=2e-5)
    assert fo["num_rendered"] == final.grad_fn.num_rendered and (npy(radii) == fo["radii"]).all() and psnr(npy(base), fo["color"]) >= 50
    assert rel_maxnorm(npy(sc.grads.view("cubemap")), ref[6]) <= 1e-4
    for k in NAMES:
        got = npy(sc.grads.view(k))
        want = gr[ORACLE_NAMES[k]].reshape(got.shape)
        assert rel_maxnorm(got, want) <= 1e-4, k
        assert grad_gate(got, want) <= GATE_BUDGET, (k, "elementwise gate")


def test_c4_ball_and_circle_cameras_two_views_against_oracle():
    """Views 0 and 3 of the C4 batch of bench.py's c4_one_gpu line (cameras on the circle, scene = ball): overwrite, then add."""
    import _gsr
    L = 128
    sc = _Scene(L, 1004, -4.75, True)
    cams = S.circle_cameras(W, H, 8)
    g = S.make_upstream_grads(H, W, 1003)
    to_c = lambda a: torch.from_numpy(a).cuda()
    sc.grads.flat.fill_(float("nan"))
    total = {k: 0.0 for k in NAMES}
    tex_total = 0.0
    for i, v in enumerate((0, 3)):
        rast, ct = _view(cams[v])
        final, refl_color, nrm, base, radii, allmap, refl_map, gw = _fused(sc, rast, ct, cams[v], i > 0)
        probe = {}
        gaussian_renderer._RasterizeReflect.probe = probe
        try:
            torch.autograd.backward([final, allmap], [to_c(g["dL_dcolor"]), to_c(g["dL_dplanes"])])
        finally:
            gaussian_renderer._RasterizeReflect.probe = None
        ref, fo, gr = _oracle_view(sc, cams[v], base, allmap, refl_map, g, L, probe)
        assert fo["num_rendered"] == final.grad_fn.num_rendered and psnr(base.detach().cpu().numpy(), fo["color"]) >= 50, v
        assert float((radii > 0).float().mean()) > 0.9           # the ball is in view
        for k in NAMES:
            total[k] = total[k] + gr[ORACLE_NAMES[k]].astype(np.float64)
        tex_total = tex_total + ref[6]
    _gsr.side_join()
    torch.cuda.synchronize()
    for k in NAMES:
        got = sc.grads.view(k).cpu().numpy()
        want = total[k].reshape(got.shape)
        assert rel_maxnorm(got, want) <= 1e-4, k
        assert grad_gate(got, want) <= 2 * GATE_BUDGET, (k, "elementwise gate")       # (two views' budgets of threshold flips)
    assert rel_maxnorm(sc.grads.view("cubemap").cpu().numpy(), tex_total) <= 1e-4
