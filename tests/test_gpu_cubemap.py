"""GPU parity: cubemap encoder (all three interpolation modes) and the fused deferred-reflection pass vs the
CPU oracle (oracle/oracle_cubemap.cpp) and vs the reference's own op-by-op composition
(gaussian_renderer/__init__.py:22-35,148,178-179,197-199) evaluated on the CPU in float64."""
import numpy as np
import pytest
import torch

from helpers import S, rel_maxnorm

pytestmark = pytest.mark.gpu


def _dirs(B, seed, L):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(B, 3, generator=g)
    # add exact face centres, edges, corners and the zero vector (fail value path)
    special = torch.tensor([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [1, -1, 0], [-1, 1, 0.0],
                            [1, 1, 1], [-1, 1, 1], [1, -1, -1], [-1, -1, -1], [0, 0, 0], [1, 0.999, 0.2], [0.3, 1, 0.9995], [1, 1 - 1.0 / L, 1 - 1.0 / L]],
                           dtype=torch.float32)
    # directions hugging cube edges / corners so that the seamless branches are hit often
    near = torch.sign(torch.randn(B // 4, 3, generator=g)) * (1 - 0.02 * torch.rand(B // 4, 3, generator=g))
    return torch.cat([d, special, near], 0).contiguous()


@pytest.mark.parametrize("interp,seamless", [(0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("L,C", [(16, 3), (8, 5)])
def test_cubemap_forward_backward_vs_oracle(interp, seamless, L, C):
    from oracle import oracle as orc
    from cubemapencoder.cubemap_encoder import _backend
    d = _dirs(20000, 3, L)
    B = d.shape[0]
    g = torch.Generator().manual_seed(5)
    cm = (torch.rand(6, C, L, L, generator=g) - 0.5)
    fv = torch.randn(C, generator=g)
    go = torch.randn(C, B, generator=g)
    out = torch.empty(C, B, device="cuda")
    _backend.cubemap_encode_forward(d.cuda(), cm.cuda(), fv.cuda(), out, interp, seamless, B, C, L)
    ref = orc.cubemap_forward(d.numpy(), cm.numpy(), fv.numpy(), interp, seamless)
    # -use_fast_math in the reference build: tolerance covers approximate division
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
    gcm = torch.zeros_like(cm, device="cuda")
    gin = torch.empty(B, 3, device="cuda")
    gf = torch.zeros(C, device="cuda")
    _backend.cubemap_encode_backward(go.cuda(), d.cuda(), cm.cuda(), gcm, gin, gf, interp, seamless, B, C, L)
    rin, rcm, rf = orc.cubemap_backward(go.numpy(), d.numpy(), cm.numpy(), interp, seamless)
    assert rel_maxnorm(gcm.cpu().numpy(), rcm) <= 1e-5
    assert rel_maxnorm(gf.cpu().numpy(), rf) <= 1e-5
    assert rel_maxnorm(gin.cpu().numpy(), rin) <= 1e-4


def test_cubemap_encoder_module_autograd():
    from oracle import oracle as orc
    from cubemapencoder import CubemapEncoder
    enc = CubemapEncoder(output_dim=3, resolution=32).cuda()
    d = _dirs(5000, 9, 32).cuda().requires_grad_(True)
    y = enc(d)
    assert y.shape == (d.shape[0], 3)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    tex = enc.params["Cubemap_texture"].detach().cpu().numpy()
    fail = enc.params["Cubemap_failv"].detach().cpu().numpy()
    rin, rcm, rf = orc.cubemap_backward(w.t().contiguous().cpu().numpy(), d.detach().cpu().numpy(), tex, 1, 1)
    assert rel_maxnorm(enc.params["Cubemap_texture"].grad.cpu().numpy(), rcm) <= 1e-5
    assert rel_maxnorm(d.grad.cpu().numpy(), rin) <= 1e-4


from helpers_chain import OracleCubemap as _OracleCubemap, reference_chain as _reference_chain  # noqa: E402,F401


@pytest.mark.parametrize("binned", [True, False])
def test_fused_deferred_reflection_vs_reference_chain(binned, monkeypatch):
    """Both backward paths of the fused op: texel gradients binned by cube-face band (default) and by float atomics."""
    import gaussian_renderer
    from gaussian_renderer import deferred_reflection
    monkeypatch.setattr(gaussian_renderer, "REFLECTION_BACKWARD_BINNED", binned)
    W, H, L = 160, 96, 16
    cam = S.look_at_camera(W, H, eye=(1.0, -0.5, -4.0))
    g = torch.Generator().manual_seed(11)
    nv = torch.randn(3, H, W, generator=g) * torch.rand(1, H, W, generator=g)
    nv[:, :4, :4] = 0.0  # zero normals: r = d, exercises len = 0
    base = torch.rand(3, H, W, generator=g)
    strength = torch.rand(1, H, W, generator=g)
    tex, fail = S.make_cubemap(L, 3, 4)
    wf, wc, wn = torch.randn(3, H, W, generator=g), torch.randn(3, H, W, generator=g), torch.randn(3, H, W, generator=g)

    # reference chain, float64 CPU
    leaf = lambda x: x.double().clone().requires_grad_(True)
    nv_r, base_r, s_r, tex_r, fail_r = leaf(nv), leaf(base), leaf(strength), leaf(torch.from_numpy(tex)), leaf(torch.from_numpy(fail))
    f_r, c_r, n_r = _reference_chain(nv_r, base_r, s_r, tex_r, fail_r, cam, W, H)
    ((f_r * wf.double()).sum() + (c_r * wc.double()).sum() + (n_r * wn.double()).sum()).backward()

    # fused HIP op
    class Env:
        pass
    cu = lambda x: x.float().cuda().clone().requires_grad_(True)
    nv_h, base_h, s_h, tex_h, fail_h = cu(nv), cu(base), cu(strength), cu(torch.from_numpy(tex)), cu(torch.from_numpy(fail))
    env = Env()
    env.params = {"Cubemap_texture": tex_h, "Cubemap_failv": fail_h}
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    f_h, c_h, n_h = deferred_reflection(nv_h, base_h, s_h, env, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"])
    ((f_h * wf.cuda()).sum() + (c_h * wc.cuda()).sum() + (n_h * wn.cuda()).sum()).backward()

    # forward: texel selection is discontinuous in the direction at texel borders only through weights (continuous),
    # so float32 vs float64 agree to rounding
    np.testing.assert_allclose(f_h.detach().cpu().numpy(), f_r.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(c_h.detach().cpu().numpy(), c_r.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(n_h.detach().cpu().numpy(), n_r.detach().numpy(), atol=2e-5)
    assert rel_maxnorm(base_h.grad.cpu().numpy(), base_r.grad.numpy()) <= 1e-5
    assert rel_maxnorm(s_h.grad.cpu().numpy(), s_r.grad.numpy()) <= 1e-4
    assert rel_maxnorm(tex_h.grad.cpu().numpy(), tex_r.grad.numpy()) <= 1e-4
    # the normal gradient passes through d(texel weights)/d(direction), piecewise constant in the direction: pixels whose
    # float32 direction lands in a different texel cell than the float64 one differ; allow a small budget of such pixels
    gn_h, gn_r = nv_h.grad.cpu().numpy(), nv_r.grad.numpy()
    bad = np.abs(gn_h - gn_r).max(axis=0) > 1e-3 * np.abs(gn_r).max()
    assert bad.mean() <= 2e-3, bad.mean()


def test_fused_matches_composed_hip_path():
    """The fused kernel and the un-fused composition (torch float32 ops on the GPU + the CubemapEncoder HIP kernels, chain in
    tests/helpers_chain.py) must agree; so must the stand-alone shading-normal kernel of the initial stage."""
    from cubemapencoder import CubemapEncoder
    from gaussian_renderer import deferred_reflection, shading_normal
    from helpers_chain import reflection_chain
    W, H, L = 200, 120, 32
    cam = S.make_camera(W, H)
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    g = torch.Generator().manual_seed(2)
    nv = torch.randn(3, H, W, generator=g).cuda()
    base = torch.rand(3, H, W, generator=g).cuda()
    s = torch.rand(1, H, W, generator=g).cuda()
    enc = CubemapEncoder(output_dim=3, resolution=L).cuda()
    f_h, c_h, n_h = deferred_reflection(nv, base, s, enc, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"])
    f_c, c_c, n_c = reflection_chain(nv, base, s, enc, ct["viewmatrix"], H, W, cam["K"], ct["R"], ct["T"])
    # float32 on both sides; a direction a few ulps apart may pick neighbouring texel weights: tolerance, not bit equality
    assert (torch.abs(f_h - f_c) > 1e-3).float().mean().item() <= 1e-3
    assert (torch.abs(n_h - n_c) > 1e-5).float().mean().item() == 0.0
    # initial stage: shading normal alone, forward and backward against the torch chain
    nv1 = nv.clone().requires_grad_(True)
    nv2 = nv.clone().requires_grad_(True)
    w = torch.randn(3, H, W, generator=g).cuda()
    n1 = shading_normal(nv1, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"])
    (n1 * w).sum().backward()
    _, _, n2 = reflection_chain(nv2, base, s, enc, ct["viewmatrix"], H, W, cam["K"], ct["R"], ct["T"])
    (n2 * w).sum().backward()
    assert (n1.detach() - n_h).abs().max().item() <= 1e-6      # two kernels, same formula (contraction may differ)
    assert (n1 - n2).abs().max().item() <= 1e-5
    assert rel_maxnorm(nv1.grad.cpu().numpy(), nv2.grad.cpu().numpy()) <= 1e-4


@pytest.mark.parametrize("L,W,H", [(128, 320, 200), (256, 320, 200), (600, 160, 96), (256, 64, 64), (16, 333, 77)])
def test_reflection_backward_paths_agree(L, W, H, monkeypatch):
    """The sorted-footprint backward against the float-atomics one across the sort configurations it selects by cubemap
    size (9-bit digits at L=128, 10-bit at L=256, the 8-bit default otherwise), with records beyond the LDS window
    (few pixels on a large cubemap) and with an odd pixel count."""
    import gaussian_renderer
    from gaussian_renderer import deferred_reflection
    cam = S.look_at_camera(W, H, eye=(0.5, -0.25, -3.0))
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    g = torch.Generator().manual_seed(L + W)
    nv = torch.randn(3, H, W, generator=g) * torch.rand(1, H, W, generator=g)
    nv[:, : H // 3] = 0.0   # a smooth region (r = d): long runs of equal texels
    base, strength = torch.rand(3, H, W, generator=g), torch.rand(1, H, W, generator=g)
    tex = torch.randn(6, 3, L, L, generator=g) * 0.5
    fail = torch.zeros(3)
    wf, wc, wn = torch.randn(3, H, W, generator=g), torch.randn(3, H, W, generator=g), torch.randn(3, H, W, generator=g)

    class Env:
        pass
    # float64 reference chain on the CPU
    leaf = lambda x: x.double().clone().requires_grad_(True)
    nv_r, base_r, s_r, tex_r, fail_r = leaf(nv), leaf(base), leaf(strength), leaf(tex), leaf(fail)
    f_r, c_r, n_r = _reference_chain(nv_r, base_r, s_r, tex_r, fail_r, cam, W, H)
    ((f_r * wf.double()).sum() + (c_r * wc.double()).sum() + (n_r * wn.double()).sum()).backward()
    grads = {}
    for binned in (True, "keys written by the backward", False):
        monkeypatch.setattr(gaussian_renderer, "REFLECTION_BACKWARD_BINNED", bool(binned))
        monkeypatch.setattr(gaussian_renderer, "REFLECTION_FORWARD_KEYS", binned is True)
        cu = lambda x: x.float().cuda().clone().requires_grad_(True)
        nv_h, base_h, s_h, tex_h, fail_h = cu(nv), cu(base), cu(strength), cu(tex), cu(fail)
        env = Env()
        env.params = {"Cubemap_texture": tex_h, "Cubemap_failv": fail_h}
        f_h, c_h, n_h = deferred_reflection(nv_h, base_h, s_h, env, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"])
        ((f_h * wf.cuda()).sum() + (c_h * wc.cuda()).sum() + (n_h * wn.cuda()).sum()).backward()
        grads[binned] = [x.grad.cpu().numpy() for x in (nv_h, base_h, s_h, tex_h)]
        # the texel gradient of each path against float64 (pixels whose float32 direction falls into the neighbouring
        # texel cell move their contribution by one texel: a handful of texels, bounded by the max-norm tolerance)
        # (the bilinear weights are differences of texel coordinates ~L/2: their float32 resolution grows with L)
        assert rel_maxnorm(grads[binned][3], tex_r.grad.numpy()) <= (1e-4 if L <= 256 else 3e-4)
        assert rel_maxnorm(grads[binned][1], base_r.grad.numpy()) <= 1e-5
    # sort keys from the forward kernel or from the backward's pixel kernel: the same stable sort of the same keys, bit for bit
    for a, b in zip(grads[True], grads["keys written by the backward"]):
        assert np.array_equal(a, b) or rel_maxnorm(a, b) <= 1e-6     # (rim pixels add with float atomics: arrival order)
    # same formulas in two kernels; contraction order differs and 1/|n| amplifies it.  The NORMAL gradient passes through d(texel
    # weights)/d(direction), piecewise constant in the direction: a pixel whose float32 direction lands in the neighbouring texel cell in
    # one of the two kernels differs outright (one of 15 360 at L = 600 since round 4, when the footprint kernel's divisions became
    # v_rcp): budgeted per pixel as in the float64 comparisons of this file, max-norm for the other two
    bad = np.abs(grads[True][0] - grads[False][0]).max(axis=0) > 1e-3 * np.abs(grads[False][0]).max()
    assert bad.mean() <= 2e-3, bad.mean()
    keep = ~bad
    assert rel_maxnorm(grads[True][0][:, keep], grads[False][0][:, keep]) <= 2e-4
    for a, b in zip(grads[True][1:3], grads[False][1:3]):
        assert rel_maxnorm(a, b) <= 1e-4
    # the two kernels contract the direction arithmetic differently: texel coordinates an ulp apart, times L
    assert rel_maxnorm(grads[True][3], grads[False][3]) <= (1e-4 if L <= 256 else 3e-4)


def test_reflection_grad_sink_routes_cubemap_gradients():
    """Extension: with a gradient sink the cubemap / fail-value gradients of THAT call land in the caller's tensors
    (overwritten: the buffers start as NaN; or added in accumulate mode) and autograd leaves the leaves' .grad alone;
    a call without a sink made in between is unaffected (the sink is per call, there is no module state)."""
    from gaussian_renderer import deferred_reflection
    W, H, L = 160, 96, 16
    cam = S.look_at_camera(W, H, eye=(1.0, -0.5, -4.0))
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    g = torch.Generator().manual_seed(3)
    nv = (torch.randn(3, H, W, generator=g) * torch.rand(1, H, W, generator=g)).cuda()
    nv[:, :2, :2] = 0.0   # zero normals: r = d
    base, strength = torch.rand(3, H, W, generator=g).cuda(), torch.rand(1, H, W, generator=g).cuda()
    tex0, fail0 = S.make_cubemap(L, 3, 4)
    wf = torch.randn(3, H, W, generator=g).cuda()

    class Env:
        pass

    def run(sink, accumulate=False, scale=1.0):
        tex = torch.from_numpy(tex0).cuda().requires_grad_(True)
        fail = torch.from_numpy(fail0).cuda().requires_grad_(True)
        env = Env()
        env.params = {"Cubemap_texture": tex, "Cubemap_failv": fail}
        f, _, _ = deferred_reflection(nv, base, strength, env, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"], grad_sink=sink,
                                      accumulate=accumulate)
        (f * wf * scale).sum().backward()
        return tex, fail
    tex_p, fail_p = run(None)
    sink = {"cubemap": torch.full((6, 3, L, L), float("nan"), device="cuda"), "fail": torch.full((3,), float("nan"), device="cuda")}
    tex_s, fail_s = run(sink)
    tex_q, fail_q = run(None)           # no sink on this call: plain autograd again
    assert tex_s.grad is None and fail_s.grad is None
    assert rel_maxnorm(tex_q.grad.cpu().numpy(), tex_p.grad.cpu().numpy()) <= 1e-5      # (atomics order differs run to run)
    assert torch.isfinite(sink["cubemap"]).all() and torch.isfinite(sink["fail"]).all()
    assert rel_maxnorm(sink["cubemap"].cpu().numpy(), tex_p.grad.cpu().numpy()) <= 1e-5
    np.testing.assert_allclose(sink["fail"].cpu().numpy(), fail_p.grad.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # accumulate: zero once, two backwards (weights 1 and 0.5) add up on the device
    acc = {"cubemap": torch.zeros((6, 3, L, L), device="cuda"), "fail": torch.zeros((3,), device="cuda")}
    run(acc, accumulate=True)
    run(acc, accumulate=True, scale=0.5)
    assert rel_maxnorm(acc["cubemap"].cpu().numpy(), 1.5 * tex_p.grad.cpu().numpy()) <= 1e-5
    np.testing.assert_allclose(acc["fail"].cpu().numpy(), 1.5 * fail_p.grad.cpu().numpy(), rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        run({"cubemap": acc["cubemap"]}, accumulate=True)


@pytest.mark.parametrize("forward_keys", [True, False])
def test_reflection_async_tail_joined_before_the_sink_is_read(forward_keys, monkeypatch):
    """Extension: async_tail=True puts the cubemap-gradient part of the backward on the library's side stream.  The per-pixel
    gradients autograd receives are unaffected; the sink is complete after _gsr.side_join() — overwrite and accumulate over
    two backwards, a busy main stream in between — and FlatGrads.all_reduce() joins by itself."""
    import _gsr
    import gaussian_renderer
    from gaussian_renderer import deferred_reflection
    from gsr_dist import FlatGrads
    monkeypatch.setattr(gaussian_renderer, "REFLECTION_FORWARD_KEYS", forward_keys)    # sort beside the pixel kernel / after it (small shape)
    W, H, L = 640, 360, 32
    cam = S.look_at_camera(W, H, eye=(1.0, -0.5, -4.0))
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    g = torch.Generator().manual_seed(5)
    nv0 = (torch.randn(3, H, W, generator=g) * torch.rand(1, H, W, generator=g)).cuda()
    base, strength = torch.rand(3, H, W, generator=g).cuda(), torch.rand(1, H, W, generator=g).cuda()
    tex0, fail0 = S.make_cubemap(L, 3, 6)
    wf = torch.randn(3, H, W, generator=g).cuda()

    class Env:
        pass

    def run(sink, accumulate=False, async_tail=False, scale=1.0):
        tex = torch.from_numpy(tex0).cuda().requires_grad_(True)
        fail = torch.from_numpy(fail0).cuda().requires_grad_(True)
        nv = nv0.clone().requires_grad_(True)
        env = Env()
        env.params = {"Cubemap_texture": tex, "Cubemap_failv": fail}
        f, _, _ = deferred_reflection(nv, base, strength, env, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"], grad_sink=sink,
                                      accumulate=accumulate, async_tail=async_tail)
        (f * wf * scale).sum().backward()
        return tex, fail, nv
    tex_p, fail_p, nv_p = run(None)
    with pytest.raises(ValueError):
        run(None, async_tail=True)          # autograd would read the gradient at once
    params = {"cubemap": torch.from_numpy(tex0).cuda().requires_grad_(True), "fail": torch.from_numpy(fail0).cuda().requires_grad_(True)}
    fg = FlatGrads(params)
    fg.flat.fill_(float("nan"))
    sink = fg.sink(names=("cubemap", "fail"))
    _, _, nv_a = run(sink, async_tail=True)
    busy = torch.randn(2048, 2048, device="cuda")
    busy = busy @ busy                       # main-stream work enqueued while the tail runs beside it
    _, _, nv_b = run(sink, accumulate=True, async_tail=True, scale=0.5)
    assert torch.equal(nv_a.grad, nv_p.grad) and rel_maxnorm(nv_b.grad.cpu().numpy(), 0.5 * nv_p.grad.cpu().numpy()) <= 1e-6
    assert len(_gsr._side_held) > 0
    fg.all_reduce()                          # joins (world size 1: nothing else)
    assert len(_gsr._side_held) == 0
    assert rel_maxnorm(fg.view("cubemap").cpu().numpy(), 1.5 * tex_p.grad.cpu().numpy()) <= 1e-5
    np.testing.assert_allclose(fg.view("fail").cpu().numpy(), 1.5 * fail_p.grad.cpu().numpy(), rtol=1e-5, atol=1e-7)
