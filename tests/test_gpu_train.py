"""GPU parity of the training-step passes through the C ABI: fused SSIM + L1 loss and the fused flat
Adam, against the oracle, the torch-generated golden vectors and (Adam) torch.optim.Adam on the same device.
Tolerances: loss values 1e-5 absolute (fp32 sums over up to 6e6 terms), image gradient 1e-4 relative to its max
(fp32 conv sums in a different order than the oracle), Adam 1e-6 relative (same op sequence, different rounding of the
reciprocal bias corrections)."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_golden.npz"))


def _orc():
    from oracle import oracle as orc
    return orc


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_photometric_loss_matches_golden_and_oracle(tag):
    from utils.loss_utils import l1_loss, photometric_loss, ssim
    img = torch.from_numpy(G[f"{tag}_img"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(G[f"{tag}_gt"]).cuda()
    assert abs(l1_loss(img, gt).item() - float(G[f"{tag}_l1"])) < 1e-5
    assert abs(ssim(img, gt).item() - float(G[f"{tag}_ssim"])) < 1e-5
    loss = photometric_loss(img, gt, 0.2)
    assert abs(loss.item() - float(G[f"{tag}_loss"])) < 1e-5
    loss.backward()
    ref = G[f"{tag}_grad"]
    assert np.abs(img.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    # unsqueezed (1,C,H,W) call shape of train.py:170
    assert abs(ssim(img.detach().unsqueeze(0), gt.unsqueeze(0)).item() - float(G[f"{tag}_ssim"])) < 1e-5


def test_l1_loss_and_ssim_on_the_same_tensors_share_one_fused_pass():
    """train.py:167-173 calls l1_loss(image, gt) and ssim(image, gt) one after the other: the second call reuses the first call's pair of sums
    (one forward kernel, one backward kernel), the loss and its gradient equal photometric_loss's; a changed image or other tensors do not hit."""
    import _gsr
    from utils.loss_utils import l1_loss, photometric_loss, ssim
    g = torch.Generator().manual_seed(21)
    gt = torch.rand(3, 97, 131, generator=g).cuda()
    base = torch.rand(3, 97, 131, generator=g).cuda()

    def run(fused):
        img = base.clone().requires_grad_(True)
        _gsr.profile_enable(True)
        if fused:
            loss = photometric_loss(img, gt, 0.2)
        else:
            loss = 0.8 * l1_loss(img, gt) + 0.2 * (1.0 - ssim(img, gt))
        loss.backward()
        torch.cuda.synchronize()
        st = _gsr.profile_collect()
        _gsr.profile_enable(False)
        return float(loss), img.grad.clone(), st["loss_fwd"][1], st["loss_bwd"][1]
    lf, gf, nf_f, nb_f = run(True)
    lp, gp, nf_p, nb_p = run(False)
    assert (nf_f, nb_f) == (1, 1) and (nf_p, nb_p) == (1, 1)
    assert abs(lf - lp) < 1e-6 and (gf - gp).abs().max().item() <= 1e-6 * gf.abs().max().item()
    # no stale hit: another image object, and the same object modified in place, are recomputed
    a = base.clone()
    v1 = float(l1_loss(a, gt))
    a.add_(1.0)                      # (in place: same object, new version)
    v2 = float(l1_loss(a, gt))
    assert abs(v1 - v2) > 0.1 and abs(v2 - float((a - gt).abs().mean())) < 1e-5
    with torch.no_grad():
        assert abs(float(ssim(a, gt)) - float(ssim(a.clone(), gt))) < 1e-7


def test_ssim_map_and_ragged_sizes_against_oracle():
    from utils.loss_utils import C1, C2, FusedSSIMMap, photometric_loss
    orc = _orc()
    rs = np.random.RandomState(11)
    for (C, H, W) in [(3, 1, 1), (3, 17, 5), (1, 33, 47), (3, 120, 160)]:
        gt = rs.rand(C, H, W).astype(np.float32)
        img = np.clip(0.6 * gt + 0.4 * rs.rand(C, H, W), 0, 1).astype(np.float32)
        s_l1, s_ss, smap = orc.ssim_l1_forward(img, gt, dtype=np.float64)
        ti, tg = torch.from_numpy(img).cuda().requires_grad_(True), torch.from_numpy(gt).cuda()
        got = FusedSSIMMap.apply(C1, C2, ti.detach(), tg).cpu().numpy()
        assert np.abs(got - smap).max() < 2e-5, (C, H, W)
        n = img.size
        loss = photometric_loss(ti, tg, 0.2)
        assert abs(loss.item() - (0.8 * s_l1 / n + 0.2 * (1 - s_ss / n))) < 1e-5
        loss.backward()
        ref = orc.ssim_l1_backward(img, gt, 0.8 / n, -0.2 / n, dtype=np.float64)
        assert np.abs(ti.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max(), (C, H, W)


def test_loss_properties_at_full_size():
    """1080p (BASELINE C3 image size): identical images -> SSIM = 1, L1 = 0, zero gradient; linearity of the backward
    in the upstream weights."""
    from utils.loss_utils import _SsimL1, C1, C2, photometric_loss, ssim
    g = torch.Generator(device="cpu").manual_seed(5)
    img = torch.rand(3, 1080, 1920, generator=g).cuda()
    a = img.clone().requires_grad_(True)
    assert abs(ssim(a, img).item() - 1.0) < 1e-5
    loss = photometric_loss(a, img, 0.2)
    assert abs(loss.item()) < 1e-5
    loss.backward()
    assert a.grad.abs().max().item() < 1e-9
    other = (0.5 * img + 0.5 * torch.rand(3, 1080, 1920, generator=g).cuda()).requires_grad_(True)
    grads = []
    for w in ([1.0, 0.0], [0.0, 1.0], [2.0, -3.0]):
        other.grad = None
        s = _SsimL1.apply(other, img, C1, C2, False)[0]
        (s * torch.tensor(w, device="cuda")).sum().backward()
        grads.append(other.grad.clone())
    comb = 2.0 * grads[0] - 3.0 * grads[1]
    assert (comb - grads[2]).abs().max().item() <= 1e-5 * grads[2].abs().max().item()


def test_fused_adam_matches_golden_oracle_and_torch():
    from gsr_train import FlatAdam, FlatParams
    # (a) golden: two groups, three steps
    fp = FlatParams(dict(a=torch.from_numpy(G["adam_p0"][:257]), b=torch.from_numpy(G["adam_p0"][257:])), "cuda")
    grad = torch.zeros_like(fp.flat)
    opt = FlatAdam(fp, grad, dict(a=0.00016, b=0.0025))
    off_b = fp.slices["b"][0]
    for step in range(3):
        g = torch.from_numpy(G["adam_grads"][step]).cuda()
        grad.zero_()
        grad[:257] = g[:257]
        grad[off_b:off_b + 96] = g[257:]
        opt.step()
        got = torch.cat([fp.p["a"].detach(), fp.p["b"].detach()]).cpu().numpy()
        ref = G[f"adam_p{step + 1}"]
        assert np.abs(got - ref).max() <= 1e-6 * np.abs(ref).max(), step
    # (b) interleaved learning rates (shs rows: 3 floats at lr, 45 at lr/20) and ragged sizes vs the oracle
    orc = _orc()
    rs = np.random.RandomState(2)
    shs = rs.randn(1001, 16, 3).astype(np.float32)
    op = rs.randn(1001, 1).astype(np.float32)
    fp = FlatParams(dict(shs=torch.from_numpy(shs), opacities=torch.from_numpy(op)), "cuda")
    grad = torch.zeros_like(fp.flat)
    opt = FlatAdam(fp, grad, dict(shs=(0.0025, 0.0025 / 20, 48, 3), opacities=0.05))
    lr = np.concatenate([np.tile(np.r_[np.full(3, 0.0025), np.full(45, 0.0025 / 20)], 1001), np.full(1001, 0.05)]).astype(np.float32)
    p = np.concatenate([shs.reshape(-1), op.reshape(-1)])
    m, v = np.zeros_like(p), np.zeros_like(p)
    a0, a1 = fp.slices["shs"]
    b0, b1 = fp.slices["opacities"]
    for step in range(1, 4):
        g = (rs.randn(p.size) * 0.1).astype(np.float32)
        grad[a0:a1] = torch.from_numpy(g[: a1 - a0]).cuda()
        grad[b0:b1] = torch.from_numpy(g[a1 - a0:]).cuda()
        opt.step()
        p, m, v = orc.adam(p, g, m, v, lr, step=step, dtype=np.float32)
        got = np.concatenate([fp.p["shs"].detach().cpu().numpy().reshape(-1), fp.p["opacities"].detach().cpu().numpy().reshape(-1)])
        assert np.abs(got - p).max() <= 1e-6 * np.abs(p).max(), step
    # (c) torch.optim.Adam on the same device, one big tensor
    w0 = torch.randn(1_000_003, device="cuda")
    fp = FlatParams(dict(w=w0.clone()), "cuda")
    grad = torch.zeros_like(fp.flat)
    opt = FlatAdam(fp, grad, dict(w=0.01))
    wt = w0.clone().requires_grad_(True)
    topt = torch.optim.Adam([wt], lr=0.01, eps=1e-15)
    for step in range(2):
        g = torch.randn(1_000_003, device="cuda")
        grad[:1_000_003] = g
        wt.grad = g.clone()
        opt.step(); topt.step()
    assert (fp.p["w"].detach() - wt.detach()).abs().max().item() <= 1e-6 * wt.detach().abs().max().item()


def test_adam_argument_errors():
    from _gsr import AdamSegment, GsrError, check, lib
    p = torch.zeros(64, device="cuda")
    seg = (AdamSegment * 1)(AdamSegment(0, 32, 0.1, 0.0, 0, 0))       # does not tile [0, 64)
    with pytest.raises(GsrError):
        check(lib.gsr_adam_step(p.data_ptr(), p.data_ptr(), p.data_ptr(), p.data_ptr(), 64, seg, 1, 0.9, 0.999, 1e-15, 1, None), "adam")
    seg = (AdamSegment * 1)(AdamSegment(0, 64, 0.1, 0.0, 0, 0))
    with pytest.raises(GsrError):                                     # step counter is 1-based
        check(lib.gsr_adam_step(p.data_ptr(), p.data_ptr(), p.data_ptr(), p.data_ptr(), 64, seg, 1, 0.9, 0.999, 1e-15, 0, None), "adam")


def test_train_state_step_end_to_end():
    """One full step of the harness on a small scene: render (surfel + deferred reflection) -> photometric loss ->
    backward into the flat gradient buffer (gradient sink) -> fused Adam; every parameter group moves by at most
    lr (|Adam update| <= lr at step 1) and by exactly lr * sign(g) where the gradient is non-zero."""
    import gsr_synth as S
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussian_renderer import deferred_reflection
    from gsr_train import GaussianTrainState
    from utils.loss_utils import photometric_loss
    P, W, H, L = 5000, 160, 120, 16
    sc = S.make_scene(P, "S", seed=9, mu=-2.6)
    tex, fail = S.make_cubemap(L, 3, 9)
    cam = S.make_camera(W, H)
    names = ["means3D", "shs", "opacities", "scales", "rotations", "refl_strengths"]
    tensors = {k: torch.from_numpy(sc[k]) for k in names}
    tensors["cubemap"], tensors["fail"] = torch.from_numpy(tex), torch.from_numpy(fail)
    st = GaussianTrainState(tensors, "cuda")
    before = st.params.flat.clone()
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                             bg=torch.zeros(3, device="cuda"), scale_modifier=1.0, viewmatrix=ct["viewmatrix"],
                                             projmatrix=ct["projmatrix"], sh_degree=3, campos=ct["campos"], prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    sink = st.grads.sink()
    rast.set_grad_sink(sink)

    class Env:
        params = {"Cubemap_texture": st.p["cubemap"], "Cubemap_failv": st.p["fail"]}
    means2D = torch.zeros(P, 3, device="cuda", requires_grad=True)
    base, radii, allmap, refl_map, gw = rast(means3D=st.p["means3D"], means2D=means2D, opacities=st.p["opacities"], shs=st.p["shs"],
                                             refl_strengths=st.p["refl_strengths"], scales=st.p["scales"], rotations=st.p["rotations"],
                                             env_scope_mask=torch.from_numpy(sc["env_scope_mask"]).cuda())
    final, _, _ = deferred_reflection(allmap[2:5], base, refl_map, Env, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"])
    gt = torch.rand(3, H, W, device="cuda")
    st.grads.zero_except_(sink)
    loss = photometric_loss(final, gt, 0.2)
    loss.backward()
    assert torch.isfinite(st.grads.flat).all()
    st.update_learning_rate(1)
    st.optimizer.step()
    moved = st.params.flat - before
    for k in st.params.names:
        a, b = st.params.slices[k]
        g, d = st.grads.flat[a:b], moved[a:b]
        lr, lr2, period, split = st.optimizer.groups[k]
        ulp = 1e-6 * max(1.0, before[a:b].abs().max().item())     # p_new - p_old is rounded at the magnitude of p
        cap = max(lr, lr2) * (1 + 1e-5) + ulp
        assert d.abs().max().item() <= cap, k
        if k != "fail":                                # (the fail value only matters for a zero reflection vector)
            assert (g != 0).any(), k                   # every group received gradient
        nz = g.abs() > 1e-12
        # step 1: m/(1-b1) = g, sqrt(v/(1-b2)) = |g|  ->  update = -lr * sign(g) (eps = 1e-15 is negligible)
        if period == 0:
            assert torch.allclose(d[nz], -lr * torch.sign(g[nz]), rtol=1e-3, atol=lr * 1e-3 + ulp), k


def test_training_loop_converges_and_survives_densification():
    """The pieces together, as train.py:120-306 strings them: render() (rasterizer + fused surface pass + fused
    reflection) -> photometric + normal-consistency loss -> backward into the flat buffer -> fused Adam, with the
    densification statistics every step and one densify_and_prune in the middle.  Target: an image rendered from the
    unperturbed scene; the perturbed model must get closer to it."""
    import gsr_synth as S
    from gaussian_renderer import render
    from gsr_densify import DensifyStats, densify_and_prune
    from gsr_train import GaussianTrainState
    from utils.loss_utils import photometric_loss
    P, W, H, L = 6000, 160, 120, 16
    sc = S.make_scene(P, "S", seed=13, mu=-2.7)
    tex, fail = S.make_cubemap(L, 3, 13)
    cam = S.make_camera(W, H)
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}

    class View:
        FoVx, FoVy = 2 * np.arctan(cam["tanfovx"]), 2 * np.arctan(cam["tanfovy"])
        image_width, image_height = W, H
        world_view_transform, full_proj_transform, camera_center = ct["viewmatrix"], ct["projmatrix"], ct["campos"]
        HWK, R, T = (H, W, cam["K"]), ct["R"], ct["T"]
        znear, zfar = 0.01, 100.0

    class Pipe:
        depth_ratio, compute_cov3D_python = 0.0, False

    def model_of(st):
        class Env:
            params = {"Cubemap_texture": st.p["cubemap"], "Cubemap_failv": st.p["fail"]}

        class PC:
            get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_refl = (st.p["means3D"], st.p["opacities"], st.p["scales"],
                                                                                       st.p["rotations"], st.p["shs"], st.p["refl_strengths"])
            active_sh_degree, get_envmap = 3, Env
        return PC

    names = ["means3D", "shs", "opacities", "scales", "rotations", "refl_strengths"]
    truth = {k: torch.from_numpy(sc[k]) for k in names}
    truth["cubemap"], truth["fail"] = torch.from_numpy(tex), torch.from_numpy(fail)
    bg = torch.zeros(3, device="cuda")
    with torch.no_grad():
        gt = render(View, model_of(GaussianTrainState(truth, "cuda")), Pipe, bg)["render"].detach().clone()
    g = torch.Generator(device="cpu").manual_seed(1)
    start = {k: v.clone() for k, v in truth.items()}
    start["shs"] = start["shs"] + 0.3 * torch.randn(start["shs"].shape, generator=g)
    start["opacities"] = start["opacities"] * 0.7
    st = GaussianTrainState(start, "cuda", spatial_lr_scale=1.0)
    stats = DensifyStats(P, "cuda")
    losses = []
    for it in range(1, 41):
        st.update_learning_rate(it)
        st.grads.zero_()
        pkg = render(View, model_of(st), Pipe, bg)
        loss = photometric_loss(pkg["render"], gt, 0.2)
        normal_error = (1 - (pkg["rend_normal"] * pkg["surf_normal"]).sum(dim=0))[None]      # train.py:182-189
        loss = loss + 0.05 * normal_error.mean()
        loss.backward()
        losses.append(float(photometric_loss(pkg["render"].detach(), gt, 0.2)))
        stats.update(pkg["viewspace_points"].grad, pkg["radii"], pkg["gaussian_weights"])
        st.optimizer.step()
        if it == 20:
            st, stats, info = densify_and_prune(st, stats, 0.0002, 0.05, torch.zeros(3), 5.0, None)
            assert info["after"] == st.p["means3D"].shape[0] and info["after"] > 0
    assert np.isfinite(losses).all()
    # before the densification step the fit improves steadily; pruning by blend weight (accum_w < 0.01, as the reference
    # does) removes many of the synthetic scene's half-hidden surfels at once, after which the fit improves again
    assert np.mean(losses[16:20]) < 0.8 * np.mean(losses[:3]), (losses[:3], losses[16:20])
    assert np.mean(losses[-3:]) < np.mean(losses[20:23]), (losses[20:23], losses[-3:])


def test_world_size_one_rccl_step_backward_sink_allreduce_adam():
    """The whole N-GPU data path on one GPU with the real backend: HIP backward -> gradient sink (two views accumulated on the
    device) -> ONE all-reduce over RCCL (backend "nccl", world size 1) -> fused Adam.  Against the same two views through plain
    autograd accumulation and no process group."""
    import torch.distributed as dist
    import gsr_synth as S
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussian_renderer import deferred_reflection
    from gsr_train import GaussianTrainState
    from utils.loss_utils import photometric_loss
    P, W, H, L = 4000, 160, 120, 16
    sc = S.make_scene(P, "S", seed=19, mu=-2.6)
    tex, fail = S.make_cubemap(L, 3, 19)
    cams = [S.look_at_camera(W, H, eye=(0.3 * k, -0.1 * k, -0.4)) for k in range(2)]
    names = ["means3D", "shs", "opacities", "scales", "rotations", "refl_strengths"]
    tensors = {k: torch.from_numpy(sc[k]) for k in names}
    tensors["cubemap"], tensors["fail"] = torch.from_numpy(tex), torch.from_numpy(fail)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).cuda()
    mask = torch.from_numpy(sc["env_scope_mask"]).cuda()

    def run(use_sinks):
        st = GaussianTrainState({k: v.clone() for k, v in tensors.items()}, "cuda")

        class Env:
            params = {"Cubemap_texture": st.p["cubemap"], "Cubemap_failv": st.p["fail"]}
        st.grads.zero_()
        for i, cam in enumerate(cams):
            ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
            rast = GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                                                    bg=torch.zeros(3, device="cuda"), scale_modifier=1.0,
                                                                    viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"], sh_degree=3,
                                                                    campos=ct["campos"], prefiltered=False, debug=False))
            rsink = None
            if use_sinks:
                rast.set_grad_sink(st.grads.sink(), accumulate=i > 0)
                rsink = st.grads.sink(names=("cubemap", "fail"))
            base, radii, allmap, refl_map, gw = rast(means3D=st.p["means3D"], means2D=torch.zeros(P, 3, device="cuda", requires_grad=True),
                                                     opacities=st.p["opacities"], shs=st.p["shs"], refl_strengths=st.p["refl_strengths"],
                                                     scales=st.p["scales"], rotations=st.p["rotations"], env_scope_mask=mask)
            final, _, _ = deferred_reflection(allmap[2:5], base, refl_map, Env, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"],
                                              grad_sink=rsink, accumulate=i > 0)
            photometric_loss(final, gt, 0.2).backward()
        if use_sinks:
            st.grads.all_reduce()
        grads = st.grads.flat.clone()
        st.update_learning_rate(1)
        st.optimizer.step()
        return grads, st.params.flat.clone()

    g_plain, p_plain = run(False)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29533", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        g_sunk, p_sunk = run(True)
    finally:
        dist.destroy_process_group()
    assert torch.isfinite(g_sunk).all() and g_sunk.abs().max() > 0
    den = g_plain.abs().max().item()
    assert (g_sunk - g_plain).abs().max().item() <= 5e-5 * den
    # identical gradients up to atomics order -> the Adam step (sign-like at step 1) lands within the learning rate
    assert (p_sunk - p_plain).abs().max().item() <= 2.1 * 0.0025 + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,use_mask", [(1080, 1920, True), (67, 131, False), (1, 1, True)])
def test_normal_consistency_loss_matches_the_torch_expression(H, W, use_mask):
    """utils.loss_utils.normal_consistency_loss against the reference's five torch ops (train.py:182-189) evaluated in float64:
    value to 1e-6 relative, both gradients to 1e-6 of their maximum; twice the same value bit for bit (fixed-order reduction)."""
    from utils.loss_utils import normal_consistency_loss
    g = torch.Generator().manual_seed(H * 1000 + W)
    rn = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0).cuda().requires_grad_(True)
    sn = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0).cuda().requires_grad_(True)
    mask = (torch.rand(1, H, W, generator=g) < 0.7).float().cuda() if use_mask else None
    lam = 0.05
    loss = normal_consistency_loss(rn, sn, lam, mask)
    (loss * 3.0).backward()
    rd, sd = rn.detach().double().requires_grad_(True), sn.detach().double().requires_grad_(True)
    err = (1 - (rd * sd).sum(dim=0))[None]
    if use_mask:
        err = err * mask.double()
    ref = lam * err.mean()
    (ref * 3.0).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6 * max(1e-6, abs(float(ref.detach())))
    for a, b in ((rn.grad, rd.grad), (sn.grad, sd.grad)):
        assert (a.double() - b).abs().max() <= 1e-6 * max(float(b.abs().max()), 1e-30)
    assert float(normal_consistency_loss(rn.detach(), sn.detach(), lam, mask)) == float(loss.detach())
