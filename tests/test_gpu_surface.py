"""GPU parity of the fused surface pass through the C ABI against the torch-autograd golden vectors
and the oracle, and of render() against the torch chain of tests/helpers_chain.py on the rasterizer's allmap.
Tolerances: surf_depth 1e-5 relative, surf_normal 2e-4 absolute (unit vectors from fp32 differences of points ~5 units
apart), gradients 1e-3 of the tensor's max (the cross product of two fp32 central differences cancels ~3 digits)."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "surface_golden.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_surface_pass_matches_golden(tag):
    from gaussian_renderer import _SurfacePass
    am = torch.from_numpy(G[f"{tag}_allmap"].astype(np.float32)).cuda().requires_grad_(True)
    ray = torch.from_numpy(G[f"{tag}_raymat"].astype(np.float32)).cuda()
    ratio = float(G[f"{tag}_ratio"])
    sd, sn = _SurfacePass.apply(am, ray, ratio)
    ref_sd, ref_sn = G[f"{tag}_surf_depth"], G[f"{tag}_surf_normal"]
    ok = np.isfinite(ref_sd)
    assert np.abs(sd.detach().cpu().numpy()[ok] - ref_sd[ok]).max() <= 1e-5 * np.abs(ref_sd[ok]).max()
    # normals next to the injected inf / lowest-float depths are ill-conditioned: compare where the reference is well inside
    got_sn = sn.detach().cpu().numpy()
    bad = np.zeros(ref_sn.shape[1:], bool)
    for (y, x) in ((3, 4), (5, 6), (7, 8)):
        bad[max(0, y - 1):y + 2, max(0, x - 1):x + 2] = True
    assert np.abs(got_sn - ref_sn)[:, ~bad].max() < 2e-4
    g_sd = torch.from_numpy(G[f"{tag}_g_sd"].astype(np.float32)).cuda()
    g_sn = torch.from_numpy(G[f"{tag}_g_sn"].astype(np.float32)).cuda()
    g_sn[:, torch.from_numpy(bad).cuda()] = 0        # keep the ill-conditioned stencils out of the gradient comparison
    ((sd * g_sd).sum() + (sn * g_sn).sum()).backward()
    from oracle import oracle as orc
    gsn = G[f"{tag}_g_sn"].copy()
    gsn[:, bad] = 0
    _, _, ref = orc.surface_pass(G[f"{tag}_allmap"], G[f"{tag}_raymat"], ratio, G[f"{tag}_g_sd"][0], gsn, dtype=np.float64)
    got = am.grad.cpu().numpy()
    far = np.zeros_like(bad)
    for (y, x) in ((3, 4), (5, 6), (7, 8)):
        far[max(0, y - 2):y + 3, max(0, x - 2):x + 3] = True
    fin = np.isfinite(ref) & ~far[None]
    assert np.abs(got[fin] - ref[fin]).max() <= 1e-3 * np.abs(ref[fin]).max()
    assert np.abs(got[[2, 3, 4, 6, 7]]).max() == 0


def test_surface_pass_1080p_properties_and_oracle_sample():
    """Full-size properties: border normals are exactly zero, interior normals have length alpha, a fronto-parallel plane
    gives the view axis; gradient w.r.t. the median plane vanishes at depth_ratio 0."""
    import gsr_synth as S
    from gaussian_renderer import _SurfacePass
    H, W = 1080, 1920
    cam = S.make_camera(W, H)
    wvt = torch.from_numpy(cam["viewmatrix"]).cuda()
    fpt = torch.from_numpy(cam["projmatrix"]).cuda()

    class V:
        world_view_transform, full_proj_transform, image_width, image_height = wvt, fpt, W, H
    from gaussian_renderer import _ray_block
    ray = _ray_block(V)
    am = torch.zeros(8, H, W, device="cuda")
    alpha = 0.25 + 0.5 * torch.rand(H, W, device="cuda")
    am[1] = alpha
    am[0] = 4.0 * alpha          # expected depth 4 everywhere: the plane z = 4 in front of an identity camera
    am[5] = 4.0
    am.requires_grad_(True)
    sd, sn = _SurfacePass.apply(am, ray, 0.0)
    assert torch.allclose(sd, torch.full_like(sd, 4.0), rtol=1e-6)
    assert sn[:, 0].abs().max() == 0 and sn[:, -1].abs().max() == 0 and sn[:, :, 0].abs().max() == 0 and sn[:, :, -1].abs().max() == 0
    inner = sn[:, 1:-1, 1:-1]
    assert torch.allclose(inner.norm(dim=0), alpha[1:-1, 1:-1], rtol=1e-4)
    assert torch.allclose(inner[2] / alpha[1:-1, 1:-1], -torch.ones_like(inner[2]), atol=1e-3)   # cross(+y step, +x step) = -z: faces the camera
    (sn * torch.randn_like(sn)).sum().backward()
    assert am.grad[5].abs().max() == 0 and torch.isfinite(am.grad).all() and am.grad[0].abs().max() > 0


def test_render_surface_outputs_equal_torch_chain():
    """render() end to end on a small scene: its surf_depth / surf_normal / rend_normal (fused HIP passes) against the torch
    chain of tests/helpers_chain.py evaluated on the rasterizer's own allmap, values and gradients through to the
    parameters (both on the GPU, float32)."""
    import gsr_synth as S
    from diff_surfel_rasterization import GaussianRasterizer
    from gaussian_renderer import _settings, render
    from helpers_chain import shading_normal_chain, surface_chain
    P, W, H = 4000, 128, 96
    sc = S.make_scene(P, "S", seed=31, mu=-2.6)
    tex, fail = S.make_cubemap(16, 3, 31)
    cam = S.make_camera(W, H)
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}

    class View:
        FoVx, FoVy = 2 * np.arctan(cam["tanfovx"]), 2 * np.arctan(cam["tanfovy"])
        image_width, image_height = W, H
        world_view_transform, full_proj_transform, camera_center = ct["viewmatrix"], ct["projmatrix"], ct["campos"]
        HWK, R, T = (H, W, cam["K"]), ct["R"], ct["T"]
        znear, zfar = 0.01, 100.0

    class Pipe:
        depth_ratio, compute_cov3D_python = 0.3, False

    def model():
        t = {k: torch.from_numpy(sc[k]).cuda().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths")}

        class Env:
            params = {"Cubemap_texture": torch.from_numpy(tex).cuda(), "Cubemap_failv": torch.from_numpy(fail).cuda()}

        class PC:
            get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_refl = (t["means3D"], t["opacities"], t["scales"], t["rotations"],
                                                                                       t["shs"], t["refl_strengths"])
            active_sh_degree, get_envmap = 3, Env
        return PC, t

    bg = torch.zeros(3, device="cuda")
    outs = []
    for initial_stage in (False, True):
        PC, t = model()
        pkg = render(View, PC, Pipe, bg, initial_stage=initial_stage)
        ((pkg["surf_normal"] * pkg["rend_normal"]).sum() + pkg["surf_depth"].mean()).backward()
        outs.append((pkg["surf_depth"].detach(), pkg["surf_normal"].detach(), pkg["rend_normal"].detach(), {k: v.grad.clone() for k, v in t.items()}))
        assert ("refl_color_map" in pkg) == (not initial_stage)
    # the same quantities through the torch chain on the rasterizer's allmap
    PC, t = model()
    rast = GaussianRasterizer(raster_settings=_settings(View, PC, bg, 1.0))
    _, _, allmap, _, _ = rast(means3D=t["means3D"], means2D=torch.zeros(P, 3, device="cuda", requires_grad=True), shs=t["shs"],
                              refl_strengths=t["refl_strengths"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                              env_scope_mask=torch.ones(P, dtype=torch.bool, device="cuda"))
    sd_c, sn_c = surface_chain(allmap, ct["viewmatrix"], ct["projmatrix"], Pipe.depth_ratio)
    rn_c = shading_normal_chain(allmap[2:5], ct["viewmatrix"]).permute(2, 0, 1)
    ((sn_c * rn_c).sum() + sd_c.mean()).backward()
    g_c = {k: v.grad.clone() for k, v in t.items()}
    for sd, sn, rn, g in outs:
        assert torch.allclose(sd, sd_c.detach(), rtol=1e-5, atol=1e-6)
        assert (sn - sn_c.detach()).abs().max().item() < 5e-4
        assert (rn - rn_c.detach()).abs().max().item() < 1e-5
        for k in g:
            assert (g[k] - g_c[k]).abs().max().item() <= 2e-3 * g_c[k].abs().max().item() + 1e-9, k


def test_camera_blocks_are_rebuilt_for_new_cameras():
    """Cameras created and freed in a loop (per-frame cameras of a video / viewer loop) get their own camera constants even
    when the allocator hands a new camera the addresses of a freed one: the fused reflection must follow the camera."""
    import gsr_synth as S
    from gaussian_renderer import deferred_reflection
    from helpers_chain import reflection_chain
    from cubemapencoder import CubemapEncoder
    W, H = 96, 64
    g = torch.Generator().manual_seed(8)
    nv = torch.randn(3, H, W, generator=g).cuda()
    base, s = torch.rand(3, H, W, generator=g).cuda(), torch.rand(1, H, W, generator=g).cuda()
    enc = CubemapEncoder(output_dim=3, resolution=16).cuda()
    seen = set()
    for k in range(6):
        cam = S.look_at_camera(W, H, eye=(0.5 * k - 1.0, 0.2 * k, -3.0 - 0.3 * k))
        ct = {kk: torch.from_numpy(np.ascontiguousarray(v)).cuda() for kk, v in cam.items() if isinstance(v, np.ndarray)}
        seen.add(ct["viewmatrix"].data_ptr())
        f_h, _, n_h = deferred_reflection(nv, base, s, enc, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"])
        f_c, _, n_c = reflection_chain(nv, base, s, enc, ct["viewmatrix"], H, W, cam["K"], ct["R"], ct["T"])
        assert (n_h - n_c).abs().max().item() <= 1e-5, k
        assert (torch.abs(f_h - f_c) > 1e-3).float().mean().item() <= 2e-3, k
        del ct, f_h, n_h, f_c, n_c
