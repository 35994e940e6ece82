"""GPU tests of the caller-side entry points as the reference's scripts use them:

  render_fast()     gaussian_renderer/__init__.py:221-325 of the reference (eval_fps.py:48-54, render.py:48)
  render_env_map()  gaussian_renderer/__init__.py:37-40 with the direction grids of utils/general_utils.py:200-240
  on-disk formats   GaussianModel.save_ply / load_ply and the `.map` state dict (scene/gaussian_model.py:225-262, 296-336):
                    written, read back, put on the device and rendered.
"""
import math
import os

import numpy as np
import pytest
import torch

from helpers import S

pytestmark = pytest.mark.gpu


def _view(cam, W, H):
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}

    class View:
        FoVx, FoVy = cam["FoVx"], cam["FoVy"]
        image_width, image_height = W, H
        world_view_transform, full_proj_transform, camera_center = ct["viewmatrix"], ct["projmatrix"], ct["campos"]
        HWK, R, T = (H, W, cam["K"]), ct["R"], ct["T"]
        znear, zfar = cam["znear"], cam["zfar"]
    return View


class _Pipe:
    depth_ratio, compute_cov3D_python = 0.0, False


def _model(t, env, degree=3):
    class PC:
        get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_refl = (t["means3D"], t["opacities"], t["scales"], t["rotations"],
                                                                                   t["shs"], t["refl_strengths"])
        active_sh_degree, get_envmap = degree, env
    return PC


def _scene(P, seed, mu, L):
    from cubemapencoder import CubemapEncoder
    sc = S.make_scene(P, "S", seed=seed, mu=mu)
    tex, fail = S.make_cubemap(L, 3, seed)
    t = {k: torch.from_numpy(sc[k]).cuda() for k in ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths")}
    env = CubemapEncoder(output_dim=3, resolution=L).cuda()
    with torch.no_grad():
        env.params["Cubemap_texture"].copy_(torch.from_numpy(tex))
        env.params["Cubemap_failv"].copy_(torch.from_numpy(fail) + 0.25)
    return t, env


@pytest.mark.parametrize("initial_stage", [False, True])
def test_render_fast_equals_render(initial_stage):
    """render_fast() is render() without the surface outputs and with an all-true env-scope mask: every map both return must be
    the same bits (same kernels, same inputs), with and without autograd recording (eval_fps.py runs it under no_grad)."""
    from gaussian_renderer import render, render_fast
    P, W, H = 30_000, 400, 240
    t, env = _scene(P, 41, -3.3, 32)
    View = _view(S.look_at_camera(W, H, eye=(0.4, -0.3, -1.0), target=(0, 0, 5)), W, H)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    PC = _model(t, env)
    full = render(View, PC, _Pipe, bg, initial_stage=initial_stage)
    keys = ("render", "rend_alpha", "rend_normal", "refl_strength_map") + (() if initial_stage else ("refl_color_map", "base_color_map"))
    for no_grad in (False, True):
        with torch.no_grad() if no_grad else torch.enable_grad():
            fast = render_fast(View, PC, _Pipe, bg, initial_stage=initial_stage)
        assert set(fast.keys()) == set(keys)
        for k in keys:
            if k == "refl_strength_map" and initial_stage:
                continue                                    # render() does not return it in the initial stage
            assert torch.equal(fast[k], full[k]), k
        assert fast["render"].shape == (3, H, W) and torch.isfinite(fast["render"]).all()
    assert float(full["rend_alpha"].max()) > 0.5            # the scene is really in view


def _env_dirs_reference(H, W):
    """The two direction grids of the reference's environment-map visualisation, restated with numpy from
    utils/general_utils.py:200-240 (grid 1: np.linspace end points included, z-up; grid 2: pixel-centre linspace, y-up)."""
    i, j = np.meshgrid(np.linspace(-np.pi, np.pi, W, dtype=np.float32), np.linspace(0, np.pi, H, dtype=np.float32), indexing='xy')
    d1 = np.stack([np.sin(j) * np.cos(i), np.sin(j) * np.sin(i), np.cos(j)], axis=-1)
    gy, gx = np.meshgrid(np.linspace(0.0 + 1.0 / H, 1.0 - 1.0 / H, H), np.linspace(-1.0 + 1.0 / W, 1.0 - 1.0 / W, W), indexing='ij')
    st, ct_, sp, cp = np.sin(gy * np.pi), np.cos(gy * np.pi), np.sin(gx * np.pi), np.cos(gx * np.pi)
    d2 = np.stack([st * sp, ct_, -st * cp], axis=-1)
    return d1.astype(np.float32), d2.astype(np.float32)


def test_render_env_map_against_oracle_cubemap():
    """render_env_map(): sigmoid(cubemap lookup) over the two panorama grids, against the oracle's cubemap on directions restated
    from the reference's grid code; default size and the keys the reference returns."""
    from gaussian_renderer import render_env_map
    from oracle import oracle as orc
    t, env = _scene(64, 5, -3.0, 64)
    out = render_env_map(_model(t, env))
    assert set(out.keys()) == {"env_cood1", "env_cood2"}
    tex, fail = env.params["Cubemap_texture"].detach().cpu().numpy(), env.params["Cubemap_failv"].detach().cpu().numpy()
    for key, dirs in zip(("env_cood1", "env_cood2"), _env_dirs_reference(512, 1024)):
        assert out[key].shape == (3, 512, 1024)
        ref = orc.cubemap_forward(dirs.reshape(-1, 3), tex, fail, 1, 1, dtype=np.float64)            # [C, B]
        ref = 1.0 / (1.0 + np.exp(-ref))
        got = out[key].detach().cpu().numpy().reshape(3, -1)
        # float32 trigonometry on both sides of a texel border moves a weight, not a texel: small everywhere but at a few pixels on cube
        # edges where the face changes (the lookup is continuous across edges with seamless filtering, so still small)
        assert np.abs(got - ref).max() < 2e-4, key
        assert np.abs(got - ref).mean() < 2e-6, key
    # a smaller panorama through the same code path
    small = render_env_map(_model(t, env), height=64, width=128)
    assert small["env_cood2"].shape == (3, 64, 128)


def test_ply_and_map_round_trip_renders_identically(tmp_path):
    """F4 on the device: save_ply (+ .map) -> load_ply -> GaussianTrainState on the GPU -> render() is bit-identical to the render from
    the original tensors; the `.map` state dict loads into a CubemapEncoder (the reference's load path: GaussianModel.load_ply
    builds the encoder at the stored resolution and calls load_state_dict) and the environment panoramas agree bit for bit."""
    from cubemapencoder import CubemapEncoder
    from gaussian_renderer import render, render_env_map
    from gsr_train import GaussianTrainState
    from scene.ply_io import load_ply, save_ply
    P, W, H, L = 20_000, 320, 200, 32
    t, env = _scene(P, 77, -3.2, L)
    View = _view(S.make_camera(W, H), W, H)
    bg = torch.zeros(3, device="cuda")
    with torch.no_grad():
        want = render(View, _model(t, env), _Pipe, bg)
        want_env = render_env_map(_model(t, env), height=64, width=128)
    path = os.path.join(tmp_path, "point_cloud", "iteration_7", "point_cloud.ply")
    save_ply(path, t["means3D"], t["shs"], t["opacities"], t["refl_strengths"], t["scales"], t["rotations"],
             cubemap=env.params["Cubemap_texture"], fail_value=env.params["Cubemap_failv"])
    assert os.path.getsize(path) > P * 62 * 4 and os.path.exists(path.replace(".ply", ".map"))
    data = load_ply(path)
    assert data["cubemap"].shape == (6, 3, L, L) and data["shs"].shape == (P, 16, 3)
    # the `.map` file is the encoder's state dict
    env2 = CubemapEncoder(output_dim=3, resolution=data["cubemap"].shape[-1]).cuda()
    missing = env2.load_state_dict(torch.load(path.replace(".ply", ".map"), map_location="cuda", weights_only=True))
    assert not missing.missing_keys and not missing.unexpected_keys
    st = GaussianTrainState({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in data.items()}, "cuda")
    loaded = {k: st.p[k] for k in ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths")}
    for k, v in loaded.items():
        assert torch.equal(v.detach(), t[k]), k
    with torch.no_grad():
        got = render(View, _model(loaded, env2), _Pipe, bg)
        got_env = render_env_map(_model(loaded, env2), height=64, width=128)
    for k in ("render", "rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal", "refl_strength_map", "refl_color_map", "base_color_map",
              "radii", "gaussian_weights"):
        assert torch.equal(got[k], want[k]), k
    for k in want_env:
        assert torch.equal(got_env[k], want_env[k]), k
    # and the optimizer state built on the loaded parameters trains: one full step moves the render
    out = render(View, _model(loaded, env2), _Pipe, bg)
    out["render"].mean().backward()
    assert math.isfinite(float(st.p["means3D"].grad.abs().max()))
