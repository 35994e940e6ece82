"""Host-side logic of the drop-in packages that runs without a GPU: argument validation mirrors the reference's
wrappers (same exceptions, same messages), settings tuples have the reference's fields, synthetic data is seeded."""
import os
import sys

import numpy as np
import pytest
import torch

import gsr_synth as S


def test_settings_fields_match_reference(hip_lib_built):
    from diff_surfel_rasterization import GaussianRasterizationSettings as SS
    from diff_gaussian_rasterization import GaussianRasterizationSettings as GS
    base = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
            "campos", "prefiltered", "debug")
    assert SS._fields == base                      # DSR __init__.py:170-182
    assert GS._fields == base + ("antialiasing",)  # DGR __init__.py:157-170


def _settings(cls, **extra):
    z = torch.zeros(3)
    return cls(image_height=8, image_width=8, tanfovx=0.5, tanfovy=0.5, bg=z, scale_modifier=1.0, viewmatrix=torch.eye(4),
               projmatrix=torch.eye(4), sh_degree=0, campos=z, prefiltered=False, debug=False, **extra)


def test_surfel_rasterizer_argument_errors(hip_lib_built):
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    r = GaussianRasterizer(_settings(GaussianRasterizationSettings))
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="either SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1))                      # neither
    with pytest.raises(Exception, match="either SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3), colors_precomp=torch.zeros(4, 3))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3), scales=torch.zeros(4, 2),
          rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 9))
    # CPU tensors are refused like CHECK_INPUT does (DSR rasterize_points.cu:27-28); nothing reaches the device
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3), refl_strengths=torch.zeros(4, 1),
          scales=torch.zeros(4, 2), rotations=torch.zeros(4, 4))


def test_gauss_rasterizer_argument_errors(hip_lib_built):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    r = GaussianRasterizer(_settings(GaussianRasterizationSettings, antialiasing=False))
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="either SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), colors_precomp=torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        from diff_gaussian_rasterization import _C
        _C.rasterize_gaussians(torch.zeros(3), torch.zeros(4), *([torch.zeros(0)] * 6), 1.0, torch.zeros(0), torch.eye(4), torch.eye(4),
                               0.5, 0.5, 8, 8, torch.zeros(0), 0, torch.zeros(3), False, False, False)


def test_output_taps_are_a_surfel_only_extension(hip_lib_built):
    from diff_gaussian_rasterization import GaussianRasterizationSettings as SG, GaussianRasterizer as RG
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    r = GaussianRasterizer(_settings(GaussianRasterizationSettings))
    r.set_output_taps(("normal_view",))
    r.set_output_taps(())
    with pytest.raises(NotImplementedError, match="no output tap"):
        r.set_output_taps(("depth",))
    with pytest.raises(NotImplementedError, match="no output tap"):
        RG(_settings(SG, antialiasing=False)).set_output_taps(("normal_view",))


def test_cubemap_encoder_host_side(hip_lib_built):
    from cubemapencoder import CubemapEncoder
    enc = CubemapEncoder(output_dim=3, resolution=8)
    assert enc.params["Cubemap_texture"].shape == (6, 3, 8, 8) and enc.params["Cubemap_failv"].shape == (3,)
    assert enc.n_elems == 6 * 3 * 8 * 8 + 3 and enc.seamless == 1 and enc.interp_id == 1
    assert -0.5 <= float(enc.params["Cubemap_texture"].min()) and float(enc.params["Cubemap_texture"].max()) <= 0.5
    enc.resize(16)                                                            # reference cubemap_encoder.py:102-105
    assert enc.params["Cubemap_texture"].shape == (6, 3, 16, 16) and enc.resolution == 16
    enc.filter(torch.sigmoid, lambda x: torch.log(x / (1 - x)))               # :107-113 (sharpen in activated space)
    assert torch.isfinite(enc.params["Cubemap_texture"]).all()
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        enc(torch.randn(4, 3))                                                # CHECK_CUDA of cubemapencoder.cu:23


def test_adjust_sharpness_identity_and_borders(hip_lib_built):
    from cubemapencoder.cubemap_encoder import _adjust_sharpness
    x = torch.rand(6, 3, 9, 9)
    assert torch.allclose(_adjust_sharpness(x, 1.0), x)          # factor 1 = original image
    y = _adjust_sharpness(x, 2.0)
    assert torch.equal(y[..., 0, :], x[..., 0, :]) and torch.equal(y[..., :, -1], x[..., :, -1])   # borders untouched


def test_synthetic_scene_is_seeded_and_shaped():
    a = S.make_scene(1000, "S", seed=1003, mu=-4.75)
    b = S.make_scene(1000, "S", seed=1003, mu=-4.75)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert a["scales"].shape == (1000, 2) and S.make_scene(10, "G")["scales"].shape == (10, 3)
    assert a["shs"].shape == (1000, 16, 3) and a["env_scope_mask"].dtype == bool
    np.testing.assert_allclose(np.linalg.norm(a["rotations"], axis=1), 1.0, atol=1e-5)
    assert (a["means3D"][:20, 2] <= 0.3).all()          # 2 % near-plane points exercise the cull
    cams = S.circle_cameras(64, 48, n=8)
    assert len(cams) == 8
    for c in cams:
        # every circle camera looks at the origin: the origin projects to the image centre
        o = np.array([0, 0, 0, 1.0], np.float32) @ c["projmatrix"]
        np.testing.assert_allclose(o[:2] / o[3], 0.0, atol=1e-5)


def test_flat_grads_views_and_accumulation():
    from gsr_dist import FlatGrads
    p = {"a": torch.zeros(5, 3, requires_grad=True), "b": torch.zeros(4, requires_grad=True)}
    fg = FlatGrads(p)
    ((p["a"] * 2).sum() + (p["b"] * 3).sum()).backward()
    # every slice starts on a 16-byte boundary (float4 stores of the backward kernels into sink views): "a" (15 floats) is padded to 16
    assert fg.slices == {"a": (0, 15), "b": (16, 20)} and fg.flat.numel() == 20
    assert torch.equal(fg.view("a"), torch.full((5, 3), 2.0)) and torch.equal(fg.view("b"), torch.full((4,), 3.0)) and fg.flat[15] == 0
    ((p["a"] * 1).sum()).backward()                      # accumulates in place into the same buffer
    assert torch.equal(fg.view("a"), torch.full((5, 3), 3.0))
    fg.all_reduce()                                      # no process group: no-op
    fg.zero_()
    assert float(fg.flat.abs().sum()) == 0 and p["a"].grad.data_ptr() == fg.flat.data_ptr()


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """bench.py: `--gpus 4` inside a torchrun environment of another size exits with status 2 before any GPU call (runs here, without a GPU);
    round 2's bench silently measured one GPU in that situation."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300)
    assert p.returncode == 2 and "--gpus 4" in p.stderr and "WORLD_SIZE=2" in p.stderr and not p.stdout.strip()
