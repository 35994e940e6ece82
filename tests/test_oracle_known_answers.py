"""Hand-derived known answers for the oracle: alpha rules, ordering, culling, saturation,
cubemap faces / edges / corners.  These pin the semantics the HIP kernels are then compared against."""
import numpy as np
import pytest

import gsr_synth as S
from oracle import oracle as orc

W = H = 65          # odd: the optical axis projects exactly onto pixel (32, 32) (ndc2Pix(0, 65) = 32)
C0 = 0.28209479177387814


def _cam():
    return S.make_camera(W, H, fovy_deg=60.0)


def _base(P, variant):
    cam = _cam()
    shs = np.zeros((P, 16, 3), np.float32)
    kw = dict(bg=np.array([0.1, 0.2, 0.3], np.float32), means3D=np.tile(np.array([[0, 0, 4.0]], np.float32), (P, 1)),
              opacities=np.full((P, 1), 0.5, np.float32), viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"],
              tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], image_height=H, image_width=W, sh_degree=0, shs=shs,
              refl_strengths=np.full((P, 1), 0.25, np.float32), rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)))
    if variant == "G":
        kw["scales"] = np.full((P, 3), 0.3, np.float32)
        kw["normals"] = np.tile(np.array([[0, 0, -1.0]], np.float32), (P, 1))
    else:
        kw["scales"] = np.full((P, 2), 0.3, np.float32)
        kw["env_scope_mask"] = np.ones(P, bool)
    return kw


def _oracle(variant):
    return orc.GaussOracle(np.float32) if variant == "G" else orc.SurfelOracle(np.float32)


def _set_rgb(kw, i, rgb):
    kw["shs"][i, 0, :] = (np.asarray(rgb, np.float32) - 0.5) / C0


@pytest.mark.parametrize("variant", ["G", "S"])
def test_single_gaussian_centre_pixel(variant):
    kw = _base(1, variant)
    _set_rgb(kw, 0, (0.9, 0.6, 0.2))
    o = _oracle(variant)
    out = o.forward(**kw)
    a = 0.5  # alpha = min(0.99, opacity * exp(0)) at the centre pixel
    np.testing.assert_allclose(out["color"][:, 32, 32], a * np.array([0.9, 0.6, 0.2]) + (1 - a) * kw["bg"], atol=2e-6)
    np.testing.assert_allclose(out["refl_strength_map"][0, 32, 32], a * 0.25, atol=1e-6)
    nc = o.state("n_contrib")
    ft = o.state("final_T")
    if variant == "G":
        assert nc[32, 32] == 1 and abs(ft[32, 32] - 0.5) < 1e-6
        np.testing.assert_allclose(out["invdepth"][0, 32, 32], a / 4.0, atol=1e-6)
        np.testing.assert_allclose(out["normal_map"][:, 32, 32], a * np.array([0, 0, -1.0]), atol=1e-6)
    else:
        assert nc[0, 32, 32] == 1 and abs(ft[0, 32, 32] - 0.5) < 1e-6
        am = out["allmap"][:, 32, 32]
        np.testing.assert_allclose(am[0], a * 4.0, atol=1e-5)          # expected depth
        np.testing.assert_allclose(am[1], a, atol=1e-6)                # alpha
        np.testing.assert_allclose(am[2:5], a * np.array([0, 0, -1.0]), atol=1e-6)   # normal flipped toward the camera
        np.testing.assert_allclose(am[5], 4.0, atol=1e-5)              # median depth: T = 1 > 0.5 before the update -> depth 4
        assert am[7] == 1.0                                            # env-scope mask
        assert abs(out["gaussian_weights"][0] - a) < 1e-6
    # a far-away pixel sees only the background
    np.testing.assert_allclose(out["color"][:, 0, 0], kw["bg"], atol=1e-6)


def test_surfel_median_depth_plane():
    kw = _base(1, "S")
    out = orc.SurfelOracle(np.float32).forward(**kw)
    # T (=1) > 0.5 before the update, so the single contributor defines the median depth
    assert abs(out["allmap"][5, 32, 32] - 4.0) < 1e-5


@pytest.mark.parametrize("variant", ["G", "S"])
def test_equal_depth_orders_by_index(variant):
    kw = _base(2, variant)
    _set_rgb(kw, 0, (1.0, 0.0, 0.0))
    _set_rgb(kw, 1, (0.0, 1.0, 0.0))
    out = _oracle(variant).forward(**kw)
    exp = 0.5 * np.array([1.0, 0, 0]) + 0.25 * np.array([0, 1.0, 0]) + 0.25 * kw["bg"]
    np.testing.assert_allclose(out["color"][:, 32, 32], exp, atol=2e-6)
    # swap the colours: the first index still blends first
    _set_rgb(kw, 0, (0.0, 1.0, 0.0))
    _set_rgb(kw, 1, (1.0, 0.0, 0.0))
    out = _oracle(variant).forward(**kw)
    exp = 0.5 * np.array([0, 1.0, 0]) + 0.25 * np.array([1.0, 0, 0]) + 0.25 * kw["bg"]
    np.testing.assert_allclose(out["color"][:, 32, 32], exp, atol=2e-6)


@pytest.mark.parametrize("variant", ["G", "S"])
def test_nearer_gaussian_blends_first(variant):
    kw = _base(2, variant)
    kw["means3D"][0, 2] = 5.0   # index 0 is further away
    _set_rgb(kw, 0, (1.0, 0.0, 0.0))
    _set_rgb(kw, 1, (0.0, 1.0, 0.0))
    out = _oracle(variant).forward(**kw)
    exp = 0.5 * np.array([0, 1.0, 0]) + 0.25 * np.array([1.0, 0, 0]) + 0.25 * kw["bg"]
    np.testing.assert_allclose(out["color"][:, 32, 32], exp, atol=2e-6)


@pytest.mark.parametrize("variant", ["G", "S"])
def test_near_plane_cull(variant):
    kw = _base(2, variant)
    kw["means3D"][0, 2] = 0.2          # p_view.z <= 0.2 -> culled
    kw["means3D"][1, 2] = 0.2001
    kw["scales"][:] = 0.01
    o = _oracle(variant)
    out = o.forward(**kw)
    assert out["radii"][0] == 0 and out["radii"][1] > 0
    assert list(orc.mark_visible(kw["means3D"], kw["viewmatrix"], kw["projmatrix"])) == [False, True]


@pytest.mark.parametrize("variant", ["G", "S"])
def test_prefiltered_trap(variant):
    kw = _base(1, variant)
    kw["means3D"][0, 2] = 0.1
    o = _oracle(variant)
    o.forward(prefiltered=True, **kw)
    assert o.trapped()


@pytest.mark.parametrize("variant", ["G", "S"])
def test_alpha_below_1_255_is_skipped(variant):
    kw = _base(1, variant)
    kw["opacities"][:] = 0.0039  # < 1/255 = 0.00392
    o = _oracle(variant)
    out = o.forward(**kw)
    np.testing.assert_allclose(out["color"][:, 32, 32], kw["bg"], atol=1e-7)
    nc = o.state("n_contrib")
    assert (nc[32, 32] if variant == "G" else nc[0, 32, 32]) == 0


@pytest.mark.parametrize("variant", ["G", "S"])
def test_saturation_stop(variant):
    P = 6
    kw = _base(P, variant)
    kw["opacities"][:] = 1.0          # alpha clamps to 0.99
    kw["means3D"][:, 2] = 4.0 + 0.01 * np.arange(P)
    o = _oracle(variant)
    o.forward(**kw)
    # float32 emulation of the loop: stop when T * (1 - alpha) < 1e-4 (that Gaussian is NOT blended)
    T, n = np.float32(1.0), 0
    for i in range(P):
        test = np.float32(T * (np.float32(1) - np.float32(0.99)))
        if test < np.float32(0.0001):
            break
        T, n = test, i + 1
    nc, ft = o.state("n_contrib"), o.state("final_T")
    assert (nc[32, 32] if variant == "G" else nc[0, 32, 32]) == n
    assert abs((ft[32, 32] if variant == "G" else ft[0, 32, 32]) - T) < 1e-9


def test_binning_keys_and_ranges():
    """tile-major, depth-minor keys; emission y outer / x inner; ranges cover exactly the list."""
    kw = _base(3, "G")
    kw["means3D"][:, 2] = [6.0, 4.0, 5.0]
    kw["scales"][:] = 0.5
    o = orc.GaussOracle(np.float32)
    out = o.forward(**kw)
    keys, pl, rg = o.state("keys"), o.state("point_list"), o.state("ranges")
    assert len(keys) == out["num_rendered"] == int(o.state("tiles_touched").sum())
    assert (np.diff(keys.astype(np.uint64)) >= 0).all()
    depths = o.state("depths")
    for t in range(rg.shape[0]):
        s, e = rg[t]
        assert ((keys[s:e] >> np.uint64(32)) == t).all()
        d = depths[pl[s:e]]
        assert (np.diff(d) >= 0).all()
    assert sum(int(e - s) for s, e in rg) == out["num_rendered"]


def test_antialiasing_scales_opacity():
    kw = _base(1, "G")
    kw["scales"][:] = 0.001   # much smaller than a pixel: det(cov)/det(cov+0.3I) is tiny -> clamp at sqrt(2.5e-5)
    o = orc.GaussOracle(np.float32)
    o.forward(antialiasing=True, **kw)
    co = o.state("conic_opacity")[0]
    cov_px = (0.001 * (H / (2 * np.tan(np.radians(30)))) / 4.0) ** 2
    expect = 0.5 * np.sqrt(max(2.5e-5, cov_px ** 2 / (cov_px + 0.3) ** 2))
    np.testing.assert_allclose(co[3], expect, rtol=2e-3)
    o.forward(antialiasing=False, **kw)
    assert abs(o.state("conic_opacity")[0][3] - 0.5) < 1e-7


# ---------------------------------------------------------------- cubemap
def _face_const_cubemap(L=8, C=1):
    cm = np.zeros((6, C, L, L), np.float32)
    for f in range(6):
        cm[f] = 10.0 * (f + 1)
    return cm


def test_cubemap_face_assignment():
    cm = _face_const_cubemap()
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [2, 0.3, -0.4], [0.1, -3, 0.2]], np.float32)
    for interp, seamless in ((0, 1), (1, 0), (1, 1)):
        out = orc.cubemap_forward(dirs, cm, np.zeros(1, np.float32), interp, seamless)[0]
        np.testing.assert_allclose(out, [10, 20, 30, 40, 50, 60, 10, 40], atol=1e-5)


def test_cubemap_zero_vector_returns_fail_value():
    cm = _face_const_cubemap(C=2)
    out = orc.cubemap_forward(np.zeros((1, 3), np.float32), cm, np.array([7.0, -3.0], np.float32), 1, 1)
    np.testing.assert_allclose(out[:, 0], [7.0, -3.0])
    gin, gcm, gf = orc.cubemap_backward(np.array([[2.0], [5.0]], np.float32), np.zeros((1, 3), np.float32), cm, 1, 1)
    np.testing.assert_allclose(gf, [2.0, 5.0])
    assert np.abs(gcm).max() == 0 and np.abs(gin).max() == 0


def test_cubemap_seamless_edge_and_corner():
    cm = _face_const_cubemap()
    # exactly on the +x/+y edge: half of each face (tie goes to x: face 0 with kx = ky-side weight 0.5)
    out = orc.cubemap_forward(np.array([[1, 1, 0]], np.float32), cm, np.zeros(1, np.float32), 1, 1)[0, 0]
    assert abs(out - 0.5 * (10 + 30)) < 1e-4
    # exactly at the +x+y+z corner: v11 = mean of the three faces, weights (1/4,1/4,1/4,1/4)
    out = orc.cubemap_forward(np.array([[1, 1, 1]], np.float32), cm, np.zeros(1, np.float32), 1, 1)[0, 0]
    exp = 0.25 * (10 + 30 + 50) + 0.25 * (10 + 30 + 50) / 3.0
    assert abs(out - exp) < 1e-4
    # non-seamless clamps inside the face
    out = orc.cubemap_forward(np.array([[1, 1, 1]], np.float32), cm, np.zeros(1, np.float32), 1, 0)[0, 0]
    assert abs(out - 10) < 1e-5


def test_cubemap_backward_weights_sum_to_grad():
    rs = np.random.RandomState(0)
    cm = rs.randn(6, 3, 8, 8).astype(np.float32)
    d = rs.randn(500, 3).astype(np.float32)
    go = rs.randn(3, 500).astype(np.float32)
    for interp, seamless in ((0, 1), (1, 0), (1, 1)):
        gin, gcm, gf = orc.cubemap_backward(go, d, cm, interp, seamless)
        # interpolation weights of every lookup sum to one -> texel gradients sum to the upstream gradient sum
        np.testing.assert_allclose(gcm.sum(axis=(0, 2, 3)), go.sum(axis=1), rtol=1e-4, atol=1e-4)
