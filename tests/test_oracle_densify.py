"""CPU: known-answer cases for the densification oracle (oracle/oracle_densify.py)."""
import numpy as np

from oracle import oracle_densify as od


def _model(P=6):
    rs = np.random.RandomState(0)
    p = dict(means3D=rs.randn(P, 3), shs=rs.randn(P, 16, 3), opacities=rs.randn(P, 1), scales=np.full((P, 2), np.log(0.001)),
             rotations=np.tile([1.0, 0, 0, 0], (P, 1)), refl_strengths=rs.randn(P, 1))
    m = {k: np.ones_like(v) for k, v in p.items()}
    v = {k: 2 * np.ones_like(v) for k, v in p.items()}
    s = dict(xyz_gradient_accum=np.zeros(P), denom=np.ones(P), accum_w=np.ones(P), denom_w=np.ones(P), max_radii2D=np.zeros(P))
    return od.Model(p, m, v, s)


def test_stats_accumulate_only_where_visible_or_weighted():
    s = {k: np.zeros(4, np.float32) for k in ("xyz_gradient_accum", "denom", "accum_w", "denom_w", "max_radii2D")}
    g = np.array([[3, 4, 0], [1, 0, 0], [0, 0, 0], [6, 8, 0]], np.float32)
    od.add_densification_stats(s, g, np.array([5, 0, 2, 7]), np.array([0.5, 0.25, 0.0, 0.0], np.float32))
    assert s["xyz_gradient_accum"].tolist() == [5.0, 0.0, 0.0, 10.0] and s["denom"].tolist() == [1, 0, 1, 1]
    assert s["max_radii2D"].tolist() == [5, 0, 2, 7] and s["accum_w"].tolist() == [0.5, 0.25, 0, 0] and s["denom_w"].tolist() == [1, 1, 0, 0]


def test_prune_clone_split_order_and_optimizer_state():
    mdl = _model(6)
    mdl.s["accum_w"][0] = 0.001                     # row 0: pruned by weight
    mdl.s["xyz_gradient_accum"][[1, 2]] = 1.0       # rows 1, 2: large gradient
    mdl.p["scales"][2] = np.log(0.5)                # row 2 is big -> split; row 1 small -> clone
    x1, x2 = mdl.p["means3D"][1].copy(), mdl.p["means3D"][2].copy()
    noise = np.array([[1.0, 0.0], [0.0, -2.0]], np.float32)
    nc, ns = mdl.densify_and_prune(0.0002, 0.05, np.zeros(3, np.float32), 1.0, None, noise)
    assert (nc, ns) == (1, 1)
    # originals 1,3,4,5 (2 removed as split parent), then the clone of 1, then the two children of 2
    assert mdl.p["means3D"].shape[0] == 7
    np.testing.assert_array_equal(mdl.p["means3D"][0], x1)
    np.testing.assert_array_equal(mdl.p["means3D"][4], x1)
    np.testing.assert_allclose(mdl.p["means3D"][5], x2 + [0.5, 0, 0], rtol=1e-6)        # identity rotation: offset = scale * noise
    np.testing.assert_allclose(mdl.p["means3D"][6], x2 + [0, -1.0, 0], rtol=1e-6)
    np.testing.assert_allclose(mdl.p["scales"][5], np.log(0.5 / 1.6), rtol=1e-6)
    assert (mdl.m["shs"][:4] == 1).all() and (mdl.m["shs"][4:] == 0).all() and (mdl.v["opacities"][4:] == 0).all()
    assert all((a == 0).all() and a.shape == (7,) for a in mdl.s.values())


def test_big_point_prune_uses_world_size_only():
    mdl = _model(5)
    mdl.p["scales"][3] = np.log(0.2)                # > 0.1 * extent and inside -> pruned when max_screen_size is set
    mdl.s["max_radii2D"][:] = 1000.0                # reset to zero by the postfix: never prunes (reference quirk)
    mdl.p["means3D"][:] *= 0.1
    mdl.densify_and_prune(10.0, 0.05, np.zeros(3, np.float32), 1.0, 20, np.zeros((0, 2), np.float32))
    assert mdl.p["means3D"].shape[0] == 4
