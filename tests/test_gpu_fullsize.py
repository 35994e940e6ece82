"""GPU parity at the full BASELINE sizes (config C3: 10^6 surfels, 1920x1080, SH 3; config C5: 5 x 10^6 Gaussians with
anti-aliasing and the inverse-depth backward): size-independent properties of the
binning state, bit-identity of the outputs with per-wave culling on and off, linearity of the backward in the upstream
gradients, and the complete oracle comparison (the OpenMP oracle needs ~40 s for this step on the GPU box's host)."""
import numpy as np
import pytest
import torch

from helpers import GATE_BUDGET, HipSurfel, S, assert_planes_psnr, grad_gate, n_contrib_ok, psnr, rel_maxnorm, scene_kwargs

pytestmark = pytest.mark.gpu
P, W, H = 1_000_000, 1920, 1080


@pytest.fixture(scope="module")
def c3():
    kw, cam, sc = scene_kwargs("S", P, W, H, 1003, -4.75, 3, (0, 0, 0))
    return kw


def test_c3_binning_state_properties(c3):
    hip = HipSurfel(c3)          # (with autograd: the workspace buffers are reached through the graph node)
    R = hip.R
    tt = hip.state("tiles_touched").astype(np.int64)
    off = hip.state("point_offsets").astype(np.int64)
    assert R == int(tt.sum()) == int(off[-1]) and (np.diff(off) == tt[1:]).all()
    keys = hip.state("keys").astype(np.uint64)
    pl = hip.state("point_list").astype(np.int64)
    rg = hip.state("ranges").astype(np.int64)
    assert (keys[1:] >= keys[:-1]).all()                                   # (tile, depth) order
    same = keys[1:] == keys[:-1]
    assert (pl[1:][same] > pl[:-1][same]).all()                            # ties broken by Gaussian index (stable sort)
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    counts = np.bincount(tiles, minlength=rg.shape[0])
    assert ((rg[:, 1] - rg[:, 0]) == counts).all() and (rg[counts > 0, 0] == np.concatenate([[0], np.cumsum(counts)[:-1]])[counts > 0]).all()
    depth_bits = hip.state("depths").view(np.uint32).astype(np.uint64)
    assert ((keys & np.uint64(0xFFFFFFFF)) == depth_bits[pl]).all()        # every instance carries its Gaussian's depth
    assert (np.bincount(pl, minlength=P) == tt).all()                      # each Gaussian appears once per touched tile
    out = hip.out()
    assert np.abs(out["allmap"][1] - (1.0 - hip.state("final_T")[0])).max() == 0
    assert out["allmap"][1].min() >= 0 and out["allmap"][1].max() <= 1.0


def test_c3_cull_bit_identity_and_backward_linearity(c3):
    import _gsr
    g1 = S.make_upstream_grads(H, W, 11)
    g2 = S.make_upstream_grads(H, W, 12)
    outs, grads = [], []
    try:
        for cull in (1, 0):
            _gsr.set_option("cull", cull)
            hip = HipSurfel(c3)
            outs.append(hip.out())
            if cull:
                n_contrib = hip.state("n_contrib")
                grads.append(hip.backward(g1["dL_dcolor"], g1["dL_dplanes"], g1["dL_drefl"]))
            else:
                assert (hip.state("n_contrib") == n_contrib).all()
    finally:
        _gsr.set_option("cull", 1)
    for k in ("color", "allmap", "refl_strength_map", "gaussian_weights", "radii"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    hip = HipSurfel(c3)
    grads.append(hip.backward(g2["dL_dcolor"], g2["dL_dplanes"], g2["dL_drefl"]))
    hip = HipSurfel(c3)
    both = hip.backward(g1["dL_dcolor"] + g2["dL_dcolor"], g1["dL_dplanes"] + g2["dL_dplanes"], g1["dL_drefl"] + g2["dL_drefl"])
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_drefl_strengths"):
        s = grads[0][k] + grads[1][k]
        assert np.isfinite(both[k]).all()
        assert rel_maxnorm(both[k], s) <= 1e-4, k


def test_c3_against_oracle(c3):
    from oracle import oracle as orc
    o = orc.SurfelOracle(np.float32)
    ref = o.forward(**c3)
    hip = HipSurfel(c3)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"]
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    np.testing.assert_array_equal(hip.state("point_list").astype(np.uint32), o.state("point_list"))
    nc_h, nc_o = hip.state("n_contrib"), o.state("n_contrib")
    assert n_contrib_ok(nc_h, nc_o)
    assert psnr(out["color"], ref["color"]) >= 50
    assert_planes_psnr(out["allmap"], ref["allmap"])          # every plane against its own peak
    g = S.make_upstream_grads(H, W, 1003)
    gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dallmap=g["dL_dplanes"], dL_drefl_strength_map=g["dL_drefl"])
    gh = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    # dL_dmeans2D is the densification signal (the reference's overwrite, DSR backward.cu:656-659), compared like the rest
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_drefl_strengths"):
        assert rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k]) <= 1e-4, k
        bad = grad_gate(gh[k], gr[k])
        assert bad <= GATE_BUDGET, (k, "elementwise gate", bad)


def test_full_size_gauss_variant_against_oracle():
    """Variant G at the same size (10^6 Gaussians, 1080p, SH 3, anti-aliasing + inverse depth)."""
    from helpers import HipGauss
    from oracle import oracle as orc
    kw, cam, sc = scene_kwargs("G", P, W, H, 1003, -4.75, 3, (0, 0, 0))
    o = orc.GaussOracle(np.float32)
    ref = o.forward(antialiasing=True, **kw)
    hip = HipGauss(kw, antialiasing=True)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"]
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    np.testing.assert_array_equal(hip.state("point_list").astype(np.uint32), o.state("point_list"))
    assert n_contrib_ok(hip.state("n_contrib").astype(np.int64), o.state("n_contrib").astype(np.int64))
    for k in ("color", "normal_map", "invdepth", "refl_strength_map"):
        assert psnr(out[k], ref[k], peak=max(1.0, float(np.abs(ref[k]).max()))) >= 50, k
    g = S.make_upstream_grads(H, W, 1003)
    gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dinvdepth=g["dL_dinvdepth"], dL_dnormal_map=g["dL_dnormal"], dL_drefl_strength_map=g["dL_drefl"])
    gh = hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dnormals", "dL_drefl_strengths"):
        assert rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k]) <= 1e-4, k
        bad = grad_gate(gh[k], gr[k])
        assert bad <= GATE_BUDGET, (k, "elementwise gate", bad)


def test_full_size_gauss_variant_cull_bit_identity():
    """Variant G with and without the per-wave footprint vote: the vote may only drop (wave, Gaussian) pairs in which no
    pixel reaches alpha >= 1/255, so every output and the contributor counts must be bit-identical."""
    import _gsr
    from helpers import HipGauss
    kw, cam, sc = scene_kwargs("G", P, W, H, 1003, -4.75, 3, (0, 0, 0))
    outs, ncs = [], []
    try:
        for cull in (1, 0):
            _gsr.set_option("cull", cull)
            hip = HipGauss(kw, antialiasing=True)
            outs.append(hip.out())
            ncs.append(hip.state("n_contrib"))
    finally:
        _gsr.set_option("cull", 1)
    assert np.array_equal(ncs[0], ncs[1])
    for k in ("color", "normal_map", "invdepth", "refl_strength_map", "radii"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_c5_gauss_5m_antialiasing_inverse_depth_against_oracle():
    """BASELINE config C5: 5 x 10^6 Gaussians, 1920x1080, SH 3, anti-aliasing on, inverse-depth (depth-regularisation)
    backward enabled with a non-zero upstream gradient — variant G end to end against the oracle (DGR forward.cu:151-269,
    backward.cu:147-326,399-449).  The OpenMP oracle needs a few seconds per direction on the GPU box's host cores."""
    from helpers import HipGauss
    from oracle import oracle as orc
    P5 = 5_000_000
    kw, cam, sc = scene_kwargs("G", P5, W, H, 1005, -5.3, 3, (0, 0, 0))
    o = orc.GaussOracle(np.float32)
    ref = o.forward(antialiasing=True, **kw)
    hip = HipGauss(kw, antialiasing=True)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"] and out["num_rendered"] > 3 * P5
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    np.testing.assert_array_equal(hip.state("point_list").astype(np.uint32), o.state("point_list"))
    assert n_contrib_ok(hip.state("n_contrib").astype(np.int64), o.state("n_contrib").astype(np.int64))
    for k in ("color", "normal_map", "invdepth", "refl_strength_map"):
        assert psnr(out[k], ref[k], peak=max(1.0, float(np.abs(ref[k]).max()))) >= 50, k
    g = S.make_upstream_grads(H, W, 1005)
    assert np.abs(g["dL_dinvdepth"]).max() > 0
    gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dinvdepth=g["dL_dinvdepth"], dL_dnormal_map=g["dL_dnormal"], dL_drefl_strength_map=g["dL_drefl"])
    gh = hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dnormals", "dL_drefl_strengths"):
        assert rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k]) <= 1e-4, k
        bad = grad_gate(gh[k], gr[k])
        assert bad <= GATE_BUDGET, (k, "elementwise gate", bad)


def test_c3_rotated_view_and_coloured_background_against_oracle():
    """A second full-size comparison with other list statistics: the C4 batch's last view (camera turned 21 degrees about the y
    axis, so that part of the scene leaves the frustum and the tile lists are lopsided), a non-black background, other seeds."""
    import math
    from oracle import oracle as orc
    a = math.radians(21.0)
    c2w = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=np.float64)
    cam = S.make_camera(W, H, R=c2w, T=np.zeros(3))
    kw, _, _ = scene_kwargs("S", P, W, H, 2024, -4.75, 3, (0.3, 0.1, 0.6), cam=cam)
    o = orc.SurfelOracle(np.float32)
    ref = o.forward(**kw)
    hip = HipSurfel(kw)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"] and (ref["radii"] == 0).mean() > 0.02      # some of the scene is culled
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    np.testing.assert_array_equal(hip.state("point_list").astype(np.uint32), o.state("point_list"))
    assert n_contrib_ok(hip.state("n_contrib"), o.state("n_contrib"))
    assert psnr(out["color"], ref["color"]) >= 50
    assert_planes_psnr(out["allmap"], ref["allmap"])
    g = S.make_upstream_grads(H, W, 2024)
    gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dallmap=g["dL_dplanes"], dL_drefl_strength_map=g["dL_drefl"])
    gh = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_drefl_strengths"):
        assert rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k]) <= 1e-4, k
        assert grad_gate(gh[k], gr[k]) <= GATE_BUDGET, (k, "elementwise gate")
