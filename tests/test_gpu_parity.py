"""GPU parity tests proper: HIP path (through the drop-in Python API -> ctypes -> C ABI) vs the CPU
oracle on identical seeded inputs.  Bars (BASELINE.md §4 / north star):
  * integer / index work bit-exact (radii, tile counts, offsets, sort keys, point list, tile ranges)
  * render PSNR >= 50 dB (we also assert much tighter absolute bounds)
  * gradients within 1e-4 relative: max-norm per tensor AND elementwise (|a-b| <= 1e-4 |b| + 1e-6 max|b|, failing
    fraction <= 1e-5: helpers.grad_gate)
n_contrib (per-pixel last contributor) depends on exp() ulps through the alpha < 1/255 and T < 1e-4
thresholds; it is compared with a mismatch budget of 3e-5 of the pixels (helpers.N_CONTRIB_BUDGET).
Absolute bounds are ~10x what tests/observed_errors.py measured on MI355X in round 3: colour max-abs 2.8e-6 (S) / 4.8e-5 (G) at C2,
gaussian_weights 8.6e-7 relative / 1.2e-7 absolute.
"""
import numpy as np
import pytest

from helpers import GATE_BUDGET, HipGauss, HipSurfel, S, assert_image_close, grad_gate, n_contrib_ok, psnr, rel_maxnorm, scene_kwargs

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-4
PSNR_MIN = 50.0


def _oracle():
    from oracle import oracle as orc
    return orc


def _check_binning(hip, o, exact_float_state=()):
    np.testing.assert_array_equal(hip.state("tiles_touched").astype(np.uint32), o.state("tiles_touched"))
    np.testing.assert_array_equal(hip.state("point_offsets").astype(np.uint32), o.state("point_offsets"))
    vis = o.state("radii") > 0
    np.testing.assert_array_equal(hip.state("depths").view(np.uint32)[vis], o.state("depths").view(np.uint32)[vis])
    np.testing.assert_array_equal(hip.state("keys").astype(np.uint64), o.state("keys"))
    np.testing.assert_array_equal(hip.state("point_list").astype(np.uint32), o.state("point_list"))
    np.testing.assert_array_equal(hip.state("ranges").astype(np.uint32), o.state("ranges"))
    np.testing.assert_array_equal(hip.state("clamped")[vis], o.state("clamped")[vis])


def _run_surfel(P, W, H, seed, mu, sh_degree, bg, mask_radius=0.0, backward=True, sh_rows=16):
    orc = _oracle()
    kw, cam, sc = scene_kwargs("S", P, W, H, seed, mu, sh_degree, bg, mask_radius)
    kw["shs"] = np.ascontiguousarray(kw["shs"][:, :sh_rows])
    o = orc.SurfelOracle(np.float32)
    ref = o.forward(**kw)
    hip = HipSurfel(kw)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"]
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    _check_binning(hip, o)
    nc_h, nc_o = hip.state("n_contrib").astype(np.uint32), o.state("n_contrib")
    assert n_contrib_ok(nc_h[0], nc_o[0]) and n_contrib_ok(nc_h[1], nc_o[1])
    assert psnr(out["color"], ref["color"]) >= PSNR_MIN
    assert_image_close(out["color"], ref["color"], 3e-5)
    assert psnr(out["refl_strength_map"], ref["refl_strength_map"]) >= PSNR_MIN
    for plane in range(8):
        scale = max(1.0, float(np.abs(ref["allmap"][plane]).max()))
        assert psnr(out["allmap"][plane], ref["allmap"][plane], peak=scale) >= PSNR_MIN, plane
    # gaussian_weights: true max in both; tolerance for exp() ulps
    gw_h, gw_o = out["gaussian_weights"].astype(np.float64), ref["gaussian_weights"].astype(np.float64)
    gw_bad = np.abs(gw_h - gw_o) > 1.5e-6 + 1e-5 * np.abs(gw_o)
    # (a surfel whose maximum comes from a pair sitting on the alpha = 1/255 threshold may differ by its whole weight, <= 1/255: same budget as n_contrib)
    assert int(gw_bad.sum()) <= max(2, int(3e-5 * gw_o.size)) and np.abs(gw_h - gw_o).max() <= 5e-3, (int(gw_bad.sum()), np.abs(gw_h - gw_o).max())
    if not backward:
        return
    g = S.make_upstream_grads(H, W, seed)
    gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dallmap=g["dL_dplanes"], dL_drefl_strength_map=g["dL_drefl"])
    gh = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_drefl_strengths", "dL_dscales", "dL_drotations"):
        err = rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k])
        assert err <= GRAD_TOL, (k, err)
        bad = grad_gate(gh[k], gr[k], GRAD_TOL)
        assert bad <= GATE_BUDGET, (k, "elementwise gate", bad)


def _run_gauss(P, W, H, seed, mu, sh_degree, bg, antialiasing=False, backward=True, sh_rows=16):
    orc = _oracle()
    kw, cam, sc = scene_kwargs("G", P, W, H, seed, mu, sh_degree, bg)
    kw["shs"] = np.ascontiguousarray(kw["shs"][:, :sh_rows])
    o = orc.GaussOracle(np.float32)
    ref = o.forward(antialiasing=antialiasing, **kw)
    hip = HipGauss(kw, antialiasing=antialiasing)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"]
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    _check_binning(hip, o)
    nc_h, nc_o = hip.state("n_contrib").astype(np.uint32)[0], o.state("n_contrib")
    assert n_contrib_ok(nc_h, nc_o)
    for k in ("color", "normal_map", "refl_strength_map", "invdepth"):
        scale = max(1.0, float(np.abs(ref[k]).max()))
        assert psnr(out[k], ref[k], peak=scale) >= PSNR_MIN, k
    assert_image_close(out["color"], ref["color"], 5e-4)
    if not backward:
        return
    g = S.make_upstream_grads(H, W, seed)
    gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dinvdepth=g["dL_dinvdepth"], dL_dnormal_map=g["dL_dnormal"],
                    dL_drefl_strength_map=g["dL_drefl"])
    gh = hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dnormals", "dL_drefl_strengths", "dL_dscales", "dL_drotations"):
        err = rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k])
        assert err <= GRAD_TOL, (k, err)
        bad = grad_gate(gh[k], gr[k], GRAD_TOL)
        assert bad <= GATE_BUDGET, (k, "elementwise gate", bad)


# ---- BASELINE config C1: 10k / 256x256 / SH deg 0, forward only (plumbing) ----
def test_c1_surfel_forward():
    _run_surfel(10_000, 256, 256, 1001, -3.0, 0, (0, 0, 0), backward=False)


def test_c1_gauss_forward():
    _run_gauss(10_000, 256, 256, 1001, -3.0, 0, (0, 0, 0), backward=False)


# ---- small fwd+bwd, SH3, white background, ragged image size (not a multiple of 16) ----
def test_small_surfel_fwd_bwd_ragged():
    _run_surfel(5_000, 200, 136, 7, -3.0, 3, (1, 1, 1), mask_radius=4.5)


def test_small_gauss_fwd_bwd_ragged_aa():
    _run_gauss(5_000, 200, 136, 7, -3.0, 3, (1, 1, 1), antialiasing=True)


def test_small_gauss_fwd_bwd_no_aa():
    _run_gauss(5_000, 256, 256, 8, -3.0, 2, (0, 0, 0), antialiasing=False)


# ---- BASELINE config C2: 100k / 800x800 / SH3 forward+backward ----
def test_c2_surfel_fwd_bwd():
    _run_surfel(100_000, 800, 800, 1002, -3.6, 3, (0, 0, 0))


def test_c2_gauss_fwd_bwd():
    _run_gauss(100_000, 800, 800, 1002, -3.6, 3, (0, 0, 0), antialiasing=True)


# ---- a tile grid with more than 255 tiles along one axis: the depth sort cannot carry the tile rectangle packed to 4 x 8 bits beside the
# index (OrderRect, csrc/gsr_common.hip), key emission gathers the 8-byte rectangle through the sorted index instead (emit_tiles_kernel<true>) ----
def test_surfel_grid_wider_than_255_tiles():
    _run_surfel(4_000, 4112, 48, 21, -3.2, 3, (0, 0, 0))


def test_gauss_grid_taller_than_255_tiles():
    _run_gauss(4_000, 40, 4104, 22, -3.2, 3, (0, 0, 0), antialiasing=True)


# ---- key emission with two Gaussians per thread (chosen from 3 M Gaussians up; forced here), on both forms of the tile rectangle ----
@pytest.mark.parametrize("variant,P,W,H", [("S", 5_000, 200, 136), ("G", 4_097, 256, 256), ("S", 3_001, 4112, 48)])
def test_key_emission_two_gaussians_per_thread(variant, P, W, H):
    import _gsr
    try:
        _gsr.set_option("emit_items", 2)
        if variant == "S":
            _run_surfel(P, W, H, 31, -3.0, 3, (0, 0, 0), backward=False)
        else:
            _run_gauss(P, W, H, 32, -3.0, 3, (0, 0, 0), antialiasing=True, backward=False)
    finally:
        _gsr.set_option("emit_items", 0)


# ---- the per-Gaussian kernels store their AoS rows wave by wave (64 rows through LDS): Gaussian counts that end inside a wave,
# exactly at one, one past it, and SH tensors whose rows are not 16 coefficients long (the row-by-row fall-back) ----
@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 5, 64, 65, 257])
def test_gaussian_counts_around_wave_boundaries(P):
    _run_surfel(P, 96, 64, 20 + P, -1.6, 3, (0.2, 0.1, 0.3))
    _run_gauss(P, 96, 64, 40 + P, -1.6, 3, (0.2, 0.1, 0.3), antialiasing=True)


@pytest.mark.gpu
@pytest.mark.parametrize("sh_degree,sh_rows", [(1, 4), (2, 9), (0, 1)])
def test_short_sh_rows(sh_degree, sh_rows):
    _run_surfel(700, 96, 64, 61, -2.0, sh_degree, (0, 0, 0), sh_rows=sh_rows)
    _run_gauss(700, 96, 64, 62, -2.0, sh_degree, (0, 0, 0), sh_rows=sh_rows)


# ---- per-wave culling must not change a single bit of any output (it only skips pairs that cannot blend) ----
@pytest.mark.parametrize("variant", ["S", "G"])
def test_cull_is_bit_exact(variant):
    import _gsr
    P, W, H = 30_000, 400, 300
    kw, cam, sc = scene_kwargs(variant, P, W, H, 21, -3.3, 3, (0.3, 0.2, 0.1))
    # a few huge / degenerate / near-plane Gaussians to exercise the unbounded-box paths
    kw["scales"][:50] *= 40.0
    kw["means3D"][50:100, 2] = 0.25
    kw["opacities"][100:150] = 0.003
    g = S.make_upstream_grads(H, W, 21)
    res = []
    for cull in (0, 1):
        _gsr.set_option("cull", cull)
        try:
            if variant == "S":
                hip = HipSurfel(kw)
                out, nc, ft = hip.out(), hip.state("n_contrib"), hip.state("final_T")
                gh = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
            else:
                hip = HipGauss(kw, antialiasing=True)
                out, nc, ft = hip.out(), hip.state("n_contrib"), hip.state("final_T")
                gh = hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])
            res.append((out, gh, nc, ft))
        finally:
            _gsr.set_option("cull", 1)
    (o0, g0, n0, t0), (o1, g1, n1, t1) = res
    for k in o0:
        if isinstance(o0[k], np.ndarray):
            np.testing.assert_array_equal(o0[k], o1[k], err_msg=k)
    np.testing.assert_array_equal(n0, n1)
    np.testing.assert_array_equal(t0, t1)
    # gradients are float-atomic sums over thousands of terms (arrival order differs from run to run): compare tightly,
    # not bitwise
    for k in g0:
        if g0[k] is not None:
            assert rel_maxnorm(g1[k], g0[k]) <= 5e-5, k
