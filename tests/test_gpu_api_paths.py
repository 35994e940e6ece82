"""GPU parity for the less-travelled API paths and the reference's edge cases: precomputed colours, precomputed
transMat / cov3D, SH degrees and coefficient counts, scale_modifier (incl. the surfel backward ignoring it), P = 0,
markVisible, the prefiltered trap, debug mode, the multi-view ball scene of config C4, maximum-size keys."""
import numpy as np
import pytest
import torch

from helpers import HipGauss, HipSurfel, S, assert_planes_psnr, psnr, rel_maxnorm, scene_kwargs

pytestmark = pytest.mark.gpu


def _orc():
    from oracle import oracle as orc
    return orc


def _cmp_surfel(kw, grads=("dL_dmeans3D", "dL_dopacity", "dL_drefl_strengths"), scale_modifier=1.0, tol=1e-4):
    orc = _orc()
    H, W = kw["image_height"], kw["image_width"]
    o = orc.SurfelOracle(np.float32)
    ref = o.forward(scale_modifier=scale_modifier, **kw)
    hip = HipSurfel(kw, scale_modifier=scale_modifier)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"]
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    assert psnr(out["color"], ref["color"]) >= 50
    assert_planes_psnr(out["allmap"], ref["allmap"])
    g = S.make_upstream_grads(H, W, 3)
    gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dallmap=g["dL_dplanes"], dL_drefl_strength_map=g["dL_drefl"])
    gh = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    for k in grads:
        assert gh[k] is not None, k
        assert rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k]) <= tol, k
    return ref, out, gr, gh


def _cmp_gauss(kw, grads=("dL_dmeans3D", "dL_dopacity", "dL_dnormals", "dL_drefl_strengths"), aa=False, scale_modifier=1.0):
    orc = _orc()
    H, W = kw["image_height"], kw["image_width"]
    o = orc.GaussOracle(np.float32)
    ref = o.forward(antialiasing=aa, scale_modifier=scale_modifier, **kw)
    hip = HipGauss(kw, antialiasing=aa, scale_modifier=scale_modifier)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"]
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    for k in ("color", "normal_map", "invdepth", "refl_strength_map"):
        assert psnr(out[k], ref[k], peak=max(1.0, float(np.abs(ref[k]).max()))) >= 50, k
    g = S.make_upstream_grads(H, W, 3)
    gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dinvdepth=g["dL_dinvdepth"], dL_dnormal_map=g["dL_dnormal"], dL_drefl_strength_map=g["dL_drefl"])
    gh = hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])
    for k in grads:
        assert gh[k] is not None, k
        assert rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k]) <= 1e-4, k
    return ref, out, gr, gh


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_sh_degrees_surfel_and_gauss(deg):
    kw, _, _ = scene_kwargs("S", 3000, 160, 120, 40 + deg, -2.8, deg, (0.1, 0.1, 0.1))
    _cmp_surfel(kw, grads=("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations"))
    kw, _, _ = scene_kwargs("G", 3000, 160, 120, 50 + deg, -2.8, deg, (0.1, 0.1, 0.1))
    ref, out, gr, gh = _cmp_gauss(kw, grads=("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations"))
    ncoef = (deg + 1) ** 2
    assert np.abs(gh["dL_dsh"][:, ncoef:, :]).max() == 0   # coefficients above the active degree get exact zeros


def test_sh_with_four_coefficients_only():
    """M = 4 (degree-1 tensor): rows are 48 bytes, exercises the non-192-byte row path."""
    kw, _, _ = scene_kwargs("G", 2000, 128, 96, 61, -2.8, 1, (0, 0, 0))
    kw["shs"] = np.ascontiguousarray(kw["shs"][:, :4, :])
    _cmp_gauss(kw, grads=("dL_dsh", "dL_dmeans3D"))
    kw, _, _ = scene_kwargs("S", 2000, 128, 96, 62, -2.8, 0, (0, 0, 0))
    kw["shs"] = np.ascontiguousarray(kw["shs"][:, :1, :])    # M = 1: 12-byte rows, scalar-load fallback
    _cmp_surfel(kw, grads=("dL_dsh", "dL_dmeans3D"))


def test_precomputed_colors():
    rs = np.random.RandomState(5)
    kw, _, _ = scene_kwargs("S", 3000, 160, 120, 70, -2.8, 3, (1, 1, 1))
    kw["colors_precomp"] = rs.rand(3000, 3).astype(np.float32)
    kw["shs"] = None
    ref, out, gr, gh = _cmp_surfel(kw, grads=("dL_dmeans3D", "dL_dcolors", "dL_dopacity"))
    kw, _, _ = scene_kwargs("G", 3000, 160, 120, 71, -2.8, 3, (1, 1, 1))
    kw["colors_precomp"] = rs.rand(3000, 3).astype(np.float32)
    kw["shs"] = None
    _cmp_gauss(kw, grads=("dL_dmeans3D", "dL_dcolors", "dL_dopacity"))


def test_surfel_precomputed_transmat():
    """pipe.compute_cov3D_python path: the caller supplies T (P,9) and gets dL_dtransMat back (DSR backward.cu:499-575)."""
    orc = _orc()
    kw, _, _ = scene_kwargs("S", 3000, 160, 120, 80, -2.8, 3, (0, 0, 0))
    o = orc.SurfelOracle(np.float32)
    o.forward(**kw)
    T = o.state("transMat")
    vis = o.state("radii") > 0
    T[~vis] = np.eye(3, dtype=np.float32).reshape(-1)     # culled Gaussians never wrote their T
    kw2 = dict(kw)
    kw2["cov3D_precomp"] = T
    kw2["scales"] = None
    kw2["rotations"] = None
    _cmp_surfel(kw2, grads=("dL_dtransMat", "dL_dmeans3D", "dL_dopacity"))


def test_gauss_precomputed_cov3d():
    orc = _orc()
    kw, _, _ = scene_kwargs("G", 3000, 160, 120, 81, -2.8, 3, (0, 0, 0))
    o = orc.GaussOracle(np.float32)
    o.forward(**kw)
    cov = o.state("cov3D")
    vis = o.state("radii") > 0
    cov[~vis] = np.array([1e-2, 0, 0, 1e-2, 0, 1e-2], np.float32)
    kw2 = dict(kw)
    kw2["cov3D_precomp"] = cov
    kw2["scales"] = None
    kw2["rotations"] = None
    _cmp_gauss(kw2, grads=("dL_dcov3D", "dL_dmeans3D"), aa=True)


def test_scale_modifier():
    """Forward honours scale_modifier; the surfel backward rebuilds T with modifier 1 (DSR backward.cu:511) and the
    oracle reproduces that, so parity must hold for modifier != 1 as well."""
    kw, _, _ = scene_kwargs("S", 3000, 160, 120, 90, -3.0, 3, (0, 0, 0))
    _cmp_surfel(kw, grads=("dL_dmeans3D", "dL_dscales", "dL_drotations"), scale_modifier=1.7)
    kw, _, _ = scene_kwargs("G", 3000, 160, 120, 91, -3.0, 3, (0, 0, 0))
    _cmp_gauss(kw, grads=("dL_dmeans3D", "dL_dscales", "dL_drotations"), scale_modifier=0.6)


def test_c4_ball_scene_circle_cameras():
    """Config C4 geometry: Gaussians in a ball around the origin seen from cameras on a circle (non-identity view matrix)."""
    cams = S.circle_cameras(200, 150, n=8)
    for v in (1, 6):
        kw, _, _ = scene_kwargs("S", 6000, 200, 150, 1004, -2.6, 3, (0, 0, 0), cam=cams[v], ball=True)
        _cmp_surfel(kw, grads=("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dopacity"))
    kw, _, _ = scene_kwargs("G", 6000, 200, 150, 1004, -2.6, 3, (0, 0, 0), cam=cams[3], ball=True)
    _cmp_gauss(kw, grads=("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dopacity"), aa=True)


def test_empty_and_tiny_inputs():
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam = S.make_camera(64, 48)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    st = GaussianRasterizationSettings(image_height=48, image_width=64, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                       bg=torch.tensor([0.2, 0.3, 0.4]).cuda(), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]),
                                       projmatrix=t(cam["projmatrix"]), sh_degree=0, campos=t(cam["campos"]), prefiltered=False, debug=False)
    r = GaussianRasterizer(st)
    z = lambda *s: torch.zeros(*s, device="cuda")
    # P = 0: zero outputs, num_rendered 0 (DSR rasterize_points.cu:111)
    color, radii, allmap, refl, gw = r(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), shs=z(0, 16, 3), refl_strengths=z(0, 1),
                                       scales=z(0, 2), rotations=z(0, 4), env_scope_mask=torch.zeros(0, dtype=torch.bool, device="cuda"))
    assert color.shape == (3, 48, 64) and float(color.abs().max()) == 0 and radii.numel() == 0 and float(allmap.abs().max()) == 0
    # everything culled (behind the near plane): image = background, no instances
    m = z(5, 3)
    color, radii, allmap, refl, gw = r(means3D=m, means2D=z(5, 3), opacities=z(5, 1) + 0.5, shs=z(5, 16, 3), refl_strengths=z(5, 1),
                                       scales=z(5, 2) + 0.1, rotations=torch.tensor([[1.0, 0, 0, 0]] * 5).cuda(),
                                       env_scope_mask=torch.ones(5, dtype=torch.bool, device="cuda"))
    assert int(radii.max()) == 0
    np.testing.assert_allclose(color[:, 10, 10].cpu().numpy(), [0.2, 0.3, 0.4], atol=1e-7)
    assert list(r.markVisible(torch.tensor([[0, 0, 1.0], [0, 0, 0.1], [0, 0, 0.2]]).cuda()).cpu().numpy()) == [True, False, False]


def test_prefiltered_trap_and_debug_mode():
    import _gsr
    kw, _, _ = scene_kwargs("S", 500, 64, 48, 7, -2.5, 0, (0, 0, 0))
    with pytest.raises(RuntimeError, match="filtered although prefiltered"):   # (GsrError through ctypes, RuntimeError through the compiled binding)
        HipSurfel(kw, prefiltered=True)          # the scene contains near-plane points: the reference would __trap()
    hip = HipSurfel(kw, debug=True)              # debug: synchronise + check after every stage
    assert hip.R > 0


def test_max_tile_id_bits_and_large_splats():
    """A few very large splats cover every tile (keys use all getHigherMsb(tiles) bits); checks the sort bit range."""
    orc = _orc()
    kw, _, _ = scene_kwargs("G", 300, 1920, 1080, 9, -1.0, 0, (0, 0, 0))
    o = orc.GaussOracle(np.float32)
    ref = o.forward(**kw)
    hip = HipGauss(kw)
    assert hip.R == ref["num_rendered"]
    np.testing.assert_array_equal(hip.state("point_list").astype(np.uint32), o.state("point_list"))
    np.testing.assert_array_equal(hip.state("ranges").astype(np.uint32), o.state("ranges"))
    assert int(o.state("keys").max() >> np.uint64(32)) == 120 * 68 - 1
    assert psnr(hip.out()["color"], ref["color"]) >= 50


def _sink_params(hip):
    return dict(means3D=hip.means3D, shs=hip.shs, opacities=hip.opac, scales=hip.scales, rotations=hip.rots, refl_strengths=hip.refl)


_GRAD_NAMES = dict(means3D="dL_dmeans3D", shs="dL_dsh", opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations",
                   refl_strengths="dL_drefl_strengths")


def test_grad_sink_routes_gradients_into_caller_buffers():
    """Extension: with a gradient sink set on a rasterizer, the backward of ITS forward calls writes the parameter gradients
    into the caller's tensors (views of one flat buffer) and autograd leaves the leaves' .grad alone; values equal the
    plain autograd path.  A second rasterizer without a sink, run in between, is unaffected (no module-level state)."""
    from gsr_dist import FlatGrads
    kw, _, _ = scene_kwargs("S", 4000, 192, 128, 77, -2.8, 3, (0, 0, 0))
    g = S.make_upstream_grads(128, 192, 5)
    plain = HipSurfel(kw).backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    box = {}

    def make_sink(hip):
        box["fg"] = FlatGrads(_sink_params(hip))
        box["fg"].flat.fill_(float("nan"))          # every element must be overwritten by the kernels
        return box["fg"].sink(), False
    hip = HipSurfel(kw, make_sink=make_sink)
    other = HipSurfel(kw)                            # forward of another rasterizer while the first one's graph is alive
    fg, params = box["fg"], _sink_params(hip)
    before = {k: p.grad.data_ptr() for k, p in params.items()}
    again = other.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    assert torch.isnan(fg.flat).all()                # the un-sunk rasterizer did not touch the first one's sink
    hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    for k, p in params.items():
        assert p.grad.data_ptr() == before[k]                      # still the flat-buffer view
        got = fg.view(k).cpu().numpy()
        assert np.isfinite(got).all(), k
        ref = plain[_GRAD_NAMES[k]].reshape(got.shape)
        assert rel_maxnorm(got, ref) <= 5e-5, k                    # atomics order differs run to run
        assert rel_maxnorm(again[_GRAD_NAMES[k]].reshape(got.shape), ref) <= 5e-5, k


def test_grad_sink_accumulates_views_on_the_device():
    """accumulate=True: two views of the same scene add their parameter gradients into ONE zeroed buffer on the device
    (kernel `+=`), equal to the sum of the two single-view backwards; overwrite mode would keep only the second."""
    from gsr_dist import FlatGrads
    kwa, _, _ = scene_kwargs("S", 3000, 160, 128, 78, -2.7, 3, (0, 0, 0))
    kwb = dict(kwa)
    camb = S.look_at_camera(160, 128, eye=(0.6, -0.2, -0.5))
    for k in ("viewmatrix", "projmatrix", "campos"):
        kwb[k] = camb[k]
    g = S.make_upstream_grads(128, 160, 6)
    ga = HipSurfel(kwa).backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    gb = HipSurfel(kwb).backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    box = {}

    def make_sink(hip):
        if "fg" not in box:
            box["fg"] = FlatGrads(_sink_params(hip))
        return box["fg"].sink(), True
    ha = HipSurfel(kwa, make_sink=make_sink)
    hb = HipSurfel(kwb, make_sink=make_sink)
    ha.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    hb.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    for k in _GRAD_NAMES:
        got = box["fg"].view(k).cpu().numpy()
        ref = (ga[_GRAD_NAMES[k]] + gb[_GRAD_NAMES[k]]).reshape(got.shape)
        assert rel_maxnorm(got, ref) <= 5e-5, k
        assert rel_maxnorm(got, gb[_GRAD_NAMES[k]].reshape(got.shape)) > 1e-3, k     # really the sum, not the last view
    with pytest.raises(ValueError):
        HipSurfel(kwa, make_sink=lambda h: ({"means3D": box["fg"].view("means3D")}, True)).backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])


@pytest.mark.parametrize("P", [257, 1023])
def test_grad_sink_with_odd_gaussian_count_stays_16_byte_aligned(P):
    """P not a multiple of 4: the packed flat buffer pads every slice to 4 floats, so the float4 stores of dL_dsh / dL_drot into
    the sink views stay aligned (overwrite, then accumulate a second backward); a deliberately misaligned sink is refused
    loudly by the binding (ValueError) instead of reaching the kernel."""
    from gsr_dist import FlatGrads
    kw, _, _ = scene_kwargs("S", P, 160, 128, 79, -2.4, 3, (0, 0, 0))
    g = S.make_upstream_grads(128, 160, 7)
    plain = HipSurfel(kw).backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    box = {}

    def make_sink(acc):
        def f(hip):
            if "fg" not in box:
                box["fg"] = FlatGrads(_sink_params(hip))
                box["fg"].flat.fill_(float("nan"))
            return box["fg"].sink(), acc
        return f
    HipSurfel(kw, make_sink=make_sink(False)).backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    fg = box["fg"]
    for k in _GRAD_NAMES:
        assert fg.view(k).data_ptr() % 16 == 0, k
        got = fg.view(k).cpu().numpy()
        assert rel_maxnorm(got, plain[_GRAD_NAMES[k]].reshape(got.shape)) <= 5e-5, k
    HipSurfel(kw, make_sink=make_sink(True)).backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    for k in _GRAD_NAMES:
        got = fg.view(k).cpu().numpy()
        assert rel_maxnorm(got, 2.0 * plain[_GRAD_NAMES[k]].reshape(got.shape)) <= 5e-5, k
    # a packed buffer WITHOUT padding: shs starts at 3 P floats, misaligned for odd P
    raw = torch.zeros(59 * P + 8, device="cuda")

    def bad_sink(hip):
        views, off = {}, 0
        for k, p in _sink_params(hip).items():
            views[k] = raw[off:off + p.numel()].view(p.shape)
            off += p.numel()
        assert views["shs"].data_ptr() % 16 != 0
        return views, False
    with pytest.raises(ValueError, match="16-byte aligned"):
        HipSurfel(kw, make_sink=bad_sink).backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])


@pytest.mark.parametrize("variant", ["S", "G"])
@pytest.mark.parametrize("factor", [2500.0, 6000.0])
def test_far_scenes_keep_the_depth_order(variant, factor):
    """Depths of 10^4: the depth pre-sort orders the raw float bits (31 bits, four 8-bit digit places whose histogram is accumulated by
    gaussian_stats_kernel), so the top digit place is exercised with more than one value — the C1-C5 scenes live in [0.2, 8].  The whole scene
    scaled about the camera (same image, depths x factor) must give the oracle's point list, keys and ranges."""
    orc = _orc()
    kw, _, _ = scene_kwargs(variant, 3000, 192, 128, 55, -2.6, 2, (0.1, 0.1, 0.1))
    kw["means3D"] = (kw["means3D"] * factor).astype(np.float32)
    kw["scales"] = (kw["scales"] * factor).astype(np.float32)
    if variant == "S":
        o = orc.SurfelOracle(np.float32)
        ref = o.forward(**kw)
        hip = HipSurfel(kw)
    else:
        o = orc.GaussOracle(np.float32)
        ref = o.forward(**kw)
        hip = HipGauss(kw)
    d = o.state("depths")[o.state("radii") > 0]
    assert d.max() > 13107.0 and (d < 4096.0).any()        # (several values of the top 8-bit digit: float exponents 2^11 .. 2^14)
    assert hip.R == ref["num_rendered"] > 0
    np.testing.assert_array_equal(hip.state("point_list").astype(np.uint32), o.state("point_list"))
    np.testing.assert_array_equal(hip.state("keys").astype(np.uint64), o.state("keys"))
    np.testing.assert_array_equal(hip.state("ranges").astype(np.uint32), o.state("ranges"))
    assert psnr(hip.out()["color"], ref["color"]) >= 50


def test_pixels_without_contributors_report_zero_median():
    """A sparse scene: most pixels see no surfel at all.  Their median-contributor entry is the reference's float -1
    converted with saturation, i.e. 0 (the C++ conversion is undefined and once compiled to lane garbage)."""
    orc = _orc()
    kw, _, _ = scene_kwargs("S", 60, 256, 192, 123, -3.5, 1, (0.2, 0.2, 0.2))
    o = orc.SurfelOracle(np.float32)
    ref = o.forward(**kw)
    hip = HipSurfel(kw)
    nh, no_ = hip.state("n_contrib").astype(np.int64), o.state("n_contrib").astype(np.int64)
    assert (no_[0] == 0).mean() > 0.5
    np.testing.assert_array_equal(nh, no_)
    np.testing.assert_array_equal(hip.out()["color"][:, no_[0] == 0], ref["color"][:, no_[0] == 0])


def test_more_tiles_than_the_dispatch_order_keeps_in_registers():
    """17 600 tiles: tile_order_kernel keeps 16 list lengths per thread in registers (16 384 tiles) and re-reads the rest; every
    tile must still be dispatched exactly once (a missing or duplicated tile shows in the image and the contributor counts)."""
    orc = _orc()
    W, H = 2816, 1600
    kw, _, _ = scene_kwargs("S", 3000, W, H, 321, -2.6, 1, (0.1, 0.2, 0.3))
    o = orc.SurfelOracle(np.float32)
    ref = o.forward(**kw)
    hip = HipSurfel(kw)
    out = hip.out()
    assert out["num_rendered"] == ref["num_rendered"]
    np.testing.assert_array_equal(hip.state("n_contrib").astype(np.int64), o.state("n_contrib").astype(np.int64))
    assert psnr(out["color"], ref["color"]) >= 50


@pytest.mark.parametrize("P,W,H,mu", [(20_000, 320, 200, -3.2), (1_000_000, 1920, 1080, -4.75)])
def test_sort_drivers_agree_bit_for_bit(P, W, H, mu):
    """The one-clear Onesweep driver (csrc/gsr_sort.hpp, on rocPRIM's private device code) against the public
    rocprim::radix_sort_pairs (gsr_set_option("sort_driver", 0)) at a small size and at C3: identical depth order, point list,
    tile ranges, contributor counts and image — for both rasterizer variants' binning and for the reflection backward's
    texel-id sort (gradient of the cubemap)."""
    import _gsr
    from gaussian_renderer import deferred_reflection
    kw, cam, sc = scene_kwargs("S", P, W, H, 1003, mu, 3, (0, 0, 0))
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    tex0, fail0 = S.make_cubemap(64, 3, 5)
    res = []
    try:
        for driver in (1, 0):
            _gsr.set_option("sort_driver", driver)
            hip2 = HipSurfel(kw)
            out = hip2.out()
            state = {k: hip2.state(k) for k in ("point_list", "ranges", "n_contrib", "keys")}

            class Env:
                params = {"Cubemap_texture": torch.from_numpy(tex0).cuda().requires_grad_(True), "Cubemap_failv": torch.from_numpy(fail0).cuda()}
            final, _, _ = deferred_reflection(hip2.allmap[2:5].detach(), hip2.color.detach(), hip2.refl_map.detach(), Env, ct["viewmatrix"],
                                              (H, W, cam["K"]), ct["R"], ct["T"])
            final.sum().backward()
            res.append((out, state, Env.params["Cubemap_texture"].grad.cpu().numpy()))
    finally:
        _gsr.set_option("sort_driver", 1)
    (o1, s1, g1), (o0, s0, g0) = res
    assert o1["num_rendered"] == o0["num_rendered"] > 0
    for k in s1:
        np.testing.assert_array_equal(s1[k], s0[k], err_msg=k)
    for k in ("color", "allmap", "radii"):
        np.testing.assert_array_equal(o1[k], o0[k], err_msg=k)
    assert rel_maxnorm(g1, g0) <= 1e-5      # same sorted order; LDS / global float atomics inside the combine differ in order


def test_normal_view_tap_equals_the_slice():
    """Extension: GaussianRasterizer.set_output_taps(("normal_view",)) returns allmap[2:5] as a sixth output that aliases the
    allmap tensor; the gradient its consumer (the reflection pass) sends back reaches the tile backward as a pointer of its own
    (gsr_surfel_backward_ex) and is added to the upstream planes while they are loaded.  Same forward values, and the same
    parameter gradients as slicing allmap by hand — with plain autograd, with a gradient sink, and when only the tap carries a
    gradient (no upstream gradient for allmap at all)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings as GS_G, GaussianRasterizer as GR_G
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussian_renderer import deferred_reflection
    from gsr_dist import FlatGrads
    P, W, H, L = 4000, 192, 128, 16
    sc = S.make_scene(P, "S", seed=31, mu=-2.7)
    tex, fail = S.make_cubemap(L, 3, 31)
    cam = S.make_camera(W, H)
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cam.items() if isinstance(v, np.ndarray)}
    g = S.make_upstream_grads(H, W, 31)
    g_final, g_allmap = torch.from_numpy(g["dL_dcolor"]).cuda(), torch.from_numpy(g["dL_dplanes"]).cuda()
    names = ["means3D", "shs", "opacities", "scales", "rotations", "refl_strengths"]
    settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                             bg=torch.zeros(3, device="cuda"), scale_modifier=1.0, viewmatrix=ct["viewmatrix"],
                                             projmatrix=ct["projmatrix"], sh_degree=3, campos=ct["campos"], prefiltered=False, debug=False)

    class Env:
        params = {"Cubemap_texture": torch.from_numpy(tex).cuda().requires_grad_(True), "Cubemap_failv": torch.from_numpy(fail).cuda().requires_grad_(True)}

    def run(tap, with_allmap_grad=True, sink=False):
        p = {k: torch.from_numpy(sc[k]).cuda().requires_grad_(True) for k in names}
        rast = GaussianRasterizer(settings)
        fg = None
        if sink:
            fg = FlatGrads(p)
            fg.flat.fill_(float("nan"))
            rast.set_grad_sink(fg.sink())
        if tap:
            rast.set_output_taps(("normal_view",))
        means2D = torch.zeros(P, 3, device="cuda", requires_grad=True)
        out = rast(means3D=p["means3D"], means2D=means2D, opacities=p["opacities"], shs=p["shs"], refl_strengths=p["refl_strengths"],
                   scales=p["scales"], rotations=p["rotations"], env_scope_mask=torch.from_numpy(sc["env_scope_mask"]).cuda())
        assert len(out) == (6 if tap else 5)
        base, radii, allmap, refl_map, gw = out[:5]
        nv = out[5] if tap else allmap[2:5]
        if tap:
            assert nv.data_ptr() == allmap[2:5].data_ptr() and tuple(nv.shape) == (3, H, W)     # an alias, not a copy
        final, _, _ = deferred_reflection(nv, base, refl_map, Env, ct["viewmatrix"], (H, W, cam["K"]), ct["R"], ct["T"])
        if with_allmap_grad:
            torch.autograd.backward([final, allmap], [g_final, g_allmap])
        else:
            torch.autograd.backward([final], [g_final])
        grads = {k: (fg.view(k) if sink else p[k].grad).detach().cpu().numpy().copy() for k in names}
        grads["means2D"] = means2D.grad.cpu().numpy().copy()
        return final.detach().cpu().numpy(), grads

    for kw in (dict(), dict(with_allmap_grad=False), dict(sink=True)):
        f0, g0 = run(False, **kw)
        f1, g1 = run(True, **kw)
        assert np.array_equal(f0, f1)
        for k in g0:
            assert np.isfinite(g1[k]).all(), (kw, k)
            assert np.abs(g0[k]).max() > 0, (kw, k)
            assert rel_maxnorm(g1[k], g0[k]) <= 5e-5, (kw, k)          # float atomics: arrival order differs run to run
    with pytest.raises(NotImplementedError):
        GR_G(GS_G(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device="cuda"), scale_modifier=1.0,
                  viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"], sh_degree=3, campos=ct["campos"], prefiltered=False, debug=False,
                  antialiasing=False)).set_output_taps(("normal_view",))
    with pytest.raises(NotImplementedError):
        GaussianRasterizer(settings).set_output_taps(("depth",))


@pytest.mark.parametrize("variant", ["S", "G"])
def test_num_rendered_mailbox_equals_the_copy_path(variant):
    """num_rendered reaches the host through a pinned mailbox word written by the last workgroup of the statistics kernel
    (default) or through the round-2 copy + event (gsr_set_option("mailbox", 0)): same count, same outputs, call after call
    (the sequence number distinguishes consecutive forwards, also across different scenes)."""
    import _gsr
    outs = {}
    try:
        for mode in (1, 0, 1):
            _gsr.set_option("mailbox", mode)
            for seed, P in ((41, 3000), (42, 5000), (41, 3000)):
                kw, _, _ = scene_kwargs(variant, P, 200, 120, seed, -2.8, 3, (0, 0, 0))
                hip = HipSurfel(kw) if variant == "S" else HipGauss(kw, antialiasing=False)
                o = hip.out()
                key = (seed, P)
                if key in outs:
                    assert o["num_rendered"] == outs[key]["num_rendered"]
                    assert np.array_equal(o["color"], outs[key]["color"]) and np.array_equal(o["radii"], outs[key]["radii"])
                else:
                    outs[key] = o
                assert o["num_rendered"] == int(hip.state("tiles_touched").astype(np.int64).sum())
    finally:
        _gsr.set_option("mailbox", 1)
    assert outs[(41, 3000)]["num_rendered"] != outs[(42, 5000)]["num_rendered"]


def test_num_rendered_beyond_int32_is_refused():
    """40 000 splats that each cover (nearly) all 65 536 tiles of a 4096 x 4096 image: the instance count (~2.6e9) does not fit the int the
    API returns.  The reference's 32-bit InclusiveSum wraps silently (rasterizer_impl.cu:282) and then allocates and sorts garbage;
    here the count is reduced in 64 bits and the call is refused before the binning workspace is sized."""
    kw, _, _ = scene_kwargs("G", 40_000, 4096, 4096, 12, 0.0, 0, (0, 0, 0))
    with pytest.raises(RuntimeError, match="does not fit"):
        HipGauss(kw, requires_grad=False)
    # the library is usable afterwards
    kw2, _, _ = scene_kwargs("G", 2000, 160, 96, 13, -2.5, 0, (0, 0, 0))
    assert (HipGauss(kw2, requires_grad=False).out()["radii"] > 0).any()
