"""Variant G (diff_gaussian_rasterization) with the extensions variant S has had since round 2: gradient sinks and accumulate mode
(gsr_gauss_backward_accum; marshaling follows DGR rasterize_points.cu:142-264).  The kernels write — or add — the seven parameter
gradients straight into views of one flat buffer; values equal plain autograd and, for one view, the oracle."""
import numpy as np
import pytest
import torch

from helpers import HipGauss, S, grad_gate, rel_maxnorm, scene_kwargs

pytestmark = pytest.mark.gpu

_NAMES = dict(means3D="dL_dmeans3D", shs="dL_dsh", opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations",
              refl_strengths="dL_drefl_strengths", normals="dL_dnormals")


def _params(h):
    return dict(means3D=h.means3D, shs=h.shs, opacities=h.opac, scales=h.scales, rotations=h.rots, refl_strengths=h.refl, normals=h.normals)


def _bwd(h, g):
    return h.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])


@pytest.mark.parametrize("P,aa", [(3001, True), (257, False)])
def test_gauss_sink_overwrite_then_add_equals_the_sum_of_two_views(P, aa):
    """First view overwrites the (NaN-filled) flat buffer, the second adds to it on the device: equal to the sum of two single-view
    plain-autograd backwards; P is odd (the 16-byte guard of the float4 stores: FlatGrads pads its slices); view A against the oracle."""
    from gsr_dist import FlatGrads
    from oracle import oracle as orc
    W, H = 176, 112
    kwa, _, _ = scene_kwargs("G", P, W, H, 91, -2.7, 3, (0.1, 0.0, 0.2))
    kwb = dict(kwa)
    camb = S.look_at_camera(W, H, eye=(0.5, -0.2, -0.4))
    for k in ("viewmatrix", "projmatrix", "campos"):
        kwb[k] = camb[k]
    g = S.make_upstream_grads(H, W, 8)
    ga = _bwd(HipGauss(kwa, antialiasing=aa), g)
    gb = _bwd(HipGauss(kwb, antialiasing=aa), g)
    o = orc.GaussOracle(np.float32)
    o.forward(antialiasing=aa, **kwa)
    go = o.backward(dL_dcolor=g["dL_dcolor"], dL_dinvdepth=g["dL_dinvdepth"], dL_dnormal_map=g["dL_dnormal"], dL_drefl_strength_map=g["dL_drefl"])
    box = {}

    def sink(acc):
        def f(h):
            if "fg" not in box:
                box["fg"] = FlatGrads(_params(h))
                box["fg"].flat.fill_(float("nan"))
            return box["fg"].sink(), acc
        return f
    ha = HipGauss(kwa, antialiasing=aa, make_sink=sink(False))
    _bwd(ha, g)
    fg = box["fg"]
    for k, name in _NAMES.items():
        got = fg.view(k).cpu().numpy()
        assert fg.view(k).data_ptr() % 16 == 0, k
        assert np.isfinite(got).all(), k                                  # every element was written
        assert rel_maxnorm(got, ga[name].reshape(got.shape)) <= 5e-5, k   # = plain autograd (atomics order differs)
        ref = go[name].reshape(got.shape)
        assert rel_maxnorm(got, ref) <= 1e-4, k                           # = the oracle
        assert grad_gate(got, ref, floor=1e-5) <= 1e-3, k                   # elementwise (small scene: no budget of elements to speak of)
        assert _params(ha)[k].grad is None or _params(ha)[k].grad.data_ptr() == fg.view(k).data_ptr()
    _bwd(HipGauss(kwb, antialiasing=aa, make_sink=sink(True)), g)
    for k, name in _NAMES.items():
        got = fg.view(k).cpu().numpy()
        ref = (ga[name] + gb[name]).reshape(got.shape)
        assert rel_maxnorm(got, ref) <= 5e-5, k
        assert rel_maxnorm(got, gb[name].reshape(got.shape)) > 1e-3, k     # really the sum, not the last view


def test_gauss_sink_argument_checks():
    from gsr_dist import FlatGrads
    kw, _, _ = scene_kwargs("G", 500, 96, 64, 92, -2.5, 3, (0, 0, 0))
    g = S.make_upstream_grads(64, 96, 9)
    box = {}

    def partial(h):
        box["fg"] = FlatGrads(_params(h))
        return {"means3D": box["fg"].view("means3D")}, True      # accumulate needs a sink for every parameter gradient
    with pytest.raises(ValueError, match="accumulate"):
        _bwd(HipGauss(kw, make_sink=partial), g)
    with pytest.raises(ValueError, match="unknown gradient"):
        _bwd(HipGauss(kw, make_sink=lambda h: ({"bogus": torch.zeros(3, device="cuda")}, False)), g)
    raw = torch.zeros(500 * 64 + 16, device="cuda")

    def misaligned(h):
        return {"rotations": raw[1:1 + 2000].view(500, 4)}, False
    with pytest.raises(ValueError, match="16-byte aligned"):
        _bwd(HipGauss(kw, make_sink=misaligned), g)
