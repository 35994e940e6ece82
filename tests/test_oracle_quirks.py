"""Known answers for the outputs of the reference that are NOT true gradients, each
derived by hand from the text of the quirk and evaluated with the dense float64 restatement (tests/dense_torch.py), then
compared with what the oracle returns.  Together with test_oracle_dense.py this pins every backward output of the oracle
against something that does not share its derivation.  CPU only."""
import numpy as np
import torch

import dense_torch as dn
from oracle import oracle as orc
from test_oracle_dense import H, TAN, W, _assert_close, _camera, _scene, _t


def _gauss_oracle(p, view, full, campos, bg):
    o = orc.GaussOracle(np.float64)
    o.forward(bg=bg, means3D=p["means3D"], opacities=p["opacities"], viewmatrix=view, projmatrix=full, campos=campos, tanfovx=TAN, tanfovy=TAN,
              image_height=H, image_width=W, sh_degree=3, shs=p["shs"], normals=p["normals"], refl_strengths=p["refl_strengths"],
              scales=p["scales"], rotations=p["rotations"])
    return o


def test_gauss_mean2d_pixels_is_the_3_2_1_weighted_colour_gradient():
    """DGR backward.cu:603-655: `dL_dalpha_means2d` is accumulated INSIDE the colour-channel loop as a running sum of the
    partial dL_dalpha, so channel c enters with weight 3 - c; normal / reflection / inverse-depth channels do not enter at
    all; the background term enters unweighted; and the result is scaled to NDC units (x 0.5 W, 0.5 H).  Hence
        grad_means2D = (0.5 W, 0.5 H) * d/d(xy) [ sum_c (3 - c) <g_c, C_c without background> + sum_c <g_c, bg_c T_final> ]
    which autograd evaluates on the dense model through a leaf added to the screen-space means."""
    P = 24
    p = _scene("G", P, 7)
    view, full, campos = _camera(7)
    bg = np.array([0.3, 0.1, 0.6])
    o = _gauss_oracle(p, view, full, campos, bg)
    rs = np.random.RandomState(70)
    wc, wn, wr, wi = rs.normal(size=(3, H, W)), rs.normal(size=(3, H, W)), rs.normal(size=(1, H, W)), rs.normal(size=(1, H, W))
    g = o.backward(dL_dcolor=wc, dL_dinvdepth=wi, dL_dnormal_map=wn, dL_drefl_strength_map=wr)
    off = torch.zeros(P, 2, dtype=torch.float64, requires_grad=True)
    t = {k: _t(p[k], False) for k in ("means3D", "scales", "rotations", "opacities", "shs", "normals", "refl_strengths")}
    out = dn.render_gauss(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], t["normals"], t["refl_strengths"], _t(view, False),
                          _t(full, False), _t(campos, False), TAN, TAN, W, H, _t(bg, False), xy_offset=off)
    k = torch.tensor([3.0, 2.0, 1.0], dtype=torch.float64)[:, None, None]
    quirk_loss = (k * _t(wc, False) * out["color_nobg"]).sum() + (_t(wc, False) * _t(bg, False)[:, None, None] * out["final_T"][None]).sum()
    quirk_loss.backward()
    expect = off.grad.numpy() * np.array([0.5 * W, 0.5 * H])
    got = g["dL_dmeans2D"]
    assert np.abs(expect).max() > 0
    _assert_close(got[:, :2], expect, 1e-8, "dL_dmean2D_pixels")
    assert np.abs(got[:, 2]).max() == 0
    # ... and it is NOT the true screen-space gradient of the loss the other outputs were differentiated for
    off2 = torch.zeros(P, 2, dtype=torch.float64, requires_grad=True)
    out2 = dn.render_gauss(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], t["normals"], t["refl_strengths"], _t(view, False),
                           _t(full, False), _t(campos, False), TAN, TAN, W, H, _t(bg, False), xy_offset=off2)
    ((out2["color"] * _t(wc, False)).sum() + (out2["normal_map"] * _t(wn, False)).sum() + (out2["refl_strength_map"] * _t(wr, False)).sum() +
     (out2["invdepth"] * _t(wi, False)).sum()).backward()
    true_ndc = off2.grad.numpy() * np.array([0.5 * W, 0.5 * H])
    assert np.abs(true_ndc - got[:, :2]).max() > 1e-3 * np.abs(true_ndc).max()
    # the internal gradient that feeds dL_dmeans3D (Appendix A.2: "dL_dmean2D, full dL_dalpha") IS that true gradient
    _assert_close(g["dL_dmeans2D_internal"][:, :2], true_ndc, 1e-8, "dL_dmean2D (internal)")


def _surfel_pair(seed, log_scale):
    P = 24
    p = _scene("S", P, seed, log_scale=log_scale)
    view, full, campos = _camera(seed)
    bg = np.array([0.2, 0.5, 0.3])
    o = orc.SurfelOracle(np.float64)
    o.forward(bg=bg, means3D=p["means3D"], opacities=p["opacities"], viewmatrix=view, projmatrix=full, campos=campos, tanfovx=TAN, tanfovy=TAN,
              image_height=H, image_width=W, sh_degree=3, shs=p["shs"], refl_strengths=p["refl_strengths"], scales=p["scales"],
              rotations=p["rotations"], env_scope_mask=p["mask"])
    return p, view, full, campos, bg, o


def test_surfel_mean2d_is_overwritten_with_the_densification_signal():
    """DSR backward.cu:656-659: whatever the tile pass accumulated in dL_dmean2D is replaced by
        dL_dmean2D.x = dL_dtransMat[2] * transMat[8] * 0.5 * W,   .y = dL_dtransMat[5] * transMat[8] * 0.5 * H
    where dL_dtransMat is the gradient the TILE PASS accumulated (the per-pair use of T; the bounding-box-centre term is
    folded in later) and transMat[2], [5], [8] are the constant-term coefficients of x w, y w and w.  In the dense model
    T[:, i, j] is the coefficient of local coordinate i in output j, so these are T[:, 2, 0], T[:, 2, 1], T[:, 2, 2] and the
    pair-path gradient is d loss / d T with the centre detached."""
    p, view, full, campos, bg, o = _surfel_pair(11, -1.7)
    rs = np.random.RandomState(71)
    wc, wr, wa = rs.normal(size=(3, H, W)), rs.normal(size=(1, H, W)), rs.normal(size=(8, H, W))
    wa[7] = 0.0
    g = o.backward(dL_dcolor=wc, dL_dallmap=wa, dL_drefl_strength_map=wr)
    t = {k: _t(p[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs", "refl_strengths")}
    out = dn.render_surfel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], t["refl_strengths"],
                           torch.from_numpy(p["mask"].astype(np.float64)), _t(view, False), _t(full, False), _t(campos, False), TAN, TAN, W, H,
                           _t(bg, False), freeze_lowpass_depth=True, pair_path_only_T=True)
    ((out["color"] * _t(wc, False)).sum() + (out["allmap"] * _t(wa, False)).sum() + (out["refl_strength_map"] * _t(wr, False)).sum()).backward()
    Tm, dT = out["Tm"].detach().numpy(), out["Tm"].grad.numpy()
    visible = out["radii"].numpy() > 0
    expect = np.stack([dT[:, 2, 0] * Tm[:, 2, 2] * 0.5 * W, dT[:, 2, 1] * Tm[:, 2, 2] * 0.5 * H], axis=1) * visible[:, None]
    assert np.abs(expect).max() > 0
    _assert_close(g["dL_dmeans2D"][:, :2], expect, 1e-7, "dL_dmean2D (densification signal)")
    assert np.abs(g["dL_dmeans2D"][:, 2]).max() == 0


def test_surfel_lowpass_branch_freezes_the_intersection_point():
    """DSR backward.cu:454-464: where the low-pass falloff wins (rho2d < rho3d) the depth gradient is propagated with the
    ray-splat intersection point s held constant.  With small surfels (many such pairs) the oracle therefore agrees with the
    dense model that freezes s there (checked in test_oracle_dense.py) and DISAGREES with plain autograd."""
    p, view, full, campos, bg, o = _surfel_pair(12, -2.6)
    rs = np.random.RandomState(72)
    wa = np.zeros((8, H, W))
    wa[0] = rs.normal(size=(H, W))
    g = o.backward(dL_dcolor=np.zeros((3, H, W)), dL_dallmap=wa, dL_drefl_strength_map=np.zeros((1, H, W)))
    res = {}
    for freeze in (True, False):
        t = {k: _t(p[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs", "refl_strengths")}
        out = dn.render_surfel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], t["refl_strengths"],
                               torch.from_numpy(p["mask"].astype(np.float64)), _t(view, False), _t(full, False), _t(campos, False), TAN, TAN, W, H,
                               _t(bg, False), freeze_lowpass_depth=freeze)
        assert out["lowpass_pairs"] > 50
        (out["allmap"] * _t(wa, False)).sum().backward()
        res[freeze] = t["scales"].grad.numpy()
    _assert_close(g["dL_dscales"], res[True], 1e-7, "dL_dscales (frozen s)")
    assert np.abs(g["dL_dscales"] - res[False]).max() > 1e-3 * np.abs(res[False]).max()
