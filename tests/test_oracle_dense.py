"""The oracle against the dense float64 torch-autograd restatement (tests/dense_torch.py): a second pin that shares no
code and no derivation with oracle/*.cpp — forward planes compared value by value, every TRUE gradient supplied by torch
autograd.  Outputs of the reference that are not true gradients are left out here exactly as in
test_oracle_fd.py and pinned by known answers in test_oracle_quirks.py.  CPU only, float64, 48 primitives, 32 x 32."""
import numpy as np
import pytest
import torch

import dense_torch as dn
from oracle import oracle as orc

W = H = 32
TAN = 0.5     # focal = 32 exactly, so the surfel backward's int(focal * tan * 2) == W quirk stays inert


def _camera(seed):
    """A general camera (rotation + translation), row-vector convention as scene/cameras.py builds it."""
    rs = np.random.RandomState(seed)
    ang = rs.uniform(-0.15, 0.15, 3)
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Rwc = Rz @ Ry @ Rx                       # world -> camera
    t = np.array([0.05, -0.03, 0.1])
    view = np.eye(4)
    view[:3, :3] = Rwc.T
    view[3, :3] = t
    znear, zfar = 0.01, 100.0
    Pm = np.zeros((4, 4))
    Pm[0, 0] = 1.0 / TAN
    Pm[1, 1] = 1.0 / TAN
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    full = view @ Pm.T
    campos = np.linalg.inv(view)[3, :3]
    return view, full, campos


def _scene(variant, P, seed, log_scale=-1.7):
    rs = np.random.RandomState(seed)
    means = np.stack([rs.uniform(-1.3, 1.3, P), rs.uniform(-1.3, 1.3, P), rs.uniform(2.5, 5.5, P)], 1)
    means[:3, 2] = rs.uniform(-0.5, 0.15, 3)          # behind / at the near plane: culled
    ns = 3 if variant == "G" else 2
    scales = np.exp(rs.normal(log_scale, 0.4, (P, ns)))
    rot = rs.normal(size=(P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opac = np.clip(1 / (1 + np.exp(-rs.normal(0.5, 1.2, (P, 1)))), 0.02, 0.97)
    opac[3] = 0.002                                   # never reaches 1/255
    shs = np.concatenate([rs.normal(size=(P, 1, 3)), 0.2 * rs.normal(size=(P, 15, 3))], 1)
    refl = 1 / (1 + np.exp(-rs.normal(-1.0, 1.0, (P, 1))))
    normals = rs.normal(size=(P, 3))
    mask = rs.uniform(size=P) < 0.6
    return dict(means3D=means, scales=scales, rotations=rot, opacities=opac, shs=shs, refl_strengths=refl, normals=normals, mask=mask)


def _t(x, grad=True):
    return torch.tensor(np.asarray(x, dtype=np.float64), dtype=torch.float64, requires_grad=grad)


def _assert_close(a, b, rtol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max()
    assert err <= rtol * scale, f"{what}: max abs diff {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("seed,antialiasing", [(0, False), (1, False), (2, True)])
def test_gauss_forward_and_true_gradients(seed, antialiasing):
    P = 48
    p = _scene("G", P, seed)
    view, full, campos = _camera(seed)
    bg = np.array([0.2, 0.5, 0.3])
    o = orc.GaussOracle(np.float64)
    ref = o.forward(bg=bg, means3D=p["means3D"], opacities=p["opacities"], viewmatrix=view, projmatrix=full, campos=campos, tanfovx=TAN,
                    tanfovy=TAN, image_height=H, image_width=W, sh_degree=3, shs=p["shs"], normals=p["normals"],
                    refl_strengths=p["refl_strengths"], scales=p["scales"], rotations=p["rotations"], antialiasing=antialiasing)
    assert ref["num_rendered"] > 0
    t = {k: _t(p[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs", "normals", "refl_strengths")}
    out = dn.render_gauss(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], t["normals"], t["refl_strengths"],
                          _t(view, False), _t(full, False), _t(campos, False), TAN, TAN, W, H, _t(bg, False), antialiasing=antialiasing)
    np.testing.assert_array_equal(out["radii"].numpy(), ref["radii"])
    for k in ("color", "normal_map", "refl_strength_map", "invdepth"):
        _assert_close(out[k].detach().numpy(), ref[k], 1e-10, k)
    _assert_close(out["final_T"].detach().numpy(), o.state("final_T"), 1e-12, "final_T")
    rs = np.random.RandomState(100 + seed)
    wc, wn, wr, wi = rs.normal(size=(3, H, W)), rs.normal(size=(3, H, W)), rs.normal(size=(1, H, W)), rs.normal(size=(1, H, W))
    loss = (out["color"] * _t(wc, False)).sum() + (out["normal_map"] * _t(wn, False)).sum() + (out["refl_strength_map"] * _t(wr, False)).sum() + \
        (out["invdepth"] * _t(wi, False)).sum()
    loss.backward()
    g = o.backward(dL_dcolor=wc, dL_dinvdepth=wi, dL_dnormal_map=wn, dL_drefl_strength_map=wr)
    names = dict(opacities="dL_dopacity", refl_strengths="dL_drefl_strengths", normals="dL_dnormals", shs="dL_dsh")
    if not antialiasing:
        # with anti-aliasing the covariance path of the reference is not a true gradient (closed form evaluated with the
        # post-blur entries, DGR backward.cu:235-245): only the quantities that do not pass through it are compared then
        names.update(means3D="dL_dmeans3D", scales="dL_dscales", rotations="dL_drotations")
    for k, gk in names.items():
        # the conic -> covariance step of the reference divides by (det^2 + 1e-7) instead of det^2 (DGR backward.cu:256-257,
        # "denom2inv"): a deliberate regulariser that moves the three covariance-path gradients by ~1e-7 / det^2 relative
        tol = 2e-6 if k in ("means3D", "scales", "rotations") else 1e-8
        _assert_close(g[gk].reshape(p[k].shape), t[k].grad.numpy(), tol, gk)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_surfel_forward_and_true_gradients(seed):
    P = 48
    p = _scene("S", P, seed)
    view, full, campos = _camera(seed)
    bg = np.array([0.2, 0.5, 0.3])
    o = orc.SurfelOracle(np.float64)
    ref = o.forward(bg=bg, means3D=p["means3D"], opacities=p["opacities"], viewmatrix=view, projmatrix=full, campos=campos, tanfovx=TAN,
                    tanfovy=TAN, image_height=H, image_width=W, sh_degree=3, shs=p["shs"], refl_strengths=p["refl_strengths"],
                    scales=p["scales"], rotations=p["rotations"], env_scope_mask=p["mask"])
    assert ref["num_rendered"] > 0
    t = {k: _t(p[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs", "refl_strengths")}
    out = dn.render_surfel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], t["refl_strengths"],
                           torch.from_numpy(p["mask"].astype(np.float64)), _t(view, False), _t(full, False), _t(campos, False), TAN, TAN, W, H,
                           _t(bg, False), freeze_lowpass_depth=True)
    np.testing.assert_array_equal(out["radii"].numpy(), ref["radii"])
    _assert_close(out["color"].detach().numpy(), ref["color"], 1e-10, "color")
    _assert_close(out["refl_strength_map"].detach().numpy(), ref["refl_strength_map"], 1e-10, "refl_strength_map")
    for plane, name in enumerate(("depth", "alpha", "normal.x", "normal.y", "normal.z", "median depth", "distortion", "env-scope mask")):
        _assert_close(out["allmap"][plane].detach().numpy(), ref["allmap"][plane], 1e-9, name)
    _assert_close(out["gaussian_weights"].numpy(), ref["gaussian_weights"], 1e-10, "gaussian_weights")
    rs = np.random.RandomState(200 + seed)
    wc, wr, wa = rs.normal(size=(3, H, W)), rs.normal(size=(1, H, W)), rs.normal(size=(8, H, W))
    # (plane 5, the median depth, is the depth of ONE contributor per pixel: its gradient goes to that contributor alone)
    wa[7] = 0.0          # mask plane carries no gradient
    wa[6] *= 1e3         # the distortion plane is ~1e-3 of the others
    loss = (out["color"] * _t(wc, False)).sum() + (out["allmap"] * _t(wa, False)).sum() + (out["refl_strength_map"] * _t(wr, False)).sum()
    loss.backward()
    g = o.backward(dL_dcolor=wc, dL_dallmap=wa, dL_drefl_strength_map=wr)
    for k, gk in dict(means3D="dL_dmeans3D", scales="dL_dscales", opacities="dL_dopacity", refl_strengths="dL_drefl_strengths", shs="dL_dsh").items():
        _assert_close(g[gk].reshape(p[k].shape), t[k].grad.numpy(), 1e-7, gk)
    # the reference returns the quaternion gradient without the normalisation Jacobian (DSR auxiliary.h:242-286); for unit
    # input the true gradient (autograd, through the normalisation) is its projection onto the tangent space
    q, ga = p["rotations"], g["dL_drotations"]
    _assert_close(ga - q * (q * ga).sum(axis=1, keepdims=True), t["rotations"].grad.numpy(), 1e-7, "dL_drotations (tangential)")


def test_surfel_depth_gradient_is_a_true_gradient_on_the_ray_splat_branch():
    """Large face-on surfels: the low-pass falloff never wins, so no part of the reference's depth / distortion gradient is
    frozen and plain autograd (freeze_lowpass_depth=False) must agree on every plane."""
    P = 24
    p = _scene("S", P, 5, log_scale=-0.6)
    q = np.tile(np.array([[1.0, 0.05, -0.04, 0.02]]), (P, 1))
    p["rotations"] = q / np.linalg.norm(q, axis=1, keepdims=True)
    view, full, campos = _camera(5)
    bg = np.zeros(3)
    o = orc.SurfelOracle(np.float64)
    o.forward(bg=bg, means3D=p["means3D"], opacities=p["opacities"], viewmatrix=view, projmatrix=full, campos=campos, tanfovx=TAN, tanfovy=TAN,
              image_height=H, image_width=W, sh_degree=3, shs=p["shs"], refl_strengths=p["refl_strengths"], scales=p["scales"],
              rotations=p["rotations"], env_scope_mask=p["mask"])
    t = {k: _t(p[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs", "refl_strengths")}
    out = dn.render_surfel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], t["refl_strengths"],
                           torch.from_numpy(p["mask"].astype(np.float64)), _t(view, False), _t(full, False), _t(campos, False), TAN, TAN, W, H,
                           _t(bg, False), freeze_lowpass_depth=False)
    rs = np.random.RandomState(9)
    wa = np.zeros((8, H, W))
    wa[0], wa[6] = rs.normal(size=(H, W)), 1e3 * rs.normal(size=(H, W))
    (out["allmap"] * _t(wa, False)).sum().backward()
    g = o.backward(dL_dcolor=np.zeros((3, H, W)), dL_dallmap=wa, dL_drefl_strength_map=np.zeros((1, H, W)))
    for k, gk in dict(means3D="dL_dmeans3D", scales="dL_dscales", opacities="dL_dopacity").items():
        _assert_close(g[gk].reshape(p[k].shape), t[k].grad.numpy(), 1e-7, gk)
