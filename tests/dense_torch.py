"""Dense float64 torch-autograd restatement of both rasterizer variants (test infrastructure only).

A SECOND, independent pin of the oracle: written from the maths of SURVEY.md Appendix A — not
from oracle/*.cpp and not kernel by kernel — as plain differentiable torch code over dense [P, H, W] tensors, for small
scenes (<= 64 primitives, 32 x 32 pixels).  Forward outputs are compared with the oracle plane by plane and torch autograd
supplies every TRUE gradient (tests/test_oracle_dense.py); the reference's non-gradient outputs are
pinned separately by hand-derived known answers (tests/test_oracle_quirks.py).

Integer / discrete decisions (cull, radius, tile rectangle, skip and stop rules, blend order) are taken on detached values
and enter the differentiable part as masks, exactly as they are piecewise constant in the reference.
"""
import numpy as np
import torch


def F(v):
    """A constant the reference writes as a float literal (0.3f, 1.3f, the SH coefficients, ...): its VALUE is the fp32
    rounding of the decimal, also when the surrounding arithmetic runs in double."""
    return float(np.float32(v))


NEAR, FAR = F(0.2), 100.0
SH_C0 = F(0.28209479177387814)
SH_C1 = F(0.4886025119029199)
SH_C2 = tuple(F(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396))
SH_C3 = tuple(F(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
                             1.445305721320277, -0.5900435899266435))
ALPHA_MAX, ALPHA_MIN, T_STOP = F(0.99), F(1.0) / F(255.0), F(0.0001)


def sh_basis(d, degree):
    """Real spherical-harmonics basis of unit directions d [P,3] up to `degree` (<= 3) -> [P, (degree+1)^2]."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, SH_C0)]
    if degree > 0:
        b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if degree > 2:
        b += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy), SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)


def sh_colour(means, campos, shs, degree):
    """max(0, SH(dir) + 0.5) with dir = normalize(mean - campos); shs [P,16,3]."""
    d = means - campos
    d = d / d.norm(dim=1, keepdim=True)
    n = (degree + 1) ** 2
    rgb = torch.einsum("pk,pkc->pc", sh_basis(d, degree), shs[:, :n]) + 0.5
    return torch.clamp(rgb, min=0.0)


def quat_to_R(q):
    """Rotation matrix of quaternion rows (r, x, y, z), as given (no normalisation)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], dim=1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], dim=1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)], dim=1)


def tile_rect_mask(xy, radius, W, H):
    """[P,H,W] bool: pixel lies in a 16x16 tile of the primitive's rectangle (centre +- integer radius, clamped to the grid)."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r = radius.to(xy.dtype)
    lo_x = torch.clamp(((xy[:, 0] - r) / 16).to(torch.int64), 0, gx)
    hi_x = torch.clamp(((xy[:, 0] + r + 15) / 16).to(torch.int64), 0, gx)
    lo_y = torch.clamp(((xy[:, 1] - r) / 16).to(torch.int64), 0, gy)
    hi_y = torch.clamp(((xy[:, 1] + r + 15) / 16).to(torch.int64), 0, gy)
    tx = (torch.arange(W) // 16)[None, None, :]
    ty = (torch.arange(H) // 16)[None, :, None]
    inside = (tx >= lo_x[:, None, None]) & (tx < hi_x[:, None, None]) & (ty >= lo_y[:, None, None]) & (ty < hi_y[:, None, None])
    nonempty = ((hi_x - lo_x) * (hi_y - lo_y)) > 0
    return inside & nonempty[:, None, None], nonempty


def composite(alpha, usable, order):
    """Front-to-back blending rule shared by both variants.  alpha [P,H,W] (already min(0.99, .)), usable [P,H,W] bool (pair
    passes its skip tests), order = primitive indices front to back.  Returns blend weights w [P,H,W] (alpha * T before the
    pair; 0 where the pair does not blend), the final transmittance [H,W] and T before each pair [P,H,W]."""
    P, H, W = alpha.shape
    T = torch.ones(H, W, dtype=alpha.dtype)
    done = torch.zeros(H, W, dtype=torch.bool)
    w = [None] * P
    T_before = [None] * P
    for i in order.tolist():
        a = alpha[i]
        test = T * (1 - a)
        live = usable[i] & ~done
        stop = live & (test.detach() < T_STOP)         # the pixel terminates; this pair is NOT blended
        ok = live & ~stop
        done = done | stop
        T_before[i] = T
        w[i] = torch.where(ok, a * T, torch.zeros_like(T))
        T = torch.where(ok, test, T)
    return torch.stack(w), T, torch.stack(T_before)


def pixel_grid(W, H, dtype):
    px = torch.arange(W, dtype=dtype)[None, None, :]
    py = torch.arange(H, dtype=dtype)[None, :, None]
    return px, py


def render_gauss(means, scales, rots, opac, shs, normals, refl, view, proj, campos, tanfovx, tanfovy, W, H, bg, degree=3, scale_modifier=1.0,
                 antialiasing=False, xy_offset=None):
    """Variant G (Appendix A.1): EWA-projected 3D Gaussians.  Returns dict(color [3,H,W], normal_map [3,H,W],
    refl_strength_map [1,H,W], invdepth [1,H,W], radii [P], n_contrib [H,W])."""
    dt = means.dtype
    P = means.shape[0]
    A = view[:3, :3].T                                    # world -> view rotation for column vectors
    p_view = means @ view[:3, :3] + view[3, :3]
    hom = torch.cat([means, torch.ones(P, 1, dtype=dt)], dim=1) @ proj
    ndc = hom[:, :3] / (hom[:, 3:4] + F(0.0000001))
    tz = p_view[:, 2]
    visible = tz.detach() > NEAR
    # 3D covariance from the quaternion as given and the modified scales
    R = quat_to_R(rots)
    S = scale_modifier * scales
    Sigma = R @ torch.diag_embed(S * S) @ R.transpose(1, 2)
    # projection Jacobian at the (clamped) view-space position
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    limx, limy = F(1.3) * tanfovx, F(1.3) * tanfovy
    tx = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -fx * tx / (tz * tz)], dim=1), torch.stack([zero, fy / tz, -fy * ty / (tz * tz)], dim=1)], dim=1)
    M = J @ A[None]
    cov = M @ Sigma @ M.transpose(1, 2)
    a0, b0, c0 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    det0 = a0 * c0 - b0 * b0
    a, c = a0 + F(0.3), c0 + F(0.3)
    det = a * c - b0 * b0
    visible = visible & (det.detach() != 0)
    o = opac[:, 0]
    if antialiasing:
        o = o * torch.sqrt(torch.clamp(det0 / det, min=F(0.000025)))
    conic_a, conic_b, conic_c = c / det, -b0 / det, a / det
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=F(0.1)))
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach()))
    xy = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    if xy_offset is not None:        # a leaf added to the screen-space means (pixel units): d loss / d xy_offset = screen-space gradient
        xy = xy + xy_offset
    in_rect, nonempty = tile_rect_mask(xy.detach(), radius, W, H)
    visible = visible & nonempty
    rgb = sh_colour(means, campos, shs, degree)
    # pairs
    px, py = pixel_grid(W, H, dt)
    dx = xy[:, 0, None, None] - px
    dy = xy[:, 1, None, None] - py
    power = -0.5 * (conic_a[:, None, None] * dx * dx + conic_c[:, None, None] * dy * dy) - conic_b[:, None, None] * dx * dy
    alpha = torch.clamp(o[:, None, None] * torch.exp(power), max=ALPHA_MAX)
    usable = in_rect & visible[:, None, None] & ~(power.detach() > 0) & ~(alpha.detach() < ALPHA_MIN)
    order = torch.argsort(tz.detach(), stable=True)
    w, T_final, _ = composite(alpha, usable, order)
    color = torch.einsum("phw,pc->chw", w, rgb) + T_final[None] * bg[:, None, None]
    normal_map = torch.einsum("phw,pc->chw", w, normals)
    refl_map = torch.einsum("phw,p->hw", w, refl[:, 0])[None]
    invdepth = torch.einsum("phw,p->hw", w, 1.0 / tz)[None]
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    colour_only = torch.einsum("phw,pc->chw", w, rgb)     # without the background term (used by the mean2D quirk test)
    return dict(color=color, normal_map=normal_map, refl_strength_map=refl_map, invdepth=invdepth, radii=radii, final_T=T_final,
                color_nobg=colour_only)


def render_surfel(means, scales, rots, opac, shs, refl, mask, view, proj, campos, tanfovx, tanfovy, W, H, bg, degree=3, scale_modifier=1.0,
                  freeze_lowpass_depth=False, pair_path_only_T=False):
    """Variant S (Appendix A.3): 2D Gaussian surfels intersected per ray.  Returns dict(color [3,H,W], allmap [8,H,W],
    refl_strength_map [1,H,W], radii [P], gaussian_weights [P]).

    freeze_lowpass_depth: where the low-pass falloff wins (rho2d < rho3d) the reference's backward differentiates the depth
    with the ray-splat intersection point held constant (Appendix A.4); True restates that by detaching s there, so that
    autograd reproduces the reference's depth / distortion gradients also on that branch."""
    dt = means.dtype
    P = means.shape[0]
    p_view = means @ view[:3, :3] + view[3, :3]
    tz = p_view[:, 2]
    visible = tz.detach() > NEAR
    q = rots / rots.norm(dim=1, keepdim=True)             # normalised inside
    R = quat_to_R(q)
    L0 = R[:, :, 0] * (scale_modifier * scales[:, 0:1])   # tangent axes scaled, world space
    L1 = R[:, :, 1] * (scale_modifier * scales[:, 1:2])
    # homography rows: (u, v, 1) -> (x w, y w, w) in pixels;  splat2world = [L0|0; L1|0; p|1] (3 x 4), then PV, then NDC -> pixel
    s2w = torch.stack([torch.cat([L0, torch.zeros(P, 1, dtype=dt)], 1), torch.cat([L1, torch.zeros(P, 1, dtype=dt)], 1),
                       torch.cat([means, torch.ones(P, 1, dtype=dt)], 1)], dim=1)                       # [P,3,4]
    n2p = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, 0, 1]], dtype=dt).T   # [4,3]
    Tm = s2w @ proj @ n2p                                  # [P,3,3]: Tm[:, i, j] = coefficient of local coordinate i in output j
    Tm.retain_grad() if Tm.requires_grad else None
    Tu, Tv, Tw = Tm[:, :, 0], Tm[:, :, 1], Tm[:, :, 2]
    normal = R[:, :, 2] @ view[:3, :3]                     # view-space normal
    cosv = -(p_view * normal).sum(dim=1)
    visible = visible & (cosv.detach() != 0)
    normal = normal * torch.where(cosv.detach() > 0, 1.0, -1.0)[:, None]
    # bounding box of the 3-sigma ellipse
    t = torch.tensor([9.0, 9.0, -1.0], dtype=dt)
    dd = (t * Tw * Tw).sum(dim=1)
    visible = visible & (dd.detach() != 0)
    f = t[None] / dd[:, None]
    cx = (f * Tu * Tw).sum(dim=1)
    cy = (f * Tv * Tw).sum(dim=1)
    hx = torch.sqrt(torch.clamp(cx * cx - (f * Tu * Tu).sum(dim=1), min=F(1e-4)))
    hy = torch.sqrt(torch.clamp(cy * cy - (f * Tv * Tv).sum(dim=1), min=F(1e-4)))
    radius = torch.ceil(torch.maximum(torch.maximum(hx, hy), torch.full_like(hx, 3.0 * F(0.707106))).detach())
    xy = torch.stack([cx, cy], dim=1)
    if pair_path_only_T:            # gradient w.r.t. Tm then counts the per-pair use of T only, not the bounding-box centre
        xy = xy.detach()
    in_rect, nonempty = tile_rect_mask(xy.detach(), radius, W, H)
    visible = visible & nonempty
    rgb = sh_colour(means, campos, shs, degree)
    # pairs: planes through the pixel ray, their intersection line meets the splat plane at s
    px, py = pixel_grid(W, H, dt)
    k = px[..., None] * Tw[:, None, None, :] - Tu[:, None, None, :]
    l = py[..., None] * Tw[:, None, None, :] - Tv[:, None, None, :]
    qv = torch.cross(k, l, dim=-1)
    unstable = qv[..., 2].detach().abs() < F(1e-4)
    qz = torch.where(unstable, torch.ones_like(qv[..., 2]), qv[..., 2])
    sx = torch.where(unstable, torch.zeros_like(qz), qv[..., 0] / qz)
    sy = torch.where(unstable, torch.zeros_like(qz), qv[..., 1] / qz)
    rho3d = torch.where(unstable, torch.full_like(qz, 1e8), sx * sx + sy * sy)
    ddx = xy[:, 0, None, None] - px
    ddy = xy[:, 1, None, None] - py
    rho2d = 2.0 * (ddx * ddx + ddy * ddy)
    lowpass = rho2d.detach() < rho3d.detach()
    rho = torch.where(lowpass, rho2d, rho3d)
    if freeze_lowpass_depth:
        sxd, syd = torch.where(lowpass, sx.detach(), sx), torch.where(lowpass, sy.detach(), sy)
    else:
        sxd, syd = sx, sy
    depth = sxd * Tw[:, None, None, 0] + syd * Tw[:, None, None, 1] + Tw[:, None, None, 2]
    alpha = torch.clamp(opac[:, 0, None, None] * torch.exp(-0.5 * rho), max=ALPHA_MAX)
    usable = in_rect & visible[:, None, None] & ~(depth.detach() < NEAR) & ~(alpha.detach() < ALPHA_MIN)
    order = torch.argsort(tz.detach(), stable=True)
    w, T_final, T_before = composite(alpha, usable, order)
    blended = w.detach() > 0
    color = torch.einsum("phw,pc->chw", w, rgb) + T_final[None] * bg[:, None, None]
    refl_map = torch.einsum("phw,p->hw", w, refl[:, 0])[None]
    safe_depth = torch.where(blended, depth, torch.ones_like(depth))
    D = (w * safe_depth).sum(dim=0)
    N = torch.einsum("phw,pc->chw", w, normal)
    m = FAR / (FAR - NEAR) * (1 - NEAR / safe_depth)
    # distortion: sum over ordered pairs i, front to back, of w_i (m_i^2 A_i + M2_i - 2 m_i M1_i) with A, M1, M2 accumulated before i
    dist = torch.zeros(H, W, dtype=dt)
    M1 = torch.zeros(H, W, dtype=dt)
    M2 = torch.zeros(H, W, dtype=dt)
    median = torch.zeros(H, W, dtype=dt)
    for i in order.tolist():
        Ai = 1 - T_before[i]
        dist = dist + w[i] * (m[i] * m[i] * Ai + M2 - 2 * m[i] * M1)
        M1 = M1 + m[i] * w[i]
        M2 = M2 + m[i] * m[i] * w[i]
        median = torch.where(blended[i] & (T_before[i].detach() > 0.5), safe_depth[i], median)
    inside_scope = (blended & (mask[:, None, None] != 0)).any(dim=0).to(dt)
    allmap = torch.stack([D, 1 - T_final, N[0], N[1], N[2], median, dist, inside_scope])
    gw = torch.where(blended, w.detach(), torch.zeros_like(w)).reshape(P, -1).max(dim=1).values
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    return dict(color=color, allmap=allmap, refl_strength_map=refl_map, radii=radii, gaussian_weights=gw, final_T=T_final, Tm=Tm,
                lowpass_pairs=int((lowpass & blended).sum()))
