"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of oracle/_build/libgsr_oracle.so, the CPU restatement of the reference
rasterizers (variant G = diff-gaussian-rasterization, variant S = diff-surfel-rasterization) and
of the cubemap encoder.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package never does.

Parity status: "parity unpinned" by reference artefacts (the reference has no tests and cannot be
compiled here); pinned by golden vectors from the importable reference utilities, known-answer
cases and float64 finite differences (tests/test_oracle_*.py).
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgsr_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with its Makefile (g++ only, a few seconds)."""
    # always through make: it is a no-op when the library is newer than every source
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for suf in ("f32", "f64"):
            getattr(_lib, f"orc_gauss_forward_{suf}").restype = ctypes.c_void_p
            getattr(_lib, f"orc_surfel_forward_{suf}").restype = ctypes.c_void_p
    return _lib


def _suf(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _real(dtype):
    return ctypes.c_float if np.dtype(dtype) == np.float32 else ctypes.c_double


def _arr(x, dtype):
    if x is None:
        return None
    a = np.ascontiguousarray(np.asarray(x), dtype=dtype)
    return a


def _p(a):
    if a is None or a.size == 0:
        return ctypes.c_void_p(0)
    return a.ctypes.data_as(ctypes.c_void_p)


_STATE_SPECS_COMMON = {
    "depths": ("real", lambda s: (s.P,)),
    "means2D": ("real", lambda s: (s.P, 2)),
    "rgb": ("real", lambda s: (s.P, 3)),
    "clamped": (np.uint8, lambda s: (s.P, 3)),
    "radii": (np.int32, lambda s: (s.P,)),
    "tiles_touched": (np.uint32, lambda s: (s.P,)),
    "point_offsets": (np.uint32, lambda s: (s.P,)),
    "keys_unsorted": (np.uint64, lambda s: (s.R,)),
    "keys": (np.uint64, lambda s: (s.R,)),
    "point_list": (np.uint32, lambda s: (s.R,)),
    "ranges": (np.uint32, lambda s: (s.tiles, 2)),
}


class _Base:
    kind = None
    extra_specs = {}

    def __init__(self, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.handle = None
        self._keep = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # interpreter shutdown
            pass

    def free(self):
        if self.handle is not None:
            getattr(lib(), f"orc_{self.kind}_free_{_suf(self.dtype)}")(ctypes.c_void_p(self.handle))
            self.handle = None

    def state(self, name):
        specs = dict(_STATE_SPECS_COMMON)
        specs.update(self.extra_specs)
        dt, shp = specs[name]
        dt = self.dtype if dt == "real" else dt
        out = np.zeros(shp(self), dtype=dt)
        if out.size:
            rc = getattr(lib(), f"orc_{self.kind}_get_{_suf(self.dtype)}")(ctypes.c_void_p(self.handle), name.encode(), _p(out))
            assert rc == 0, name
        return out

    def trapped(self):
        return bool(getattr(lib(), f"orc_{self.kind}_trapped_{_suf(self.dtype)}")(ctypes.c_void_p(self.handle)))


class GaussOracle(_Base):
    """Variant G (DGR).  forward()/backward() mirror RasterizeGaussiansCUDA / ...BackwardCUDA
    (DGR rasterize_points.cu:38-140, 142-264)."""
    kind = "gauss"
    extra_specs = {
        "cov3D": ("real", lambda s: (s.P, 6)),
        "conic_opacity": ("real", lambda s: (s.P, 4)),
        "final_T": ("real", lambda s: (s.H, s.W)),
        "n_contrib": (np.uint32, lambda s: (s.H, s.W)),
    }

    def forward(self, *, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height, image_width,
                sh_degree=0, shs=None, colors_precomp=None, normals=None, refl_strengths=None, scales=None, rotations=None,
                cov3D_precomp=None, scale_modifier=1.0, prefiltered=False, antialiasing=False):
        self.free()
        dt = self.dtype
        a = lambda x: _arr(x, dt)
        means3D = a(means3D)
        P = means3D.shape[0]
        H, W = int(image_height), int(image_width)
        shs = a(shs)
        M = 0 if shs is None or shs.size == 0 else shs.shape[1]
        ins = dict(bg=a(bg), means3D=means3D, shs=shs, colors_precomp=a(colors_precomp), normals=a(normals),
                   refl=a(refl_strengths), opacities=a(opacities), scales=a(scales), rotations=a(rotations),
                   cov3D_precomp=a(cov3D_precomp), view=a(viewmatrix), proj=a(projmatrix), campos=a(campos))
        self.P, self.H, self.W, self.M, self.D = P, H, W, M, int(sh_degree)
        self.tiles = ((W + 15) // 16) * ((H + 15) // 16)
        self.cfg = dict(scale_modifier=float(scale_modifier), tanfovx=float(tanfovx), tanfovy=float(tanfovy),
                        antialiasing=bool(antialiasing))
        out = dict(color=np.zeros((3, H, W), dt), normal_map=np.zeros((3, H, W), dt), refl_strength_map=np.zeros((1, H, W), dt),
                   invdepth=np.zeros((1, H, W), dt), radii=np.zeros((P,), np.int32))
        nr = ctypes.c_int(0)
        Rt = _real(dt)
        self.R = 0
        if P > 0:
            h = getattr(lib(), f"orc_gauss_forward_{_suf(dt)}")(
                P, self.D, M, _p(ins["bg"]), W, H, _p(ins["means3D"]), _p(ins["shs"]), _p(ins["colors_precomp"]), _p(ins["normals"]),
                _p(ins["refl"]), _p(ins["opacities"]), _p(ins["scales"]), Rt(scale_modifier), _p(ins["rotations"]),
                _p(ins["cov3D_precomp"]), _p(ins["view"]), _p(ins["proj"]), _p(ins["campos"]), Rt(tanfovx), Rt(tanfovy),
                int(bool(prefiltered)), int(bool(antialiasing)), _p(out["color"]), _p(out["normal_map"]),
                _p(out["refl_strength_map"]), _p(out["invdepth"]), _p(out["radii"]), ctypes.byref(nr))
            self.handle = h
            self.R = nr.value
        self._keep = ins
        out["num_rendered"] = self.R
        return out

    def backward(self, *, dL_dcolor, dL_dinvdepth=None, dL_dnormal_map=None, dL_drefl_strength_map=None):
        dt = self.dtype
        ins = self._keep
        P, H, W, M = self.P, self.H, self.W, self.M
        a = lambda x, shp: np.zeros(shp, dt) if x is None else _arr(x, dt)
        g_pix = a(dL_dcolor, (3, H, W))
        g_nrm = a(dL_dnormal_map, (3, H, W))
        g_refl = a(dL_drefl_strength_map, (1, H, W))
        g_inv = None if dL_dinvdepth is None else _arr(dL_dinvdepth, dt)
        out = dict(dL_dmeans2D_internal=np.zeros((P, 3), dt), dL_dmeans2D=np.zeros((P, 3), dt), dL_dconic=np.zeros((P, 2, 2), dt),
                   dL_dopacity=np.zeros((P, 1), dt), dL_dcolors=np.zeros((P, 3), dt), dL_dnormals=np.zeros((P, 3), dt),
                   dL_drefl_strengths=np.zeros((P, 1), dt), dL_dinvdepths=np.zeros((P, 1), dt), dL_dmeans3D=np.zeros((P, 3), dt),
                   dL_dcov3D=np.zeros((P, 6), dt), dL_dsh=np.zeros((P, M, 3), dt), dL_dscales=np.zeros((P, 3), dt),
                   dL_drotations=np.zeros((P, 4), dt))
        if P == 0:
            return out
        Rt = _real(dt)
        c = self.cfg
        getattr(lib(), f"orc_gauss_backward_{_suf(dt)}")(
            ctypes.c_void_p(self.handle), P, self.D, M, _p(ins["bg"]), W, H, _p(ins["means3D"]), _p(ins["shs"]),
            _p(ins["colors_precomp"]), _p(ins["normals"]), _p(ins["refl"]), _p(ins["opacities"]), _p(ins["scales"]),
            Rt(c["scale_modifier"]), _p(ins["rotations"]), _p(ins["cov3D_precomp"]), _p(ins["view"]), _p(ins["proj"]),
            _p(ins["campos"]), Rt(c["tanfovx"]), Rt(c["tanfovy"]), int(c["antialiasing"]), _p(g_pix), _p(g_nrm), _p(g_refl),
            _p(g_inv), _p(out["dL_dmeans2D_internal"]), _p(out["dL_dmeans2D"]), _p(out["dL_dconic"]), _p(out["dL_dopacity"]),
            _p(out["dL_dcolors"]), _p(out["dL_dnormals"]), _p(out["dL_drefl_strengths"]),
            _p(out["dL_dinvdepths"]) if g_inv is not None else ctypes.c_void_p(0), _p(out["dL_dmeans3D"]), _p(out["dL_dcov3D"]),
            _p(out["dL_dsh"]), _p(out["dL_dscales"]), _p(out["dL_drotations"]))
        return out


class SurfelOracle(_Base):
    """Variant S (DSR).  forward()/backward() mirror RasterizeGaussiansCUDA / ...BackwardCUDA
    (DSR rasterize_points.cu:39-151, 153-267)."""
    kind = "surfel"
    extra_specs = {
        "transMat": ("real", lambda s: (s.P, 9)),
        "normal_opacity": ("real", lambda s: (s.P, 4)),
        "final_T": ("real", lambda s: (3, s.H, s.W)),
        "n_contrib": (np.uint32, lambda s: (2, s.H, s.W)),
    }

    def forward(self, *, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height, image_width,
                sh_degree=0, shs=None, colors_precomp=None, refl_strengths=None, scales=None, rotations=None, cov3D_precomp=None,
                env_scope_mask=None, scale_modifier=1.0, prefiltered=False):
        self.free()
        dt = self.dtype
        a = lambda x: _arr(x, dt)
        means3D = a(means3D)
        P = means3D.shape[0]
        H, W = int(image_height), int(image_width)
        shs = a(shs)
        M = 0 if shs is None or shs.size == 0 else shs.shape[1]
        mask = None if env_scope_mask is None else np.ascontiguousarray(np.asarray(env_scope_mask).astype(np.uint8))
        ins = dict(bg=a(bg), means3D=means3D, mask=mask, shs=shs, colors_precomp=a(colors_precomp), refl=a(refl_strengths),
                   opacities=a(opacities), scales=a(scales), rotations=a(rotations), transMat_precomp=a(cov3D_precomp),
                   view=a(viewmatrix), proj=a(projmatrix), campos=a(campos))
        self.P, self.H, self.W, self.M, self.D = P, H, W, M, int(sh_degree)
        self.tiles = ((W + 15) // 16) * ((H + 15) // 16)
        self.cfg = dict(scale_modifier=float(scale_modifier), tanfovx=float(tanfovx), tanfovy=float(tanfovy))
        out = dict(color=np.zeros((3, H, W), dt), allmap=np.zeros((8, H, W), dt), refl_strength_map=np.zeros((1, H, W), dt),
                   radii=np.zeros((P,), np.int32), gaussian_weights=np.zeros((P,), dt))
        nr = ctypes.c_int(0)
        Rt = _real(dt)
        self.R = 0
        if P > 0:
            h = getattr(lib(), f"orc_surfel_forward_{_suf(dt)}")(
                P, self.D, M, _p(ins["bg"]), W, H, _p(ins["means3D"]), _p(ins["mask"]), _p(ins["shs"]), _p(ins["colors_precomp"]),
                _p(ins["refl"]), _p(ins["opacities"]), _p(ins["scales"]), Rt(scale_modifier), _p(ins["rotations"]),
                _p(ins["transMat_precomp"]), _p(ins["view"]), _p(ins["proj"]), _p(ins["campos"]), Rt(tanfovx), Rt(tanfovy),
                int(bool(prefiltered)), _p(out["color"]), _p(out["allmap"]), _p(out["refl_strength_map"]), _p(out["radii"]),
                _p(out["gaussian_weights"]), ctypes.byref(nr))
            self.handle = h
            self.R = nr.value
        self._keep = ins
        out["num_rendered"] = self.R
        return out

    def backward(self, *, dL_dcolor, dL_dallmap=None, dL_drefl_strength_map=None):
        dt = self.dtype
        ins = self._keep
        P, H, W, M = self.P, self.H, self.W, self.M
        a = lambda x, shp: np.zeros(shp, dt) if x is None else _arr(x, dt)
        g_pix = a(dL_dcolor, (3, H, W))
        g_all = a(dL_dallmap, (8, H, W))
        g_refl = a(dL_drefl_strength_map, (1, H, W))
        out = dict(dL_dmeans2D=np.zeros((P, 3), dt), dL_dnormal=np.zeros((P, 3), dt), dL_dopacity=np.zeros((P, 1), dt),
                   dL_dcolors=np.zeros((P, 3), dt), dL_drefl_strengths=np.zeros((P, 1), dt), dL_dmeans3D=np.zeros((P, 3), dt),
                   dL_dtransMat=np.zeros((P, 9), dt), dL_dsh=np.zeros((P, M, 3), dt), dL_dscales=np.zeros((P, 2), dt),
                   dL_drotations=np.zeros((P, 4), dt))
        if P == 0:
            return out
        Rt = _real(dt)
        c = self.cfg
        getattr(lib(), f"orc_surfel_backward_{_suf(dt)}")(
            ctypes.c_void_p(self.handle), P, self.D, M, _p(ins["bg"]), W, H, _p(ins["means3D"]), _p(ins["shs"]),
            _p(ins["colors_precomp"]), _p(ins["refl"]), _p(ins["scales"]), Rt(c["scale_modifier"]), _p(ins["rotations"]),
            _p(ins["transMat_precomp"]), _p(ins["view"]), _p(ins["proj"]), _p(ins["campos"]), Rt(c["tanfovx"]), Rt(c["tanfovy"]),
            _p(g_pix), _p(g_all), _p(g_refl), _p(out["dL_dmeans2D"]), _p(out["dL_dnormal"]), _p(out["dL_dopacity"]),
            _p(out["dL_dcolors"]), _p(out["dL_drefl_strengths"]), _p(out["dL_dmeans3D"]), _p(out["dL_dtransMat"]), _p(out["dL_dsh"]),
            _p(out["dL_dscales"]), _p(out["dL_drotations"]))
        return out


def mark_visible(means3D, viewmatrix, projmatrix, dtype=np.float32):
    m = _arr(means3D, dtype)
    v = _arr(viewmatrix, dtype)
    p = _arr(projmatrix, dtype)
    out = np.zeros((m.shape[0],), np.uint8)
    if m.shape[0]:
        getattr(lib(), f"orc_mark_visible_{_suf(dtype)}")(m.shape[0], _p(m), _p(v), _p(p), _p(out))
    return out.astype(bool)


def cubemap_forward(inputs, cubemap, fail_value, interp=1, seamless=1, dtype=np.float32):
    """CME cubemap_encode_forward; returns outputs [C, B]."""
    x = _arr(inputs, dtype)
    cm = _arr(cubemap, dtype)
    fv = _arr(fail_value, dtype)
    B, C, L = x.shape[0], cm.shape[1], cm.shape[2]
    out = np.zeros((C, B), dtype)
    if B:
        getattr(lib(), f"orc_cubemap_forward_{_suf(dtype)}")(_p(x), _p(cm), _p(fv), _p(out), int(interp), int(seamless), B, C, L)
    return out


def cubemap_backward(grad_outputs, inputs, cubemap, interp=1, seamless=1, dtype=np.float32):
    """CME cubemap_encode_backward; returns (grad_inputs [B,3], grad_cubemap, grad_fail [C])."""
    g = _arr(grad_outputs, dtype)
    x = _arr(inputs, dtype)
    cm = _arr(cubemap, dtype)
    B, C, L = x.shape[0], cm.shape[1], cm.shape[2]
    gcm = np.zeros_like(cm)
    gin = np.zeros((B, 3), dtype)
    gf = np.zeros((C,), dtype)
    if B:
        getattr(lib(), f"orc_cubemap_backward_{_suf(dtype)}")(_p(g), _p(x), _p(cm), _p(gcm), _p(gin), _p(gf), int(interp), int(seamless), B, C, L)
    return gin, gcm, gf


def sh_forward(deg, means, campos, shs, dtype=np.float32):
    """computeColorFromSH forward (DSR/DGR forward.cu:20-71): returns (rgb [N,3] clamped at 0, clamped flags [N,3])."""
    m, c, s = _arr(means, dtype), _arr(campos, dtype), _arr(shs, dtype)
    N, M = m.shape[0], s.shape[1]
    rgb = np.zeros((N, 3), dtype)
    cl = np.zeros((N, 3), np.uint8)
    getattr(lib(), f"orc_sh_forward_{_suf(dtype)}")(N, int(deg), M, _p(m), _p(c), _p(s), _p(rgb), _p(cl))
    return rgb, cl


def sh_backward(deg, means, campos, shs, clamped, dL_dcolor, dtype=np.float32):
    m, c, s, g = _arr(means, dtype), _arr(campos, dtype), _arr(shs, dtype), _arr(dL_dcolor, dtype)
    cl = np.ascontiguousarray(clamped, dtype=np.uint8)
    N, M = m.shape[0], s.shape[1]
    dm = np.zeros((N, 3), dtype)
    ds = np.zeros((N, M, 3), dtype)
    getattr(lib(), f"orc_sh_backward_{_suf(dtype)}")(N, int(deg), M, _p(m), _p(c), _p(s), _p(cl), _p(g), _p(dm), _p(ds))
    return dm, ds


def ssim_l1_forward(img1, img2, C1=0.01 ** 2, C2=0.03 ** 2, dtype=np.float32):
    """utils/loss_utils.py l1_loss + _ssim pieces: returns (sum |x-y|, sum ssim_map, ssim_map [C,H,W])."""
    a, b = _arr(img1, dtype), _arr(img2, dtype)
    C, H, W = a.shape
    sums = np.zeros(2, dtype)
    smap = np.zeros((C, H, W), dtype)
    r = _real(dtype)
    getattr(lib(), f"orc_ssim_l1_forward_{_suf(dtype)}")(_p(a), _p(b), C, H, W, r(C1), r(C2), _p(sums), _p(smap))
    return float(sums[0]), float(sums[1]), smap


def ssim_l1_backward(img1, img2, w_l1, w_ssim, C1=0.01 ** 2, C2=0.03 ** 2, dtype=np.float32):
    """d(w_l1 * sum|x-y| + w_ssim * sum ssim_map) / d img1, [C,H,W]."""
    a, b = _arr(img1, dtype), _arr(img2, dtype)
    C, H, W = a.shape
    d = np.zeros((C, H, W), dtype)
    r = _real(dtype)
    getattr(lib(), f"orc_ssim_l1_backward_{_suf(dtype)}")(_p(a), _p(b), C, H, W, r(C1), r(C2), r(w_l1), r(w_ssim), _p(d))
    return d


def adam(param, grad, exp_avg, exp_avg_sq, lr_per_elem, beta1=0.9, beta2=0.999, eps=1e-15, step=1, dtype=np.float32):
    """torch.optim.Adam single step (no amsgrad / weight decay); returns updated (param, exp_avg, exp_avg_sq)."""
    p, g, m, v, lr = (np.array(_arr(x, dtype).reshape(-1), copy=True) for x in (param, grad, exp_avg, exp_avg_sq, lr_per_elem))
    getattr(lib(), f"orc_adam_{_suf(dtype)}")(_p(p), _p(g), _p(m), _p(v), ctypes.c_uint64(p.size), _p(lr), ctypes.c_double(beta1),
                                             ctypes.c_double(beta2), ctypes.c_double(eps), int(step))
    return p, m, v


def surface_pass(allmap, raymat, depth_ratio, g_surf_depth=None, g_surf_normal=None, dtype=np.float32):
    """render()'s depth select + depth_to_normal (gaussian_renderer/__init__.py:151-176, utils/point_utils.py:9-37).
    Returns (surf_depth [H,W], surf_normal [3,H,W], g_allmap [8,H,W] or None when no cotangent is given)."""
    am, ray = _arr(allmap, dtype), _arr(raymat, dtype)
    H, W = am.shape[1], am.shape[2]
    sd, sn = np.zeros((H, W), dtype), np.zeros((3, H, W), dtype)
    want = g_surf_depth is not None or g_surf_normal is not None
    gsd, gsn = _arr(g_surf_depth, dtype), _arr(g_surf_normal, dtype)
    gam = np.zeros((8, H, W), dtype) if want else None
    getattr(lib(), f"orc_surface_{_suf(dtype)}")(_p(am), _p(ray), _real(dtype)(depth_ratio), H, W, _p(sd), _p(sn), _p(gsd), _p(gsn), _p(gam))
    return sd, sn, gam
