"""The C ABI without Python in the loop: tests/cabi_host/cabi_host.cpp (HIP runtime + include/gsr_hip.h, no torch, no ctypes) is built with
hipcc, run on the scene this test writes to disk, and its outputs are compared with the Python path's (forward bit for bit, gradients to the
noise of float atomics).  What a maintainer's binding in any language sees: pointers, sizes, an allocation callback, a stream, error codes."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import HipSurfel, S, rel_maxnorm, scene_kwargs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting-reflection_amd")


def test_c_host_program_matches_the_python_path(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "cabi_host"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_host", "cabi_host.cpp"),
                        "-L", PKG, "-lgsr_hip", "-Wl,-rpath," + PKG, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    P, W, H = 6000, 200, 136
    kw, cam, sc = scene_kwargs("S", P, W, H, 61, -2.8, 3, (0.2, 0.1, 0.3))
    g = S.make_upstream_grads(H, W, 61)
    d = tmp_path / "scene"
    d.mkdir()
    M = kw["shs"].shape[1]
    (d / "meta.txt").write_text("%d %d %d %d %d %.9g %.9g\n" % (P, 3, M, W, H, kw["tanfovx"], kw["tanfovy"]))
    arrays = {"means3D": kw["means3D"], "shs": kw["shs"], "opacities": kw["opacities"], "scales": kw["scales"], "rotations": kw["rotations"],
              "refl": kw["refl_strengths"], "mask": kw["env_scope_mask"].astype(np.uint8), "view": kw["viewmatrix"], "proj": kw["projmatrix"],
              "campos": kw["campos"], "bg": np.asarray(kw["bg"], np.float32), "g_color": g["dL_dcolor"], "g_others": g["dL_dplanes"], "g_refl": g["dL_drefl"]}
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        assert a.dtype in (np.float32, np.uint8), (name, a.dtype)
        a.tofile(d / (name + ".bin"))
    r = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    hip = HipSurfel(kw)
    out = hip.out()
    grads = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    rd = lambda name, dt=np.float32: np.fromfile(d / ("out_" + name + ".bin"), dtype=dt)
    assert int((d / "num_rendered.txt").read_text()) == out["num_rendered"]
    assert np.array_equal(rd("radii", np.int32), out["radii"].reshape(-1))
    assert np.array_equal(rd("color"), out["color"].reshape(-1))
    assert np.array_equal(rd("others"), out["allmap"].reshape(-1))
    assert np.array_equal(rd("refl"), out["refl_strength_map"].reshape(-1))
    assert np.array_equal(rd("weights"), out["gaussian_weights"].reshape(-1))
    for name, key in (("dmeans3D", "dL_dmeans3D"), ("dsh", "dL_dsh"), ("dopacity", "dL_dopacity"), ("dscales", "dL_dscales"), ("drot", "dL_drotations"),
                      ("drefl", "dL_drefl_strengths"), ("dmeans2D", "dL_dmeans2D")):
        want = grads[key].reshape(-1)
        got = rd(name)
        assert got.shape == want.shape and np.isfinite(got).all(), name
        assert rel_maxnorm(got, want) <= 5e-5, name
