"""Pins the pieces of the oracle / host logic that CAN be checked against the reference itself: golden
vectors produced by the reference's importable utilities (tests/golden/make_golden.py, run in the build
container where /root/reference is mounted).  Everything else in the oracle is "parity unpinned" by reference
artefacts and pinned by known answers + float64 finite differences (test_oracle_known_answers.py,
test_oracle_fd.py)."""
import os

import numpy as np

import gsr_synth as S
from oracle import oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sh_colour_matches_reference_eval_sh():
    g = np.load(os.path.join(G, "sh_golden.npz"))
    for deg in range(4):
        rgb, cl = orc.sh_forward(deg, g["means"], g["campos"], g["shs"])
        np.testing.assert_allclose(rgb, g[f"rgb_deg{deg}"], rtol=0, atol=1e-6)
        # clamp flags = (unclamped colour + 0.5 < 0); skip values within rounding of the threshold
        raw = g[f"raw_deg{deg}"] + 0.5
        sure = np.abs(raw) > 1e-6
        assert ((cl == 1) == (raw < 0))[sure].all()
        # float64 instantiation of the same text
        rgb64, _ = orc.sh_forward(deg, g["means"], g["campos"], g["shs"], dtype=np.float64)
        np.testing.assert_allclose(rgb64, g[f"rgb_deg{deg}"], rtol=0, atol=1e-6)


def test_sh_layout_is_P_M_3():
    """The kernels take (P, 16, 3); the reference's Python helper takes (P, 3, 16) (gaussian_renderer/__init__.py:118)."""
    g = np.load(os.path.join(G, "sh_golden.npz"))
    shs = g["shs"].copy()
    shs[:, 1:, :] = 0.0  # only DC left: colour = C0 * dc + 0.5
    rgb, _ = orc.sh_forward(3, g["means"], g["campos"], shs)
    np.testing.assert_allclose(rgb, np.maximum(0.28209479177387814 * shs[:, 0, :] + 0.5, 0), atol=1e-6)


def test_camera_matrices_match_reference_graphics_utils():
    g = np.load(os.path.join(G, "camera_golden.npz"))
    for i in range(int(g["n"])):
        fovx, fovy, W, H = g[f"fov{i}"]
        cam = S.make_camera(int(W), int(H), fovy_deg=np.degrees(fovy), R=g[f"R{i}"], T=g[f"T{i}"])
        np.testing.assert_allclose(cam["FoVx"], fovx, rtol=1e-12)
        np.testing.assert_allclose(cam["viewmatrix"], g[f"wvt{i}"], atol=1e-6)
        np.testing.assert_allclose(cam["projmatrix"], g[f"full{i}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(cam["campos"], g[f"center{i}"], atol=2e-5)
        np.testing.assert_allclose(S.projection_matrix(0.01, 100.0, fovx, fovy).T, g[f"proj{i}"], rtol=1e-6, atol=1e-7)
        # intrinsics used for the camera rays: K with principal point at the image centre gives the same projection
        np.testing.assert_allclose(cam["K"][0, 0], g[f"focal{i}"][0], rtol=1e-6)
        np.testing.assert_allclose(cam["K"][1, 1], g[f"focal{i}"][1], rtol=1e-6)
        np.testing.assert_allclose(g[f"projc{i}"], g[f"proj{i}"], rtol=1e-4, atol=1e-5)
