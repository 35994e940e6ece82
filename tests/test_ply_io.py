"""Host-side on-disk formats: the reference's PLY vertex layout and `.map` state dict."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-reflection_amd"))
from scene.ply_io import attribute_names, load_ply, read_ply_vertices, save_ply  # noqa: E402
import gsr_synth as S  # noqa: E402


def test_attribute_order_is_the_reference_layout():
    n = attribute_names(16, 2)
    assert n[:9] == ['x', 'y', 'z', 'nx', 'ny', 'nz', 'f_dc_0', 'f_dc_1', 'f_dc_2']
    assert n[9] == 'f_rest_0' and n[53] == 'f_rest_44' and n[54:] == ['opacity', 'refl', 'scale_0', 'scale_1', 'rot_0', 'rot_1', 'rot_2', 'rot_3']
    assert len(n) == 62


def test_ply_and_map_round_trip(tmp_path):
    sc = S.make_scene(257, "S", seed=3, mu=-3.0)
    tex, fail = S.make_cubemap(8, 3, 3)
    path = str(tmp_path / "iteration_7" / "point_cloud.ply")
    save_ply(path, sc["means3D"], sc["shs"], sc["opacities"], sc["refl_strengths"], sc["scales"], sc["rotations"], cubemap=tex, fail_value=fail)
    head = open(path, "rb").read(2000).split(b"end_header\n")[0].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 257\nproperty float x\n")
    assert [l.split()[-1] for l in head.splitlines() if l.startswith("property")] == attribute_names(16, 2)
    assert os.path.getsize(path) == len(head) + len("end_header\n") + 257 * 62 * 4
    back = load_ply(path)
    for k in ("means3D", "shs", "opacities", "refl_strengths", "scales", "rotations"):
        np.testing.assert_array_equal(back[k], sc[k].astype(np.float32).reshape(back[k].shape))
    np.testing.assert_array_equal(back["cubemap"], tex)
    np.testing.assert_array_equal(back["fail"], fail)
    # channel-major storage of the SH rest block: f_rest_{c*15 + k} = shs[:, 1 + k, c]
    v = read_ply_vertices(path)
    np.testing.assert_array_equal(v["f_rest_17"], sc["shs"][:, 1 + 2, 1])
    np.testing.assert_array_equal(v["f_dc_2"], sc["shs"][:, 0, 2])
    assert np.abs(v["nx"]).max() == 0
    # the .map is a plain state dict with the reference's parameter names, loadable without unpickling code
    data = torch.load(path.replace(".ply", ".map"), weights_only=True)
    assert sorted(data.keys()) == ["params.Cubemap_failv", "params.Cubemap_texture"] and tuple(data["params.Cubemap_texture"].shape) == (6, 3, 8, 8)


def test_reads_ascii_and_extra_elements(tmp_path):
    p = tmp_path / "a.ply"
    names = attribute_names(1, 2)   # SH degree 0: no f_rest
    rows = np.arange(2 * len(names), dtype=np.float32).reshape(2, -1)
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n" + "".join(f"property float {n}\n" for n in names) +
                "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for r in rows:
            f.write(" ".join(str(float(x)) for x in r) + "\n")
    out = load_ply(str(p), max_sh_degree=0)
    assert out["shs"].shape == (2, 1, 3) and out["scales"].shape == (2, 2) and out["rotations"].shape == (2, 4)
    np.testing.assert_array_equal(out["means3D"], rows[:, :3])
    np.testing.assert_array_equal(out["rotations"], rows[:, -4:])
