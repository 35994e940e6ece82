"""On-disk formats of the reference, host-side only, no third-party PLY package:

* `point_cloud.ply` as written by GaussianModel.save_ply (scene/gaussian_model.py:225-258 of the reference): one binary
  little-endian `vertex` element of float32 properties
      x y z nx ny nz f_dc_0..2 f_rest_0..(3(D+1)^2-4) opacity refl scale_0..1 rot_0..3
  with f_dc / f_rest stored channel-major (`_features_*.transpose(1, 2).flatten(1)`), raw (pre-activation) values;
* `point_cloud.map` = `torch.save(env_map.state_dict())` with keys `params.Cubemap_texture` (6,3,L,L) and
  `params.Cubemap_failv` (3) (scene/gaussian_model.py:260-262, 331-336); read back with `weights_only=True`.

The in-memory layout is the one the rasterizer takes: `shs` (P, (D+1)^2, 3) = cat(f_dc, f_rest) coefficient-major.
"""
import os

import numpy as np
import torch


def attribute_names(n_sh_coeffs=16, n_scales=2):
    # scene/gaussian_model.py:225-238 (construct_list_of_attributes)
    names = ['x', 'y', 'z', 'nx', 'ny', 'nz']
    names += [f'f_dc_{i}' for i in range(3)]
    names += [f'f_rest_{i}' for i in range(3 * (n_sh_coeffs - 1))]
    names += ['opacity', 'refl']
    names += [f'scale_{i}' for i in range(n_scales)]
    names += [f'rot_{i}' for i in range(4)]
    return names


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save_ply(path, means3D, shs, opacities, refl_strengths, scales, rotations, cubemap=None, fail_value=None):
    """Writes the reference's PLY (+ the `.map` next to it when a cubemap is given).  All values raw (pre-activation)."""
    xyz, shs_, op, rf, sc, rot = (_np(a).astype(np.float32) for a in (means3D, shs, opacities, refl_strengths, scales, rotations))
    P, M = xyz.shape[0], shs_.shape[1]
    f_dc = shs_[:, 0, :]                                                   # (P,3)   = _features_dc.transpose(1,2).flatten(1)
    f_rest = np.transpose(shs_[:, 1:, :], (0, 2, 1)).reshape(P, -1)        # (P,3*(M-1)) channel-major
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, op.reshape(P, 1), rf.reshape(P, 1), sc.reshape(P, -1), rot.reshape(P, 4)],
                          axis=1).astype('<f4')
    names = attribute_names(M, sc.reshape(P, -1).shape[1])
    assert cols.shape[1] == len(names)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(cols).tobytes())
    if cubemap is not None:
        state = {"params.Cubemap_texture": torch.as_tensor(_np(cubemap)).float().contiguous(),
                 "params.Cubemap_failv": torch.as_tensor(_np(fail_value if fail_value is not None else np.zeros(3))).float().contiguous()}
        torch.save(state, path.replace('.ply', '.map'))


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
              "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply_vertices(path):
    """Minimal PLY reader: the first element must be `vertex` with scalar properties (binary little-endian or ascii).
    Returns a structured numpy array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, seen_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if seen_vertex:
                    seen_vertex = "done"   # further elements follow the vertex block; their properties are not ours
                elif tok[1] == "vertex":
                    seen_vertex, count = True, int(tok[2])
                else:
                    raise ValueError(f"{path}: first element is '{tok[1]}', expected 'vertex'")
            elif tok[0] == "property" and seen_vertex is True:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("binary_little_endian", "ascii") or count is None:
            raise ValueError(f"{path}: unsupported PLY format '{fmt}'")
        dt = np.dtype(props)
        if fmt == "ascii":
            raw = np.loadtxt(f, max_rows=count, ndmin=2)
            out = np.zeros(count, dt)
            for i, (n, _) in enumerate(props):
                out[n] = raw[:, i]
            return out
        return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)


def load_ply(path, max_sh_degree=3):
    """GaussianModel.load_ply (scene/gaussian_model.py:298-336): returns a dict of float32 arrays in the rasterizer's layout
    (means3D (P,3), shs (P,(D+1)^2,3), opacities (P,1), refl_strengths (P,1), scales (P,S), rotations (P,4)) plus
    'cubemap' / 'fail' when the sibling `.map` exists."""
    v = read_ply_vertices(path)
    names = v.dtype.names
    P = v.shape[0]
    col = lambda n: np.asarray(v[n], dtype=np.float32)
    xyz = np.stack([col('x'), col('y'), col('z')], axis=1)
    f_dc = np.stack([col('f_dc_0'), col('f_dc_1'), col('f_dc_2')], axis=1)                        # (P,3)
    rest_names = sorted([n for n in names if n.startswith('f_rest_')], key=lambda x: int(x.split('_')[-1]))
    M = (max_sh_degree + 1) ** 2
    if len(rest_names) != 3 * M - 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties, expected {3 * M - 3} for SH degree {max_sh_degree}")
    f_rest = np.stack([col(n) for n in rest_names], axis=1).reshape(P, 3, M - 1) if M > 1 else np.zeros((P, 3, 0), np.float32)
    shs = np.concatenate([f_dc[:, None, :], np.transpose(f_rest, (0, 2, 1))], axis=1)             # (P,M,3)
    scale_names = sorted([n for n in names if n.startswith('scale_')], key=lambda x: int(x.split('_')[-1]))
    rot_names = sorted([n for n in names if n.startswith('rot')], key=lambda x: int(x.split('_')[-1]))
    out = dict(means3D=xyz, shs=np.ascontiguousarray(shs, dtype=np.float32), opacities=col('opacity')[:, None],
               refl_strengths=(col('refl')[:, None] if 'refl' in names else np.zeros((P, 1), np.float32)),
               scales=np.stack([col(n) for n in scale_names], axis=1), rotations=np.stack([col(n) for n in rot_names], axis=1))
    map_path = path.replace('.ply', '.map')
    if os.path.exists(map_path):
        data = torch.load(map_path, map_location="cpu", weights_only=True)
        out["cubemap"] = data["params.Cubemap_texture"].float().numpy()
        out["fail"] = data["params.Cubemap_failv"].float().numpy()
    return out
