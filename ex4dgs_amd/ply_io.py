"""The reference's checkpoint format: the two binary PLY files CGaussianModel.save_ply writes and load_ply
reads -- `point_cloud.ply` (static Gaussians) and `dynamic_point_cloud.ply` next to it
(scene/c_gaussian_model.py:473-547, :560-666 of the reference).

Each file is one `vertex` element of float32 properties, binary little-endian (what plyfile's PlyData([el]).write() emits
for a native little-endian structured array).  Column order:
  static : x y z nx ny nz | f_dc_0..2 | f_rest_0..44 | opacity | scale_0..2 | rot_0..3 | xyz_disp_0..2   (:473-488)
  dynamic: motion_xyz_{k}_{c} | motion_f_dc_0..2 | motion_f_rest_0..44 | motion_scale_0..2 | motion_opacity |
           motion_opacity_c_0..1 | motion_opacity_v_0..1 | motion_rot_{k}_{c}                               (:490-512)
SH blocks are stored channel-major: `[N,C,3] -> transpose(1,2) -> [N,3*C]` (:518-519).
The reference fills / reads these files column by column through plyfile (~60 + 7K numpy copies per file); here a file is
ONE [N,F] float32 matrix: written with a single tofile(), read with a single fromfile() and sliced, then moved to the GPU.
"""
import os

import numpy as np
import torch


def static_attributes(sh_rest=15, disp=3):
    l = ["x", "y", "z", "nx", "ny", "nz"]
    l += [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * sh_rest)]
    l += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)] + [f"xyz_disp_{i}" for i in range(disp)]
    return l


def dynamic_attributes(K, sh_rest=15, xyz_width=3, opacity_c=2, opacity_v=2):
    l = [f"motion_xyz_{i}_{j}" for i in range(K) for j in range(xyz_width)]
    l += [f"motion_f_dc_{i}" for i in range(3)] + [f"motion_f_rest_{i}" for i in range(3 * sh_rest)]
    l += [f"motion_scale_{i}" for i in range(3)] + ["motion_opacity"]
    l += [f"motion_opacity_c_{i}" for i in range(opacity_c)] + [f"motion_opacity_v_{i}" for i in range(opacity_v)]     # :503-506
    l += [f"motion_rot_{i}_{j}" for i in range(K) for j in range(4)]
    return l


def _write(path, names, matrix):
    matrix = np.ascontiguousarray(matrix, dtype="<f4")
    assert matrix.shape[1] == len(names), (matrix.shape, len(names))
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {matrix.shape[0]}\n" + \
             "".join(f"property float {n}\n" for n in names) + "end_header\n"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        matrix.tofile(f)


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
              "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4"}


def _read(path):
    """Returns (names, [N,F] float32 matrix) of the first element of a binary/ascii PLY with scalar properties."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, seen_element = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if seen_element:                  # only the first element is used (plydata.elements[0]); its data follows the header
                    while tok[0] != "end_header":
                        line = f.readline()
                        if not line:
                            raise ValueError(f"{path}: truncated PLY header")
                        tok = line.decode("ascii").split() or [""]
                    break
                seen_element, count = True, int(tok[2])
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not part of this format")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2).astype(np.float32)
        else:
            order = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(n, order + t) for n, t in props])
            raw = np.fromfile(f, dtype=dt, count=count)
            if all(t == "f4" for _, t in props) and order == "<":
                data = raw.view("<f4").reshape(count, len(props))                 # zero-copy: the whole file is one matrix
            else:
                data = np.stack([raw[n].astype(np.float32) for n, _ in props], 1) if props else np.zeros((count, 0), np.float32)
    return [n for n, _ in props], data


def _columns(names, data, prefix, key):
    sel = sorted((n for n in names if n.startswith(prefix)), key=key)
    idx = [names.index(n) for n in sel]
    return data[:, idx]


def save_ply(model, path):
    """Writes `path` (static) and its `dynamic_point_cloud.ply` sibling like CGaussianModel.save_ply (:514-547)."""
    g = lambda n: getattr(model, n).detach()
    cm = lambda t: t.transpose(1, 2).flatten(start_dim=1)                          # [N,C,3] -> channel-major [N,3C]
    xyz = g("_xyz")
    static = torch.cat([xyz, torch.zeros_like(xyz), cm(g("_features_dc")), cm(g("_features_rest")), g("_opacity"), g("_scaling"),
                        g("_rotation"), g("_xyz_disp")], dim=1).cpu().numpy()
    _write(path, static_attributes(g("_features_rest").shape[1], g("_xyz_disp").shape[1]), static)
    xm = g("_xyz_motion")
    dynamic = torch.cat([xm.flatten(start_dim=1), cm(g("_features_dc_motion")), cm(g("_features_rest_motion")), g("_scaling_motion"),
                         g("_opacity_motion"), g("_opacity_duration_center").flatten(start_dim=1),
                         g("_opacity_duration_var").flatten(start_dim=1), g("_rotation_motion").flatten(start_dim=1)], dim=1).cpu().numpy()
    _write(path.replace("point_cloud.ply", "dynamic_point_cloud.ply"),
           dynamic_attributes(xm.shape[1], g("_features_rest_motion").shape[1], xm.shape[2],
                              g("_opacity_duration_center").shape[1], g("_opacity_duration_var").shape[1]), dynamic)


def load_ply(path, device="cuda", max_sh_degree=3):
    """Returns the 15 parameter tensors (dict, names and shapes of CGaussianModel) read like load_ply (:560-666); properties are
    looked up by NAME and ordered by their numeric suffixes, so files with permuted columns load identically."""
    last = lambda x: int(x.split("_")[-1])
    last2 = lambda x: (int(x.split("_")[-2]), int(x.split("_")[-1]))
    names, d = _read(path)
    col = lambda n: d[:, names.index(n)]
    n_rest = 3 * (max_sh_degree + 1) ** 2 - 3
    rest = _columns(names, d, "f_rest_", last)
    assert rest.shape[1] == n_rest, (rest.shape, n_rest)                           # :574
    out = {
        "_xyz": np.stack([col("x"), col("y"), col("z")], 1),
        "_features_dc": np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], 1)[:, None, :],             # [N,1,3]
        "_features_rest": rest.reshape(-1, 3, n_rest // 3).transpose(0, 2, 1),                                # [N,15,3]
        "_opacity": col("opacity")[:, None],
        "_scaling": _columns(names, d, "scale_", last),
        "_rotation": _columns(names, d, "rot_", last),
        "_xyz_disp": np.stack([col("xyz_disp_0"), col("xyz_disp_1"), col("xyz_disp_2")], 1),
    }
    names, d = _read(path.replace("point_cloud.ply", "dynamic_point_cloud.ply"))
    col = lambda n: d[:, names.index(n)]
    N = d.shape[0]
    mxyz = _columns(names, d, "motion_xyz_", last2)
    mrot = _columns(names, d, "motion_rot_", last2)
    K = mrot.shape[1] // 4
    mrest = _columns(names, d, "motion_f_rest_", last)
    assert mrest.shape[1] == n_rest, (mrest.shape, n_rest)                         # :617
    out.update({
        "_xyz_motion": mxyz.reshape(N, K, mxyz.shape[1] // max(K, 1)),
        "_features_dc_motion": np.stack([col("motion_f_dc_0"), col("motion_f_dc_1"), col("motion_f_dc_2")], 1)[:, None, :],
        "_features_rest_motion": mrest.reshape(-1, 3, n_rest // 3).transpose(0, 2, 1),
        "_scaling_motion": _columns(names, d, "motion_scale_", last),
        "_opacity_motion": col("motion_opacity")[:, None],
        "_opacity_duration_center": _columns(names, d, "motion_opacity_c_", last)[:, :, None],
        "_opacity_duration_var": _columns(names, d, "motion_opacity_v_", last)[:, :, None],
        "_rotation_motion": mrot.reshape(N, K, 4),
    })
    return {k: torch.tensor(np.ascontiguousarray(v), dtype=torch.float32, device=device) for k, v in out.items()}
