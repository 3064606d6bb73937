"""`point_cloud.ply` in the reference's layout, without the plyfile package.

Wire format (scene/gaussian_model.py:607-638, written through plyfile as
binary little endian): one `vertex` element of float32 properties in the order
x y z nx ny nz | f_dc_0..2 | f_rest_0..(3(M-1)-1) | opacity | scale_0..2 |
rot_0..3, all pre-activation values; the SH blocks are channel-major (the
[P, K, 3] tensors are transposed to [P, 3, K] before flattening).  `load_ply`
restores the tensors as scene/gaussian_model.py:554-604 does.  The fork does
not store its extra `reps` tensor; `save_ply(..., reps=...)` appends optional
`rep_*` properties after the reference's ones and `load_ply` returns them
when present, so files of the reference load unchanged.

Parity of this format is pinned only by the reference's code (no .ply fixture
ships with it and plyfile is not available to write one).
"""
import os

import numpy as np


def _np(t):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t, np.float32)


def attribute_names(n_dc, n_rest, n_scale=3, n_rot=4, n_rep=0):
    """construct_list_of_attributes (scene/gaussian_model.py:607-619) plus optional rep_*."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    names += [f"rep_{i}" for i in range(n_rep)]
    return names


def save_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation, reps=None):
    xyz, f_dc, f_rest = _np(xyz), _np(features_dc), _np(features_rest)
    opacity, scaling, rotation = _np(opacity).reshape(-1, 1), _np(scaling), _np(rotation)
    P = xyz.shape[0]
    f_dc = f_dc.reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)        # channel-major, as the reference
    f_rest = f_rest.reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, opacity, scaling, rotation]
    n_rep = 0
    if reps is not None and _np(reps).size:
        r = _np(reps).reshape(P, -1)
        n_rep = r.shape[1]
        cols.append(r)
    table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], scaling.shape[1], rotation.shape[1], n_rep)
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
          "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
          "uint": "u4", "uint32": "u4"}


def read_vertices(path):
    """The `vertex` element of a binary or ascii PLY as a structured numpy array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, count, props, in_vertex, first_element = None, 0, [], False, True
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header without end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    if not first_element:
                        raise ValueError("the vertex element must come first")
                    count = int(tok[2])
                first_element = False
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported in the vertex element")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            flat = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2)
            out = np.empty(count, dtype=[(n, "<" + t) for n, t in props])
            for k, (n, _) in enumerate(props):
                out[n] = flat[:, k]
            return out
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)


def load_ply(path, max_sh_degree=None):
    """Returns a dict of float32 arrays in the model's layout: xyz [P,3],
    features_dc [P,1,3], features_rest [P,M-1,3], opacity [P,1], scaling [P,3],
    rotation [P,4] and reps [P,DIM] or None (scene/gaussian_model.py:567-604)."""
    v = read_vertices(path)
    names = v.dtype.names
    P = v.shape[0]
    col = lambda n: np.asarray(v[n], np.float32)
    numbered = lambda prefix: sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    f_dc = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], axis=1).reshape(P, 3, 1)
    rest_names = numbered("f_rest_")
    if max_sh_degree is not None and len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError("the file's SH degree does not match max_sh_degree")
    f_rest = np.stack([col(n) for n in rest_names], axis=1).reshape(P, 3, -1) if rest_names else np.zeros((P, 3, 0), np.float32)
    scales = np.stack([col(n) for n in numbered("scale_")], axis=1)
    rots = np.stack([col(n) for n in numbered("rot_")], axis=1)
    rep_names = numbered("rep_")
    return dict(
        xyz=xyz,
        features_dc=np.ascontiguousarray(f_dc.transpose(0, 2, 1)),
        features_rest=np.ascontiguousarray(f_rest.transpose(0, 2, 1)),
        opacity=col("opacity").reshape(P, 1),
        scaling=scales,
        rotation=rots,
        reps=np.stack([col(n) for n in rep_names], axis=1) if rep_names else None,
    )


# ------------------------------------------------------------------ training checkpoints (.pth)
# train.py:371-376 saves `torch.save((gaussians.capture(), iteration), path)` and train.py:77-80
# loads it back into `gaussians.restore(model_params, opt)`.  The capture tuple of the fork has 13
# entries (scene/gaussian_model.py:166-182); its order is restated here once so that a trainer built
# on the raw-parameter path reads and writes files the reference's train.py can resume from.
CAPTURE_FIELDS = ("active_sh_degree", "xyz", "reps", "features_dc", "features_rest", "scaling", "rotation", "opacity",
                  "max_radii2D", "xyz_gradient_accum", "denom", "optimizer_state_dict", "spatial_lr_scale")


def save_checkpoint(path, iteration, **fields):
    """Writes the reference's checkpoint: ((13-tuple in CAPTURE_FIELDS order), iteration).
    Pass every field by name; tensors are stored as given (nn.Parameter or plain, like capture())."""
    import torch
    missing = [k for k in CAPTURE_FIELDS if k not in fields]
    extra = [k for k in fields if k not in CAPTURE_FIELDS]
    if missing or extra:
        raise ValueError(f"checkpoint fields: missing {missing}, unknown {extra}")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save((tuple(fields[k] for k in CAPTURE_FIELDS), int(iteration)), path)


def load_checkpoint(path, map_location=None):
    """Reads a checkpoint written by the reference's train.py (or by save_checkpoint) into
    ({field: value}, iteration).  Loaded with weights_only=True: tensors, nn.Parameter and plain
    containers only, nothing in the file is executed."""
    import torch
    allow = getattr(torch.serialization, "safe_globals", None)
    ctx = allow([torch.nn.Parameter]) if allow else None
    if ctx is not None:
        with ctx:
            model_params, iteration = torch.load(path, map_location=map_location, weights_only=True)
    else:
        model_params, iteration = torch.load(path, map_location=map_location, weights_only=True)
    if len(model_params) != len(CAPTURE_FIELDS):
        raise ValueError(f"expected the fork's {len(CAPTURE_FIELDS)}-entry capture tuple, got {len(model_params)} entries")
    return dict(zip(CAPTURE_FIELDS, model_params)), int(iteration)
