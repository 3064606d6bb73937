"""The point_cloud.ply wire format (scene/gaussian_model.py:554-638) without plyfile:
header text, property order, channel-major SH blocks, round trip, foreign files."""
import os
import struct

import numpy as np
import pytest

import gsplat_io


def _model(P=37, M=16, dim=4, seed=0):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    return dict(xyz=f(P, 3), features_dc=f(P, 1, 3), features_rest=f(P, M - 1, 3), opacity=f(P, 1), scaling=f(P, 3),
                rotation=f(P, 4), reps=f(P, dim))


def test_header_and_layout(tmp_path):
    m = _model()
    path = os.path.join(tmp_path, "a", "point_cloud.ply")
    gsplat_io.save_ply(path, m["xyz"], m["features_dc"], m["features_rest"], m["opacity"], m["scaling"], m["rotation"])
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    props = [l.split()[2] for l in lines[3:]]
    # construct_list_of_attributes (scene/gaussian_model.py:607-619) for degree 3
    want = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
           ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert props == want and all(l.split()[:2] == ["property", "float"] for l in lines[3:])
    assert len(body) == 37 * len(want) * 4
    row0 = struct.unpack("<%df" % len(want), body[:len(want) * 4])
    assert row0[:3] == tuple(m["xyz"][0]) and row0[3:6] == (0.0, 0.0, 0.0)
    # SH blocks are channel-major: f_rest_k = features_rest[:, k % 15, k // 15]
    assert row0[9 + 0] == m["features_rest"][0, 0, 0] and row0[9 + 1] == m["features_rest"][0, 1, 0]
    assert row0[9 + 15] == m["features_rest"][0, 0, 1] and row0[9 + 44] == m["features_rest"][0, 14, 2]
    assert row0[9 + 45] == m["opacity"][0, 0]


@pytest.mark.parametrize("M,with_reps", [(16, True), (16, False), (4, True), (1, False)])
def test_round_trip(tmp_path, M, with_reps):
    m = _model(P=101, M=M, seed=M)
    path = os.path.join(tmp_path, "pc.ply")
    gsplat_io.save_ply(path, m["xyz"], m["features_dc"], m["features_rest"], m["opacity"], m["scaling"], m["rotation"],
                       reps=m["reps"] if with_reps else None)
    deg = int(round(np.sqrt(M))) - 1
    r = gsplat_io.load_ply(path, max_sh_degree=deg)
    for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        assert r[k].dtype == np.float32 and np.array_equal(r[k], m[k]), k
    assert (r["reps"] is None) == (not with_reps)
    if with_reps:
        assert np.array_equal(r["reps"], m["reps"])
    with pytest.raises(ValueError):
        gsplat_io.load_ply(path, max_sh_degree=deg + 1)


def test_torch_tensors_and_ascii_file(tmp_path):
    import torch
    m = _model(P=5, M=4)
    path = os.path.join(tmp_path, "t.ply")
    gsplat_io.save_ply(path, *[torch.tensor(m[k], requires_grad=True) for k in
                               ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")])
    assert np.array_equal(gsplat_io.load_ply(path)["rotation"], m["rotation"])
    # an ascii file with the properties in another order and an extra element after the vertices
    names = gsplat_io.attribute_names(3, 9)
    order = names[::-1]
    table = np.arange(2 * len(names), dtype=np.float64).reshape(2, -1)
    with open(os.path.join(tmp_path, "ascii.ply"), "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n")
        f.write("".join(f"property float {n}\n" for n in order))
        f.write("element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for row in table:
            f.write(" ".join(repr(float(x)) for x in row) + "\n")
    r = gsplat_io.load_ply(os.path.join(tmp_path, "ascii.ply"), max_sh_degree=1)
    assert r["xyz"][1, 0] == table[1, order.index("x")] and r["features_rest"].shape == (2, 3, 3)
    assert r["features_rest"][0, 2, 1] == table[0, order.index("f_rest_5")]     # coefficient 2, channel 1 -> 1*3+2


def test_checkpoint_round_trip_in_the_reference_layout(tmp_path):
    """((capture 13-tuple), iteration) as train.py:371-376 writes it; restore order of
    scene/gaussian_model.py:145-165."""
    import torch
    import gsplat_io
    P = 37
    g = torch.Generator().manual_seed(3)
    params = dict(xyz=torch.nn.Parameter(torch.randn(P, 3, generator=g)), reps=torch.nn.Parameter(torch.randn(P, 4, generator=g)),
                  features_dc=torch.nn.Parameter(torch.randn(P, 1, 3, generator=g)),
                  features_rest=torch.nn.Parameter(torch.randn(P, 15, 3, generator=g)),
                  scaling=torch.nn.Parameter(torch.randn(P, 3, generator=g)),
                  rotation=torch.nn.Parameter(torch.randn(P, 4, generator=g)),
                  opacity=torch.nn.Parameter(torch.randn(P, 1, generator=g)))
    opt = torch.optim.Adam([{"params": [p], "lr": 1e-3, "name": n} for n, p in params.items()], lr=0.0, eps=1e-15)
    for p in params.values():
        p.grad = torch.ones_like(p)
    opt.step()
    fields = dict(params, active_sh_degree=2, max_radii2D=torch.rand(P), xyz_gradient_accum=torch.rand(P, 1),
                  denom=torch.rand(P, 1), optimizer_state_dict=opt.state_dict(), spatial_lr_scale=4.5)
    path = str(tmp_path / "chkpnt7000.pth")
    gsplat_io.save_checkpoint(path, 7000, **fields)
    # what the reference's train.py does with the file
    model_params, first_iter = torch.load(path, weights_only=False)
    assert first_iter == 7000 and len(model_params) == 13
    assert model_params[0] == 2 and model_params[12] == 4.5
    assert torch.equal(model_params[1], params["xyz"]) and torch.equal(model_params[2], params["reps"])
    assert torch.equal(model_params[7], params["opacity"]) and torch.equal(model_params[8], fields["max_radii2D"])
    got, it = gsplat_io.load_checkpoint(path)
    assert it == 7000 and set(got) == set(gsplat_io.CAPTURE_FIELDS)
    for k in params:
        assert torch.equal(got[k], params[k])
    opt2 = torch.optim.Adam([{"params": [p], "lr": 1e-3, "name": n} for n, p in params.items()], lr=0.0, eps=1e-15)
    opt2.load_state_dict(got["optimizer_state_dict"])
    for p in params.values():
        assert torch.equal(opt2.state[p]["exp_avg"], opt.state[p]["exp_avg"])
    with pytest.raises(ValueError):
        gsplat_io.save_checkpoint(path, 1, xyz=params["xyz"])
