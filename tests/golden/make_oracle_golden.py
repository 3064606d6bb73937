"""Writes tests/golden/raster_tiny.npz: inputs seeds and the CPU checker's
outputs for two tiny scenes.  The fixture pins the checker against drift and
gives the GPU tests committed expected values.  Run:  python tests/golden/make_oracle_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gs_oracle  # noqa: E402
import synth  # noqa: E402

CASES = {"a": dict(P=96, width=40, height=24, seed=7, aa=False, bg=(0.3, 0.6, 0.1)),
         "b": dict(P=300, width=75, height=45, seed=9, aa=True, bg=(1.0, 1.0, 1.0))}


def run_case(c):
    sc = synth.tiny_scene(P=c["P"], width=c["width"], height=c["height"], seed=c["seed"])
    f = gs_oracle.forward_from_scene(sc, bg=c["bg"], antialiasing=c["aa"])
    rng = np.random.default_rng(c["seed"] + 100)
    H, W = f.H, f.W
    gc = rng.standard_normal((3, H, W)).astype(np.float32)
    gd = rng.standard_normal((1, H, W)).astype(np.float32)
    gr = rng.standard_normal((4, H, W)).astype(np.float32)
    g = f.backward(gc, gd, gr)
    return sc, f, (gc, gd, gr), g


def main():
    out = {}
    for name, c in CASES.items():
        sc, f, (gc, gd, gr), g = run_case(c)
        out.update({f"{name}_radii": f.radii, f"{name}_keys": f.keys, f"{name}_point_list": f.point_list,
                    f"{name}_ranges": f.ranges, f"{name}_n_contrib": f.n_contrib, f"{name}_gauss_count": f.gauss_count,
                    f"{name}_last_contr": f.last_contr, f"{name}_color": f.out_color, f"{name}_invdepth": f.invdepth,
                    f"{name}_reps": f.out_reps, f"{name}_gauss_sum": f.gauss_sum, f"{name}_R": np.int64(f.R)})
        for k, v in g.items():
            out[f"{name}_{k}"] = v
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "raster_tiny.npz"), **out)
    print("wrote raster_tiny.npz")


if __name__ == "__main__":
    main()
