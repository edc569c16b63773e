"""Regenerates the committed golden vectors from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).

The reference (pure Julia) cannot be run in this image, so these are NOT reference outputs: they freeze the oracle's
behaviour on two small seeded cases so that any later change to the restatement -- or to the HIP path -- shows up as a
diff against data, not just against whatever the oracle computes today.  Inputs come from raycore.jl_amd/scenes.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raycore_jl_amd as rc  # noqa: E402
from helpers import build_oracle, random_rays  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case_c1():
    cfg = rc.scenes.config_c1()
    o = build_oracle(po, cfg)
    rays = o.ray_grid(cfg["viewdir"], cfg["grid"])
    return {"rays": rays, "closest": o.trace(rays), "any": o.trace(rays, mode="any"), "tlas_nodes": o.tlas_nodes,
            "blas_nodes": o.blas_nodes, "illumination": o.get_illumination(cfg["viewdir"], cfg["grid"]),
            "blas4_nodes": o.blas4_nodes(1), "closest4": o.trace4(1, rays), "any4": o.trace4(1, rays, mode="any"),
            "triangles": o.triangles}


def case_instanced():
    sc = rc.scenes
    xf, _, _ = sc.lattice_transforms(3, 2, 2, 1.3, 123)
    cfg = {"blas": [(sc.fan_sphere(10, 6), None), (sc.random_triangles(200, 8, lo=-0.5, hi=0.5, edge=0.2), None)],
           "instances": [(1, xf[:7], np.arange(7, dtype=np.uint32) + 1), (2, xf[7:], np.arange(5, dtype=np.uint32) + 50)]}
    o = build_oracle(po, cfg)
    wb = o.world_bound
    rays = random_rays(rc, 4096, 77, wb[:3], wb[3:])
    rays["tmin"][::9] = 0.4
    rays["tmax"][::4] = 3.0
    n = len(o.blas_prims)
    return {"xforms": xf, "rays": rays, "closest": o.trace(rays), "any": o.trace(rays, mode="any"), "tlas_nodes": o.tlas_nodes,
            "blas_nodes": o.blas_nodes, "instances": o.instances, "view_factors_16": o.view_factors(16, seed=5), "n_prims": np.array([n]),
            "contacts": o.collide_instances()[0], "blas4_nodes_2": o.blas4_nodes(2)}


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "c1_sphere.npz"), **case_c1())
    np.savez_compressed(os.path.join(HERE, "instanced_small.npz"), **case_instanced())
    print("wrote", os.listdir(HERE))
