"""Dev tool: the trace kernel in an HBM-bound regime -- trees far larger than L2 (32 MiB) + Infinity Cache (256 MiB), incoherent rays.

For each scene (random small triangles in the unit cube, one BLAS) it reports Grays/s, the reference algorithm's node fetches per ray
(counted by the product's own instrumented kernel, `stats` option: interior + leaf lane-visits), hence ALGORITHMIC bytes per launch
(32 + 32 + 60 x fetches).  Run it under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/gpu_hbm_regime.sh) to get the PHYSICAL
HBM bytes; their ratio is the fetch amplification (a 64-byte node read drags a 128-byte line in)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from perf_probe import time_trace

sc = rc.scenes


def incoherent_rays(n, seed):
    g = np.random.default_rng(seed)
    o = g.random((n, 3))
    d = g.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return sc.make_rays(o, d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", default="4000000,16000000")
    ap.add_argument("--rays", type=int, default=4_194_304)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-stats", action="store_true", help="timed launches only (the form profiled under rocprofv3)")
    args = ap.parse_args()
    res = {}
    rays = incoherent_rays(args.rays, 7)
    for nt in [int(x) for x in args.tris.split(",")]:
        t = rc.TLAS(0)
        dv = torch.from_numpy(sc.random_triangles(nt, 42, edge=0.01)).cuda()
        t.add_geometry_device(dv.data_ptr(), nt)
        t.push_instances(1)
        t.sync()
        del dv
        t.set_option("kernel", -1)
        ms, hits = time_trace(t, rays, "closest", 5)
        entry = {"triangles": nt, "node_bytes": (2 * nt - 1) * 64, "rays": len(rays), "ms": round(ms, 4), "mrays_s": round(len(rays) / ms / 1e3, 1), "hit_fraction": float(hits["hit"].mean())}
        if not args.no_stats:
            t.set_option("kernel", 3); t.set_option("stats", 1)
            time_trace(t, rays, "closest", 1)
            v = [t.get_option(f"stat{i}") for i in range(8)]
            t.set_option("stats", 0); t.set_option("kernel", -1)
            fetches = (v[3] + v[5]) / len(rays) + 1.0   # interior + BLAS-leaf lane-visits, + the TLAS leaf of the single instance
            entry.update({"node_fetches_per_ray": round(fetches, 3), "instance_entries_per_ray": 1.0,
                          "algorithmic_bytes_per_launch": (64 + 60.0 * fetches + 140.0) * len(rays)})
        res[str(nt)] = entry
        print(json.dumps({str(nt): entry}), flush=True)
        t.free()
        torch.cuda.empty_cache()
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
