"""VERDICT r3 #5a: the claim order is learned from the previous launch of the same SHAPE -- what if consecutive same-size batches differ
(two cameras, alternating light samples)?  Two different 1 M-ray batches A / B alternating on one stream, cost_order 1 vs 0, against each
batch repeated on its own; C2 (two view directions), random geometry, and two 1 Mi-ray pinhole views of the C3 scene."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev
sc = rc.scenes


def run(name, t, a, b, mode="closest", reps=12):
    n = len(a)
    assert len(b) == n
    da, db = to_dev(a), to_dev(b)
    h = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    out = {}
    for label, co, seq in (("A A A A ... cost_order 0", 0, "AA"), ("A A A A ... cost_order 1", 1, "AA"), ("B B B B ... cost_order 1", 1, "BB"),
                           ("A B A B ... cost_order 0", 0, "AB"), ("A B A B ... cost_order 1", 1, "AB")):
        t.set_option("cost_order", co)
        ms = []
        for k in range(reps):
            d = da if seq[k % 2] == "A" else db
            t.trace_device(d.data_ptr(), h.data_ptr(), n, mode=mode)
            ms.append(t.last_kernel_ms())
        steady = np.array(ms[4:])
        out[label] = steady.mean()
        print(f"   {name:22s} {label:28s} mean of launches 5..{reps}: {steady.mean():.4f} ms  ({n / steady.mean() / 1e3:7.1f} Mrays/s)  min {steady.min():.4f} max {steady.max():.4f}", flush=True)
    loss = out["A B A B ... cost_order 1"] / out["A B A B ... cost_order 0"] - 1
    print(f"   {name:22s} alternating batches, learned order vs none: {100 * loss:+.1f} % time", flush=True)
    t.set_option("cost_order", 1)


if __name__ == "__main__":
    cfg2 = sc.config_c2(); t2 = build(cfg2)
    run("C2 two view dirs", t2, rc.generate_ray_grid(t2, (0.3, 0.2, 1.0), 1000), rc.generate_ray_grid(t2, (-0.8, 0.4, 0.3), 1000))
    run("C2 mirrored image", t2, rc.generate_ray_grid(t2, (0.3, 0.2, 1.0), 1000), np.ascontiguousarray(rc.generate_ray_grid(t2, (0.3, 0.2, 1.0), 1000)[::-1]))
    cfg3 = sc.config_c3(); t3 = build(cfg3)
    eye2 = cfg3["lattice_centre"] + np.array([14.0, 3.0, -6.0])
    run("C3 1Mi two cameras", t3, sc.c3_primary_rays(cfg3, 1024, 1024), sc.pinhole_rays(1024, 1024, eye2, cfg3["lattice_centre"], 45.0))
    tb = rc.TLAS(0); tb.add_geometry(sc.random_triangles(1_000_000, 42, edge=0.01)); tb.push_instances(1); tb.sync()
    run("random 1M tris", tb, rc.generate_ray_grid(tb, (0.3, 0.2, 1.0), 1000), rc.generate_ray_grid(tb, (1.0, -0.2, 0.1), 1000))
