"""VERDICT r3 #5b: what does a batch cost the FIRST time it is traced?  Eight different 1 M-ray batches per scene (view directions / camera
positions), launched one after the other on one stream so that every launch is its batch's first (the history's four batch slots are
recycled), under cost_order 0 (natural order), cost_order 1 + first_order 0 (machinery on, nothing predicted) and cost_order 1 +
first_order 1 (box-count prediction).  Times are HIP-event launch times including the small order kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev
sc = rc.scenes


def run(name, t, batches, mode="closest", rounds=3):
    n = len(batches[0])
    dev = [to_dev(b) for b in batches]
    h = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    res = {}
    for label, opts in (("cost_order 0", {"cost_order": 0}), ("cost_order 1, first_order 0", {"cost_order": 1, "first_order": 0}), ("cost_order 1, first_order 1", {"cost_order": 1, "first_order": 1})):
        for k, v in opts.items():
            t.set_option(k, v)
        ms = []
        for r in range(rounds + 1):
            for d in dev:
                t.trace_device(d.data_ptr(), h.data_ptr(), n, mode=mode)
                if r:  # round 0 warms the allocator / the history entry
                    ms.append(t.last_kernel_ms())
        res[label] = float(np.mean(ms))
        print(f"   {name:20s} {label:30s} mean first-launch time {np.mean(ms):.4f} ms  ({n / np.mean(ms) / 1e3:7.1f} Mrays/s)  min {np.min(ms):.4f} max {np.max(ms):.4f}", flush=True)
    # and the steady state of ONE batch for reference
    t.set_option("cost_order", 1); t.set_option("first_order", 1)
    ms = []
    for k in range(8):
        t.trace_device(dev[0].data_ptr(), h.data_ptr(), n, mode=mode)
        ms.append(t.last_kernel_ms())
    print(f"   {name:20s} {'batch 0 repeated (learned)':30s} launches 4..8: {np.mean(ms[3:]):.4f} ms  ({n / np.mean(ms[3:]) / 1e3:7.1f} Mrays/s); its first (predicted) launch {ms[0]:.4f}, second {ms[1]:.4f}", flush=True)
    return res


def dirs(k):
    g = np.random.default_rng(5)
    v = g.normal(size=(k, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[0] = sc.normalize(np.array([0.3, 0.2, 1.0]))
    return v.astype(np.float32)


if __name__ == "__main__":
    which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "r1m", "c3", "shadow"]
    if "c2" in which:
        cfg2 = sc.config_c2(); t2 = build(cfg2)
        run("C2 1M rays", t2, [rc.generate_ray_grid(t2, d, 1000) for d in dirs(8)])
    if "r1m" in which:
        tb = rc.TLAS(0); tb.add_geometry(sc.random_triangles(1_000_000, 42, edge=0.01)); tb.push_instances(1); tb.sync()
        run("random 1M tris", tb, [rc.generate_ray_grid(tb, d, 1000) for d in dirs(8)])
    if "c3" in which or "shadow" in which:
        cfg3 = sc.config_c3(); t3 = build(cfg3)
        eyes = [cfg3["lattice_centre"] + 16.0 * d for d in dirs(8).astype(np.float64)]
        eyes[0] = cfg3["eye"]
        prim = [sc.pinhole_rays(1024, 1024, e, cfg3["lattice_centre"], 45.0) for e in eyes]
        if "c3" in which:
            run("C3 1Mi primary", t3, prim)
            run("C3 4Mi primary", t3, [sc.pinhole_rays(2048, 2048, e, cfg3["lattice_centre"], 45.0) for e in eyes[:4]] * 2)
        if "shadow" in which:
            sh = []
            for p in prim:
                s = sc.c3_shadow_rays(cfg3, p, t3.trace(p))
                sh.append(s)
            m = min(len(s) for s in sh)
            run("C3 shadow (any)", t3, [np.ascontiguousarray(s[:m]) for s in sh], mode="any")
