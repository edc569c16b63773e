"""Dev tool: time the trace kernels on the BASELINE configs (device-resident rays, HIP-event kernel time)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from raycore_jl_amd._capi import check, lib, ptr


def to_dev(a):
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()


def time_trace(t, rays, mode, reps=int(os.environ.get("RC_PROBE_REPS", "5"))):
    n = len(rays)
    d_rays = to_dev(rays)
    d_hits = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    best = 1e9
    loop = int(os.environ.get("RC_PROBE_LOOP", "0"))
    if loop:  # like bench.py's steps: `loop` launches back to back between two events (what a launch costs INCLUDING whatever the library enqueues around the kernel)
        for _ in range(12):
            t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode=mode)
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(loop):
                t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode=mode, stream=torch.cuda.current_stream().cuda_stream)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / loop)
        reps = 0
    for _ in range(reps):
        t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode=mode)
        best = min(best, t.last_kernel_ms())
    hits = d_hits.cpu().numpy().view(rc.HIT_DT)
    return best, hits


def build(cfg):
    t = rc.TLAS(0)
    for verts, meta in cfg["blas"]:
        t.add_geometry(verts, meta)
    for b, xf, ids in cfg["instances"]:
        t.push_instances(b, xf, ids)
    t0 = time.time()
    t.sync()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="kernel=0;kernel=1", help="';'-separated option sets, each 'k=v,k=v'")
    ap.add_argument("--c3res", type=int, default=2048)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--workloads", default="c2,c3,shadow,c4")
    args = ap.parse_args()
    sc = rc.scenes
    cfg2 = sc.config_c2()
    t0 = time.time(); t2 = build(cfg2); print(f"C2 build+sync {time.time()-t0:.3f}s")
    rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
    cfg3 = sc.config_c3()
    t0 = time.time(); t3 = build(cfg3); print(f"C3 build+sync {time.time()-t0:.3f}s")
    rays3 = sc.c3_primary_rays(cfg3, args.c3res, args.c3res)
    ms, hits3 = time_trace(t3, rays3, "closest", 1)
    shadow = sc.c3_shadow_rays(cfg3, rays3, hits3)
    bounce = sc.c4_bounce_rays(cfg3, rays3, hits3, 4 * len(rays3))
    wl_extra = {}
    for key, nt in (("r1m", 1_000_000), ("r4m", 4_000_000)):
        if key in args.workloads.split(","):
            tb = rc.TLAS(0)
            tb.add_geometry(sc.random_triangles(nt, 42, edge=0.01))
            tb.push_instances(1)
            tb.sync()
            wl_extra[key] = (f"random {nt//1000}k tris", tb, rc.generate_ray_grid(tb, (0.3, 0.2, 1.0), 1000), "closest")
    # the memory-bound regimes of bench.py's extras: 4 194 304 incoherent rays on a 4 M-triangle BLAS (512 MB of nodes: beyond L2, partly in the
    # 256 MiB Infinity Cache) and on a 16 M-triangle BLAS (2 GB of nodes: DRAM must serve)
    for key, nt in (("hbm", 4_000_000), ("hbm16", 16_000_000)):
        if key in args.workloads.split(","):
            g = np.random.default_rng(7)
            ro = g.random((4194304, 3))
            rd = g.standard_normal((4194304, 3))
            rd /= np.linalg.norm(rd, axis=1, keepdims=True)
            tb = rc.TLAS(0)
            dv = torch.from_numpy(sc.random_triangles(nt, 42, edge=0.01)).cuda()
            tb.add_geometry_device(dv.data_ptr(), nt)
            tb.push_instances(1)
            tb.sync()
            del dv
            wl_extra[key] = (f"{nt // 1_000_000} M tris incoherent", tb, sc.make_rays(ro, rd), "closest")
    wl = {"c2": ("C2 closest", t2, rays2, "closest"), "c3": ("C3 primary", t3, rays3, "closest"),
          "shadow": ("C3 shadow-any", t3, shadow, "any"), "c4": ("C4 bounce", t3, bounce, "closest")}
    wl.update(wl_extra)
    defaults = {"kernel": -1, "blocks_per_cu": 0, "lds_stack": 24, "refill": 20, "stats": 0, "sched_thr": 36, "pool": 0, "claim_shards": 16, "taper": 12, "cost_order": 1, "cost_thr": 64, "entry_cull": 1, "stack16": 1,
                }
    first_hits = {}
    for var in args.variants.split(";"):
        opts = dict(defaults)
        for kv in var.split(","):
            if kv:
                k, v = kv.split("=")
                opts[k] = int(v)
        for t in [t2, t3] + [w[1] for w in wl_extra.values()]:
            for k, v in opts.items():
                t.set_option(k, v)
        for w in args.workloads.split(","):
            name, t, rays, mode = wl[w]
            ms, hits = time_trace(t, rays, mode)
            extra = ""
            if w not in first_hits:
                first_hits[w] = hits.copy()
            elif hits.tobytes() != first_hits[w].tobytes():
                extra = f" !!! {int((hits.view(np.uint8).reshape(len(hits), -1) != first_hits[w].view(np.uint8).reshape(len(hits), -1)).any(axis=1).sum())} HITS DIFFER from the first variant"
            if args.stats and opts["kernel"] in (1, 2, 3):
                t.set_option("stats", 1)
                ms_stats = time_trace(t, rays, mode, 2)[0]
                v = [t.get_option(f"stat{i}") for i in range(8)]
                tv = [t.get_option(f"stat{c}") for c in "89abcd"]
                t.set_option("stats", 0)
                if opts["kernel"] == 1:
                    extra = f" wave_steps={v[0]} lanes/step={v[1]/max(v[0],1):.1f} max_sp={v[2]}"
                elif opts["kernel"] == 3:
                    extra = f" | I {v[2]} x{v[3]/max(v[2],1):.1f} | L {v[4]} x{v[5]/max(v[4],1):.1f} | S {v[6]} x{v[7]/max(v[6],1):.1f} | refills {v[0]}"
                    m64 = (1 << 64) - 1
                    t0, tx, te = m64 - (tv[0] & m64), m64 - (tv[1] & m64), tv[2]
                    nw = max(tv[5], 1)
                    extra += (f" | stats run {ms_stats * 1e3:.0f} us; ticks/100: total {(te - t0) / 100:.0f}, first wave out of work at {(tx - t0) / 100:.0f}, mean wave idle-at-end "
                              f"{(te - t0 - tv[4] / nw) / 100:.0f}, mean wave drain {tv[3] / nw / 100:.0f}")
                else:
                    extra = (f" iters={v[0]} live/iter={v[1]/max(v[0],1):.1f} | I {v[2]} x{v[3]/max(v[2],1):.1f} | L {v[4]} x{v[5]/max(v[4],1):.1f}"
                             f" | E {v[6]} x{v[7]/max(v[6],1):.1f}")
            extra += f" drift={t.get_option('claim_drift')}"
            print(f"[{var:40s}] {name:14s} n={len(rays):9d} {ms:9.3f} ms  {len(rays)/ms/1e3:9.1f} Mrays/s  hit={hits['hit'].mean():.3f}{extra}", flush=True)


if __name__ == "__main__":
    main()
