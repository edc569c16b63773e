"""Dev (GPU): FIRST-LAUNCH rates of the trace kernels for one build of the library -- every launch traces rays it has never seen, read from HBM
(distinct buffers, far more bytes than the Infinity Cache holds), default options.  Run once per library variant inside ONE gpurun call
(tools/ab_build.sh builds variants into tools/ab/):

    python tools/ab_fresh.py                      # the in-tree library
    python tools/ab_fresh.py tools/ab/noprefetch.so

Workloads: C3 4 Mi jittered primaries (the bench headline), C3 1 Mi jittered primaries, C2 1 M grid rays from perturbed view directions, C3 shadow
rays of jittered primaries (any_hit), and for reference the same-buffer figures (one cache-warm buffer, cost_order 0)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import raycore_jl_amd as rc
    if len(sys.argv) > 1:
        path = os.path.abspath(sys.argv[1])
        sys.modules[rc.lib.__module__].LIB_PATH = path
    import torch
    sc = rc.scenes
    opts = dict(kv.split("=") for kv in os.environ.get("RC_AB_OPTS", "").split(",") if kv)   # e.g. RC_AB_OPTS=refill=28,pool=256
    label = (os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "in-tree") + ("".join(f" {k}={v}" for k, v in opts.items()))
    stream = torch.cuda.current_stream()

    def up(a):
        return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()

    def b2b(t, bufs, out, n, mode="closest", rounds=1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(rounds):
            for b in bufs:
                t.trace_device(b.data_ptr(), out.data_ptr(), n, mode=mode, stream=stream.cuda_stream)
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / (rounds * len(bufs))

    def report(name, n, fresh_ms, warm_ms):
        print(f"[{label:24s}] {name:34s} fresh {n / fresh_ms / 1e3:8.1f} Mrays/s ({fresh_ms:.4f} ms)   same buffer, natural order {n / warm_ms / 1e3:8.1f} Mrays/s ({warm_ms:.4f} ms)", flush=True)

    cfg = sc.config_c3()
    t = rc.TLAS(0)
    for verts, meta in cfg["blas"]:
        t.add_geometry(verts, meta)
    for b, xf, ids in cfg["instances"]:
        t.push_instances(b, xf, ids)
    t.sync()
    for k, v in opts.items():
        t.set_option(k, int(v))
    for res, count in ((2048, 20), (1024, 40)):
        n = res * res
        bufs = [up(sc.c3_primary_rays(cfg, res, res, jitter_seed=100 + i)) for i in range(count)]
        out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        for _ in range(30):   # clocks up, on a buffer that is not among the timed ones
            t.trace_device(bufs[0].data_ptr(), out.data_ptr(), n, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        fresh = min(b2b(t, bufs[1:], out, n) for _ in range(1))
        t.set_option("cost_order", 0)
        warm = min(b2b(t, bufs[:1], out, n, rounds=20) for _ in range(3))
        t.set_option("cost_order", 1)
        report(f"C3 {res}x{res} jittered primaries", n, fresh, warm)
        if res == 2048:
            # shadow rays of three jittered batches (host side: needs the hits)
            sh = []
            for i in range(6):
                rays = sc.c3_primary_rays(cfg, res, res, jitter_seed=300 + i)
                sh.append(sc.c3_shadow_rays(cfg, rays, t.trace(rays)))
            m = min(len(s) for s in sh)
            sbufs = [up(np.ascontiguousarray(s[:m])) for s in sh]
            fresh = b2b(t, sbufs[1:], out, m, mode="any")
            t.set_option("cost_order", 0)
            warm = min(b2b(t, sbufs[:1], out, m, mode="any", rounds=20) for _ in range(3))
            t.set_option("cost_order", 1)
            report("C3 shadow rays (any_hit)", m, fresh, warm)
            del sbufs, sh
        del bufs
        torch.cuda.empty_cache()
    t.free()
    cfg2 = sc.config_c2()
    t2 = rc.TLAS(0)
    t2.add_geometry(*cfg2["blas"][0])
    t2.push_instances(1, cfg2["instances"][0][1], cfg2["instances"][0][2])
    t2.sync()
    for k, v in opts.items():
        t2.set_option(k, int(v))
    g = np.random.default_rng(5)
    vd = np.asarray(cfg2["viewdir"], dtype=np.float64)
    bufs = [up(rc.generate_ray_grid(t2, tuple(vd + 0.02 * g.standard_normal(3)), cfg2["grid"])) for _ in range(40)]
    n = cfg2["grid"] ** 2
    out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    for _ in range(30):
        t2.trace_device(bufs[0].data_ptr(), out.data_ptr(), n, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    fresh = b2b(t2, bufs[1:], out, n)
    t2.set_option("cost_order", 0)
    warm = min(b2b(t2, bufs[:1], out, n, rounds=20) for _ in range(3))
    report("C2 1M grid rays, perturbed views", n, fresh, warm)
    t2.free()


if __name__ == "__main__":
    main()
