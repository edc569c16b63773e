"""Does splitting ONE mid-size batch over two streams shorten it?  The second part's persistent workgroups can only start where the first part's
exit (the LDS is full), so they fill the first launch's tail.  Wall time by events around the fork / join, against the single launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev
sc = rc.scenes


def run(name, t, rays, mode="closest"):
    n = len(rays)
    d, h = to_dev(rays), torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    def timed(parts, reps=12):
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s0):
                e0.record()
                streams = [s0, s1, s2]
                evs = []
                for k, (a, b) in enumerate(parts):
                    st = streams[k]
                    if k:
                        st.wait_event(e0)
                    t.trace_device(d.data_ptr() + 32 * a, h.data_ptr() + 32 * a, b - a, mode=mode, stream=st.cuda_stream)
                    if k:
                        ev = torch.cuda.Event(); ev.record(st); evs.append(ev)
                for ev in evs:
                    s0.wait_event(ev)
                e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return float(np.mean(ms[-8:])), min(ms)
    for co in (1, 0):
        t.set_option("cost_order", co)
        base = timed([(0, n)])
        print(f"   {name:18s} cost_order {co}: one launch            mean {base[0]:.4f} ms  best {base[1]:.4f}  ({n / base[0] / 1e3:7.1f} Mrays/s)", flush=True)
        for frac in (0.5, 0.6, 0.7, 0.8):
            n1 = (int(n * frac) // 128) * 128
            r = timed([(0, n1), (n1, n)])
            print(f"   {name:18s} cost_order {co}: split {frac:.1f} / {1 - frac:.1f}     mean {r[0]:.4f} ms  best {r[1]:.4f}  ({n / r[0] / 1e3:7.1f} Mrays/s)  {100 * (r[0] / base[0] - 1):+.1f} %", flush=True)
        n1, n2 = (int(n * 0.5) // 128) * 128, (int(n * 0.8) // 128) * 128
        r = timed([(0, n1), (n1, n2), (n2, n)])
        print(f"   {name:18s} cost_order {co}: split 0.5 / 0.3 / 0.2 mean {r[0]:.4f} ms  best {r[1]:.4f}  ({n / r[0] / 1e3:7.1f} Mrays/s)  {100 * (r[0] / base[0] - 1):+.1f} %", flush=True)
    t.set_option("cost_order", 1)


if __name__ == "__main__":
    cfg2 = sc.config_c2(); t2 = build(cfg2)
    run("C2 1M", t2, rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"]))
    cfg3 = sc.config_c3(); t3 = build(cfg3)
    prim = sc.c3_primary_rays(cfg3, 2048, 2048)
    run("C3 1Mi", t3, sc.c3_primary_rays(cfg3, 1024, 1024))
    run("C3 shadow", t3, sc.c3_shadow_rays(cfg3, prim, t3.trace(prim)), mode="any")
    run("C3 4Mi", t3, prim)
