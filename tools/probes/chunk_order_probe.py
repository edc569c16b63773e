"""dev: does a STATIC permutation of the 128-ray chunks help a batch's first launch?  The learned claim order is worth 5-10 % on repeats; is that
LPT (long chunks first, which needs knowledge) or MIXING (cheap and expensive chunks in flight together, which a fixed permutation gives for
free)?  The rays are permuted physically, chunk by chunk (coherence inside a claim untouched), and traced with cost_order 0: natural claim order
over the permuted array = that chunk order.  40 launches back to back, best of 5.
  python3 tools/probes/chunk_order_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev

def loop_ms(t, d_r, d_h, n, mode, loop=40, reps=5):
    for _ in range(6):
        t.trace_device(d_r.data_ptr(), d_h.data_ptr(), n, mode=mode)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(loop):
            t.trace_device(d_r.data_ptr(), d_h.data_ptr(), n, mode=mode, stream=torch.cuda.current_stream().cuda_stream)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / loop)
    return best

def permuted(rays, chunk_order):
    nc = len(chunk_order)
    idx = (chunk_order[:, None].astype(np.int64) * 128 + np.arange(128)[None, :]).reshape(-1)
    return np.concatenate([rays[idx], rays[nc * 128:]]), idx

def run(name, t, rays, mode):
    n = len(rays)
    nc = n // 128
    d_h = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    base_hits = None
    g = np.random.default_rng(5)
    stride = int(nc * 0.6180339887) | 1
    while np.gcd(stride, nc) != 1:
        stride += 2
    orders = {"natural": np.arange(nc), "random chunks": g.permutation(nc), f"stride {stride}": (np.arange(nc, dtype=np.int64) * stride) % nc,
              "16-chunk groups, stride": None, "reversed": np.arange(nc)[::-1].copy()}
    ng = nc // 16
    sg = int(ng * 0.6180339887) | 1
    while np.gcd(sg, ng) != 1:
        sg += 2
    grp = (np.arange(ng, dtype=np.int64) * sg) % ng
    orders["16-chunk groups, stride"] = np.concatenate([(grp[:, None] * 16 + np.arange(16)[None, :]).reshape(-1), np.arange(ng * 16, nc)])
    for label, co in orders.items():
        pr, idx = permuted(rays, co)
        d_r = to_dev(pr)
        for opt in (0, 1):
            t.set_option("cost_order", opt)
            ms = loop_ms(t, d_r, d_h, n, mode)
            hits = d_h.cpu().numpy().view(rc.HIT_DT)
            if base_hits is None:
                base_hits = hits.copy()
            else:   # same per-ray results, whatever the order
                assert hits[:nc * 128].tobytes() == base_hits[idx].tobytes() if label != "natural" else True
            print(f"{name:12s} {label:28s} cost_order={opt}  {ms:.4f} ms  {n / ms / 1e3:8.1f} Mrays/s", flush=True)
        del d_r
    t.set_option("cost_order", 1)

sc = rc.scenes
cfg3 = sc.config_c3(); t3 = build(cfg3)
rays3 = sc.c3_primary_rays(cfg3, 2048, 2048)
hits3 = t3.trace(rays3)
shadow = sc.c3_shadow_rays(cfg3, rays3, hits3)
cfg2 = sc.config_c2(); t2 = build(cfg2)
rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
run("C3 primary", t3, rays3, "closest")
run("C3 shadow", t3, shadow, "any")
run("C2", t2, rays2, "closest")
mid = sc.c3_primary_rays(cfg3, 1024, 1024)
run("C3 1Mi", t3, mid, "closest")
