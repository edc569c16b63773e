"""dev: how much does the learned claim order's quality vary from one recording to the next?  One batch repeated; per launch its kernel time,
whether it recorded, and the time level of the launches between two recordings (each rebuilt order is in use for ~7 launches)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev
sc = rc.scenes
hip = ctypes.CDLL("libamdhip64.so")
def header(t):
    h = torch.empty(48, dtype=torch.int32, device="cuda")
    hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(192), 3)
    torch.cuda.synchronize()
    return h.cpu().numpy().view(np.uint32)
def run(name, t, rays, mode, launches=int(os.environ.get("RC_Q_LAUNCHES", "160"))):
    d_r = to_dev(rays); d_h = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
    rows = []
    for k in range(launches):
        t.trace_device(d_r.data_ptr(), d_h.data_ptr(), len(rays), mode=mode)
        ms = t.last_kernel_ms()
        w = header(t)
        rows.append((ms, int(w[5]), int(w[1]), int(w[2]), [int(x) for x in w[16:20]]))
    # levels between recordings
    levels, cur = [], []
    for ms, rec, valid, thr, scale in rows[6:]:
        if rec:
            if len(cur) >= 3: levels.append((np.mean(cur), len(cur), last_scale))
            cur = []
            last_scale = scale
        else:
            cur.append(ms)
    print(f"== {name}: recording launches {[r[0] for r in rows if r[1]][:6]} ... mean {np.mean([r[0] for r in rows if r[1]]):.4f} ms")
    for m, n, scale in levels:
        print(f"   order in use for {n} launches: mean {m:.4f} ms   scale words {scale}")
    lv = [m for m, _, _ in levels]
    late = rows[launches // 2:]
    rec_late = [r[0] for r in late if r[1]]
    print(f"   levels: min {min(lv):.4f} max {max(lv):.4f} spread {100 * (max(lv) / min(lv) - 1):.1f} %")
    print(f"   SUMMARY {name}: second half of the run: mean of all launches {np.mean([r[0] for r in late]):.4f} ms, of the launches that did not record {np.mean([r[0] for r in late if not r[1]]):.4f}, "
          f"of the {len(rec_late)} that did {np.mean(rec_late):.4f}; last threshold {rows[-1][4][0]}")
cfg3 = sc.config_c3(); t3 = build(cfg3)
rays3 = sc.c3_primary_rays(cfg3, 2048, 2048)
hits3 = t3.trace(rays3)
run("C3 primary", t3, rays3, "closest")
run("C3 shadow", t3, sc.c3_shadow_rays(cfg3, rays3, hits3), "any")
cfg2 = sc.config_c2(); t2 = build(cfg2)
run("C2", t2, rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"]), "closest")
run("C3 1Mi", t3, sc.c3_primary_rays(cfg3, 1024, 1024), "closest")
tb = rc.TLAS(0); tb.add_geometry(sc.random_triangles(1_000_000, 42, edge=0.01)); tb.push_instances(1); tb.sync()
run("random 1M tris", tb, rc.generate_ray_grid(tb, (0.3, 0.2, 1.0), 1000), "closest")
