"""How often a wave of the dominant trace kernel runs each of its phases, per launch (GPU box): kernel 5's STATS instantiation
(option "stats") counts, per wave and summed over the launch, the interior-loop iterations, the leaf / switch / refill passes that
had at least one lane to serve, the lanes served, and the outer iterations.  tools/isa_mix.py weights the static opcode histogram of
each phase with these.  python3 tools/phase_passes.py > profiles/r06_phase_passes_kernel5.json"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc

sc = rc.scenes
def build(cfg):
    t = rc.TLAS(0)
    for v, m in cfg["blas"]: t.add_geometry(v, m)
    for b, xf, ids in cfg["instances"]: t.push_instances(b, xf, ids)
    return t.sync()
def dev(a): return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()
cfg3 = sc.config_c3(); t3 = build(cfg3)
rays3 = sc.c3_primary_rays(cfg3, 2048, 2048)
hits3 = t3.trace(rays3)
cfg2 = sc.config_c2(); t2 = build(cfg2)
wl = {"c3": (t3, rays3, "closest"), "c3_shadow": (t3, sc.c3_shadow_rays(cfg3, rays3, hits3), "any"),
      "c4": (t3, sc.c4_bounce_rays(cfg3, rays3, hits3, 4 * len(rays3)), "closest"), "c2": (t2, rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"]), "closest")}
out = {"kernel": "k_trace_phased_lds<ANY, 768, 16, 6, false, true, STACK16> (the STATS instantiation of trace kernel 5, in the shape the product runs the scene with)", "workloads": {}}
for t_ in (t2, t3):
    if len(sys.argv) > 1 and sys.argv[1] == "nocull":
        t_.set_option("entry_cull", 0)
for name, (t, rays, mode) in wl.items():
    d_r, d_h = dev(rays), torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
    t.set_option("kernel", 5); t.set_option("stats", 1)
    t.trace_device(d_r.data_ptr(), d_h.data_ptr(), len(rays), mode=mode)
    torch.cuda.synchronize()
    v = [t.get_option(f"stat{c}") for c in "0123456789abcdef"] + [t.get_option(f"stat{i}") for i in range(16, 20)]
    t.set_option("stats", 0); t.set_option("kernel", -1)
    t.trace_device(d_r.data_ptr(), d_h.data_ptr(), len(rays), mode=mode); t.trace_device(d_r.data_ptr(), d_h.data_ptr(), len(rays), mode=mode)
    ms = t.last_kernel_ms()
    n = len(rays)
    out["workloads"][name] = {"rays": n, "mode": mode, "product_kernel_ms": round(ms, 4), "waves": v[13],
                              "refill_passes": v[0], "interior_passes": v[2], "interior_lanes": v[3], "leaf_passes": v[4], "leaf_lanes": v[5],
                              "switch_passes": v[6], "switch_lanes": v[7], "outer_iterations": v[14],
                              "exit_passes": v[15], "entry_passes": v[16], "writeout_passes": v[17], "refill_rounds": v[18], "entries_skipped_by_the_entry_cull": v[19],
                              "lanes_per_pass": {"interior": round(v[3] / max(v[2], 1), 2), "leaf": round(v[5] / max(v[4], 1), 2), "switch": round(v[7] / max(v[6], 1), 2)},
                              "per_ray": {"interior_visits": round(v[3] / n, 3), "leaf_visits": round(v[5] / n, 3), "switch_events": round(v[7] / n, 3)}}
json.dump(out, sys.stdout, indent=1); print()
