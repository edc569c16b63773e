import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import raycore_jl_amd as rc
sc=rc.scenes
for nt in (1_000_000, 4_000_000, 16_000_000):
    v=sc.random_triangles(nt, 42, edge=0.01)
    t=rc.TLAS(0); t.add_geometry(v); t.push_instances(1)
    t0=time.perf_counter(); t.sync(); t.wait_for_gpu(); dt=time.perf_counter()-t0
    # transform-only update: refit path (k_inst_recs again, no radius kernel)
    print(nt, "first sync (build + flat arrays + cull radius) %.1f ms" % (dt*1e3), flush=True)
    t.free()
