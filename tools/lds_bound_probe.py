"""Dev (GPU): upper bound of the texture-path lever for L2-resident scenes (VERDICT r5 'next' #3a): how much faster is the phased kernel when EVERY
interior node it visits comes from the LDS planes instead of the top ~84 % of the visits?  Scene: C3's lattice with a 1 024-triangle sphere BLAS
(32 x 17 fan sphere: 1 023 interior nodes + the TLAS's 255 = 1 278 plane entries), which a build with RC_LDS_PLANES16=1278 holds completely at
ONE workgroup per CU (111 504 B of LDS).  Compared at the same occupancy (option blocks_per_cu=1) with the shipped 748 planes (TLAS + the BLAS's
top 493 nodes) and with the BLAS top switched off (TLAS only); the shipped shape at two workgroups per CU is printed for scale.

    tools/ab_build.sh planes1278 -DRC_LDS_PLANES16=1278
    python tools/lds_bound_probe.py ; python tools/lds_bound_probe.py tools/ab/planes1278.so"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import raycore_jl_amd as rc
    if len(sys.argv) > 1:
        sys.modules[rc.lib.__module__].LIB_PATH = os.path.abspath(sys.argv[1])
    import torch
    label = os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "in-tree (748)"
    sc = rc.scenes
    cfg = sc.config_c3(lon=32, bands=17)
    t = rc.TLAS(0)
    for verts, meta in cfg["blas"]:
        t.add_geometry(verts, meta)
    for b, xf, ids in cfg["instances"]:
        t.push_instances(b, xf, ids)
    t.sync()
    stream = torch.cuda.current_stream()
    rays = sc.c3_primary_rays(cfg, 2048, 2048)
    hits = t.trace(rays)
    bounce = sc.c4_bounce_rays(cfg, rays, hits, 4 * len(rays))
    print(f"[{label}] scene: {t.n_primitives()} triangles per BLAS x {t.n_instances()} instances; hit fraction {hits['hit'].mean():.3f}", flush=True)
    t.set_option("cost_order", 0)
    ref = {}
    for name, r in (("primary 4 Mi", rays), ("bounce 16 Mi", bounce)):
        d = torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda()
        out = torch.empty(len(r) * 32, dtype=torch.uint8, device="cuda")
        for bpc, blas_top in ((1, 1), (1, 0), (0, 1)):
            t.set_option("blocks_per_cu", bpc)
            t.set_option("blas_top", blas_top)
            for _ in range(10):
                t.trace_device(d.data_ptr(), out.data_ptr(), len(r), stream=stream.cuda_stream)
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(10):
                    t.trace_device(d.data_ptr(), out.data_ptr(), len(r), stream=stream.cuda_stream)
                e1.record(stream); e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            h = out.cpu().numpy().tobytes()
            ref.setdefault(name, h)
            same = h == ref[name]
            print(f"[{label}] {name:13s} workgroups/CU {'1' if bpc == 1 else '2 (default)'}  BLAS top in LDS {'yes' if blas_top else 'no '}  {best:8.4f} ms  {len(r) / best / 1e3:8.1f} Mrays/s  same hits: {same}", flush=True)
        del d, out
    t.free()


if __name__ == "__main__":
    main()
