"""Dev tool: BVH4 (closest_hit4) vs the instanced BVH2 path on the single-BLAS C2 scene, plus collapse timings."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc


def to_dev(a):
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()


def best_ms(fn, obj, reps=20):
    b = 1e9
    for _ in range(reps):
        fn()
        b = min(b, obj.last_kernel_ms())
    return b


def main():
    sc = rc.scenes
    cfg = sc.config_c2()
    verts, meta = cfg["blas"][0]
    t = rc.TLAS(0)
    t.push(verts, meta=meta)
    t.sync()
    t0 = time.time()
    blas = rc.build_blas4(verts, meta)
    print(f"build_blas4 100k (BVH2 build + collapse + sync, host wall) {1e3 * (time.time() - t0):.2f} ms, nodes4 {blas.num_interior}")
    n4 = blas.nodes
    print("child_count histogram", np.bincount(n4["child_count"], minlength=5))
    coh = rc.generate_ray_grid(t, cfg["viewdir"], cfg["grid"])
    g = np.random.default_rng(3)
    n = 1 << 20
    o = (g.random((n, 3)) * 2 - 0.5).astype(np.float32)
    tgt = g.random((n, 3)).astype(np.float32)
    inc = sc.make_rays(o, (tgt - o) / np.linalg.norm(tgt - o, axis=1, keepdims=True))
    # warm the clocks: the first kernels after an idle period run at lower frequency
    d_w = to_dev(inc)
    d_wh = torch.empty(len(inc) * 32, dtype=torch.uint8, device="cuda")
    for _ in range(200):
        t.trace_device(d_w.data_ptr(), d_wh.data_ptr(), len(inc))
    torch.cuda.synchronize()
    for name, rays in (("coherent 1M", coh), ("incoherent 1M", inc)):
        d_r = to_dev(rays)
        d_h2 = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
        d_h4 = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
        for mode in ("closest", "any"):
            ms2 = best_ms(lambda: t.trace_device(d_r.data_ptr(), d_h2.data_ptr(), len(rays), mode=mode), t)
            ms4 = best_ms(lambda: blas.trace_device(d_r.data_ptr(), d_h4.data_ptr(), len(rays), mode=mode), blas)
            torch.cuda.synchronize()
            h2, h4 = d_h2.cpu().numpy().view(rc.HIT_DT), d_h4.cpu().numpy().view(rc.HIT_DT)
            agree = np.array_equal(h2["hit"], h4["hit"]) and (mode == "any" or np.array_equal(h2["t"].view(np.uint32), h4["t"].view(np.uint32)))
            print(f"{name} {mode}: BVH2 {ms2:.3f} ms ({len(rays) / ms2 / 1e3:.0f} Mrays/s)  BVH4 {ms4:.3f} ms ({len(rays) / ms4 / 1e3:.0f} Mrays/s)  same hit/t: {agree}")
    for n_tri in (250_000, 1_000_000):
        v = sc.random_soup(n_tri, 0xB4 + n_tri) if hasattr(sc, "random_soup") else None
        if v is None:
            gg = np.random.default_rng(n_tri)
            c = gg.random((n_tri, 1, 3)).astype(np.float32)
            v = (c + (gg.random((n_tri, 3, 3)).astype(np.float32) - 0.5) * np.float32(0.01)).reshape(n_tri, 9)
        s = rc.TLAS(0)
        bi = s.add_geometry(v)
        import ctypes as C
        from raycore_jl_amd._capi import check, lib
        best = 1e9
        for _ in range(3):
            nn = C.c_uint32(0)
            t0 = time.time()
            check(lib().rc_blas4_build(s._h, bi - 1, C.byref(nn)))
            best = min(best, 1e3 * (time.time() - t0))
        print(f"collapse {n_tri} tris -> {nn.value} BVH4 nodes: {best:.2f} ms host wall, {s.last_kernel_ms():.2f} ms device events")


if __name__ == "__main__":
    main()
