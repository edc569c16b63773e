"""CPU campaign (no GPU): the entry cull's claim on random hostile scenes -- the generator of tests/test_gpu_fuzz.py (mirrors, shears, singular /
huge / tiny transforms, single triangles, duplicate instances) -- for every instance entry the oracle records: where the numpy restatement of
the product's test (tests/cull_model.py) says skip, the reference tested no triangle.   python3 tools/cull_predicate_campaign.py [seeds] [workers]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(seed):
    import raycore_jl_amd as rc
    from oracle import pyoracle as po
    import cull_model as cm
    from helpers import build_oracle
    from test_gpu_fuzz import hostile_transform
    sc = rc.scenes
    g = np.random.default_rng(5000 + seed)
    n_blas = int(g.integers(1, 5))
    blas = []
    for b in range(n_blas):
        nt = int(g.choice([1, 2, 3, 17, 200, 1500]))
        verts = sc.random_triangles(nt, 50 * seed + b, lo=-0.5, hi=0.5, edge=float(g.choice([0.05, 0.3, 1.0])))
        if nt > 3 and g.random() < 0.5:
            verts[1] = verts[0]
        blas.append((verts, None))
    instances = []
    for b in range(n_blas):
        m = int(g.integers(1, 7))
        xf = np.stack([hostile_transform(g, int(g.integers(0, 7)) if g.random() < 0.4 else 0) for _ in range(m)])
        if m > 1 and g.random() < 0.3:
            xf[1] = xf[0]
        instances.append((b + 1, xf, g.integers(0, 100, m).astype(np.uint32)))
    o = build_oracle(po, {"blas": blas, "instances": instances})
    n = 1500
    org = g.uniform(-5, 5, size=(n, 3)); tgt = g.uniform(-3.5, 3.5, size=(n, 3))
    d = tgt - org; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = sc.make_rays(org, d)
    rays["tmin"][::7] = g.uniform(-1, 1, len(rays["tmin"][::7]))
    rays["tmax"][::5] = g.uniform(0, 8, len(rays["tmax"][::5]))
    sph = cm.instance_spheres(o.instances, o.blas_descs, cm.blas_radii(o.blas_descs, o.blas_prims))
    tot = np.zeros(3, np.int64)
    for mode in ("closest", "any"):
        for r in rays:
            inst, ct, lf = o.trace_entries(r, mode)
            tmin = np.float32(0) if mode == "any" else r["tmin"]
            for i, c, l in zip(inst, ct, lf):
                tot[0] += 1
                if cm.skip_entry(sph[int(i)], r["o"], r["d"], tmin, c):
                    tot[1] += 1; tot[2] += int(l > 0)
    return seed, tot


if __name__ == "__main__":
    import multiprocessing as mp
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    total = np.zeros(3, np.int64)
    with mp.get_context("spawn").Pool(workers) as pool:
        for seed, tot in pool.imap_unordered(one, range(seeds)):
            total += tot
            if tot[2]:
                print("VIOLATION in seed", seed, tot, flush=True)
    print(f"{seeds} hostile scenes x 1500 rays x (closest, any): {total[0]} instance entries, {total[1]} the cull would skip, {total[2]} of those with a triangle test")
