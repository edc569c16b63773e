"""CPU measurement (no GPU) for the TLAS SUBTREE cull: from the oracle's step traces of sampled C3 primary / C3 shadow / C4 bounce rays,
how many of the reference's TLAS-level visits lie in a subtree whose traversal tests no triangle (what any exact subtree cull could at
most skip), and how many of those two realisable conservative bounds catch:
  * sphere: per TLAS node the (near-)minimal sphere around the entry-cull spheres (c_w, A) of the instances below it, B = max B, put
    through the product's own entry-cull segment test (tests/cull_model.py: same constants, same float32 steps);
  * box:    per TLAS node the AABB of those spheres cut with the node's own (reference) box, inflated by the ray's share of the margin,
            against the segment [t_min, closest_t] (float64 slab test: an upper bound on what a float32 implementation would catch).
A bound that says "skip" on a subtree with a triangle test is reported as a VIOLATION (none may occur: the bounds are conservative).
Visits are attributed at VISIT time (closest t of the step that visits the node), which is what a test at the pop / descend would see and
an upper bound for a test made at the parent.    python3 tools/tlas_subtree_bound.py [rays per workload] [workers]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

F = np.float32
_G = {}


def subtree_instances(nodes, n_inst):
    """per TLAS node (1-based index -> list of instance indices below it); leaves are nodes n .. 2n-1"""
    below = {}

    def rec(i):
        nd = nodes[i - 1]
        if nd["child0"] == 0xFFFFFFFF:
            below[i] = [int(nd["child1"])]
        else:
            below[i] = rec(int(nd["child0"])) + rec(int(nd["child1"]))
        return below[i]
    sys.setrecursionlimit(10000)
    rec(1)
    return below


def bounding_sphere(c, r):
    """near-minimal sphere around spheres (c_i, r_i): Badoiu-Clarkson steps from the box centre"""
    if not np.all(np.isfinite(r)):
        return c.mean(axis=0), np.inf
    lo, hi = (c - r[:, None]).min(axis=0), (c + r[:, None]).max(axis=0)
    x = 0.5 * (lo + hi)
    for k in range(1, 400):
        d = np.linalg.norm(c - x, axis=1) + r
        j = int(np.argmax(d))
        if np.linalg.norm(c[j] - x) < 1e-12:
            break
        x = x + (c[j] - x) * (1.0 / (k + 1))
    return x, float((np.linalg.norm(c - x, axis=1) + r).max())


def node_bounds(o, cm):
    nodes = o.tlas_nodes
    inst = o.instances
    n_inst = len(inst)
    sph = cm.instance_spheres(inst, o.blas_descs, cm.blas_radii(o.blas_descs, o.blas_prims))
    cw = np.array([s[0] for s in sph], np.float64); A = np.array([s[1] for s in sph], np.float64); B = np.array([s[2] for s in sph], np.float64)
    below = subtree_instances(nodes, n_inst)
    # the reference box of every non-root node = the child box its parent stores
    ref_box = {}
    for i in range(1, n_inst):
        nd = nodes[i - 1]
        ref_box[int(nd["child0"])] = (nd["aabb0_min"].astype(np.float64), nd["aabb0_max"].astype(np.float64))
        ref_box[int(nd["child1"])] = (nd["aabb1_min"].astype(np.float64), nd["aabb1_max"].astype(np.float64))
    out = {}
    for i, lst in below.items():
        if i == 1:
            continue
        c, r = cw[lst], A[lst]
        C, rho = bounding_sphere(c, r)
        rho = rho * 1.000001 + 8.0e-5 * np.abs(C).sum()   # the union's own rounding share, as k_inst_recs adds for an instance
        lo, hi = (c - r[:, None]).min(axis=0), (c + r[:, None]).max(axis=0)
        rb = ref_box[i]
        lo, hi = np.maximum(lo, rb[0]), np.minimum(hi, rb[1])
        out[i] = dict(sphere=(C.astype(np.float32), F(rho), F(B[lst].max())), lo=lo, hi=hi, bmax=float(B[lst].max()), n=len(lst))
    return out, sph


def box_skip(b, o, d, tmin, ct):
    """segment vs the tight box, inflated by the ray's margin at the farthest parameter the box can be reached at (float64)"""
    if not np.all(np.isfinite(b["lo"])) or not np.all(np.isfinite(b["hi"])):
        return False
    o = o.astype(np.float64); d = d.astype(np.float64)
    dl = np.linalg.norm(d)
    ctr, half = 0.5 * (b["lo"] + b["hi"]), 0.5 * (b["hi"] - b["lo"])
    tfar = 2.0 * (np.linalg.norm(ctr - o) + np.linalg.norm(half)) / dl
    infl = 8.0e-5 * np.abs(o).sum() + (b["bmax"] + 5.0e-6 * dl) * tfar + 8.0e-5 * np.abs(ctr).sum()
    lo, hi = b["lo"] - infl, b["hi"] + infl
    with np.errstate(all="ignore"):
        inv = 1.0 / np.where(np.abs(d) > 1e-12, d, 1e-12)
        t0, t1 = (lo - o) * inv, (hi - o) * inv
        tn, tf = np.minimum(t0, t1).max(), np.maximum(t0, t1).min()
        tn, tf = max(tn, float(tmin)), min(tf, float(ct))
    return bool(tn > tf)


def one(args):
    name, mode, rays = args
    if "o" not in _G:
        import raycore_jl_amd as rc
        from oracle import pyoracle as po
        import cull_model as cm
        from helpers import build_oracle
        cfg = rc.scenes.config_c3()
        o = build_oracle(po, cfg)
        nb, sph = node_bounds(o, cm)
        _G.update(o=o, cm=cm, nb=nb, sph=sph, n_inst=len(o.instances), leaf_inst={len(o.instances) - 1 + 1 + j: int(o.tlas_nodes[len(o.instances) - 1 + j]["child1"]) for j in range(len(o.instances))})
    o, cm, nb, sph, n_inst = _G["o"], _G["cm"], _G["nb"], _G["sph"], _G["n_inst"]
    # tallies: [steps by kind 0..4] total; per strategy: TLAS interior / entry / BLAS interior steps saved; violations
    tot = np.zeros(5, np.int64)
    strat = {k: np.zeros(6, np.int64) for k in ("ideal", "entry_cull", "sphere", "box", "sphere_at_parent")}  # [tlas_int, entries, blas_int, tests(=violations), subtree roots, of them leaves]
    for r in rays:
        ev, dp, nd, ct = o.trace_steps(r, mode)
        n = len(ev)
        kind = ev & 7
        tot += np.bincount(kind, minlength=5)[:5]
        tmin = F(0) if mode == "any" else r["tmin"]
        level_tlas = np.zeros(n, bool)   # the step visits a TLAS node (interior: kind 0, leaf: kind 2)
        level_tlas[(kind == 0) | (kind == 2)] = True
        before = np.concatenate([[1], dp[:-1]]).astype(np.int32)
        tests = np.concatenate([[0], np.cumsum((kind == 3) | (kind == 4))])
        k0 = np.concatenate([[0], np.cumsum(kind == 0)]); k1 = np.concatenate([[0], np.cumsum(kind == 1)]); k2 = np.concatenate([[0], np.cumsum(kind == 2)])
        # end of every TLAS step's subtree: first j >= k with depth-after < depth-before(k)
        ends = np.full(n, n - 1, np.int32)
        stack = []
        for k in range(n):
            # close every open subtree whose depth-before exceeds the depth after this step (they opened at or before k)
            if level_tlas[k]:
                stack.append(k)
            while stack and dp[k] < before[stack[-1]]:
                ends[stack.pop()] = k
        # parent time closest t of a TLAS step: the closest t at the step that pushed / descended to it is not recorded; approximate "at parent" by
        # the closest t of the parent's own visit = the nearest earlier TLAS-interior step whose subtree contains k
        parent_ct = ct.copy()
        open_ = []
        for k in range(n):
            while open_ and ends[open_[-1]] < k:
                open_.pop()
            if level_tlas[k]:
                if open_:
                    parent_ct[k] = ct[open_[-1]]
                if kind[k] == 0:
                    open_.append(k)
        for name_s in strat:
            acc = strat[name_s]
            k = 0
            while k < n:
                if not level_tlas[k] or (nd[k] == 1 and kind[k] == 0 and name_s != "ideal"):
                    k += 1
                    continue
                e = int(ends[k])
                is_leaf = kind[k] == 2
                node = int(nd[k])
                skip = False
                if name_s == "ideal":
                    skip = tests[e + 1] - tests[k] == 0 and not (nd[k] == 1 and kind[k] == 0)
                elif name_s == "entry_cull":
                    if is_leaf:
                        skip = cm.skip_entry(sph[_G["leaf_inst"][node]], r["o"], r["d"], tmin, ct[k])
                elif name_s in ("sphere", "sphere_at_parent"):
                    c_here = ct[k] if name_s == "sphere" else parent_ct[k]
                    s = sph[_G["leaf_inst"][node]] if is_leaf else nb[node]["sphere"]
                    skip = cm.skip_entry(s, r["o"], r["d"], tmin, c_here)
                elif name_s == "box":
                    if is_leaf:
                        skip = cm.skip_entry(sph[_G["leaf_inst"][node]], r["o"], r["d"], tmin, ct[k])
                    else:
                        skip = box_skip(nb[node], r["o"], r["d"], tmin, ct[k])
                if skip:
                    acc[0] += k0[e + 1] - k0[k]; acc[1] += k2[e + 1] - k2[k]; acc[2] += k1[e + 1] - k1[k]; acc[3] += tests[e + 1] - tests[k]
                    acc[4] += 1; acc[5] += int(is_leaf)
                    k = e + 1
                else:
                    k += 1
    return name, tot, strat


def workloads(n_rays):
    import raycore_jl_amd as rc
    from oracle import pyoracle as po
    from helpers import build_oracle
    sc = rc.scenes
    cfg = sc.config_c3()
    o = build_oracle(po, cfg)
    prim = sc.c3_primary_rays(cfg, 512, 512)
    hits = o.trace(prim, nthreads=8)
    g = np.random.default_rng(4)
    out = {"c3_primary": ("closest", prim[g.choice(len(prim), n_rays, replace=False)])}
    sh = sc.c3_shadow_rays(cfg, prim, hits)
    out["c3_shadow"] = ("any", sh[g.choice(len(sh), min(n_rays, len(sh)), replace=False)])
    c4 = sc.c4_bounce_rays(cfg, prim, hits, 1 << 18)
    out["c4_bounce"] = ("closest", c4[g.choice(len(c4), n_rays, replace=False)])
    return out


if __name__ == "__main__":
    import multiprocessing as mp
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    wl = workloads(n_rays)
    jobs = []
    for name, (mode, rays) in wl.items():
        for part in np.array_split(rays, workers * 2):
            jobs.append((name, mode, part))
    res = {}
    with mp.get_context("spawn").Pool(workers) as pool:
        for name, tot, strat in pool.imap_unordered(one, jobs):
            if name not in res:
                res[name] = [np.zeros(5, np.int64), {k: np.zeros(6, np.int64) for k in strat}]
            res[name][0] += tot
            for k in strat:
                res[name][1][k] += strat[k]
    print("TLAS subtree cull: what the reference's visits leave to skip (oracle step traces, C3 scene = 256 instances; per ray)")
    for name, (tot, strat) in res.items():
        n = len(wl[name][1])
        ti, bi, en, lt = tot[0], tot[1], tot[2], tot[3] + tot[4]
        print(f"\n{name}: {n} rays; per ray {ti / n:.2f} TLAS interior visits, {en / n:.2f} instance entries, {bi / n:.2f} BLAS interior visits, {lt / n:.2f} triangle tests")
        base = strat["entry_cull"]
        for k in ("ideal", "entry_cull", "sphere", "sphere_at_parent", "box"):
            a = strat[k]
            extra = a[0] - base[0]
            print(f"  {k:17s}: skips {a[0] / n:5.2f} TLAS interior ({100.0 * a[0] / max(ti, 1):4.1f} %), {a[1] / n:4.2f} entries ({100.0 * a[1] / max(en, 1):4.1f} %), "
                  f"{a[2] / n:5.2f} BLAS interior ({100.0 * a[2] / max(bi, 1):4.1f} %); subtree roots {a[4] / n:4.2f} per ray ({a[5] / n:4.2f} of them leaves); "
                  f"triangle tests inside skipped subtrees (violations): {a[3]}")
        after = ti + bi - strat["entry_cull"][2]
        print(f"  interior visits left after the entry cull: {after / n:.2f} per ray; a sphere subtree cull removes {100.0 * (strat['sphere'][0] + strat['sphere'][2] - base[2]) / after:.1f} % of them "
              f"(at-parent closest t: {100.0 * (strat['sphere_at_parent'][0] + strat['sphere_at_parent'][2] - base[2]) / after:.1f} %), a box one {100.0 * (strat['box'][0] + strat['box'][2] - base[2]) / after:.1f} %, "
              f"the ideal one {100.0 * (strat['ideal'][0] + strat['ideal'][2] - base[2]) / after:.1f} %")
