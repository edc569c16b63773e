"""How much headroom do the entry cull's safety margins have?  CPU campaign (no GPU) over the hostile
scenes of tools/cull_predicate_campaign.py: for every instance entry the oracle records,
  * nominal constants, with the kernel's two hardware reciprocals moved by +-2 ulp in all four sign combinations: violations must be 0;
  * the TIGHTEST fruitful entry: max over entries in which the reference tested a triangle of seg / R^2 (an entry is skipped iff > 1);
  * every margin scaled together by m = 0.5, 0.25, ... 0: at which m does a skipped entry with a triangle test first appear;
  * each margin alone set to zero (the others nominal): which of them the claim actually leans on.
python3 tools/cull_margin_slack.py [seeds] [workers]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

SWEEP = [1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.0]
SINGLES = ["r_pad", "k_abs", "k_b", "k_d", "k_ll", "k_seg"]
ULPS = [(2, 2), (2, -2), (-2, 2), (-2, -2)]


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
    n = 600
    org = g.uniform(-5, 5, size=(n, 3)); tgt = g.uniform(-3.5, 3.5, size=(n, 3))
    # half of the rays graze an instance's sphere (where the margins decide), the rest are the campaign's random rays
    radii = cm.blas_radii(o.blas_descs, o.blas_prims)
    nominal = cm.instance_spheres(o.instances, o.blas_descs, radii)
    finite = [s for s in nominal if np.isfinite(s[1])]
    if finite:
        for k in range(n // 2):
            cw, A, _ = finite[int(g.integers(0, len(finite)))]
            dirn = g.normal(size=3); dirn /= np.linalg.norm(dirn)
            u = np.cross(dirn, g.normal(size=3)); u /= np.linalg.norm(u)
            r_w = float(A) / 1.01
            p = cw.astype(np.float64) + u * r_w * float(g.choice([0.9, 0.97, 0.99, 1.0, 1.005, 1.01, 1.02, 1.05]))
            back = float(g.choice([0.0, 0.5, 5.0, 40.0]))
            org[k] = p - dirn * back; tgt[k] = p + dirn
    d = tgt - org; d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= g.choice([1.0, 1.0, 0.3, 3.0, 30.0], size=(n, 1))      # unnormalised directions inside the regime |d|^2 in [1e-2, 1e6]
    # the rays the margins exist for: aimed, in the instance's LOCAL frame, along a coordinate axis (the other two direction components
    # below safe_invdir's 1e-5 clamp, so the slab test follows a ray bent by up to 1.74e-5 |t|) past the leaf-box corner that defines
    # the BLAS's radius, from far away, offset outwards by a fraction of what the clamp can bend them back
    adv_o, adv_d = [], []
    offs = list(o.blas_descs["primitives_offset"]) + [len(o.blas_prims)]
    for inst in o.instances:
        b = int(inst["blas_index"]) - 1
        v = o.blas_prims["v"][offs[b]:offs[b + 1]].astype(np.float64)
        if len(v) < 2:
            continue
        m = inst["inv_transform"].astype(np.float64).reshape(3, 4)
        try:
            w = np.linalg.inv(m[:, :3])
        except np.linalg.LinAlgError:
            continue
        if not np.all(np.isfinite(w)) or np.linalg.cond(m[:, :3]) > 16:
            continue
        cl = radii[b][0]
        lo, hi = v.min(axis=1), v.max(axis=1)
        far = np.where(np.abs(lo - cl) > np.abs(hi - cl), lo, hi)               # per triangle: the corner of its box farthest from the centre
        k = int(np.argmax(((far - cl) ** 2).sum(axis=1)))
        corner, nrm = far[k], (far[k] - cl) / max(np.linalg.norm(far[k] - cl), 1e-30)
        for _ in range(24):
            ax = int(g.integers(0, 3))
            dl = np.zeros(3); dl[ax] = float(g.choice([1.0, -1.0]))
            tiny = g.uniform(-2e-5, 2e-5, 3); tiny[ax] = 0.0
            dl = dl + tiny * float(g.choice([0.0, 0.3, 1.0]))
            T = float(g.choice([1.0, 10.0, 100.0, 1000.0, 3000.0]))
            side = nrm - nrm.dot(dl) * dl                                        # outwards, across the ray
            if np.linalg.norm(side) < 1e-9:
                continue
            side /= np.linalg.norm(side)
            delta = float(g.choice([0.0, 0.2, 0.5, 0.9, 1.0, 1.1, 1.5, 2.0, 3.0])) * 1.0e-5 * T + float(g.choice([0.0, 1e-7, 1e-6])) * np.linalg.norm(far[k] - cl)
            ol = corner + side * delta - dl * T
            adv_o.append(w @ (ol - m[:, 3])); adv_d.append(w @ dl)
    if adv_o:
        ao, ad = np.array(adv_o), np.array(adv_d)
        ok = np.all(np.isfinite(ao), axis=1) & np.all(np.isfinite(ad), axis=1) & (np.abs(ao).max(axis=1) < 1e6)
        org, d = np.concatenate([org, ao[ok]]), np.concatenate([d, ad[ok]])
    rays = sc.make_rays(org, d)
    rays["tmin"][::7] = g.uniform(-1, 1, len(rays["tmin"][::7]))
    rays["tmax"][::5] = g.uniform(0, 8, len(rays["tmax"][::5]))
    sph = {("m", m): cm.instance_spheres(o.instances, o.blas_descs, radii, mg=cm.margins(m)) for m in SWEEP}
    for name in SINGLES:
        sph[("z", name)] = cm.instance_spheres(o.instances, o.blas_descs, radii, mg=cm.margins(1.0, **{name: 0.0}))
    out = {"entries": 0, "fruitful": 0, "skipped_nominal": 0, "max_ratio_fruitful": 0.0, "ulp_violations": 0}
    out.update({f"viol_m{m}": 0 for m in SWEEP}); out.update({f"skip_m{m}": 0 for m in SWEEP}); out.update({f"viol_zero_{k}": 0 for k in SINGLES})
    for mode in ("closest", "any"):
        for r in rays:
            inst, ct, lf = o.trace_entries(r, mode)
            tmin = np.float32(0) if mode == "any" else r["tmin"]
            for i, c, l in zip(inst, ct, lf):
                i = int(i)
                out["entries"] += 1
                fruit = l > 0
                out["fruitful"] += int(fruit)
                skip_nom = cm.skip_entry(sph[("m", 1.0)][i], r["o"], r["d"], tmin, c)
                out["skipped_nominal"] += int(skip_nom)
                if fruit:
                    out["max_ratio_fruitful"] = max(out["max_ratio_fruitful"], cm.skip_entry(sph[("m", 1.0)][i], r["o"], r["d"], tmin, c, ratio=True))
                    for a, b in ULPS:
                        out["ulp_violations"] += int(cm.skip_entry(sph[("m", 1.0)][i], r["o"], r["d"], tmin, c, ulp_idd=a, ulp_idl=b))
                for m in SWEEP:
                    sk = skip_nom if m == 1.0 else cm.skip_entry(sph[("m", m)][i], r["o"], r["d"], tmin, c, mg=cm.margins(m))
                    out[f"skip_m{m}"] += int(sk)
                    out[f"viol_m{m}"] += int(sk and fruit)
                if fruit:
                    for name in SINGLES:
                        out[f"viol_zero_{name}"] += int(cm.skip_entry(sph[("z", name)][i], r["o"], r["d"], tmin, c, mg=cm.margins(1.0, **{name: 0.0})))
    return seed, out


if __name__ == "__main__":
    import multiprocessing as mp
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    total = None
    with mp.get_context("spawn").Pool(workers) as pool:
        for seed, out in pool.imap_unordered(one, range(seeds)):
            if total is None:
                total = dict(out)
            else:
                for k, v in out.items():
                    total[k] = max(total[k], v) if k == "max_ratio_fruitful" else total[k] + v
    print(f"{seeds} hostile scenes x (600 rays, half of them grazing an instance's cull sphere, + 24 clamp-bent far rays per instance) x (closest, any): {total['entries']} instance entries, "
          f"{total['fruitful']} with a triangle test, {total['skipped_nominal']} skipped by the nominal test")
    print(f"nominal constants, hardware reciprocals moved by +-2 ulp (4 sign combinations per fruitful entry): {total['ulp_violations']} violations")
    r = total["max_ratio_fruitful"]
    print(f"tightest fruitful entry: seg / R^2 = {r:.4f} (skipped iff > 1): the squared clearance could shrink by {1 / max(r, 1e-30):.2f} x, the radius by {1 / np.sqrt(max(r, 1e-30)):.3f} x before that entry flips")
    print("all margins scaled together by m:")
    for m in SWEEP:
        print(f"   m = {m:<8}: {total[f'skip_m{m}']:9d} entries skipped, {total[f'viol_m{m}']:6d} of them with a triangle test")
    print("one margin at zero, the others nominal (violations among fruitful entries):")
    for name in SINGLES:
        print(f"   {name:6s} = 0: {total[f'viol_zero_{name}']}")
