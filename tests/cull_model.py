"""numpy restatement of the product's entry cull (test infrastructure): the per-BLAS leaf-box radius (k_cull_radius), the per-instance sphere
and margins (k_inst_recs, raycore.jl_amd/csrc/rc_build.hip) and the kernel's segment test (switch phase of rc_traverse_core.h), with the same
constants.  tests/test_entry_cull_predicate.py holds it against the oracle's record of what the reference does inside every instance."""
import numpy as np

F = np.float32


def blas_radii(descs, prims):
    """(centre', r') per BLAS: centre of the root box, radius = farthest corner of any triangle's own box, x 1.000001 (k_cull_radius)."""
    out = []
    offs = list(descs["primitives_offset"]) + [len(prims)]
    for b, d in enumerate(descs):
        v = prims["v"][offs[b]:offs[b + 1]].astype(np.float32)               # (n, 3 vertices, 3)
        c = F(0.5) * (d["root_min"] + d["root_max"])
        lo, hi = v.min(axis=1), v.max(axis=1)
        m = np.maximum(np.abs(lo - c), np.abs(hi - c)).astype(np.float32)
        acc = (m[:, 0] * m[:, 0] + m[:, 1] * m[:, 1] + m[:, 2] * m[:, 2]).astype(np.float32)
        r = (np.sqrt(acc).astype(np.float32) * F(1.000001)).max() if len(v) else F(0)
        if np.isnan(v).any():
            r = F(np.nan)
        out.append((c.astype(np.float64), float(r), offs[b + 1] - offs[b]))
    return out


def _sigma_ub(q):
    return np.sqrt(np.abs(q.T @ q).sum(axis=1).max())


def instance_spheres(instances, descs, radii, a_scale=1.0):
    """[(c_w (3,), A, B)] as float32, A = inf where the instance is outside the cull's regime (k_inst_recs).  a_scale: test mutants."""
    out = []
    for inst in instances:
        b = int(inst["blas_index"]) - 1
        m = inst["inv_transform"].astype(np.float64).reshape(3, 4)
        mi, t = m[:, :3], m[:, 3]
        cl, rl, n_prims = radii[b]
        with np.errstate(all="ignore"):
            try:
                w = np.linalg.inv(mi)
            except np.linalg.LinAlgError:
                w = np.full((3, 3), np.nan)
            sW, sI = _sigma_ub(w), _sigma_ub(mi)
            cw = w @ (cl - t)
            rw = rl * sW * 1.00001
            A = a_scale * 1.01 * rw + 8.0e-5 * (np.abs(cw).sum() + rw + sW * np.abs(cl).sum())
            B = 4.0e-5 * sW
            ok = n_prims >= 2 and sW <= 100.0 and sW * sI <= 16.0 and A < 1.0e30 and bool((np.abs(cw) < 1.0e30).all())
        if not ok:
            out.append((cw.astype(np.float32), F(np.inf), F(0)))
        else:
            out.append((cw.astype(np.float32), F(A) * F(1.000001), F(B) * F(1.000001)))
    return out


def fma(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def skip_entry(sphere, o, d, tmin, closest_t):
    """The kernel's test, float32 step by step (fused multiply-adds as the kernel writes them)."""
    cw, A, B = sphere
    with np.errstate(all="ignore"):
        o, d = o.astype(np.float32), d.astype(np.float32)
        d = np.where(d == 0, F(0), d)                                   # check_direction
        dd = fma(d[2], d[2], fma(d[1], d[1], F(d[0] * d[0])))
        o1 = F(F(abs(o[0]) + abs(o[1])) + abs(o[2]))
        regime = bool(dd >= F(1.0e-2)) and bool(dd <= F(1.0e6)) and bool(o1 < F(1.0e30))
        idd = F(1.0) / dd if regime else F(np.nan)
        idl = F(1.0) / np.sqrt(dd).astype(np.float32) if dd > 0 else F(np.inf)
        c_ray = F(8.0e-5) * o1
        L = (cw - o).astype(np.float32)
        LL = fma(L[2], L[2], fma(L[1], L[1], F(L[0] * L[0])))
        bq = fma(L[2], d[2], fma(L[1], d[1], F(L[0] * d[0])))
        tc = F(bq * idd)
        d2 = fma(F(-4.0e-6), LL, fma(-tc, bq, LL))
        med = np.float32(sorted([tc, F(tmin), F(closest_t)])[1]) if not (np.isnan(tc) or np.isnan(tmin) or np.isnan(closest_t)) else F(np.nan)
        ts = F(med - tc)
        seg = fma(F(F(0.999) * ts) * ts, dd, d2)
        Ae = F(A + c_ray)
        tb = F(2.0) * fma(Ae, idl, F(abs(tc)))
        R = fma(fma(F(5.0e-6), F(dd * idl), B), tb, Ae)
        return bool(seg > F(R * R))
