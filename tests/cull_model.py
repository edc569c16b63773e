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


# The margins of the derivation (rc_build.hip above k_inst_recs; the kernel's switch phase), by name.  `m` scales every one of them together
# (tools/cull_margin_slack.py: how far can the margins shrink before the claim breaks?); single entries may be overridden.
NOMINAL = dict(r_pad=0.01,      # A = (1 + r_pad) r_w ...
               k_abs=8.0e-5,    # ... + k_abs (|c_w|_1 + r_w + sigma(W) |c'|_1);  A_ray = k_abs |o|_1
               k_b=4.0e-5,      # B = k_b sigma(W)
               k_d=5.0e-6,      # (B + k_d |d|) t_bound
               k_ll=4.0e-6,     # squared distance reduced by k_ll |c_w - o|^2
               k_seg=1.0e-3)    # the segment's share of the squared distance counted (1 - k_seg) times


def margins(m=1.0, **override):
    out = {k: v * m for k, v in NOMINAL.items()}
    out.update(override)
    return out


def instance_spheres(instances, descs, radii, a_scale=1.0, mg=None):
    """[(c_w (3,), A, B)] as float32, A = inf where the instance is outside the cull's regime (k_inst_recs).  a_scale: test mutants."""
    mg = mg or NOMINAL
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
            A = a_scale * (1.0 + mg["r_pad"]) * rw + mg["k_abs"] * (np.abs(cw).sum() + rw + sW * np.abs(cl).sum())
            B = mg["k_b"] * sW
            ok = n_prims >= 2 and sW <= 100.0 and sW * sI <= 16.0 and A < 1.0e30 and bool((np.abs(cw) < 1.0e30).all())
        if not ok:
            out.append((cw.astype(np.float32), F(np.inf), F(0)))
        else:
            out.append((cw.astype(np.float32), F(A) * F(1.000001), F(B) * F(1.000001)))
    return out


def fma(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def ulps(x, k):
    """x moved by k units in the last place (float32)"""
    x = F(x)
    for _ in range(abs(int(k))):
        x = np.nextafter(x, F(np.inf) if k > 0 else F(-np.inf), dtype=np.float32)
    return x


def skip_entry(sphere, o, d, tmin, closest_t, mg=None, ulp_idd=0, ulp_idl=0, ratio=False):
    """The kernel's test, float32 step by step (fused multiply-adds as the kernel writes them).
    The kernel takes 1 / (d . d) and 1 / |d| from v_rcp_f32 / v_rsq_f32, which are 1-ulp approximations, where this model divides exactly
    (ADVICE r3): ulp_idd / ulp_idl move the two by that many ulps so that a campaign can cover what the hardware may return.  The median
    of (tc, t_min, closest_t) is v_med3_f32; with a NaN among its inputs the hardware returns min3 / an input rather than NaN -- the
    model returns NaN (skip = False) there, and the regime test has already replaced idd by NaN for such rays, so tc is NaN in both and
    every later comparison is false in both.
    ratio=True: return seg / R^2 (skip iff > 1) instead of the decision -- how close an entry came to being skipped."""
    mg = mg or NOMINAL
    cw, A, B = sphere
    with np.errstate(all="ignore"):
        o, d = o.astype(np.float32), d.astype(np.float32)
        d = np.where(d == 0, F(0), d)                                   # check_direction
        dd = fma(d[2], d[2], fma(d[1], d[1], F(d[0] * d[0])))
        o1 = F(F(abs(o[0]) + abs(o[1])) + abs(o[2]))
        regime = bool(dd >= F(1.0e-2)) and bool(dd <= F(1.0e6)) and bool(o1 < F(1.0e30))
        idd = ulps(F(1.0) / dd, ulp_idd) if regime else F(np.nan)
        idl = ulps(F(1.0) / np.sqrt(dd).astype(np.float32), ulp_idl) if dd > 0 else F(np.inf)
        c_ray = F(mg["k_abs"]) * o1
        L = (cw - o).astype(np.float32)
        LL = fma(L[2], L[2], fma(L[1], L[1], F(L[0] * L[0])))
        bq = fma(L[2], d[2], fma(L[1], d[1], F(L[0] * d[0])))
        tc = F(bq * idd)
        d2 = fma(F(-mg["k_ll"]), LL, fma(-tc, bq, LL))
        med = np.float32(sorted([tc, F(tmin), F(closest_t)])[1]) if not (np.isnan(tc) or np.isnan(tmin) or np.isnan(closest_t)) else F(np.nan)
        ts = F(med - tc)
        seg = fma(F(F(1.0 - mg["k_seg"]) * ts) * ts, dd, d2)
        Ae = F(A + c_ray)
        tb = F(2.0) * fma(Ae, idl, F(abs(tc)))
        R = fma(fma(F(mg["k_d"]), F(dd * idl), B), tb, Ae)
        if ratio:
            return float(seg) / float(F(R * R)) if np.isfinite(R) and R > 0 else 0.0
        return bool(seg > F(R * R))
