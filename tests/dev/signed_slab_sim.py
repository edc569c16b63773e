"""Dev tool (CPU only; under tests/ because it drives the oracle): what would a SIGN-RESOLVED slab test save?  

fast_intersect_bbox (src/instanced-bvh.jl:1841-1859) computes f = p_max * inv + ox, n = p_min * inv + ox and then max.(f, n) / min.(f, n).
For a well-formed box (p_min <= p_max, every refit output) IEEE multiply and add are monotone, so which of f / n is the larger is decided by
sign(inv) alone: 12 of the 20 min / max of an interior visit are SELECTIONS.  A per-lane select costs what a min / max costs (v_cndmask is in
the same issue class), so the selection has to come for free -- from WHERE a lane reads the planes.  Candidates priced here, per interior
pass of a wave (the kernel issues a pass's instructions however few lanes take part):

  A  (the verdict's a + b)  LDS-resident nodes read through six per-lane plane offsets; buffer-loaded nodes get static octant variants of
     the tail, usable only when every lane of the pass has the same octant.  A pass whose lanes are all LDS lanes runs the short tail; a
     mixed pass runs it only if the octant is uniform.
  B  (what was built)  the traversal copy of an interior node stores each axis as a RING (min pair, max pair, min pair): one 16-byte load
     at a per-lane offset of 0 or 8 bytes returns (near pair, far pair), for LDS lanes and buffer lanes alike.  Every pass runs the short tail.

The replay is tests/dev/sched_sim.py's (the kernel's phase policy, thr 36 / refill 20) on the oracle's step traces, with the entry cull
applied (tests/cull_model.py) and per step: LDS-resident or not (TLAS interior nodes; the single BLAS's breadth-first top), and the octant of
the level's direction.  Issue costs: the two classes of profiles/r03_valu_probe3.txt (full rate 2.8 cycles, the rest 4.55).

    python tests/dev/signed_slab_sim.py --workload c3 --waves 32
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402  (dev tool: allowed to use the oracle)
import cull_model as cm  # noqa: E402

K_INT, K_ENTRY, K_LEAF, K_EXIT, K_DONE = 0, 1, 2, 3, 4
FULL, SLOW = 2.8, 4.55  # cycles per wave-instruction of the two issue classes
LDS_PLANE_NODES = 310


def load_scene(cfg):
    o = po.Scene()
    for verts, meta in cfg["blas"]:
        o.add_blas(verts, meta)
    for b, xf, ids in cfg["instances"]:
        for x, i in zip(xf, ids):
            o.add_instance(b, x, int(i))
    o.build()
    return o


def bfs_top(nodes, n_leaves, k):
    """the first k internal nodes of a breadth-first walk from the root (child0 before child1): k_top_remap, rc_build.hip"""
    top, level = [], [1]
    while level and len(top) < k:
        nxt = []
        for i in level:
            if len(top) >= k:
                break
            top.append(i)
            nd = nodes[i - 1]
            for c in (int(nd["child0"]), int(nd["child1"])):
                if c < n_leaves:
                    nxt.append(c)
        level = nxt
    return set(top)


class Model:
    def __init__(self, o, cull=True):
        self.o = o
        self.tl = o.tlas_nodes
        self.inst = o.instances
        self.n_inst = len(self.inst)
        descs = o.blas_descs
        self.single_blas = len(descs) == 1
        room = LDS_PLANE_NODES - (self.n_inst - 1)
        self.top = set()
        if self.single_blas and room > 0:
            n_leaves = len(o.blas_prims)
            self.top = bfs_top(o.blas_nodes, n_leaves, min(room, n_leaves - 1))
        self.spheres = cm.instance_spheres(self.inst, descs, cm.blas_radii(descs, o.blas_prims)) if cull else None
        self.inv = [i["inv_transform"].astype(np.float64).reshape(3, 4)[:, :3] for i in self.inst]

    def steps(self, ray, mode):
        """kernel-side step list of one ray: rows (kind, lds, octant)"""
        ev, dp, nd, ct = self.o.trace_steps(ray, mode)
        d = np.asarray(ray["d"], np.float64)
        d = np.where(d == 0, 0.0, d)
        o3 = np.asarray(ray["o"], np.float32)
        tmin = 0.0 if mode == "any" else float(ray["tmin"])

        def octant(v):
            return int(np.signbit(v[0])) | (int(np.signbit(v[1])) << 1) | (int(np.signbit(v[2])) << 2)
        w_oct = octant(d)
        out = []
        cur_oct, i, n = w_oct, 0, len(ev)
        while i < n:
            e = int(ev[i]); k = e & 7
            if k == 0:
                out.append((K_INT, 1, w_oct))
            elif k == 1:
                out.append((K_INT, 1 if int(nd[i]) in self.top else 0, cur_oct))
            elif k == 2:
                inst = int(self.tl[int(nd[i]) - 1]["child1"])
                out.append((K_ENTRY, 0, w_oct))
                if self.spheres is not None and cm.skip_entry(self.spheres[inst], o3, d.astype(np.float32), tmin, float(ct[i])):
                    i += 1
                    while i < n and not (int(ev[i]) & 0x80):  # the reference's fruitless visit of the instance: not executed
                        assert (int(ev[i]) & 7) in (1,), "a skipped entry reached a triangle test"
                        i += 1
                    i += 1
                    continue
                cur_oct = octant(self.inv[inst] @ d)
            else:
                out.append((K_LEAF, 0, cur_oct))
            if e & 0x80:
                out.append((K_EXIT, 0, w_oct))
                cur_oct = w_oct
            i += 1
        out.append((K_DONE, 0, 0))
        return np.array(out, np.int16)


def simulate(seq, thr=36, refill=20, buf_min=1, buf_wait=0):
    """One wave under the kernel's phase policy; returns the pass statistics of the interior loop + the phase counts."""
    nxt = 0
    cur = [None] * 64
    pos = np.zeros(64, np.int64)
    kind = np.full(64, 255, np.int16)
    lds = np.zeros(64, np.int16)
    octv = np.zeros(64, np.int16)
    st = dict(I=0, I_lanes=0, I_lds_lanes=0, all_lds=0, any_lds=0, any_buf=0, uni=0, uni_mixed=0, uni_buf=0, L=0, S=0, refill=0, outer=0, W=0)
    n_total = len(seq)

    def advance(mask):
        for l in np.nonzero(mask)[0]:
            pos[l] += 1
            kind[l], lds[l], octv[l] = cur[l][pos[l]]

    while True:
        st["outer"] += 1
        waited = 0
        while True:
            m = kind == K_INT
            n = int(m.sum())
            if n == 0:
                break
            # TD experiment (profiles/r05_td_model.txt): the texture path is charged per wave-INSTRUCTION, so a pass's buffer loads cost the same
            # for 1 lane as for 64.  Policy: buffer lanes sit a pass out unless >= buf_min of them are ready, nothing else is, or they have waited buf_wait passes
            nb_ready = int((m & (lds == 0)).sum())
            if buf_min > 1 and 0 < nb_ready < buf_min and nb_ready < n and waited < buf_wait:
                m = m & (lds == 1)
                n = int(m.sum())
                waited += 1
            else:
                waited = 0
            nl = int(lds[m].sum())
            uniform = len(set(octv[m].tolist())) == 1
            st["I"] += 1; st["I_lanes"] += n; st["I_lds_lanes"] += nl
            st["all_lds"] += nl == n; st["any_lds"] += nl > 0; st["any_buf"] += nl < n
            st["uni"] += uniform; st["uni_mixed"] += uniform and nl < n
            st["uni_buf"] += nl < n and len(set(octv[m & (lds == 0)].tolist())) == 1
            advance(m)
            live = int(((kind != 255) & (kind != K_DONE)).sum())
            thr_eff = thr if nxt < n_total else min(thr, max(live // 2, 1))
            if n < thr_eff:
                break
        m = kind == K_LEAF
        if m.any():
            st["L"] += 1
            advance(m)
        m = (kind == K_ENTRY) | (kind == K_EXIT)
        if m.any():
            st["S"] += 1
            advance(m)
        free = (kind == K_DONE) | (kind == 255)
        n_free = int(free.sum())
        can_refill = nxt < n_total
        if n_free == 64 and not can_refill:
            break
        if n_free >= refill or n_free == 64 or not can_refill:
            if (kind == K_DONE).any():
                st["W"] += 1
            kind[kind == K_DONE] = 255
            if can_refill:
                st["refill"] += 1
                for l in np.nonzero(kind == 255)[0]:
                    if nxt >= n_total:
                        break
                    cur[l] = seq[nxt]; nxt += 1
                    pos[l] = 0
                    kind[l], lds[l], octv[l] = cur[l][0]
    return st


def workload(name, res):
    import importlib.util
    spec = importlib.util.spec_from_file_location("scenes", os.path.join(ROOT, "raycore.jl_amd", "scenes.py"))
    sc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc)
    if name == "c2":
        cfg = sc.config_c2()
        o = load_scene(cfg)
        return o, o.ray_grid(cfg["viewdir"], cfg["grid"]), "closest"
    cfg = sc.config_c3()
    o = load_scene(cfg)
    rays = sc.c3_primary_rays(cfg, res, res)
    if name == "c3":
        return o, rays, "closest"
    hits = o.trace(rays, "closest", nthreads=8)
    if name == "shadow":
        sh = sc.c3_shadow_rays(cfg, rays, hits)
        return o, sh[hits["hit"] != 0] if len(sh) == len(rays) else sh, "any"
    if name == "c4":
        return o, sc.c4_bounce_rays(cfg, rays, hits, 4 * res * res), "closest"
    raise SystemExit("workload: c2 | c3 | shadow | c4")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--waves", type=int, default=24)
    ap.add_argument("--res", type=int, default=2048)
    args = ap.parse_args()
    o, rays, mode = workload(args.workload, args.res)
    md = Model(o)
    total_waves, pool = 6144, 128
    n_chunks = (len(rays) + pool - 1) // pool
    rng = np.random.default_rng(1)
    tot = None
    n_rays = 0
    for w in rng.choice(total_waves, args.waves, replace=False):
        seq, c = [], int(w)
        while c < n_chunks:
            for i in range(c * pool, min((c + 1) * pool, len(rays))):
                seq.append(md.steps(rays[i], mode))
            c += total_waves
        n_rays += len(seq)
        st = simulate(seq)
        tot = st if tot is None else {k: tot[k] + st[k] for k in st}
    I = tot["I"]
    print(f"{args.workload}: {args.waves} waves, {n_rays} rays ({mode}); interior passes {I} ({I / n_rays:.2f} per ray), {tot['I_lanes'] / I:.1f} lanes per pass")
    print(f"  interior VISITS served from LDS            {tot['I_lds_lanes'] / tot['I_lanes']:.3f}")
    print(f"  interior PASSES: every lane an LDS lane     {tot['all_lds'] / I:.3f}   some LDS lane {tot['any_lds'] / I:.3f}   some buffer lane {tot['any_buf'] / I:.3f}")
    print(f"  passes with ONE octant over all lanes       {tot['uni'] / I:.3f}   of the passes with a buffer lane: {tot['uni_mixed'] / max(tot['any_buf'], 1):.3f}"
          f"   (one octant over the buffer lanes only: {tot['uni_buf'] / max(tot['any_buf'], 1):.3f})")
    # issue cost of an interior pass, cycles.  Today (profiles/r04_isa_mix_kernel5.json, 50 instructions): 20 min/max + 12 packed + 14 other of the
    # slow class, 4 full-rate adds; the LDS / buffer fetch address is one slow-class instruction each.
    base = 20 * SLOW + 12 * SLOW + 14 * SLOW + 4 * FULL
    short = base - 12 * SLOW
    # A: LDS path pays 6 full-rate adds (5 more than today) whenever it runs; the short tail runs when all lanes are LDS lanes, or the octant is uniform
    #    (+ 2 slow instructions per pass to establish that: readfirstlane + compare)
    a_short = tot["all_lds"] + tot["uni_mixed"]
    cost_a = a_short * short + (I - a_short) * base + tot["any_lds"] * 5 * FULL + tot["any_buf"] * 2 * SLOW
    # B: every pass runs the short tail; the LDS path pays 6 full-rate adds instead of its one address instruction, the buffer path 4 full-rate
    #    (three ring offsets + the children's) instead of one
    cost_b = I * short + tot["any_lds"] * (6 * FULL - SLOW) + tot["any_buf"] * (4 * FULL - SLOW)
    cost_0 = I * base
    other = tot["L"] * 67 + tot["S"] * 60 + tot["outer"] * 25 + tot["refill"] * 70 + tot["W"] * 25  # instructions of the other phases (r04 ISA mix), ~4 cycles each
    other_c = other * 4.0
    for name, c in (("today", cost_0), ("A  LDS offsets + octant variants", cost_a), ("B  ring loads on both paths", cost_b)):
        print(f"  {name:34s} interior issue cycles per ray {c / n_rays:8.1f}  ({c / cost_0 - 1:+.1%})   whole kernel {(c + other_c) / n_rays:8.1f} ({(c + other_c) / (cost_0 + other_c) - 1:+.1%})")


if __name__ == "__main__":
    main()
