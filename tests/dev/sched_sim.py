"""Dev tool (CPU only; lives under tests/ because it drives the oracle, which only test infrastructure may do): replay the reference algorithm's per-ray step sequences through wave-scheduling policies.

Per-ray results are fixed by the reference algorithm (bit-exact parity), so a ray's sequence of steps -- interior box tests,
leaf triangle tests, instance entries, returns to the top level -- is the same whatever the kernel does; only WHEN a lane's next
step runs relative to the other 63 lanes is the kernel's choice.  The trace kernels are bound by VALU issue (profiles/r02_*), so the
cost of a policy is the number of wave-level VALU instructions it issues: every phase a wave executes costs its full instruction
count however few lanes take part.  This tool samples waves' worth of rays from a BASELINE workload, records each ray's steps with
the oracle (rco_trace_events) and counts phase executions under the kernel's current policy and under candidates.

    python tests/dev/sched_sim.py --workload c3 --waves 48
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as po  # noqa: E402  (dev tool: allowed to use the oracle)

# phase costs in VALU wave-instructions, counted in the ISA of k_trace_phased_lds<false,768,16,6> (round 2)
C = {"I": 48, "L": 76, "S": 104, "Fchk": 25, "refill": 88, "outer": 6}

K_INT, K_ENTRY, K_LEAF, K_EXIT, K_DONE = 0, 1, 2, 3, 4


def load_scene(cfg):
    o = po.Scene()
    for verts, meta in cfg["blas"]:
        o.add_blas(verts, meta)
    for b, xf, ids in cfg["instances"]:
        for x, i in zip(xf, ids):
            o.add_instance(b, x, int(i))
    o.build()
    return o


def expand(ev):
    """oracle events -> kernel step kinds (an exit is its own step in the kernel: the switch phase)."""
    out = []
    for e in ev:
        k = e & 7
        out.append(K_INT if k <= 1 else (K_ENTRY if k == 2 else K_LEAF))
        if e & 0x80:
            out.append(K_EXIT)
    out.append(K_DONE)
    return np.array(out, np.uint8)


def sample_streams(o, rays, n_waves, total_waves, pool, mode):
    """Chunks the sampled waves would claim (wave w takes chunks w, w + total_waves, ...): per wave a list of ray step arrays."""
    n_chunks = (len(rays) + pool - 1) // pool
    waves = []
    rng = np.random.default_rng(1)
    picks = rng.choice(total_waves, n_waves, replace=False)
    for w in picks:
        seq = []
        c = int(w)
        while c < n_chunks:
            for i in range(c * pool, min((c + 1) * pool, len(rays))):
                ev, _ = o.trace_events(rays[i], mode)
                seq.append(expand(ev))
            c += total_waves
        waves.append(seq)
    return waves


def simulate(seq, thr=36, refill=20, leaf_min=1, switch_min=1, max_defer=0, cost=C):
    """One wave under the phased policy.  leaf_min / switch_min: run the leaf / switch phase only when that many lanes wait for it
    (or nothing else can run), at most max_defer outer iterations late.  Returns dict of counts."""
    nxt = 0                      # next ray of the wave's sequence
    cur = [None] * 64            # per lane: step array
    pos = np.zeros(64, np.int64)
    kind = np.full(64, 255, np.uint8)   # 255 = empty lane
    st = {"I": 0, "L": 0, "S": 0, "refill": 0, "outer": 0, "I_lanes": 0, "L_lanes": 0, "S_lanes": 0, "valu": 0}
    defer_l = defer_s = 0
    n_total = len(seq)

    def advance(mask):
        for l in np.nonzero(mask)[0]:
            pos[l] += 1
            kind[l] = cur[l][pos[l]]

    while True:
        st["outer"] += 1
        st["valu"] += cost["outer"]
        # interior loop
        while True:
            m = kind == K_INT
            n = int(m.sum())
            if n == 0:
                break
            st["I"] += 1; st["I_lanes"] += n; st["valu"] += cost["I"]
            advance(m)
            live = int(((kind != 255) & (kind != K_DONE)).sum())
            thr_eff = thr
            if nxt >= n_total:  # drain: threshold follows the lanes still alive
                thr_eff = min(thr, max(live // 2, 1))
            if n < thr_eff:
                break
        n_int = int((kind == K_INT).sum())
        # leaf phase
        m = kind == K_LEAF
        n = int(m.sum())
        if n and (n >= leaf_min or n_int == 0 or defer_l >= max_defer):
            st["L"] += 1; st["L_lanes"] += n; st["valu"] += cost["L"]
            advance(m)
            defer_l = 0
        elif n:
            defer_l += 1
        # switch phase
        m = (kind == K_ENTRY) | (kind == K_EXIT)
        n = int(m.sum())
        if n and (n >= switch_min or n_int == 0 or defer_s >= max_defer):
            st["S"] += 1; st["S_lanes"] += n; st["valu"] += cost["S"]
            advance(m)
            defer_s = 0
        elif n:
            defer_s += 1
        # finished lanes / refill
        st["valu"] += cost["Fchk"]
        free = (kind == K_DONE) | (kind == 255)
        n_free = int(free.sum())
        can_refill = nxt < n_total
        if n_free == 64 and not can_refill:
            break
        if n_free >= refill or n_free == 64 or not can_refill:
            kind[kind == K_DONE] = 255
            if can_refill:
                st["refill"] += 1; st["valu"] += cost["refill"]
                for l in np.nonzero(kind == 255)[0]:
                    if nxt >= n_total:
                        break
                    cur[l] = seq[nxt]; nxt += 1
                    pos[l] = 0
                    kind[l] = cur[l][0]
    return st


def simulate_pool(seq, pool=0, thr=36, refill=20, swap_cost=0, min_swap=1, cost=C):
    """One wave that holds 64 + `pool` rays: the extra ones are PARKED (in LDS: state + stack column), and a phase pass serves up to 64 of the
    resident rays that wait for that phase, wherever they are.  A pass that needs a parked ray first swaps it with a lane whose ray does not
    take part (one wave-wide swap pass of `swap_cost` VALU instructions per phase pass that needs any swap: ray re-read, safe_inv3, entry
    transform, the two states exchanged through LDS).  swap_cost = 0 is the upper bound of what a pool can give: VALU wave-instructions per ray
    at the higher lane fill, same per-ray step chains.  Policy otherwise as `simulate` (the kernel's): interior passes while >= thr rays wait for
    one, then one leaf pass, one switch pass, write-out and refill when >= refill slots are free."""
    R = 64 + pool
    nxt = 0
    cur = [None] * R
    pos = np.zeros(R, np.int64)
    kind = np.full(R, 255, np.uint8)
    in_lane = np.zeros(R, bool)
    in_lane[:64] = True
    st = {"I": 0, "L": 0, "S": 0, "refill": 0, "outer": 0, "I_lanes": 0, "L_lanes": 0, "S_lanes": 0, "valu": 0, "swaps": 0, "swap_passes": 0}
    n_total = len(seq)

    def run(mask, name):
        idx = np.nonzero(mask)[0]
        if len(idx) > 64:   # more waiting than lanes: the ones already in lanes first
            order = np.argsort(~in_lane[idx], kind="stable")
            idx = idx[order[:64]]
        parked = idx[~in_lane[idx]]
        if 0 < len(parked) < min_swap and len(parked) < len(idx):   # not worth a swap pass: the pass runs with the rays that sit in lanes (a pass of parked rays only must swap)
            idx = idx[in_lane[idx]]
            parked = idx[:0]
        if len(parked):
            chosen = np.zeros(R, bool); chosen[idx] = True
            out = np.nonzero(in_lane & ~chosen)[0][:len(parked)]   # lanes whose ray sits this pass out (there are enough: at most 64 take part)
            in_lane[out] = False
            in_lane[parked] = True
            st["swaps"] += len(parked); st["swap_passes"] += 1; st["valu"] += swap_cost
        st[name] += 1; st[name + "_lanes"] += len(idx); st["valu"] += cost[name]
        for l in idx:
            pos[l] += 1
            kind[l] = cur[l][pos[l]]
        return len(idx)

    while True:
        st["outer"] += 1
        st["valu"] += cost["outer"]
        while True:
            m = kind == K_INT
            n = int(m.sum())
            if n == 0:
                break
            served = run(m, "I")
            live = int(((kind != 255) & (kind != K_DONE)).sum())
            thr_eff = thr if nxt < n_total else min(thr, max(min(live, 64) // 2, 1))
            if int((kind == K_INT).sum()) < thr_eff and served <= n:
                break
        m = kind == K_LEAF
        if m.any():
            run(m, "L")
        m = (kind == K_ENTRY) | (kind == K_EXIT)
        if m.any():
            run(m, "S")
        st["valu"] += cost["Fchk"]
        free = (kind == K_DONE) | (kind == 255)
        n_free = int(free.sum())
        can_refill = nxt < n_total
        if n_free == R and not can_refill:
            break
        if n_free >= refill or n_free == R or not can_refill:
            kind[kind == K_DONE] = 255
            if can_refill:
                st["refill"] += 1; st["valu"] += cost["refill"]
                for l in np.nonzero(kind == 255)[0]:
                    if nxt >= n_total:
                        break
                    cur[l] = seq[nxt]; nxt += 1
                    pos[l] = 0
                    kind[l] = cur[l][0]
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--waves", type=int, default=24)
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--pool", action="store_true", help="the parked-ray pool bound instead of the phase-threshold variants")
    args = ap.parse_args()
    import importlib.util
    spec = importlib.util.spec_from_file_location("scenes", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "raycore.jl_amd", "scenes.py"))
    sc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc)
    if args.workload == "c3":
        cfg = sc.config_c3()
        o = load_scene(cfg)
        rays = sc.c3_primary_rays(cfg, args.res, args.res)
        mode = "closest"
    elif args.workload == "c2":
        cfg = sc.config_c2()
        o = load_scene(cfg)
        rays = o.ray_grid(cfg["viewdir"], cfg["grid"])
        mode = "closest"
    elif args.workload == "c4":
        cfg = sc.config_c3()
        o = load_scene(cfg)
        prim = sc.c3_primary_rays(cfg, 1024, 1024)
        rays = sc.c4_bounce_rays(cfg, prim, o.trace(prim, nthreads=8), 4 * len(prim))
        mode = "closest"
    else:
        raise SystemExit("workload: c3 | c2 | c4")
    total_waves = 6144
    waves = sample_streams(o, rays, args.waves, total_waves, 128, mode)
    n_rays = sum(len(w) for w in waves)
    steps = sum(len(r) - 1 for w in waves for r in w)
    print(f"{args.workload}: {args.waves} waves, {n_rays} rays, {steps / n_rays:.1f} steps per ray")
    variants = [
        ("current (thr 36, refill 20)", dict()),
        ("thr 24", dict(thr=24)),
        ("thr 48", dict(thr=48)),
        ("leaf>=8 switch>=8 defer<=2", dict(leaf_min=8, switch_min=8, max_defer=2)),
        ("leaf>=12 switch>=12 defer<=3", dict(leaf_min=12, switch_min=12, max_defer=3)),
        ("leaf>=16 switch>=16 defer<=4", dict(leaf_min=16, switch_min=16, max_defer=4)),
        ("leaf>=16 switch>=16 defer<=8, thr 24", dict(leaf_min=16, switch_min=16, max_defer=8, thr=24)),
        ("leaf>=24 switch>=24 defer<=16, thr 16", dict(leaf_min=24, switch_min=24, max_defer=16, thr=16)),
    ]
    if args.pool:
        # VERDICT r5 'next' #3b: the lane-fill bound of a parked-ray pool.  swap 0 = free swaps (upper bound); 70 = ray re-read + safe_inv3 + entry transform + state
        # exchange (~60 VALU + 7 ds_read_b64 + the writes), one wave-wide swap pass per phase pass that needs one.
        base = None
        for pool, swap, min_swap in ((0, 0, 1), (16, 0, 1), (32, 0, 1), (16, 70, 1), (32, 70, 1), (16, 70, 8), (32, 70, 8), (16, 70, 16), (32, 70, 16), (32, 70, 24), (32, 100, 16)):
            tot = None
            for w in waves:
                stp = simulate_pool(w, pool=pool, swap_cost=swap, min_swap=min_swap)
                tot = stp if tot is None else {k: tot[k] + stp[k] for k in stp}
            if base is None:
                base = tot["valu"]
            print(f"pool {pool:2d} (+{pool / 64:.2f} rays per lane), swap pass {swap:3d} VALU when >= {min_swap:2d} rays move: VALU/ray {tot['valu'] / n_rays:7.1f} ({100 * (base / tot['valu'] - 1):+5.1f} % rays per VALU instruction)"
                  f" | I {tot['I']:6d} x{tot['I_lanes'] / max(tot['I'], 1):4.1f} | L {tot['L']:5d} x{tot['L_lanes'] / max(tot['L'], 1):4.1f} | S {tot['S']:5d} x{tot['S_lanes'] / max(tot['S'], 1):4.1f}"
                  f" | swap passes {tot['swap_passes']} moving {tot['swaps']} rays | outer {tot['outer']} refills {tot['refill']}", flush=True)
        return
    for name, kw in variants:
        tot = None
        for w in waves:
            st = simulate(w, **kw)
            tot = st if tot is None else {k: tot[k] + st[k] for k in st}
        scale = (len(rays) / n_rays)
        print(f"{name:42s} VALU/ray {tot['valu'] / n_rays:7.1f} (x{scale * tot['valu'] / 1e6:6.1f} M per launch) | I {tot['I']:6d} x{tot['I_lanes'] / max(tot['I'], 1):4.1f}"
              f" | L {tot['L']:5d} x{tot['L_lanes'] / max(tot['L'], 1):4.1f} | S {tot['S']:5d} x{tot['S_lanes'] / max(tot['S'], 1):4.1f} | outer {tot['outer']} refills {tot['refill']}")


if __name__ == "__main__":
    main()
