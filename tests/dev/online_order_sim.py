"""Dev tool (CPU only; under tests/ because it drives the oracle): can a claim order LEARNED INSIDE a launch help the FIRST launch of a batch?
(VERDICT r5 'next' #2.)  A timeline model of one launch of the persistent trace kernel, fed with the reference algorithm's per-ray step
counts from the instrumented oracle:

* 6 144 waves (256 CUs x 2 workgroups x 12 waves; 6 per SIMD), 64 lanes each; a lane holds one ray, a ray needs `steps` wave iterations
  (node visits + 2 per instance entry: entry and exit are switch passes), every iteration advances every live lane by one step;
* an iteration of a wave takes tau(k) = a + b k microseconds, k = waves still alive on its SIMD (profiles/r02_step_latency.txt: 0.42 us at 3,
  0.66 us at 6 waves per SIMD); the pair is scaled by ONE factor per workload so that the natural order reproduces the measured first-launch time;
* claims as in rc_claim_chunk: 16 shard counters, rotation per round, 128-ray chunks dealt whole, then in halves / quarters / eighths
  (taper 12); a wave refills its free lanes from its claimed range when >= 20 are free, claiming again when the range is used up.

Policies compared (same rays, same model):
  natural          chunk p of the claim order is chunk p                                       (what a first launch gets today)
  LPT              chunks sorted by their longest ray, longest first, everything known up front (the upper bound, measured on the GPU in round 3)
  late-LPT         the chunks the natural order hands out in the waves' FIRST claims stay where they are; only the rest -- everything a wave
                   can still influence once the first rays have reported -- is sorted by longest ray.  Upper bound of ANY in-launch scheme.
  promote(T, age)  neighbour promotion: at a refill, a wave that sees a finished ray whose lifetime was >= T x the running mean lifetime
                   (or, with age, a ray STILL in flight that old) pushes the unclaimed neighbours of that ray's chunk (previous / next chunk,
                   +- one image row; "ahead": the nearest unclaimed chunks of the same image column, however many rows on) onto a priority
                   list; a claim takes from the list before the shard counter.

    python tests/dev/online_order_sim.py            ->  profiles/r06_online_order.txt
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402  (dev tool: allowed to use the oracle)

W, LANES, SHARDS, POOL, REFILL, TAPER = 6144, 64, 16, 128, 20, 12
SIMDS = 1024


def claim_tables(n_items):
    """g1, g2, g3, c1, c2, c3, n_claims exactly as rc_claim_fill (rc_traverse.hip)."""
    n_base = (n_items + POOL - 1) // POOL
    chunks, done = [n_base, 0, 0, 0], 0
    for k in range(3):
        keep = TAPER * (POOL >> k) * W // 8
        upto = (n_items - keep) // POOL if n_items > keep else 0
        chunks[k] = max(upto - done, 0)
        done += chunks[k]
    chunks[3] = n_base - done
    g1, g2, g3 = chunks[0], chunks[0] + 2 * chunks[1], chunks[0] + 2 * chunks[1] + 4 * chunks[2]
    c1, c2, c3 = chunks[0], chunks[0] + chunks[1], chunks[0] + chunks[1] + chunks[2]
    return g1, g2, g3, c1, c2, c3, g3 + 8 * chunks[3], n_base


def claim_to_range(v, tabs, order, n_items):
    g1, g2, g3, c1, c2, c3, n_claims, n_base = tabs
    if v < g1:
        pos, part, shift = v, 0, 0
    elif v < g2:
        u = v - g1; pos, part, shift = c1 + (u >> 1), u & 1, 1
    elif v < g3:
        u = v - g2; pos, part, shift = c2 + (u >> 2), u & 3, 2
    else:
        u = v - g3; pos, part, shift = c3 + (u >> 3), u & 7, 3
    chunk = int(order[pos]) if order is not None else pos
    size = POOL >> shift
    a = chunk * POOL + part * size
    b = min(a + size, n_items)
    return min(a, b), b, chunk, pos


def simulate(cost, order=None, promote=None, row_chunks=0, tau=(0.18, 0.08), dt=0.05, scale=1.0, collect=False):
    """Returns the launch's end time in microseconds (and statistics).  promote = dict(T=..., age=bool, both_rows=bool) or None."""
    n = len(cost)
    tabs = claim_tables(n)
    n_claims, n_base = tabs[6], tabs[7]
    rem = np.zeros((W, LANES), np.int32)
    life = np.zeros((W, LANES), np.int32)          # lifetime so far of the lane's ray
    lane_ray = np.full((W, LANES), -1, np.int64)
    pool_next = np.zeros(W, np.int64); pool_end = np.zeros(W, np.int64)
    exhausted = np.zeros(W, bool)
    alive = np.ones(W, bool)
    prog = np.zeros(W)
    simd = (np.arange(W) // 12 % 256) * 4 + (np.arange(W) % 12) % 4
    shard_of = np.arange(W) & (SHARDS - 1)
    counter = np.zeros(SHARDS, np.int64)
    full_rounds, remc = n_claims >> 4, n_claims & 15
    my_chunks = np.array([full_rounds + (1 if ((s + full_rounds * 5) & 15) < remc else 0) for s in range(SHARDS)])
    # neighbour promotion: a priority list of claim POSITIONS (whole chunks only: positions below c1), a taken flag per position
    taken = np.zeros(n_base, bool)
    part_taken = np.zeros(n_base, bool)              # a part of the chunk went out through the counters (the chunk can no longer be promoted whole)
    prio = []                                        # global list (a per-shard list in a kernel; the model is generous)
    in_prio = np.zeros(n_base, bool)
    fin_sum, fin_cnt = 0.0, 0
    stats = {"promoted": 0, "claimed_from_prio": 0, "skipped": 0}
    c1 = tabs[3]

    def next_range(w):
        """the wave's next claimed range, or None"""
        nonlocal prio
        if promote is not None:
            while prio:
                p = prio.pop()
                if not taken[p]:
                    taken[p] = True
                    stats["claimed_from_prio"] += 1
                    a = p * POOL
                    return a, min(a + POOL, n)
        s = shard_of[w]
        while True:
            cs = counter[s]; counter[s] += 1
            if cs >= my_chunks[s]:
                return None
            v = (cs << 4) + ((s + cs * 5) & 15)
            a, b, chunk, pos = claim_to_range(v, tabs, order, n)
            if promote is not None:
                if taken[pos] and not part_taken[pos]:   # promoted and claimed whole from the list: its parts are skipped
                    stats["skipped"] += 1
                    continue
                taken[pos] = True; part_taken[pos] = True
            if b > a:
                return a, b

    def refill(w):
        nonlocal fin_sum, fin_cnt
        free = np.nonzero(rem[w] == 0)[0]
        if promote is not None:
            done = free[lane_ray[w, free] >= 0]
            mean = fin_sum / fin_cnt if fin_cnt else 0.0
            if len(done):
                l = life[w, done]
                fin_sum += float(l.sum()); fin_cnt += len(l)
                hot = done[l >= promote["T"] * max(mean, 1.0)] if fin_cnt > 64 else []
                for ln in hot:
                    push_neighbours(int(lane_ray[w, ln]) // POOL)
            if promote.get("age") and fin_cnt > 64:
                liv = np.nonzero(rem[w] > 0)[0]
                for ln in liv[life[w, liv] >= promote["T"] * max(mean, 1.0)]:
                    push_neighbours(int(lane_ray[w, ln]) // POOL)
            lane_ray[w, free] = -1
        i = 0
        while i < len(free):
            if pool_next[w] == pool_end[w]:
                if exhausted[w]:
                    break
                r = next_range(w)
                if r is None:
                    exhausted[w] = True
                    break
                pool_next[w], pool_end[w] = r
            k = int(min(len(free) - i, pool_end[w] - pool_next[w]))
            ids = np.arange(pool_next[w], pool_next[w] + k)
            rem[w, free[i:i + k]] = cost[ids]
            life[w, free[i:i + k]] = 0
            lane_ray[w, free[i:i + k]] = ids
            pool_next[w] += k
            i += k

    def push_neighbours(chunk):
        cand = [chunk - 1, chunk + 1]
        if row_chunks:
            cand += [chunk + row_chunks, chunk - row_chunks]
            if promote.get("both_rows"):
                cand += [chunk + 2 * row_chunks, chunk + row_chunks - 1, chunk + row_chunks + 1]
            if promote.get("ahead"):   # the nearest chunks of the SAME IMAGE COLUMN that nobody has claimed yet, however many rows ahead
                c, found = chunk + row_chunks, 0
                while c < n_base and found < promote["ahead"]:
                    if not taken[c]:
                        cand.append(c); found += 1
                    c += row_chunks
        for c in cand:
            if 0 <= c < n_base and not taken[c] and not in_prio[c]:
                in_prio[c] = True
                prio.append(c)
                stats["promoted"] += 1

    for w in range(W):
        refill(w)
    t = 0.0
    a_, b_ = tau[0] * scale, tau[1] * scale
    first_dry = None
    while alive.any():
        k = np.bincount(simd[alive], minlength=SIMDS)
        prog[alive] += dt / (a_ + b_ * k[simd[alive]])
        step = np.nonzero(alive & (prog >= 1.0))[0]
        t += dt
        if len(step) == 0:
            continue
        prog[step] -= 1.0
        n_live = (rem[step] > 0).sum(axis=1)
        can_refill = ~(exhausted[step] & (pool_next[step] == pool_end[step]))
        do_refill = (can_refill & (LANES - n_live >= REFILL)) | (n_live == 0)
        adv = step[~do_refill]
        live_mask = rem[adv] > 0
        rem[adv] -= live_mask
        life[adv] += live_mask
        for w in step[do_refill]:
            if not can_refill[np.searchsorted(step, w)]:
                alive[w] = False   # nothing live, nothing to claim
                continue
            refill(w)
            if exhausted[w] and first_dry is None:
                first_dry = t
            if (rem[w] > 0).sum() == 0 and exhausted[w] and pool_next[w] == pool_end[w]:
                alive[w] = False
    return (t, first_dry, stats) if collect else t


def workload(name):
    import raycore_jl_amd as rc
    sc = rc.scenes
    from helpers import build_oracle
    if name == "c2":
        cfg = sc.config_c2()
        o = build_oracle(po, cfg)
        rays = o.ray_grid(cfg["viewdir"], cfg["grid"])
        mode, row, measured = "closest", 1000 / POOL, 0.405   # natural-order first launch: profiles/r05_bench.json c2 first launch 2.47 Grays/s
    elif name == "r1m":
        verts = sc.random_triangles(1_000_000, 42, edge=0.01)
        cfg = {"blas": [(verts, None)], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
        o = build_oracle(po, cfg)
        rays = o.ray_grid((0.3, 0.2, 1.0), 1000)
        mode, row, measured = "closest", 1000 / POOL, 0.418
    elif name == "c3_1mi":
        cfg = sc.config_c3()
        o = build_oracle(po, cfg)
        rays = sc.c3_primary_rays(cfg, 1024, 1024)
        mode, row, measured = "closest", 1024 / POOL, 0.250
    elif name == "shadow":
        cfg = sc.config_c3()
        o = build_oracle(po, cfg)
        prim = sc.c3_primary_rays(cfg, 2048, 2048)
        rays = sc.c3_shadow_rays(cfg, prim, o.trace(prim, nthreads=8))
        mode, row, measured = "any", 0, 0.372
    else:
        raise SystemExit(name)
    _, cnt = o.trace(rays, mode=mode, nthreads=8, counters=True)
    cost = (cnt[:, 0].astype(np.int64) + 2 * cnt[:, 1].astype(np.int64)).astype(np.int32)
    cost = np.maximum(cost, 1)
    return cost, int(round(row)), measured


READING = """# Reading (VERDICT r5 'next' #2: "build only if the simulation recovers >= 40 % of the LPT gain").
# * The model is trustworthy where it can be checked: the LPT gain it predicts for C2 (+10.6 %) is the one measured on the GPU in round 3 with the chunks physically
#   reordered (0.402 -> 0.363 ms, +10.7 %, profiles/r03_lpt_upper_bound.txt); for the shadow batch and C3 at 1 Mi rays it predicts +12 % / +10 % where +19 % / +21 % were
#   measured -- it under-states gains there, it does not invent them.
# * Neighbour promotion recovers 0 %.  A 1 M-ray launch has 7 813 chunks for 6 144 waves and deals all of them in parts (taper 12): the waves' FIRST claims take chunks
#   0 .. 3 071, the first wave is out of work at 102-139 us of 250-418, and by then every previous / next chunk and every chunk one image row up or down of a chunk
#   that has reported a long ray (or holds one that is already old) has been claimed: "promoted 0" in every variant.  The shadow batch (16 230 chunks) promotes 1.
# * Looking further -- the nearest UNCLAIMED chunks of the same image column, which at that point lie tens to hundreds of rows below -- promotes 1 500-4 600 chunks and LOSES
#   8-23 %: cost is not coherent over that distance, and a promoted chunk is claimed whole where the taper would have dealt it in eighths.
# * The bound of ANY in-launch scheme (late-LPT: perfect knowledge of every chunk still unclaimed after the first round) is +5.3 % C2, +4.4 % random 1 M, +6.1 % shadow,
#   +13 % C3 1 Mi: about half of the LPT gain, because the other half is long rays that sat in first-round chunks and could only have been started earlier by knowing them
#   before the launch.  A predictor good enough to collect that half needs samples spread over the whole image in the first round -- i.e. a strided / interleaved first round,
#   which by itself costs 5-6 % on these batches (profiles/r05_chunk_order_probe.txt: no static permutation beats the natural order) -- or a pre-pass, which costs the
#   15-25 us it saves (round 4).  The box-count predictor of round 4 (rank correlation 0.94 / 0.66 / 0.60 / 0.23 with the true chunk cost on C2 / random 1 M / C3 1 Mi /
#   shadow) applied to the late chunks only would net about +4 / +2.5 / +7 / +1 % before its own cost.
# * Not built.  The first-launch targets of the verdict (C2 >= 2.65, random 1 M >= 2.55, shadow >= 5.9 Grays/s) are not reachable by reordering claims inside the launch;
#   what a first launch of a mid-size batch can still use is overlap with OTHER launches (rc_trace_*_device_batches, round 6: +21 % for four C2-sized batches).
#
# Appendix (tests/dev/coarse_first_round_sim.py, same model): the one scheme that WOULD give a first launch knowledge about unclaimed chunks -- a strided first round
# (every s-th image row, so that every later chunk has a first-round neighbour <= s/2 rows away in its column) followed by claims in the order of the neighbours'
# running cost estimates, idealised (one global priority queue, estimates exact at every refill, whole chunks):
#   C3 2048 x 2048 (natural, taper 12: 504.9 us; LPT 460.4, +9.7 %; late-LPT +9.3 %):  s = 6: 486.2 us (+3.8 %), s = 8: 481.7 (+4.8 %), s = 16: 495.0 (+2.0 %);
#                                                                                      with a PERFECT predictor behind the strided round: +5.3 / +6.8 / +6.4 %
#   C3 1024 x 1024 (natural, taper 12: 226.0 us; LPT 204.6, +10.5 %):                  s = 2 / 8 / 16: 259.1 / 250.5 / 250.6 us (-13 / -10 / -10 %); perfect predictor -1 ... +1 %
# For the 4 Mi-ray batch the idealised scheme is worth +4-5 % before the 1-2.4 % a strided claim order costs in cache locality (profiles/r05_chunk_order_probe.txt:
# "16-chunk groups, stride" -2.4 %, random chunks -1.7 % on this batch) -- net +2-3 % for a global queue no kernel has; for 1 M-ray batches it loses the taper and 10 %.  Not built.
# And the round-4 a-priori predictor (boxes of the TLAS the chunk's middle ray passes; rank correlation 0.61 with the true chunk cost on C3's 4 Mi primaries), applied in the
# same model: late chunks ordered by it 497.2 us (+1.5 %), all chunks ordered by it 502.8 (+0.4 %) against 504.9 natural -- where perfect knowledge gives +9.3 / +9.7 %.  An
# in-launch predictor pass is not worth its own cost either.
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c2,r1m,c3_1mi,shadow")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_online_order.txt"))
    args = ap.parse_args()
    lines = [__doc__.split("\n\n")[0].replace("\n", " "), ""]
    for name in args.workloads.split(","):
        t0 = time.time()
        cost, row, measured = workload(name)
        n = len(cost)
        n_base = (n + POOL - 1) // POOL
        cmax = np.array([cost[c * POOL:(c + 1) * POOL].max() for c in range(n_base)])
        # calibrate: one scale factor so that the natural order lands on the measured first-launch time
        t_nat = simulate(cost)
        scale = measured * 1e3 / t_nat
        t_nat, dry, _ = simulate(cost, scale=scale, collect=True)
        lpt = np.argsort(-cmax, kind="stable")
        t_lpt = simulate(cost, order=lpt, scale=scale)
        # the position of the claim order that the waves' first claims (claims 0 .. W - 1) reach
        tabs = claim_tables(n)
        first_round = min(claim_to_range(min(W, tabs[6]) - 1, tabs, None, n)[3] + 1, n_base)
        late = np.arange(n_base)
        late[first_round:] = first_round + np.argsort(-cmax[first_round:], kind="stable")
        t_late = simulate(cost, order=late, scale=scale)
        head = (f"== {name}: {n} rays, {n_base} chunks for {W} waves ({n_base / W:.2f} per wave; the waves' first claims take chunks 0 .. {first_round - 1}), steps per ray mean {cost.mean():.1f} p50 {np.median(cost):.0f} "
                f"p99 {np.percentile(cost, 99):.0f} max {cost.max()}; scale {scale:.3f}; natural: first wave out of work at {dry:.0f} us of {t_nat:.0f}")
        print(head, flush=True)
        lines.append(head)
        gain_lpt = t_nat - t_lpt

        def row_out(label, t, extra=""):
            rec = (t_nat - t) / gain_lpt * 100 if gain_lpt > 0 else 0.0
            s = f"   {label:58s} {t:7.1f} us  {n / t:8.1f} Mrays/s  {100 * (t_nat / t - 1):+6.1f} %   recovers {rec:5.1f} % of the LPT gain{extra}"
            print(s, flush=True)
            lines.append(s)

        row_out("natural (calibrated to the measured first launch)", t_nat)
        row_out("LPT, all chunks known up front", t_lpt)
        row_out("late-LPT: only chunks after the first round sorted (bound)", t_late)
        for T, age, both, ahead in ((2.0, False, False, 0), (1.5, True, True, 0), (2.0, True, False, 1), (1.5, True, False, 2), (3.0, True, False, 2)):
            t, _, st = simulate(cost, promote={"T": T, "age": age, "both_rows": both, "ahead": ahead}, row_chunks=row, scale=scale, collect=True)
            row_out(f"promote T={T} x mean{' +age' if age else ''}{' +wide' if both else ''}{f' +{ahead} ahead in the column' if ahead else ''}", t,
                    f"  (promoted {st['promoted']}, claimed from the list {st['claimed_from_prio']}, skipped {st['skipped']})")
        lines.append(f"   ({time.time() - t0:.0f} s)")
        lines.append("")
    open(args.out, "w").write("\n".join(lines) + "\n" + (READING if args.workloads == "c2,r1m,c3_1mi,shadow" else ""))   # (the reading belongs to the full table)


if __name__ == "__main__":
    main()
