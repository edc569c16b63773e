"""Dev tool (CPU only): the schedule simulator with leaf tests / instance entries pooled across the workgroup and answered after a delay."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'dev'))
import numpy as np
import sched_sim as ss
import importlib.util
spec = importlib.util.spec_from_file_location("scenes", os.path.join(ROOT, 'raycore.jl_amd', 'scenes.py'))
sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
K_INT, K_ENTRY, K_LEAF, K_EXIT, K_DONE = 0, 1, 2, 3, 4

def simulate_coop(seq, delay, thr=36, refill=20, C=ss.C, c_post=12, c_collect=18, c_serve_leaf=105/64, c_serve_entry=110/64, c_exit=14):
    """Leaf tests and instance entries are posted to a workgroup ring and served by whichever wave has 64 of them: the owner pays
    post + collect at its phase points, the service (full lane fill) is charged per request, and the lane is blocked for `delay`
    further outer iterations.  Exits stay with the owner (cheap)."""
    nxt = 0
    cur = [None]*64
    pos = np.zeros(64, np.int64)
    kind = np.full(64, 255, np.uint8)
    wait = np.zeros(64, np.int64)   # > 0: request in flight, counts outer iterations
    valu = 0.0; I = 0; Il = 0; outer = 0
    n_total = len(seq)
    def advance(mask):
        for l in np.nonzero(mask)[0]:
            pos[l] += 1; kind[l] = cur[l][pos[l]]
    while True:
        outer += 1; valu += C["outer"]
        while True:
            m = (kind == K_INT) & (wait == 0)
            n = int(m.sum())
            if n == 0: break
            I += 1; Il += n; valu += C["I"]
            advance(m)
            live = int(((kind != 255) & (kind != K_DONE)).sum())
            thr_eff = thr if nxt < n_total else min(thr, max(live // 2, 1))
            if n < thr_eff: break
        # collect finished requests
        done = (wait == 1)
        if done.any():
            valu += c_collect
            advance(done)
        wait[wait > 0] -= 1
        # post new requests
        m = ((kind == K_LEAF) | (kind == K_ENTRY)) & (wait == 0)
        if m.any():
            valu += c_post
            valu += c_serve_leaf * int(((kind == K_LEAF) & m).sum()) + c_serve_entry * int(((kind == K_ENTRY) & m).sum())
            wait[m] = delay + 1
        m = (kind == K_EXIT) & (wait == 0)
        if m.any():
            valu += c_exit; advance(m)
        valu += C["Fchk"]
        free = (kind == K_DONE) | (kind == 255)
        n_free = int(free.sum())
        can_refill = nxt < n_total
        if n_free == 64 and not can_refill: break
        if n_free >= refill or n_free == 64 or not can_refill:
            kind[kind == K_DONE] = 255
            if can_refill:
                valu += C["refill"]
                for l in np.nonzero(kind == 255)[0]:
                    if nxt >= n_total: break
                    cur[l] = seq[nxt]; nxt += 1; pos[l] = 0; kind[l] = cur[l][0]; wait[l] = 0
    return dict(valu=valu, I=I, Il=Il, outer=outer)

cfg = sc.config_c3(); o = ss.load_scene(cfg)
rays = sc.c3_primary_rays(cfg, 2048, 2048)
waves = ss.sample_streams(o, rays, 24, 6144, 128, "closest")
n_rays = sum(len(w) for w in waves)
tot = None
for w in waves:
    r = ss.simulate(w); tot = r if tot is None else {k: tot[k]+r[k] for k in r}
print("current: VALU/ray %.1f, interior iterations %d x%.1f, outer %d" % (tot["valu"]/n_rays, tot["I"], tot["I_lanes"]/tot["I"], tot["outer"]))
for d in (0, 1, 2, 3):
    t = dict(valu=0, I=0, Il=0, outer=0)
    for w in waves:
        r = simulate_coop(w, d); t = {k: t[k]+r[k] for k in t}
    print("pooled leaf/entry, served after %d further outer iterations: VALU/ray %.1f, interior iterations %d x%.1f, outer %d" % (d, t["valu"]/n_rays, t["I"], t["Il"]/t["I"], t["outer"]))
