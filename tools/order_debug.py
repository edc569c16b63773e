import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from oracle import pyoracle as po
from tools.perf_probe import build, to_dev
from tools.lpt_probe import oracle_of
import ctypes
sc = rc.scenes
WHICH = sys.argv[1] if len(sys.argv) > 1 else "c2"
MODE = "closest"
if WHICH == "c2":
    cfg2 = sc.config_c2(); t = build(cfg2); o = oracle_of(cfg2)
    rays = rc.generate_ray_grid(t, cfg2["viewdir"], cfg2["grid"])
else:  # the C3 shadow rays (any_hit)
    cfg3 = sc.config_c3(); t = build(cfg3); o = oracle_of(cfg3)
    prim = sc.c3_primary_rays(cfg3, 2048, 2048)
    rays = sc.c3_shadow_rays(cfg3, prim, t.trace(prim))
    MODE = "any"
n = len(rays)
_, cnt = o.trace(rays, mode=MODE, nthreads=16, counters=True)
cost = cnt[:, 0].astype(np.int64) + 2 * cnt[:, 1]
nb = -(-n // 128)
true_cost = np.array([int(cost[k * 128:(k + 1) * 128].max()) for k in range(nb)], dtype=np.uint32)
d_r, d_h = to_dev(rays), torch.empty(n * 32, dtype=torch.uint8, device="cuda")
def read(ptr, count):
    out = torch.empty(count, dtype=torch.int32, device="cuda")
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(4 * count), 3)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)
def write(ptr, arr):
    src = torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).cuda()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(4 * len(arr)), 3)
    torch.cuda.synchronize()
learned_cost = None
for mode in ("learned", "true_through_the_machinery", "learned_costs_no_reporting"):
    t.set_option("cost_order", 1); t.set_option("taper", 12)
    t.set_option("cost_thr", 64)
    n_used = {"learned": n - 128, "true_through_the_machinery": n - 256}.get(mode, n - 384)  # another launch shape each: a fresh history
    for rep in range(8):
        if mode == "learned" and rep == 3:  # what launch 2 recorded (before launch 3's scatter clears it)
            learned_cost = read(t.get_option("debug_cost_ptr"), -(-n // 128)).copy()
        if mode != "learned" and rep > 0:  # overwrite what the previous launch recorded with the TRUE chunk costs, classed over their whole range
            BIAS = 100000   # the reporting threshold of the launch = the bottom of the scale: biased so that NO ray reports (the cost of reporting is not what is measured here)
            src = true_cost if mode == "true_through_the_machinery" else np.where(learned_cost > 0, learned_cost, 0).astype(np.uint32)
            write(t.get_option("debug_cost_ptr"), np.where(src[: -(-n_used // 128)] > 0, src[: -(-n_used // 128)] + BIAS, 0).astype(np.uint32))
            hdr = read(t.get_option("debug_ctl_ptr"), 32)
            hdr[16:20] = [BIAS, BIAS + int(src.max()), BIAS, BIAS + int(src.max())]
            write(t.get_option("debug_ctl_ptr"), hdr)
        t.trace_device(d_r.data_ptr(), d_h.data_ptr(), n_used, mode=MODE)
        torch.cuda.synchronize()
        nn = t.get_option("debug_order_n")
        order = read(t.get_option("debug_order_ptr"), nn)
        ctl = read(t.get_option("debug_ctl_ptr"), 32)[[0, 1, 2, 16, 17, 18, 19]]  # sel, order valid, reporting threshold, slot 0's (thr, top) x 2
        perm_ok = np.array_equal(np.sort(order), np.arange(nn)) if rep > 0 else None
        tc = true_cost[:nn][order] if rep > 0 else None
        corr = None
        if rep > 0:
            # how well does the claim order put the truly long chunks first?  mean true cost of the first 10 % vs all; rank correlation
            first = tc[: nn // 10].mean(); allm = tc.mean()
            ranks = np.empty(nn); ranks[np.argsort(-true_cost[:nn], kind="stable")] = np.arange(nn)
            corr = np.corrcoef(ranks[order], np.arange(nn))[0, 1]
            print(f"{mode} rep {rep}: {t.last_kernel_ms():.3f} ms  permutation {perm_ok}  ctl {ctl.tolist()}  mean true cost of the first tenth {first:.1f} vs all {allm:.1f}; rank correlation with the true order {corr:.3f}; first 12 chunks' true costs {tc[:12].tolist()}", flush=True)
        else:
            print(f"{mode} rep 0: {t.last_kernel_ms():.3f} ms  ctl {ctl.tolist()}", flush=True)
