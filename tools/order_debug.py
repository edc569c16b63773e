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
cfg2 = sc.config_c2(); t = build(cfg2); o = oracle_of(cfg2)
rays = rc.generate_ray_grid(t, cfg2["viewdir"], cfg2["grid"])
n = len(rays)
_, cnt = o.trace(rays, nthreads=16, counters=True)
cost = cnt[:, 0].astype(np.int64)
nb = -(-n // 128)
true_cost = np.array([int(cost[k * 128:(k + 1) * 128].max()) for k in range(nb)], dtype=np.uint32)
d_r, d_h = to_dev(rays), torch.empty(n * 32, dtype=torch.uint8, device="cuda")
def read(ptr, count):
    out = torch.empty(count, dtype=torch.int32, device="cuda")
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(4 * count), 3)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)
for mode in ("learned",):
    t.set_option("cost_order", 1); t.set_option("taper", 12)
    if mode == "true":
        d_cost = torch.from_numpy(true_cost.view(np.int32)).cuda()
        t.set_option("cost_thr", int(np.percentile(true_cost, 60))); t.set_option("debug_cost_ptr", d_cost.data_ptr())
        n_used = n
    else:
        t.set_option("cost_thr", 64)
        n_used = n - 128  # another launch shape: a fresh history
    for rep in range(6):
        t.trace_device(d_r.data_ptr(), d_h.data_ptr(), n_used)
        torch.cuda.synchronize()
        nn = t.get_option("debug_order_n")
        order = read(t.get_option("debug_order_ptr"), nn)
        ctl = read(t.get_option("debug_ctl_ptr"), 4)
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
