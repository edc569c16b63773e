"""Why does the CPU baseline (the oracle, C restatement of the reference algorithm) not scale with the GPU box's 256 hardware threads?
Rate of the C3 primary batch at 1 .. N threads (preallocated result, pinned pool), next to what the container is allowed to use:
the cgroup CPU quota and the throttling counters before / after.  python3 tools/cpu_scaling_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raycore_jl_amd as rc
from oracle import pyoracle as po

def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None
def cgroup():
    out = {}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.cpus"):
        v = read(p)
        if v is not None:
            out[p] = v.replace("\n", "; ")
    return out
print("os.cpu_count", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)), "loadavg", read("/proc/loadavg"))
for k, v in cgroup().items():
    print(k, "=", v)
sc = rc.scenes
cfg = sc.config_c3()
o = po.Scene()
for v, m in cfg["blas"]: o.add_blas(v, m)
for b, xf, ids in cfg["instances"]:
    for x, i in zip(xf, ids): o.add_instance(b, x, int(i))
o.build()
rays = sc.c3_primary_rays(cfg, 2048, 2048)
n = len(rays)
out = np.zeros(n, dtype=rc.HIT_DT)
po.pool_pin(True)
o.trace(rays, nthreads=po.allowed_cpus(), out=out)
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if nt > po.allowed_cpus():
        break
    sub = rays if nt >= 8 else rays[:: 8 // nt * 2]
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter(); o.trace(sub, nthreads=nt, out=out[:len(sub)]); best = min(best, time.perf_counter() - t0)
    print(f"threads {nt:4d}: {len(sub) / best / 1e6:8.2f} Mrays/s   ({len(sub) / best / 1e6 / nt:.3f} per thread, {best:.3f} s)", flush=True)
for k, v in cgroup().items():
    if "stat" in k:
        print("after:", k, "=", v)
