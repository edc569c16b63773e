"""End-to-end cost of view_factors as the API returns it (a HOST N x N UInt32 matrix, src/kernels.jl:74-78) at BASELINE config C5:
rc_view_factors with a fresh / reused / registered matrix, two scenes on one device in ROWS mode, the RCCL (RAYS) path on one device,
chunk-size sweep.  python3 tools/vf_e2e_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raycore_jl_amd as rc

sc = rc.scenes
cfg = sc.config_c5()
def build():
    t = rc.TLAS(0)
    t.add_geometry(*cfg["blas"][0])
    t.push_instances(1, cfg["instances"][0][1], cfg["instances"][0][2])
    return t.sync()
t = build()
n, rpt = t.n_primitives(), cfg["rays_per_triangle"]
gb = 4 * n * n / 1e9
print(f"C5: {n} triangles, {rpt} rays each = {n * rpt / 1e6:.1f} M rays, matrix {gb:.2f} GB; PCIe floor at 56 GB/s = {gb / 56:.3f} s", flush=True)
t0 = time.perf_counter(); out = np.empty((n, n), np.uint32, order="F"); rc.view_factors(t, rpt, 7, out=out); dt = time.perf_counter() - t0
print(f"rc_view_factors, fresh matrix (pages faulted in inside the call): {dt:.3f} s   device pipeline {t.last_kernel_ms():.1f} ms   counted {int(out.sum(dtype=np.int64))}", flush=True)
ref_sum, ref_chk = out.sum(axis=1, dtype=np.int64), int((out[:, ::97].astype(np.int64) * np.arange(1, n + 1)[:, None]).sum())
for rep in range(3):
    t0 = time.perf_counter(); rc.view_factors(t, rpt, 7, out=out); dt = time.perf_counter() - t0
    print(f"rc_view_factors, reused matrix: {dt:.3f} s = {gb / dt:.1f} GB/s into host memory, {n * rpt / dt / 1e6:.0f} Mrays/s   device pipeline {t.last_kernel_ms():.1f} ms", flush=True)
t.host_register(out)
for rep in range(2):
    t0 = time.perf_counter(); rc.view_factors(t, rpt, 7, out=out); dt = time.perf_counter() - t0
    print(f"rc_view_factors, reused + registered matrix: {dt:.3f} s = {gb / dt:.1f} GB/s", flush=True)
for mb in (32, 64, 128, 384, 768):
    t.set_option("vf_chunk_bytes", mb << 20)
    t0 = time.perf_counter(); rc.view_factors(t, rpt, 7, out=out); dt = time.perf_counter() - t0
    print(f"  chunk {mb} MB: {dt:.3f} s", flush=True)
t.set_option("vf_chunk_bytes", 192 << 20)
t2 = build()
out[:] = 0
t0 = time.perf_counter(); rc.view_factors_multi([t, t2], rpt, 7, mode="rows", out=out); dt = time.perf_counter() - t0
ok = np.array_equal(out.sum(axis=1, dtype=np.int64), ref_sum) and int((out[:, ::97].astype(np.int64) * np.arange(1, n + 1)[:, None]).sum()) == ref_chk
print(f"rc_view_factors_multi ROWS, two scenes on device 0 (one PCIe link): {dt:.3f} s   same matrix: {ok}", flush=True)
t2.free()
out[:] = 0
t0 = time.perf_counter(); rc.view_factors_multi([t], rpt, 7, mode="rays", out=out); dt = time.perf_counter() - t0
ok = np.array_equal(out.sum(axis=1, dtype=np.int64), ref_sum) and int((out[:, ::97].astype(np.int64) * np.arange(1, n + 1)[:, None]).sum()) == ref_chk
print(f"rc_view_factors_multi RAYS (RCCL ncclReduce, one rank; includes ncclCommInitAll + the 10 GB accumulator): {dt:.3f} s   same matrix: {ok}", flush=True)
t0 = time.perf_counter(); rc.view_factors_multi([t], rpt, 7, mode="rays", out=out); dt = time.perf_counter() - t0
print(f"rc_view_factors_multi RAYS again (communicator cached): {dt:.3f} s", flush=True)
t.host_unregister(out)
from raycore_jl_amd import distributed as rd
t0 = time.perf_counter(); m = rd.view_factors_host_matrix(t, rpt, 7); dt = time.perf_counter() - t0
print(f"distributed.view_factors_host_matrix, one rank (creates + faults in a /dev/shm matrix): {dt:.3f} s   same: {np.array_equal(np.asarray(m).sum(axis=1, dtype=np.int64), ref_sum)}", flush=True)
