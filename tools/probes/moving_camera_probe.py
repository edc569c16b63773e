"""dev: a camera that moves every frame, back-to-back launches between two events (what bench.py's c3_moving_camera measures with kernel
times, here including whatever the library enqueues around the kernels): cost_order 1 against 0, 16 ray buffers in rotation.
  python3 tools/probes/moving_camera_probe.py [step]      (eye shift per frame, default 0.02)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raycore_jl_amd as rc

step = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
sc = rc.scenes
cfg = sc.config_c3()
t = rc.TLAS(0)
for verts, meta in cfg["blas"]:
    t.add_geometry(verts, meta)
for b, xf, ids in cfg["instances"]:
    t.push_instances(b, xf, ids)
t.sync()
res = 2048
n = res * res
nf = 16
frames = [torch.from_numpy(sc.pinhole_rays(res, res, cfg["eye"] + np.array([step * k, 0.5 * step * k, 0.0]), cfg["lattice_centre"], 45.0).view(np.uint8).reshape(-1)).cuda() for k in range(nf)]
dh = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
for co in (0, 1, 0, 1):
    t.set_option("cost_order", co)
    for rep in range(2):   # warm: 32 frames (with cost_order 1: past the first pause decision)
        for f in frames:
            t.trace_device(f.data_ptr(), dh.data_ptr(), n)
    best, mean = 1e9, []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f in frames:
            t.trace_device(f.data_ptr(), dh.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / nf
        best = min(best, ms); mean.append(ms)
    print(f"step {step}: cost_order={co}  {best:.4f} ms best, {np.mean(mean):.4f} mean per frame   {n / np.mean(mean) / 1e3:.1f} Mrays/s", flush=True)
