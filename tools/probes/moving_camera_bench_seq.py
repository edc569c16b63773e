"""dev: bench.py's c3_moving_camera sequence with per-run times and the history's launch clock (which launches took part in the mechanism)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raycore_jl_amd as rc
sc = rc.scenes
cfg = sc.config_c3()
t = rc.TLAS(0)
for verts, meta in cfg["blas"]:
    t.add_geometry(verts, meta)
for b, xf, ids in cfg["instances"]:
    t.push_instances(b, xf, ids)
t.sync()
res = 2048; n = res * res
stream = torch.cuda.Stream()
hip = ctypes.CDLL("libamdhip64.so")
def clock():
    h = torch.empty(48, dtype=torch.int32, device="cuda")
    hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(192), 3)
    torch.cuda.synchronize()
    w = h.cpu().numpy().view(np.uint32)
    return int(w[3]), int(w[36])
rays = sc.c3_primary_rays(cfg, res, res)
d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
dh = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
with torch.cuda.stream(stream):
    for _ in range(30):   # the headline: one buffer repeated
        t.trace_device(d_rays.data_ptr(), dh.data_ptr(), n, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    print("after the headline: clock, streak =", clock())
    frames = [torch.from_numpy(sc.pinhole_rays(res, res, cfg["eye"] + np.array([0.02 * k, 0.01 * k, 0.0]), cfg["lattice_centre"], 45.0).view(np.uint8).reshape(-1)).cuda() for k in range(16)]
    def b2b(buffers):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for b in buffers:
            t.trace_device(b.data_ptr(), dh.data_ptr(), n, stream=stream.cuda_stream)
        e1.record(stream); e1.synchronize()
        return e0.elapsed_time(e1) / len(buffers)
    for f in frames:
        t.trace_device(f.data_ptr(), dh.data_ptr(), n, stream=stream.cuda_stream)
    for rep in range(4):
        on = b2b(frames); c_on = clock()
        t.set_option("cost_order", 0)
        off = b2b(frames); c_off = clock()
        t.set_option("cost_order", 1)
        print(f"run {rep}: on {on:.4f} ms (clock, streak after: {c_on})   off {off:.4f} ms ({c_off})", flush=True)
