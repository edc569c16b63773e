import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes
import raycore_jl_amd as rc
sc = rc.scenes
cfg = sc.config_c3()
t = rc.TLAS(0)
for v, m in cfg["blas"]: t.add_geometry(v, m)
for b, xf, ids in cfg["instances"]: t.push_instances(b, xf, ids)
t.sync()
rays = sc.c3_primary_rays(cfg, 2048, 2048); n = len(rays)
d = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda(); out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream()
def b2b(rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(rounds): t.trace_device(d.data_ptr(), out.data_ptr(), n, stream=stream.cuda_stream)
    e1.record(stream); e1.synchronize()
    return e0.elapsed_time(e1) / rounds
def hdr():
    h = torch.zeros(48, dtype=torch.int32, device="cuda")
    ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(192), 3)
    h = h.cpu().numpy(); return dict(sel=int(h[0]), valid=int(h[1]), fresh=int(h[4]), rec=int(h[5]), skip=int(h[38]), gen=[int(x) for x in h[12:16]])
for k in range(6):
    ms = b2b(20); print("cost_order=1 round", k, f"{ms:.4f} ms {n/ms/1e3:.0f} Mrays/s", hdr(), flush=True)
t.set_option("cost_order", 0)
for k in range(3):
    ms = b2b(20); print("cost_order=0 round", k, f"{ms:.4f} ms {n/ms/1e3:.0f} Mrays/s", flush=True)
t.set_option("cost_order", 1)
for k in range(3):
    ms = b2b(20); print("cost_order=1 again", k, f"{ms:.4f} ms {n/ms/1e3:.0f} Mrays/s", hdr(), flush=True)
ms = t.recent_kernel_ms(20); print("kernel-only ms of the last 20:", [round(x, 4) for x in ms])
