import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import raycore_jl_amd as rc
sc = rc.scenes
cfg = sc.config_c3(lattice=(3, 3, 2))
t = rc.TLAS(0)
for v, m in cfg["blas"]: t.add_geometry(v, m)
for b, xf, ids in cfg["instances"]: t.push_instances(b, xf, ids)
t.sync()
rays = sc.c3_primary_rays(cfg, 512, 512)
n = len(rays)
dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
dh = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
dsh = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
docc = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
light = np.array([10, 10, 10], np.float32)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    # warm-up on the capture stream: first-use allocations (spill region, counters) must not happen during capture
    t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=s.cuda_stream)
    t.shadow_rays_device(dr.data_ptr(), dh.data_ptr(), n, light, dsh.data_ptr(), stream=s.cuda_stream)
    t.trace_device(dsh.data_ptr(), docc.data_ptr(), n, mode="any", stream=s.cuda_stream)
torch.cuda.synchronize()
want_h, want_o = dh.cpu().numpy().copy(), docc.cpu().numpy().copy()
dh.zero_(); docc.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
    t.shadow_rays_device(dr.data_ptr(), dh.data_ptr(), n, light, dsh.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    t.trace_device(dsh.data_ptr(), docc.data_ptr(), n, mode="any", stream=torch.cuda.current_stream().cuda_stream)
for rep in range(5):
    dh.zero_(); docc.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(dh.cpu().numpy(), want_h) and np.array_equal(docc.cpu().numpy(), want_o), rep
print("graph replay ok, drift", t.get_option("claim_drift"))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); print("replay: %.1f us per 3-launch frame" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for _ in range(200):
    t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=s.cuda_stream)
    t.shadow_rays_device(dr.data_ptr(), dh.data_ptr(), n, light, dsh.data_ptr(), stream=s.cuda_stream)
    t.trace_device(dsh.data_ptr(), docc.data_ptr(), n, mode="any", stream=s.cuda_stream)
torch.cuda.synchronize(); print("eager: %.1f us per 3-launch frame" % ((time.perf_counter() - t0) / 200 * 1e6))
