"""dev: which library calls survive beside an open global-mode stream capture (torch.cuda.graph)?  One process per call: a capture that
was invalidated leaves torch's graph bookkeeping unusable."""
import os, sys
N_CASES = 9
if len(sys.argv) == 1:  # parent: no GPU use here
    import subprocess
    for i in range(N_CASES):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(i)])
    sys.exit(0)
CASE = int(sys.argv[1])
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import raycore_jl_amd as rc
from helpers import build_product

sc = rc.scenes
cfg = sc.config_c3(lattice=(3, 3, 2))
t = build_product(rc, cfg)
rays = sc.c3_primary_rays(cfg, 320, 200)
n = len(rays)
dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
dh = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=s.cuda_stream)
torch.cuda.synchronize()
small = sc.config_c3(lattice=(2, 1, 1))
big_batch = np.concatenate([rays] * 52)[:3_300_000]


KEEP = []


def attempt(name, fn):
    a, b = build_product(rc, small), build_product(rc, small)
    g = torch.cuda.CUDAGraph()
    KEEP.append(g)  # (a graph whose capture failed aborts the process in its destructor)
    try:
        with torch.cuda.graph(g, stream=s):
            t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
            fn(a, b)
            t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
        g.replay(); torch.cuda.synchronize()
        print(f"{name:40s} capture survived", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name:40s} INVALIDATED: {str(e)[:90]}", flush=True)
        try:
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001
            pass
    a.free(); b.free()


CASES = [
    ("nothing", lambda a, b: None),
    ("small host trace", lambda a, b: a.trace(rays[:1000])),
    ("pipelined host trace (3.3 M rays)", lambda a, b: a.trace(big_batch)),
    ("view_factors one scene", lambda a, b: rc.view_factors(a, 8, seed=3)),
    ("view_factors_multi two scenes", lambda a, b: rc.view_factors_multi([a, b], 8, seed=3)),
    ("get_illumination", lambda a, b: rc.get_illumination(a, (0, 0, 1), 64)),
    ("trace_multi two scenes", lambda a, b: rc.trace_multi([a, b], rays)),
    ("build + free a scene", lambda a, b: build_product(rc, small).free()),
    ("collide_instances", lambda a, b: rc.collide_instances(a)),
]
attempt(*CASES[CASE])
sys.stdout.flush()
os._exit(0)
