"""dev: read back the entry-cull spheres of C3 and compare with the geometry (numpy, float64)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import raycore_jl_amd as rc
from helpers import build_product
cfg = rc.scenes.config_c3()
t = build_product(rc, cfg)
n = t.n_instances() if hasattr(t, "n_instances") else 256
ptr = t.get_option("debug_inst_cull_ptr")
hip = ctypes.CDLL("libamdhip64.so")
buf = np.zeros((256, 8), np.float32)
rcode = hip.hipMemcpy(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(ptr), buf.nbytes, 2)
print("hipMemcpy", rcode)
b, xf, ids = cfg["instances"][0]
xf = np.asarray(xf, dtype=np.float64).reshape(-1, 3, 4)
s = np.array([np.linalg.svd(x[:, :3], compute_uv=False).max() for x in xf]); c = xf[:, :, 3]
print("first records:\n", buf[:4])
print("centre error max", np.abs(buf[:, :3] - c).max())
print("A / s: min", (buf[:, 3] / s).min(), "max", (buf[:, 3] / s).max(), " B / s", (buf[:, 4] / s).min(), (buf[:, 4] / s).max())
