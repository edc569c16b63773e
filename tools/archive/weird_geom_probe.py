import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import raycore_jl_amd as rc
from oracle import pyoracle as po
sc = rc.scenes
g = np.random.default_rng(0)
def case(name, verts):
    try:
        t = rc.TLAS(); t.push(verts); t.sync()
        o = po.Scene(); o.add_instance(o.add_blas(verts)); o.build()
        st = t.adapt()
        same_nodes = st.all_blas_nodes.tobytes() == o.blas_nodes.tobytes()
        same_tlas = st.nodes.tobytes() == o.tlas_nodes.tobytes()
        rays = sc.make_rays(g.uniform(-2, 2, (2000, 3)), sc.normalize(g.normal(size=(2000, 3))))
        a, b = t.trace(rays), o.trace(rays)
        same_hits = np.array_equal(a["hit"], b["hit"]) and np.array_equal(a["primitive_id"], b["primitive_id"])
        print(f"{name:28s} blas nodes same={same_nodes} tlas same={same_tlas} hits same={same_hits} nhit={int(a['hit'].sum())}")
    except Exception as e:
        print(f"{name:28s} EXC {type(e).__name__}: {e}")
base = sc.random_triangles(500, 3, lo=-1, hi=1, edge=0.3)
v = base.copy(); v[7, 0] = np.nan; case("one NaN coordinate", v)
v = base.copy(); v[7, 4] = np.inf; case("one +Inf coordinate", v)
v = base.copy(); v[7, 4] = -np.inf; v[9, 2] = np.inf; case("+-Inf coordinates", v)
v = base.copy(); v[:, 2::3] = 0.25; case("coplanar (z const)", v)
v = np.tile(base[:1], (300, 1)); case("300 identical triangles", v)
v = base.copy() * np.float32(1e30); case("huge 1e30", v)
v = base.copy() * np.float32(1e-30); case("tiny 1e-30", v)
v = base.copy(); v[3] = np.float32(3e38); case("near-FLT_MAX triangle", v)
