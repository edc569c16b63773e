"""Dev tool: BLAS build time vs the merge-sort / Onesweep switch point of the key sort."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raycore_jl_amd as rc

for n in (1000, 5000, 20000, 65536, 100_000, 250_000, 1_000_000, 1_048_577, 4_000_000):
    verts = rc.scenes.random_triangles(n, 42, edge=0.01)
    d_verts = torch.from_numpy(verts).cuda()
    out = []
    ref = None
    for mode, lim in (("merge", 1 << 30), ("onesweep", 0)):
        t = rc.TLAS(0)
        t.set_option("onesweep_min", lim)
        ms = []
        for _ in range(8):
            b = t.add_geometry_device(d_verts.data_ptr(), n)
            torch.cuda.synchronize()
            ms.append(t.last_kernel_ms())
        t.push_instances(b)
        nodes = t.adapt().all_blas_nodes[-(2 * n - 1):].tobytes()  # the last build's tree
        if ref is None:
            ref = nodes
        else:
            assert ref == nodes, "trees differ"
        out.append(f"{mode} {min(ms):.3f}")
        t.free()
    print(f"n={n:8d}  " + "  ".join(out), flush=True)
