"""Dev tool: does overlapping consecutive batches on two streams hide the fill/drain of the persistent waves?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from perf_probe import build


def main():
    sc = rc.scenes
    cfg = sc.config_c3()
    t = build(cfg)
    for res in (1024, 1448, 2048):
        rays = sc.c3_primary_rays(cfg, res, res)
        n = len(rays)
        d_r = [torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda() for _ in range(2)]
        d_h = [torch.empty(n * 32, dtype=torch.uint8, device="cuda") for _ in range(2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for kern in (5, 3):
            t.set_option("kernel", kern)
            for n_streams in (1, 2):
                for _ in range(5):
                    t.trace_device(d_r[0].data_ptr(), d_h[0].data_ptr(), n, stream=streams[0].cuda_stream)
                torch.cuda.synchronize()
                k = 40
                t0 = time.perf_counter()
                for i in range(k):
                    j = i % n_streams
                    t.trace_device(d_r[j].data_ptr(), d_h[j].data_ptr(), n, stream=streams[j].cuda_stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(f"{res}x{res} kernel {kern} streams {n_streams}: {1e3 * dt / k:.3f} ms/batch  {n * k / dt / 1e6:.0f} Mrays/s", flush=True)


if __name__ == "__main__":
    main()
