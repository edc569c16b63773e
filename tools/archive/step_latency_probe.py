"""Dev tool: unloaded latency of one traversal step -- 64 copies of the longest ray of a sample, alone on the machine (one wave)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from oracle import pyoracle as po  # dev tool: the oracle only counts the steps
from tools.perf_probe import build, time_trace


def main():
    sc = rc.scenes
    for name, cfg, rays in (("C2", sc.config_c2(), None), ("C3", sc.config_c3(), None)):
        t = build(cfg)
        o = po.Scene()
        for verts, meta in cfg["blas"]:
            o.add_blas(verts, meta)
        for b, xf, ids in cfg["instances"]:
            for x, i in zip(xf, ids):
                o.add_instance(b, x, int(i))
        o.build()
        rays = rc.generate_ray_grid(t, cfg["viewdir"], cfg["grid"]) if name == "C2" else sc.c3_primary_rays(cfg, 512, 512)
        idx = np.random.default_rng(0).choice(len(rays), 30000, replace=False)
        steps = np.array([len(o.trace_events(rays[i])[0]) for i in idx])
        for q in (100, 50):
            k = idx[np.argsort(steps)[int((len(steps) - 1) * q / 100)]]
            n_steps = len(o.trace_events(rays[k])[0])
            for copies in (64, 64 * 12, 64 * 24 * 256):
                batch = np.repeat(rays[k:k + 1], copies)
                for kern in (5, 0):
                    t.set_option("kernel", kern)
                    ms, _ = time_trace(t, batch, "closest", 5)
                    print(f"{name}: ray with {n_steps} steps (percentile {q}) x {copies:6d} copies, kernel {kern}: {ms * 1e3:8.1f} us = {ms * 1e3 / n_steps:.3f} us per step")
        t.set_option("kernel", -1)


if __name__ == "__main__":
    main()
