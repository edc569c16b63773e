"""Shared test helpers: build the same scene in the product (C ABI) and in the CPU oracle."""
import numpy as np


def build_product(rc, cfg, device=0):
    t = rc.TLAS(device)
    for verts, meta in cfg["blas"]:
        t.add_geometry(verts, meta)
    for b, xf, ids in cfg["instances"]:
        t.push_instances(b, xf, ids)
    return t.sync()


def build_oracle(po, cfg):
    s = po.Scene()
    for verts, meta in cfg["blas"]:
        s.add_blas(verts, meta)
    for b, xf, ids in cfg["instances"]:
        for x, i in zip(xf, ids):
            s.add_instance(b, x, int(i))
    return s.build()


def assert_hits_equal(got, want, what=""):
    """Bit-exact comparison of RTHitResult arrays: ids exact, t/u/v identical bit patterns (NaN payloads aside)."""
    assert len(got) == len(want)
    for f in ("hit", "primitive_id", "instance_id", "instance_custom_index"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert len(bad) == 0, f"{what}: {len(bad)} rays differ in {f}, first {bad[:5]}: got {got[f][bad[:5]]} want {want[f][bad[:5]]}"
    for f in ("t", "bary_u", "bary_v"):
        a, b = got[f].view(np.uint32), want[f].view(np.uint32)
        # a NaN must be a NaN on both sides; its sign / payload bits are the hardware's (x86 generates 0xFFC00000 for 0*inf,
        # gfx950 0x7FC00000), not the algorithm's
        bad = np.nonzero((a != b) & ~(np.isnan(got[f]) & np.isnan(want[f])))[0]
        assert len(bad) == 0, f"{what}: {len(bad)} rays differ in {f} bits, first {bad[:5]}: got {got[f][bad[:5]]} want {want[f][bad[:5]]}"


def random_rays(rc, n, seed, lo, hi):
    g = rc.scenes.rng(seed)
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    o = g.uniform(lo - 0.5 * (hi - lo), hi + 0.5 * (hi - lo), size=(n, 3))
    target = g.uniform(lo, hi, size=(n, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return rc.scenes.make_rays(o, d)
