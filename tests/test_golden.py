"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle): the oracle must keep
reproducing them (CPU), and the HIP path must match them bit for bit (GPU)."""
import os

import numpy as np
import pytest

from helpers import assert_hits_equal, build_oracle, build_product

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    return raycore_jl_amd


def cfg_c1(rc):
    return rc.scenes.config_c1()


def cfg_instanced(rc, g):
    sc = rc.scenes
    xf = g["xforms"]
    return {"blas": [(sc.fan_sphere(10, 6), None), (sc.random_triangles(200, 8, lo=-0.5, hi=0.5, edge=0.2), None)],
            "instances": [(1, xf[:7], np.arange(7, dtype=np.uint32) + 1), (2, xf[7:], np.arange(5, dtype=np.uint32) + 50)]}


def test_scene_generators_are_stable(rc):
    g = np.load(os.path.join(HERE, "instanced_small.npz"))
    xf, _, _ = rc.scenes.lattice_transforms(3, 2, 2, 1.3, 123)
    assert xf.tobytes() == g["xforms"].tobytes()  # Philox-seeded numpy generators: identical on every machine


def test_oracle_reproduces_golden(rc, oracle):
    g = np.load(os.path.join(HERE, "c1_sphere.npz"))
    cfg = cfg_c1(rc)
    o = build_oracle(oracle, cfg)
    assert o.ray_grid(cfg["viewdir"], cfg["grid"]).tobytes() == g["rays"].tobytes()
    assert o.blas_nodes.tobytes() == g["blas_nodes"].tobytes() and o.tlas_nodes.tobytes() == g["tlas_nodes"].tobytes()
    assert_hits_equal(o.trace(g["rays"]), g["closest"], "golden C1 closest")
    assert_hits_equal(o.trace(g["rays"], mode="any"), g["any"], "golden C1 any")
    assert np.array_equal(o.get_illumination(cfg["viewdir"], cfg["grid"]), g["illumination"])
    assert o.blas4_nodes(1).tobytes() == g["blas4_nodes"].tobytes() and o.triangles.tobytes() == g["triangles"].tobytes()
    assert_hits_equal(o.trace4(1, g["rays"]), g["closest4"], "golden C1 closest4")
    assert_hits_equal(o.trace4(1, g["rays"], mode="any"), g["any4"], "golden C1 any4")
    g2 = np.load(os.path.join(HERE, "instanced_small.npz"))
    o2 = build_oracle(oracle, cfg_instanced(rc, g2))
    assert o2.instances.tobytes() == g2["instances"].tobytes() and o2.tlas_nodes.tobytes() == g2["tlas_nodes"].tobytes()
    assert_hits_equal(o2.trace(g2["rays"]), g2["closest"], "golden instanced closest")
    assert_hits_equal(o2.trace(g2["rays"], mode="any"), g2["any"], "golden instanced any")
    assert np.array_equal(o2.view_factors(16, seed=5), g2["view_factors_16"])
    assert np.array_equal(o2.collide_instances()[0], g2["contacts"]) and o2.blas4_nodes(2).tobytes() == g2["blas4_nodes_2"].tobytes()


@pytest.mark.gpu
def test_hip_path_matches_golden(rc):
    g = np.load(os.path.join(HERE, "c1_sphere.npz"))
    cfg = cfg_c1(rc)
    t = build_product(rc, cfg)
    st = t.adapt()
    assert st.all_blas_nodes.tobytes() == g["blas_nodes"].tobytes() and st.nodes.tobytes() == g["tlas_nodes"].tobytes()
    assert_hits_equal(t.trace(g["rays"]), g["closest"], "golden C1 closest")
    assert_hits_equal(t.trace(g["rays"], mode="any"), g["any"], "golden C1 any")
    assert np.array_equal(rc.get_illumination(t, cfg["viewdir"], cfg["grid"]), g["illumination"])
    g2 = np.load(os.path.join(HERE, "instanced_small.npz"))
    t2 = build_product(rc, cfg_instanced(rc, g2))
    assert t2.adapt().instances.tobytes() == g2["instances"].tobytes()
    assert_hits_equal(t2.trace(g2["rays"]), g2["closest"], "golden instanced closest")
    assert_hits_equal(t2.trace(g2["rays"], mode="any"), g2["any"], "golden instanced any")
    assert np.array_equal(rc.view_factors(t2, rays_per_triangle=16, seed=5), g2["view_factors_16"])
    # BVH4, full triangles, collision pairs
    assert st.all_blas_triangles.tobytes() == g["triangles"].tobytes()
    b4 = rc.build_blas4(*cfg["blas"][0])
    assert b4.nodes.tobytes() == g["blas4_nodes"].tobytes()
    assert_hits_equal(b4.trace(g["rays"]), g["closest4"], "golden C1 closest4")
    assert_hits_equal(b4.trace(g["rays"], mode="any"), g["any4"], "golden C1 any4")
    res = rc.collide_instances(t2)
    assert np.array_equal(np.stack([res.contacts["instance_a"], res.contacts["instance_b"]], axis=1), g2["contacts"])
    assert rc.build_blas4(*cfg_instanced(rc, g2)["blas"][1]).nodes.tobytes() == g2["blas4_nodes_2"].tobytes()
