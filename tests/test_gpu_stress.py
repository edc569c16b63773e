"""Lifecycle stress tests restated from the reference's test/test_tlas_stress.jl against the C ABI (GPU)."""
import numpy as np
import pytest

from helpers import assert_hits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0
    return raycore_jl_amd


def xlat(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = [x, y, z]
    return m


def test_200_blases_alive_then_delete_every_other(rc):  # test/test_tlas_stress.jl:187-227
    t = rc.TLAS()
    n_blas, spacing = 200, 4.0
    handles = []
    for i in range(1, n_blas + 1):
        n = 4 + (i % 6)
        handles.append(t.push(rc.scenes.fan_sphere(2 * n, n, radius=1.0), xlat(i * spacing, 0, 0)))
    t.sync()
    assert t.n_geometries() == n_blas and t.n_instances() == n_blas
    o = np.array([[i * spacing, 0, 5] for i in range(1, n_blas + 1)], dtype=np.float32)
    h = t.trace(rc.scenes.make_rays(o, [0, 0, -1]))
    assert np.all(h["hit"] == 1) and np.all(np.abs(h["t"] - 4.0) < 0.1)
    for i, hd in enumerate(handles, start=1):
        if i % 2 == 0:
            assert t.delete(hd)
    t.sync()
    assert t.n_geometries() == n_blas // 2 and t.n_instances() == n_blas // 2
    h2 = t.trace(rc.scenes.make_rays(o, [0, 0, -1]))
    assert list(h2["hit"]) == [i % 2 for i in range(1, n_blas + 1)]
    # compaction: blas_index values are dense 1..n_geometries and instance order follows handle order
    inst = t.adapt().instances
    assert sorted(set(inst["blas_index"].tolist())) == list(range(1, n_blas // 2 + 1))


def test_5000_instances_refit_keeps_topology(rc, oracle):  # test/test_tlas_stress.jl:233-273
    sc = rc.scenes
    g = sc.rng(9)
    pos = g.uniform(-50, 50, size=(5000, 3))
    xf = np.tile(sc.IDENTITY3x4, (5000, 1)).astype(np.float32)
    xf[:, [3, 7, 11]] = pos
    t = rc.TLAS()
    h = t.push(sc.fan_sphere(8, 5, radius=0.4), xf)
    t.sync()
    assert t.n_instances() == 5000 and len(t.adapt().nodes) == 9999
    topo = t.adapt().nodes[["child0", "child1", "parent"]].copy()
    for it in range(3):
        xf[:, 3] += 0.25
        t.update_transforms(h, xf)
        t.sync()
        assert t.last_sync_action == "refit"
        assert np.array_equal(t.adapt().nodes[["child0", "child1", "parent"]], topo)
    o = oracle.Scene()
    b = o.add_blas(sc.fan_sphere(8, 5, radius=0.4))
    for x in xf:
        o.add_instance(b, x, 0)
    o.build()
    # a refit tree has the old topology but the new boxes: hits must equal a fresh build's (closest hit is unique here)
    rays = sc.make_rays(pos[:2000] + [0, 0, 5] + [0.75, 0, 0], [0, 0, -1])
    got, want = t.trace(rays), o.trace(rays, nthreads=8)
    assert np.array_equal(got["hit"], want["hit"]) and got["hit"].mean() > 0.9
    m = got["hit"] == 1
    assert np.array_equal(got["t"][m].view(np.uint32), want["t"][m].view(np.uint32))
    assert np.array_equal(got["instance_id"][m], want["instance_id"][m])


def test_random_churn_matches_model(rc, oracle):  # test/test_tlas_stress.jl:101-181 (invariants), checked against the oracle
    sc = rc.scenes
    g = sc.rng(2024)
    meshes = [sc.fan_sphere(8, 5, radius=0.5), sc.random_triangles(60, 3, lo=-0.5, hi=0.5, edge=0.3), sc.fan_sphere(12, 7, radius=0.3)]
    t = rc.TLAS()
    live = {}  # handle -> (mesh_idx, xforms, ids)
    deleted = []
    for op in range(160):
        r = g.uniform()
        if r < 0.45 or not live:
            k = int(g.integers(1, 4))
            xf = np.tile(sc.IDENTITY3x4, (k, 1)).astype(np.float32)
            xf[:, [3, 7, 11]] = g.uniform(-4, 4, size=(k, 3))
            mi = int(g.integers(0, len(meshes)))
            ids = g.integers(0, 1000, size=k).astype(np.uint32)
            h = t.push(meshes[mi], xf, instance_ids=ids)
            live[h] = [mi, xf, ids]
        elif r < 0.65:
            h = list(live)[int(g.integers(0, len(live)))]
            assert t.delete(h) and not t.delete(h)
            deleted.append(h)
            del live[h]
        elif r < 0.85:
            h = list(live)[int(g.integers(0, len(live)))]
            xf = live[h][1].copy()
            xf[:, [3, 7, 11]] += g.uniform(-0.5, 0.5, size=(len(xf), 3)).astype(np.float32)
            t.update_transforms(h, xf)
            live[h][1] = xf
        else:
            h = list(live)[int(g.integers(0, len(live)))]
            mi = int(g.integers(0, len(meshes)))
            t.update(h, meshes[mi])
            live[h][0] = mi
        if op % 8 == 7 or op == 159:
            t.sync()
            assert t.sync().last_sync_action == "noop"
            assert t.n_instances() == sum(len(v[1]) for v in live.values()) == t.n_total_instances()
            assert all(t.is_valid(h) for h in live) and not any(t.is_valid(h) for h in deleted)
            for h in deleted[-3:]:
                with pytest.raises(rc.RaycoreError):
                    t.update_transforms(h, np.eye(4, dtype=np.float32)[None])
            # model: surviving handles in ascending id order; one BLAS per live handle that still references it
            o = oracle.Scene()
            for h in sorted(live, key=lambda x: x.id):
                mi, xf, ids = live[h]
                b = o.add_blas(meshes[mi])
                for x, i in zip(xf, ids):
                    o.add_instance(b, x, int(i))
            o.build()
            if live:
                st = t.adapt()
                assert st.instances[["instance_id", "transform", "inv_transform"]].tobytes() == o.instances[["instance_id", "transform", "inv_transform"]].tobytes()
                wb = o.world_bound
                rays = sc.make_rays(g.uniform(wb[:3] - 1, wb[3:] + 1, size=(3000, 3)), sc.normalize(g.normal(size=(3000, 3))))
                got, want = t.trace(rays), o.trace(rays, nthreads=4)
                for f in ("hit", "instance_id", "instance_custom_index"):
                    assert np.array_equal(got[f], want[f]), (op, f)
                m = got["hit"] == 1
                assert np.array_equal(got["t"][m].view(np.uint32), want["t"][m].view(np.uint32))
            else:
                assert t.n_geometries() == 0
                assert not t.trace(sc.make_rays([[0, 0, 5]], [0, 0, -1]))["hit"][0]


def test_grow_and_shrink(rc):  # test/test_tlas_stress.jl:517-548
    t = rc.TLAS()
    base = rc.scenes.fan_sphere(6, 4, radius=0.4)
    hs = []
    for it in range(60):
        hs.append(t.push(base, xlat(it, 0, 0)))
        if it % 3 == 2:
            t.delete(hs.pop(0))
        t.sync()
        assert t.n_instances() == len(hs) == t.n_geometries()
    o = np.array([[t.get_instance(h)["transform"][3], 0, 5] for h in hs], dtype=np.float32)
    assert np.all(t.trace(rc.scenes.make_rays(o, [0, 0, -1]))["hit"] == 1)


def test_mesh_swap_keeps_analytic_depth(rc):  # test/test_mesh_update.jl:96-116: sphere t = 4 - z_off after mesh swaps
    t = rc.TLAS()
    h = t.push(rc.scenes.fan_sphere(32, 17, centre=(0, 0, 0), radius=1.0))
    for z_off in (0.0, 0.5, -1.0, 2.0, 0.25):
        t.update(h, rc.scenes.fan_sphere(32, 17, centre=(0, 0, z_off), radius=1.0))
        assert t.sync().last_sync_action == "rebuild"
        hit, _, dist, _, inst = rc.closest_hit(t, rc.Ray((0.0, 0.0, 5.0), (0, 0, -1)))
        assert hit and inst == 1 and abs(dist - (4.0 - z_off)) < 0.05
    # refit path (test/test_mesh_update.jl:184-227): move the instance instead of the mesh
    for dz in (0.5, 1.5):
        t.update_transform(h, xlat(0, 0, dz))
        assert t.sync().last_sync_action == "refit"
        hit, _, dist, _, _ = rc.closest_hit(t, rc.Ray((0.0, 0.0, 5.0), (0, 0, -1)))
        assert hit and abs(dist - (4.0 - 0.25 - dz)) < 0.05


def test_contract_counters(rc):  # test/test_abstract_accel_contract.jl:22-34
    t, hs = rc.TLAS_from_meshes([rc.scenes.fan_sphere(8, 5), rc.scenes.fan_sphere(8, 5, centre=(3, 0, 0))])
    assert t.n_instances() == 2 and t.n_geometries() == 2
    wb = t.world_bound()
    assert np.all(wb.p_min < wb.p_max) and wb.p_max[0] > 3.0
    assert t.wait_for_gpu() is t


def test_c_client_example_runs(rc, tmp_path):
    """examples/trace_quad.c through the C ABI on the GPU (the reference's KAT shape: t = 1, 2, miss, 3; instance ids 0/1)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "trace_quad"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "trace_quad.c"),
                           "-L", os.path.dirname(rc.LIB_PATH), "-lraycore_mi355x", "-Wl,-rpath," + os.path.dirname(rc.LIB_PATH), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert out[0].startswith("ray 0: hit=1 t=1 ") and "instance=0 custom=11" in out[0]
    assert out[1].startswith("ray 1: hit=1 t=2 ") and "instance=1 custom=22" in out[1]
    assert out[2].startswith("ray 2: hit=0")
    assert out[3].startswith("ray 3: hit=1 t=3 ") and "instance=0" in out[3]


def test_instance_buffer_device_refit(rc):
    """instance_buffer + refit_tlas! without a host round trip (src/Raycore.jl:117-128): descriptors rewritten in device memory
    through a torch view of the library's buffer give the same arrays and hits as update_transforms! + sync! from the host."""
    import torch
    sc = rc.scenes
    verts = sc.fan_sphere(10, 6, radius=0.5)
    xf0, _, _ = sc.lattice_transforms(4, 4, 2, 1.6, 3)
    xf1, _, _ = sc.lattice_transforms(4, 4, 2, 1.9, 4)
    n = len(xf0)
    a, b = rc.TLAS(), rc.TLAS()
    ha = a.push(verts, xf0.reshape(n, 12), instance_ids=np.arange(n, dtype=np.uint32))
    hb = b.push(verts, xf0.reshape(n, 12), instance_ids=np.arange(n, dtype=np.uint32))
    a.sync(); b.sync()
    # host path
    a.update_transforms(ha, xf1.reshape(n, 12))
    a.sync()
    assert a.last_sync_action == "refit"
    # device path: alias the 108-byte records as 27 floats each and overwrite the transform words [2, 14)
    ptr, cnt = b.instance_buffer(hb)
    assert cnt == n

    class Alias:
        __cuda_array_interface__ = {"shape": (n, 27), "typestr": "<f4", "data": (ptr, False), "version": 2}

    recs = torch.as_tensor(Alias(), device="cuda")
    recs[:, 2:14] = torch.from_numpy(xf1.reshape(n, 12)).cuda()
    torch.cuda.synchronize()
    b.refit_device(recompute_inverse=True)
    sa, sb = a.adapt(), b.adapt()
    assert sa.nodes.tobytes() == sb.nodes.tobytes()
    assert sa.instances.tobytes() == sb.instances.tobytes()      # host mirror pulled back lazily, inverses recomputed on the device
    assert np.array_equal(a.world_bound().p_min, b.world_bound().p_min) and np.array_equal(a.world_bound().p_max, b.world_bound().p_max)
    g = np.random.default_rng(5)
    rays = sc.make_rays(g.uniform(-2, 8, (30000, 3)), sc.normalize(g.normal(size=(30000, 3))))
    ha_, hb_ = a.trace(rays), b.trace(rays)
    assert ha_.tobytes() == hb_.tobytes() and ha_["hit"].sum() > 100
    got = b.get_instances(hb)
    assert np.array_equal(got["transform"], xf1.reshape(n, 12))
    # a later host-side mutation starts from the refreshed mirror
    b.update_transform(b.push(verts[:4]), np.eye(4, dtype=np.float32))
    b.sync()
    assert np.array_equal(b.get_instances(hb)["transform"], xf1.reshape(n, 12))


def test_overlapping_launches_on_two_streams(rc):
    """Launches of one scene on different streams may overlap; each stream has its own stack spill region and every launch its own
    counter slot, so results equal the serial ones even when the stacks run through the spill path (deep chain scene)."""
    import torch
    def chain(levels, fat=0.3):
        tris = []
        for j in range(1, levels + 1):
            for axis in range(3):
                size = 2.0 ** (-j + 1)
                p = np.full(3, fat * size); p[axis] = size
                q = p.copy(); q[(axis + 1) % 3] += 0.5 * size * fat
                r = np.full(3, -1e-4 * (1 + 0.5 * j))
                tris.append(np.concatenate([r, p, q]))
        return np.array(tris, dtype=np.float32)
    sc = rc.scenes
    xf = np.tile(sc.IDENTITY3x4, (4, 1)).astype(np.float32)
    xf[1, [0, 5, 10]] = 0.5
    xf[2, [0, 5, 10]] = 0.25
    xf[3, [3, 7, 11]] = [0.01, 0.0, 0.0]
    t = rc.TLAS()
    t.push(chain(10), xf, instance_ids=np.arange(4, dtype=np.uint32))
    t.sync()
    g = sc.rng(21)
    n = 1_500_000
    sets = []
    for k in range(2):
        rays = sc.make_rays(g.uniform(-0.2, 0.0, size=(n, 3)), sc.normalize(g.uniform(0.05, 1.0, size=(n, 3))))
        sets.append((rays, t.trace(rays)))  # serial reference results
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    d_rays = [torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda() for r, _ in sets]
    d_hits = [torch.zeros(n * 32, dtype=torch.uint8, device="cuda") for _ in sets]
    torch.cuda.synchronize()
    for rep in range(6):
        for k in (0, 1):
            t.trace_device(d_rays[k].data_ptr(), d_hits[k].data_ptr(), n, stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    for k in (0, 1):
        got = d_hits[k].cpu().numpy().view(rc.HIT_DT)
        assert got.tobytes() == sets[k][1].tobytes(), f"stream {k}"


def test_large_host_batches_are_chunked_and_identical(rc):
    """Host-buffer trace calls of >= 3 Mi rays run as an upload / trace / download pipeline over 512 Ki-ray chunks (rc_capi.hip);
    the chunked result, the single-launch result and the device-buffer result must be the same bytes, with and without `out=`."""
    import torch
    sc = rc.scenes
    cfg = sc.config_c3()
    t = rc.TLAS(0)
    for verts, meta in cfg["blas"]:
        t.add_geometry(verts, meta)
    for b, xf, ids in cfg["instances"]:
        t.push_instances(b, xf, ids)
    t.sync()
    rays = sc.c3_primary_rays(cfg, 1800, 1800)  # 3.24 M rays: seven chunks, the last one partial
    assert len(rays) >= 3 * (1 << 20) and len(rays) % (1 << 19) != 0
    for mode in ("closest", "any"):
        t.set_option("host_pipeline", 1)
        chunked = t.trace(rays, mode=mode)
        assert t.last_kernel_ms() > 0
        reuse = np.full(len(rays), 0xAB, dtype=np.uint8).repeat(32).view(rc.HIT_DT)
        assert t.trace(rays, mode=mode, out=reuse) is reuse
        t.set_option("host_pipeline", 0)
        single = t.trace(rays, mode=mode)
        d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
        d_hits = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
        t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), mode=mode)
        torch.cuda.synchronize()
        dev = d_hits.cpu().numpy().view(rc.HIT_DT)
        assert chunked.tobytes() == single.tobytes() == dev.tobytes() == reuse.tobytes(), mode
        assert 0 < int(chunked["hit"].sum()) < len(rays)
    with pytest.raises(ValueError):
        t.trace(rays, out=np.zeros(5, dtype=rc.HIT_DT))
    t.free()
