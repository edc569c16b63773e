"""Seeded synthetic scenes and ray sets for BASELINE.json's five configs.

Pure numpy (Philox bit generator => identical on every machine).  Used by tests/ and bench.py; the
reference has no scene generators of its own for these configs (its benchmark inputs are a
downloaded mesh), so these are this repo's definitions of "the configuration the metric is quoted on".
"""
import numpy as np

RAY_DT = np.dtype([("o", "<f4", 3), ("tmin", "<f4"), ("d", "<f4", 3), ("tmax", "<f4")])  # RTRay, src/rt_transport.jl:10-19
IDENTITY3x4 = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32)


def rng(seed):
    return np.random.Generator(np.random.Philox(key=int(seed)))


def make_rays(origins, directions, tmin=0.0, tmax=np.inf):
    origins = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
    directions = np.broadcast_to(np.asarray(directions, dtype=np.float32).reshape(-1, 3), origins.shape)
    r = np.zeros(len(origins), dtype=RAY_DT)
    r["o"], r["d"], r["tmin"], r["tmax"] = origins, directions, tmin, tmax
    return r


# ------------------------------------------------------------------------------------------------
# meshes (triangle soup: (n, 9) float32 = v0 v1 v2)
# ------------------------------------------------------------------------------------------------
def uv_sphere_grid(nu, nv, centre=(0, 0, 0), radius=1.0):
    """nu x nv vertex grid -> 2(nu-1)(nv-1) faces INCLUDING the degenerate pole faces (C1: the
    degenerate filter of src/instanced-bvh.jl:573-600 must drop them)."""
    th = np.linspace(0.0, np.pi, nv, dtype=np.float64)
    ph = np.linspace(0.0, 2 * np.pi, nu, dtype=np.float64)
    P = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)),
                  np.outer(np.cos(th), np.ones_like(ph))], axis=-1)
    P[0, :, :] = [0, 0, 1]     # exact poles => exactly degenerate faces
    P[-1, :, :] = [0, 0, -1]
    P = (P * radius + np.asarray(centre, dtype=np.float64)).astype(np.float32)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.concatenate([a, b, c], axis=-1).reshape(-1, 9)
    t2 = np.concatenate([a, c, d], axis=-1).reshape(-1, 9)
    return np.ascontiguousarray(np.concatenate([t1, t2], axis=0), dtype=np.float32)


def fan_sphere(lon, bands, centre=(0, 0, 0), radius=0.5):
    """Closed sphere with pole fans: 2*lon*(bands-1) triangles, none degenerate, outward winding."""
    th = np.linspace(0.0, np.pi, bands + 1, dtype=np.float64)[1:-1]
    ph = np.arange(lon, dtype=np.float64) * (2 * np.pi / lon)
    ring = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)),
                     np.outer(np.cos(th), np.ones_like(ph))], axis=-1)  # (bands-1, lon, 3)
    nxt = np.roll(ring, -1, axis=1)
    north, south = np.array([0, 0, 1.0]), np.array([0, 0, -1.0])
    tris = [np.concatenate([np.broadcast_to(north, (lon, 3)), ring[0], nxt[0]], axis=-1)]
    for k in range(bands - 2):
        a, b, c, d = ring[k], ring[k + 1], nxt[k + 1], nxt[k]
        tris.append(np.concatenate([a, b, c], axis=-1))
        tris.append(np.concatenate([a, c, d], axis=-1))
    tris.append(np.concatenate([np.broadcast_to(south, (lon, 3)), nxt[-1], ring[-1]], axis=-1))
    T = np.concatenate(tris, axis=0).reshape(-1, 3, 3) * radius + np.asarray(centre, dtype=np.float64)
    return np.ascontiguousarray(T.reshape(-1, 9), dtype=np.float32)


def random_triangles(n, seed, lo=0.0, hi=1.0, edge=0.01):
    g = rng(seed)
    c = g.uniform(lo, hi, size=(n, 1, 3))
    e = g.uniform(-edge, edge, size=(n, 3, 3))
    return np.ascontiguousarray((c + e).reshape(n, 9), dtype=np.float32)


def box_room(lo, hi, k):
    """Inward-facing axis-aligned room, each wall k x k quads => 12 k^2 triangles."""
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    out = []
    s = np.linspace(0.0, 1.0, k + 1)
    for axis in range(3):
        u, v = (axis + 1) % 3, (axis + 2) % 3
        for side in (0, 1):
            for i in range(k):
                for j in range(k):
                    q = np.zeros((4, 3))
                    for n_, (a, b) in enumerate(((s[i], s[j]), (s[i + 1], s[j]), (s[i + 1], s[j + 1]), (s[i], s[j + 1]))):
                        q[n_, axis] = lo[axis] if side == 0 else hi[axis]
                        q[n_, u] = lo[u] + a * (hi[u] - lo[u])
                        q[n_, v] = lo[v] + b * (hi[v] - lo[v])
                    if side == 1:
                        q = q[::-1]
                    out.append(np.concatenate([q[0], q[1], q[2]]))
                    out.append(np.concatenate([q[0], q[2], q[3]]))
    return np.ascontiguousarray(np.array(out), dtype=np.float32)


def random_rotations(n, seed):
    """Uniform random rotations (Shoemake quaternions) -> (n, 3, 3)."""
    g = rng(seed)
    u = g.uniform(size=(n, 3))
    q = np.stack([np.sqrt(1 - u[:, 0]) * np.sin(2 * np.pi * u[:, 1]), np.sqrt(1 - u[:, 0]) * np.cos(2 * np.pi * u[:, 1]),
                  np.sqrt(u[:, 0]) * np.sin(2 * np.pi * u[:, 2]), np.sqrt(u[:, 0]) * np.cos(2 * np.pi * u[:, 2])], axis=1)
    x, y, z, w = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(n, 3, 3)
    return R


def lattice_transforms(nx, ny, nz, pitch, seed, smin=0.5, smax=1.0):
    """Vulkan row-major 3x4 (Mat3x4f bytes, src/instanced-bvh.jl:28-31): rotation*scale | lattice position."""
    n = nx * ny * nz
    R = random_rotations(n, seed)
    s = rng(seed + 1).uniform(smin, smax, size=n)
    idx = np.arange(n)
    pos = np.stack([idx % nx, (idx // nx) % ny, idx // (nx * ny)], axis=1).astype(np.float64) * pitch
    M = np.concatenate([R * s[:, None, None], pos[:, :, None]], axis=2)
    return np.ascontiguousarray(M.reshape(n, 12), dtype=np.float32), pos, s


# ------------------------------------------------------------------------------------------------
# ray sets
# ------------------------------------------------------------------------------------------------
def normalize(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def pinhole_rays(width, height, eye, look_at, fov_deg=45.0, up=(0, 1, 0), jitter_seed=None):
    """One primary ray per pixel through the pixel's centre; jitter_seed: through a uniformly random point of the pixel instead (a
    progressive renderer's frames: the same image, different rays every launch)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = normalize(np.asarray(look_at, dtype=np.float64) - eye)
    r = normalize(np.cross(f, np.asarray(up, dtype=np.float64)))
    u = np.cross(r, f)
    half = np.tan(np.radians(fov_deg) / 2)
    px, py = np.meshgrid(np.arange(width) + 0.5, np.arange(height) + 0.5)
    if jitter_seed is not None:
        g = rng(jitter_seed)
        px = px + g.uniform(-0.5, 0.5, size=px.shape)
        py = py + g.uniform(-0.5, 0.5, size=py.shape)
    X = (px / width * 2 - 1) * half * (width / height)
    Y = (py / height * 2 - 1) * half
    d = normalize(f + X[..., None] * r + Y[..., None] * u).reshape(-1, 3)
    return make_rays(np.broadcast_to(eye, d.shape), d)


def cosine_hemisphere(normals, seed):
    n = normalize(normals)
    g = rng(seed)
    u1, u2 = g.uniform(size=len(n)), g.uniform(size=len(n))
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    helper = np.where(np.abs(n[:, :1]) < 0.9, [[1.0, 0, 0]], [[0, 1.0, 0]])
    t = normalize(np.cross(n, helper))
    b = np.cross(n, t)
    return normalize(t * (r * np.cos(phi))[:, None] + b * (r * np.sin(phi))[:, None] + n * np.sqrt(1 - u1)[:, None])


# ------------------------------------------------------------------------------------------------
# configs: each returns a dict {blas: [(verts, meta)], instances: [(blas_idx1, xforms(m,12), ids)], ...}
# ------------------------------------------------------------------------------------------------
def config_c1():
    """C1: UV sphere 24x24 vertex grid (1058 faces, the 46 pole faces are degenerate), centre (0,0,2)."""
    verts = uv_sphere_grid(24, 24, centre=(0, 0, 2), radius=1.0)
    return {"name": "C1", "blas": [(verts, None)], "instances": [(1, IDENTITY3x4[None], np.zeros(1, np.uint32))],
            "viewdir": np.array([0, 0, 1], np.float32), "grid": 64}


def config_c2(n_tris=100_000, grid=1000):
    """C2: 100k random small triangles in [0,1]^3 (Philox key 0xC2), one BLAS, identity;
    rays = grid x grid generate_ray_grid along normalize(0.3,0.2,1) (get_illumination's shape)."""
    verts = random_triangles(n_tris, 0xC2)
    return {"name": "C2", "blas": [(verts, np.arange(1, n_tris + 1, dtype=np.uint32))],
            "instances": [(1, IDENTITY3x4[None], np.zeros(1, np.uint32))],
            "viewdir": np.array([0.3, 0.2, 1.0], np.float32), "grid": grid}


def config_c3(lon=64, bands=33, lattice=(8, 8, 4), pitch=1.5, seed=0xC3):
    """C3/C4 scene: one 4096-triangle sphere BLAS (64 x 33 fan sphere, radius 0.5) x 256 instances on an
    8x8x4 lattice, pitch 1.5, seeded random rotation x uniform scale in [0.5,1] => 1 048 576 triangles."""
    verts = fan_sphere(lon, bands, radius=0.5)
    xf, pos, scale = lattice_transforms(*lattice, pitch, seed)
    ext = (np.array(lattice) - 1) * pitch
    centre = ext / 2
    return {"name": "C3", "blas": [(verts, np.arange(1, len(verts) + 1, dtype=np.uint32))],
            "instances": [(1, xf, np.arange(1, len(xf) + 1, dtype=np.uint32))],
            "centres": pos, "scales": scale, "lattice_centre": centre,
            "eye": centre + np.array([0.0, 0.0, -(ext[2] / 2 + 14.0)]), "light": np.array([10.0, 10.0, 10.0])}


def c3_primary_rays(cfg, width=2048, height=2048, jitter_seed=None):
    return pinhole_rays(width, height, cfg["eye"], cfg["lattice_centre"], 45.0, jitter_seed=jitter_seed)


def c3_hit_frames(cfg, rays, hits):
    """Hit points and analytic outward normals (instances are spheres) for the rays that hit."""
    m = hits["hit"] == 1
    p = rays["o"][m].astype(np.float64) + hits["t"][m].astype(np.float64)[:, None] * rays["d"][m].astype(np.float64)
    n = normalize(p - cfg["centres"][hits["instance_id"][m]])
    return p, n


def c3_shadow_rays(cfg, rays, hits):
    """any_hit shadow rays from the primary hit points (+1e-3 n) toward the point light; t_max = distance."""
    p, n = c3_hit_frames(cfg, rays, hits)
    o = p + 1e-3 * n
    to = cfg["light"] - o
    dist = np.linalg.norm(to, axis=1)
    return make_rays(o, to / dist[:, None], 0.0, dist.astype(np.float32))


def c4_bounce_rays(cfg, rays, hits, n_rays, seed=0xC4):
    """C4: incoherent diffuse bounce rays: origins = primary hit points (+1e-3 n), directions cosine-weighted
    about the normal; hit points are reused round-robin to reach n_rays."""
    p, n = c3_hit_frames(cfg, rays, hits)
    idx = np.arange(n_rays) % len(p)
    d = cosine_hemisphere(n[idx], seed)
    return make_rays(p[idx] + 1e-3 * n[idx], d)


def config_c5(lon=96, bands=51, wall_k=13):
    """C5: closed scene for view_factors: 5 spheres (2*lon*(bands-1) tris each) inside a box room with
    12*wall_k^2 wall triangles; ONE BLAS at identity, metadata = 1..N (view_factors reads local-space
    vertices and indexes the matrix by metadata, src/kernels.jl:80-104).  Defaults: 50 028 triangles."""
    cs = [(-1.2, -1.2, -0.6), (1.2, -1.2, 0.5), (-1.2, 1.2, 0.4), (1.2, 1.2, -0.5), (0.0, 0.0, 0.0)]
    parts = [fan_sphere(lon, bands, centre=c, radius=0.7) for c in cs]
    parts.append(box_room((-2.5, -2.5, -2.0), (2.5, 2.5, 2.0), wall_k))
    verts = np.ascontiguousarray(np.concatenate(parts, axis=0), dtype=np.float32)
    n = len(verts)
    return {"name": "C5", "blas": [(verts, np.arange(1, n + 1, dtype=np.uint32))],
            "instances": [(1, IDENTITY3x4[None], np.zeros(1, np.uint32))], "rays_per_triangle": 4096}
