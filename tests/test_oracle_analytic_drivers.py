"""Checks of the oracle's get_illumination / view_factors restatements against closed forms -- answers that do not come from the
restated code (the reference has no tests for src/kernels.jl, so equality of the HIP path and the oracle alone would only show that
two implementations by the same author agree).

* view_factors (src/kernels.jl:80-104) shoots `random_hemisphere_uniform` directions (src/math.jl:125-141: cos(theta) uniform in
  [0, 1], i.e. uniform in SOLID ANGLE, not cosine weighted) from `random_triangle_point` origins lifted 0.01 along the normal: the
  expected fraction of a small source's rays that land on a target is the target's solid angle over 2 pi.  For a square
  [-a, a]^2 at height h above the source point that is 4 atan(a^2 / (h sqrt(2 a^2 + h^2))) (the classical pyramid formula).
* get_illumination (src/kernels.jl:10-56, 112-124) counts the rays of a grid_size^2 grid spanning the projected bounds plus a 5 %
  margin on each side: a rectangle that fills the bounds and faces the view direction is hit by exactly the grid points inside it.
"""
import numpy as np
import pytest


def square(z, a, flip=False):
    """[-a, a]^2 at height z as two triangles; normal +z (counter-clockwise seen from above) or -z with flip."""
    p = [(-a, -a, z), (a, -a, z), (a, a, z), (-a, a, z)]
    tris = [(p[0], p[1], p[2]), (p[0], p[2], p[3])]
    if flip:
        tris = [(t[0], t[2], t[1]) for t in tris]
    return np.array(tris, np.float32).reshape(-1, 9)


def solid_angle_fraction(a, h):
    return 4.0 * np.arctan(a * a / (h * np.sqrt(2 * a * a + h * h))) / (2 * np.pi)


@pytest.mark.parametrize("a,h", [(1.0, 1.0), (0.5, 1.0), (2.0, 0.5)])
def test_view_factor_counts_follow_the_uniform_hemisphere_solid_angle(oracle, a, h):
    eps = 1e-3  # source: a tiny triangle around the origin, normal +z
    src = np.array([[-eps, -eps, 0, eps, -eps, 0, 0, eps, 0]], np.float32)
    verts = np.concatenate([src, square(h, a, flip=True)])  # the target faces the source (its own rays go down and hit nothing)
    s = oracle.Scene()
    s.add_instance(s.add_blas(verts, meta=[1, 2, 3]))
    s.build()
    rays = 400_000
    m = s.view_factors(rays, seed=7, nthreads=8)
    row = m[0].astype(np.int64)  # rows / columns are metadata - 1: row 0 = the source
    assert row[0] == 0
    got = row[1:].sum() / rays
    want = solid_angle_fraction(a, h - 0.01)  # origins are lifted 0.01 along the normal (src/kernels.jl:93)
    sigma = np.sqrt(want * (1 - want) / rays)
    assert abs(got - want) < 5 * sigma, (got, want, sigma)
    # a cosine-weighted sampler (the textbook view factor) would give a clearly different number: the test can tell them apart
    x = a / np.sqrt(a * a + (h - 0.01) ** 2)
    cosine = 4.0 / np.pi * x * np.arctan(x)  # differential element to a centred parallel square, Lambertian
    assert abs(cosine - want) > 20 * sigma


def test_view_factor_origins_are_uniform_over_the_source_triangle(oracle):
    """random_triangle_point (src/math.jl:158-175): with a source much LARGER than the gap to a covering plane, every ray lands right
    above its origin's neighbourhood, so the split of the hits between the plane's parts is the split of the source's area."""
    src = np.array([[0, 0, 0, 2, 0, 0, 0, 2, 0]], np.float32)  # right triangle, legs 2: area 2; the part with x < 1 has area 1.5
    h = 0.02                                                   # (rays start at z = 0.01)
    def strip(x0, x1):
        p = [(x0, -50, h), (x1, -50, h), (x1, 50, h), (x0, 50, h)]
        return np.array([(p[0], p[2], p[1]), (p[0], p[3], p[2])], np.float32).reshape(-1, 9)
    verts = np.concatenate([src, strip(-50, 1), strip(1, 50)])
    s = oracle.Scene()
    s.add_instance(s.add_blas(verts, meta=[1, 2, 3, 4, 5]))
    s.build()
    rays = 200_000
    row = s.view_factors(rays, seed=3, nthreads=8)[0].astype(np.int64)
    left, right = row[1] + row[2], row[3] + row[4]
    assert left + right > 0.97 * rays  # nearly every direction reaches the plane 0.01 above (the rest leaves sideways)
    frac = left / (left + right)
    # directions spread the landing point by ~0.01 tan(theta): only origins within ~0.1 of x = 1 can cross over, symmetric to first order
    assert abs(frac - 0.75) < 0.01, frac


@pytest.mark.parametrize("grid", [110, 220])
def test_illumination_counts_the_grid_points_inside_a_facing_rectangle(oracle, grid):
    verts = np.array([[0, 0, 0, 1, 0, 0, 1, 2, 0], [0, 0, 0, 1, 2, 0, 0, 2, 0]], np.float32)  # 1 x 2 rectangle in z = 0
    s = oracle.Scene()
    s.add_instance(s.add_blas(verts, meta=[1, 2]))
    s.build()
    counts = s.get_illumination((0, 0, -1), grid, nthreads=4)
    # grid spans 1.1 x the larger projected extent (2) in BOTH directions (square cells: margin = 5 % of the larger extent on each side,
    # width = extent + 2 margins per axis): cells 2.2 / grid along the long side, (1 + 0.2) / grid along the short one
    n_long = sum(1 for i in range(1, grid + 1) if abs((i - (grid + 1) / 2) * (2.2 / grid)) <= 1.0)
    n_short = sum(1 for i in range(1, grid + 1) if abs((i - (grid + 1) / 2) * (1.2 / grid)) <= 0.5)
    total = int(counts.sum())
    assert abs(total - n_long * n_short) <= max(n_long, n_short)  # rays exactly on the outline or the diagonal may go either way
    assert abs(int(counts[0]) - int(counts[1])) <= 2 * max(n_long, n_short)  # the diagonal splits the rectangle in halves
