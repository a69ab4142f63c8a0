"""Geometry chosen to hit the places where the ORDER of the reference's traversal decides the result (intersection.cuh:247, :561:
`t < best`, strict -- of several triangles at exactly the same distance the one found first wins) and where its arithmetic
leaves the ordinary range (zero-area and collinear triangles: |det| < 1e-6 rejects, intersection.cuh:231; direction components
of exactly 0: the +-1e30 reciprocal, intersection.cuh:23-40; origins on box faces).  Every loop shape of the kernel has its own
way of visiting pairs out of order (pair queues, subtree stealing, merged shadow rays, TLAS rounds) and its own tie rule that has
to reproduce first-found-wins; all of them against the oracle, bit for bit."""
import numpy as np
import pytest

from common import assert_frames_equal, render_both

pytestmark = pytest.mark.gpu


# (options, the loop shape -- PMODE -- they must lead to; None: whichever the first frames' timing picks)
VARIANTS = [(dict(), None), (dict(merged=0, csteal=0), 2), (dict(merged=0, csteal=2, csteal_min=0), 2),
            (dict(merged=1, csteal=0, steal=0), 4), (dict(merged=1, csteal=2, csteal_min=0, steal=1), 4),
            (dict(merged=0, csteal=1, leaf_pairs=0), 2), (dict(force_geom=2), 3), (dict(force_geom=2, merged=0, tlas_rounds=1), 3),
            (dict(pair_trace=0), 0), (dict(force_geom=2, pair_trace=0), 0)]


@pytest.mark.parametrize("opts,pmode", VARIANTS, ids=[",".join(f"{k}={v}" for k, v in o.items()) or "defaults" for o, _ in VARIANTS])
@pytest.mark.parametrize("leaf", [2, 8])
def test_ties_and_degenerate_triangles(P, O, blue_noise, opts, pmode, leaf):
    """leaf 8: the eight meshes are one TLAS leaf (the pair queues, PMODE 2 / 4); leaf 2: the reference's builder splits the TLAS
    with the same target (scene.cuh:556), a real TLAS of eight meshes two of which fill the same box (TLAS rounds, PMODE 3)."""
    if leaf == 2:
        pmode = 0 if opts.get("pair_trace") == 0 else 3
    s = P.Scene(88, 64)
    P.scenes.coincident(s, leaf=leaf)
    for k, v in opts.items():
        s.set_option(k, v)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 5, 2)
    assert pmode is None or s.get_option("pmode") == pmode
    assert_frames_equal(gpu, cpu)
    ids = gpu[0]["object_id"]
    assert (ids == 0).any() and not (ids == 1).any() and (ids == 3).any() and not (ids == 4).any(), \
        "of two identical surfaces the first mesh is the one that is seen"
    s.close()


@pytest.mark.parametrize("merged", [0, 1])
def test_stolen_subtrees_among_coincident_triangles(P, O, blue_noise, merged):
    """A finer wall (2 x 24 x 24 quads, every triangle twice) in a larger frame: whole waves of rays whose walks are stolen from
    (csteal_min 0: at once) while equal distances come from different leaves of one BLAS -- about one ray in a hundred is marked
    by a thief and traced again without stealing (tools/trav_stats.py coincident: 162,034 of 13.7 M at 1080p).  The oracle's bits."""
    s = P.Scene(480, 270)
    P.scenes.coincident(s, n=24, leaf=8)
    for k, v in dict(merged=merged, steal=1, csteal=2, csteal_min=0).items():
        s.set_option(k, v)
    gpu, cpu = render_both(P, O, s, blue_noise, 2, 4, 1)
    assert s.get_option("pmode") == (4 if merged else 2)
    assert_frames_equal(gpu, cpu)
    s.close()


def test_axis_parallel_rays_and_origins_on_box_faces(P, O):
    """traceSingleRay (scene.cuh:1367-1391) over the same scene: directions with components of exactly zero, origins in the planes
    of the wall, the floors and the leaf boxes' faces, rays along triangle edges and through grid vertices."""
    s = P.Scene(32, 32)
    P.scenes.coincident(s, n=6, leaf=2)
    s.uploadToGPU()
    axes = np.array([[0, 0, -1], [0, 0, 1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [1, 1, 0], [0, -1, -1], [1, 0, -1]], np.float32)
    axes /= np.linalg.norm(axes, axis=1, keepdims=True).astype(np.float32)
    grid = np.linspace(-3.0, 3.0, 13, dtype=np.float32)  # every second value is a quad edge of the 6 x 6 wall
    o, d = [], []
    for z in (4.0, -6.0, -5.0, -7.0):
        for y in (-3.0, 0.0, 1.0, 3.0):
            for x in grid:
                for a in axes:
                    o.append((x, y, z))
                    d.append(a)
    o, d = np.array(o, np.float32), np.array(d, np.float32)
    g = s.trace_rays(o, d)
    c = O.trace_rays(s.flatten(), o, d)
    for name in g.dtype.names:
        a, b = g[name], c[name]
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), name
    assert 0 < g["hit"].sum() < len(o)
    s.close()
