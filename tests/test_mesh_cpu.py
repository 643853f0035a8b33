"""The marching-cubes table and walker of ssdnerf_amd/mesh.py (the mesh half of ``extract_geometry``,
lib/core/utils/nerf_utils.py:82-112) -- generated table, not a typed-in one, so its properties are checked here; the GPU kernels are checked
against the walker in tests/test_mesh_gpu.py."""
import numpy as np

from ssdnerf_amd import mesh as M


def test_every_case_closes_into_loops_and_fits_five_triangles():
    counts, table = M.triangle_table()
    assert counts[0] == 0 and counts[255] == 0 and counts.max() == M.MAX_TRIS
    for case in range(256):
        loops = M._case_loops(case)
        crossing = {e for e, (a, b) in enumerate(M.EDGES) if ((case >> a) & 1) != ((case >> b) & 1)}
        assert {e for loop in loops for e in loop} == crossing and sum(len(loop) for loop in loops) == len(crossing)   # every crossing edge once
        assert counts[case] == sum(len(loop) - 2 for loop in loops)
        used = table[case, :3 * counts[case]]
        assert (used >= 0).all() and (table[case, 3 * counts[case]:] == -1).all() and set(used.tolist()) == crossing
        assert {e for loop in M._case_loops(255 - case) for e in loop} == crossing                                      # the complement cuts the same edges
        # no triangle edge lies in a cube face unless it is a segment of the loop itself (see _triangulate_loop)
        boundary = {frozenset((loop[i], loop[(i + 1) % len(loop)])) for loop in loops for i in range(len(loop))}
        for t in range(counts[case]):
            tri = table[case, 3 * t:3 * t + 3].tolist()
            for k in range(3):
                a, b = tri[k], tri[(k + 1) % 3]
                assert frozenset((a, b)) in boundary or not M._on_one_face(a, b), (case, tri)


def _lattice(n):
    g = np.arange(n, dtype=np.float32)
    return np.meshgrid(g, g, g, indexing="ij")


def test_walker_on_analytic_and_random_volumes():
    n = 24
    X, Y, Z = _lattice(n)
    c = (n - 1) / 2
    sphere = (8.0 - np.sqrt((X - c) ** 2 + (Y - c + 0.3) ** 2 + (Z - c - 0.2) ** 2)).astype(np.float32)
    v, t = M.marching_cubes_reference(sphere, 0.0)
    st = M.mesh_stats(v, t)
    assert st["closed_and_oriented"] and st["euler"] == 2 and st["degenerate"] == 0
    assert abs(st["area"] / (4 * np.pi * 64) - 1) < 0.01 and abs(st["volume"] / (4 / 3 * np.pi * 512) - 1) < 0.015      # positive: outward normals
    r = np.sqrt((X - c) ** 2 + (Y - c) ** 2)
    torus = (3.0 - np.sqrt((r - 7) ** 2 + (Z - c) ** 2)).astype(np.float32)
    st = M.mesh_stats(*M.marching_cubes_reference(torus, 0.0))
    assert st["closed_and_oriented"] and st["euler"] == 0 and st["volume"] > 0
    # vertices sit on lattice edges, at the linear interpolant of the two corner values
    for p in v[::37]:
        frac = p - np.floor(p)
        assert (frac > 0).sum() <= 1
    rng = np.random.default_rng(0)
    for _ in range(3):                                                          # white noise: every ambiguous face / cell configuration shows up
        vol = np.full((14, 14, 14), -5, np.float32)
        vol[1:-1, 1:-1, 1:-1] = rng.standard_normal((12, 12, 12))
        st = M.mesh_stats(*M.marching_cubes_reference(vol, 0.0))
        assert st["closed_and_oriented"] and st["degenerate"] == 0 and st["volume"] > 0
