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


# ---------------------------------------------------------------------------------------------- r06: the generated table against the CLASSIC one (PyMCubes' table)
def _loops_of(tris):
    """the boundary of a set of triangles (cube-edge ids) as a set of undirected cyclic loops, canonical form; None if it is not a union of simple cycles"""
    import collections
    cnt = collections.Counter()
    for a, b, c in tris:
        for p, q in ((a, b), (b, c), (c, a)):
            cnt[frozenset((p, q))] += 1
    boundary = [tuple(e) for e, n in cnt.items() if n == 1]
    adj = collections.defaultdict(list)
    for p, q in boundary:
        adj[p].append(q); adj[q].append(p)
    if any(len(v) != 2 for v in adj.values()):
        return None
    loops, seen = set(), set()
    for start in sorted(adj):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        while True:
            seen.add(cur)
            nxt = [v for v in adj[cur] if v != prev]
            nxt = nxt[0] if nxt else adj[cur][0]
            if nxt == start:
                break
            loop.append(nxt); prev, cur = cur, nxt
        k = loop.index(min(loop))
        fwd = tuple(loop[k:] + loop[:k])
        bwd = tuple([fwd[0]] + list(reversed(fwd[1:])))
        loops.add(min(fwd, bwd))
    return frozenset(loops)


def test_generated_table_against_the_classic_table():
    """The reference's mesh step is PyMCubes (lib/core/utils/nerf_utils.py:88), which walks the classic 256-case table (fixture mc_classic_table.npy: made by
    tests/golden/make_mc_classic.py from the copy in scikit-image's sources).  The table generated here uses the same corner and edge numbering; a corner counts as inside
    when its value is ABOVE the threshold, the classic index sets a bit when it is BELOW, so case c here is case 255 - c there.  Checked per case: the same cube edges
    carry vertices (all 256); where no cube face has alternating corners the surface patches have the same boundary loops and the same number of triangles -- the two
    meshes are the same surface, cell by cell, up to which diagonal splits a patch --; the cases that differ are all ambiguous ones, where the classic table's fixed choice
    is known to leave holes between neighbouring cells and the generated table cuts around the inside corners (mesh.py)."""
    import os
    from ssdnerf_amd import mesh as M
    classic = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mc_classic_table.npy"))
    counts, table = M.triangle_table()
    same_loops = differing = 0
    for c in range(256):
        mine = [tuple(int(e) for e in table[c, 3 * t:3 * t + 3]) for t in range(int(counts[c]))]
        row = classic[255 - c]
        theirs = [tuple(int(e) for e in row[3 * t:3 * t + 3]) for t in range(5) if row[3 * t] >= 0]
        assert {e for t in mine for e in t} == {e for t in theirs for e in t}, c                      # the same crossing edges
        inside = [(c >> i) & 1 for i in range(8)]
        ambiguous = any(inside[f[0]] == inside[f[2]] != inside[f[1]] == inside[f[3]] for f in M.FACES)
        la, lb = _loops_of(mine), _loops_of(theirs)
        assert la is not None
        if la == lb:
            same_loops += 1
            assert len(mine) == len(theirs), c
        else:
            differing += 1
            assert ambiguous, (c, mine, theirs)                                                       # only ambiguous cases may be cut differently
    # all 136 cases without an alternating face agree; all 120 with one are cut the other way round (the classic index is the COMPLEMENT of the index here, and the
    # classic rule separates the corners that are set in ITS index)
    n_unamb = sum(1 for c in range(256) if not any(((c >> f[0]) & 1) == ((c >> f[2]) & 1) != ((c >> f[1]) & 1) == ((c >> f[3]) & 1) for f in M.FACES))
    assert (same_loops, differing, n_unamb) == (136, 120, 136)


def test_walker_with_the_classic_table_gives_the_same_surface_on_a_smooth_volume():
    """a sphere sampled finely enough has no cell with an alternating face: the classic table and the generated one must then produce the same vertices, the same number
    of triangles, and the same surface (area and enclosed volume to 2e-4 relative: only the diagonals inside a patch may differ)"""
    import os
    from ssdnerf_amd import mesh as M
    classic = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mc_classic_table.npy"))
    n = 20
    x = np.linspace(-1, 1, n, dtype=np.float32)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    vol = (0.7 - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32)
    v_a, t_a = M.marching_cubes_reference(vol, 0.0)
    # the same walker with the classic table: index 255 - case (its triangles then wind like the generated ones: outward normals)
    counts, table = M.triangle_table()
    c_counts = np.array([int((classic[255 - c] >= 0).sum()) // 3 for c in range(256)], dtype=np.uint8)
    c_table = np.full((256, 15), -1, dtype=np.int8)
    for c in range(256):
        row = classic[255 - c]
        for t in range(int(c_counts[c])):
            a, b, cc = (int(e) for e in row[3 * t:3 * t + 3])
            c_table[c, 3 * t:3 * t + 3] = (a, b, cc)
    saved = M.triangle_table
    try:
        M.triangle_table = lambda: (c_counts, c_table)
        v_b, t_b = M.marching_cubes_reference(vol, 0.0)
    finally:
        M.triangle_table = saved
    assert np.array_equal(v_a, v_b) and t_a.shape == t_b.shape

    def area_volume(v, t):
        p, q, r = v[t[:, 0]].astype(np.float64), v[t[:, 1]].astype(np.float64), v[t[:, 2]].astype(np.float64)
        return 0.5 * np.linalg.norm(np.cross(q - p, r - p), axis=1).sum(), np.einsum("ij,ij->i", p, np.cross(q, r)).sum() / 6.0
    (a1, w1), (a2, w2) = area_volume(v_a, t_a), area_volume(v_b, t_b)
    assert abs(a1 - a2) < 2e-4 * a1 and abs(w1 - w2) < 2e-4 * abs(w1) and w1 * w2 > 0          # (same orientation too)
