"""Iso-surface extraction for ``extract_geometry`` (reference: lib/core/utils/nerf_utils.py:82-112, which hands the density volume to PyMCubes'
``mcubes.marching_cubes``).  PyMCubes is a third-party CPU library and not installable here, so the step is native:
marching cubes on the GPU over the volume the fused density decode has just assembled there (csrc/marching_cubes.hip) -- classify every cell,
prefix-sum the counts, emit SHARED vertices (one per crossing lattice edge, linearly interpolated like PyMCubes) and indexed triangles.  No copy of
the 256^3 volume to the host, no Python loop over cells.

The 256-case triangle table is GENERATED here from first principles instead of being typed in from the classic listing: per case, every cube face
contributes the segments that separate its inside corners from its outside corners (a face with alternating corners is cut around its INSIDE
corners -- a rule that depends on the face's four signs only, so the two cells that share the face agree and the mesh is watertight), the directed
segments close into loops, and every loop is cut into triangles by diagonals that never lie in a cube face (``_triangulate_loop``: a diagonal in
a face could coincide with an edge of the neighbouring cell -- two sheets through one edge).  ``tests/test_mesh_cpu.py`` checks the table (every
case closes, <= 5 triangles, complementary cases use the same edges) and the pure-Python walker below on analytic and random volumes (closed
oriented manifold, Euler characteristic, area, enclosed volume); the GPU tests check the kernels against this walker.

Conventions (PyMCubes'): ``volume[x, y, z]``, vertices in index coordinates (float), a corner is INSIDE when its value is greater than the
iso-value, triangles wind counter-clockwise seen from the outside (normals point from high density to low)."""
from __future__ import annotations

from functools import lru_cache
from typing import List, Tuple

import numpy as np

# corner i of a cell at (x, y, z) sits at (x, y, z) + CORNERS[i]; edge e joins corners EDGES[e]
CORNERS = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.int64)
EDGES = np.array([[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4], [0, 4], [1, 5], [2, 6], [3, 7]], dtype=np.int64)
# the six faces as corner cycles, counter-clockwise seen from OUTSIDE the cube
FACES = [[0, 3, 2, 1], [4, 5, 6, 7], [0, 1, 5, 4], [2, 3, 7, 6], [0, 4, 7, 3], [1, 2, 6, 5]]
# lattice edge that carries cube edge e: (offset of its owning lattice point, axis)
EDGE_OWNER = np.array([[0, 0, 0, 0], [1, 0, 0, 1], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0], [1, 0, 1, 1], [0, 1, 1, 0], [0, 0, 1, 1],
                       [0, 0, 0, 2], [1, 0, 0, 2], [1, 1, 0, 2], [0, 1, 0, 2]], dtype=np.int64)
MAX_TRIS = 5


def _edge_of(a: int, b: int) -> int:
    for e, (p, q) in enumerate(EDGES):
        if (p, q) == (a, b) or (p, q) == (b, a):
            return e
    raise KeyError((a, b))


def _case_loops(case: int) -> List[List[int]]:
    """directed loops of cube edges for one corner-sign case (bit i set = corner i inside)"""
    inside = [(case >> i) & 1 for i in range(8)]
    nxt = {}
    for face in FACES:
        # walk the face's corner cycle (CCW from outside); at every inside -> outside transition (corner k inside, k+1 outside) a segment STARTS on
        # edge (k, k+1) ... and it ends on the edge of the next outside -> inside transition.  With the inside corners kept on the left of the
        # directed segment this is: start = the edge where the cycle LEAVES the inside set, end = the edge where it ENTERS it again.  A face with
        # alternating corners has two leave/enter pairs: pairing each "enter" edge with the "leave" edge that FOLLOWS it around the same inside
        # corner cuts around the inside corners.
        s = [inside[c] for c in face]
        for k in range(4):
            if s[k] == 0 and s[(k + 1) % 4] == 1:                       # the cycle enters the inside set at corner k+1 through edge (k, k+1)
                j = (k + 1) % 4
                while s[(j + 1) % 4] == 1:                               # ... and leaves it after the run of inside corners
                    j = (j + 1) % 4
                e_in = _edge_of(face[k], face[(k + 1) % 4])
                e_out = _edge_of(face[j], face[(j + 1) % 4])
                assert e_out not in nxt
                nxt[e_out] = e_in                                        # inside on the LEFT of the segment e_out -> e_in (seen from outside)
    loops, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, e = [], start
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == start, "segments of a case must close into loops"
        loops.append(loop)
    return loops


_FACE_EDGES = None


def _on_one_face(a: int, b: int) -> bool:
    global _FACE_EDGES
    if _FACE_EDGES is None:
        _FACE_EDGES = [{_edge_of(f[k], f[(k + 1) % 4]) for k in range(4)} for f in FACES]
    return any(a in fe and b in fe for fe in _FACE_EDGES)


def _triangulations(poly: List[int]) -> List[List[Tuple[int, int, int]]]:
    """every triangulation of a convex polygon given by its vertex labels in order (Catalan many; loops have <= 7 vertices)"""
    if len(poly) == 3:
        return [[(poly[0], poly[1], poly[2])]]
    out = []
    for k in range(1, len(poly) - 1):
        lefts = _triangulations(poly[:k + 1]) if k >= 2 else [[]]
        rights = _triangulations(poly[k:]) if len(poly) - k >= 3 else [[]]
        for left in lefts:
            for right in rights:
                out.append(left + [(poly[0], poly[k], poly[-1])] + right)
    return out


def _triangulate_loop(loop: List[int]) -> List[Tuple[int, int, int]]:
    """A triangulation of the loop none of whose DIAGONALS joins two vertices of one cube face.  Such a diagonal lies in that face; the cell on the
    other side of the face holds the same two vertices and may put an edge between them too -- two sheets through one edge, a non-manifold mesh
    (a plain fan does this on faces with alternating corners).  Every loop of every case has such a triangulation (tests/test_mesh_cpu.py)."""
    n = len(loop)
    boundary = {(loop[i], loop[(i + 1) % n]) for i in range(n)} | {(loop[(i + 1) % n], loop[i]) for i in range(n)}
    for tris in _triangulations(loop):
        diagonals = {(t[i], t[(i + 1) % 3]) for t in tris for i in range(3)} - boundary
        if not any(_on_one_face(a, b) for a, b in diagonals):
            return tris
    raise AssertionError(f"no face-free triangulation of loop {loop}")


@lru_cache(maxsize=None)
def triangle_table() -> Tuple[np.ndarray, np.ndarray]:
    """(tri_count[256] uint8, tri_edges[256, 15] int8): the triangles of every case as triples of cube-edge ids, -1 padded"""
    counts = np.zeros(256, dtype=np.uint8)
    table = np.full((256, 3 * MAX_TRIS), -1, dtype=np.int8)
    for case in range(256):
        tris = []
        for loop in _case_loops(case):
            tris += [(a, c, b) for a, b, c in _triangulate_loop(loop)]      # (the loops run clockwise seen from outside: swap to wind counter-clockwise)
        assert len(tris) <= MAX_TRIS, (case, len(tris))
        counts[case] = len(tris)
        for t, tri in enumerate(tris):
            table[case, 3 * t:3 * t + 3] = tri
    return counts, table


def marching_cubes_reference(volume: np.ndarray, iso: float) -> Tuple[np.ndarray, np.ndarray]:
    """Pure-numpy walker over the cells with the same table, vertex sharing and interpolation as the kernels (test infrastructure for small
    volumes; the product path is ``marching_cubes`` below)."""
    counts, table = triangle_table()
    nx, ny, nz = volume.shape
    vol = volume.astype(np.float32)
    vert_id, verts, tris = {}, [], []
    for x in range(nx):                                                   # vertex order: lattice points x-major, axes 0, 1, 2 (the kernels' order)
        for y in range(ny):
            for z in range(nz):
                for axis, (dx, dy, dz) in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
                    x2, y2, z2 = x + dx, y + dy, z + dz
                    if x2 >= nx or y2 >= ny or z2 >= nz:
                        continue
                    a, b = vol[x, y, z], vol[x2, y2, z2]
                    if (a > iso) != (b > iso):
                        t = (np.float32(iso) - a) / (b - a)
                        p = np.array([x, y, z], np.float32)
                        p[axis] += t
                        vert_id[(x, y, z, axis)] = len(verts)
                        verts.append(p)
    for x in range(nx - 1):
        for y in range(ny - 1):
            for z in range(nz - 1):
                case = 0
                for i, (cx, cy, cz) in enumerate(CORNERS):
                    case |= int(vol[x + cx, y + cy, z + cz] > iso) << i
                for t in range(counts[case]):
                    tri = []
                    for e in table[case, 3 * t:3 * t + 3]:
                        ox, oy, oz, axis = EDGE_OWNER[e]
                        tri.append(vert_id[(x + ox, y + oy, z + oz, axis)])
                    tris.append(tri)
    return (np.array(verts, np.float32).reshape(-1, 3), np.array(tris, np.int32).reshape(-1, 3))


def marching_cubes(volume, iso: float):
    """volume (nx, ny, nz) float32 CUDA tensor -> (vertices (V, 3) float32 in index coordinates, triangles (T, 3) int32), both on the device.
    Three launches around two prefix sums; one host read (the two totals, to size the outputs)."""
    import torch

    from . import _cabi as C
    assert volume.is_cuda and volume.dim() == 3, "marching_cubes: a (nx, ny, nz) CUDA tensor"
    vol = volume.detach().to(torch.float32).contiguous()
    nx, ny, nz = (int(v) for v in vol.shape)
    n = nx * ny * nz
    # the prefix sums of triangles (<= 5 per cell) and vertices (<= 3 per lattice point) and the emit kernel's offsets are int32
    if 5 * n >= 2 ** 31:
        raise ValueError(f"marching_cubes: a {nx} x {ny} x {nz} lattice can hold more than 2^31 triangle slots (int32 offsets); extract it in chunks")
    dev = vol.device
    counts, table = triangle_table()
    key = str(dev)
    if key not in _TABLES:
        _TABLES[key] = (torch.from_numpy(counts).to(dev), torch.from_numpy(table.astype(np.int8)).to(dev))
    d_counts, d_table = _TABLES[key]
    cell_tris = torch.empty(n, dtype=torch.int32, device=dev)
    point_verts = torch.empty(n, dtype=torch.int32, device=dev)
    point_mask = torch.empty(n, dtype=torch.uint8, device=dev)
    C.check(C.lib().ssdnerf_marching_cubes_count(C.ptr(vol), C.u32(nx), C.u32(ny), C.u32(nz), C.f32(float(iso)), C.ptr(d_counts), C.ptr(cell_tris),
                                                 C.ptr(point_verts), C.ptr(point_mask), C.stream()), "marching_cubes_count")
    tri_off = torch.cumsum(cell_tris, 0, dtype=torch.int32)
    vert_off = torch.cumsum(point_verts, 0, dtype=torch.int32)
    totals = torch.stack([tri_off[-1], vert_off[-1]]).tolist()                    # the one device -> host read
    n_tris, n_verts = int(totals[0]), int(totals[1])
    vertices = torch.empty(n_verts, 3, dtype=torch.float32, device=dev)
    triangles = torch.empty(n_tris, 3, dtype=torch.int32, device=dev)
    if n_tris:
        C.check(C.lib().ssdnerf_marching_cubes_emit(C.ptr(vol), C.u32(nx), C.u32(ny), C.u32(nz), C.f32(float(iso)), C.ptr(d_counts), C.ptr(d_table),
                                                    C.ptr(tri_off), C.ptr(vert_off), C.ptr(point_mask), C.ptr(vertices), C.ptr(triangles), C.stream()), "marching_cubes_emit")
    return vertices, triangles


_TABLES: dict = {}


def mesh_stats(vertices: np.ndarray, triangles: np.ndarray) -> dict:
    """closedness / orientation / size figures of an indexed triangle mesh (tests)"""
    tri = triangles.astype(np.int64)
    he = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], axis=0)
    n = int(vertices.shape[0])
    fwd = he[:, 0] * n + he[:, 1]
    rev = he[:, 1] * n + he[:, 0]
    uf, cf = np.unique(fwd, return_counts=True)
    p = vertices[tri].astype(np.float64)
    cross = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    return dict(directed_edges_unique=bool((cf == 1).all()), closed_and_oriented=bool((cf == 1).all() and np.array_equal(uf, np.unique(rev))),
                euler=n - len(uf) // 2 + len(tri), area=float(0.5 * np.linalg.norm(cross, axis=1).sum()),
                volume=float((p[:, 0] * cross).sum() / 6.0), degenerate=int((np.linalg.norm(cross, axis=1) == 0).sum()))
