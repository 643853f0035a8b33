// ssdnerf_amd/csrc/marching_cubes.hip -- iso-surface of a density volume on the GPU.
//
// Reference: lib/core/utils/nerf_utils.py:82-112 `extract_geometry` copies the 256^3 density volume chunk by chunk to the host and runs PyMCubes'
// `mcubes.marching_cubes` on it.  Here the volume never leaves the device: `nerf.extract_density_volume` assembles it with the fused density
// decode, and the two kernels below turn it into an indexed triangle mesh -- SHARED vertices, one per crossing lattice edge, linearly interpolated
// along the edge (PyMCubes' rule), triangles from a 256-case table that ssdnerf_amd/mesh.py generates (and explains).
//
//   k_mc_count : one lane per lattice point p = (x, y, z): which of its three owned edges (+x, +y, +z) cross the iso-value (3-bit mask, vertex
//                count) and how many triangles its cell emits (table lookup by the 8-corner sign case).
//   (host)     : inclusive prefix sums of the two count arrays (torch.cumsum), one host read of the two totals.
//   k_mc_emit  : the same lane writes its vertices at its offset and its cell's triangles; a triangle corner on cube edge e is the vertex of the
//                lattice point that OWNS e: offset of that point + rank of the edge's axis in the point's mask.
// Bound: HBM stream (4 B/point read ~3x from cache, 8 B/point of counts written and read back, ~1 B mask).  Index math in 32 bits (n < 2^31).
#include "common.h"

namespace {

constexpr unsigned MC_TPB = 256;

struct McGeo { uint32_t nx, ny, nz, sx, sy; };   // sx = ny * nz, sy = nz: strides of the x-major volume[x][y][z]

SSD_DEV uint32_t mc_mask(const float* __restrict__ vol, const McGeo& g, uint32_t p, uint32_t x, uint32_t y, uint32_t z, float iso) {
    const bool in0 = vol[p] > iso;
    uint32_t m = 0;
    if (x + 1 < g.nx && (vol[p + g.sx] > iso) != in0) m |= 1u;
    if (y + 1 < g.ny && (vol[p + g.sy] > iso) != in0) m |= 2u;
    if (z + 1 < g.nz && (vol[p + 1] > iso) != in0) m |= 4u;
    return m;
}

SSD_DEV uint32_t mc_case(const float* __restrict__ vol, const McGeo& g, uint32_t p, float iso) {
    // corner order of ssdnerf_amd/mesh.py CORNERS: (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) (1,0,1) (1,1,1) (0,1,1)
    uint32_t c = 0;
    c |= (uint32_t)(vol[p] > iso) << 0;
    c |= (uint32_t)(vol[p + g.sx] > iso) << 1;
    c |= (uint32_t)(vol[p + g.sx + g.sy] > iso) << 2;
    c |= (uint32_t)(vol[p + g.sy] > iso) << 3;
    c |= (uint32_t)(vol[p + 1] > iso) << 4;
    c |= (uint32_t)(vol[p + g.sx + 1] > iso) << 5;
    c |= (uint32_t)(vol[p + g.sx + g.sy + 1] > iso) << 6;
    c |= (uint32_t)(vol[p + g.sy + 1] > iso) << 7;
    return c;
}

__global__ void __launch_bounds__(MC_TPB) k_mc_count(const float* __restrict__ vol, McGeo g, float iso, const uint8_t* __restrict__ tri_count,
                                                     int32_t* __restrict__ cell_tris, int32_t* __restrict__ point_verts, uint8_t* __restrict__ point_mask) {
    const uint32_t p = blockIdx.x * MC_TPB + threadIdx.x;
    if (p >= g.nx * g.ny * g.nz) return;
    const uint32_t z = p % g.nz, y = (p / g.nz) % g.ny, x = p / g.sx;
    const uint32_t m = mc_mask(vol, g, p, x, y, z, iso);
    point_mask[p] = (uint8_t)m;
    point_verts[p] = (int32_t)__popc(m);
    const bool cell = x + 1 < g.nx && y + 1 < g.ny && z + 1 < g.nz;
    cell_tris[p] = cell ? (int32_t)tri_count[mc_case(vol, g, p, iso)] : 0;
}

__global__ void __launch_bounds__(MC_TPB) k_mc_emit(const float* __restrict__ vol, McGeo g, float iso, const uint8_t* __restrict__ tri_count,
                                                    const int8_t* __restrict__ tri_edges, const int32_t* __restrict__ tri_off,
                                                    const int32_t* __restrict__ vert_off, const uint8_t* __restrict__ point_mask,
                                                    float* __restrict__ vertices, int32_t* __restrict__ triangles) {
    const uint32_t p = blockIdx.x * MC_TPB + threadIdx.x;
    if (p >= g.nx * g.ny * g.nz) return;
    const uint32_t z = p % g.nz, y = (p / g.nz) % g.ny, x = p / g.sx;
    const uint32_t m = point_mask[p];
    if (m) {                                                             // this point's vertices, axes in ascending order
        uint32_t v = (uint32_t)vert_off[p] - __popc(m);
        const float a = vol[p];
        const uint32_t step[3] = {g.sx, g.sy, 1u};
#pragma unroll
        for (int axis = 0; axis < 3; ++axis) {
            if (!(m & (1u << axis))) continue;
            const float b = vol[p + step[axis]];
            const float t = (iso - a) / (b - a);                         // linear interpolation along the edge (IEEE division: matches the host walker)
            float pos[3] = {(float)x, (float)y, (float)z};
            pos[axis] += t;
            vertices[3ull * v + 0] = pos[0]; vertices[3ull * v + 1] = pos[1]; vertices[3ull * v + 2] = pos[2];
            ++v;
        }
    }
    if (!(x + 1 < g.nx && y + 1 < g.ny && z + 1 < g.nz)) return;
    const uint32_t c = mc_case(vol, g, p, iso);
    const uint32_t nt = tri_count[c];
    if (nt == 0) return;
    // owner lattice point (offset from p) and axis of each cube edge: ssdnerf_amd/mesh.py EDGE_OWNER
    const uint32_t own_off[12] = {0u, g.sx, g.sy, 0u, 1u, g.sx + 1u, g.sy + 1u, 1u, 0u, g.sx, g.sx + g.sy, g.sy};
    constexpr uint32_t own_axis[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
    uint32_t t0 = (uint32_t)tri_off[p] - nt;
    for (uint32_t t = 0; t < nt; ++t) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t e = (uint32_t)tri_edges[c * 15u + 3u * t + k];
            const uint32_t q = p + own_off[e];
            const uint32_t mq = point_mask[q];
            const uint32_t rank = __popc(mq & ((1u << own_axis[e]) - 1u));
            triangles[3ull * (t0 + t) + k] = (int32_t)((uint32_t)vert_off[q] - __popc(mq) + rank);
        }
    }
}

}  // namespace

extern "C" int ssdnerf_marching_cubes_count(const float* volume, uint32_t nx, uint32_t ny, uint32_t nz, float iso, const uint8_t* tri_count,
                                            int32_t* cell_tris, int32_t* point_verts, uint8_t* point_mask, void* stream) {
    SSD_REQUIRE(volume && tri_count && cell_tris && point_verts && point_mask, "marching_cubes_count: null pointer");
    const uint64_t n = (uint64_t)nx * ny * nz;
    SSD_REQUIRE(nx >= 2 && ny >= 2 && nz >= 2 && n < (1ull << 31), "marching_cubes_count: need 2 <= n per axis and fewer than 2^31 lattice points");
    const McGeo g = {nx, ny, nz, ny * nz, nz};
    hipLaunchKernelGGL(k_mc_count, dim3(ssd_blocks((uint32_t)n, MC_TPB)), dim3(MC_TPB), 0, (hipStream_t)stream, volume, g, iso, tri_count, cell_tris, point_verts,
                       point_mask);
    SSD_CHECK_LAUNCH("marching_cubes_count");
    return SSDNERF_OK;
}

extern "C" int ssdnerf_marching_cubes_emit(const float* volume, uint32_t nx, uint32_t ny, uint32_t nz, float iso, const uint8_t* tri_count,
                                           const int8_t* tri_edges, const int32_t* tri_offsets, const int32_t* vert_offsets, const uint8_t* point_mask,
                                           float* vertices, int32_t* triangles, void* stream) {
    SSD_REQUIRE(volume && tri_count && tri_edges && tri_offsets && vert_offsets && point_mask && vertices && triangles, "marching_cubes_emit: null pointer");
    const uint64_t n = (uint64_t)nx * ny * nz;
    SSD_REQUIRE(nx >= 2 && ny >= 2 && nz >= 2 && n < (1ull << 31), "marching_cubes_emit: need 2 <= n per axis and fewer than 2^31 lattice points");
    const McGeo g = {nx, ny, nz, ny * nz, nz};
    hipLaunchKernelGGL(k_mc_emit, dim3(ssd_blocks((uint32_t)n, MC_TPB)), dim3(MC_TPB), 0, (hipStream_t)stream, volume, g, iso, tri_count, tri_edges, tri_offsets,
                       vert_offsets, point_mask, vertices, triangles);
    SSD_CHECK_LAUNCH("marching_cubes_emit");
    return SSDNERF_OK;
}
