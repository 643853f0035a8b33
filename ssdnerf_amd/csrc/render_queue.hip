// ssdnerf_amd/csrc/render_queue.hip -- fused eval-branch renderer, two-stage form (the fast path).
//
// Why two stages: on SRN-Cars-like scenes ~90 % of the rays cross the whole [-1,1]^3 box without ever meeting an
// occupied voxel, and the reference's stepping rule has to be replayed probe by probe for them (bit-exact sample
// positions are decided by that float chain).  Profiling the single-kernel version (profiles/r01/a_*) showed the
// wave spending most of its instructions marching empty rays at ~5 % lane utilisation while the lanes that own
// shading work wait.  So:
//
//   stage A  k_view_masks -> k_ray_cull -> k_survivor_march -> k_queue_close (described in front of each kernel below): every ray that
//                          cannot meet an occupied voxel is finished with the background colour (a view-level tile mask, then the conservative
//                          coarse-occupancy scan, then the exact march of the survivors), every ray that does is filed in its scene's hit
//                          queue as (ray id | tail bound << 24, t_first), rays that may still take many samples FIRST.
//   stage B  k_shade_mfma (shade_mfma.hip, the default) or k_shade_queue (below, its VALU-only predecessor): persistent waves; each wave
//                          owns a slice of one scene's hit queue and keeps 64 LIVE hitting
//                          rays, refilling finished lanes from the slice (ballot + mbcnt compaction, no atomics):
//                          gather -> tiny MLP -> composite -> advance to the next occupied sample, all in registers.
//                          blockIdx is mapped so that all workgroups of scene s run on XCD (s mod 8): that XCD's 4 MiB
//                          L2 holds the scene's 1.5 MiB of planes + 32 KiB bitfield.
//
// Both stages use the specialised probe below, valid when cascades == 1 and the grid size is a power of two
// (always true for the renderer's configs: base_volume_renderer.py:113 hard-wires C = 1, grid_size = 64).  It is
// BIT-IDENTICAL to ssd_probe/ssd_skip_empty of common.h (every folded constant is an exact power-of-two scaling);
// tests/test_render_gpu.py checks the fused result against the oracle and against the generic single-kernel path.
// The density bitfield is re-ordered once per call from Morton to linear order (32 KiB per scene) so that the
// per-probe index is two integer multiply-adds instead of a 27-instruction Morton encode.
#include "decode_core.h"

static constexpr unsigned RQ_TPB = 256;
static constexpr unsigned RQ_SLICE = 256;        // hit-queue entries per shading wave
static constexpr unsigned RQ_COARSE_MAX_BYTES = 4096; // coarse occupancy ((H >> RQ_COARSE_LOG2B)^3 bits) staged in LDS by k_ray_cull
static constexpr unsigned RQ_HD_STRIDE = 68;     // floats per LDS row of the per-ray direction term (64 + 4 pad)

struct FastMarch {
    float bound, dt_gamma, dt_min, dt_max;
    float mip_bound, rb;     // cascade 0: min(1, bound) and its reciprocal
    float half_H;            // 0.5 * H   (exact power of two)
    float two_rH;            // 2 / H     (exact power of two)
    float Hm1f;
    uint32_t H, log2H;
};

struct QueueCfg {
    FastMarch m;
    PlaneGeom g;
    float aabb[6];
    float min_near, T_thresh, bg, sat;
    uint32_t N, S, cap;
    uint64_t plane_stride;      // elements per scene
    uint32_t bitfield_stride;   // bytes per scene
    const float* dt_gammas;     // [S] on device or null
    uint8_t* image_u8;          // [S][N][3] or null: the quantised image, written next to the float one (saves the separate quantisation pass)
};

// cell index of v = p * rb + 1: only the upper clamp can bind (v > -1 always, and (int) truncates (-1, 0) to 0 exactly like a clamp at 0)
SSD_DEV int rq_cell(const FastMarch& m, float v) { return (int)fminf(v * m.half_H, m.Hm1f); }

struct FastProbe { float x, y, z, dt; int nx, ny, nz; bool occ; };

// DTG0: dt_gamma == 0 (the uncond render), where clamp(t * 0, dt_min, dt_max) == dt_min: the march step is a constant
#ifndef RQ_RUN_TO_CLOSED
#define RQ_RUN_TO_CLOSED 0                     // (r05) 1: `do t += dt while (t < tt)` with the constant step as ssd_run_to_const (common.h): bit-identical, and no
                                               // faster here (stage A 0.80 ms either way: profiles/r05) -- off
#endif
template <bool DTG0>
SSD_DEV float rq_dt(const FastMarch& m, float t) { return DTG0 ? m.dt_min : ssd_clamp(t * m.dt_gamma, m.dt_min, m.dt_max); }

template <bool DTG0 = false>
SSD_DEV FastProbe rq_probe(const FastMarch& m, const uint8_t* __restrict__ lin_bits, const RayGeom& r, float t) {
    FastProbe p;
    p.x = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dx, r.ox), -m.bound, m.bound);      // == min(bound, max(-bound, v)) for the finite values that occur
    p.y = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dy, r.oy), -m.bound, m.bound);
    p.z = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dz, r.oz), -m.bound, m.bound);
    p.dt = rq_dt<DTG0>(m, t);
    p.nx = rq_cell(m, ssd_fma(p.x, m.rb, 1.0f));
    p.ny = rq_cell(m, ssd_fma(p.y, m.rb, 1.0f));
    p.nz = rq_cell(m, ssd_fma(p.z, m.rb, 1.0f));
    const uint32_t idx = ((((uint32_t)p.nz << m.log2H) + (uint32_t)p.ny) << m.log2H) + (uint32_t)p.nx;
    p.occ = (lin_bits[idx >> 3] >> (idx & 7u)) & 1u;
    return p;
}

// sgn{x,y,z} = 0.5 + 0.5*sign(d): 0 or 1, so  nx + 0.5 + 0.5*sign(dx) == (float)(nx + sgn)  exactly.
template <bool DTG0 = false>
SSD_DEV float rq_skip(const FastMarch& m, const RayGeom& r, const FastProbe& p, float sgx, float sgy, float sgz, float t) {
    const float tx = ssd_fma(ssd_fma((float)p.nx + sgx, m.two_rH, -1.0f), m.mip_bound, -p.x) * r.rdx;
    const float ty = ssd_fma(ssd_fma((float)p.ny + sgy, m.two_rH, -1.0f), m.mip_bound, -p.y) * r.rdy;
    const float tz = ssd_fma(ssd_fma((float)p.nz + sgz, m.two_rH, -1.0f), m.mip_bound, -p.z) * r.rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    if (DTG0 && RQ_RUN_TO_CLOSED) return ssd_run_to_const(m.dt_min, t, tt);      // (r05) the same chain of additions in closed form: common.h
    do {
        t += rq_dt<DTG0>(m, t);
    } while (t < tt);
    return t;
}

// ------------------------------------------------------------------------------------------------
// Morton-ordered bitfield -> linear (z-major, x-minor) bitfield.  One lane per output byte.
__global__ void k_bitfield_linearize(const uint8_t* __restrict__ morton_bits, uint32_t H, uint32_t log2H, uint32_t bytes_per_scene,
                                     uint8_t* __restrict__ lin_bits) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bytes_per_scene) return;
    const uint8_t* src = morton_bits + (uint64_t)blockIdx.y * bytes_per_scene;
    unsigned out = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t lin = b * 8 + i;
        const uint32_t x = lin & (H - 1), y = (lin >> log2H) & (H - 1), z = lin >> (2 * log2H);
        const uint32_t mi = ssd_morton(x, y, z);
        out |= ((src[mi >> 3] >> (mi & 7u)) & 1u) << i;
    }
    lin_bits[(uint64_t)blockIdx.y * bytes_per_scene + b] = (uint8_t)out;
}

// ------------------------------------------------------------------------------------------------
// The bitfield once more, BLOCK-MAJOR: the 64 cells of a 4^3 block in one 64-bit word (bit = (z & 3) << 4 | (y & 3) << 2 | (x & 3), words in linear
// z/y/x block order).  One 8-byte load tells k_survivor_march whether the whole block, the 2^3 sub-block or the cell at a position is occupied.
__global__ void __launch_bounds__(RQ_TPB) k_bitfield_blocks64(const uint8_t* __restrict__ lin_bits_all, uint32_t H, uint32_t log2H, uint32_t bytes_per_scene,
                                                              uint64_t* __restrict__ blocks_all) {
    const uint32_t Hb = H >> 2, lb = log2H - 2, n_blocks = Hb * Hb * Hb;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint8_t* lin = lin_bits_all + (uint64_t)blockIdx.y * bytes_per_scene;
    const uint32_t bx = b & (Hb - 1), by = (b >> lb) & (Hb - 1), bz = b >> (2 * lb);
    uint64_t w = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {                                       // row (z, y) of the block: 4 bits at a bit offset that is a multiple of 4
        const uint32_t bit = ((((4 * bz + (k >> 2)) << log2H) + 4 * by + (k & 3u)) << log2H) + 4 * bx;
        w |= (uint64_t)((lin[bit >> 3] >> (bit & 4u)) & 0xfu) << (4 * k);
    }
    blocks_all[(uint64_t)blockIdx.y * n_blocks + b] = w;
}

// exit parameter of the SZ^3-cell block around probe p, `eps` inside the exit face (see k_survivor_march)
template <int SZ>
SSD_DEV float rq_block_exit(const FastMarch& m, const RayGeom& r, const FastProbe& p, float sgx, float sgy, float sgz, float ex, float ey, float ez, float t) {
    const float tx = (ssd_fma(ssd_fma((float)(p.nx & ~(SZ - 1)) + (float)SZ * sgx, m.two_rH, -1.0f), m.mip_bound, -p.x) - ex) * r.rdx;
    const float ty = (ssd_fma(ssd_fma((float)(p.ny & ~(SZ - 1)) + (float)SZ * sgy, m.two_rH, -1.0f), m.mip_bound, -p.y) - ey) * r.rdy;
    const float tz = (ssd_fma(ssd_fma((float)(p.nz & ~(SZ - 1)) + (float)SZ * sgz, m.two_rH, -1.0f), m.mip_bound, -p.z) - ez) * r.rdz;
    return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
}

// ------------------------------------------------------------------------------------------------
// Conservative coarse occupancy: one bit per block of B^3 cells (B = 2^RQ_COARSE_LOG2B), set if ANY cell of the block dilated by B/2 cells
// is occupied.  k_ray_cull walks it with points RQ_COARSE_STEP = B - 0.1 cells apart: every point q of the segment is then within
// (B - 0.1)/2 cells, per axis, of a test point p, so the cell of q -- and the neighbour the reference's fp32 rounding may pick instead --
// has an index within B/2 of p's (the test point's block is taken from ITS exact cell index, >> LOG2B).  If every point lands in a clear
// cell of the ray is occupied, so the exact march would test nothing but empty cells and need not run at all (see k_ray_cull).
static constexpr int RQ_COARSE_B = 1 << RQ_COARSE_LOG2B;                 // (common.h)
static constexpr int RQ_COARSE_DILATE = RQ_COARSE_B / 2;
static constexpr float RQ_COARSE_STEP = SSD_COARSE_STEP;                 // in cells

// Eight lanes per coarse block, one z-range each (r02: one lane per block ran 512 bit tests in series on 128 blocks in all -- 50 us per call).
__global__ void __launch_bounds__(RQ_TPB) k_bitfield_coarsen(const uint8_t* __restrict__ lin_bits_all, uint32_t H, uint32_t log2H, uint32_t bytes_per_scene,
                                                              uint8_t* __restrict__ coarse_all) {
    const uint32_t Hc = H >> RQ_COARSE_LOG2B, log2Hc = log2H - RQ_COARSE_LOG2B;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = t >> 3, part = t & 7u;                               // coarse block (x fastest), eighth of its dilated z range
    const uint8_t* lin = lin_bits_all + (uint64_t)blockIdx.y * bytes_per_scene;
    bool occ = false;
    if (i < Hc * Hc * Hc) {
        const int cx = (int)(i & (Hc - 1)), cy = (int)((i >> log2Hc) & (Hc - 1)), cz = (int)(i >> (2 * log2Hc));
        constexpr int B = RQ_COARSE_B, D = RQ_COARSE_DILATE, SPAN = B + 2 * D, PER = (SPAN + 7) / 8;
        const int x0 = max(B * cx - D, 0), x1 = min(B * cx + B - 1 + D, (int)H - 1);
        const int zb = B * cz - D + (int)part * PER;
        for (int z = max(zb, 0); z <= min(min(zb + PER - 1, B * cz + B - 1 + D), (int)H - 1); ++z)
            for (int y = max(B * cy - D, 0); y <= min(B * cy + B - 1 + D, (int)H - 1); ++y)
                for (int x = x0; x <= x1; ++x) {
                    const uint32_t idx = ((((uint32_t)z << log2H) + (uint32_t)y) << log2H) + (uint32_t)x;
                    occ |= (lin[idx >> 3] >> (idx & 7u)) & 1u;
                }
    }
    const uint64_t any = __ballot(occ);                                      // 8 consecutive lanes = one coarse block: 8 blocks = one byte per wave
    if ((threadIdx.x & 63) == 0 && i < Hc * Hc * Hc) {
        uint32_t byte = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) byte |= ((any >> (8 * k)) & 0xffull) ? (1u << k) : 0u;
        coarse_all[(uint64_t)blockIdx.y * (Hc * Hc * Hc / 8) + (i >> 3)] = (uint8_t)byte;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage A: first occupied sample of every ray, as two dense passes.
//
//   k_ray_cull        one lane per ray.  Ray from the arrays or generated from the view's camera (RaySrc), slab test, then the conservative
//                     EMPTY-SPACE PRE-TEST: the reference's march only ever tests the cell that contains a point o + t d of the ray
//                     (t in [near, far)), in fp32, i.e. a cell within one cell of the true segment.  The lane samples the segment every
//                     RQ_COARSE_STEP cells and looks each sample up in the dilated coarse bitfield (LDS); if all samples are clear, every
//                     cell the march could test is empty, the ray has zero samples whatever its stepping sequence, and the background is
//                     written at once (65 % of the bench's rays).  Otherwise the ray is appended -- one atomic per wave -- to the scene's
//                     SURVIVOR list as (ray id | tail bound << 24): past the last non-clear test point no occupied cell can be met.
//   k_survivor_march  one lane per SURVIVOR: the exact march from `near` to the first occupied probe (hit -> the scene's hit queue) or to
//                     the tail bound (background).  The pre-test never advances a ray, so a marched ray keeps the reference's stepping
//                     sequence.  r01 ran both phases in one kernel, where the march executed at ~22 live lanes of 64 (survivors are the
//                     silhouette band of each view) and was 2/3 of the kernel's VALU issue; compacted through the list it runs full waves.
// ---- view-level cull (cameras only).  Every sample of a ray lies in an occupied cell, hence inside a SET coarse block; a pinhole ray through a
// pixel meets such a block only if the pixel lies inside the bounding rectangle of the block's eight projected corners (all in front of the
// camera).  One block of 256 threads per (view, scene) projects every set coarse block -- grown by one cell -- and marks the 16 x 16 image tiles its
// rectangle (grown by 1.5 pixels) touches; k_ray_cull then writes the background for pixels in unmarked tiles without generating their rays.
// The margins cover the fp32 differences between this test and the rays the kernels build; what it removes are rays the coarse pre-test would
// have finished with the same outputs.  A view whose pose is not rigid, or with a set block that is not entirely in front of the camera,
// gets a full mask (no cull).
// r03: the same pass records, per tile, the CAMERA-DEPTH RANGE of the (grown) set blocks that project into it.  Camera depth is linear along a ray
// (z_c(o + t d) = t (e_z . d), e_z the camera's viewing axis in world coordinates), so a ray of that tile can test an occupied cell only for
// t in [z_lo, z_hi] / (e_z . d): k_ray_cull scans its test points inside that range instead of from the box's near to its far face
// (a car fills about a third of the box's depth).  Stored as two bf16, rounded outwards.
__global__ void __launch_bounds__(RQ_TPB) k_view_masks(FastMarch m, RaySrc src, uint32_t views_cap, uint32_t zr_cap, const uint8_t* __restrict__ coarse_all,
                                                        uint32_t* __restrict__ view_masks, uint32_t* __restrict__ view_zr) {
    __shared__ uint32_t mask[8];
    __shared__ uint32_t give_up;
    __shared__ uint32_t z_lo[256], z_hi[256];                                // fp32 bit patterns of positive depths: unsigned order == float order
    z_lo[threadIdx.x] = 0xffffffffu; z_hi[threadIdx.x] = 0u;
    const uint32_t view = blockIdx.x, scene = blockIdx.y;
    const uint32_t Hc = m.H >> RQ_COARSE_LOG2B, log2Hc = m.log2H - RQ_COARSE_LOG2B, n_blocks = Hc * Hc * Hc;
    if (threadIdx.x < 8) mask[threadIdx.x] = 0u;
    if (threadIdx.x == 0) give_up = 0u;
    __syncthreads();
    const float* M = src.c2w + ((uint64_t)scene * src.V + view) * 16;
    const float* K = src.intr + ((uint64_t)scene * src.V + view) * 4;
    const float c00 = ssd_fma(M[0], M[0], ssd_fma(M[4], M[4], M[8] * M[8])), c11 = ssd_fma(M[1], M[1], ssd_fma(M[5], M[5], M[9] * M[9])),
                c22 = ssd_fma(M[2], M[2], ssd_fma(M[6], M[6], M[10] * M[10])), c01 = ssd_fma(M[0], M[1], ssd_fma(M[4], M[5], M[8] * M[9])),
                c02 = ssd_fma(M[0], M[2], ssd_fma(M[4], M[6], M[8] * M[10])), c12 = ssd_fma(M[1], M[2], ssd_fma(M[5], M[6], M[9] * M[10]));
    const bool rigid = fabsf(c00 - 1.0f) < 1e-4f && fabsf(c11 - 1.0f) < 1e-4f && fabsf(c22 - 1.0f) < 1e-4f && fabsf(c01) < 1e-4f && fabsf(c02) < 1e-4f &&
                       fabsf(c12) < 1e-4f && K[0] > 0.0f && K[1] > 0.0f;
    const uint32_t h = src.hw / src.w, tw = (src.w + 15u) / 16u, th = (h + 15u) / 16u;          // tile size in pixels: always 16 x 16 tiles per view
    const float cell = m.two_rH * m.mip_bound, blockw = cell * (float)RQ_COARSE_B;
    const uint8_t* coarse = coarse_all + (uint64_t)scene * (n_blocks >> 3);
    bool bad = !rigid;
    for (uint32_t i = threadIdx.x; i < n_blocks && !bad; i += RQ_TPB) {
        if (!((coarse[i >> 3] >> (i & 7u)) & 1u)) continue;
        const uint32_t bx = i & (Hc - 1), by = (i >> log2Hc) & (Hc - 1), bz = i >> (2 * log2Hc);
        const float lo[3] = {(float)bx * blockw - m.mip_bound - cell, (float)by * blockw - m.mip_bound - cell, (float)bz * blockw - m.mip_bound - cell};
        const float span = blockw + 2.0f * cell;
        float ulo = 3.0e38f, uhi = -3.0e38f, vlo = 3.0e38f, vhi = -3.0e38f, zmin = 3.0e38f, zmax = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float rx = lo[0] + ((k & 1) ? span : 0.0f) - M[3], ry = lo[1] + ((k & 2) ? span : 0.0f) - M[7], rz = lo[2] + ((k & 4) ? span : 0.0f) - M[11];
            const float xc = ssd_fma(M[0], rx, ssd_fma(M[4], ry, M[8] * rz)), yc = ssd_fma(M[1], rx, ssd_fma(M[5], ry, M[9] * rz)),
                        zc = ssd_fma(M[2], rx, ssd_fma(M[6], ry, M[10] * rz));           // camera coordinates: R^T (X - o)
            if (!(zc > 1e-3f)) { bad = true; break; }
            const float u = ssd_fma(K[0], xc / zc, K[2]), v = ssd_fma(K[1], yc / zc, K[3]);   // pixel-centre coordinates (pixel p has centre p + 0.5)
            ulo = fminf(ulo, u); uhi = fmaxf(uhi, u); vlo = fminf(vlo, v); vhi = fmaxf(vhi, v);
            zmin = fminf(zmin, zc); zmax = fmaxf(zmax, zc);
        }
        if (bad) break;
        // pixels whose centre lies in [ulo - 1.5, uhi + 1.5]: p + 0.5 >= ulo - 1.5  ->  p >= ulo - 2
        const float p0 = floorf(ulo - 2.0f), p1 = ceilf(uhi + 1.0f), q0 = floorf(vlo - 2.0f), q1 = ceilf(vhi + 1.0f);
        if (p1 < 0.0f || q1 < 0.0f || p0 > (float)(src.w - 1u) || q0 > (float)(h - 1u)) continue;
        const uint32_t tx0 = (uint32_t)fmaxf(p0, 0.0f) / tw, tx1 = min((uint32_t)fminf(p1, (float)(src.w - 1u)) / tw, 15u);
        const uint32_t ty0 = (uint32_t)fmaxf(q0, 0.0f) / th, ty1 = min((uint32_t)fminf(q1, (float)(h - 1u)) / th, 15u);
        const uint32_t row = ((2u << tx1) - 1u) & ~((1u << tx0) - 1u);       // bits tx0 .. tx1 of a 16-bit tile row
        for (uint32_t ty = ty0; ty <= ty1; ++ty) {
            atomicOr(&mask[ty >> 1], row << ((ty & 1u) * 16u));
            for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                atomicMin(&z_lo[ty * 16u + tx], __float_as_uint(zmin));
                atomicMax(&z_hi[ty * 16u + tx], __float_as_uint(zmax));
            }
        }
    }
    if (bad) give_up = 1u;
    __syncthreads();
    if (threadIdx.x < 8) view_masks[((uint64_t)scene * views_cap + view) * 8 + threadIdx.x] = give_up ? 0xffffffffu : mask[threadIdx.x];
    {   // bf16 halves, rounded outwards; no cull -> [0, inf)
        const uint32_t lo16 = give_up ? 0u : (z_lo[threadIdx.x] >> 16), hi16 = give_up ? 0x7f80u : min((z_hi[threadIdx.x] >> 16) + 1u, 0x7f80u);
        view_zr[((uint64_t)scene * zr_cap + view) * 256 + threadIdx.x] = lo16 | (hi16 << 16);
    }
}

struct CullGrid { uint32_t group; uint32_t views_cap; uint32_t zr_cap; int32_t tile_w_shift, tile_h_shift; };   // shifts: log2 of the tile size in pixels, or -1   // rays per blockIdx.y group: hw (one view per y, cameras) or N (arrays, gridDim.y == 1)
#ifndef RQ_LONG_STEPS
#define RQ_LONG_STEPS 48                       // a hitting ray whose remaining segment (first hit .. tail bound) is longer than this many minimum steps goes to the FRONT of the queue
#endif
// (r05, measured and removed: writing the background of culled pixels as dense 16-byte stores -- per wave for 8 x 8-pixel waves that lie entirely in unmarked
// tiles, or per block for whole bands of 8 image rows whose tile row has no marked tile -- instead of 8 stores of 4-byte / 1-byte pieces per lane.  Bit-identical and
// no faster, stage A 0.812 / 0.817 against 0.799 / 0.807 ms and 0.722 - 0.725 against 0.716 - 0.721: k_ray_cull is not bound by its stores but by the rays of the MARKED
// tiles (ray generation, slab test, coarse scan).  profiles/r05/y_cull_wave_fill_ab.txt.)
#ifndef RQ_CULL_CHUNKS
#define RQ_CULL_CHUNKS 8
#endif
#ifndef RQ_MARCH_CHUNKS
#define RQ_MARCH_CHUNKS 1                      // (r05; 8 = k_ray_cull's until then) a block walks its chunks one after the other and a chunk lasts as long as its slowest ray: 2048-ray
#endif                                         // blocks were 2 441 blocks on 2 048 slots, a second mostly empty round.  Stage A 0.797 -> 0.711 ms; 16 / 4 / 2 chunks 0.93 / 0.76 / 0.76;
                                               // blocks of 128 / 64 threads 0.85 / 1.11 (profiles/r05/z_march_block_granularity_ab.txt)
static constexpr unsigned RQ_CHUNKS_DEFAULT = RQ_CULL_CHUNKS;   // 256-ray chunks per block of k_ray_cull: the block stages its list in LDS and reserves global slots ONCE
static constexpr unsigned RQ_CHUNKS_SMALL = 1;                  // ... with SSDNERF_FIRST_HIT_SMALL_BLOCKS (r06): 256-ray blocks with 3.6 KB of LDS (the coarse bitfield of a 64^3 grid is 512 B) instead
static constexpr unsigned RQ_COARSE_SMALL = 512;                // of 2048 rays and 21 KB, so that THREE blocks find room on a CU whose LDS the shading kernel of the previous render holds (2 x 74 KB
                                                                // of 160) -- with 9 KB (two chunks, 4 KB for the coarse bits) only one did and the kernel took the whole 4.4 ms of the shading
                                                                // kernel beside it, leaving k_survivor_march exposed behind it (profiles/r06/i_pipeline_*.txt)
static constexpr unsigned RQ_MCHUNKS = RQ_MARCH_CHUNKS;      // ... and of k_survivor_march
#ifndef RQ_MARCH_TPB
#define RQ_MARCH_TPB 256
#endif
static constexpr unsigned RQ_MTPB = RQ_MARCH_TPB;            // threads per block of k_survivor_march

// Appends `item` of every lane with `take` to the block's LDS list (one LDS atomic per wave).
template <typename T>
SSD_DEV void rq_lds_append(bool take, const T& item, T* list, uint32_t* list_count) {
    const uint64_t m = __ballot(take);
    if (m == 0) return;
    uint32_t base = 0;
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) base = atomicAdd(list_count, (uint32_t)__popcll(m));
    base = __shfl(base, __builtin_ctzll(m), 64);
    if (take) list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = item;
}
// Copies the block's LDS list to its global array behind ONE atomic reservation (coalesced stores).  Ends with every thread past the barrier.
template <typename T>
SSD_DEV void rq_flush(const T* list, const uint32_t* list_count, uint32_t* slot /* LDS */, uint32_t* global_counter, T* global_list) {
    __syncthreads();
    const uint32_t n = *list_count;
    if (threadIdx.x == 0 && n != 0) *slot = atomicAdd(global_counter, n);
    __syncthreads();
    const uint32_t base = *slot;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) global_list[base + i] = list[i];
}

template <unsigned RQ_CHUNKS, unsigned RQ_COARSE_LDS>
__global__ void __launch_bounds__(RQ_TPB) k_ray_cull(QueueCfg c, RaySrc src, CullGrid cg, const uint8_t* __restrict__ coarse_bits,
                                                      float* __restrict__ image, float* __restrict__ depth, float* __restrict__ weights_sum,
                                                      int32_t* __restrict__ sample_counts, uint2* __restrict__ survivors,
                                                      uint32_t* __restrict__ counters, const uint32_t* __restrict__ view_masks,
                                                      const uint32_t* __restrict__ view_zr) {
    const uint32_t scene = blockIdx.z;
    __shared__ uint32_t tile_mask[8];                                         // this block's view (cameras: one view per blockIdx.y), k_view_masks
    __shared__ uint32_t tile_zr[256];                                         // ... and its tiles' depth ranges
    const bool view_cull = view_masks != nullptr;
    if (view_cull && threadIdx.x < 8) tile_mask[threadIdx.x] = view_masks[((uint64_t)scene * cg.views_cap + blockIdx.y) * 8 + threadIdx.x];
    if (view_cull) tile_zr[threadIdx.x] = view_zr[((uint64_t)scene * cg.zr_cap + blockIdx.y) * 256 + threadIdx.x];
    const uint32_t tile_w = view_cull ? (src.w + 15u) / 16u : 1u, tile_h = view_cull ? (src.hw / src.w + 15u) / 16u : 1u;
    const bool tiled = view_cull && (src.w & 7u) == 0 && ((src.hw / src.w) & 7u) == 0;
    const bool pow2 = src.w_shift >= 3 && cg.tile_w_shift >= 0 && cg.tile_h_shift >= 0;          // power-of-two view and tile sizes: shifts instead of divisions
    // this scene's coarse bitfield -> LDS ((H/4)^3 bits; 512 B for H = 64)
    __shared__ __attribute__((aligned(16))) uint8_t coarse_lds[RQ_COARSE_LDS];
    __shared__ uint2 list[RQ_CHUNKS * RQ_TPB];
    __shared__ uint32_t list_count, slot;
    const uint32_t Hc = c.m.H >> RQ_COARSE_LOG2B, log2Hc = c.m.log2H - RQ_COARSE_LOG2B, coarse_bytes = (Hc * Hc * Hc) >> 3;
    const bool use_coarse = coarse_bits != nullptr;
    if (threadIdx.x == 0) list_count = 0;
    if (use_coarse) {
        const uint4* cb = reinterpret_cast<const uint4*>(coarse_bits + (uint64_t)scene * coarse_bytes);
        for (uint32_t i = threadIdx.x; i < coarse_bytes / 16; i += RQ_TPB) reinterpret_cast<uint4*>(coarse_lds)[i] = cb[i];
    }
    __syncthreads();
    const bool packing = c.N <= SSD_RAY_ID_MASK + 1u;
#pragma unroll 1
    for (uint32_t chunk = 0; chunk < RQ_CHUNKS; ++chunk) {
        uint32_t in_group = (blockIdx.x * RQ_CHUNKS + chunk) * RQ_TPB + threadIdx.x;
        uint32_t px = 0, py = 0;
        if (tiled && in_group < cg.group) {      // a wave takes an 8 x 8 pixel block of the view instead of 64 pixels of a row: whole waves fall in unmarked tiles
            const uint32_t blk = in_group >> 6, l = in_group & 63u, bpr = src.w >> 3;
            const uint32_t by = pow2 ? blk >> (src.w_shift - 3) : blk / bpr, bx = blk - by * bpr;       // (runtime divisions were ~100 of the ~190 instructions of a ray here)
            py = by * 8u + (l >> 3); px = bx * 8u + (l & 7u);
            in_group = py * src.w + px;
        }
        const uint32_t n = blockIdx.y * cg.group + in_group;
        const uint64_t gi = (uint64_t)scene * c.N + n;
        bool alive = false;
        uint32_t tail = SSD_TAIL_NONE;
        float t_start = 0.f;
        if (in_group < cg.group && n < c.N) {
            RayGeom r = {};
            bool outside = false;
            float t_zlo = 0.0f, t_zhi = 3.0e38f;                             // the tile's depth range as ray parameters (view cull only)
            if (src.c2w != nullptr) {       // the view is uniform over the block (blockIdx.y): pose and intrinsics are scalar loads
                const uint64_t cam = (uint64_t)scene * src.V + blockIdx.y;
                if (!tiled) {
                    py = src.w_shift >= 0 ? in_group >> src.w_shift : in_group / src.w;
                    px = in_group - py * src.w;
                }
                uint32_t tile = 0;
                if (view_cull) {
                    tile = pow2 ? ((py >> cg.tile_h_shift) << 4) + (px >> cg.tile_w_shift) : (py / tile_h) * 16u + px / tile_w;
                    outside = !((tile_mask[tile >> 5] >> (tile & 31u)) & 1u);
                }
                if (!outside) {
                    float o[3], d[3];
                    ssd_cam_ray(src.c2w + cam * 16, src.intr + cam * 4, px, py, o, d);
                    r = ssd_ray_geom(o[0], o[1], o[2], d[0], d[1], d[2]);
                    if (view_cull) {
                        const uint32_t zr = tile_zr[tile];
                        const float* M = src.c2w + cam * 16;
                        const float ez_d = ssd_fma(M[2], d[0], ssd_fma(M[6], d[1], M[10] * d[2]));     // z_c of the point o + t d is t * ez_d (k_view_masks' own formula)
                        if (ez_d > 1e-6f) {
                            const float rz = 1.0f / ez_d;
                            t_zlo = __uint_as_float(zr << 16) * rz * 0.998f;                        // (rounding of the division / of k_view_masks' products: far below the margin)
                            t_zhi = __uint_as_float(zr & 0xffff0000u) * rz * 1.002f;
                        }
                    }
                }
            } else {
                r = ssd_load_ray(src.rays_o + 3 * gi, src.rays_d + 3 * gi);
            }
            float t = 0.f, far_ = 0.f;
            if (!outside) {
                ssd_near_far(c.aabb, r, c.min_near, t, far_);
                alive = t < far_;
                t_start = t;
            }
            if (use_coarse && alive) {
                const float step_t = ssd_coarse_step_t(r, c.m.two_rH * c.m.mip_bound);          // RQ_COARSE_STEP cells of world length, in t
                // The test points run in COARSE-BLOCK coordinates, advanced by one add per axis: q(u) = ((o + u d) rb + 1) half_H / B.  The
                // accumulated rounding of <= 128 adds on |q| <= H/B is < 1e-3 cell, far inside the 0.05-cell margin RQ_COARSE_STEP = B - 0.1
                // leaves on either side (a test point only has to lie within that margin of the ray; it takes no part in the exact march).
                const float hb = c.m.half_H * (1.0f / (float)(1 << RQ_COARSE_LOG2B));
                float qx = ssd_fma(ssd_fma(t, r.dx, r.ox), c.m.rb, 1.0f) * hb, qy = ssd_fma(ssd_fma(t, r.dy, r.oy), c.m.rb, 1.0f) * hb,
                      qz = ssd_fma(ssd_fma(t, r.dz, r.oz), c.m.rb, 1.0f) * hb;
                const float sx = step_t * r.dx * c.m.rb * hb, sy = step_t * r.dy * c.m.rb * hb, sz = step_t * r.dz * c.m.rb * hb;
                const uint32_t top = Hc - 1;
                const float topf = (float)top;
                auto occupied = [&](float x, float y, float z) {
                    const uint32_t bx = (uint32_t)__builtin_amdgcn_fmed3f(x, 0.0f, topf), by = (uint32_t)__builtin_amdgcn_fmed3f(y, 0.0f, topf),
                                   bz = (uint32_t)__builtin_amdgcn_fmed3f(z, 0.0f, topf);
                    const uint32_t ci = (((bz << log2Hc) + by) << log2Hc) + bx;
                    return ((coarse_lds[ci >> 3] >> (ci & 7u)) & 1u) != 0;
                };
                int j_last = -1, j_first = -1, j = 0;
                float tc = t;
                // test points in front of / behind the tile's depth range lie in no set block (a set block the ray passes through projects onto
                // the ray's pixel, hence into its tile, and is part of the range): they are clear without being looked up.  One step of slack each side.
                if (t_zlo > t + 2.0f * step_t) {
                    j = (int)((t_zlo - t) / step_t) - 1;
                    const float fj = (float)j;
                    tc = ssd_fma(fj, step_t, t); qx = ssd_fma(fj, sx, qx); qy = ssd_fma(fj, sy, qy); qz = ssd_fma(fj, sz, qz);
                }
                const float t_scan_end = fminf(far_, t_zhi + step_t);
                for (; tc < t_scan_end; tc += step_t, ++j) {                      // test points near, near + step, ... below far
                    if (occupied(qx, qy, qz)) { j_last = j; if (j_first < 0) j_first = j; }
                    qx += sx; qy += sy; qz += sz;
                }
                // ... and the segment's end point itself (when the range reaches it)
                if (t_scan_end >= far_ &&
                    occupied(ssd_fma(ssd_fma(far_, r.dx, r.ox), c.m.rb, 1.0f) * hb, ssd_fma(ssd_fma(far_, r.dy, r.oy), c.m.rb, 1.0f) * hb,
                             ssd_fma(ssd_fma(far_, r.dz, r.oz), c.m.rb, 1.0f) * hb)) j_last = j;
                alive = j_last >= 0;                                             // nothing within a cell of this ray: no march
                // every test point after j_last is clear: past near + (j_last + 1) steps no cell the march could test is occupied, so the
                // march (k_survivor_march and the shading kernel, via ssd_tail_far) may stop there; it still starts at `near`
                if (alive && packing && j_last < (int)SSD_TAIL_NONE) tail = (uint32_t)j_last;
                // HEAD SKIP (r03), the mirror image of the tail bound: every test point before j_first is clear, so up to near + (j_first - 1) steps
                // (- 0.5 step of slack, as for the tail) no cell the march could test is occupied.  The reference's march visits the parameters
                // t_0 = near, t_(k+1) = t_k + dt(t_k) -- a sequence that does not depend on the cells (in an empty cell it runs the same additions up
                // to the cell's exit) -- and its first sample is the first t_k whose cell is occupied.  So k_survivor_march may run those additions
                // WITHOUT probing while t_k < head (the survivor entry carries `head`): its probes then start at an exact member of the sequence
                // ~10 cells in front of the object instead of ~45 cells away at the box (k_survivor_march 0.68 -> 0.41 ms, profiles/r03).
                if (alive && j_first >= 2) t_start = fminf(ssd_fma((float)j_first - 1.5f, step_t, t), far_);
            }
            if (!alive) {  // misses the box, or nothing within a cell of the ray: background only
                image[3 * gi + 0] = c.bg; image[3 * gi + 1] = c.bg; image[3 * gi + 2] = c.bg;
                if (c.image_u8) { const uint8_t q8 = ssd_quant_u8(c.bg); c.image_u8[3 * gi + 0] = q8; c.image_u8[3 * gi + 1] = q8; c.image_u8[3 * gi + 2] = q8; }
                depth[gi] = 0.f; weights_sum[gi] = 0.f;
                if (sample_counts) sample_counts[gi] = 0;
            }
        }
        rq_lds_append(alive, make_uint2(packing ? (n | (tail << 24)) : n, __float_as_uint(t_start)), list, &list_count);
    }
    rq_flush(list, &list_count, &slot, counters + ssd_counter(SSD_CNT_SURVIVORS, c.S, scene), survivors + (uint64_t)scene * c.N);
}

// (r03, measured and dropped: staging the scene's 32 KiB linear bitfield in LDS per block -- 48 KiB of LDS, 3 blocks per CU instead of 8 -- made the
// step 0.17 ms SLOWER, 6.62 against 6.45 ms on one box: the probes are L1 / L2 hits and eight waves per SIMD hide them better than three with
// LDS-latency probes.)
template <bool DTG0>
__global__ void __launch_bounds__(RQ_MTPB) k_survivor_march(QueueCfg c, RaySrc src, const uint8_t* __restrict__ lin_bits,
                                                            const uint64_t* __restrict__ blocks64 /* k_bitfield_blocks64 or null */,
                                                            const uint2* __restrict__ survivors, float* __restrict__ image,
                                                            float* __restrict__ depth, float* __restrict__ weights_sum,
                                                            int32_t* __restrict__ sample_counts, uint2* __restrict__ queue,
                                                            uint32_t* __restrict__ counters, uint8_t* __restrict__ qkey /* or null */, uint32_t key_stride) {
    const uint32_t scene = blockIdx.y;
    const uint32_t count = counters[ssd_counter(SSD_CNT_SURVIVORS, c.S, scene)];
    if (blockIdx.x * (RQ_MCHUNKS * RQ_MTPB) >= count) return;                  // the grid covers the worst case (every ray survives)
    __shared__ uint2 list[RQ_MCHUNKS * RQ_MTPB];                                // long rays from the front, short rays from the back
    __shared__ uint8_t keys[RQ_MCHUNKS * RQ_MTPB];                              // (ticket order) per list entry: upper bound of the remaining march steps
    __shared__ uint32_t list_count, short_count, slot, slot_short;
    if (threadIdx.x == 0) { list_count = 0; short_count = 0; }
    __syncthreads();
    lin_bits += (uint64_t)scene * c.bitfield_stride;
    const uint32_t lb = c.m.log2H - 2;
    if (blocks64) blocks64 += (uint64_t)scene * (c.bitfield_stride >> 3);
    const float blk_eps = c.m.two_rH * c.m.mip_bound * (1.0f / 4096.0f);      // 2^-12 cell: >> the fp32 error of a position (<= 3e-5 cell up to H = 512)
    if (c.dt_gammas) c.m.dt_gamma = c.dt_gammas[scene];
    const bool packing = c.N <= SSD_RAY_ID_MASK + 1u;
#pragma unroll 1
    for (uint32_t chunk = 0; chunk < RQ_MCHUNKS; ++chunk) {
        const uint32_t i = (blockIdx.x * RQ_MCHUNKS + chunk) * RQ_MTPB + threadIdx.x;
        bool hit = false;
        uint32_t e = 0;
        float t = 0.f, far_b = 0.f;
        if (i < count) {
            const uint2 sv = survivors[(uint64_t)scene * c.N + i];
            e = sv.x;
            const uint32_t n = packing ? (e & SSD_RAY_ID_MASK) : e;
            const uint64_t gi = (uint64_t)scene * c.N + n;
            const RayGeom r = ssd_fetch_ray(src, scene, c.N, n);
            float near_, far_;
            ssd_near_far(c.aabb, r, c.min_near, near_, far_);
            far_ = ssd_tail_far(r, c.m.two_rH * c.m.mip_bound, near_, far_, e, packing);
            far_b = far_;
            const float sgx = ssd_fma(0.5f, ssd_sign1(r.dx), 0.5f), sgy = ssd_fma(0.5f, ssd_sign1(r.dy), 0.5f), sgz = ssd_fma(0.5f, ssd_sign1(r.dz), 0.5f);
            t = near_;
            const float head = __uint_as_float(sv.y);                        // k_ray_cull's head skip: nothing the march could test is occupied before it
            if (DTG0) {                                                      // the march's own parameter sequence, without the probes: four additions per test
                const float dt = c.m.dt_min;
                while (t < head) {
                    const float a = t + dt, b = a + dt, c2 = b + dt, d = c2 + dt;
                    if (c2 < head) { t = d; continue; }                      // (d is the first member that may reach the head)
                    t = a >= head ? a : b >= head ? b : c2;
                    break;
                }
            } else {
                while (t < head) t += rq_dt<DTG0>(c.m, t);
            }
            if (blocks64) {
                // BLOCK SKIP (r03).  The march's parameters t_k do not depend on the cells (see the head skip), and the first sample is the first t_k whose
                // cell is occupied.  If the 4^3-cell block (or its 2^3 sub-block) around the current position holds no occupied cell, every t_k whose
                // position is still inside it -- positions are monotone per axis in t, and `blk_eps` inside the exit face covers their rounding -- tests
                // an empty cell: those parameters are run through without probes (the rays that graze the object, 2/3 of the survivors, walk ~30 cells
                // of its hull).  One 8-byte load per probe serves all three levels.
                const float ex = blk_eps * ssd_sign1(r.dx), ey = blk_eps * ssd_sign1(r.dy), ez = blk_eps * ssd_sign1(r.dz);
                while (t < far_) {
                    FastProbe p;
                    p.x = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dx, r.ox), -c.m.bound, c.m.bound);
                    p.y = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dy, r.oy), -c.m.bound, c.m.bound);
                    p.z = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dz, r.oz), -c.m.bound, c.m.bound);
                    p.nx = rq_cell(c.m, ssd_fma(p.x, c.m.rb, 1.0f)); p.ny = rq_cell(c.m, ssd_fma(p.y, c.m.rb, 1.0f)); p.nz = rq_cell(c.m, ssd_fma(p.z, c.m.rb, 1.0f));
                    const uint32_t bi = (((((uint32_t)p.nz >> 2) << lb) + ((uint32_t)p.ny >> 2)) << lb) + ((uint32_t)p.nx >> 2);
                    const uint64_t w = blocks64[bi];
                    float tt;
                    if (w == 0) {
                        tt = rq_block_exit<4>(c.m, r, p, sgx, sgy, sgz, ex, ey, ez, t);
                    } else {
                        const uint64_t sub = w >> ((((uint32_t)p.nz & 2u) << 4) | (((uint32_t)p.ny & 2u) << 2) | ((uint32_t)p.nx & 2u));
                        if ((sub & 0x00330033ull) == 0) {
                            tt = rq_block_exit<2>(c.m, r, p, sgx, sgy, sgz, ex, ey, ez, t);
                        } else {
                            if ((sub >> ((((uint32_t)p.nz & 1u) << 4) | (((uint32_t)p.ny & 1u) << 2) | ((uint32_t)p.nx & 1u))) & 1ull) { hit = true; break; }
                            t = rq_skip<DTG0>(c.m, r, p, sgx, sgy, sgz, t);
                            continue;
                        }
                    }
                    if (DTG0 && RQ_RUN_TO_CLOSED) t = ssd_run_to_const(c.m.dt_min, t, tt);
                    else do { t += rq_dt<DTG0>(c.m, t); } while (t < tt);
                }
            } else {
                while (t < far_) {
                    const FastProbe p = rq_probe<DTG0>(c.m, lin_bits, r, t);
                    if (p.occ) { hit = true; break; }
                    t = rq_skip<DTG0>(c.m, r, p, sgx, sgy, sgz, t);
                }
            }
            if (!hit) {  // the ray left the object's neighbourhood without a sample: background only
                image[3 * gi + 0] = c.bg; image[3 * gi + 1] = c.bg; image[3 * gi + 2] = c.bg;
                if (c.image_u8) { const uint8_t q8 = ssd_quant_u8(c.bg); c.image_u8[3 * gi + 0] = q8; c.image_u8[3 * gi + 1] = q8; c.image_u8[3 * gi + 2] = q8; }
                depth[gi] = 0.f; weights_sum[gi] = 0.f;
                if (sample_counts) sample_counts[gi] = 0;
            }
        }
        // ticket order (r05, qkey != null): every hit goes to the front part IN ARRIVAL ORDER -- a wave of this kernel is an 8 x 8 pixel block of a
        // view (k_ray_cull), so 64 consecutive entries are neighbouring rays -- with its bound on the steps it can still take; k_ticket_order then
        // sorts the queue's 64-entry SLICES, not the rays.  Otherwise two classes, long rays first (the r02 - r04 form).
        const float steps_left = (far_b - t) / c.m.dt_min;
        const bool is_long = hit && (qkey != nullptr || (far_b - t) > (float)RQ_LONG_STEPS * c.m.dt_min);
        {
            const uint64_t m = __ballot(is_long);
            if (m != 0) {
                uint32_t base = 0;
                if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) base = atomicAdd(&list_count, (uint32_t)__popcll(m));
                base = __shfl(base, __builtin_ctzll(m), 64);
                if (is_long) {
                    const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    list[at] = make_uint2(e, __float_as_uint(t));
                    keys[at] = (uint8_t)fminf(255.0f, fmaxf(steps_left, 0.0f));
                }
            }
        }
        {   // short rays: the same append, growing down from the end of the list
            const uint64_t m = __ballot(hit && !is_long);
            if (m != 0) {
                uint32_t base = 0;
                if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) base = atomicAdd(&short_count, (uint32_t)__popcll(m));
                base = __shfl(base, __builtin_ctzll(m), 64);
                if (hit && !is_long)
                    list[RQ_MCHUNKS * RQ_MTPB - 1u - (base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)))] =
                        make_uint2(e, __float_as_uint(t));
            }
        }
    }
    // ---- the hit queue is filled from BOTH ends: rays that may still take many samples (upper bound: tail bound - first hit, in minimum steps)
    // at the front, the others at the back.  The persistent shading kernel consumes the queue front to back, so the long rays are in flight
    // early and the launch ends on short ones (its tail -- waves draining their last rays with no ticket left -- was 0.77 ms of 6.97 ms with the
    // rays in arrival order).  k_queue_close then moves the back part up against the front part: one contiguous queue again.
    __syncthreads();
    const uint32_t n_long = list_count, n_short = short_count;
    if (threadIdx.x == 0) {
        slot = n_long ? atomicAdd(counters + ssd_counter(SSD_CNT_HITS, c.S, scene), n_long) : 0u;
        slot_short = n_short ? atomicAdd(counters + ssd_counter(SSD_CNT_HITS_SHORT, c.S, scene), n_short) : 0u;
    }
    __syncthreads();
    uint2* q = queue + (uint64_t)scene * c.N;
    for (uint32_t i = threadIdx.x; i < n_long; i += blockDim.x) q[slot + i] = list[i];
    if (qkey != nullptr)
        for (uint32_t i = threadIdx.x; i < n_long; i += blockDim.x) qkey[(uint64_t)scene * key_stride + slot + i] = keys[i];
    for (uint32_t i = threadIdx.x; i < n_short; i += blockDim.x) q[c.N - 1u - (slot_short + i)] = list[RQ_MCHUNKS * RQ_MTPB - 1u - i];
}

// The short-ray entries sit at [N - n_short, N); the final queue is [0, n_long + n_short).  Entries beyond that range move into the holes
// [n_long, N - n_short) (as many holes as such entries; source and destination ranges are disjoint); order inside a class is irrelevant.
__global__ void __launch_bounds__(RQ_TPB) k_queue_close(uint32_t S, uint32_t N, uint2* __restrict__ queue, const uint32_t* __restrict__ counters) {
    const uint32_t scene = blockIdx.y;
    const uint32_t n_long = counters[ssd_counter(SSD_CNT_HITS, S, scene)], n_short = counters[ssd_counter(SSD_CNT_HITS_SHORT, S, scene)];
    const uint32_t moves = min(n_short, N - n_long - n_short);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= moves) return;
    uint2* q = queue + (uint64_t)scene * N;
    q[n_long + k] = q[N - moves + k];
}
__global__ void k_queue_total(uint32_t S, uint32_t* __restrict__ counters) {
    const uint32_t scene = blockIdx.x * blockDim.x + threadIdx.x;
    if (scene < S) counters[ssd_counter(SSD_CNT_HITS, S, scene)] += counters[ssd_counter(SSD_CNT_HITS_SHORT, S, scene)];
}

// Ticket order (r05).  The persistent shading kernel ends when its last wave has drained the rays it still holds after the last ticket; a ray's
// samples are sequential, so that drain lasts as long as the longest ray that was STARTED late.  Two classes (long / short at 48 steps) do not help on
// object scenes -- 98.8 % of the bench's hitting rays take at most 48 samples (tools/ray_length_hist.py), the launch ended 0.3 ms (6 %) after its
// mean wave with or without them -- and a finer sort of RAYS would scatter neighbouring pixels over the queue.  So the queue stays in arrival order
// and its 64-entry slices (= tickets = stage fills) are handed out longest first: key = the largest step bound of the slice's rays, 64 buckets
// of 4 steps, one block per scene (a few thousand slices), counting sort.  order[pos] = slice index; the position inside a bucket is whatever the
// atomics give -- the order of tickets never changes a ray's result.  Also folds the short-class count into the total like k_queue_total.
__global__ void __launch_bounds__(1024) k_ticket_order(uint32_t S, const uint8_t* __restrict__ qkey, uint32_t key_stride, uint32_t* __restrict__ counters,
                                                        uint32_t* __restrict__ order, uint32_t order_stride) {
    const uint32_t scene = blockIdx.x;
    __shared__ uint32_t hist[64], cursor[64];
    const uint32_t count = counters[ssd_counter(SSD_CNT_HITS, S, scene)] + counters[ssd_counter(SSD_CNT_HITS_SHORT, S, scene)];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n_slices = (count + 63u) / 64u;
    const uint8_t* k = qkey + (uint64_t)scene * key_stride;
    auto bucket_of = [&](uint32_t sl) -> uint32_t {
        uint32_t mx = 0;
        const uint32_t n = min(64u, count - sl * 64u);
        if (n == 64u) {
            const uint4* p4 = reinterpret_cast<const uint4*>(k + (uint64_t)sl * 64u);            // 64-byte aligned: key_stride is a multiple of 64
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 v = p4[j];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) mx = max(mx, max(max(w[q] & 0xffu, (w[q] >> 8) & 0xffu), max((w[q] >> 16) & 0xffu, w[q] >> 24)));
            }
        } else {
            for (uint32_t j = 0; j < n; ++j) mx = max(mx, (uint32_t)k[(uint64_t)sl * 64u + j]);
        }
        return 63u - (mx >> 2);                                                                  // bucket 0 = the longest rays
    };
    for (uint32_t sl = threadIdx.x; sl < n_slices; sl += blockDim.x) atomicAdd(&hist[bucket_of(sl)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 64; ++b) { cursor[b] = run; run += hist[b]; }
        counters[ssd_counter(SSD_CNT_HITS, S, scene)] = count;
        if (scene == 0) counters[ssd_counter(SSD_CNT_TICKETS, S, 0) + 1] = 1u;      // "this workspace holds an order table" (r05 advisor): the shading kernel looks HERE, not at its own
    }                                                                              // launch's view of SSDNERF_TICKET_ORDER, before it indexes the table
    __syncthreads();
    for (uint32_t sl = threadIdx.x; sl < n_slices; sl += blockDim.x) order[(uint64_t)scene * order_stride + atomicAdd(&cursor[bucket_of(sl)], 1u)] = sl;
}

// ------------------------------------------------------------------------------------------------
// Stage B: shade the hit queue.
template <typename PT>
__global__ void __launch_bounds__(RQ_TPB) k_shade_queue(QueueCfg c, uint32_t slices_per_scene, const PT* __restrict__ planes,
                                                         const float* __restrict__ P, const uint8_t* __restrict__ lin_bits,
                                                         const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const uint2* __restrict__ queue, const uint32_t* __restrict__ queue_count,
                                                         float* __restrict__ image, float* __restrict__ depth, float* __restrict__ weights_sum,
                                                         int32_t* __restrict__ sample_counts, int32_t* __restrict__ overflow_flag) {
    __shared__ __attribute__((aligned(16))) float hd_lds[(RQ_TPB / 64) * 64 * RQ_HD_STRIDE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup -> (scene, slice block): workgroups are dispatched round-robin over the 8 XCDs, so b % 8 picks the XCD;
    // scene s is served by XCD s % 8 (speed only - correctness does not depend on the placement).
    const uint32_t b = blockIdx.x;
    const uint32_t wg_per_scene = (slices_per_scene + (RQ_TPB / 64) - 1) / (RQ_TPB / 64);
    uint32_t scene, wg;
    if ((c.S & 7u) == 0) {            // S multiple of 8: pin scene s to XCD s % 8
        const uint32_t xcd = b & 7u, j = b >> 3;
        scene = xcd + 8u * (j / wg_per_scene);
        wg = j % wg_per_scene;
    } else {                          // few scenes: let every XCD work on every scene (their planes fit the L2s anyway)
        scene = b / wg_per_scene;
        wg = b % wg_per_scene;
    }
    if (scene >= c.S) return;
    const uint32_t count = queue_count[ssd_counter(SSD_CNT_HITS, c.S, scene)];
    uint32_t next = __builtin_amdgcn_readfirstlane((wg * (RQ_TPB / 64) + wave) * RQ_SLICE);
    if (next >= count) return;
    const uint32_t end = min(next + RQ_SLICE, count);

    const uint64_t ray0 = (uint64_t)scene * c.N;
    planes += scene * c.plane_stride;
    lin_bits += (uint64_t)scene * c.bitfield_stride;
    queue += ray0;
    if (c.dt_gammas) c.m.dt_gamma = c.dt_gammas[scene];

    float* hd_wave = hd_lds + wave * 64 * RQ_HD_STRIDE;
    const float* hd_row = hd_wave + lane * RQ_HD_STRIDE;
    float wd[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) wd[m] = P[MLP_OFF_WD + lane * 16 + m];
    const float bd = P[MLP_OFF_BD + lane];

    int ray = -1;
    RayGeom r = {};
    float sgx = 0.f, sgy = 0.f, sgz = 0.f;
    float t = 0.f, far_ = 0.f, ws = 0.f, dep = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    uint32_t cnt = 0;
    float sx = 0.f, sy = 0.f, sz = 0.f, sdt = 0.f;
    bool fresh = false;

    auto finish = [&]() {
        const uint64_t gi = ray0 + (uint32_t)ray;
        const float bgk = c.bg * (1.0f - ws);
        image[3 * gi + 0] = cr + bgk;
        image[3 * gi + 1] = cg + bgk;
        image[3 * gi + 2] = cb + bgk;
        depth[gi] = dep;
        weights_sum[gi] = ws;
        if (sample_counts) sample_counts[gi] = (int32_t)cnt;
        ray = -1;
    };

    for (;;) {
        // ---- refill idle lanes from this wave's slice of the hit queue -------------------------------
        const uint64_t idle = __ballot(ray < 0);
        if (idle != 0 && next < end) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t cand = next + rank;
            if (ray < 0 && cand < end) {
                const uint2 e = queue[cand];
                ray = (int)(c.N <= SSD_RAY_ID_MASK + 1u ? (e.x & SSD_RAY_ID_MASK) : e.x);       // (the tail bound in the upper bits is ignored here)
                t = __uint_as_float(e.y);
                const uint64_t gi = ray0 + (uint32_t)ray;
                r = ssd_load_ray(rays_o + 3 * gi, rays_d + 3 * gi);
                float near_;
                ssd_near_far(c.aabb, r, c.min_near, near_, far_);
                sgx = ssd_fma(0.5f, ssd_sign1(r.dx), 0.5f); sgy = ssd_fma(0.5f, ssd_sign1(r.dy), 0.5f); sgz = ssd_fma(0.5f, ssd_sign1(r.dz), 0.5f);
                const FastProbe p = rq_probe(c.m, lin_bits, r, t);   // the queued t is an occupied probe by construction
                sx = p.x; sy = p.y; sz = p.z; sdt = p.dt;
                ws = dep = cr = cg = cb = 0.f;
                cnt = 0; fresh = true;
            }
            next = __builtin_amdgcn_readfirstlane(min(next + (uint32_t)__popcll(idle), end));
        }
        if (__ballot(ray >= 0) == 0) break;

        // ---- per-ray direction term dir_net(SH4(d)): once per ray, lane i computes hidden unit i ------
        uint64_t need = __ballot(fresh);
        while (need) {
            const int L = __builtin_ctzll(need);
            need &= need - 1;
            const float dx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.dx), L));
            const float dy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.dy), L));
            const float dz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.dz), L));
            float sh[16];
            shb::eval<4, false>(dx, dy, dz, sh, nullptr, nullptr, nullptr);
            float h = bd;
#pragma unroll
            for (int m = 0; m < 16; ++m) h = ssd_fma(wd[m], sh[m], h);
            hd_wave[L * RQ_HD_STRIDE + lane] = h;
        }
        fresh = false;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- shade one sample per live lane, composite, advance to the next occupied sample -----------
        if (ray >= 0) {
            float f[18];
            ssd_gather18<PT>(planes, c.g, sx, sy, sz, f);
            float sigma, sr, sg, sb;
            ssd_mlp<2>(P, f, nullptr, hd_row, c.sat, sigma, sr, sg, sb);
            const float alpha = 1.0f - __expf(-sigma * sdt);
            const float T = 1.0f - ws;
            const float w = alpha * T;
            ws += w;
            dep = ssd_fma(w, t, dep);
            cr = ssd_fma(w, sr, cr);
            cg = ssd_fma(w, sg, cg);
            cb = ssd_fma(w, sb, cb);
            t += sdt;
            ++cnt;
            if (T < c.T_thresh) {
                finish();
            } else {
                for (;;) {
                    if (!(t < far_)) { finish(); break; }
                    if (cnt >= c.cap) {  // the reference's global step cap would cut this ray: schedule dependent, reported
                        if (overflow_flag) atomicAdd(overflow_flag, 1);
                        finish();
                        break;
                    }
                    const FastProbe p = rq_probe(c.m, lin_bits, r, t);
                    if (p.occ) { sx = p.x; sy = p.y; sz = p.z; sdt = p.dt; break; }
                    t = rq_skip(c.m, r, p, sgx, sgy, sgz, t);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
static bool rq_is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

static int rq_make_cfg(QueueCfg& c, uint32_t Hp, uint32_t Wp, uint32_t grid_size, uint32_t S, uint32_t N, float bound, float min_near,
                       float dt_gamma, const float* dt_gammas, uint32_t max_steps, float T_thresh, float bg_color, float sat) {
    SSD_REQUIRE(rq_is_pow2(grid_size) && grid_size >= 8 && grid_size <= 512, "render (queued): grid_size must be a power of two in [8, 512]");
    const MarchCfg mc = ssd_make_march_cfg(bound, dt_gamma, max_steps, 1, grid_size, nullptr);
    c.m.bound = bound; c.m.dt_gamma = dt_gamma; c.m.dt_min = mc.dt_min; c.m.dt_max = mc.dt_max;
    c.m.mip_bound = fminf(1.0f, bound);
    c.m.rb = 1.0f / c.m.mip_bound;
    c.m.half_H = 0.5f * (float)grid_size;
    c.m.two_rH = 2.0f / (float)grid_size;
    c.m.Hm1f = (float)(grid_size - 1);
    c.m.H = grid_size;
    c.m.log2H = (uint32_t)__builtin_ctz(grid_size);
    c.g = ssd_plane_geom(Hp, Wp);
    c.aabb[0] = c.aabb[1] = c.aabb[2] = -bound;
    c.aabb[3] = c.aabb[4] = c.aabb[5] = bound;
    c.min_near = min_near; c.T_thresh = T_thresh; c.bg = bg_color; c.sat = sat;
    c.N = N; c.S = S; c.cap = max_steps;
    c.plane_stride = (uint64_t)3 * Hp * Wp * 8;
    c.bitfield_stride = (grid_size * grid_size * grid_size) / 8;
    c.dt_gammas = dt_gammas;
    c.image_u8 = nullptr;
    return SSDNERF_OK;
}

extern "C" size_t ssdnerf_render_queue_workspace(uint32_t S, uint32_t N, uint32_t grid_size) { return ssd_render_ws(nullptr, S, N, grid_size).bytes; }

// Rays come from the (S, N, 3) arrays, or -- c2w != NULL -- from S x V cameras of h x w pixels (N == V * h * w).
static int rq_ray_src(RaySrc& src, const char* who, const float* rays_o, const float* rays_d, const float* c2w, const float* intrinsics, uint32_t V,
                      uint32_t h, uint32_t w, uint32_t N) {
    if (c2w != nullptr) {
        SSD_REQUIRE(intrinsics && V > 0 && h > 0 && w > 0 && (uint64_t)V * h * w == N, "%s: cameras need intrinsics and N == V * h * w", who);
        SSD_REQUIRE(V <= 65535, "%s: at most 65535 views per scene", who);
        src = ssd_ray_src_cams(c2w, intrinsics, V, h, w);
    } else {
        SSD_REQUIRE(rays_o && rays_d, "%s: null ray arrays", who);
        src = ssd_ray_src_arrays(rays_o, rays_d);
    }
    return SSDNERF_OK;
}

static int rq_first_hit(const uint8_t* bitfield, uint32_t grid_size, const RaySrc& src, uint32_t S, uint32_t N, float bound, float min_near,
                        float dt_gamma, const float* dt_gammas, uint32_t max_steps, float bg_color, float* image, float* depth, float* weights_sum,
                        int32_t* sample_counts, uint8_t* image_u8, void* workspace, size_t workspace_bytes, void* stream) {
    bool small_blocks = (grid_size & SSDNERF_FIRST_HIT_SMALL_BLOCKS) != 0;                   // flags ride in the upper half of `grid_size` (include/ssdnerf_hip.h)
    grid_size &= 0xffffu;
    SSD_REQUIRE(bitfield && image && depth && weights_sum && workspace, "render_first_hit: null pointer");
    if (workspace_bytes < ssdnerf_render_queue_workspace(S, N, grid_size))
        return ssdnerf_fail(SSDNERF_E_WORKSPACE, "render_first_hit: workspace %zu < %zu bytes", workspace_bytes, ssdnerf_render_queue_workspace(S, N, grid_size));
    SSD_REQUIRE(S <= 65535, "render_first_hit: at most 65535 scenes per launch");
    QueueCfg c;
    int rc = rq_make_cfg(c, 1, 1, grid_size, S, N, bound, min_near, dt_gamma, dt_gammas, max_steps, 0.f, bg_color, 0.f);
    if (rc) return rc;
    c.image_u8 = image_u8;
    hipStream_t s = (hipStream_t)stream;
    const RenderWs w = ssd_render_ws(workspace, S, N, grid_size);
    if (hipMemsetAsync(w.counters, 0, w.counter_bytes, s) != hipSuccess) return ssdnerf_fail(SSDNERF_E_LAUNCH, "render_first_hit: memset failed");   // hit counts, slice tickets, survivor counts, boundary tests
    hipLaunchKernelGGL(k_bitfield_linearize, dim3(ssd_blocks(c.bitfield_stride, RQ_TPB), S), dim3(RQ_TPB), 0, s, bitfield, grid_size, c.m.log2H, c.bitfield_stride, w.lin_bits);
    const uint32_t hc = grid_size >> RQ_COARSE_LOG2B;    // (the workspace reserves room for the finest block size, 2 cells)
    const bool coarse_ok = hc >= 8 && (hc * hc * hc / 8) <= RQ_COARSE_MAX_BYTES && (hc * hc * hc / 8) % 16 == 0 && bound <= 1.0f && getenv("SSDNERF_NO_COARSE") == nullptr;
    if (coarse_ok)
        hipLaunchKernelGGL(k_bitfield_coarsen, dim3(ssd_blocks(hc * hc * hc * 8, RQ_TPB), S), dim3(RQ_TPB), 0, s, w.lin_bits, grid_size, c.m.log2H, c.bitfield_stride, w.coarse);
    const uint64_t* blocks64 = nullptr;
    if (grid_size >= 16 && getenv("SSDNERF_NO_COARSE") == nullptr) {
        const uint32_t hb = grid_size >> 2;
        hipLaunchKernelGGL(k_bitfield_blocks64, dim3(ssd_blocks(hb * hb * hb, RQ_TPB), S), dim3(RQ_TPB), 0, s, w.lin_bits, grid_size, c.m.log2H, c.bitfield_stride, w.blocks64);
        blocks64 = w.blocks64;
    }
    CullGrid cg;
    cg.views_cap = N / 64 + 1;
    cg.zr_cap = N / 256 + 1;
    cg.tile_w_shift = cg.tile_h_shift = -1;
    if (src.c2w != nullptr && src.w > 0) {
        const uint32_t tw = (src.w + 15u) / 16u, th = (src.hw / src.w + 15u) / 16u;
        if (tw && !(tw & (tw - 1))) cg.tile_w_shift = __builtin_ctz(tw);
        if (th && !(th & (th - 1))) cg.tile_h_shift = __builtin_ctz(th);
    }
    dim3 grid;
    small_blocks = small_blocks && (!coarse_ok || (hc * hc * hc / 8) <= RQ_COARSE_SMALL);      // (a finer grid's coarse bits do not fit the small form: the default form then)
    const unsigned chunks = small_blocks ? RQ_CHUNKS_SMALL : RQ_CHUNKS_DEFAULT;
    if (src.c2w != nullptr) { cg.group = src.hw; grid = dim3(ssd_blocks(src.hw, RQ_TPB * chunks), src.V, S); }      // one view per blockIdx.y: camera loads are scalar
    else { cg.group = N; grid = dim3(ssd_blocks(N, RQ_TPB * chunks), 1, S); }
    const bool view_cull = coarse_ok && src.c2w != nullptr && src.hw >= 64 && src.w >= 16 && src.hw / src.w >= 16 && src.hw % src.w == 0;
    if (view_cull)
        hipLaunchKernelGGL(k_view_masks, dim3(src.V, S), dim3(RQ_TPB), 0, s, c.m, src, cg.views_cap, cg.zr_cap, w.coarse, w.view_masks, w.view_zr);
    if (small_blocks)
        hipLaunchKernelGGL((k_ray_cull<RQ_CHUNKS_SMALL, RQ_COARSE_SMALL>), grid, dim3(RQ_TPB), 0, s, c, src, cg, coarse_ok ? w.coarse : (const uint8_t*)nullptr, image, depth, weights_sum, sample_counts,
                           w.survivors, w.counters, view_cull ? w.view_masks : (const uint32_t*)nullptr, w.view_zr);
    else
        hipLaunchKernelGGL((k_ray_cull<RQ_CHUNKS_DEFAULT, RQ_COARSE_MAX_BYTES>), grid, dim3(RQ_TPB), 0, s, c, src, cg, coarse_ok ? w.coarse : (const uint8_t*)nullptr, image, depth, weights_sum, sample_counts,
                           w.survivors, w.counters, view_cull ? w.view_masks : (const uint32_t*)nullptr, w.view_zr);
    const char* to_env = getenv("SSDNERF_TICKET_ORDER");            // =0: the two-class queue of r02 - r04 (A/B runs, the bit-identity test); read per call,
    const bool ticket_order = !(to_env && to_env[0] == '0');        // and the same way by the shading launch (shade_mfma.hip, sm_shade)
    uint8_t* qkey = ticket_order ? w.qkey : nullptr;
    if (dt_gammas == nullptr && dt_gamma == 0.0f)
        hipLaunchKernelGGL(k_survivor_march<true>, dim3(ssd_blocks(N, RQ_MTPB * RQ_MCHUNKS), S), dim3(RQ_MTPB), 0, s, c, src, w.lin_bits, blocks64, w.survivors, image, depth, weights_sum,
                           sample_counts, w.queue, w.counters, qkey, w.key_stride);
    else
        hipLaunchKernelGGL(k_survivor_march<false>, dim3(ssd_blocks(N, RQ_MTPB * RQ_MCHUNKS), S), dim3(RQ_MTPB), 0, s, c, src, w.lin_bits, blocks64, w.survivors, image, depth, weights_sum,
                           sample_counts, w.queue, w.counters, qkey, w.key_stride);
    if (ticket_order) {
        hipLaunchKernelGGL(k_ticket_order, dim3(S), dim3(1024), 0, s, S, (const uint8_t*)w.qkey, w.key_stride, w.counters, w.order, w.order_stride);
    } else {
        hipLaunchKernelGGL(k_queue_close, dim3(ssd_blocks(N / 2 + 1, RQ_TPB), S), dim3(RQ_TPB), 0, s, S, N, (uint2*)w.queue, w.counters);   // moves <= N/2 entries
        hipLaunchKernelGGL(k_queue_total, dim3(ssd_blocks(S, 64)), dim3(64), 0, s, S, w.counters);
    }
    SSD_CHECK_LAUNCH("render_first_hit");
    return SSDNERF_OK;
}

extern "C" int ssdnerf_render_first_hit(const uint8_t* bitfield, uint32_t grid_size, const float* rays_o, const float* rays_d, uint32_t S, uint32_t N,
                                        float bound, float min_near, float dt_gamma, const float* dt_gammas, uint32_t max_steps, float bg_color,
                                        float* image, float* depth, float* weights_sum, int32_t* sample_counts, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (N == 0 || S == 0) return SSDNERF_OK;
    RaySrc src;
    if (int rc = rq_ray_src(src, "render_first_hit", rays_o, rays_d, nullptr, nullptr, 0, 0, 0, N)) return rc;
    return rq_first_hit(bitfield, grid_size, src, S, N, bound, min_near, dt_gamma, dt_gammas, max_steps, bg_color, image, depth, weights_sum, sample_counts,
                        nullptr, workspace, workspace_bytes, stream);
}

extern "C" int ssdnerf_render_first_hit_cams(const uint8_t* bitfield, uint32_t grid_size, const float* c2w, const float* intrinsics, uint32_t S, uint32_t V,
                                             uint32_t h, uint32_t w, float bound, float min_near, float dt_gamma, const float* dt_gammas,
                                             uint32_t max_steps, float bg_color, float* image, float* depth, float* weights_sum,
                                             int32_t* sample_counts, uint8_t* image_u8, void* workspace, size_t workspace_bytes, void* stream) {
    const uint64_t N64 = (uint64_t)V * h * w;
    if (N64 == 0 || S == 0) return SSDNERF_OK;
    SSD_REQUIRE(N64 <= 0xffffffffull, "render_first_hit_cams: more than 2^32 rays per scene");
    RaySrc src;
    if (int rc = rq_ray_src(src, "render_first_hit_cams", nullptr, nullptr, c2w, intrinsics, V, h, w, (uint32_t)N64)) return rc;
    return rq_first_hit(bitfield, grid_size, src, S, (uint32_t)N64, bound, min_near, dt_gamma, dt_gammas, max_steps, bg_color, image, depth, weights_sum,
                        sample_counts, image_u8, workspace, workspace_bytes, stream);
}

extern "C" int ssdnerf_render_shade_queue(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params, uint32_t grid_size,
                                          const float* rays_o, const float* rays_d, uint32_t S, uint32_t N, float bound, float min_near,
                                          float dt_gamma, const float* dt_gammas, uint32_t max_steps, float T_thresh, float bg_color,
                                          float sigmoid_saturation, float* image, float* depth, float* weights_sum, int32_t* sample_counts,
                                          int32_t* overflow_flag, void* workspace, size_t workspace_bytes, void* stream) {
    if (N == 0 || S == 0) return SSDNERF_OK;
    SSD_REQUIRE(planes && mlp_params && rays_o && rays_d && image && depth && weights_sum && workspace, "render_shade_queue: null pointer");
    SSD_REQUIRE(planes_dtype == 0 || planes_dtype == 1, "render_shade_queue: unsupported plane dtype");
    if (workspace_bytes < ssdnerf_render_queue_workspace(S, N, grid_size))
        return ssdnerf_fail(SSDNERF_E_WORKSPACE, "render_shade_queue: workspace too small");
    QueueCfg c;
    int rc = rq_make_cfg(c, Hp, Wp, grid_size, S, N, bound, min_near, dt_gamma, dt_gammas, max_steps, T_thresh, bg_color, sigmoid_saturation);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const RenderWs w = ssd_render_ws(workspace, S, N, grid_size);
    const uint32_t slices = ssd_blocks(N, RQ_SLICE);                       // worst case: every ray hits
    const uint32_t wg_per_scene = ssd_blocks(slices, RQ_TPB / 64);
    dim3 g(S * wg_per_scene), b(RQ_TPB);
    if (planes_dtype == 0) hipLaunchKernelGGL((k_shade_queue<float>), g, b, 0, s, c, slices, (const float*)planes, mlp_params, w.lin_bits, rays_o, rays_d, w.queue, w.counters, image, depth, weights_sum, sample_counts, overflow_flag);
    else hipLaunchKernelGGL((k_shade_queue<__half>), g, b, 0, s, c, slices, (const __half*)planes, mlp_params, w.lin_bits, rays_o, rays_d, w.queue, w.counters, image, depth, weights_sum, sample_counts, overflow_flag);
    SSD_CHECK_LAUNCH("render_shade_queue");
    return SSDNERF_OK;
}
