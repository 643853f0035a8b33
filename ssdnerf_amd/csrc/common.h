// ssdnerf_amd/csrc/common.h -- shared host/device helpers for libssdnerf_hip.so (gfx950 only).
//
// Arithmetic contract: this library is compiled with -ffp-contract=off; a fused
// multiply-add exists exactly where the source says __builtin_fmaf.  The marching arithmetic below
// is fp32 IEEE with correctly rounded division (hipcc default), so that integer outputs (per-ray
// sample counts, voxel indices, alive flags) are reproducible bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ssdnerf_hip.h"

// ------------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------------
extern thread_local char g_ssdnerf_err[512];
static inline int ssdnerf_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ssdnerf_err, sizeof(g_ssdnerf_err), fmt, ap);
    va_end(ap);
    return code;
}
#define SSD_REQUIRE(cond, ...) do { if (!(cond)) return ssdnerf_fail(SSDNERF_E_INVALID, __VA_ARGS__); } while (0)
#define SSD_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) return ssdnerf_fail(SSDNERF_E_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); } while (0)

static inline unsigned ssd_blocks(uint64_t work, unsigned threads) { return (unsigned)((work + threads - 1) / threads); }

// ------------------------------------------------------------------------------------------------
// conservative coarse occupancy shared by k_ray_cull (pre-test, render_queue.hip) and the shading kernels (tail bound):
// one bit per block of B^3 cells, B = 2^SSD_COARSE_LOG2B, dilated by B/2 cells; rays are sampled every SSD_COARSE_STEP cells.
#ifndef RQ_COARSE_LOG2B
#define RQ_COARSE_LOG2B 2
#endif
#define SSD_COARSE_STEP ((float)(1 << RQ_COARSE_LOG2B) - 0.1f)
// hit-queue entries carry, above the 24-bit ray index, the index of the LAST coarse test point of the ray that was not clear
// (0..126; 127 = no bound): beyond near + (j + 1) steps the ray cannot meet an occupied cell, so marching may stop there.
#define SSD_RAY_ID_MASK 0x00ffffffu
#define SSD_TAIL_NONE 127u

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
#define SSD_DEV __device__ __forceinline__
#define SSD_SQRT3 1.7320508075688772f

SSD_DEV float ssd_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// the output quantisation of eval_and_viz (base_nerf.py:551-553; k_quantize_u8): clamp to [0, 1], x 255, round half to even
SSD_DEV uint8_t ssd_quant_u8(float v) { return (uint8_t)rintf(fminf(fmaxf(v, 0.f), 1.f) * 255.f); }
SSD_DEV float ssd_clamp(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }
SSD_DEV float ssd_sign1(float v) { return copysignf(1.0f, v); }

// 10-bit-per-axis Morton code (what the reference's density grid is indexed by).
SSD_DEV uint32_t ssd_spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
SSD_DEV uint32_t ssd_morton(uint32_t x, uint32_t y, uint32_t z) { return ssd_spread3(x) | (ssd_spread3(y) << 1) | (ssd_spread3(z) << 2); }
SSD_DEV uint32_t ssd_compact3(uint32_t v) {
    v &= 0x49249249u;
    v = (v | (v >> 2)) & 0xC30C30C3u;
    v = (v | (v >> 4)) & 0x0F00F00Fu;
    v = (v | (v >> 8)) & 0xFF0000FFu;
    v = (v | (v >> 16)) & 0x0000FFFFu;
    return v;
}

// ------------------------------------------------------------------------------------------------
// Density-grid guided marching: one "probe" at depth t and the empty-cell skip.
// Restates the stepping rule of the reference (lib/ops/raymarching/src/raymarching.cu:359-399,
// repeated at :427-480 and :755-810); see oracle/raymarching_oracle.c for the CPU statement of the
// same contract.  MarchCfg holds everything that is uniform over a launch.
// ------------------------------------------------------------------------------------------------
struct MarchCfg {
    float bound, dt_gamma, dt_min, dt_max, rH, Hf, H3f, Cf;
    uint32_t H, C;
    int h_pow2;  // H is a power of two: 0.5*(v)*H is exact in fp32, the double detour can be skipped bit-exactly
    const uint8_t* grid;
};

static inline MarchCfg ssd_make_march_cfg(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid) {
    MarchCfg c;
    c.bound = bound; c.dt_gamma = dt_gamma;
    c.dt_min = 2.0f * SSD_SQRT3 / (float)max_steps;
    c.dt_max = 2.0f * SSD_SQRT3 * (float)(1u << (C - 1)) / (float)H;
    c.H = H; c.C = C; c.Hf = (float)H; c.rH = 1.0f / (float)H; c.H3f = (float)(H * H * H); c.Cf = (float)C;
    c.h_pow2 = (H & (H - 1)) == 0;
    c.grid = grid;
    return c;
}

struct RayGeom { float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz; };

SSD_DEV RayGeom ssd_load_ray(const float* __restrict__ o, const float* __restrict__ d) {
    RayGeom r;
    r.ox = o[0]; r.oy = o[1]; r.oz = o[2];
    r.dx = d[0]; r.dy = d[1]; r.dz = d[2];
    r.rdx = 1.0f / r.dx; r.rdy = 1.0f / r.dy; r.rdz = 1.0f / r.dz;  // IEEE division (contract C4)
    return r;
}

struct Probe { float x, y, z, dt, mip_bound; int nx, ny, nz; bool occ; };

SSD_DEV int ssd_cell(const MarchCfg& c, float v /* x*rbound+1 */) {
    float s;
    if (c.h_pow2) s = (0.5f * v) * c.Hf;                    // exact: both factors are powers of two
    else s = (float)(0.5 * (double)v * (double)c.H);        // reference promotes through double (raymarching.cu:374-376)
    return (int)ssd_clamp(s, 0.0f, (float)(c.H - 1));
}

SSD_DEV Probe ssd_probe(const MarchCfg& c, const RayGeom& r, float t) {
    Probe p;
    p.x = ssd_clamp(ssd_fma(t, r.dx, r.ox), -c.bound, c.bound);
    p.y = ssd_clamp(ssd_fma(t, r.dy, r.oy), -c.bound, c.bound);
    p.z = ssd_clamp(ssd_fma(t, r.dz, r.oz), -c.bound, c.bound);
    p.dt = ssd_clamp(t * c.dt_gamma, c.dt_min, c.dt_max);
    int level = 0;
    if (c.C > 1) {  // cascade selection (raymarching.cu:42-54); with C == 1 both terms clamp to 0
        int e1, e2;
        (void)frexpf(fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))), &e1);
        (void)frexpf((float)((double)(p.dt * c.Hf) * 0.5), &e2);
        const int l1 = (int)fminf(c.Cf - 1.0f, fmaxf(0.0f, (float)e1));
        const int l2 = (int)fminf(c.Cf - 1.0f, fmaxf(0.0f, (float)e2));
        level = l1 > l2 ? l1 : l2;
    }
    p.mip_bound = fminf(ldexpf(1.0f, level), c.bound);
    const float rb = 1.0f / p.mip_bound;
    p.nx = ssd_cell(c, ssd_fma(p.x, rb, 1.0f));
    p.ny = ssd_cell(c, ssd_fma(p.y, rb, 1.0f));
    p.nz = ssd_cell(c, ssd_fma(p.z, rb, 1.0f));
    const uint32_t idx = (uint32_t)ssd_fma((float)level, c.H3f, (float)ssd_morton((uint32_t)p.nx, (uint32_t)p.ny, (uint32_t)p.nz));
    p.occ = (c.grid[idx >> 3] >> (idx & 7u)) & 1u;
    return p;
}

// Advance t past the empty voxel the probe landed in: DDA distance to the next voxel face, then
// fixed-size steps until that distance is covered (at least one step is always taken).
SSD_DEV float ssd_skip_empty(const MarchCfg& c, const RayGeom& r, const Probe& p, float t) {
    const float fx = ssd_fma(0.5f, ssd_sign1(r.dx), (float)p.nx + 0.5f);
    const float fy = ssd_fma(0.5f, ssd_sign1(r.dy), (float)p.ny + 0.5f);
    const float fz = ssd_fma(0.5f, ssd_sign1(r.dz), (float)p.nz + 0.5f);
    const float tx = ssd_fma(ssd_fma(fx * c.rH, 2.0f, -1.0f), p.mip_bound, -p.x) * r.rdx;
    const float ty = ssd_fma(ssd_fma(fy * c.rH, 2.0f, -1.0f), p.mip_bound, -p.y) * r.rdy;
    const float tz = ssd_fma(ssd_fma(fz * c.rH, 2.0f, -1.0f), p.mip_bound, -p.z) * r.rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        t += ssd_clamp(t * c.dt_gamma, c.dt_min, c.dt_max);
    } while (t < tt);
    return t;
}

// ------------------------------------------------------------------------------------------------
// `do { t += dt; } while (t < tt);` with a CONSTANT step, without the loop (r05).  The march's parameters are a chain of fp32 additions, so the kernels
// may not replace k steps by t + k * dt -- but inside one binade the chain IS an exact arithmetic progression: t is a multiple of u = ulp(t), so
// fl(t + dt) = t + D with D = dt rounded to a multiple of u, the same D at every step (a tie -- dt's residue exactly u / 2 -- would alternate with the
// parity of t: excluded by comparing the first two differences), and t + k D is a multiple of u below the binade's end, hence representable: the chain's
// k-th member is fma(k, D, t), exactly.  The count k comes from a reciprocal estimate and is corrected by one step either way; anything that leaves the
// binade (or a tie) takes the loop.  tests/test_march_closed_form_cpu.py replays the identity in numpy; the GPU parity tests pin the kernels.
SSD_DEV float ssd_run_to_const(float dt, float t, float tt) {
    const float t1 = t + dt;
    if (!(t1 < tt)) return t1;                                           // one step is enough (the loop takes at least one)
    const float t2 = t1 + dt;
    const float D = t1 - t, D2 = t2 - t1;                                // exact differences of neighbours in one binade
    float k = ceilf((tt - t) * __builtin_amdgcn_rcpf(D));
    float c = ssd_fma(k, D, t);
    c = c < tt ? c + D : c;                                              // the estimate may be one short ...
    c = (c - D < tt) ? c : c - D;                                        // ... or one long
    const bool ok = D == D2 && ((__float_as_uint(t) ^ __float_as_uint(c)) >> 23) == 0 && !(c < tt) && (c - D < tt);
    if (ok) return c;
    t = t2;                                                              // (two steps are already known to be needed ... t1 < tt; t2 is the chain's second member)
    while (t < tt) t += dt;
    return t;
}

// ------------------------------------------------------------------------------------------------
// Camera rays (reference: get_ray_directions / get_rays / get_cam_rays, lib/core/utils/nerf_utils.py:17-61).  ONE statement of the arithmetic,
// shared by k_cam_rays (raygen.hip, materialises the arrays the reference API hands around) and by the render kernels when they are given
// cameras instead of ray arrays -- so a ray generated in a kernel is bit-identical to the one k_cam_rays would have stored.
//     d_cam = ((x + 0.5 - cx) / fx, (y + 0.5 - cy) / fy, 1);  d = R d_cam / max(|R d_cam|, 1e-12);  o = c2w[:3, 3]
SSD_DEV void ssd_cam_ray(const float* __restrict__ M /* c2w, 16 floats row-major */, const float* __restrict__ K /* fx fy cx cy */, uint32_t px_i,
                         uint32_t py_i, float o[3], float d[3]) {
    const float px = (float)px_i + 0.5f, py = (float)py_i + 0.5f;
    const float dx = (px - K[2]) / K[0], dy = (py - K[3]) / K[1];
    const float vx = ssd_fma(dx, M[0], ssd_fma(dy, M[1], M[2]));
    const float vy = ssd_fma(dx, M[4], ssd_fma(dy, M[5], M[6]));
    const float vz = ssd_fma(dx, M[8], ssd_fma(dy, M[9], M[10]));
    const float inv = 1.0f / fmaxf(sqrtf(ssd_fma(vx, vx, ssd_fma(vy, vy, vz * vz))), 1e-12f);
    d[0] = vx * inv; d[1] = vy * inv; d[2] = vz * inv;
    o[0] = M[3]; o[1] = M[7]; o[2] = M[11];
}

// Where a render kernel gets ray n of a scene from: the (S, N, 3) arrays of the reference API, or -- c2w != NULL -- the cameras themselves
// (c2w [S][V][16], intr [S][V][4], N = V * hw rays per scene, ray n = pixel n % hw of view n / hw): 80 B per VIEW instead of 24 B per RAY.
struct RaySrc {
    const float* rays_o; const float* rays_d;
    const float* c2w; const float* intr;
    uint32_t V, hw, w;
    int hw_shift, w_shift;        // log2 when hw / w are powers of two (the 128 x 128 views of the hot path), else -1
};
static inline RaySrc ssd_ray_src_arrays(const float* o, const float* d) { RaySrc s = {}; s.rays_o = o; s.rays_d = d; s.hw_shift = s.w_shift = -1; return s; }
static inline RaySrc ssd_ray_src_cams(const float* c2w, const float* intr, uint32_t V, uint32_t h, uint32_t w) {
    RaySrc s = {};
    s.c2w = c2w; s.intr = intr; s.V = V; s.hw = h * w; s.w = w;
    s.hw_shift = (s.hw & (s.hw - 1)) == 0 ? __builtin_ctz(s.hw) : -1;
    s.w_shift = (w & (w - 1)) == 0 ? __builtin_ctz(w) : -1;
    return s;
}
SSD_DEV RayGeom ssd_ray_geom(float ox, float oy, float oz, float dx, float dy, float dz) {
    RayGeom r;
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    r.rdx = 1.0f / dx; r.rdy = 1.0f / dy; r.rdz = 1.0f / dz;
    return r;
}
// ray n of scene `scene` (N rays per scene)
SSD_DEV RayGeom ssd_fetch_ray(const RaySrc& s, uint32_t scene, uint32_t N, uint32_t n) {
    if (s.c2w == nullptr) {
        const uint64_t gi = (uint64_t)scene * N + n;
        return ssd_load_ray(s.rays_o + 3 * gi, s.rays_d + 3 * gi);
    }
    const uint32_t view = s.hw_shift >= 0 ? n >> s.hw_shift : n / s.hw;
    const uint32_t pix = n - view * s.hw;
    const uint32_t py = s.w_shift >= 0 ? pix >> s.w_shift : pix / s.w;
    const uint32_t px = pix - py * s.w;
    const uint64_t cam = (uint64_t)scene * s.V + view;
    float o[3], d[3];
    ssd_cam_ray(s.c2w + cam * 16, s.intr + cam * 4, px, py, o, d);
    return ssd_ray_geom(o[0], o[1], o[2], d[0], d[1], d[2]);
}

// Tail bound of a ray that passed the coarse pre-test (render_queue.hip, k_ray_cull): its list / queue word carries, above the 24-bit ray
// index, the index j_last of the last coarse test point that was not clear; past near + (j_last + 1) coarse steps no occupied cell can be
// met, + 0.5 step of slack for the difference between the scan's accumulated test parameters and this product form.
SSD_DEV float ssd_coarse_step_t(const RayGeom& q, float cell_world /* 2 / H * mip_bound */) {
    const float len = sqrtf(ssd_fma(q.dx, q.dx, ssd_fma(q.dy, q.dy, q.dz * q.dz)));
    return (SSD_COARSE_STEP * cell_world) / fmaxf(len, 1e-20f);
}
SSD_DEV float ssd_tail_far(const RayGeom& q, float cell_world, float near_, float far_, uint32_t packed, bool packing) {
    const uint32_t jl = packed >> 24;
    if (!packing || jl >= SSD_TAIL_NONE) return far_;
    return fminf(far_, ssd_fma((float)jl + 1.5f, ssd_coarse_step_t(q, cell_world), near_));
}

// Workspace of the two-stage renderer (caller-owned, ssdnerf_render_queue_workspace bytes), shared by render_queue.hip and shade_mfma.hip:
//   counters : 4 kinds x S scenes, ONE 128-BYTE LINE EACH (ssd_counter): hits per scene | slice tickets | survivors per scene | termination
//              tests within 2e-6 of T_thresh (diagnostic).  Device-scope atomics on one address are resolved one after the other at the memory
//              side; r02 measured 2.7 ms for ~250 k wave-level appends that all landed in ONE cache line (profiles/r02/a_kernel_stats_*.csv),
//              so every counter has its own line and the producers reserve per BLOCK, not per wave.
//   lin_bits [S][H^3/8]: the bitfield in linear z/y/x order         coarse [S][(H/2)^3/8] (room for the finest block size)
//   queue [S][N] uint2 {ray | tail << 24, t_first}                    survivors [S][N] uint2 {ray | tail << 24, t_start (r03: head skip)}
#define SSD_COUNTER_STRIDE 32u     // u32 words per counter (128 B)
enum { SSD_CNT_HITS = 0, SSD_CNT_TICKETS = 1, SSD_CNT_SURVIVORS = 2, SSD_CNT_BOUNDARY = 3, SSD_CNT_HITS_SHORT = 4, SSD_CNT_KINDS = 5 };
__host__ __device__ static inline uint32_t ssd_counter(uint32_t kind, uint32_t S, uint32_t scene) { return (kind * S + scene) * SSD_COUNTER_STRIDE; }
//   view_masks [S][<= N/64 views][8] u32: per view, a 16 x 16-tile mask of the image tiles that a set coarse block projects into (k_view_masks)
//   view_zr [S][<= N/256 views][256] u32: per image tile, the camera-depth range of the set coarse blocks that project into it, as two bf16
//              (low half: lower end rounded down, high half: upper end rounded up): k_ray_cull scans a ray's test points inside that range only
//   qkey [S][key_stride] u8: per queue entry, an upper bound of the ray's remaining march steps (<= 255); order [S][order_stride] u32: the queue's 64-entry
//              slices in the order the shading kernel's tickets take them (k_ticket_order: longest slice first, r05)
struct RenderWs { uint32_t* counters; uint8_t* lin_bits; uint8_t* coarse; uint2* queue; uint2* survivors; uint32_t* view_masks; uint32_t* view_zr; uint64_t* blocks64; uint8_t* qkey; uint32_t* order; uint32_t key_stride, order_stride; size_t counter_bytes, bytes; };
static inline RenderWs ssd_render_ws(void* base, uint32_t S, uint32_t N, uint32_t grid_size) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t counters = up((size_t)SSD_CNT_KINDS * S * SSD_COUNTER_STRIDE * 4), bits = up((size_t)S * grid_size * grid_size * grid_size / 8);
    const size_t hc = grid_size / 2, coarse = up((size_t)S * (hc * hc * hc / 8)), queue = up((size_t)S * N * sizeof(uint2));
    RenderWs w;
    uint8_t* b = (uint8_t*)base;
    w.counters = (uint32_t*)b;
    w.lin_bits = b + counters;
    w.coarse = b + counters + bits;
    w.queue = (uint2*)(b + counters + bits + coarse);
    w.survivors = (uint2*)(b + counters + bits + coarse + queue);
    const size_t surv = up((size_t)S * N * sizeof(uint2));
    w.view_masks = (uint32_t*)(b + counters + bits + coarse + queue + surv);
    const size_t masks = up((size_t)S * (N / 64 + 1) * 32);
    w.view_zr = (uint32_t*)(b + counters + bits + coarse + queue + surv + masks);
    const size_t zr = up((size_t)S * (N / 256 + 1) * 1024);
    w.blocks64 = (uint64_t*)(b + counters + bits + coarse + queue + surv + masks + zr);      // [S][(H/4)^3] u64: the 64 cells of a 4^3 block in one word (k_survivor_march)
    w.key_stride = (uint32_t)(((size_t)N + 63) / 64 * 64);
    w.order_stride = (uint32_t)(((size_t)N + 63) / 64);
    const size_t keys = up((size_t)S * w.key_stride), order = up((size_t)S * w.order_stride * 4);
    w.qkey = b + counters + bits + coarse + queue + surv + masks + zr + bits;
    w.order = (uint32_t*)(b + counters + bits + coarse + queue + surv + masks + zr + bits + keys);
    w.counter_bytes = counters;
    w.bytes = counters + bits + coarse + queue + surv + masks + zr + bits + keys + order;
    return w;
}

// Slab test of one ray against the scene box.  Returns false on a miss (near = far = FLT_MAX).
SSD_DEV bool ssd_near_far(const float* __restrict__ aabb, const RayGeom& r, float min_near, float& near_, float& far_) {
    float lo = (aabb[0] - r.ox) * r.rdx, hi = (aabb[3] - r.ox) * r.rdx;
    if (lo > hi) { const float s = lo; lo = hi; hi = s; }
    float a = (aabb[1] - r.oy) * r.rdy, b = (aabb[4] - r.oy) * r.rdy;
    if (a > b) { const float s = a; a = b; b = s; }
    bool miss = (lo > b) || (a > hi);
    if (!miss) {
        if (a > lo) lo = a;
        if (b < hi) hi = b;
        a = (aabb[2] - r.oz) * r.rdz; b = (aabb[5] - r.oz) * r.rdz;
        if (a > b) { const float s = a; a = b; b = s; }
        miss = (lo > b) || (a > hi);
        if (!miss) {
            if (a > lo) lo = a;
            if (b < hi) hi = b;
        }
    }
    if (miss) { near_ = far_ = 3.402823466e+38f; return false; }
    if (lo < min_near) lo = min_near;
    near_ = lo; far_ = hi;
    return true;
}
