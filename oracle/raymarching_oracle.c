/* oracle/raymarching_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), NOT product code.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (ssdnerf_amd/) never links, imports or falls back to it.
 *
 * A plain-C restatement of the ten ray-marching routines of the reference's CUDA
 * extension (reference: lib/ops/raymarching/src/raymarching.cu).  Each function cites
 * the lines it follows.  The restatement is written against an explicit ARITHMETIC
 * CONTRACT (DESIGN.md "Arithmetic contract"), because integer outputs (per-ray sample
 * counts, alive flags) are decided by chains of fp32 compares:
 *
 *   C1. all ray/sample arithmetic is IEEE binary32, round-to-nearest-even;
 *   C2. compiled with -ffp-contract=off: a fused multiply-add happens exactly where this
 *       file writes fmaf(), nowhere else.  fmaf() is used where the reference expression
 *       is a single-use product feeding an add/sub (what nvcc -fmad=true contracts);
 *   C3. the voxel index goes through double exactly as the reference's literal `0.5 *`
 *       forces (raymarching.cu:374-376), then narrows to float, clamps, truncates;
 *   C4. divisions are IEEE (the reference does not use -use_fast_math);
 *   C5. exp in compositing: the reference uses the hardware intrinsic __expf; the oracle
 *       uses expf().  Composited floats are therefore compared with a tolerance, not
 *       bitwise (tests/ state it).
 *
 * Pinning: tests/test_oracle_vs_reference.py checks every function here against
 * oracle/_ref (the reference's own .cu compiled for the CPU by oracle/build_ref.sh):
 * bit-exact for all integer outputs and for the marched sample positions.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_SQRT3 1.7320508075688772f

/* cap the OpenMP team (bench.py's cpu_baseline reports exactly this number as "cores") */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static inline float orc_clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }
static inline float orc_sign1(float v) { return copysignf(1.0f, v); }

/* ---- Morton code, 10 bits per axis (raymarching.cu:56-81) ------------------------- */
static inline uint32_t orc_spread3(uint32_t v) {
    /* put two zero bits between each of the low 10 bits of v */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t orc_morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return orc_spread3(x) | (orc_spread3(y) << 1) | (orc_spread3(z) << 2);
}
static inline uint32_t orc_compact3(uint32_t v) {
    v &= 0x49249249u;
    v = (v | (v >> 2)) & 0xC30C30C3u;
    v = (v | (v >> 4)) & 0x0F00F00Fu;
    v = (v | (v >> 8)) & 0xFF0000FFu;
    v = (v | (v >> 16)) & 0x0000FFFFu;
    return v;
}

/* ---- cascade level helpers (raymarching.cu:42-54) ---------------------------------- */
static inline int orc_level_from_pos(float x, float y, float z, float n_cascade) {
    const float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    (void)frexpf(m, &e);
    return (int)fminf(n_cascade - 1.0f, fmaxf(0.0f, (float)e));
}
static inline int orc_level_from_dt(float dt, float H, float n_cascade) {
    /* dt*H in fp32, the literal 0.5 promotes to double, narrowed back to float (C3) */
    const float m = (float)((double)(dt * H) * 0.5);
    int e;
    (void)frexpf(m, &e);
    return (int)fminf(n_cascade - 1.0f, fmaxf(0.0f, (float)e));
}

/* One marching "probe": everything the three marchers share for the sample at depth t.
 * (raymarching.cu:359-379 == :427-448 == :755-775) */
typedef struct {
    float x, y, z, dt;   /* clamped position and step size              */
    int nx, ny, nz;      /* voxel of the cascade the probe lands in     */
    float mip_bound;     /* half-extent of that cascade                 */
    int occupied;        /* bit of the Morton-ordered density bitfield  */
} orc_probe;

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3f, Hf, Cf;
    uint32_t H;
    const uint8_t* grid;
} orc_ray;

static inline void orc_ray_init(orc_ray* r, const float* o, const float* d, const uint8_t* grid, float bound,
                                float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    r->ox = o[0]; r->oy = o[1]; r->oz = o[2];
    r->dx = d[0]; r->dy = d[1]; r->dz = d[2];
    r->rdx = 1.0f / r->dx; r->rdy = 1.0f / r->dy; r->rdz = 1.0f / r->dz;
    r->bound = bound; r->dt_gamma = dt_gamma;
    r->dt_min = 2.0f * ORC_SQRT3 / (float)max_steps;                 /* .cu:345, :744 */
    r->dt_max = 2.0f * ORC_SQRT3 * (float)(1u << (C - 1)) / (float)H; /* .cu:346, :745 */
    r->H = H; r->Hf = (float)H; r->rH = 1.0f / (float)H;
    r->H3f = (float)(H * H * H);
    r->Cf = (float)C; r->grid = grid;
}

static inline void orc_probe_at(const orc_ray* r, float t, orc_probe* p) {
    p->x = orc_clampf(fmaf(t, r->dx, r->ox), -r->bound, r->bound);
    p->y = orc_clampf(fmaf(t, r->dy, r->oy), -r->bound, r->bound);
    p->z = orc_clampf(fmaf(t, r->dz, r->oz), -r->bound, r->bound);
    p->dt = orc_clampf(t * r->dt_gamma, r->dt_min, r->dt_max);
    const int la = orc_level_from_pos(p->x, p->y, p->z, r->Cf);
    const int lb = orc_level_from_dt(p->dt, r->Hf, r->Cf);
    const int level = la > lb ? la : lb;
    p->mip_bound = fminf(scalbnf(1.0f, level), r->bound);
    const float mip_rbound = 1.0f / p->mip_bound;
    /* (x*rb + 1) in fp32 (fused), then *0.5 and *H in double, narrowed, clamped, truncated (C3) */
    const double Hd = (double)r->H;
    const float hi = (float)(r->H - 1);
    p->nx = (int)orc_clampf((float)(0.5 * (double)fmaf(p->x, mip_rbound, 1.0f) * Hd), 0.0f, hi);
    p->ny = (int)orc_clampf((float)(0.5 * (double)fmaf(p->y, mip_rbound, 1.0f) * Hd), 0.0f, hi);
    p->nz = (int)orc_clampf((float)(0.5 * (double)fmaf(p->z, mip_rbound, 1.0f) * Hd), 0.0f, hi);
    /* index = level*H^3 + morton, evaluated in fp32 like the reference (exact for H<=128) */
    const uint32_t idx = (uint32_t)fmaf((float)level, r->H3f, (float)orc_morton_encode((uint32_t)p->nx, (uint32_t)p->ny, (uint32_t)p->nz));
    p->occupied = (r->grid[idx >> 3] >> (idx & 7u)) & 1;
}

/* Distance-to-next-voxel skip (raymarching.cu:390-398 == :472-479 == :802-809).  Returns new t. */
static inline float orc_skip_empty(const orc_ray* r, const orc_probe* p, float t) {
    const float fx = fmaf(0.5f, orc_sign1(r->dx), (float)p->nx + 0.5f);
    const float fy = fmaf(0.5f, orc_sign1(r->dy), (float)p->ny + 0.5f);
    const float fz = fmaf(0.5f, orc_sign1(r->dz), (float)p->nz + 0.5f);
    const float tx = fmaf(fmaf(fx * r->rH, 2.0f, -1.0f), p->mip_bound, -p->x) * r->rdx;
    const float ty = fmaf(fmaf(fy * r->rH, 2.0f, -1.0f), p->mip_bound, -p->y) * r->rdy;
    const float tz = fmaf(fmaf(fz * r->rH, 2.0f, -1.0f), p->mip_bound, -p->z) * r->rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        t += orc_clampf(t * r->dt_gamma, r->dt_min, r->dt_max);
    } while (t < tt);
    return t;
}

/* ===================================================================================== */
/* near_far_from_aabb  (raymarching.cu:91-145)                                            */
void orc_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                            float min_near, float* nears, float* fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const float* o = rays_o + 3 * n;
        const float* d = rays_d + 3 * n;
        float lo = -FLT_MAX, hi = FLT_MAX; /* running slab intersection */
        int miss = 0;
        for (int a = 0; a < 3 && !miss; ++a) {
            const float rd = 1.0f / d[a];
            float t0 = (aabb[a] - o[a]) * rd;
            float t1 = (aabb[a + 3] - o[a]) * rd;
            if (t0 > t1) { const float s = t0; t0 = t1; t1 = s; }
            if (a == 0) { lo = t0; hi = t1; continue; }
            /* the reference tests overlap with strict '>' before tightening (.cu:121,133) */
            if (lo > t1 || t0 > hi) { miss = 1; break; }
            if (t0 > lo) lo = t0;
            if (t1 < hi) hi = t1;
        }
        if (miss) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (lo < min_near) lo = min_near;
        nears[n] = lo;
        fars[n] = hi;
    }
}

/* sph_from_ray (raymarching.cu:162-198): exported by the reference, never called. */
void orc_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    const float rpi = 0.3183098861837907f;
    for (uint32_t n = 0; n < N; ++n) {
        const float* o = rays_o + 3 * n;
        const float* d = rays_d + 3 * n;
        const float A = fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0]));
        const float Bh = fmaf(o[2], d[2], fmaf(o[1], d[1], o[0] * d[0]));
        const float Cq = fmaf(o[2], o[2], fmaf(o[1], o[1], o[0] * o[0])) - radius * radius;
        const float t = (-Bh + sqrtf(Bh * Bh - A * Cq)) / A;
        const float x = fmaf(t, d[0], o[0]), y = fmaf(t, d[1], o[1]), z = fmaf(t, d[2], o[2]);
        const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[2 * n + 0] = fmaf(2.0f * theta, rpi, -1.0f);
        coords[2 * n + 1] = phi * rpi;
    }
}

/* morton3D / morton3D_invert (raymarching.cu:214-254) */
void orc_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
    for (uint32_t n = 0; n < N; ++n)
        indices[n] = (int32_t)orc_morton_encode((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}
void orc_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    for (uint32_t n = 0; n < N; ++n) {
        const int32_t v = indices[n]; /* arithmetic shift of the signed value, as in the reference */
        coords[3 * n + 0] = (int32_t)orc_compact3((uint32_t)(v >> 0));
        coords[3 * n + 1] = (int32_t)orc_compact3((uint32_t)(v >> 1));
        coords[3 * n + 2] = (int32_t)orc_compact3((uint32_t)(v >> 2));
    }
}

/* packbits (raymarching.cu:267-289): N output bytes, bit i of byte n = grid[8n+i] > thresh */
void orc_packbits(const float* grid, uint32_t N, float thresh, uint8_t* bitfield) {
    for (uint32_t n = 0; n < N; ++n) {
        unsigned b = 0;
        for (unsigned i = 0; i < 8; ++i) b |= (unsigned)(grid[8u * n + i] > thresh) << i;
        bitfield[n] = (uint8_t)b;
    }
}

/* ===================================================================================== */
/* march_rays_train (raymarching.cu:311-482).
 * The reference reserves output slots with atomicAdd in whatever order threads arrive; the oracle
 * uses ray order (the order the serial execution of the reference produces), which is one valid
 * schedule.  counter[0] += total samples, counter[1] += N.                                       */
static uint32_t orc_march_count(const orc_ray* r, float t0, float far_, uint32_t max_steps) {
    uint32_t k = 0;
    float t = t0;
    orc_probe p;
    while (t < far_ && k < max_steps) {
        orc_probe_at(r, t, &p);
        if (p.occupied) { ++k; t += p.dt; }
        else t = orc_skip_empty(r, &p, t);
    }
    return k;
}

void orc_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                          uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                          const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                          const float* noises) {
    uint32_t* counts = (uint32_t*)malloc(sizeof(uint32_t) * (N ? N : 1));
    float* t0s = (float*)malloc(sizeof(float) * (N ? N : 1));
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        orc_ray r;
        orc_ray_init(&r, rays_o + 3 * n, rays_d + 3 * n, grid, bound, dt_gamma, max_steps, C, H);
        const float near_ = nears[n];
        /* jitter: t0 = near + clamp(near*dt_gamma)*noise, fused (.cu:351) */
        const float t0 = fmaf(orc_clampf(near_ * dt_gamma, r.dt_min, r.dt_max), noises[n], near_);
        t0s[n] = t0;
        counts[n] = orc_march_count(&r, t0, fars[n], max_steps);
    }
    uint32_t point_base = (uint32_t)counter[0], ray_base = (uint32_t)counter[1];
    uint32_t* offs = (uint32_t*)malloc(sizeof(uint32_t) * (N ? N : 1));
    for (uint32_t n = 0; n < N; ++n) { offs[n] = point_base; point_base += counts[n]; }
    counter[0] = (int32_t)point_base;
    counter[1] = (int32_t)(ray_base + N);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const uint32_t slot = ray_base + (uint32_t)n;
        rays[3 * slot + 0] = (int32_t)n;
        rays[3 * slot + 1] = (int32_t)offs[n];
        rays[3 * slot + 2] = (int32_t)counts[n];
        if (counts[n] == 0 || offs[n] + counts[n] > M) continue; /* dropped ray (.cu:415-416) */
        orc_ray r;
        orc_ray_init(&r, rays_o + 3 * n, rays_d + 3 * n, grid, bound, dt_gamma, max_steps, C, H);
        float* px = xyzs + 3 * (size_t)offs[n];
        float* pd = dirs + 3 * (size_t)offs[n];
        float* pl = deltas + 2 * (size_t)offs[n];
        float t = t0s[n];
        const float far_ = fars[n];
        uint32_t k = 0;
        orc_probe p;
        while (t < far_ && k < counts[n]) {
            orc_probe_at(&r, t, &p);
            if (p.occupied) {
                px[0] = p.x; px[1] = p.y; px[2] = p.z;
                pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                pl[0] = p.dt; pl[1] = t; /* (dt, absolute depth)  .cu:460-462 */
                t += p.dt;
                px += 3; pd += 3; pl += 2; ++k;
            } else t = orc_skip_empty(&r, &p, t);
        }
    }
    free(counts); free(t0s); free(offs);
}

/* composite_rays_train_forward (raymarching.cu:502-581) */
void orc_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                      uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth, float* image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const uint32_t id = (uint32_t)rays[3 * n], off = (uint32_t)rays[3 * n + 1], cnt = (uint32_t)rays[3 * n + 2];
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
        if (cnt != 0 && off + cnt <= M) {
            for (uint32_t s = 0; s < cnt; ++s) {
                const uint32_t i = off + s;
                const float alpha = 1.0f - expf(-sigmas[i] * deltas[2 * i]);
                const float w = alpha * T;
                r = fmaf(w, rgbs[3 * i + 0], r);
                g = fmaf(w, rgbs[3 * i + 1], g);
                b = fmaf(w, rgbs[3 * i + 2], b);
                d = fmaf(w, deltas[2 * i + 1], d);
                ws += w;
                T *= 1.0f - alpha;
                if (T < T_thresh) break; /* tested AFTER the update in the train branch (.cu:558-561) */
            }
        }
        weights_sum[id] = ws; depth[id] = d;
        image[3 * id] = r; image[3 * id + 1] = g; image[3 * id + 2] = b;
    }
}

/* composite_rays_train_backward (raymarching.cu:605-687).  grad_depth is ignored by the reference. */
void orc_composite_rays_train_backward(const float* grad_ws, const float* grad_image, const float* sigmas, const float* rgbs,
                                       const float* deltas, const int32_t* rays, const float* weights_sum, const float* image,
                                       uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const uint32_t id = (uint32_t)rays[3 * n], off = (uint32_t)rays[3 * n + 1], cnt = (uint32_t)rays[3 * n + 2];
        if (cnt == 0 || off + cnt > M) continue;
        const float gr = grad_image[3 * id], gg = grad_image[3 * id + 1], gb = grad_image[3 * id + 2], gw = grad_ws[id];
        const float rF = image[3 * id], gF = image[3 * id + 1], bF = image[3 * id + 2], wsF = weights_sum[id];
        float T = 1.0f, r = 0, g = 0, b = 0;
        for (uint32_t s = 0; s < cnt; ++s) {
            const uint32_t i = off + s;
            const float alpha = 1.0f - expf(-sigmas[i] * deltas[2 * i]);
            const float w = alpha * T;
            r = fmaf(w, rgbs[3 * i + 0], r);
            g = fmaf(w, rgbs[3 * i + 1], g);
            b = fmaf(w, rgbs[3 * i + 2], b);
            T *= 1.0f - alpha;
            if (T < T_thresh) break; /* the sample that trips the threshold gets no gradient (.cu:657-660) */
            grad_rgbs[3 * i + 0] = gr * w;
            grad_rgbs[3 * i + 1] = gg * w;
            grad_rgbs[3 * i + 2] = gb * w;
            float acc = gr * fmaf(T, rgbs[3 * i + 0], -(rF - r));
            acc = fmaf(gg, fmaf(T, rgbs[3 * i + 1], -(gF - g)), acc);
            acc = fmaf(gb, fmaf(T, rgbs[3 * i + 2], -(bF - b)), acc);
            acc = fmaf(gw, 1.0f - wsF, acc);
            grad_sigmas[i] = deltas[2 * i] * acc;
        }
    }
}

/* ===================================================================================== */
/* march_rays, inference (raymarching.cu:705-812): alive ray n owns slots [n*n_step, (n+1)*n_step). */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                    const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                    const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                    const float* noises) {
    (void)nears;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)n_alive; ++n) {
        const int32_t id = rays_alive[n];
        orc_ray r;
        orc_ray_init(&r, rays_o + 3 * (size_t)id, rays_d + 3 * (size_t)id, grid, bound, dt_gamma, max_steps, C, H);
        float* px = xyzs + 3 * (size_t)n * n_step;
        float* pd = dirs + 3 * (size_t)n * n_step;
        float* pl = deltas + 2 * (size_t)n * n_step;
        float t = rays_t[id];
        const float far_ = fars[id];
        t = fmaf(orc_clampf(t * dt_gamma, r.dt_min, r.dt_max), noises[n], t); /* .cu:751 */
        uint32_t k = 0;
        orc_probe p;
        while (t < far_ && k < n_step) {
            orc_probe_at(&r, t, &p);
            if (p.occupied) {
                px[0] = p.x; px[1] = p.y; px[2] = p.z;
                pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                pl[0] = p.dt; pl[1] = t;
                t += p.dt;
                px += 3; pd += 3; pl += 2; ++k;
            } else t = orc_skip_empty(&r, &p, t);
        }
    }
}

/* composite_rays, inference (raymarching.cu:825-913): in place.  T = 1 - sum(w) is read BEFORE the sample is
 * added and compared with T_thresh AFTER it was accumulated (.cu:875-890).                        */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                        const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                        float* image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; ++n) {
        const int32_t id = rays_alive[n];
        const float* ps = sigmas + (size_t)n * n_step;
        const float* pc = rgbs + 3 * (size_t)n * n_step;
        const float* pl = deltas + 2 * (size_t)n * n_step;
        float ws = weights_sum[id], d = depth[id];
        float r = image[3 * id], g = image[3 * id + 1], b = image[3 * id + 2];
        uint32_t k = 0;
        while (k < n_step) {
            if (pl[2 * k] == 0.0f) break; /* an unused slot ends the ray */
            const float alpha = 1.0f - expf(-ps[k] * pl[2 * k]);
            const float T = 1.0f - ws;
            const float w = alpha * T;
            ws += w;
            d = fmaf(w, pl[2 * k + 1], d);
            r = fmaf(w, pc[3 * k + 0], r);
            g = fmaf(w, pc[3 * k + 1], g);
            b = fmaf(w, pc[3 * k + 2], b);
            if (T < T_thresh) break;
            ++k;
        }
        if (k < n_step) rays_alive[n] = -1;
        else rays_t[id] = pl[2 * (n_step - 1) + 1] + pl[2 * (n_step - 1)]; /* t_last + dt_last (.cu:905) */
        weights_sum[id] = ws; depth[id] = d;
        image[3 * id] = r; image[3 * id + 1] = g; image[3 * id + 2] = b;
    }
}

/* Whole-ray reference-shaped render of ONE scene's rays with an externally supplied decoder is
 * done in Python (oracle/render.py), which drives orc_march_rays / orc_composite_rays exactly the
 * way lib/models/decoders/base_volume_renderer.py:79-123 drives the CUDA ops.                     */
