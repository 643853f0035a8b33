// ssdnerf_amd/csrc/raymarching_ops.hip -- Part 1 of the C ABI: the ten ray-marching operators the
// reference exposes through its `_raymarching` pybind module (lib/ops/raymarching/src/bindings.cpp:5-18),
// written for gfx950.  Integer/byte-bound, HBM-streaming kernels: 256-thread blocks (4 waves), one
// work item per lane, coalesced SoA-style accesses where the reference's [N,3] layout allows it.
#include "common.h"

thread_local char g_ssdnerf_err[512] = {0};

extern "C" const char* ssdnerf_last_error(void) { return g_ssdnerf_err; }
extern "C" int ssdnerf_abi_version(void) { return 3; }   // 2 (r03): conv2d_nhwc_f32x2 takes a split-K scratch; group_norm_nhwc_runs; render workspace holds 8-byte survivor entries
                                                           // 3 (r04): conv2d_nhwc_f32x2_presplit(_supported) (ksize, hints, scratch), marching cubes, act bit 1 of the GroupNorm calls

static constexpr unsigned TPB = 256;

// ------------------------------------------------------------------------------------------------
__global__ void k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ aabb,
                           uint32_t N, float min_near, float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const RayGeom r = ssd_load_ray(rays_o + 3ull * n, rays_d + 3ull * n);
    float a, b;
    ssd_near_far(aabb, r, min_near, a, b);
    nears[n] = a;
    fars[n] = b;
}

extern "C" int ssdnerf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                                          float* nears, float* fars, void* stream) {
    if (N == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb: null pointer");
    hipLaunchKernelGGL(k_near_far, dim3(ssd_blocks(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, rays_o, rays_d, aabb, N, min_near, nears, fars);
    SSD_CHECK_LAUNCH("near_far_from_aabb");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Background-sphere coordinates (exported by the reference, never called by it).
__global__ void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float radius, uint32_t N,
                               float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* o = rays_o + 3ull * n;
    const float* d = rays_d + 3ull * n;
    const float rpi = 0.3183098861837907f;
    const float A = ssd_fma(d[2], d[2], ssd_fma(d[1], d[1], d[0] * d[0]));
    const float Bh = ssd_fma(o[2], d[2], ssd_fma(o[1], d[1], o[0] * d[0]));
    const float Cq = ssd_fma(o[2], o[2], ssd_fma(o[1], o[1], o[0] * o[0])) - radius * radius;
    const float t = (-Bh + sqrtf(Bh * Bh - A * Cq)) / A;  // far intersection
    const float x = ssd_fma(t, d[0], o[0]), y = ssd_fma(t, d[1], o[1]), z = ssd_fma(t, d[2], o[2]);
    const float theta = atan2f(sqrtf(ssd_fma(z, z, x * x)), y);
    const float phi = atan2f(z, x);
    coords[2ull * n + 0] = ssd_fma(2.0f * theta, rpi, -1.0f);
    coords[2ull * n + 1] = phi * rpi;
}

extern "C" int ssdnerf_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream) {
    if (N == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(rays_o && rays_d && coords, "sph_from_ray: null pointer");
    hipLaunchKernelGGL(k_sph_from_ray, dim3(ssd_blocks(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, rays_o, rays_d, radius, N, coords);
    SSD_CHECK_LAUNCH("sph_from_ray");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void k_morton3D(const int32_t* __restrict__ coords, uint32_t N, int32_t* __restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)ssd_morton((uint32_t)coords[3ull * n], (uint32_t)coords[3ull * n + 1], (uint32_t)coords[3ull * n + 2]);
}
__global__ void k_morton3D_invert(const int32_t* __restrict__ indices, uint32_t N, int32_t* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t v = indices[n];
    coords[3ull * n + 0] = (int32_t)ssd_compact3((uint32_t)(v >> 0));
    coords[3ull * n + 1] = (int32_t)ssd_compact3((uint32_t)(v >> 1));
    coords[3ull * n + 2] = (int32_t)ssd_compact3((uint32_t)(v >> 2));
}

extern "C" int ssdnerf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream) {
    if (N == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(coords && indices, "morton3D: null pointer");
    hipLaunchKernelGGL(k_morton3D, dim3(ssd_blocks(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, coords, N, indices);
    SSD_CHECK_LAUNCH("morton3D");
    return SSDNERF_OK;
}
extern "C" int ssdnerf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream) {
    if (N == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(coords && indices, "morton3D_invert: null pointer");
    hipLaunchKernelGGL(k_morton3D_invert, dim3(ssd_blocks(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, indices, N, coords);
    SSD_CHECK_LAUNCH("morton3D_invert");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// packbits: one lane produces one output byte from 8 consecutive cells.  fp32: two 16-byte loads,
// fp16: one 16-byte load per lane -> fully coalesced 2 KiB / 1 KiB per wave.
template <typename T> SSD_DEV float ssd_cellf(T v);
template <> SSD_DEV float ssd_cellf<float>(float v) { return v; }
template <> SSD_DEV float ssd_cellf<__half>(__half v) { return __half2float(v); }

template <typename T, bool DEV_THRESH>
__global__ void k_packbits(const T* __restrict__ grid, uint32_t N, const float* __restrict__ mean, float thresh, uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (DEV_THRESH) thresh = fminf(*mean, thresh);
    const T* g = grid + 8ull * n;
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) bits |= (unsigned)(ssd_cellf<T>(g[i]) > thresh) << i;
    bitfield[n] = (uint8_t)bits;
}

static int packbits_impl(const void* grid, int grid_dtype, uint32_t N, const float* mean, float thresh, uint8_t* bitfield, void* stream) {
    if (N == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(grid && bitfield, "packbits: null pointer");
    SSD_REQUIRE(grid_dtype == SSDNERF_DTYPE_F32 || grid_dtype == SSDNERF_DTYPE_F16, "packbits: unsupported grid dtype %d", grid_dtype);
    dim3 g(ssd_blocks(N, TPB)), b(TPB);
    hipStream_t s = (hipStream_t)stream;
    if (grid_dtype == SSDNERF_DTYPE_F32) {
        if (mean) hipLaunchKernelGGL((k_packbits<float, true>), g, b, 0, s, (const float*)grid, N, mean, thresh, bitfield);
        else hipLaunchKernelGGL((k_packbits<float, false>), g, b, 0, s, (const float*)grid, N, mean, thresh, bitfield);
    } else {
        if (mean) hipLaunchKernelGGL((k_packbits<__half, true>), g, b, 0, s, (const __half*)grid, N, mean, thresh, bitfield);
        else hipLaunchKernelGGL((k_packbits<__half, false>), g, b, 0, s, (const __half*)grid, N, mean, thresh, bitfield);
    }
    SSD_CHECK_LAUNCH("packbits");
    return SSDNERF_OK;
}
extern "C" int ssdnerf_packbits(const void* grid, int grid_dtype, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream) {
    return packbits_impl(grid, grid_dtype, N, nullptr, density_thresh, bitfield, stream);
}
extern "C" int ssdnerf_packbits_dev_thresh(const void* grid, int grid_dtype, uint32_t N, const float* mean, float density_thresh,
                                           uint8_t* bitfield, void* stream) {
    SSD_REQUIRE(mean, "packbits_dev_thresh: null mean pointer");
    return packbits_impl(grid, grid_dtype, N, mean, density_thresh, bitfield, stream);
}

// ------------------------------------------------------------------------------------------------
// march_rays_train: count -> exclusive scan (ray order) -> write.  The scan is a classic
// two-level block scan (1024 rays per block, wave64 shuffles), deterministic by construction.
// ------------------------------------------------------------------------------------------------
SSD_DEV float ssd_jittered_start(const MarchCfg& c, float near_, float noise) {
    return ssd_fma(ssd_clamp(near_ * c.dt_gamma, c.dt_min, c.dt_max), noise, near_);
}

// Several scenes in one launch: ray n belongs to scene n / rays_per_scene, whose bitfield starts grid_stride bytes further and whose cone angle is
// dt_gammas[scene] (rays_per_scene == N, stride 0, dt_gammas == NULL: the single-scene operator of the reference).
struct SceneBatch { uint32_t rays_per_scene; uint64_t grid_stride; const float* dt_gammas; };
SSD_DEV void ssd_select_scene(MarchCfg& c, const SceneBatch& sb, uint32_t n) {
    const uint32_t scene = n / sb.rays_per_scene;
    c.grid += scene * sb.grid_stride;
    if (sb.dt_gammas) c.dt_gamma = sb.dt_gammas[scene];
}

__global__ void k_march_train_count(MarchCfg c, SceneBatch sb, const float* __restrict__ rays_o, const float* __restrict__ rays_d, uint32_t N,
                                    uint32_t max_steps, const float* __restrict__ nears, const float* __restrict__ fars,
                                    const float* __restrict__ noises, uint32_t* __restrict__ counts) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    ssd_select_scene(c, sb, n);
    const RayGeom r = ssd_load_ray(rays_o + 3ull * n, rays_d + 3ull * n);
    const float far_ = fars[n];
    float t = ssd_jittered_start(c, nears[n], noises[n]);
    uint32_t k = 0;
    while (t < far_ && k < max_steps) {
        const Probe p = ssd_probe(c, r, t);
        if (p.occ) { ++k; t += p.dt; }
        else t = ssd_skip_empty(c, r, p, t);
    }
    counts[n] = k;
}

SSD_DEV uint32_t ssd_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// Block-level exclusive scan of `cnt` (one value per thread of a 1024-thread block); returns the exclusive
// prefix and leaves the block total in *total (valid for all threads after the call).
SSD_DEV uint32_t ssd_block_excl_scan_1024(uint32_t cnt, uint32_t* total) {
    __shared__ uint32_t wave_sums[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = ssd_wave_incl_scan(cnt, lane);
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t s = lane < 16 ? wave_sums[lane] : 0u;
        s = ssd_wave_incl_scan(s, lane);
        if (lane < 16) wave_sums[lane] = s;
    }
    __syncthreads();
    const uint32_t wave_off = wave == 0 ? 0u : wave_sums[wave - 1];
    *total = wave_sums[15];
    return wave_off + incl - cnt;
}

__global__ void __launch_bounds__(1024) k_scan_block_sums(const uint32_t* __restrict__ counts, uint32_t N, uint32_t* __restrict__ block_sums) {
    const uint32_t n = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t cnt = n < N ? counts[n] : 0u;
    uint32_t total;
    (void)ssd_block_excl_scan_1024(cnt, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Single block: exclusive scan over the block sums (chunks of 1024), also bumps the two counters.
__global__ void __launch_bounds__(1024) k_scan_top(uint32_t* __restrict__ block_sums, uint32_t n_blocks, uint32_t N, int32_t* __restrict__ counter,
                                                   uint32_t* __restrict__ bases /* [2]: point base, ray base before this call */) {
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = ssd_block_excl_scan_1024(v, &total);
        const uint32_t carry = carry_s;
        if (i < n_blocks) block_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bases[0] = (uint32_t)counter[0];
        bases[1] = (uint32_t)counter[1];
        counter[0] = (int32_t)((uint32_t)counter[0] + carry_s);
        counter[1] = (int32_t)((uint32_t)counter[1] + N);
    }
}

__global__ void __launch_bounds__(1024) k_march_train_write(MarchCfg c, SceneBatch sb, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                            uint32_t N, uint32_t M, const float* __restrict__ nears,
                                                            const float* __restrict__ fars, const float* __restrict__ noises,
                                                            const uint32_t* __restrict__ counts, const uint32_t* __restrict__ block_offs,
                                                            const uint32_t* __restrict__ bases, int32_t* __restrict__ rays, float* __restrict__ xyzs, float* __restrict__ dirs,
                                                            float* __restrict__ deltas) {
    const uint32_t n = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t cnt = n < N ? counts[n] : 0u;
    uint32_t total;
    const uint32_t ex = ssd_block_excl_scan_1024(cnt, &total);
    if (n >= N) return;
    const uint32_t off = bases[0] + block_offs[blockIdx.x] + ex;
    // the reference writes the (id, offset, count) triple into slot atomicAdd(counter+1, 1); in ray order that is
    // ray_base + n (ray_base != 0 only when the caller reuses a step_counter across calls).
    const uint32_t slot = bases[1] + n;
    if (slot < N) {  // the reference would write out of bounds here; we refuse
        rays[3ull * slot + 0] = (int32_t)n;
        rays[3ull * slot + 1] = (int32_t)off;
        rays[3ull * slot + 2] = (int32_t)cnt;
    }
    if (cnt == 0 || off + cnt > M) return;
    ssd_select_scene(c, sb, n);
    const RayGeom r = ssd_load_ray(rays_o + 3ull * n, rays_d + 3ull * n);
    const float far_ = fars[n];
    float t = ssd_jittered_start(c, nears[n], noises[n]);
    float* px = xyzs + 3ull * off;
    float* pd = dirs + 3ull * off;
    float* pl = deltas + 2ull * off;
    uint32_t k = 0;
    while (t < far_ && k < cnt) {
        const Probe p = ssd_probe(c, r, t);
        if (p.occ) {
            px[0] = p.x; px[1] = p.y; px[2] = p.z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            pl[0] = p.dt; pl[1] = t;
            t += p.dt;
            px += 3; pd += 3; pl += 2; ++k;
        } else t = ssd_skip_empty(c, r, p, t);
    }
}

extern "C" size_t ssdnerf_march_rays_train_workspace(uint32_t N) { return ((size_t)N + ssd_blocks(N, 1024) + 2) * sizeof(uint32_t); }

extern "C" int ssdnerf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                        uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                        const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                        const float* noises, void* workspace, size_t workspace_bytes, void* stream) {
    if (N == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter && noises, "march_rays_train: null pointer");
    SSD_REQUIRE(C >= 1 && C <= 8 && H >= 1 && H <= 1024 && max_steps >= 1, "march_rays_train: bad C=%u H=%u max_steps=%u", C, H, max_steps);
    if (!workspace || workspace_bytes < ssdnerf_march_rays_train_workspace(N))
        return ssdnerf_fail(SSDNERF_E_WORKSPACE, "march_rays_train: workspace %zu < %zu bytes", workspace_bytes, ssdnerf_march_rays_train_workspace(N));
    hipStream_t s = (hipStream_t)stream;
    const MarchCfg c = ssd_make_march_cfg(bound, dt_gamma, max_steps, C, H, grid);
    const uint32_t nb = ssd_blocks(N, 1024);
    uint32_t* counts = (uint32_t*)workspace;
    uint32_t* block_sums = counts + N;
    uint32_t* bases = block_sums + nb;
    const SceneBatch one = {N, 0, nullptr};
    hipLaunchKernelGGL(k_march_train_count, dim3(ssd_blocks(N, TPB)), dim3(TPB), 0, s, c, one, rays_o, rays_d, N, max_steps, nears, fars, noises, counts);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(1024), 0, s, counts, N, block_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, s, block_sums, nb, N, counter, bases);
    hipLaunchKernelGGL(k_march_train_write, dim3(nb), dim3(1024), 0, s, c, one, rays_o, rays_d, N, M, nears, fars, noises, counts, block_sums, bases, rays, xyzs, dirs,
                       deltas);
    SSD_CHECK_LAUNCH("march_rays_train");
    return SSDNERF_OK;
}

// ---- the train-branch march of VolumeRenderer.forward for ALL scenes of a batch (base_volume_renderer.py:59-77 calls march_rays_train once per
// scene, each call ending in a device->host read of the sample count, raymarching.py:268-274).  Two calls around ONE host read:
//   _count : sample count per ray (S*N rays in one launch) -> block sums -> exclusive offsets; scene_offsets[s] = first sample of scene s,
//            scene_offsets[S] = total.  The caller reads these S+1 integers, allocates EXACTLY total rows (the per-scene operator allocates N*max_steps)
//   _write : the same march again, writing (xyz, dir, dt, t) at the scanned offsets; rays[n] = (n, offset, count) with n and offset GLOBAL over the
//            batch, which is the form batch_composite_rays_train builds by rebasing.  Samples are packed without the per-scene 128-row padding of
//            the reference (padding rows are never composited; their only effect there is extra decode work).
__global__ void k_scene_offsets(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ block_offs, uint32_t S, uint32_t rays_per_scene,
                                uint32_t total_rays, const uint32_t* __restrict__ bases, int32_t* __restrict__ scene_offsets) {
    // block s: exclusive prefix of counts at ray s * rays_per_scene = offset of its 1024-block + the counts of the block's rays before it
    const uint32_t s = blockIdx.x;
    const uint32_t ray = s * rays_per_scene;                      // s == S: one past the end
    const uint32_t blk = min(ray, total_rays - 1u) / 1024u;
    const uint32_t n = blk * 1024u + threadIdx.x;
    uint32_t v = (n < ray && n < total_rays) ? counts[n] : 0u;
    uint32_t total;
    (void)ssd_block_excl_scan_1024(v, &total);
    if (threadIdx.x == 0) scene_offsets[s] = (int32_t)(bases[0] + block_offs[blk] + total);
}

extern "C" size_t ssdnerf_march_rays_train_batch_workspace(uint32_t S, uint32_t N) {
    return ssdnerf_march_rays_train_workspace(S * N) + 2 * sizeof(int32_t);
}

static int ssd_train_batch_args(const char* what, uint32_t S, uint32_t N, uint32_t C, uint32_t H, uint32_t max_steps, const void* workspace, size_t workspace_bytes) {
    SSD_REQUIRE((uint64_t)S * N <= 0x7fffffffull, "%s: too many rays", what);
    SSD_REQUIRE(C >= 1 && C <= 8 && H >= 1 && H <= 1024 && max_steps >= 1, "%s: bad C=%u H=%u max_steps=%u", what, C, H, max_steps);
    if (!workspace || workspace_bytes < ssdnerf_march_rays_train_batch_workspace(S, N))
        return ssdnerf_fail(SSDNERF_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, workspace_bytes, ssdnerf_march_rays_train_batch_workspace(S, N));
    return SSDNERF_OK;
}

extern "C" int ssdnerf_march_rays_train_batch_count(const float* rays_o, const float* rays_d, const uint8_t* grids, float bound, float dt_gamma,
                                                    const float* dt_gammas, uint32_t max_steps, uint32_t S, uint32_t N, uint32_t C, uint32_t H,
                                                    const float* nears, const float* fars, const float* noises, int32_t* scene_offsets,
                                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (S == 0 || N == 0) return SSDNERF_OK;
    SSD_REQUIRE(rays_o && rays_d && grids && nears && fars && noises && scene_offsets, "march_rays_train_batch_count: null pointer");
    if (int rc = ssd_train_batch_args("march_rays_train_batch_count", S, N, C, H, max_steps, workspace, workspace_bytes)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t R = S * N, nb = ssd_blocks(R, 1024);
    const MarchCfg c = ssd_make_march_cfg(bound, dt_gamma, max_steps, C, H, grids);
    const SceneBatch sb = {N, (uint64_t)C * H * H * H / 8, dt_gammas};
    uint32_t* counts = (uint32_t*)workspace;
    uint32_t* block_sums = counts + R;
    uint32_t* bases = block_sums + nb;
    int32_t* counter = (int32_t*)(bases + 2);
    if (hipMemsetAsync(counter, 0, 2 * sizeof(int32_t), s) != hipSuccess) return ssdnerf_fail(SSDNERF_E_LAUNCH, "march_rays_train_batch_count: memset failed");
    hipLaunchKernelGGL(k_march_train_count, dim3(ssd_blocks(R, TPB)), dim3(TPB), 0, s, c, sb, rays_o, rays_d, R, max_steps, nears, fars, noises, counts);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(1024), 0, s, counts, R, block_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, s, block_sums, nb, R, counter, bases);
    hipLaunchKernelGGL(k_scene_offsets, dim3(S + 1), dim3(1024), 0, s, counts, block_sums, S, N, R, bases, scene_offsets);
    SSD_CHECK_LAUNCH("march_rays_train_batch_count");
    return SSDNERF_OK;
}

extern "C" int ssdnerf_march_rays_train_batch_write(const float* rays_o, const float* rays_d, const uint8_t* grids, float bound, float dt_gamma,
                                                    const float* dt_gammas, uint32_t max_steps, uint32_t S, uint32_t N, uint32_t C, uint32_t H,
                                                    uint32_t M, const float* nears, const float* fars, const float* noises, float* xyzs, float* dirs,
                                                    float* deltas, int32_t* rays, void* workspace, size_t workspace_bytes, void* stream) {
    if (S == 0 || N == 0) return SSDNERF_OK;
    SSD_REQUIRE(rays_o && rays_d && grids && nears && fars && noises && rays && (M == 0 || (xyzs && dirs && deltas)), "march_rays_train_batch_write: null pointer");
    if (int rc = ssd_train_batch_args("march_rays_train_batch_write", S, N, C, H, max_steps, workspace, workspace_bytes)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t R = S * N, nb = ssd_blocks(R, 1024);
    const MarchCfg c = ssd_make_march_cfg(bound, dt_gamma, max_steps, C, H, grids);
    const SceneBatch sb = {N, (uint64_t)C * H * H * H / 8, dt_gammas};
    uint32_t* counts = (uint32_t*)workspace;                       // as left by _count on the same workspace
    uint32_t* block_sums = counts + R;
    uint32_t* bases = block_sums + nb;
    hipLaunchKernelGGL(k_march_train_write, dim3(nb), dim3(1024), 0, s, c, sb, rays_o, rays_d, R, M, nears, fars, noises, counts, block_sums, bases, rays, xyzs, dirs,
                       deltas);
    SSD_CHECK_LAUNCH("march_rays_train_batch_write");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Packed-ray compositing, train branch.
__global__ void k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                      const int32_t* __restrict__ rays, uint32_t M, uint32_t N, float T_thresh,
                                      float* __restrict__ weights_sum, float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t id = (uint32_t)rays[3ull * n], off = (uint32_t)rays[3ull * n + 1], cnt = (uint32_t)rays[3ull * n + 2];
    float T = 1.0f, r = 0.f, g = 0.f, b = 0.f, ws = 0.f, d = 0.f;
    if (cnt != 0 && off + cnt <= M) {
        for (uint32_t s = 0; s < cnt; ++s) {
            const uint64_t i = (uint64_t)off + s;
            const float2 dl = *reinterpret_cast<const float2*>(deltas + 2 * i);
            const float alpha = 1.0f - __expf(-sigmas[i] * dl.x);
            const float w = alpha * T;
            r = ssd_fma(w, rgbs[3 * i + 0], r);
            g = ssd_fma(w, rgbs[3 * i + 1], g);
            b = ssd_fma(w, rgbs[3 * i + 2], b);
            d = ssd_fma(w, dl.y, d);
            ws += w;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
        }
    }
    weights_sum[id] = ws; depth[id] = d;
    image[3ull * id] = r; image[3ull * id + 1] = g; image[3ull * id + 2] = b;
}

__global__ void k_composite_train_bwd(const float* __restrict__ grad_ws, const float* __restrict__ grad_image, const float* __restrict__ sigmas,
                                      const float* __restrict__ rgbs, const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                      const float* __restrict__ weights_sum, const float* __restrict__ image, uint32_t M, uint32_t N,
                                      float T_thresh, float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t id = (uint32_t)rays[3ull * n], off = (uint32_t)rays[3ull * n + 1], cnt = (uint32_t)rays[3ull * n + 2];
    if (cnt == 0 || off + cnt > M) return;
    const float gr = grad_image[3ull * id], gg = grad_image[3ull * id + 1], gb = grad_image[3ull * id + 2], gw = grad_ws[id];
    const float rF = image[3ull * id], gF = image[3ull * id + 1], bF = image[3ull * id + 2], wsF = weights_sum[id];
    float T = 1.0f, r = 0.f, g = 0.f, b = 0.f;
    for (uint32_t s = 0; s < cnt; ++s) {
        const uint64_t i = (uint64_t)off + s;
        const float dt = deltas[2 * i];
        const float c0 = rgbs[3 * i], c1 = rgbs[3 * i + 1], c2 = rgbs[3 * i + 2];
        const float alpha = 1.0f - __expf(-sigmas[i] * dt);
        const float w = alpha * T;
        r = ssd_fma(w, c0, r); g = ssd_fma(w, c1, g); b = ssd_fma(w, c2, b);
        T *= 1.0f - alpha;
        if (T < T_thresh) break;  // the sample that trips the threshold receives no gradient
        grad_rgbs[3 * i + 0] = gr * w;
        grad_rgbs[3 * i + 1] = gg * w;
        grad_rgbs[3 * i + 2] = gb * w;
        float acc = gr * ssd_fma(T, c0, -(rF - r));
        acc = ssd_fma(gg, ssd_fma(T, c1, -(gF - g)), acc);
        acc = ssd_fma(gb, ssd_fma(T, c2, -(bF - b)), acc);
        acc = ssd_fma(gw, 1.0f - wsF, acc);
        grad_sigmas[i] = dt * acc;
    }
}

extern "C" int ssdnerf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                                    uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth, float* image,
                                                    void* stream) {
    if (N == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    // M == 0 (no ray of the batch found an occupied cell: an empty scene, e.g. the first fitting iterations from a blank code): the sample arrays are
    // empty -- null -- and never dereferenced (every ray record says zero steps); the per-ray outputs are still written (zeros), as the reference's launch does
    SSD_REQUIRE((M == 0 || (sigmas && rgbs && deltas)) && rays && weights_sum && depth && image, "composite_rays_train_forward: null pointer");
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(ssd_blocks(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, sigmas, rgbs, deltas, rays, M, N, T_thresh,
                       weights_sum, depth, image);
    SSD_CHECK_LAUNCH("composite_rays_train_forward");
    return SSDNERF_OK;
}
extern "C" int ssdnerf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas, const float* rgbs,
                                                     const float* deltas, const int32_t* rays, const float* weights_sum, const float* image,
                                                     uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs, void* stream) {
    if (N == 0 || M == 0) return SSDNERF_OK;          // no samples: no gradient to write (empty arrays may be null)
    SSD_REQUIRE(grad_weights_sum && grad_image && sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs,
                "composite_rays_train_backward: null pointer");
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(ssd_blocks(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, grad_weights_sum, grad_image, sigmas, rgbs,
                       deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs);
    SSD_CHECK_LAUNCH("composite_rays_train_backward");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Inference pair: fixed-slot march and in-place composite over the alive list.
__global__ void k_march_rays(MarchCfg c, uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive,
                             const float* __restrict__ rays_t, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                             const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                             const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t id = (uint32_t)rays_alive[n];
    const RayGeom r = ssd_load_ray(rays_o + 3ull * id, rays_d + 3ull * id);
    const float far_ = fars[id];
    float t = rays_t[id];
    t = ssd_fma(ssd_clamp(t * c.dt_gamma, c.dt_min, c.dt_max), noises[n], t);
    float* px = xyzs + 3ull * n * n_step;
    float* pd = dirs + 3ull * n * n_step;
    float* pl = deltas + 2ull * n * n_step;
    uint32_t k = 0;
    while (t < far_ && k < n_step) {
        const Probe p = ssd_probe(c, r, t);
        if (p.occ) {
            px[0] = p.x; px[1] = p.y; px[2] = p.z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            pl[0] = p.dt; pl[1] = t;
            t += p.dt;
            px += 3; pd += 3; pl += 2; ++k;
        } else t = ssd_skip_empty(c, r, p, t);
    }
}

__global__ void k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* __restrict__ rays_alive, float* __restrict__ rays_t,
                                 const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                 float* __restrict__ weights_sum, float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t id = (uint32_t)rays_alive[n];
    const float* ps = sigmas + (uint64_t)n * n_step;
    const float* pc = rgbs + 3ull * n * n_step;
    const float* pl = deltas + 2ull * n * n_step;
    float ws = weights_sum[id], d = depth[id];
    float r = image[3ull * id], g = image[3ull * id + 1], b = image[3ull * id + 2];
    uint32_t k = 0;
    while (k < n_step) {
        const float dt = pl[2 * k];
        if (dt == 0.0f) break;  // an unused slot: the ray left the volume
        const float alpha = 1.0f - __expf(-ps[k] * dt);
        const float T = 1.0f - ws;   // transmittance BEFORE this sample ...
        const float w = alpha * T;
        ws += w;
        d = ssd_fma(w, pl[2 * k + 1], d);
        r = ssd_fma(w, pc[3 * k + 0], r);
        g = ssd_fma(w, pc[3 * k + 1], g);
        b = ssd_fma(w, pc[3 * k + 2], b);
        if (T < T_thresh) break;     // ... is what the threshold sees, after the sample was added
        ++k;
    }
    if (k < n_step) rays_alive[n] = -1;
    else rays_t[id] = pl[2 * (n_step - 1) + 1] + pl[2 * (n_step - 1)];
    weights_sum[id] = ws; depth[id] = d;
    image[3ull * id] = r; image[3ull * id + 1] = g; image[3ull * id + 2] = b;
}

extern "C" int ssdnerf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                                  const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                  const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                  const float* noises, void* stream) {
    (void)nears;
    if (n_alive == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas && noises, "march_rays: null pointer");
    SSD_REQUIRE(C >= 1 && C <= 8 && H >= 1 && H <= 1024 && max_steps >= 1 && n_step >= 1, "march_rays: bad C=%u H=%u max_steps=%u n_step=%u", C, H, max_steps, n_step);
    const MarchCfg c = ssd_make_march_cfg(bound, dt_gamma, max_steps, C, H, grid);
    hipLaunchKernelGGL(k_march_rays, dim3(ssd_blocks(n_alive, TPB)), dim3(TPB), 0, (hipStream_t)stream, c, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d,
                       fars, xyzs, dirs, deltas, noises);
    SSD_CHECK_LAUNCH("march_rays");
    return SSDNERF_OK;
}

extern "C" int ssdnerf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t, const float* sigmas,
                                      const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image, void* stream) {
    if (n_alive == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, "composite_rays: null pointer");
    SSD_REQUIRE(n_step >= 1, "composite_rays: n_step must be >= 1");
    hipLaunchKernelGGL(k_composite_rays, dim3(ssd_blocks(n_alive, TPB)), dim3(TPB), 0, (hipStream_t)stream, n_alive, n_step, T_thresh, rays_alive, rays_t,
                       sigmas, rgbs, deltas, weights_sum, depth, image);
    SSD_CHECK_LAUNCH("composite_rays");
    return SSDNERF_OK;
}
