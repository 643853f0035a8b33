// ssdnerf_amd/csrc/decode.hip -- Part 2 of the C ABI: triplane repack, fused point decode and the fused
// density-grid refresh (BaseNeRF.update_extra_state), for gfx950.
#include "decode_core.h"
#include "decode_bwd_math.h"

static_assert(SSDB_OFF_WD == MLP_OFF_WD && SSDB_OFF_BD == MLP_OFF_BD && SSDB_OFF_TAIL == MLP_OFF_TAIL, "parameter block layout: decode_core.h and decode_bwd_math.h must agree");

static constexpr unsigned DEC_TPB = 256;

// ------------------------------------------------------------------------------------------------
// (S,3,C,H,W) NCHW -> (S,3,H,W,8) channel-last, zero padded.  One lane per texel: C strided-but-coalesced
// 4-byte reads (lanes walk W), one 32-byte (fp32) / 16-byte (fp16) store.
template <typename IT, typename OT> SSD_DEV OT ssd_cvt(IT v);
template <> SSD_DEV float ssd_cvt<float, float>(float v) { return v; }
template <> SSD_DEV float ssd_cvt<__half, float>(__half v) { return __half2float(v); }
template <> SSD_DEV __half ssd_cvt<float, __half>(float v) { return __float2half(v); }
template <> SSD_DEV __half ssd_cvt<__half, __half>(__half v) { return v; }

template <typename IT, typename OT>
__global__ void k_triplane_pack(const IT* __restrict__ code, uint32_t n_planes, uint32_t Cch, uint32_t HW, OT* __restrict__ planes) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_planes * HW) return;
    const uint32_t pl = (uint32_t)(t / HW), px = (uint32_t)(t - (uint64_t)pl * HW);
    const IT* src = code + (uint64_t)pl * Cch * HW + px;
    OT v[8];
#pragma unroll
    for (uint32_t c = 0; c < 8; ++c) v[c] = c < Cch ? ssd_cvt<IT, OT>(src[(uint64_t)c * HW]) : ssd_cvt<float, OT>(0.0f);
    OT* dst = planes + t * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) dst[c] = v[c];
}

extern "C" int ssdnerf_triplane_pack(const void* code, int code_dtype, uint32_t S, uint32_t Cch, uint32_t Hp, uint32_t Wp, void* planes,
                                     int planes_dtype, void* stream) {
    SSD_REQUIRE(code && planes, "triplane_pack: null pointer");
    SSD_REQUIRE(Cch >= 1 && Cch <= 8, "triplane_pack: channels per plane must be in [1, 8] (got %u)", Cch);
    SSD_REQUIRE((code_dtype == 0 || code_dtype == 1) && (planes_dtype == 0 || planes_dtype == 1), "triplane_pack: unsupported dtype");
    const uint64_t total = (uint64_t)S * 3 * Hp * Wp;
    if (total == 0) return SSDNERF_OK;
    dim3 g(ssd_blocks(total, DEC_TPB)), b(DEC_TPB);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t np = S * 3, HW = Hp * Wp;
    if (code_dtype == 0 && planes_dtype == 0) hipLaunchKernelGGL((k_triplane_pack<float, float>), g, b, 0, s, (const float*)code, np, Cch, HW, (float*)planes);
    else if (code_dtype == 0) hipLaunchKernelGGL((k_triplane_pack<float, __half>), g, b, 0, s, (const float*)code, np, Cch, HW, (__half*)planes);
    else if (planes_dtype == 0) hipLaunchKernelGGL((k_triplane_pack<__half, float>), g, b, 0, s, (const __half*)code, np, Cch, HW, (float*)planes);
    else hipLaunchKernelGGL((k_triplane_pack<__half, __half>), g, b, 0, s, (const __half*)code, np, Cch, HW, (__half*)planes);
    SSD_CHECK_LAUNCH("triplane_pack");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused point decode: one sample per lane.
template <typename PT, bool COLOR>
__global__ void __launch_bounds__(DEC_TPB) k_point_decode(const PT* __restrict__ planes, PlaneGeom g, const float* __restrict__ P,
                                                           const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t n, float sat,
                                                           float* __restrict__ sigmas, float* __restrict__ rgbs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float f[18];
    ssd_gather18<PT>(planes, g, xyzs[3ull * i], xyzs[3ull * i + 1], xyzs[3ull * i + 2], f);
    float sigma, r = 0.f, gg = 0.f, b = 0.f;
    if (COLOR) {
        float sh[16];
        shb::eval<4, false>(dirs[3ull * i], dirs[3ull * i + 1], dirs[3ull * i + 2], sh, nullptr, nullptr, nullptr);
        ssd_mlp<1>(P, f, sh, nullptr, sat, sigma, r, gg, b);
        rgbs[3ull * i] = r; rgbs[3ull * i + 1] = gg; rgbs[3ull * i + 2] = b;
    } else {
        ssd_mlp<0>(P, f, nullptr, nullptr, sat, sigma, r, gg, b);
    }
    sigmas[i] = sigma;
}

extern "C" int ssdnerf_point_decode(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params, const float* xyzs,
                                    const float* dirs, uint32_t n, float sigmoid_saturation, float* sigmas, float* rgbs, void* stream) {
    if (n == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(planes && mlp_params && xyzs && sigmas, "point_decode: null pointer");
    SSD_REQUIRE((rgbs == nullptr) == (dirs == nullptr), "point_decode: rgbs and dirs must both be given or both be NULL");
    SSD_REQUIRE(planes_dtype == 0 || planes_dtype == 1, "point_decode: unsupported plane dtype");
    SSD_REQUIRE(Hp >= 1 && Wp >= 1, "point_decode: empty plane");
    const PlaneGeom g = ssd_plane_geom(Hp, Wp);
    dim3 gr(ssd_blocks(n, DEC_TPB)), b(DEC_TPB);
    hipStream_t s = (hipStream_t)stream;
    const bool color = rgbs != nullptr;
    if (planes_dtype == 0) {
        if (color) hipLaunchKernelGGL((k_point_decode<float, true>), gr, b, 0, s, (const float*)planes, g, mlp_params, xyzs, dirs, n, sigmoid_saturation, sigmas, rgbs);
        else hipLaunchKernelGGL((k_point_decode<float, false>), gr, b, 0, s, (const float*)planes, g, mlp_params, xyzs, dirs, n, sigmoid_saturation, sigmas, rgbs);
    } else {
        if (color) hipLaunchKernelGGL((k_point_decode<__half, true>), gr, b, 0, s, (const __half*)planes, g, mlp_params, xyzs, dirs, n, sigmoid_saturation, sigmas, rgbs);
        else hipLaunchKernelGGL((k_point_decode<__half, false>), gr, b, 0, s, (const __half*)planes, g, mlp_params, xyzs, dirs, n, sigmoid_saturation, sigmas, rgbs);
    }
    SSD_CHECK_LAUNCH("point_decode");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Gradient of the point decode w.r.t. the scene codes (decoder frozen), all scenes of a batch in three launches.
//
// The r01 form -- one lane per sample scattering its 72 corner contributions with global fp32 atomics -- ran at the device's atomic
// rate (~18 G/s, tools/ubench/atomic_scope.hip): 31 ms of a 65 ms guided DDIM step (5 M samples with a gradient out of ~20 M marched),
// 40 x the forward decode of the same samples.  Now the scatter is a BINNED REDUCTION in LDS and the only global writes are plain stores:
//
//   k_decode_bwd_feat  sample -> dL/df[18] (the arithmetic of decode_bwd_math.h: re-gather, hidden units twice).  Samples WITH a gradient
//                      (not the 128-alignment padding, not the samples behind the T_thresh cut) are compacted per scene -- one atomic
//                      ticket per block on a per-scene counter -- and stored per plane as {key = y0 << 16 | x0, clipped texel coordinates
//                      (ix, iy), gf[6]}.
//   k_decode_bwd_bin   block = (scene, plane, 32 x 32-texel tile, split k of the scene's compacted samples).  Waves scan the 4-byte keys
//                      (four coalesced loads in flight, the next step's issued before this step's are looked at), collect the samples
//                      whose 2 x 2 footprint touches the tile in an LDS list (ballot + mbcnt), and 64 listed samples at a time add their
//                      corner contributions -- same weights, same products as ssdb_scatter18 -- to a 24 KiB tile image with LDS atomics.
//                      The tile is then STORED to partial[k] (every texel of every tile is written: nothing to zero-fill).
//   k_decode_bwd_sum   grad_code[s][p][c][y][x] = sum_k partial[k][s][p][y][x][c]: the gradient in the code's own NCHW layout.
//
// The order of the additions inside a tile is not fixed, so the gradient is reproducible to rounding only, as before (and as with
// ATen's grid_sampler_2d_backward, which this replaces).
#ifndef DB_TILE_EDGE
#define DB_TILE_EDGE 32
#endif
static constexpr uint32_t DB_TILE = DB_TILE_EDGE;             // texels per tile edge
static constexpr uint32_t DB_TILE_FLOATS = DB_TILE * DB_TILE * 6;
static constexpr uint32_t DB_LIST = 512;                      // per-wave ring of collected sample slots (a 256-key scan step adds at most 256 to < 64 waiting)
#ifndef DB_MAX_SPLIT_N
#define DB_MAX_SPLIT_N 32
#endif
static constexpr uint32_t DB_MAX_SPLIT = DB_MAX_SPLIT_N;                  // sample splits per (scene, plane, tile): many short blocks, so that the tiles most samples
                                                              // fall in (the object's, or where a view's rays enter the box) do not end the launch with a few long ones
static constexpr size_t DB_PARTIAL_BUDGET = (size_t)256 << 20;  // bytes of per-split tile images (r02 advisor: K was chosen from the sample count alone)
static constexpr uint32_t DB_COUNTER_STRIDE = 32;             // one 128-byte line per scene counter (same-line device atomics serialise)

struct DecodeBwdWs {
    uint32_t* counters;  // [S][32]: samples with a gradient, per scene
    uint32_t* keys;      // [3][total]      the three arrays below are indexed by compacted slot: offsets[s] + rank within the scene
    float2* pos;         // [3][total]
    float* gfeat;        // [3][total][6]
    float* partial;      // [K][S][3][Hp][Wp][6]
    uint32_t K;
    size_t counter_bytes, bytes;
};
static DecodeBwdWs db_workspace(void* base, uint32_t S, uint32_t total, uint32_t Hp, uint32_t Wp) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    DecodeBwdWs w;
    const uint32_t per_scene = S ? total / S : 0;
    w.K = per_scene / 16384u;
    w.K = w.K < 1 ? 1 : (w.K > DB_MAX_SPLIT ? DB_MAX_SPLIT : w.K);
    // the per-split tile images are written and re-read once per backward: keep them under DB_PARTIAL_BUDGET whatever the plane size
    // (8 scenes of 128 x 128 planes: 9.4 MB per split -> 27 splits; 256 x 256 planes: 37.7 MB per split -> 6)
    const size_t per_split = (size_t)(S ? S : 1) * 3 * Hp * Wp * 6 * sizeof(float);
    const size_t fit = per_split ? DB_PARTIAL_BUDGET / per_split : DB_MAX_SPLIT;
    if (w.K > fit) w.K = fit < 1 ? 1u : (uint32_t)fit;
    char* p = (char*)base;
    size_t off = 0;
    w.counter_bytes = up((size_t)S * DB_COUNTER_STRIDE * sizeof(uint32_t));
    w.counters = (uint32_t*)(p + off); off += w.counter_bytes;
    w.keys = (uint32_t*)(p + off);     off += up((size_t)3 * total * sizeof(uint32_t));
    w.pos = (float2*)(p + off);        off += up((size_t)3 * total * sizeof(float2));
    w.gfeat = (float*)(p + off);       off += up((size_t)3 * total * 6 * sizeof(float));
    w.partial = (float*)(p + off);     off += up((size_t)w.K * S * 3 * Hp * Wp * 6 * sizeof(float));
    w.bytes = off;
    return w;
}

extern "C" size_t ssdnerf_point_decode_backward_workspace(uint32_t S, uint32_t total, uint32_t Hp, uint32_t Wp) {
    return db_workspace(nullptr, S, total, Hp, Wp).bytes;
}

SSD_DEV uint32_t db_scene_of(const uint32_t* __restrict__ offsets, uint32_t S, uint32_t i) {
    uint32_t scene = 0;
    while (scene + 1 < S && i >= offsets[scene + 1]) ++scene;
    return scene;
}

template <typename PT, bool COLOR>
__global__ void __launch_bounds__(DEC_TPB) k_decode_bwd_feat(const PT* __restrict__ planes, PlaneGeom g, uint64_t plane_stride, const float* __restrict__ P,
                                                              const float* __restrict__ xyzs, const float* __restrict__ dirs,
                                                              const uint32_t* __restrict__ offsets, uint32_t S, uint32_t total, float sat,
                                                              const float* __restrict__ g_sigmas, const float* __restrict__ g_rgbs,
                                                              uint32_t* __restrict__ counters, uint32_t* __restrict__ keys, float2* __restrict__ pos,
                                                              float* __restrict__ gfeat) {
    __shared__ uint32_t wave_count[DEC_TPB / 64];
    __shared__ uint32_t block_base;
    const uint32_t first = blockIdx.x * blockDim.x, i = first + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gs = 0.0f, gc[3] = {0.0f, 0.0f, 0.0f};
    if (i < total) {
        gs = g_sigmas ? g_sigmas[i] : 0.0f;
        if (COLOR) { gc[0] = g_rgbs[3ull * i]; gc[1] = g_rgbs[3ull * i + 1]; gc[2] = g_rgbs[3ull * i + 2]; }
    }
    const bool active = gs != 0.0f || gc[0] != 0.0f || gc[1] != 0.0f || gc[2] != 0.0f;
    // ---- compacted slot of this sample within its scene
    const uint32_t scene_first = db_scene_of(offsets, S, first), scene_last = db_scene_of(offsets, S, min(first + blockDim.x, total) - 1u);
    uint32_t scene = scene_first, slot = 0;
    const uint64_t am = __ballot(active);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
    if (scene_first == scene_last) {                 // the usual case: ONE ticket per block
        if (lane == 0) wave_count[wave] = (uint32_t)__popcll(am);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t n = 0;
            for (uint32_t w = 0; w < DEC_TPB / 64; ++w) n += wave_count[w];
            block_base = n ? atomicAdd(counters + scene * DB_COUNTER_STRIDE, n) : 0u;
        }
        __syncthreads();
        slot = block_base + rank;
        for (int w = 0; w < wave; ++w) slot += wave_count[w];
    } else {                                         // a block that straddles a scene boundary (at most S - 1 of them): one ticket per (wave, scene)
        scene = i < total ? db_scene_of(offsets, S, i) : scene_last;
        uint64_t todo = am;
        while (todo != 0) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t s0 = __builtin_amdgcn_readlane(scene, leader);
            const uint64_t m = __ballot(active && scene == s0);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(counters + s0 * DB_COUNTER_STRIDE, (uint32_t)__popcll(m));
            base = __builtin_amdgcn_readlane(base, leader);
            if (active && scene == s0) slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            todo &= ~m;
        }
    }
    if (!active) return;
    const uint64_t j = (uint64_t)offsets[scene] + slot;
    const float x = xyzs[3ull * i], y = xyzs[3ull * i + 1], z = xyzs[3ull * i + 2];
    float f[18], gf[18], sh[16];
    ssd_gather18<PT>(planes + scene * plane_stride, g, x, y, z, f);
    if (COLOR) shb::eval<4, false>(dirs[3ull * i], dirs[3ull * i + 1], dirs[3ull * i + 2], sh, nullptr, nullptr, nullptr);
    ssdb_mlp_backward(P, f, COLOR ? sh : f, sat, gs, gc, COLOR ? 1 : 0, gf);
    const float us[3] = {x, x, y}, vs[3] = {y, z, z};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float ix = ssdb_unnormalise(us[p], g.Wp), iy = ssdb_unnormalise(vs[p], g.Hp);
        keys[(uint64_t)p * total + j] = ((uint32_t)floorf(iy) << 16) | (uint32_t)floorf(ix);
        pos[(uint64_t)p * total + j] = make_float2(ix, iy);
        float2* dst = reinterpret_cast<float2*>(gfeat + ((uint64_t)p * total + j) * 6);
        dst[0] = make_float2(gf[0 + p], gf[3 + p]);
        dst[1] = make_float2(gf[6 + p], gf[9 + p]);
        dst[2] = make_float2(gf[12 + p], gf[15 + p]);
    }
}

__global__ void __launch_bounds__(DEC_TPB) k_decode_bwd_bin(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counters, uint32_t S,
                                                             uint32_t total, uint32_t Hp, uint32_t Wp, uint32_t K, uint32_t tiles_x,
                                                             const uint32_t* __restrict__ keys, const float2* __restrict__ pos,
                                                             const float* __restrict__ gfeat, float* __restrict__ partial) {
    __shared__ float acc[DB_TILE_FLOATS];
    __shared__ uint32_t lists[DEC_TPB / 64][DB_LIST];
    const uint32_t tile = blockIdx.x, p = blockIdx.y / K, k = blockIdx.y % K, scene = blockIdx.z;
    const uint32_t tx0 = (tile % tiles_x) * DB_TILE, ty0 = (tile / tiles_x) * DB_TILE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t e = threadIdx.x; e < DB_TILE_FLOATS; e += DEC_TPB) acc[e] = 0.0f;
    __syncthreads();
    const uint32_t o0 = offsets[scene], o1 = o0 + counters[scene * DB_COUNTER_STRIDE];      // the scene's compacted samples
    const uint32_t chunk = (((o1 - o0) + K - 1) / K + DEC_TPB - 1) / DEC_TPB * DEC_TPB;
    const uint32_t k0 = min(o1, o0 + k * chunk), k1 = min(o1, k0 + chunk);
    const uint32_t* kp = keys + (uint64_t)p * total;
    const float2* pp = pos + (uint64_t)p * total;
    const float* gp = gfeat + (uint64_t)p * total * 6;
    uint32_t* list = lists[wave];
    uint32_t head = 0, cnt = 0;
    // lanes [0, n): one listed sample each.  Listed samples are in march order, so runs of neighbouring lanes sit on the SAME texel -- whole
    // waves of them on the plane a view looks down on -- and same-address LDS atomics serialise (~20 cycles per lane measured: 31 k cycles per
    // 64-sample call on such a plane).  So equal-key runs are first summed inside each 16-lane row (segmented inclusive scan, 4 DPP row-shift
    // steps: one v_fmac with a DPP operand per value and step) and only the last lane of a run adds to the tile.
    auto add_listed = [&](uint32_t n) {
        const bool on = (uint32_t)lane < n;
        uint32_t x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        float v[4][6];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 6; ++c) v[q][c] = 0.0f;
        uint32_t key = 0xffffffffu;
        if (on) {
            const uint32_t j = list[(head + lane) % DB_LIST];
            const float2 xy = pp[j];
            float wx0, wx1, wy0, wy1;
            ssdb_corners(xy.x, Wp, &x0, &x1, &wx0, &wx1);
            ssdb_corners(xy.y, Hp, &y0, &y1, &wy0, &wy1);
            key = (y0 << 16) | x0;
            const float2* src = reinterpret_cast<const float2*>(gp + (uint64_t)j * 6);
            const float2 g01 = src[0], g23 = src[1], g45 = src[2];
            const float gv[6] = {g01.x, g01.y, g23.x, g23.y, g45.x, g45.y};
            const float w[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 6; ++c) v[q][c] = gv[c] * w[q];
        }
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, true);          // row_shr:1
        const uint64_t heads = __ballot((lane & 15) == 0 || key != prev);                                           // first lane of every run (per 16-lane row)
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int d = 1 << step;
            // lane - d is in this lane's run iff no run starts in (lane - d, lane]
            const bool same = (lane & 15) >= d && ((heads >> (lane - d + 1)) & ((1ull << d) - 1ull)) == 0;
            const float take = same ? 1.0f : 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float up = __int_as_float(step == 0 ? __builtin_amdgcn_update_dpp(0, __float_as_int(v[q][c]), 0x111, 0xf, 0xf, true)
                                                    : step == 1 ? __builtin_amdgcn_update_dpp(0, __float_as_int(v[q][c]), 0x112, 0xf, 0xf, true)
                                                    : step == 2 ? __builtin_amdgcn_update_dpp(0, __float_as_int(v[q][c]), 0x114, 0xf, 0xf, true)
                                                                : __builtin_amdgcn_update_dpp(0, __float_as_int(v[q][c]), 0x118, 0xf, 0xf, true));
                    v[q][c] = __builtin_fmaf(up, take, v[q][c]);
                }
        }
        const bool tail = on && ((lane & 15) == 15 || (uint32_t)lane + 1 == n || ((heads >> (lane + 1)) & 1ull));
        if (tail) {
            const uint32_t cx[4] = {x0, x1, x0, x1}, cy[4] = {y0, y0, y1, y1};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t lx = cx[q] - tx0, ly = cy[q] - ty0;
                // (a clamped border corner, x1 == x0, carries weight 0 in every sample of the run: its sum is +-0 and adding it changes nothing)
                if (lx < DB_TILE && ly < DB_TILE) {
                    float* t = acc + (ly * DB_TILE + lx) * 6;
#ifdef DB_EXP_NO_ATOMICS                                                         // (measurement only: racy plain adds -- what the pass costs without the LDS atomics)
#pragma unroll
                    for (int c = 0; c < 6; ++c) t[c] += v[q][c];
#else
#pragma unroll
                    for (int c = 0; c < 6; ++c) atomicAdd(t + c, v[q][c]);
#endif
                }
            }
        }
    };
    constexpr uint32_t STEP = 4 * 64;
    auto load_keys = [&](uint32_t base, uint32_t (&kk)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = base + 64 * j + lane;
            kk[j] = i < k1 ? kp[i] : 0xffffffffu;
        }
    };
    uint32_t cur[4], nxt[4];
    uint32_t base = k0 + wave * STEP;
    load_keys(base, cur);
    for (; base < k1; base += (DEC_TPB / 64) * STEP) {
        load_keys(base + (DEC_TPB / 64) * STEP, nxt);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the 2 x 2 footprint {x0, x0+1} x {y0, y0+1} touches the tile iff x0 in [tx0 - 1, tx0 + 31] and the same for y
            const bool hit = ((cur[j] & 0xffffu) + 1u - tx0) <= DB_TILE && ((cur[j] >> 16) + 1u - ty0) <= DB_TILE;
            const uint64_t hm = __ballot(hit);
            if (hm == 0) continue;
            if (hit) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                list[(head + cnt + rank) % DB_LIST] = base + 64 * j + lane;
            }
            cnt += (uint32_t)__popcll(hm);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        while (cnt >= 64) {
#ifndef DB_EXP_SCAN_ONLY                                                          // (measurement only: the key scan and the lists alone)
            add_listed(64);
#endif
            head = (head + 64) % DB_LIST;
            cnt -= 64;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
    add_listed(cnt);
    __syncthreads();
    float* out = partial + (((uint64_t)k * S + scene) * 3 + p) * Hp * Wp * 6;
    for (uint32_t e = threadIdx.x; e < DB_TILE_FLOATS; e += DEC_TPB) {
        const uint32_t ly = e / (DB_TILE * 6), r = e % (DB_TILE * 6);
        const uint32_t y = ty0 + ly, xc = tx0 * 6 + r;
        if (y < Hp && xc < Wp * 6) out[(uint64_t)y * Wp * 6 + xc] = acc[e];
    }
}

// r06: the same reduction WITHOUT floating-point LDS atomics.  tools/r06_decode_bwd_split.sh: of k_decode_bwd_bin's 2.7 ms (7 M uniform samples; 1.9 ms ray-ordered)
// 0.64 ms are the key scan, the lists, the contributions and the tile stores -- the rest is `ds_add_f32`, which the LDS retires at ~2.4 cycles PER LANE (a 64-lane
// instruction ~150 cycles; plain 4-byte reads and writes run at 32 lanes per cycle).  So the additions become plain read-modify-writes, made safe by OWNERSHIP:
//   * the tile's four 16 x 16 quadrants belong to the block's four waves -- no two waves ever touch the same texel.  The waves still share the key scan (each scans a quarter
//     of a step's 1024 keys) and hand the samples over through four per-quadrant rings in LDS (a sample whose 2 x 2 footprint crosses a quadrant border is listed in each
//     quadrant it touches; every consumer adds only the corners inside its own); one barrier behind a step's listing, one behind its draining (rings of 1152: fewer than 64
//     left over + at most 1024 new);
//   * inside a wave, lanes that target the same texel are serialised by a one-word election per round: every pending lane writes its id to owner[texel], reads it back, the
//     one that finds itself there adds its six values (three 8-byte reads and writes) and retires; LDS instructions of a wave execute in order, so a later round reads what an
//     earlier round wrote.  Equal-texel RUNS of neighbouring lanes (march order: whole rays on one texel of the plane a view looks down on) are summed in registers first,
//     as before (DPP row scans), so an election sees at most one lane per run.
// Same products, same per-run sums; the order of the additions into a texel differs (it was arbitrary before as well).
static constexpr uint32_t DBQ_CAP = 1152;
__global__ void __launch_bounds__(DEC_TPB) k_decode_bwd_bin_owned(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counters, uint32_t S,
                                                                   uint32_t total, uint32_t Hp, uint32_t Wp, uint32_t K, uint32_t tiles_x,
                                                                   const uint32_t* __restrict__ keys, const float2* __restrict__ pos,
                                                                   const float* __restrict__ gfeat, float* __restrict__ partial) {
    static_assert(DB_TILE == 32 && DEC_TPB == 256, "four waves, four 16 x 16 quadrants");
    __shared__ __attribute__((aligned(16))) float acc[DB_TILE_FLOATS];
    __shared__ uint32_t qlist[4][DBQ_CAP];
    __shared__ uint32_t owner[4][256];
    __shared__ uint32_t qtail[4];
    const uint32_t tile = blockIdx.x, p = blockIdx.y / K, k = blockIdx.y % K, scene = blockIdx.z;
    const uint32_t tx0 = (tile % tiles_x) * DB_TILE, ty0 = (tile / tiles_x) * DB_TILE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t e = threadIdx.x; e < DB_TILE_FLOATS; e += DEC_TPB) acc[e] = 0.0f;
    if (threadIdx.x < 4) qtail[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t o0 = offsets[scene], o1 = o0 + counters[scene * DB_COUNTER_STRIDE];
    const uint32_t chunk = (((o1 - o0) + K - 1) / K + DEC_TPB - 1) / DEC_TPB * DEC_TPB;
    const uint32_t k0 = min(o1, o0 + k * chunk), k1 = min(o1, k0 + chunk);
    const uint32_t* kp = keys + (uint64_t)p * total;
    const float2* pp = pos + (uint64_t)p * total;
    const float* gp = gfeat + (uint64_t)p * total * 6;
    uint32_t* const list = qlist[wave];
    uint32_t* const own = owner[wave];
    const uint32_t qx0 = (uint32_t)(wave & 1) * 16u, qy0 = (uint32_t)(wave >> 1) * 16u;       // this wave's quadrant inside the tile
    uint32_t head = 0;                                                                          // items of this quadrant's ring consumed so far
    auto add_listed = [&](uint32_t n) {
        const bool on = (uint32_t)lane < n;
        uint32_t x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        float v[4][6];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 6; ++c) v[q][c] = 0.0f;
        uint32_t key = 0xffffffffu;
        if (on) {
            const uint32_t j = list[(head + lane) % DBQ_CAP];
            const float2 xy = pp[j];
            float wx0, wx1, wy0, wy1;
            ssdb_corners(xy.x, Wp, &x0, &x1, &wx0, &wx1);
            ssdb_corners(xy.y, Hp, &y0, &y1, &wy0, &wy1);
            key = (y0 << 16) | x0;
            const float2* src = reinterpret_cast<const float2*>(gp + (uint64_t)j * 6);
            const float2 g01 = src[0], g23 = src[1], g45 = src[2];
            const float gv[6] = {g01.x, g01.y, g23.x, g23.y, g45.x, g45.y};
            const float w[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 6; ++c) v[q][c] = gv[c] * w[q];
        }
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, true);          // row_shr:1
        const uint64_t heads = __ballot((lane & 15) == 0 || key != prev);                                           // first lane of every run (per 16-lane row)
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int d = 1 << step;
            const bool same = (lane & 15) >= d && ((heads >> (lane - d + 1)) & ((1ull << d) - 1ull)) == 0;
            const float take = same ? 1.0f : 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float up = __int_as_float(step == 0 ? __builtin_amdgcn_update_dpp(0, __float_as_int(v[q][c]), 0x111, 0xf, 0xf, true)
                                                    : step == 1 ? __builtin_amdgcn_update_dpp(0, __float_as_int(v[q][c]), 0x112, 0xf, 0xf, true)
                                                    : step == 2 ? __builtin_amdgcn_update_dpp(0, __float_as_int(v[q][c]), 0x114, 0xf, 0xf, true)
                                                                : __builtin_amdgcn_update_dpp(0, __float_as_int(v[q][c]), 0x118, 0xf, 0xf, true));
                    v[q][c] = __builtin_fmaf(up, take, v[q][c]);
                }
        }
        const bool tail = on && ((lane & 15) == 15 || (uint32_t)lane + 1 == n || ((heads >> (lane + 1)) & 1ull));
        const uint32_t cx[4] = {x0, x1, x0, x1}, cy[4] = {y0, y0, y1, y1};
        // (corner by corner: ONE election over all four corners per round -- fewer dependent LDS round trips -- was measured slower, 2.38 against 2.25 ms: 256 (lane, corner)
        // pairs on the quadrant's 256 texels collide far more often than 64 lanes do)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (a clamped border corner, x1 == x0, carries weight 0 in every sample of the run: its sum is +-0 and adding it changes nothing)
            const uint32_t lx = cx[q] - tx0 - qx0, ly = cy[q] - ty0 - qy0;                      // inside this wave's quadrant iff both < 16 (unsigned)
            bool pending = tail && lx < 16u && ly < 16u;
            const uint32_t tex = (ly & 15u) * 16u + (lx & 15u);
            float2* const t2 = reinterpret_cast<float2*>(acc + (((ly & 15u) + qy0) * DB_TILE + (lx & 15u) + qx0) * 6);
            while (__ballot(pending) != 0ull) {
                if (pending) own[tex] = (uint32_t)lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t who = own[tex];                                                // (the election's answer and the texel's values in ONE round trip: every
                float2 a = t2[0], b = t2[1], c = t2[2];                                       //  address is inside the wave's own arrays, so the reads are unconditional)
                const bool win = pending && who == (uint32_t)lane;
                if (win) {
                    a.x += v[q][0]; a.y += v[q][1]; b.x += v[q][2]; b.y += v[q][3]; c.x += v[q][4]; c.y += v[q][5];
                    t2[0] = a; t2[1] = b; t2[2] = c;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                pending = pending && !win;
            }
        }
    };
    constexpr uint32_t STEP = 4 * 64;
    auto load_keys = [&](uint32_t base, uint32_t (&kk)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = base + 64 * j + lane;
            kk[j] = i < k1 ? kp[i] : 0xffffffffu;
        }
    };
    uint32_t cur[4], nxt[4];
    load_keys(k0 + wave * STEP, cur);
    for (uint32_t it = k0; it < k1; it += (DEC_TPB / 64) * STEP) {                              // (block-uniform trip count: the barriers below)
        const uint32_t base = it + wave * STEP;
        load_keys(base + (DEC_TPB / 64) * STEP, nxt);
        // the footprint {x0, x0 + 1} touches quadrant column 0 (tile columns 0 - 15) iff x0 - tx0 in [-1, 15], column 1 iff in [15, 31]; rows alike.
        // All sixteen (key slot, quadrant) masks first, then ONE LDS atomic per quadrant and step (lanes 0 - 3, one instruction): a returning atomic per non-empty
        // mask was up to sixteen dependent LDS round trips per step on uniformly scattered samples (0.8 ms of the launch)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t ux = (cur[j] & 0xffffu) + 1u - tx0, uy = (cur[j] >> 16) + 1u - ty0;
            const bool hx[2] = {ux <= 16u, ux - 16u <= 16u}, hy[2] = {uy <= 16u, uy - 16u <= 16u};
            if (__ballot((hx[0] || hx[1]) && (hy[0] || hy[1])) == 0ull) continue;            // (march-ordered samples: most key slots of most blocks end here)
            // the four quadrant masks of this key slot, then ONE LDS atomic instruction (lanes 0 - 3) for their ring positions: a returning atomic per non-empty mask
            // was up to sixteen dependent LDS round trips per step on uniformly scattered samples
            uint64_t hm[4];
            uint32_t nq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hm[q] = __ballot(hx[q & 1] && hy[q >> 1]);
                nq[q] = (uint32_t)__popcll(hm[q]);
            }
            const uint32_t mine = lane == 0 ? nq[0] : lane == 1 ? nq[1] : lane == 2 ? nq[2] : nq[3];
            uint32_t first = 0;
            if (lane < 4 && mine != 0u) first = atomicAdd(&qtail[lane], mine);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (hm[q] == 0ull) continue;
                const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)first, q);
                if (hx[q & 1] && hy[q >> 1]) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm[q], 0u));
                    qlist[q][(at + rank) % DBQ_CAP] = base + 64 * j + lane;
                }
            }
        }
        __syncthreads();                                                                      // this step's samples are listed
        const uint32_t tail_now = *reinterpret_cast<volatile uint32_t*>(&qtail[wave]);
#pragma unroll 1
        while (tail_now - head >= 64u) {
#ifndef DB_EXP_SCAN_ONLY
            add_listed(64);
#endif
            head += 64u;
        }
        __syncthreads();                                                                      // every ring is below 64 before the next step's (at most 1024) items arrive
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
    add_listed(*reinterpret_cast<volatile uint32_t*>(&qtail[wave]) - head);
    __syncthreads();
    float* out = partial + (((uint64_t)k * S + scene) * 3 + p) * Hp * Wp * 6;
    for (uint32_t e = threadIdx.x; e < DB_TILE_FLOATS; e += DEC_TPB) {
        const uint32_t ly = e / (DB_TILE * 6), r = e % (DB_TILE * 6);
        const uint32_t y = ty0 + ly, xc = tx0 * 6 + r;
        if (y < Hp && xc < Wp * 6) out[(uint64_t)y * Wp * 6 + xc] = acc[e];
    }
}

__global__ void __launch_bounds__(DEC_TPB) k_decode_bwd_sum(const float* __restrict__ partial, uint32_t K, uint64_t n_texels, uint32_t HW,
                                                             float* __restrict__ grad_code) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;    // (scene, plane, y, x)
    if (t >= n_texels) return;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (uint32_t k = 0; k < K; ++k) {
        const float2* src = reinterpret_cast<const float2*>(partial + ((uint64_t)k * n_texels + t) * 6);
        const float2 a = src[0], b = src[1], c = src[2];
        s[0] += a.x; s[1] += a.y; s[2] += b.x; s[3] += b.y; s[4] += c.x; s[5] += c.y;
    }
    const uint64_t pl = t / HW, px = t - pl * HW;
#pragma unroll
    for (int c = 0; c < 6; ++c) grad_code[(pl * 6 + c) * HW + px] = s[c];
}

extern "C" int ssdnerf_point_decode_backward(const void* planes, int planes_dtype, uint32_t S, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                             const float* xyzs, const float* dirs, const uint32_t* offsets, uint32_t total, float sigmoid_saturation,
                                             const float* grad_sigmas, const float* grad_rgbs, float* grad_code, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    if (S == 0) return SSDNERF_OK;
    SSD_REQUIRE(grad_code, "point_decode_backward: null grad_code");
    SSD_REQUIRE(Hp >= 1 && Wp >= 1 && Hp <= 32768 && Wp <= 32768, "point_decode_backward: plane size out of range");
    hipStream_t s = (hipStream_t)stream;
    const uint64_t n_texels = (uint64_t)S * 3 * Hp * Wp;
    if (total == 0) {
        if (hipMemsetAsync(grad_code, 0, n_texels * 6 * sizeof(float), s) != hipSuccess) return ssdnerf_fail(SSDNERF_E_LAUNCH, "point_decode_backward: memset failed");
        return SSDNERF_OK;
    }
    SSD_REQUIRE(planes && mlp_params && xyzs && offsets && workspace, "point_decode_backward: null pointer");
    SSD_REQUIRE((grad_rgbs == nullptr) == (dirs == nullptr), "point_decode_backward: grad_rgbs and dirs must both be given or both be NULL");
    SSD_REQUIRE(grad_sigmas || grad_rgbs, "point_decode_backward: no upstream gradient given");
    SSD_REQUIRE(planes_dtype == 0 || planes_dtype == 1, "point_decode_backward: unsupported plane dtype");
    const DecodeBwdWs w = db_workspace(workspace, S, total, Hp, Wp);
    if (workspace_bytes < w.bytes) return ssdnerf_fail(SSDNERF_E_WORKSPACE, "point_decode_backward: workspace too small");
    if (hipMemsetAsync(w.counters, 0, w.counter_bytes, s) != hipSuccess) return ssdnerf_fail(SSDNERF_E_LAUNCH, "point_decode_backward: memset failed");
    const PlaneGeom g = ssd_plane_geom(Hp, Wp);
    const uint64_t plane_stride = (uint64_t)3 * Hp * Wp * 8;
    dim3 gr(ssd_blocks(total, DEC_TPB)), b(DEC_TPB);
    const bool color = grad_rgbs != nullptr;
#define SSD_LAUNCH_FEAT(PT, COL) hipLaunchKernelGGL((k_decode_bwd_feat<PT, COL>), gr, b, 0, s, (const PT*)planes, g, plane_stride, mlp_params, xyzs, dirs, offsets, S, \
                                                    total, sigmoid_saturation, grad_sigmas, grad_rgbs, w.counters, w.keys, w.pos, w.gfeat)
    if (planes_dtype == 0) { if (color) SSD_LAUNCH_FEAT(float, true); else SSD_LAUNCH_FEAT(float, false); }
    else { if (color) SSD_LAUNCH_FEAT(__half, true); else SSD_LAUNCH_FEAT(__half, false); }
#undef SSD_LAUNCH_FEAT
    SSD_CHECK_LAUNCH("point_decode_backward (features)");
    const uint32_t tiles_x = (Wp + DB_TILE - 1) / DB_TILE, tiles_y = (Hp + DB_TILE - 1) / DB_TILE;
    static const bool lds_atomics = [] { const char* e = getenv("SSDNERF_DECODE_BWD_ATOMICS"); return e != nullptr && e[0] != '\0' && e[0] != '0'; }();   // the r02 - r05 kernel (A/B runs)
    if (lds_atomics)
        hipLaunchKernelGGL(k_decode_bwd_bin, dim3(tiles_x * tiles_y, 3 * w.K, S), b, 0, s, offsets, w.counters, S, total, Hp, Wp, w.K, tiles_x, w.keys, w.pos,
                           w.gfeat, w.partial);
    else
        hipLaunchKernelGGL(k_decode_bwd_bin_owned, dim3(tiles_x * tiles_y, 3 * w.K, S), b, 0, s, offsets, w.counters, S, total, Hp, Wp, w.K, tiles_x, w.keys, w.pos,
                           w.gfeat, w.partial);
    SSD_CHECK_LAUNCH("point_decode_backward (binned reduction)");
    hipLaunchKernelGGL(k_decode_bwd_sum, dim3(ssd_blocks(n_texels, DEC_TPB)), b, 0, s, w.partial, w.K, n_texels, Hp * Wp, grad_code);
    SSD_CHECK_LAUNCH("point_decode_backward (sum)");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused density-grid refresh: cell centre (+ injected jitter) -> density decode -> max-EMA into the
// Morton-ordered grid -> block-reduced contribution to mean(max(grid, 0)).
template <typename T> SSD_DEV float ssd_grid_ld(const T* p);
template <> SSD_DEV float ssd_grid_ld<float>(const float* p) { return *p; }
template <> SSD_DEV float ssd_grid_ld<__half>(const __half* p) { return __half2float(*p); }
template <typename T> SSD_DEV T ssd_grid_round(float v);
template <> SSD_DEV float ssd_grid_round<float>(float v) { return v; }
template <> SSD_DEV __half ssd_grid_round<__half>(float v) { return __float2half(v); }
template <typename T> SSD_DEV float ssd_grid_rt(float v) { T t = ssd_grid_round<T>(v); return ssd_grid_ld<T>(&t); }  // round to the grid dtype, back to fp32
template <typename T> SSD_DEV float ssd_grid_max();
template <> SSD_DEV float ssd_grid_max<float>() { return 3.402823466e+38f; }
template <> SSD_DEV float ssd_grid_max<__half>() { return 65504.0f; }

static constexpr uint32_t DU_CHUNKS = 8;          // 256-cell chunks per block of k_density_update
template <typename PT, typename GT>
__global__ void __launch_bounds__(DEC_TPB) k_density_update(const PT* __restrict__ planes, PlaneGeom g, const float* __restrict__ P, uint32_t H,
                                                             float centre, float cell, float half_cell, const float* __restrict__ jitter,
                                                             float decay, GT* __restrict__ grid, float inv_count, float* __restrict__ mean_out) {
    const uint32_t H3 = H * H * H;
    const uint32_t s = blockIdx.y;
    float contrib = 0.0f;
#pragma unroll 1
    for (uint32_t chunk = 0; chunk < DU_CHUNKS; ++chunk) {
    const uint32_t n = (blockIdx.x * DU_CHUNKS + chunk) * blockDim.x + threadIdx.x;  // cell in x-major order (custom_meshgrid ij)
    if (n < H3) {
        const uint32_t cz = n % H, cy = (n / H) % H, cx = n / (H * H);
        float xyz[3];
        const uint32_t cc[3] = {cx, cy, cz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = ((float)cc[a] - centre) * cell;                   // (coords - (H-1)/2) * (2*bound/H)
            if (jitter) v = v + (jitter[3ull * n + a] * (2.0f * half_cell) - half_cell);
            xyz[a] = v;
        }
        float f[18];
        const PT* pl = planes + (uint64_t)s * 3 * g.Hp * g.Wp * 8;
        ssd_gather18<PT>(pl, g, xyz[0], xyz[1], xyz[2], f);
        float sigma, r, gg, b;
        ssd_mlp<0>(P, f, nullptr, nullptr, 0.0f, sigma, r, gg, b);
        const uint32_t idx = ssd_morton(cx, cy, cz);
        GT* cellp = grid + (uint64_t)s * H3 + idx;
        const float old = ssd_grid_ld<GT>(cellp);
        const float fresh = ssd_grid_rt<GT>(fminf(sigma, ssd_grid_max<GT>()));
        float out = old;
        if (old >= 0.0f && fresh >= 0.0f) {
            const float decayed = ssd_grid_rt<GT>(old * decay);  // product rounded to the grid dtype first
            out = fmaxf(decayed, fresh);
            *cellp = ssd_grid_round<GT>(out);
        }
        contrib += fmaxf(out, 0.0f);
    }
    }
    if (mean_out) {
        // ONE atomic per block of DU_CHUNKS x 256 cells (r03).  Every wave used to add its sum to the one address: 32 768 same-address device-scope
        // atomics per refresh of 8 scenes, resolved one after the other at ~10 ns each -- 330 of the kernel's 434 us (profiles/r03/s_pmc_final.txt:
        // 1 883 VALU instructions per wave in 154 k cycles).
        __shared__ float wave_sum[DEC_TPB / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
        if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = contrib;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (uint32_t w = 0; w < DEC_TPB / 64; ++w) t += wave_sum[w];
            atomicAdd(mean_out, t * inv_count);
        }
    }
}

extern "C" int ssdnerf_density_grid_update(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params, uint32_t S,
                                           uint32_t grid_size, float bound, const float* jitter, float decay, void* density_grid, int grid_dtype,
                                           float* mean_out, void* stream) {
    if (S == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(planes && mlp_params && density_grid, "density_grid_update: null pointer");
    SSD_REQUIRE((planes_dtype == 0 || planes_dtype == 1) && (grid_dtype == 0 || grid_dtype == 1), "density_grid_update: unsupported dtype");
    SSD_REQUIRE(grid_size >= 1 && grid_size <= 1024, "density_grid_update: grid_size out of range");
    const PlaneGeom g = ssd_plane_geom(Hp, Wp);
    const uint32_t H3 = grid_size * grid_size * grid_size;
    const float centre = (float)((double)(grid_size - 1) / 2.0);
    const float cell = (float)(2.0 * (double)bound / (double)grid_size);
    const float half_cell = (float)((double)bound / (double)grid_size);
    const float inv_count = (float)(1.0 / ((double)S * (double)H3));
    dim3 gr(ssd_blocks(H3, DEC_TPB * DU_CHUNKS), S), b(DEC_TPB);
    hipStream_t s = (hipStream_t)stream;
#define SSD_LAUNCH_DU(PT, GT) hipLaunchKernelGGL((k_density_update<PT, GT>), gr, b, 0, s, (const PT*)planes, g, mlp_params, grid_size, centre, cell, \
                                                 half_cell, jitter, decay, (GT*)density_grid, inv_count, mean_out)
    if (planes_dtype == 0 && grid_dtype == 0) SSD_LAUNCH_DU(float, float);
    else if (planes_dtype == 0) SSD_LAUNCH_DU(float, __half);
    else if (grid_dtype == 0) SSD_LAUNCH_DU(__half, float);
    else SSD_LAUNCH_DU(__half, __half);
#undef SSD_LAUNCH_DU
    SSD_CHECK_LAUNCH("density_grid_update");
    return SSDNERF_OK;
}
