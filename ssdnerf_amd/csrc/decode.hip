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
#ifdef DB_EXP_NO_MLP                                                             // (measurement only: everything of the kernel but the MLP)
#pragma unroll
    for (int k = 0; k < 18; ++k) gf[k] = f[k] * gs + (COLOR ? sh[k & 15] * gc[k % 3] : 0.0f);
#else
    ssdb_mlp_backward(P, f, COLOR ? sh : f, sat, gs, gc, COLOR ? 1 : 0, gf);
#endif
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

// ================================================================================================================================
// r06: k_decode_bwd_feat on the matrix cores (colour + density heads; the density-only form keeps the kernel above).
// The per-lane form computes every hidden unit twice (registers) with 6 300 fused multiply-adds and 512 transcendentals per sample on the vector ALUs -- 1.6 ms for 7 M
// samples, the larger half of the decode backward once the reduction lost its LDS atomics.  Here a wave takes 64 samples through three small matrix products
//   H  [64 hidden x 64 samples] = W1 [64 x 19] F [19 x 64]      (features + a bias row)            24 MFMA 32x32x16 (bf16 pair split: hi hi + hi lo + lo hi)
//   D  [64 x 64]                = Wd [64 x 16] SH [16 x 64]                                       12
//   GF [18 x 64]                = W1^T [18 x 64] dPRE [64 x 64]                                    24
// and everything between them stays in the products' C layout (a lane holds one sample column -- two, one per 32-sample tile -- and 16 of a row tile's 32 hidden
// units): silu, the two heads' partial dot products (their other half sits in lane ^ 32: one v_permlane32_swap per value), silu' stored over H and D, the combination
// dPRE = dsa w_sigma silu'(h) + (dz . Wc) silu'(u), whose registers ARE the B operand of the third product (the k order of W1^T's fragments is permuted to the C layout's
// row order).  Hidden units once, 256 transcendentals per sample.  Persistent blocks (the weights' fragments are built once per block into LDS).
// Arithmetic class: fp32 accumulation of bf16-pair products (>= 16 significand bits per factor), as the UNet's fp32-class kernels; the reduction downstream is unchanged.
typedef __bf16 db_bf16x8 __attribute__((ext_vector_type(8)));
typedef float db_f32x16 __attribute__((ext_vector_type(16)));
typedef float db_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 db_b2 __attribute__((ext_vector_type(2)));
SSD_DEV uint32_t db_cvt2(float even, float odd) {                            // {bf16(even), bf16(odd)}, round to nearest even: v_cvt_pk_bf16_f32
    const db_b2 r = __builtin_convertvector(db_f2{even, odd}, db_b2);
    return *reinterpret_cast<const uint32_t*>(&r);
}
// two values -> their packed hi terms and packed lo terms (x ~= hi + lo to 2^-17: both roundings to nearest)
SSD_DEV void db_split_pair(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = db_cvt2(x0, x1);
    lo = db_cvt2(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}
SSD_DEV db_bf16x8 db_op(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint4 u = make_uint4(a, b, c, d);
    return *reinterpret_cast<const db_bf16x8*>(&u);
}
// v_permlane32_swap(a, b): lanes 32..63 of a trade places with lanes 0..31 of b
SSD_DEV void db_swap(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// 8 fp32 values -> the hi and lo operand vectors
SSD_DEV void db_operands(const float* v, db_bf16x8& hi, db_bf16x8& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) db_split_pair(v[2 * i], v[2 * i + 1], h[i], l[i]);
    hi = db_op(h[0], h[1], h[2], h[3]);
    lo = db_op(l[0], l[1], l[2], l[3]);
}
SSD_DEV db_f32x16 db_mfma3(const db_bf16x8& a_hi, const db_bf16x8& a_lo, const db_bf16x8& b_hi, const db_bf16x8& b_lo, db_f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc, 0, 0, 0);
}
// LDS fragment tables (uint4 = one lane's 8 bf16 of an A operand), built once per block
struct DbFrags {
    uint4 a1[2][2][2][64];     // layer 1: [row tile][k step][hi | lo][lane]; k step 0 = features 0 - 15, k step 1 = {16, 17, bias, 0 ...}
    uint4 ad[2][2][64];        // direction layer: [row tile][hi | lo][lane]
    uint4 at[4][2][64];        // W1^T: [k step = (row tile, half)][hi | lo][lane]; rows = features (18 used), k = hidden units in the C layout's order
    float4 head[64];           // {w_sigma, w_c0, w_c1, w_c2} per hidden unit
    float bd[64];
};
template <typename PT>
__global__ void __launch_bounds__(DEC_TPB, 2) k_decode_bwd_feat_mfma(const PT* __restrict__ planes, PlaneGeom g, uint64_t plane_stride, const float* __restrict__ P,
                                                                   const float* __restrict__ xyzs, const float* __restrict__ dirs,
                                                                   const uint32_t* __restrict__ offsets, uint32_t S, uint32_t total, float sat,
                                                                   const float* __restrict__ g_sigmas, const float* __restrict__ g_rgbs,
                                                                   uint32_t* __restrict__ counters, uint32_t* __restrict__ keys, float2* __restrict__ pos,
                                                                   float* __restrict__ gfeat) {
    __shared__ DbFrags fr;
    __shared__ uint32_t wave_count[DEC_TPB / 64];
    __shared__ uint32_t block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t hf = (uint32_t)lane >> 5, ln = (uint32_t)lane & 31u;
    // ---- the weights' operand fragments
    auto put = [&](uint4* hi_dst, uint4* lo_dst, const float (&v)[8]) {
        uint32_t h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) db_split_pair(v[2 * i], v[2 * i + 1], h[i], l[i]);
        *hi_dst = make_uint4(h[0], h[1], h[2], h[3]);
        *lo_dst = make_uint4(l[0], l[1], l[2], l[3]);
    };
    for (uint32_t e = threadIdx.x; e < 2 * 2 * 64 + 2 * 64 + 4 * 64; e += DEC_TPB) {
        float v[8];
        if (e < 256) {                                                        // layer 1: (mt, ks, lane)
            const uint32_t mt = e >> 7, ks = (e >> 6) & 1u, l = e & 63u, i = 32u * mt + (l & 31u);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = 16u * ks + 8u * (l >> 5) + (uint32_t)j;
                v[j] = k <= 18u ? P[i * 24u + k] : 0.0f;                      // (k == 18: the bias, multiplied by the 1.0 row of the features)
            }
            put(&fr.a1[mt][ks][0][l], &fr.a1[mt][ks][1][l], v);
        } else if (e < 384) {                                                 // direction layer: (mt, lane)
            const uint32_t q = e - 256u, mt = q >> 6, l = q & 63u, i = 32u * mt + (l & 31u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = P[SSDB_OFF_WD + i * 16u + 8u * (l >> 5) + (uint32_t)j];
            put(&fr.ad[mt][0][l], &fr.ad[mt][1][l], v);
        } else {                                                              // W1^T: (k step s = 2 mt + e2, lane); row = feature l % 32
            const uint32_t q = e - 384u, s4 = q >> 6, l = q & 63u, k = l & 31u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t i = 32u * (s4 >> 1) + 16u * (s4 & 1u) + (uint32_t)(j < 4 ? j : j + 4) + 4u * (l >> 5);      // the C layout's row of slot j in this lane half
                v[j] = k < 18u ? P[i * 24u + k] : 0.0f;
            }
            put(&fr.at[s4][0][l], &fr.at[s4][1][l], v);
        }
    }
    if (threadIdx.x < 64) {
        const float* rec = P + threadIdx.x * 24u;
        fr.head[threadIdx.x] = make_float4(rec[19], rec[20], rec[21], rec[22]);
        fr.bd[threadIdx.x] = P[SSDB_OFF_BD + threadIdx.x];
    }
    __syncthreads();
    const float b_sigma = P[SSDB_OFF_TAIL + 0], b_c0 = P[SSDB_OFF_TAIL + 1], b_c1 = P[SSDB_OFF_TAIL + 2], b_c2 = P[SSDB_OFF_TAIL + 3];
    const uint4* a1 = &fr.a1[0][0][0][0];
    for (uint32_t first = blockIdx.x * DEC_TPB; first < total; first += gridDim.x * DEC_TPB) {       // (block-uniform trip count)
        const uint32_t i = first + threadIdx.x;
        float gs = 0.0f, gc[3] = {0.0f, 0.0f, 0.0f};
        if (i < total) {
            gs = g_sigmas ? g_sigmas[i] : 0.0f;
            gc[0] = g_rgbs[3ull * i]; gc[1] = g_rgbs[3ull * i + 1]; gc[2] = g_rgbs[3ull * i + 2];
        }
        const bool active = gs != 0.0f || gc[0] != 0.0f || gc[1] != 0.0f || gc[2] != 0.0f;
        // ---- compacted slot of this sample within its scene (as k_decode_bwd_feat)
        const uint32_t scene_first = db_scene_of(offsets, S, first), scene_last = db_scene_of(offsets, S, min(first + DEC_TPB, total) - 1u);
        uint32_t scene = scene_first, slot = 0;
        const uint64_t am = __ballot(active);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
        if (scene_first == scene_last) {
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
            __syncthreads();                                                  // (wave_count / block_base are reused by the next trip)
        } else {
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
        if (am == 0ull) continue;                                             // (wave-uniform: nothing of this wave's 64 samples carries a gradient)
        float x = 0.f, y = 0.f, z = 0.f, f[18], sh[16];
#pragma unroll
        for (int k = 0; k < 18; ++k) f[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sh[k] = 0.0f;
        if (active) {
            x = xyzs[3ull * i]; y = xyzs[3ull * i + 1]; z = xyzs[3ull * i + 2];
            ssd_gather18<PT>(planes + scene * plane_stride, g, x, y, z, f);
            shb::eval<4, false>(dirs[3ull * i], dirs[3ull * i + 1], dirs[3ull * i + 2], sh, nullptr, nullptr, nullptr);
        }
        // ---- B operands.  After f[k] <-> f[8 + k] a lane holds, for tile 0 (samples 0 - 31) in f[0..7] and for tile 1 (samples 32 - 63) in f[8..15], the features
        // 8 hf .. 8 hf + 7 of the tile's column lane % 32: exactly its k slots of the k step
#pragma unroll
        for (int k = 0; k < 8; ++k) { db_swap(f[k], f[8 + k]); db_swap(sh[k], sh[8 + k]); }
        float a16 = f[16], b16 = f[16], a17 = f[17], b17 = f[17];            // swap(a = b = X): a = X of samples 0 - 31 in both halves, b = X of samples 32 - 63
        db_swap(a16, b16); db_swap(a17, b17);
        float gsa = gs, gsb = gs, g0a = gc[0], g0b = gc[0], g1a = gc[1], g1b = gc[1], g2a = gc[2], g2b = gc[2];      // ... and the upstream gradients of the tiles' columns
        db_swap(gsa, gsb); db_swap(g0a, g0b); db_swap(g1a, g1b); db_swap(g2a, g2b);
        // ---- one 32-sample tile at a time (the two tiles are independent up to the last exchange; together their accumulators do not fit two waves per SIMD)
        auto tile = [&](const float* fv, const float* sv, float v16, float v17, float ugs, float ug0, float ug1, float ug2) -> db_f32x16 {
            db_bf16x8 bfh, bfl, bsh, bsl, bkh, bkl;                          // features 0 - 15, SH, and the k step {f16, f17, 1, 0 ...}
            db_operands(fv, bfh, bfl);
            db_operands(sv, bsh, bsl);
            {
                const float one = hf == 0u ? 1.0f : 0.0f;                     // the k step's slots 0 - 7 belong to lane half 0: half 1 multiplies zeros
                const float tk[8] = {hf == 0u ? v16 : 0.f, hf == 0u ? v17 : 0.f, one, 0.f, 0.f, 0.f, 0.f, 0.f};
                db_operands(tk, bkh, bkl);
            }
            // H = W1 F + b1, D = Wd SH   (C layout: register r of lane l = hidden unit 32 mt + 8 (r / 4) + 4 hf + r % 4 of the tile's sample l % 32)
            db_f32x16 H[2], Dd[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { H[mt][r] = 0.0f; Dd[mt][r] = 0.0f; }
                const db_bf16x8 w0h = *reinterpret_cast<const db_bf16x8*>(&fr.a1[mt][0][0][lane]), w0l = *reinterpret_cast<const db_bf16x8*>(&fr.a1[mt][0][1][lane]);
                const db_bf16x8 w1h = *reinterpret_cast<const db_bf16x8*>(&fr.a1[mt][1][0][lane]), w1l = *reinterpret_cast<const db_bf16x8*>(&fr.a1[mt][1][1][lane]);
                const db_bf16x8 wdh = *reinterpret_cast<const db_bf16x8*>(&fr.ad[mt][0][lane]), wdl = *reinterpret_cast<const db_bf16x8*>(&fr.ad[mt][1][lane]);
                H[mt] = db_mfma3(w1h, w1l, bkh, bkl, H[mt]);
                H[mt] = db_mfma3(w0h, w0l, bfh, bfl, H[mt]);
                Dd[mt] = db_mfma3(wdh, wdl, bsh, bsl, Dd[mt]);
            }
            // pass 1: the heads' partial dot products over this lane's 32 hidden units; silu'(h) over H, silu'(u) over D
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t iu = 32u * mt + 8u * (r >> 2) + 4u * hf + (r & 3);
                    const float4 hw = fr.head[iu];
                    const float h = H[mt][r], u = h + (Dd[mt][r] + fr.bd[iu]);
                    const float sg = ssdb_sigmoid(h), su = ssdb_sigmoid(u);
                    part[0] = fmaf(hw.x, h * sg, part[0]);
                    const float c = u * su;
                    part[1] = fmaf(hw.y, c, part[1]); part[2] = fmaf(hw.z, c, part[2]); part[3] = fmaf(hw.w, c, part[3]);
                    H[mt][r] = sg * fmaf(h, 1.0f - sg, 1.0f);
                    Dd[mt][r] = su * fmaf(u, 1.0f - su, 1.0f);
                }
            // the other 32 hidden units of the column sit in lane ^ 32: swap(a = b = v) leaves the lower half's values in a and the upper half's in b, in every lane
            float tot[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { float a = part[q], b = part[q]; db_swap(a, b); tot[q] = a + b; }
            const float sa = tot[0] + b_sigma, z0 = tot[1] + b_c0, z1 = tot[2] + b_c1, z2 = tot[3] + b_c2;
            const float e = SSDB_EXP2(sa * 1.4426950408889634f);
            const float dsa = ugs * fminf(1e6f, fmaxf(e, 1e-6f));
            const float kk = fmaf(sat, 2.0f, 1.0f);
            const float s0 = ssdb_sigmoid(z0), s1 = ssdb_sigmoid(z1), s2 = ssdb_sigmoid(z2);
            const float dz0 = ug0 * kk * (s0 * (1.0f - s0)), dz1 = ug1 * kk * (s1 * (1.0f - s1)), dz2 = ug2 * kk * (s2 * (1.0f - s2));
            // pass 2 and GF = W1^T dPRE: the registers of H are the B operand (k step (mt, e2) = registers 8 e2 .. 8 e2 + 7)
            db_f32x16 G;
#pragma unroll
            for (int r = 0; r < 16; ++r) G[r] = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t iu = 32u * mt + 8u * (r >> 2) + 4u * hf + (r & 3);
                    const float4 hw = fr.head[iu];
                    const float dc = fmaf(dz2, hw.w, fmaf(dz1, hw.z, dz0 * hw.y));
                    H[mt][r] = fmaf(dc, Dd[mt][r], dsa * hw.x * H[mt][r]);
                }
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    float v8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v8[j] = H[mt][8 * e2 + j];
                    db_bf16x8 bh, bl;
                    db_operands(v8, bh, bl);
                    const db_bf16x8 ah = *reinterpret_cast<const db_bf16x8*>(&fr.at[2 * mt + e2][0][lane]), al = *reinterpret_cast<const db_bf16x8*>(&fr.at[2 * mt + e2][1][lane]);
                    G = db_mfma3(ah, al, bh, bl, G);
                }
            }
            return G;
        };
        db_f32x16 G[2];
        G[0] = tile(f, sh, a16, a17, gsa, g0a, g1a, g2a);
        __builtin_amdgcn_sched_barrier(0);
        G[1] = tile(f + 8, sh + 8, b16, b17, gsb, g0b, g1b, g2b);
        // ---- back to one sample per lane: register r holds feature 8 (r / 4) + 4 hf + r % 4 of the tiles' columns; swap(a = G[0][r], b = G[1][r]) leaves in a the
        // hf = 0 rows and in b the hf = 1 rows of the lane's OWN sample (tile 0 for lanes 0 - 31, tile 1 for lanes 32 - 63)
        float gf[18];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            float a = G[0][r], b = G[1][r];
            db_swap(a, b);
            const int k0 = 8 * (r >> 2) + (r & 3);
            if (k0 < 18) gf[k0] = a;
            if (k0 + 4 < 18 && r < 8) gf[k0 + 4] = b;
        }
        if (!active) continue;
        const uint64_t j = (uint64_t)offsets[scene] + slot;
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
        // pairs on the quadrant's 256 texels collide far more often than 64 lanes do; two elections over DIAGONAL corner pairs: 1.59 against 1.53 ms on march-ordered samples)
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
    const bool feat_mfma = (planes_dtype & SSDNERF_DECODE_BWD_FEAT_MFMA) != 0;       // (r06, opt-in) the per-sample feature gradients on the matrix cores
    planes_dtype &= 0xff;
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
    if (feat_mfma && color) {                                                // (r06) the matrix-core form: persistent blocks, two per CU
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        }
        const uint32_t want = ssd_blocks(total, DEC_TPB), cap = (uint32_t)n_cu * 2u;
        dim3 gm(want < cap ? want : cap);
        if (planes_dtype == 0)
            hipLaunchKernelGGL((k_decode_bwd_feat_mfma<float>), gm, b, 0, s, (const float*)planes, g, plane_stride, mlp_params, xyzs, dirs, offsets, S, total,
                               sigmoid_saturation, grad_sigmas, grad_rgbs, w.counters, w.keys, w.pos, w.gfeat);
        else
            hipLaunchKernelGGL((k_decode_bwd_feat_mfma<__half>), gm, b, 0, s, (const __half*)planes, g, plane_stride, mlp_params, xyzs, dirs, offsets, S, total,
                               sigmoid_saturation, grad_sigmas, grad_rgbs, w.counters, w.keys, w.pos, w.gfeat);
    } else if (planes_dtype == 0) { if (color) SSD_LAUNCH_FEAT(float, true); else SSD_LAUNCH_FEAT(float, false); }
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
