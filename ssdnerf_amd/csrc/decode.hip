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
// Gradient of the point decode w.r.t. the planes (decoder frozen): one sample per lane, the arithmetic of decode_bwd_math.h --
// re-gather the 18 features, run the 64 hidden units twice (outputs, then gradient), scatter d/df to the 3 x 4 bilinear corners
// with fp32 hardware atomics into a (3, Hp, Wp, 8) gradient image (1.5 MiB per scene at 128^2: L2-resident, like the planes).
// Points whose upstream gradient is exactly zero (the 128-alignment padding, samples behind the T_thresh cut) return at once.
template <typename PT, bool COLOR>
__global__ void __launch_bounds__(DEC_TPB) k_point_decode_bwd(const PT* __restrict__ planes, PlaneGeom g, const float* __restrict__ P,
                                                               const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t n, float sat,
                                                               const float* __restrict__ g_sigmas, const float* __restrict__ g_rgbs,
                                                               float* __restrict__ gplanes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gs = g_sigmas ? g_sigmas[i] : 0.0f;
    float gc[3] = {0.0f, 0.0f, 0.0f};
    if (COLOR) { gc[0] = g_rgbs[3ull * i]; gc[1] = g_rgbs[3ull * i + 1]; gc[2] = g_rgbs[3ull * i + 2]; }
    if (gs == 0.0f && gc[0] == 0.0f && gc[1] == 0.0f && gc[2] == 0.0f) return;
    const float x = xyzs[3ull * i], y = xyzs[3ull * i + 1], z = xyzs[3ull * i + 2];
    float f[18], gf[18], sh[16];
    ssd_gather18<PT>(planes, g, x, y, z, f);
    if (COLOR) shb::eval<4, false>(dirs[3ull * i], dirs[3ull * i + 1], dirs[3ull * i + 2], sh, nullptr, nullptr, nullptr);
    ssdb_mlp_backward(P, f, COLOR ? sh : f, sat, gs, gc, COLOR ? 1 : 0, gf);
    ssdb_scatter18(gplanes, g.Hp, g.Wp, x, y, z, gf);
}

extern "C" int ssdnerf_point_decode_backward(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params, const float* xyzs,
                                             const float* dirs, uint32_t n, float sigmoid_saturation, const float* grad_sigmas, const float* grad_rgbs,
                                             float* grad_planes, void* stream) {
    if (n == 0) return SSDNERF_OK;
    SSD_REQUIRE(planes && mlp_params && xyzs && grad_planes, "point_decode_backward: null pointer");
    SSD_REQUIRE((grad_rgbs == nullptr) == (dirs == nullptr), "point_decode_backward: grad_rgbs and dirs must both be given or both be NULL");
    SSD_REQUIRE(grad_sigmas || grad_rgbs, "point_decode_backward: no upstream gradient given");
    SSD_REQUIRE(planes_dtype == 0 || planes_dtype == 1, "point_decode_backward: unsupported plane dtype");
    SSD_REQUIRE(Hp >= 1 && Wp >= 1, "point_decode_backward: empty plane");
    const PlaneGeom g = ssd_plane_geom(Hp, Wp);
    dim3 gr(ssd_blocks(n, DEC_TPB)), b(DEC_TPB);
    hipStream_t s = (hipStream_t)stream;
    const bool color = grad_rgbs != nullptr;
    if (planes_dtype == 0) {
        if (color) hipLaunchKernelGGL((k_point_decode_bwd<float, true>), gr, b, 0, s, (const float*)planes, g, mlp_params, xyzs, dirs, n, sigmoid_saturation, grad_sigmas, grad_rgbs, grad_planes);
        else hipLaunchKernelGGL((k_point_decode_bwd<float, false>), gr, b, 0, s, (const float*)planes, g, mlp_params, xyzs, dirs, n, sigmoid_saturation, grad_sigmas, grad_rgbs, grad_planes);
    } else {
        if (color) hipLaunchKernelGGL((k_point_decode_bwd<__half, true>), gr, b, 0, s, (const __half*)planes, g, mlp_params, xyzs, dirs, n, sigmoid_saturation, grad_sigmas, grad_rgbs, grad_planes);
        else hipLaunchKernelGGL((k_point_decode_bwd<__half, false>), gr, b, 0, s, (const __half*)planes, g, mlp_params, xyzs, dirs, n, sigmoid_saturation, grad_sigmas, grad_rgbs, grad_planes);
    }
    SSD_CHECK_LAUNCH("point_decode_backward");
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

template <typename PT, typename GT>
__global__ void __launch_bounds__(DEC_TPB) k_density_update(const PT* __restrict__ planes, PlaneGeom g, const float* __restrict__ P, uint32_t H,
                                                             float centre, float cell, float half_cell, const float* __restrict__ jitter,
                                                             float decay, GT* __restrict__ grid, float inv_count, float* __restrict__ mean_out) {
    const uint32_t H3 = H * H * H;
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;  // cell in x-major order (custom_meshgrid ij)
    const uint32_t s = blockIdx.y;
    float contrib = 0.0f;
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
        contrib = fmaxf(out, 0.0f);
    }
    if (mean_out) {
        // wave reduce, then one atomic per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(mean_out, contrib * inv_count);
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
    dim3 gr(ssd_blocks(H3, DEC_TPB), S), b(DEC_TPB);
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
