// ssdnerf_amd/csrc/raygen.hip -- camera rays for whole batches of views (reference: get_ray_directions / get_rays / get_cam_rays,
// lib/core/utils/nerf_utils.py:17-38,41-54,57-61).
//
// The reference builds the ray arrays with ~10 eager tensor ops (linspace, broadcasted subtract/divide, stack, batched matmul,
// normalise, expand) that each stream the (S,V,H,W,3) arrays through HBM; for the bench's 8 x 251 views that chain takes ~220 ms
// on an MI355X, 17x the render it feeds.  Here it is one pass: a lane owns a pixel, the view's pose and intrinsics are wave-uniform
// scalar loads, 24 bytes are written per ray.
//     d_cam = ((x + 0.5 - cx) / fx, (y + 0.5 - cy) / fy, 1)      pixel-centre pinhole direction (IEEE divisions, as the reference)
//     d     = R d_cam,  rays_d = d / max(|d|, 1e-12)              rotate by c2w[:3,:3], L2-normalise (F.normalize's eps)
//     rays_o = c2w[:3, 3]
// HBM-bound: 24 B written per ray.
#include "common.h"

__global__ void __launch_bounds__(256) k_cam_rays(const float* __restrict__ c2w, const float* __restrict__ intrinsics, uint32_t h, uint32_t w,
                                                   float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const uint32_t view = blockIdx.y, hw = h * w;
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= hw) return;
    float o3[3], d3[3];
    ssd_cam_ray(c2w + (size_t)view * 16, intrinsics + (size_t)view * 4, pix % w, pix / w, o3, d3);       // common.h: the one statement of the arithmetic
    const size_t o = ((size_t)view * hw + pix) * 3;
    rays_d[o + 0] = d3[0]; rays_d[o + 1] = d3[1]; rays_d[o + 2] = d3[2];
    rays_o[o + 0] = o3[0]; rays_o[o + 1] = o3[1]; rays_o[o + 2] = o3[2];
}

extern "C" int ssdnerf_cam_rays(const float* c2w, const float* intrinsics, uint32_t n_views, uint32_t h, uint32_t w, float* rays_o, float* rays_d,
                                void* stream) {
    if (n_views == 0 || h == 0 || w == 0) return SSDNERF_OK;
    SSD_REQUIRE(c2w && intrinsics && rays_o && rays_d, "cam_rays: null pointer");
    SSD_REQUIRE(n_views <= 65535, "cam_rays: at most 65535 views per launch");
    hipLaunchKernelGGL(k_cam_rays, dim3(ssd_blocks((uint64_t)h * w, 256), n_views), dim3(256), 0, (hipStream_t)stream, c2w, intrinsics, h, w, rays_o, rays_d);
    SSD_CHECK_LAUNCH("cam_rays");
    return SSDNERF_OK;
}

// ------------------------------------------------------------------------------------------------
// Output quantisation of eval_and_viz (lib/models/autodecoders/base_nerf.py:551-553): clamp to [0, 1], scale by 255, round half to even
// (torch.round), store as uint8 -- the form the views are all-gathered and written in.  One pass: 16 B read, 4 B written per lane.
__global__ void __launch_bounds__(256) k_quantize_u8(const float* __restrict__ x, uint64_t n, uint8_t* __restrict__ y) {
    const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const uint32_t a = (uint32_t)rintf(fminf(fmaxf(v.x, 0.f), 1.f) * 255.f), b = (uint32_t)rintf(fminf(fmaxf(v.y, 0.f), 1.f) * 255.f);
        const uint32_t c = (uint32_t)rintf(fminf(fmaxf(v.z, 0.f), 1.f) * 255.f), d = (uint32_t)rintf(fminf(fmaxf(v.w, 0.f), 1.f) * 255.f);
        reinterpret_cast<uint32_t*>(y)[i] = a | (b << 8) | (c << 16) | (d << 24);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint64_t i = n4 * 4 + threadIdx.x;
        y[i] = (uint8_t)rintf(fminf(fmaxf(x[i], 0.f), 1.f) * 255.f);
    }
}

extern "C" int ssdnerf_quantize_u8(const float* x, uint64_t n, uint8_t* y, void* stream) {
    if (n == 0) return SSDNERF_OK;
    SSD_REQUIRE(x && y, "quantize_u8: null pointer");
    const uint64_t n4 = n / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 + 1 : 4096);
    hipLaunchKernelGGL(k_quantize_u8, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, y);
    SSD_CHECK_LAUNCH("quantize_u8");
    return SSDNERF_OK;
}
