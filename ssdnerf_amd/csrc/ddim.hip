// ssdnerf_amd/csrc/ddim.hip -- the per-step latent update of the DDIM loop as one elementwise kernel
// (reference: lib/models/diffusions/gaussian_diffusion.py:213 V-parameterisation, :235 clamp, :281-283 eps / x_prev;
// ~10 eager PyTorch kernels and as many HBM round trips over the (S,18,128,128) latent per step in the reference).
// HBM-bound streaming: 8 B read + 8 B written per element, float4 per lane.
#include "common.h"

// (no __restrict__: the sampling loop runs this in place, xprev_out == x_t and, when x0 is not kept, x0_out == v; element i is read before it is written)
__global__ void k_ddim_step_v(const float4* x_t, const float4* v, uint64_t n4, float a, float b, float inv_b, float c,
                              float d, float lo, float hi, float4* x0_out, float4* xprev_out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 xt = x_t[i], vv = v[i];
        float4 x0, xp;
#define SSD_DDIM_LANE(f)                                                                          \
        {                                                                                         \
            float p = a * xt.f - b * vv.f;               /* x0 = sqrt(ab) x_t - sqrt(1-ab) v  */ \
            p = fminf(fmaxf(p, lo), hi);                 /* clip_denoised                     */ \
            const float eps = (xt.f - a * p) * inv_b;    /* (x_t - sqrt(ab) x0) / sqrt(1-ab)  */ \
            x0.f = p;                                                                             \
            xp.f = c * p + d * eps;                      /* sqrt(ab_prev) x0 + sqrt(1-ab_prev) eps */ \
        }
        SSD_DDIM_LANE(x) SSD_DDIM_LANE(y) SSD_DDIM_LANE(z) SSD_DDIM_LANE(w)
#undef SSD_DDIM_LANE
        x0_out[i] = x0;
        xprev_out[i] = xp;
    }
}

extern "C" int ssdnerf_ddim_step_v(const float* x_t, const float* v, uint64_t n, float sqrt_ab, float sqrt_1mab, float sqrt_ab_prev, float dir_coef,
                                   float clip_lo, float clip_hi, float* x0_out, float* xprev_out, void* stream) {
    if (n == 0) return SSDNERF_OK;
    SSD_REQUIRE(x_t && v && x0_out && xprev_out, "ddim_step_v: null pointer");
    SSD_REQUIRE(n % 4 == 0, "ddim_step_v: element count must be a multiple of 4 (latents are (S,18,128,128))");
    const uint64_t n4 = n / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_ddim_step_v, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x_t, (const float4*)v, n4, sqrt_ab, sqrt_1mab,
                       1.0f / sqrt_1mab, sqrt_ab_prev, dir_coef, clip_lo, clip_hi, (float4*)x0_out, (float4*)xprev_out);
    SSD_CHECK_LAUNCH("ddim_step_v");
    return SSDNERF_OK;
}
