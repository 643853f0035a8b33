/* tests/host/decode_bwd_host.c -- TEST INFRASTRUCTURE.  Host build (gcc) of the arithmetic the device kernel k_point_decode_bwd runs
 * (ssdnerf_amd/csrc/decode_bwd_math.h), so that the CPU test suite can check it against PyTorch autograd without a GPU. */
#include "../../ssdnerf_amd/csrc/decode_bwd_math.h"

/* planes (3,Hp,Wp,8) fp32, P packed decoder parameters, xyzs (n,3), shs (n,16) SH values of the view directions (NULL: density only),
 * g_sigmas (n), g_rgbs (n,3) -> grad_planes (3,Hp,Wp,8) accumulated in place; feats (n,18) optional forward features */
void decode_bwd_points(const float* planes, uint32_t Hp, uint32_t Wp, const float* P, const float* xyzs, const float* shs, uint32_t n, float sat,
                       const float* g_sigmas, const float* g_rgbs, float* grad_planes, float* feats) {
    for (uint32_t i = 0; i < n; ++i) {
        float f[18], gf[18];
        const float zero3[3] = {0.f, 0.f, 0.f};
        ssdb_gather18(planes, Hp, Wp, xyzs[3 * i], xyzs[3 * i + 1], xyzs[3 * i + 2], f);
        if (feats) for (int k = 0; k < 18; ++k) feats[18 * (uint64_t)i + k] = f[k];
        ssdb_mlp_backward(P, f, shs ? shs + 16 * (uint64_t)i : f, sat, g_sigmas[i], g_rgbs ? g_rgbs + 3 * (uint64_t)i : zero3, shs != 0, gf);
        ssdb_scatter18(grad_planes, Hp, Wp, xyzs[3 * i], xyzs[3 * i + 1], xyzs[3 * i + 2], gf);
    }
}
