/* tests/host/gn_bwd_host.c -- TEST INFRASTRUCTURE.  Host build (gcc) of the arithmetic the device kernels k_gn_bwd_stats / k_gn_bwd_apply run
 * (ssdnerf_amd/csrc/gn_bwd_math.h), laid out as the kernels lay it out: channel-last x, dy [B][HW][C], per-(sample, group) sums. */
#include <stdlib.h>
#include "../../ssdnerf_amd/csrc/gn_bwd_math.h"

void gn_bwd(const float* x, const float* dy, uint32_t B, uint32_t HW, uint32_t C, uint32_t G, const float* gamma, const float* beta,
            const float* scale_shift /* [B][2C] or NULL */, float eps, int act, float* dx) {
    const uint32_t cpg = C / G;
    const double inv_n = 1.0 / ((double)HW * cpg);
    double* fs = (double*)calloc((size_t)B * G * 2, sizeof(double));
    double* bs = (double*)calloc((size_t)B * G * 2, sizeof(double));
    float* co = (float*)malloc((size_t)C * 5 * sizeof(float));
    for (uint32_t b = 0; b < B; ++b) {
        for (uint32_t r = 0; r < HW; ++r)
            for (uint32_t c = 0; c < C; ++c) {
                const double v = x[((size_t)b * HW + r) * C + c];
                fs[((size_t)b * G + c / cpg) * 2] += v;
                fs[((size_t)b * G + c / cpg) * 2 + 1] += v * v;
            }
        for (uint32_t c = 0; c < C; ++c)
            ssdg_coeffs(fs[((size_t)b * G + c / cpg) * 2], fs[((size_t)b * G + c / cpg) * 2 + 1], inv_n, eps, gamma[c], beta[c], scale_shift != 0,
                        scale_shift ? scale_shift[(size_t)b * 2 * C + c] : 0.f, scale_shift ? scale_shift[(size_t)b * 2 * C + C + c] : 0.f,
                        co + c * 5, co + c * 5 + 1, co + c * 5 + 2, co + c * 5 + 3, co + c * 5 + 4);
        for (uint32_t r = 0; r < HW; ++r)
            for (uint32_t c = 0; c < C; ++c) {
                float p, xh;
                const size_t i = ((size_t)b * HW + r) * C + c;
                ssdg_elem(x[i], dy[i], co[c * 5], co[c * 5 + 1], co[c * 5 + 2], co[c * 5 + 3], co[c * 5 + 4], act, &p, &xh);
                bs[((size_t)b * G + c / cpg) * 2] += p;
                bs[((size_t)b * G + c / cpg) * 2 + 1] += (double)p * xh;
            }
        for (uint32_t r = 0; r < HW; ++r)
            for (uint32_t c = 0; c < C; ++c) {
                float p, xh;
                const size_t i = ((size_t)b * HW + r) * C + c;
                ssdg_elem(x[i], dy[i], co[c * 5], co[c * 5 + 1], co[c * 5 + 2], co[c * 5 + 3], co[c * 5 + 4], act, &p, &xh);
                dx[i] = ssdg_dx(p, xh, co[c * 5 + 4], (float)(bs[((size_t)b * G + c / cpg) * 2] * inv_n), (float)(bs[((size_t)b * G + c / cpg) * 2 + 1] * inv_n));
            }
    }
    free(fs); free(bs); free(co);
}
