// ssdnerf_amd/csrc/gn_bwd_math.h -- input gradient of  y = act( GroupNorm(x) * (1 + scale) + shift ),  per element.
//
// The forward (groupnorm.hip, k_gn_apply) folds everything into one affine map per (sample, channel):
//     v = x * A + O,   A = rstd * gamma * (1 + scale),   O = (beta - mean * rstd * gamma) * (1 + scale) + shift,   y = silu(v) or v
// (GroupNorm of the residual blocks / NormWithEmbedding / the output head: lib/models/architecture/ddpm/modules.py:51-110,
// denoising.py:178-187).  With frozen gamma / beta and a time embedding that does not depend on x, the only
// gradient needed by guidance and fine-tuning is d/dx:
//     dv = dy * silu'(v)                       silu'(v) = sg(v) (1 + v (1 - sg(v)))
//     p  = dv * k,   k = gamma (1 + scale)     (= d v / d xhat),   xhat = (x - mean) rstd
//     dx = rstd * ( p - mean_g(p) - xhat * mean_g(p * xhat) )      means over the group's channels x pixels
// i.e. one statistics pass (sum p, sum p*xhat per sample and group) and one apply pass, both recomputing v from x: nothing but the
// forward's (sum, sum of squares) is saved.  Plain C: compiled by hipcc into the kernels and by gcc into the CPU test's harness.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define SSDG_FN __device__ __forceinline__
#define SSDG_EXP2(x) __builtin_amdgcn_exp2f(x)
#define SSDG_RCP(x) __builtin_amdgcn_rcpf(x)
#else
#define SSDG_FN static inline
#define SSDG_EXP2(x) exp2f(x)
#define SSDG_RCP(x) (1.0f / (x))
#endif

// per-(sample, channel) coefficients from the group's (mean, rstd) as floats (r06: the kernels compute those once per block and group)
SSDG_FN void ssdg_coeffs_from(float mean, float rstd, float gamma, float beta, int has_ss, float scale, float shift,
                              float* A, float* O, float* k, float* mean_f, float* rstd_f) {
    float a = rstd * gamma;
    float o = fmaf(-mean, a, beta);
    float kk = gamma;
    if (has_ss) {
        const float sc = 1.0f + scale;
        a *= sc;
        o = fmaf(o, sc, shift);
        kk *= sc;
    }
    *A = a; *O = o; *k = kk; *mean_f = mean; *rstd_f = rstd;
}

// per-(sample, channel) coefficients from the forward's fp64 sums; mirrors k_gn_apply's fold (pre_bias = 0)
SSDG_FN void ssdg_coeffs(double sum, double sumsq, double inv_n, float eps, float gamma, float beta, int has_ss, float scale, float shift,
                         float* A, float* O, float* k, float* mean_f, float* rstd_f) {
    const double mean = sum * inv_n;
    double var = sumsq * inv_n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    ssdg_coeffs_from((float)mean, rstd, gamma, beta, has_ss, scale, shift, A, O, k, mean_f, rstd_f);
}

// p = dL/dxhat contribution of one element, and its xhat
SSDG_FN void ssdg_elem(float x, float dy, float A, float O, float k, float mean, float rstd, int act, float* p, float* xhat) {
    float dv = dy;
    if (act) {
        const float v = fmaf(x, A, O);
        const float sg = SSDG_RCP(1.0f + SSDG_EXP2(v * -1.4426950408889634f));
        dv = dy * (sg * fmaf(v, 1.0f - sg, 1.0f));
    }
    *p = dv * k;
    *xhat = (x - mean) * rstd;
}

SSDG_FN float ssdg_dx(float p, float xhat, float rstd, float m1, float m2) { return rstd * (p - m1 - xhat * m2); }
