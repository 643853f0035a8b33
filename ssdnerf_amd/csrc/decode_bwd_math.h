// ssdnerf_amd/csrc/decode_bwd_math.h -- gradient of the triplane decode w.r.t. the planes, per sample point.
//
// What the reference gets from autograd through grid_sample + 4 nn.Linear + SiLU/TruncExp/Sigmoid
// (lib/models/decoders/triplane_decoder.py:136-179, lib/ops/activation.py:8-20) when the rendering loss is differentiated
// w.r.t. the scene code with the decoder frozen (guidance: lib/models/autodecoders/diffusion_nerf.py:282-294; fine-tuning:
// lib/models/autodecoders/base_nerf.py:446-470):
//
//   h_i = b1_i + W1_i . f            s_i = silu(h_i)          sa = b_s + sum_i w_s,i s_i          sigma = exp(sa)
//   d_i = bd_i + Wd_i . SH(dir)      c_i = silu(h_i + d_i)    z_j = bc_j + sum_i Wc_ji c_i        rgb_j = sigmoid(z_j) (1 + 2 sat) - sat
//
//   d sigma / d sa = clamp(exp(sa), 1e-6, 1e6)                                (TruncExp.backward)
//   dL/dh_i = dL/dsa w_s,i silu'(h_i) + (sum_j dL/dz_j Wc_ji) silu'(h_i + d_i),    silu'(u) = sg(u) (1 + u (1 - sg(u)))
//   dL/df_k = sum_i dL/dh_i W1_ik ;   dL/dplane[corner][c] += w_corner dL/df_{3c+p}   (bilinear, border padding: grid_sampler_2d_backward)
//
// Nothing is saved by the forward: the 64 hidden units are recomputed twice here (once to reach the outputs, once to push the
// gradient back), ~5.5 kFMA per point, instead of storing 2 x 64 floats per point.
//
// The file is plain C on purpose: the device kernel (decode.hip) and a gcc-built host harness (tests/host/decode_bwd_host.c, which
// the CPU test checks against PyTorch autograd) compile the SAME arithmetic.  Parameter block layout: decode_core.h.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define SSDB_FN __device__ __forceinline__
#define SSDB_EXP2(x) __builtin_amdgcn_exp2f(x)
#define SSDB_RCP(x) __builtin_amdgcn_rcpf(x)
#define SSDB_ATOMIC_ADD(p, v) unsafeAtomicAdd((p), (v))
#define SSDB_UNROLL _Pragma("unroll")
#else
#define SSDB_FN static inline
#define SSDB_EXP2(x) exp2f(x)
#define SSDB_RCP(x) (1.0f / (x))
#define SSDB_ATOMIC_ADD(p, v) (*(p) += (v))
#define SSDB_UNROLL
#endif

#define SSDB_OFF_WD (64 * 24)
#define SSDB_OFF_BD (64 * 24 + 64 * 16)
#define SSDB_OFF_TAIL (64 * 24 + 64 * 16 + 64)

SSDB_FN float ssdb_sigmoid(float u) { return SSDB_RCP(1.0f + SSDB_EXP2(u * -1.4426950408889634f)); }

// unnormalise + clip one coordinate exactly as the forward gather does (decode_core.h ssd_grid_coord), in two halves so that the binned
// reduction of decode.hip can store the clipped coordinate between them
SSDB_FN float ssdb_unnormalise(float u, uint32_t size) {
    const float size_f = (float)size;
    const float ix = ((u + 1.0f) * size_f - 1.0f) * 0.5f;
    return fminf(size_f - 1.0f, fmaxf(ix, 0.0f));
}
SSDB_FN void ssdb_corners(float ix, uint32_t size, uint32_t* i0, uint32_t* i1, float* w0, float* w1) {
    const float fl = floorf(ix);
    *i0 = (uint32_t)fl;
    *i1 = (*i0 + 1u < size) ? *i0 + 1u : size - 1u;
    *w1 = ix - fl;
    *w0 = (fl + 1.0f) - ix;
}
SSDB_FN void ssdb_grid_coord(float u, uint32_t size, uint32_t* i0, uint32_t* i1, float* w0, float* w1) {
    ssdb_corners(ssdb_unnormalise(u, size), size, i0, i1, w0, w1);
}

// f[c*3 + p] from fp32 planes (3, Hp, Wp, 8); the host harness's forward (the device uses ssd_gather18)
SSDB_FN void ssdb_gather18(const float* planes, uint32_t Hp, uint32_t Wp, float x, float y, float z, float f[18]) {
    const float us[3] = {x, x, y}, vs[3] = {y, z, z};
    SSDB_UNROLL
    for (int p = 0; p < 3; ++p) {
        uint32_t x0, x1, y0, y1;
        float wx0, wx1, wy0, wy1;
        ssdb_grid_coord(us[p], Wp, &x0, &x1, &wx0, &wx1);
        ssdb_grid_coord(vs[p], Hp, &y0, &y1, &wy0, &wy1);
        const float* base = planes + (uint64_t)p * Hp * Wp * 8;
        const float* t00 = base + ((uint64_t)y0 * Wp + x0) * 8;
        const float* t01 = base + ((uint64_t)y0 * Wp + x1) * 8;
        const float* t10 = base + ((uint64_t)y1 * Wp + x0) * 8;
        const float* t11 = base + ((uint64_t)y1 * Wp + x1) * 8;
        const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
        SSDB_UNROLL
        for (int c = 0; c < 6; ++c) f[c * 3 + p] = fmaf(t11[c], w11, fmaf(t10[c], w10, fmaf(t01[c], w01, t00[c] * w00)));
    }
}

// dL/df[18] of one point.  ``color`` = 0: density head only (g_rgb, sh ignored).
SSDB_FN void ssdb_mlp_backward(const float* __restrict__ P, const float f[18], const float sh[16], float sat, float g_sigma, const float g_rgb[3],
                               int color, float gf[18]) {
    float sa = P[SSDB_OFF_TAIL + 0];
    float z0 = P[SSDB_OFF_TAIL + 1], z1 = P[SSDB_OFF_TAIL + 2], z2 = P[SSDB_OFF_TAIL + 3];
    for (int i = 0; i < 64; ++i) {
        const float* __restrict__ rec = P + i * 24;
        float h = rec[18];
        SSDB_UNROLL
        for (int k = 0; k < 18; ++k) h = fmaf(rec[k], f[k], h);
        sa = fmaf(rec[19], h * ssdb_sigmoid(h), sa);
        if (color) {
            float d = P[SSDB_OFF_BD + i];
            const float* __restrict__ wd = P + SSDB_OFF_WD + i * 16;
            SSDB_UNROLL
            for (int m = 0; m < 16; ++m) d = fmaf(wd[m], sh[m], d);
            const float u = h + d, c = u * ssdb_sigmoid(u);
            z0 = fmaf(rec[20], c, z0);
            z1 = fmaf(rec[21], c, z1);
            z2 = fmaf(rec[22], c, z2);
        }
    }
    const float e = SSDB_EXP2(sa * 1.4426950408889634f);
    const float dsa = g_sigma * fminf(1e6f, fmaxf(e, 1e-6f));
    float dz0 = 0.0f, dz1 = 0.0f, dz2 = 0.0f;
    if (color) {
        const float k = fmaf(sat, 2.0f, 1.0f);
        const float s0 = ssdb_sigmoid(z0), s1 = ssdb_sigmoid(z1), s2 = ssdb_sigmoid(z2);
        dz0 = g_rgb[0] * k * (s0 * (1.0f - s0));
        dz1 = g_rgb[1] * k * (s1 * (1.0f - s1));
        dz2 = g_rgb[2] * k * (s2 * (1.0f - s2));
    }
    SSDB_UNROLL
    for (int k = 0; k < 18; ++k) gf[k] = 0.0f;
    for (int i = 0; i < 64; ++i) {
        const float* __restrict__ rec = P + i * 24;
        float h = rec[18];
        SSDB_UNROLL
        for (int k = 0; k < 18; ++k) h = fmaf(rec[k], f[k], h);
        const float sg = ssdb_sigmoid(h);
        float dh = dsa * rec[19] * (sg * fmaf(h, 1.0f - sg, 1.0f));
        if (color) {
            float d = P[SSDB_OFF_BD + i];
            const float* __restrict__ wd = P + SSDB_OFF_WD + i * 16;
            SSDB_UNROLL
            for (int m = 0; m < 16; ++m) d = fmaf(wd[m], sh[m], d);
            const float u = h + d, su = ssdb_sigmoid(u);
            const float dc = fmaf(dz2, rec[22], fmaf(dz1, rec[21], dz0 * rec[20]));
            dh = fmaf(dc, su * fmaf(u, 1.0f - su, 1.0f), dh);
        }
        SSDB_UNROLL
        for (int k = 0; k < 18; ++k) gf[k] = fmaf(dh, rec[k], gf[k]);
    }
}

// scatter dL/df into the gradient planes (3, Hp, Wp, 8) fp32 with the forward's corner weights; zero-weight corners are skipped
SSDB_FN void ssdb_scatter18(float* gplanes, uint32_t Hp, uint32_t Wp, float x, float y, float z, const float gf[18]) {
    const float us[3] = {x, x, y}, vs[3] = {y, z, z};
    SSDB_UNROLL
    for (int p = 0; p < 3; ++p) {
        uint32_t x0, x1, y0, y1;
        float wx0, wx1, wy0, wy1;
        ssdb_grid_coord(us[p], Wp, &x0, &x1, &wx0, &wx1);
        ssdb_grid_coord(vs[p], Hp, &y0, &y1, &wy0, &wy1);
        float* base = gplanes + (uint64_t)p * Hp * Wp * 8;
        const uint64_t o[4] = {((uint64_t)y0 * Wp + x0) * 8, ((uint64_t)y0 * Wp + x1) * 8, ((uint64_t)y1 * Wp + x0) * 8, ((uint64_t)y1 * Wp + x1) * 8};
        const float w[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
        SSDB_UNROLL
        for (int q = 0; q < 4; ++q) {
            if (w[q] == 0.0f) continue;
            SSDB_UNROLL
            for (int c = 0; c < 6; ++c) SSDB_ATOMIC_ADD(base + o[q] + c, gf[c * 3 + p] * w[q]);
        }
    }
}
