// ssdnerf_amd/csrc/decode_core.h -- triplane bilinear gather + tiny MLP, device-side building blocks.
//
// What the reference does with ~15 eager PyTorch kernels per decode call
// (lib/models/decoders/triplane_decoder.py:136-179: grid_sample x3 planes, permute, 4 nn.Linear, SiLU x2,
// TruncExp, Sigmoid, saturation affine) happens here in registers for one sample per lane:
//
//   planes  : (3, Hp, Wp, 8) channel-last, 6 real channels zero-padded to 8 -> one bilinear corner is two
//             16-byte loads (fp32) or one (fp16); the 4 corners of a plane sit in two 64-byte row segments.
//             A scene's planes are 1.5 MiB (fp32) and stay L2-resident; the ALGORITHMIC traffic the roofline
//             is quoted on is 288 B / sample (3 planes x 4 corners x 6 ch x 4 B).
//   weights : the packed parameter block (SSDNERF_MLP_PARAM_FLOATS floats, layout below) is wave-uniform, so
//             every weight is read with scalar loads and enters the FMAs as an SGPR operand: no VGPRs, no LDS.
//   gather arithmetic follows ATen's grid_sample(bilinear, padding_mode='border', align_corners=False):
//             ix = ((u + 1) * W - 1) / 2, clipped to [0, W-1]; corner weights (x1 - ix)(y1 - iy) etc.
//
// Parameter block layout (floats):
//   [0, 64*24)            rec[i] = { W1[i][0..17], b1[i], w_sigma[i], Wc[0][i], Wc[1][i], Wc[2][i], 0 }
//   [1536, 1536+64*16)    Wd[i][0..15]          (dir_net weight, row i = hidden unit i)
//   [2560, 2624)          bd[i]
//   [2624, 2628)          b_sigma, bc[0], bc[1], bc[2]
#pragma once
#include "common.h"
#include "sh_basis.h"


#define MLP_OFF_WD (64 * 24)
#define MLP_OFF_BD (64 * 24 + 64 * 16)
#define MLP_OFF_TAIL (64 * 24 + 64 * 16 + 64)

template <typename PT> struct Texel;
template <> struct Texel<float> {
    static SSD_DEV void load6(const float* __restrict__ p, float v[6]) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float2 b = *reinterpret_cast<const float2*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y;
    }
};
template <> struct Texel<__half> {
    static SSD_DEV void load6(const __half* __restrict__ p, float v[6]) {
        const uint4 raw = *reinterpret_cast<const uint4*>(p);  // 8 halfs = 16 B
        const __half2* h = reinterpret_cast<const __half2*>(&raw);
        const float2 a = __half22float2(h[0]), b = __half22float2(h[1]), c = __half22float2(h[2]);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
    }
};

struct PlaneGeom { uint32_t Hp, Wp; float Hf, Wf; };
static inline PlaneGeom ssd_plane_geom(uint32_t Hp, uint32_t Wp) { PlaneGeom g; g.Hp = Hp; g.Wp = Wp; g.Hf = (float)Hp; g.Wf = (float)Wp; return g; }

// Unnormalise + clip one coordinate, split into integer cell and fraction.
SSD_DEV void ssd_grid_coord(float u, float size_f, uint32_t size, uint32_t& i0, uint32_t& i1, float& w0, float& w1) {
    float ix = ((u + 1.0f) * size_f - 1.0f) * 0.5f;
    ix = fminf(size_f - 1.0f, fmaxf(ix, 0.0f));
    const float fl = floorf(ix);
    i0 = (uint32_t)fl;
    i1 = min(i0 + 1u, size - 1u);      // the out-of-range neighbour only ever carries weight 0
    w1 = ix - fl;                      // exact
    w0 = (fl + 1.0f) - ix;             // ATen's (ix_se - ix): the correctly rounded 1 - w1
}

// f[c*3 + p] = bilinear sample of channel c of plane p; planes (xy, xz, yz), first coordinate on the width axis.
// PLANE_BY_PLANE: a scheduling barrier after each plane keeps at most one plane's texels (4 x 8 registers) in flight instead of all
// three (96 registers) -- for kernels that trade memory-level parallelism per wave for a third wave per SIMD.
template <typename PT, bool PLANE_BY_PLANE = false>
SSD_DEV void ssd_gather18(const PT* __restrict__ planes, const PlaneGeom& g, float x, float y, float z, float f[18]) {
    const float us[3] = {x, x, y};
    const float vs[3] = {y, z, z};
    const uint64_t plane_stride = (uint64_t)g.Hp * g.Wp * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        uint32_t x0, x1, y0, y1;
        float wx0, wx1, wy0, wy1;
        ssd_grid_coord(us[p], g.Wf, g.Wp, x0, x1, wx0, wx1);
        ssd_grid_coord(vs[p], g.Hf, g.Hp, y0, y1, wy0, wy1);
        const PT* base = planes + p * plane_stride;
        float t00[6], t01[6], t10[6], t11[6];
        Texel<PT>::load6(base + ((uint64_t)y0 * g.Wp + x0) * 8, t00);
        Texel<PT>::load6(base + ((uint64_t)y0 * g.Wp + x1) * 8, t01);
        Texel<PT>::load6(base + ((uint64_t)y1 * g.Wp + x0) * 8, t10);
        Texel<PT>::load6(base + ((uint64_t)y1 * g.Wp + x1) * 8, t11);
        const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
        // the four-term chain  t11 w11 + (t10 w10 + (t01 w01 + t00 w00))  per channel, written on channel PAIRS of one texel (adjacent registers of the
        // load, one shared weight) so that it maps onto v_pk_mul / v_pk_fma without operand assembly.  The scalar form is auto-vectorised ACROSS texels,
        // which costs 56 register moves per sample to assemble the pairs: -2 % on the shading kernel since it became issue-bound (r03,
        // profiles/r03/h_shade_valu_diet.txt; r02 had measured the two forms equal); element-wise identical arithmetic, bit-identical results
        typedef float ssd_f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
            const ssd_f2 a00 = {t00[c], t00[c + 1]}, a01 = {t01[c], t01[c + 1]}, a10 = {t10[c], t10[c + 1]}, a11 = {t11[c], t11[c + 1]};
            ssd_f2 r = a00 * ssd_f2{w00, w00};
            r = __builtin_elementwise_fma(a01, ssd_f2{w01, w01}, r);
            r = __builtin_elementwise_fma(a10, ssd_f2{w10, w10}, r);
            r = __builtin_elementwise_fma(a11, ssd_f2{w11, w11}, r);
            f[c * 3 + p] = r.x;
            f[(c + 1) * 3 + p] = r.y;
        }
        if (PLANE_BY_PLANE) __builtin_amdgcn_sched_barrier(0);
    }
}

// silu(h) = h / (1 + e^-h) with the hardware exp2 / rcp units (each ~1 ulp; DESIGN.md "float tolerance").
SSD_DEV float ssd_silu(float h) {
    const float e = __builtin_amdgcn_exp2f(h * -1.4426950408889634f);
    return h * __builtin_amdgcn_rcpf(1.0f + e);
}
SSD_DEV float ssd_sigmoid(float h) {
    const float e = __builtin_amdgcn_exp2f(h * -1.4426950408889634f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
SSD_DEV float ssd_exp(float h) { return __builtin_amdgcn_exp2f(h * 1.4426950408889634f); }

// HD_MODE 0: density only.  1: direction term from 16 SH values held in registers (16 FMAs per hidden unit).
// 2: direction term precomputed per ray, read from `hd` (LDS or global row of 64 floats).
template <int HD_MODE>
SSD_DEV void ssd_mlp(const float* __restrict__ P, const float f[18], const float* sh, const float* hd, float sat, float& sigma, float& cr,
                     float& cg, float& cb) {
    float sa = P[MLP_OFF_TAIL + 0];
    float r = P[MLP_OFF_TAIL + 1], g = P[MLP_OFF_TAIL + 2], b = P[MLP_OFF_TAIL + 3];
#pragma unroll 4
    for (int i = 0; i < 64; ++i) {
        const float* __restrict__ rec = P + i * 24;
        float h = rec[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) h = ssd_fma(rec[k], f[k], h);
        sa = ssd_fma(rec[19], ssd_silu(h), sa);
        if (HD_MODE != 0) {
            float d;
            if (HD_MODE == 1) {
                d = P[MLP_OFF_BD + i];
                const float* __restrict__ wd = P + MLP_OFF_WD + i * 16;
#pragma unroll
                for (int m = 0; m < 16; ++m) d = ssd_fma(wd[m], sh[m], d);
            } else {
                d = hd[i];
            }
            const float c = ssd_silu(h + d);
            r = ssd_fma(rec[20], c, r);
            g = ssd_fma(rec[21], c, g);
            b = ssd_fma(rec[22], c, b);
        }
    }
    sigma = ssd_exp(sa);  // TruncExp forward == exp (lib/ops/activation.py:8-13)
    if (HD_MODE != 0) {
        const float k = ssd_fma(sat, 2.0f, 1.0f);
        cr = ssd_fma(ssd_sigmoid(r), k, -sat);
        cg = ssd_fma(ssd_sigmoid(g), k, -sat);
        cb = ssd_fma(ssd_sigmoid(b), k, -sat);
    }
}
