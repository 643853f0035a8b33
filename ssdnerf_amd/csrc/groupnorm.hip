// ssdnerf_amd/csrc/groupnorm.hip -- GroupNorm (+ optional per-sample scale/shift from the time embedding, + optional SiLU)
// over channel-last activations, the memory-bound glue of the denoising UNet's residual and attention blocks.
//
// Reference: every DenoisingResBlockMod is  GN -> SiLU -> conv3x3 -> GN*(1+scale)+shift -> SiLU -> conv3x3  (+ shortcut)
// (lib/models/architecture/ddpm/modules.py:51-110 builds the block, mmgen 0.7.2 supplies the forward),
// every MultiHeadAttentionMod starts with a plain GN (modules.py:12-48) and the output head is GN -> SiLU -> conv
// (lib/models/architecture/ddpm/denoising.py:178-187).  In eager PyTorch each of these is 4-7 kernels (moments, fused-params,
// normalise, dtype casts, mul/add for scale-shift, SiLU) plus MIOpen's NCHW<->NHWC transposes around the neighbouring
// convolution; at bf16 they cost more time than the convolutions themselves (tools/bench_unet.py).
//
// Here: activations stay NHWC ([B][H*W][C], the layout the MFMA convolution kernels consume), and the whole chain is
//   k_gn_stats : one pass, per-(sample, group) sum and sum of squares (fp32 in-block, fp64 atomics across blocks)
//   k_gn_apply : one pass, y = act(x * A[b][c] + B[b][c]) with A, B folded per block from (mean, rstd, gamma, beta, scale, shift)
// = 2 reads + 1 write of the activation, 16-byte vector accesses, HBM-bound.  Statistics and the affine fold are fp32/fp64
// whatever the storage type, so the bf16/fp16 paths are *more* accurate than the eager chain they replace.
#include "common.h"
#include "gn_bwd_math.h"

#include <hip/hip_fp16.h>

namespace {

enum { GN_F32 = 0, GN_F16 = 1, GN_BF16 = 2 };

template <int DT> struct GnVec;                       // one 16-byte access
template <> struct GnVec<GN_F32> {
    static constexpr int V = 4;
    __device__ static void load(const void* p, size_t idx, float* f) {
        const float4 v = reinterpret_cast<const float4*>(p)[idx];
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    __device__ static uint4 raw(const void* p, size_t idx) { return reinterpret_cast<const uint4*>(p)[idx]; }       // the 16 bytes as they are (r06: loads issued ahead of their use)
    __device__ static void cvt(const uint4& v, float* f) { f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w); }
    __device__ static void store(void* p, size_t idx, const float* f) { reinterpret_cast<float4*>(p)[idx] = make_float4(f[0], f[1], f[2], f[3]); }
    __device__ static float round(float x) { return x; }               // value as it reads back from storage
    // PRE-SPLIT store (r04): the four fp32 values leave as their bf16 pair split -- hi = truncation to bf16, lo = truncation of the exact remainder,
    // the arithmetic of the fp32-class convolution kernels (conv_igemm.hip) -- in the layout the two-group kernel's K-tiles take: per pixel and block
    // of 32 channels, 128 bytes = [32 hi terms | 32 lo terms].  Same bytes per element as fp32; `idx` is the 4-channel vector index as in store().
    __device__ static void store_split(void* p, size_t idx, const float* f) {
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hi[i] = __float_as_uint(f[i]) & 0xffff0000u;
            lo[i] = __float_as_uint(f[i] - __uint_as_float(hi[i]));
        }
        unsigned char* base = reinterpret_cast<unsigned char*>(p) + (idx >> 3) * 128 + (idx & 7) * 8;      // 8 vectors of 4 channels per 32-channel block
        *reinterpret_cast<uint2*>(base) = make_uint2(__builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u), __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u));
        *reinterpret_cast<uint2*>(base + 64) = make_uint2(__builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u), __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u));
    }
};
template <> struct GnVec<GN_BF16> {
    static constexpr int V = 8;
    __device__ static void load(const void* p, size_t idx, float* f) {
        const uint4 v = reinterpret_cast<const uint4*>(p)[idx];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static uint4 raw(const void* p, size_t idx) { return reinterpret_cast<const uint4*>(p)[idx]; }
    __device__ static void cvt(const uint4& v, float* f) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    // fp32 -> bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32, one instruction per PAIR (r03; the integer form -- add, shift, and, add,
    // shift per element plus the pack -- was a third of k_gn_apply's VALU work, which at bf16 is what bounds that kernel, not HBM)
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    __device__ static uint32_t pack2(float lo, float hi) {
        const b2 r = __builtin_convertvector(f2{lo, hi}, b2);
        return *reinterpret_cast<const uint32_t*>(&r);
    }
    __device__ static void store(void* p, size_t idx, const float* f) {
        reinterpret_cast<uint4*>(p)[idx] = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    __device__ static float round(float x) { return __uint_as_float(pack2(x, 0.f) << 16); }
};
template <> struct GnVec<GN_F16> {
    static constexpr int V = 8;
    __device__ static void load(const void* p, size_t idx, float* f) {
        union { uint4 u; _Float16 h[8]; } v;
        v.u = reinterpret_cast<const uint4*>(p)[idx];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v.h[i];
    }
    __device__ static uint4 raw(const void* p, size_t idx) { return reinterpret_cast<const uint4*>(p)[idx]; }
    __device__ static void cvt(const uint4& r, float* f) {
        union { uint4 u; _Float16 h[8]; } v;
        v.u = r;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v.h[i];
    }
    __device__ static void store(void* p, size_t idx, const float* f) {
        union { uint4 u; _Float16 h[8]; } v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v.h[i] = (_Float16)f[i];
        reinterpret_cast<uint4*>(p)[idx] = v.u;
    }
    __device__ static float round(float x) { return (float)(_Float16)x; }
};

constexpr int GN_TPB = 256;
constexpr int GN_MAX_C = 2048;

// grid (HW / rows_per_block, B).  Threads are laid out [row-in-flight][channel vector]; a thread keeps V running sums for
// its fixed channel vector, the block folds them over the rows-in-flight through LDS (conflict-free: consecutive threads,
// consecutive channels), then per group, then adds to the global fp64 sums.  pre_bias (nullable, fp32 [C]) is the bias of the
// convolution that produced x, added on load so that the producer does not need a pass of its own for it.
// x may be the channel concatenation [x | x2] of two tensors (C1 channels from x, C - C1 from x2) that is never materialised: the
// skip connections of the UNet's decoder half (denoising.py:209-213 `torch.cat([h, hs.pop()], dim=1)`).
template <int DT>
__global__ __launch_bounds__(GN_TPB) void k_gn_stats(const void* __restrict__ x, const void* __restrict__ x2, uint32_t C1, const float* __restrict__ pre_bias,
                                                      uint32_t HW, uint32_t C, uint32_t G, uint32_t rows_per_block, double* __restrict__ sums) {
    constexpr int V = GnVec<DT>::V;
    __shared__ float part_s[GN_TPB * V], part_q[GN_TPB * V];              // [row-in-flight][C]  (rif * C <= 256 * V)
    const uint32_t tpr = C / V, rif = GN_TPB / tpr;
    const uint32_t lane_row = threadIdx.x / tpr, cv = threadIdx.x % tpr;
    const uint32_t b = blockIdx.y, row0 = blockIdx.x * rows_per_block;
    const uint32_t tpr1 = C1 / V;
    const bool second = cv >= tpr1;
    const void* src = second ? x2 : x;
    const uint32_t stpr = second ? tpr - tpr1 : tpr1, scv = second ? cv - tpr1 : cv;
    if (lane_row < rif) {
        float s[V], q[V], pb[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { s[i] = 0.f; q[i] = 0.f; pb[i] = pre_bias ? pre_bias[cv * V + i] : 0.f; }
        const size_t base = ((size_t)b * HW + row0) * stpr + scv;
        for (uint32_t r = lane_row; r < rows_per_block; r += rif) {
            float f[V];
            GnVec<DT>::load(src, base + (size_t)r * stpr, f);
#pragma unroll
            for (int i = 0; i < V; ++i) { const float v = f[i] + pb[i]; s[i] += v; q[i] = __builtin_fmaf(v, v, q[i]); }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) { part_s[lane_row * C + cv * V + i] = s[i]; part_q[lane_row * C + cv * V + i] = q[i]; }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < C; c += GN_TPB) {                   // fold the rows-in-flight, result in row 0
        float s = 0.f, q = 0.f;
        for (uint32_t r = 0; r < rif; ++r) { s += part_s[r * C + c]; q += part_q[r * C + c]; }
        part_s[c] = s; part_q[c] = q;                                      // row 0 slot c is only read by this thread above
    }
    __syncthreads();
    const uint32_t cpg = C / G;
    for (uint32_t g = threadIdx.x; g < G; g += GN_TPB) {
        double ds = 0.0, dq = 0.0;
        for (uint32_t i = 0; i < cpg; ++i) { ds += (double)part_s[g * cpg + i]; dq += (double)part_q[g * cpg + i]; }
        atomicAdd(&sums[((size_t)b * G + g) * 2 + 0], ds);
        atomicAdd(&sums[((size_t)b * G + g) * 2 + 1], dq);
    }
}

// RUNS (r03): the statistics arrive per (sample, RUN of 4 consecutive channels) instead of per group -- `sums` [B][C1 / 4][2] for x, `sums2`
// [B][(C - C1) / 4][2] for x2 -- and a group's sums are the sums of its runs.  Every producer (convolution epilogue, split-K finishing pass) writes
// them without knowing how its consumer groups the channels, so the concatenated skip connections of the decoder half -- groups of 12 or 24
// channels that straddle the two tensors -- no longer need a statistics pass of their own (k_gn_stats: 20 launches, 0.28 / 0.45 ms per forward).
template <int DT, bool RUNS = false>
__global__ __launch_bounds__(GN_TPB) void k_gn_apply(const void* __restrict__ x, const void* __restrict__ x2, uint32_t C1, const float* __restrict__ pre_bias,
                                                      uint32_t HW, uint32_t C, uint32_t G, uint32_t rows_per_block, const double* __restrict__ sums, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ scale_shift, uint32_t ss_stride, float eps,
                                                      int act, void* __restrict__ y, const double* __restrict__ sums2 = nullptr) {
    constexpr int V = GnVec<DT>::V;
    __shared__ float fa[GN_MAX_C], fb[GN_MAX_C];
    const uint32_t tpr = C / V, rif = GN_TPB / tpr;
    const uint32_t lane_row = threadIdx.x / tpr, cv = threadIdx.x % tpr;
    const uint32_t b = blockIdx.y, row0 = blockIdx.x * rows_per_block;
    const uint32_t cpg = C / G;
    const double inv_n = 1.0 / ((double)HW * (double)cpg);
    __shared__ double grp[RUNS ? 2 * GN_TPB : 2];                           // RUNS: one thread per group adds up the group's runs (all loads in flight at once)
    if (RUNS) {
        const uint32_t R1 = C1 / 4, R2 = C / 4 - R1, rpg = cpg / 4;
        for (uint32_t g = threadIdx.x; g < G; g += GN_TPB) {
            double gs = 0.0, gq = 0.0;
            for (uint32_t r0 = g * rpg; r0 < (g + 1) * rpg; r0 += 8) {
                double2 v[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint32_t r = r0 + k;
                    const double* p = r < R1 ? sums + ((size_t)b * R1 + r) * 2 : sums2 + ((size_t)b * R2 + (r - R1)) * 2;
                    v[k] = r < (g + 1) * rpg ? *reinterpret_cast<const double2*>(p) : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) { gs += v[k].x; gq += v[k].y; }
            }
            grp[2 * g] = gs; grp[2 * g + 1] = gq;
        }
        __syncthreads();
    }
    for (uint32_t c = threadIdx.x; c < C; c += GN_TPB) {
        const uint32_t g = c / cpg;
        const double gs = RUNS ? grp[2 * g] : sums[((size_t)b * G + g) * 2 + 0], gq = RUNS ? grp[2 * g + 1] : sums[((size_t)b * G + g) * 2 + 1];
        const double mean = gs * inv_n;
        double var = gq * inv_n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        float a = rstd * gamma[c];
        float o = __builtin_fmaf((pre_bias ? pre_bias[c] : 0.f) - (float)mean, a, beta[c]);      // ((x + pb) - mean) * a + beta
        if (scale_shift) {                             // NormWithEmbedding, use_scale_shift: norm(x) * (1 + scale) + shift
            const float sc = 1.0f + scale_shift[(size_t)b * ss_stride + c], sh = scale_shift[(size_t)b * ss_stride + C + c];
            a *= sc;
            o = __builtin_fmaf(o, sc, sh);
        }
        fa[c] = a; fb[c] = o;
    }
    __syncthreads();
    if (lane_row >= rif) return;
    float a[V], o[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { a[i] = fa[cv * V + i]; o[i] = fb[cv * V + i]; }
    const uint32_t tpr1 = C1 / V;
    const bool second = cv >= tpr1;
    const void* src = second ? x2 : x;
    const uint32_t stpr = second ? tpr - tpr1 : tpr1, scv = second ? cv - tpr1 : cv;
    const size_t sbase = ((size_t)b * HW + row0) * stpr + scv;
    const size_t base = ((size_t)b * HW + row0) * tpr + cv;
    // GN_APPLY_UNROLL rows per trip, their 16-byte loads issued before the first is used (r03 A/B, cars UNet forward: 1: 4.32 / 9.13 ms, 2: 4.29 / 9.05,
    // 4: 4.33 / 9.04 -- bf16 / fp32; one load in flight per thread left the pass at ~55 % of
    // the HBM rate -- 8 waves per SIMD x 16 bytes is 32 KiB in flight per CU against ~2 us of latency)
#ifndef GN_APPLY_UNROLL
#define GN_APPLY_UNROLL 2
#endif
    constexpr int UR = GN_APPLY_UNROLL;
    for (uint32_t r0 = lane_row; r0 < rows_per_block; r0 += rif * UR) {
        float f[UR][V];
#pragma unroll
        for (int k = 0; k < UR; ++k)
            if (r0 + k * rif < rows_per_block) GnVec<DT>::load(src, sbase + (size_t)(r0 + k * rif) * stpr, f[k]);
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            if (r0 + k * rif >= rows_per_block) break;
#pragma unroll
            for (int i = 0; i < V; i += 2) {                                  // pairs: packed fp32 math around the two transcendentals (SiLU = v / (1 + 2^(-log2(e) v)))
                f2 v = __builtin_elementwise_fma(f2{f[k][i], f[k][i + 1]}, f2{a[i], a[i + 1]}, f2{o[i], o[i + 1]});
                if (act & 1) {
                    const f2 u = v * f2{-1.4426950408889634f, -1.4426950408889634f};
                    const f2 d = f2{__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)} + f2{1.0f, 1.0f};
                    v = v * f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                }
                f[k][i] = v.x; f[k][i + 1] = v.y;
            }
            if constexpr (DT == GN_F32) {
                if (act & 2) { GnVec<DT>::store_split(y, base + (size_t)(r0 + k * rif) * tpr, f[k]); continue; }     // (host: C % 32 == 0)
            }
            GnVec<DT>::store(y, base + (size_t)(r0 + k * rif) * tpr, f[k]);
        }
    }
}

// y = x + bias[c] + residual  (either addend optional): the epilogue of a bias-less convolution -- conv_2 of a residual block
// plus its skip (modules.py:51-110) -- or the `h + x` that closes an attention block; optionally accumulates the GroupNorm sums
// of y (per sample and group) for the norm that reads y next.  grid (row slabs, B); a thread keeps one channel vector.
template <int DT>
__global__ __launch_bounds__(GN_TPB) void k_bias_residual(const void* __restrict__ x, const float* __restrict__ bias, const void* __restrict__ residual,
                                                           uint32_t HW, uint32_t C, uint32_t rows_per_block, void* __restrict__ y, double* __restrict__ sums,
                                                           uint32_t G) {
    constexpr int V = GnVec<DT>::V;
    __shared__ float part_s[GN_TPB * V], part_q[GN_TPB * V];
    const uint32_t tpr = C / V, rif = GN_TPB / tpr;
    const uint32_t lane_row = threadIdx.x / tpr, cv = threadIdx.x % tpr;
    const uint32_t b = blockIdx.y, row0 = blockIdx.x * rows_per_block;
    float s[V], q[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s[i] = 0.f; q[i] = 0.f; }
    if (lane_row < rif) {
        float bv[V];
#pragma unroll
        for (int i = 0; i < V; ++i) bv[i] = bias ? bias[cv * V + i] : 0.f;
        const size_t base = ((size_t)b * HW + row0) * tpr + cv;
        for (uint32_t r = lane_row; r < rows_per_block; r += rif) {
            float f[V], rr[V];
            GnVec<DT>::load(x, base + (size_t)r * tpr, f);
#pragma unroll
            for (int i = 0; i < V; ++i) f[i] += bv[i];
            if (residual) {
                GnVec<DT>::load(residual, base + (size_t)r * tpr, rr);
#pragma unroll
                for (int i = 0; i < V; ++i) f[i] += rr[i];
            }
            GnVec<DT>::store(y, base + (size_t)r * tpr, f);
            if (sums) {
#pragma unroll
                for (int i = 0; i < V; ++i) { const float v = GnVec<DT>::round(f[i]); s[i] += v; q[i] = __builtin_fmaf(v, v, q[i]); }   // what the next norm will read
            }
        }
    }
    if (!sums) return;
    if (lane_row < rif) {
#pragma unroll
        for (int i = 0; i < V; ++i) { part_s[lane_row * C + cv * V + i] = s[i]; part_q[lane_row * C + cv * V + i] = q[i]; }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < C; c += GN_TPB) {
        float ss = 0.f, qq = 0.f;
        for (uint32_t r = 0; r < rif; ++r) { ss += part_s[r * C + c]; qq += part_q[r * C + c]; }
        part_s[c] = ss; part_q[c] = qq;
    }
    __syncthreads();
    const uint32_t cpg = C / G;
    for (uint32_t g = threadIdx.x; g < G; g += GN_TPB) {
        double ds = 0.0, dq = 0.0;
        for (uint32_t i = 0; i < cpg; ++i) { ds += (double)part_s[g * cpg + i]; dq += (double)part_q[g * cpg + i]; }
        atomicAdd(&sums[((size_t)b * G + g) * 2 + 0], ds);
        atomicAdd(&sums[((size_t)b * G + g) * 2 + 1], dq);
    }
}

// ---- input gradient (frozen gamma / beta, time embedding independent of x): the arithmetic of gn_bwd_math.h over the same
// [row-in-flight][channel vector] thread layout as the forward.  A thread owns one channel vector for the whole slab, so its
// V x {A, O, k, mean, rstd} coefficients live in registers (no LDS table).
//   k_gn_bwd_stats : per (sample, group)  sum p,  sum p * xhat   (p = dy * silu'(v) * gamma (1 + scale)), fp64 atomics across blocks
//   k_gn_bwd_apply : dx = rstd * (p - mean(p) - xhat * mean(p * xhat))
// = 4 reads + 1 write of the activation for what eager autograd does with ~10 kernels (native_group_norm_backward, silu_backward,
// the scale/shift mul/add pair, and an NHWC <-> NCHW copy on either side).
// r06: a group's mean / rstd (an fp64 square root and division) are computed once per block, by one thread per group, into LDS; until then every thread
// derived them for each of its V channels.  Same expressions in the same precisions (ssdg_coeffs): bit-identical results.
__device__ __forceinline__ void gn_bwd_group_table(uint32_t b, uint32_t HW, uint32_t C, uint32_t G, const double* __restrict__ fsums, float eps, int act,
                                                   float* __restrict__ g_mean, float* __restrict__ g_rstd, const double* __restrict__ fsums2 = nullptr, uint32_t C1 = 0) {
    const uint32_t cpg = C / G;
    const double inv_n = 1.0 / ((double)HW * (double)cpg);
    for (uint32_t g = threadIdx.x; g < G; g += GN_TPB) {
        double gs, gq;
        if (act & 4) {                                                        // (r06) fsums per RUN of 4 channels, [B][C / 4][2], as the convolutions' epilogues leave them
            gs = 0.0; gq = 0.0;
            const uint32_t rpg = cpg / 4, R1 = fsums2 ? C1 / 4 : C / 4, R2 = C / 4 - R1;          // (two sources: the runs of x, then the runs of x2, as k_gn_apply<RUNS>)
            for (uint32_t r = g * rpg; r < (g + 1) * rpg; ++r) {
                const double2 v = *reinterpret_cast<const double2*>(r < R1 ? fsums + ((size_t)b * R1 + r) * 2 : fsums2 + ((size_t)b * R2 + (r - R1)) * 2);
                gs += v.x; gq += v.y;
            }
        } else {
            gs = fsums[((size_t)b * G + g) * 2 + 0]; gq = fsums[((size_t)b * G + g) * 2 + 1];
        }
        const double mean = gs * inv_n;
        double var = gq * inv_n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        g_mean[g] = (float)mean;
        g_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

#ifndef GN_BWD_UNROLL
#define GN_BWD_UNROLL 4
#endif
#ifndef GN_BWD_REPLICAS
#define GN_BWD_REPLICAS 8
#endif
// ... and a channel's coefficients (A, O, k) once per block by ONE thread per channel into LDS -- consecutive threads, consecutive channels: coalesced loads of
// gamma / beta / scale / shift.  Each thread used to load them for its own V channels: 4 V scalar loads whose 64 lanes stride by V floats (bf16: 32 loads of 16
// cache lines each), which made the bf16 passes of the mid-size tensors twice as slow as the fp32 ones (tools/bench_gn_bwd.py: 32 x 32 x 256: 18.5 against 9 us).
// tab: [3][C] (A, O, k); g_mean / g_rstd: the group table.  Callers synchronise behind it.
__device__ __forceinline__ void gn_bwd_channel_table(uint32_t b, uint32_t C, uint32_t G, const float* __restrict__ g_mean, const float* __restrict__ g_rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale_shift,
                                                     uint32_t ss_stride, float* __restrict__ tab) {
    const uint32_t cpg = C / G;
    for (uint32_t c = threadIdx.x; c < C; c += GN_TPB) {
        const uint32_t g = c / cpg;
        float A, O, K, M, R;
        ssdg_coeffs_from(g_mean[g], g_rstd[g], gamma[c], beta[c], scale_shift != nullptr,
                         scale_shift ? scale_shift[(size_t)b * ss_stride + c] : 0.f, scale_shift ? scale_shift[(size_t)b * ss_stride + C + c] : 0.f, &A, &O, &K, &M, &R);
        tab[c] = A; tab[C + c] = O; tab[2 * C + c] = K;
    }
}
template <int V>
__device__ __forceinline__ void gn_bwd_coeffs(uint32_t cv, uint32_t C, uint32_t G, const float* __restrict__ g_mean, const float* __restrict__ g_rstd,
                                              const float* __restrict__ tab, float* A, float* O, float* K, float* M, float* R) {
    const uint32_t cpg = C / G;
#pragma unroll
    for (int i = 0; i < V; i += 4) {                                           // (V is 4 or 8; cv * V + i is a multiple of 4: 16-byte LDS reads)
        const float4 a = *reinterpret_cast<const float4*>(tab + cv * V + i), o = *reinterpret_cast<const float4*>(tab + C + cv * V + i),
                     k = *reinterpret_cast<const float4*>(tab + 2 * C + cv * V + i);
        A[i] = a.x; A[i + 1] = a.y; A[i + 2] = a.z; A[i + 3] = a.w;
        O[i] = o.x; O[i + 1] = o.y; O[i + 2] = o.z; O[i + 3] = o.w;
        K[i] = k.x; K[i + 1] = k.y; K[i + 2] = k.z; K[i + 3] = k.w;
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const uint32_t g = (cv * V + i) / cpg;
        M[i] = g_mean[g]; R[i] = g_rstd[g];
    }
}

template <int DT>
__global__ __launch_bounds__(GN_TPB) void k_gn_bwd_stats(const void* __restrict__ x, const void* __restrict__ dy, uint32_t HW, uint32_t C, uint32_t G,
                                                          uint32_t rows_per_block, const double* __restrict__ fsums, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ scale_shift, uint32_t ss_stride, float eps,
                                                          int act, double* __restrict__ bsums, const void* __restrict__ x2, uint32_t C1, const double* __restrict__ fsums2) {
    constexpr int V = GnVec<DT>::V;
    __shared__ float part_s[GN_TPB * V], part_q[GN_TPB * V];
    const uint32_t tpr = C / V, rif = GN_TPB / tpr;
    const uint32_t lane_row = threadIdx.x / tpr, cv = threadIdx.x % tpr;
    const uint32_t b = blockIdx.y, row0 = blockIdx.x * rows_per_block;
    // r06: a block's life was a chain of exposed latencies -- coefficients (global loads), then one 16-byte load pair per row trip, each trip waiting for
    // its own -- over 32 rows, and the passes ran at 1.3 - 1.7 TB/s (bf16: SLOWER than fp32 on the same tensor, tools/bench_gn_bwd.py).  Now a trip is
    // GN_BWD_UNROLL rows whose loads are all issued before the first is used, the first trip's loads go out ahead of the coefficient prologue, and the host
    // hands a block enough rows for several trips where the tensor has them and one trip where it is small (gn_bwd_rows).
    constexpr int UR = GN_BWD_UNROLL;
    const size_t base = ((size_t)b * HW + row0) * tpr + cv;
    // (r06) x may be the never-materialised concatenation [x | x2] (C1 channels from x): a thread's channel vector lies in one of the two
    const uint32_t tpr1 = (x2 ? C1 : C) / V;
    const bool second = cv >= tpr1;
    const void* const xs = second ? x2 : x;
    const uint32_t stpr = second ? tpr - tpr1 : tpr1;
    const size_t sbase = ((size_t)b * HW + row0) * stpr + (second ? cv - tpr1 : cv);
    uint4 rx[UR], rd[UR];
    auto issue = [&](uint32_t r0) {
#pragma unroll
        for (int k = 0; k < UR; ++k)
            if (r0 + k * rif < rows_per_block) {
                rx[k] = GnVec<DT>::raw(xs, sbase + (size_t)(r0 + k * rif) * stpr);
                rd[k] = GnVec<DT>::raw(dy, base + (size_t)(r0 + k * rif) * tpr);
            }
    };
    if (lane_row < rif) issue(lane_row);
    extern __shared__ __attribute__((aligned(16))) float ctab[];             // dynamic: [3][C] A, O, k per channel (16-byte reads) | [2][G] mean, rstd per group
    float* const gtab = ctab + 3 * C;
    gn_bwd_group_table(b, HW, C, G, fsums, eps, act, gtab, gtab + G, fsums2, C1);
    __syncthreads();
    gn_bwd_channel_table(b, C, G, gtab, gtab + G, gamma, beta, scale_shift, ss_stride, ctab);
    __syncthreads();
    float A[V], O[V], K[V], M[V], R[V];
    if (lane_row < rif) gn_bwd_coeffs<V>(cv, C, G, gtab, gtab + G, ctab, A, O, K, M, R);
    if (lane_row < rif) {
        float s[V], q[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { s[i] = 0.f; q[i] = 0.f; }
        for (uint32_t r0 = lane_row; r0 < rows_per_block; r0 += rif * UR) {
            if (r0 != lane_row) issue(r0);
#pragma unroll
            for (int k = 0; k < UR; ++k) {
                if (r0 + k * rif >= rows_per_block) break;
                float f[V], d[V];
                GnVec<DT>::cvt(rx[k], f);
                GnVec<DT>::cvt(rd[k], d);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float p, xh;
                    ssdg_elem(f[i], d[i], A[i], O[i], K[i], M[i], R[i], act & 1, &p, &xh);
                    s[i] += p;
                    q[i] = __builtin_fmaf(p, xh, q[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) { part_s[lane_row * C + cv * V + i] = s[i]; part_q[lane_row * C + cv * V + i] = q[i]; }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < C; c += GN_TPB) {
        float ss = 0.f, qq = 0.f;
        for (uint32_t r = 0; r < rif; ++r) { ss += part_s[r * C + c]; qq += part_q[r * C + c]; }
        part_s[c] = ss; part_q[c] = qq;
    }
    __syncthreads();
    const uint32_t cpg = C / G;
    for (uint32_t g = threadIdx.x; g < G; g += GN_TPB) {
        double ds = 0.0, dq = 0.0;
        for (uint32_t i = 0; i < cpg; ++i) { ds += (double)part_s[g * cpg + i]; dq += (double)part_q[g * cpg + i]; }
        // r06: GN_BWD_REPLICAS copies of the sums, a block adds to copy (blockIdx.x mod replicas): device-scope fp64 atomics on ONE address are performed one after
        // the other at the memory side (~0.15 us each, measured: 1024 blocks per launch on 512 addresses cost 20 us); k_gn_bwd_apply adds the copies up
        double* rep = bsums + (size_t)(blockIdx.x % GN_BWD_REPLICAS) * gridDim.y * G * 2;
        atomicAdd(&rep[((size_t)b * G + g) * 2 + 0], ds);
        atomicAdd(&rep[((size_t)b * G + g) * 2 + 1], dq);
    }
}

template <int DT>
__global__ __launch_bounds__(GN_TPB) void k_gn_bwd_apply(const void* __restrict__ x, const void* __restrict__ dy, uint32_t HW, uint32_t C, uint32_t G,
                                                          uint32_t rows_per_block, const double* __restrict__ fsums, const double* __restrict__ bsums,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale_shift,
                                                          uint32_t ss_stride, float eps, int act, void* __restrict__ dx, const void* __restrict__ x2, uint32_t C1,
                                                          const double* __restrict__ fsums2, void* __restrict__ dx2) {
    constexpr int V = GnVec<DT>::V;
    const uint32_t tpr = C / V, rif = GN_TPB / tpr;
    const uint32_t lane_row = threadIdx.x / tpr, cv = threadIdx.x % tpr;
    const uint32_t b = blockIdx.y, row0 = blockIdx.x * rows_per_block;
    extern __shared__ __attribute__((aligned(16))) float ctab[];             // dynamic: [3][C] A, O, k per channel (16-byte reads) | [4][G] mean, rstd, mean(p), mean(p * xhat) per group
    float* const gtab = ctab + 3 * C;
    const uint32_t cpg = C / G;
    constexpr int UR = GN_BWD_UNROLL;
    const size_t base = ((size_t)b * HW + row0) * tpr + cv;
    const uint32_t tpr1 = (x2 ? C1 : C) / V;                                  // (r06) two sources / two gradients: see k_gn_bwd_stats
    const bool second = cv >= tpr1;
    const void* const xs = second ? x2 : x;
    void* const dxs = second ? dx2 : dx;
    const uint32_t stpr = second ? tpr - tpr1 : tpr1;
    const size_t sbase = ((size_t)b * HW + row0) * stpr + (second ? cv - tpr1 : cv);
    uint4 rx[UR], rd[UR];
    auto issue = [&](uint32_t r0) {
#pragma unroll
        for (int k = 0; k < UR; ++k)
            if (r0 + k * rif < rows_per_block) {
                rx[k] = GnVec<DT>::raw(xs, sbase + (size_t)(r0 + k * rif) * stpr);
                rd[k] = GnVec<DT>::raw(dy, base + (size_t)(r0 + k * rif) * tpr);
            }
    };
    if (lane_row < rif) issue(lane_row);                                     // (ahead of the prologue's own loads, as in k_gn_bwd_stats)
    gn_bwd_group_table(b, HW, C, G, fsums, eps, act, gtab, gtab + G, fsums2, C1);
    {
        const double inv_n = 1.0 / ((double)HW * (double)cpg);
        for (uint32_t g = threadIdx.x; g < G; g += GN_TPB) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (uint32_t r = 0; r < GN_BWD_REPLICAS; ++r) {                    // (fixed order: the same sums in every block)
                const double2 v = *reinterpret_cast<const double2*>(bsums + ((size_t)r * gridDim.y * G + (size_t)b * G + g) * 2);
                s0 += v.x; s1 += v.y;
            }
            gtab[2 * G + g] = (float)(s0 * inv_n);
            gtab[3 * G + g] = (float)(s1 * inv_n);
        }
    }
    __syncthreads();
    gn_bwd_channel_table(b, C, G, gtab, gtab + G, gamma, beta, scale_shift, ss_stride, ctab);
    __syncthreads();
    if (lane_row >= rif) return;
    float A[V], O[V], K[V], M[V], R[V], m1[V], m2[V];
    gn_bwd_coeffs<V>(cv, C, G, gtab, gtab + G, ctab, A, O, K, M, R);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const uint32_t g = (cv * V + i) / cpg;
        m1[i] = gtab[2 * G + g];
        m2[i] = gtab[3 * G + g];
    }
    for (uint32_t r0 = lane_row; r0 < rows_per_block; r0 += rif * UR) {
        if (r0 != lane_row) issue(r0);
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            if (r0 + k * rif >= rows_per_block) break;
            float f[V], d[V];
            GnVec<DT>::cvt(rx[k], f);
            GnVec<DT>::cvt(rd[k], d);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float p, xh;
                ssdg_elem(f[i], d[i], A[i], O[i], K[i], M[i], R[i], act & 1, &p, &xh);
                f[i] = ssdg_dx(p, xh, R[i], m1[i], m2[i]);
            }
            if constexpr (DT == GN_F32) {
                if (act & 2) { GnVec<DT>::store_split(dx, base + (size_t)(r0 + k * rif) * tpr, f); continue; }      // (host: single source only)   // (host: C % 32 == 0) dx PRE-SPLIT for the backward convolution
            }
            GnVec<DT>::store(dxs, sbase + (size_t)(r0 + k * rif) * stpr, f);
        }
    }
}

// fp32 [pixel][C] -> the PRE-SPLIT layout (store_split) and nothing else: for a convolution operand that does not come out of a norm (the gradient path's
// accumulated dy in front of a residual block's second convolution).  One 4-channel vector per thread, grid-stride.  The source may be a CHANNEL SLICE of a wider
// channel-last tensor (x_stride floats from pixel to pixel: what autograd hands back for one input of a concatenation) -- the dense copy a consumer would
// otherwise make first (125 us for 8 x 256 x 128 x 128) is folded into this pass.
__global__ __launch_bounds__(256) void k_split_f32(const float* __restrict__ x, void* __restrict__ y, size_t n_vec, uint32_t vpp, size_t x_stride) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / vpp;
        const float4 v = *reinterpret_cast<const float4*>(x + pix * x_stride + (i - pix * vpp) * 4);
        const float f[4] = {v.x, v.y, v.z, v.w};
        GnVec<GN_F32>::store_split(y, i, f);
    }
}

// least blocks / rows per block of the backward's passes.  r04 A/B, one guided step (71 norms, profiles/r04/w_gn_bwd_grid_ab.txt): statistics pass with
// (1024 blocks, >= 64 rows) 1.94 ms, (2048, 32) 1.52, (4096, 16) 1.67 -- the rows of a block are walked serially by eight row groups, two 16-byte loads in flight
// each: more, shorter blocks hide the load latency better until the fp64 atomics per block take over; the normalisation pass does not care (1.24 ms either way).
// (forward grids: r04 A/B of (4096 blocks, 8 rows) / (8192, 4) for the normalisation pass and (2048, 32) for the statistics pass: UNet step 8.16 -> 8.12 / 8.19 ms fp32,
//  4.25 -> 4.27 / 4.36 ms bf16 -- nothing to gain, unlike the backward's statistics pass below)
#ifndef GN_FWD_STATS_BLOCKS
#define GN_FWD_STATS_BLOCKS 1024
#define GN_FWD_STATS_ROWS 64
#endif
#ifndef GN_FWD_APPLY_BLOCKS
#define GN_FWD_APPLY_BLOCKS 2048
#define GN_FWD_APPLY_ROWS 16
#endif
#ifndef GN_BWD_STATS_BLOCKS
#define GN_BWD_STATS_BLOCKS 1024
#endif
#ifndef GN_BWD_MIN_TRIPS
#define GN_BWD_MIN_TRIPS 1
#endif
#ifndef GN_BWD_APPLY_BLOCKS
#define GN_BWD_APPLY_BLOCKS 1024
#endif
uint32_t gn_rows_per_block(uint32_t B, uint32_t HW, uint32_t min_blocks, uint32_t min_rows) {
    uint32_t rows = HW;                                 // largest power-of-two split of HW that still leaves >= min_blocks blocks
    while (rows > min_rows && (rows % 2 == 0) && (uint64_t)B * (HW / rows) < min_blocks) rows /= 2;
    return rows;
}

}  // namespace

extern "C" size_t ssdnerf_group_norm_workspace(uint32_t B, uint32_t G) { return (size_t)B * G * 2 * sizeof(double); }
extern "C" size_t ssdnerf_group_norm_backward_workspace(uint32_t B, uint32_t G) { return (size_t)GN_BWD_REPLICAS * B * G * 2 * sizeof(double); }

extern "C" int ssdnerf_group_norm_nhwc(const void* x, const void* x2, uint32_t C1, int dtype, uint32_t B, uint32_t HW, uint32_t C, uint32_t G, const float* pre_bias, const float* gamma,
                                       const float* beta, const float* scale_shift, uint32_t scale_shift_stride, float eps, int act, void* workspace,
                                       int workspace_state, void* y, void* stream) {
    if (B == 0 || HW == 0 || C == 0) return SSDNERF_OK;
    SSD_REQUIRE(x && y && gamma && beta && workspace, "group_norm_nhwc: null pointer");
    SSD_REQUIRE(dtype == GN_F32 || dtype == GN_F16 || dtype == GN_BF16, "group_norm_nhwc: dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    const uint32_t V = dtype == GN_F32 ? 4 : 8;
    SSD_REQUIRE(G > 0 && C % G == 0, "group_norm_nhwc: channels must be divisible by groups");
    SSD_REQUIRE(!scale_shift || scale_shift_stride >= 2 * C, "group_norm_nhwc: scale_shift_stride must be >= 2*C");
    SSD_REQUIRE(C % V == 0 && C / V <= GN_TPB && C <= GN_MAX_C, "group_norm_nhwc: channel count must be a multiple of the 16-byte vector and <= 1024 (f32) / 2048 (16-bit)");
    if (!x2) C1 = C;
    SSD_REQUIRE(C1 <= C && C1 % V == 0 && (x2 || C1 == C), "group_norm_nhwc: the first tensor's channel count must be a multiple of the 16-byte vector and <= C");
    SSD_REQUIRE(!(act & 2) || (dtype == GN_F32 && C % 32 == 0), "group_norm_nhwc: the pre-split output (act & 2) needs fp32 and C % 32 == 0");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t rows_s = gn_rows_per_block(B, HW, GN_FWD_STATS_BLOCKS, GN_FWD_STATS_ROWS), rows_a = gn_rows_per_block(B, HW, GN_FWD_APPLY_BLOCKS, GN_FWD_APPLY_ROWS);
    const dim3 grid_s(HW / rows_s, B), grid_a(HW / rows_a, B), block(GN_TPB);
    SSD_REQUIRE(workspace_state >= 0 && workspace_state <= 2 && !(workspace_state == 2 && pre_bias), "group_norm_nhwc: bad workspace_state");
    const bool stats_ready = workspace_state == 2;
    if (workspace_state == 0 && hipMemsetAsync(workspace, 0, ssdnerf_group_norm_workspace(B, G), st) != hipSuccess)
        return ssdnerf_fail(SSDNERF_E_LAUNCH, "group_norm_nhwc: memset failed");
    double* sums = (double*)workspace;
#define SSD_GN_LAUNCH(DT)                                                                                                                    \
    if (!stats_ready) hipLaunchKernelGGL(k_gn_stats<DT>, grid_s, block, 0, st, x, x2, C1, pre_bias, HW, C, G, rows_s, sums);                                            \
    hipLaunchKernelGGL(k_gn_apply<DT>, grid_a, block, 0, st, x, x2, C1, pre_bias, HW, C, G, rows_a, (const double*)sums, gamma, beta, scale_shift,    \
                       scale_shift_stride, eps, act, y);
    if (dtype == GN_F32) { SSD_GN_LAUNCH(GN_F32) } else if (dtype == GN_F16) { SSD_GN_LAUNCH(GN_F16) } else { SSD_GN_LAUNCH(GN_BF16) }
#undef SSD_GN_LAUNCH
    SSD_CHECK_LAUNCH("group_norm_nhwc");
    return SSDNERF_OK;
}

// The normalisation pass alone, from RUN-level statistics (k_gn_apply<RUNS>): runs1 fp64 [B][C1 / 4][2] for x, runs2 [B][(C - C1) / 4][2] for x2.
extern "C" int ssdnerf_group_norm_nhwc_runs(const void* x, const void* x2, uint32_t C1, int dtype, uint32_t B, uint32_t HW, uint32_t C, uint32_t G, const float* gamma,
                                            const float* beta, const float* scale_shift, uint32_t scale_shift_stride, float eps, int act, const void* runs1,
                                            const void* runs2, void* y, void* stream) {
    if (B == 0 || HW == 0 || C == 0) return SSDNERF_OK;
    SSD_REQUIRE(x && y && gamma && beta && runs1 && (!x2 || runs2), "group_norm_nhwc_runs: null pointer");
    SSD_REQUIRE(dtype == GN_F32 || dtype == GN_F16 || dtype == GN_BF16, "group_norm_nhwc_runs: dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    const uint32_t V = dtype == GN_F32 ? 4 : 8;
    SSD_REQUIRE(G > 0 && G <= GN_TPB && C % G == 0 && (C / G) % 4 == 0, "group_norm_nhwc_runs: groups must be multiples of 4 channels (and at most 256 groups)");
    SSD_REQUIRE(!scale_shift || scale_shift_stride >= 2 * C, "group_norm_nhwc_runs: scale_shift_stride must be >= 2*C");
    SSD_REQUIRE(C % V == 0 && C / V <= GN_TPB && C <= GN_MAX_C, "group_norm_nhwc_runs: channel count must be a multiple of the 16-byte vector and <= 1024 (f32) / 2048 (16-bit)");
    if (!x2) C1 = C;
    SSD_REQUIRE(C1 <= C && C1 % V == 0 && (x2 || C1 == C), "group_norm_nhwc_runs: the first tensor's channel count must be a multiple of the 16-byte vector and <= C");
    SSD_REQUIRE(!(act & 2) || (dtype == GN_F32 && C % 32 == 0), "group_norm_nhwc_runs: the pre-split output (act & 2) needs fp32 and C % 32 == 0");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t rows_a = gn_rows_per_block(B, HW, GN_FWD_APPLY_BLOCKS, GN_FWD_APPLY_ROWS);
    const dim3 grid_a(HW / rows_a, B), block(GN_TPB);
#define SSD_GN_LAUNCH(DT)                                                                                                                    \
    hipLaunchKernelGGL((k_gn_apply<DT, true>), grid_a, block, 0, st, x, x2, C1, (const float*)nullptr, HW, C, G, rows_a, (const double*)runs1, gamma, beta, scale_shift,    \
                       scale_shift_stride, eps, act, y, (const double*)runs2);
    if (dtype == GN_F32) { SSD_GN_LAUNCH(GN_F32) } else if (dtype == GN_F16) { SSD_GN_LAUNCH(GN_F16) } else { SSD_GN_LAUNCH(GN_BF16) }
#undef SSD_GN_LAUNCH
    SSD_CHECK_LAUNCH("group_norm_nhwc_runs");
    return SSDNERF_OK;
}

// d/dx of ssdnerf_group_norm_nhwc (no pre_bias) given dy; `fwd_sums` is the forward's workspace (per sample and group:
// sum, sum of squares of x), `bwd_workspace` ssdnerf_group_norm_backward_workspace(B, G) bytes (r06: replicated sums), zero-filled here unless
// `bwd_workspace_is_zero` (stream capture: let the caller zero it with a kernel).
// _cat (r06): x is the channel concatenation [x | x2] (C1 channels from x) that the forward never built, and the gradient leaves as TWO dense tensors, dx
// [B][HW][C1] and dx2 [B][HW][C - C1] -- the skip connections of the UNet's decoder half: autograd's torch.cat returned channel SLICES of one tensor, which every
// consumer copied dense first.  fwd_sums2 (act & 4 only): the run-level statistics of x2 (fwd_sums: those of x).
extern "C" int ssdnerf_group_norm_nhwc_backward_cat(const void* x, const void* x2, uint32_t C1, const void* dy, int dtype, uint32_t B, uint32_t HW, uint32_t C, uint32_t G,
                                                    const float* gamma, const float* beta, const float* scale_shift, uint32_t scale_shift_stride, float eps, int act,
                                                    const void* fwd_sums, const void* fwd_sums2, void* bwd_workspace, int bwd_workspace_is_zero, void* dx, void* dx2,
                                                    void* stream) {
    if (B == 0 || HW == 0 || C == 0) return SSDNERF_OK;
    SSD_REQUIRE(x && dy && dx && gamma && beta && fwd_sums && bwd_workspace, "group_norm_nhwc_backward: null pointer");
    SSD_REQUIRE(dtype == GN_F32 || dtype == GN_F16 || dtype == GN_BF16, "group_norm_nhwc_backward: dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    const uint32_t V = dtype == GN_F32 ? 4 : 8;
    SSD_REQUIRE(G > 0 && C % G == 0, "group_norm_nhwc_backward: channels must be divisible by groups");
    SSD_REQUIRE(!scale_shift || scale_shift_stride >= 2 * C, "group_norm_nhwc_backward: scale_shift_stride must be >= 2*C");
    SSD_REQUIRE(C % V == 0 && C / V <= GN_TPB && C <= GN_MAX_C, "group_norm_nhwc_backward: channel count must be a multiple of the 16-byte vector and <= 1024 (f32) / 2048 (16-bit)");
    SSD_REQUIRE(!(act & 2) || (dtype == GN_F32 && C % 32 == 0), "group_norm_nhwc_backward: the pre-split dx (act & 2) needs fp32 and C % 32 == 0");
    SSD_REQUIRE(!(act & 4) || (C / G) % 4 == 0, "group_norm_nhwc_backward: run-level forward sums (act & 4) need groups of a multiple of 4 channels");
    if (x2) {
        SSD_REQUIRE(dx2 && C1 > 0 && C1 < C && C1 % V == 0, "group_norm_nhwc_backward_cat: needs dx2 and 0 < C1 < C, C1 a multiple of the 16-byte vector");
        SSD_REQUIRE(!(act & 2), "group_norm_nhwc_backward_cat: no pre-split dx for two sources");
        SSD_REQUIRE(!(act & 4) || (fwd_sums2 && C1 % 4 == 0), "group_norm_nhwc_backward_cat: run-level sums need fwd_sums2 and C1 % 4 == 0");
    } else {
        C1 = C; fwd_sums2 = nullptr; dx2 = nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    // r06: rows per block from the thread layout -- at least one row per row group (C / V threads share a row, 256 / (C / V) rows are in flight), so that the
    // small tensors (8 x 8, 16 x 16: 64 - 256 rows per sample) spread over 100+ blocks instead of 16 blocks walking 8 trips each
    const uint32_t rif = GN_TPB / (C / V) ? GN_TPB / (C / V) : 1;
    const uint32_t rows_s = gn_rows_per_block(B, HW, GN_BWD_STATS_BLOCKS, rif * GN_BWD_MIN_TRIPS), rows_a = gn_rows_per_block(B, HW, GN_BWD_APPLY_BLOCKS, rif * GN_BWD_MIN_TRIPS);
    const dim3 grid_s(HW / rows_s, B), grid_a(HW / rows_a, B), block(GN_TPB);
    if (!bwd_workspace_is_zero && hipMemsetAsync(bwd_workspace, 0, ssdnerf_group_norm_backward_workspace(B, G), st) != hipSuccess)
        return ssdnerf_fail(SSDNERF_E_LAUNCH, "group_norm_nhwc_backward: memset failed");
#define SSD_GNB_LAUNCH(DT)                                                                                                                              \
    hipLaunchKernelGGL(k_gn_bwd_stats<DT>, grid_s, block, ((size_t)2 * G + 3 * C) * 4, st, x, dy, HW, C, G, rows_s, (const double*)fwd_sums, gamma, beta, scale_shift, scale_shift_stride, \
                       eps, act, (double*)bwd_workspace, x2, C1, (const double*)fwd_sums2);                                                            \
    hipLaunchKernelGGL(k_gn_bwd_apply<DT>, grid_a, block, ((size_t)4 * G + 3 * C) * 4, st, x, dy, HW, C, G, rows_a, (const double*)fwd_sums, (const double*)bwd_workspace, gamma, beta,   \
                       scale_shift, scale_shift_stride, eps, act, dx, x2, C1, (const double*)fwd_sums2, dx2);
    if (dtype == GN_F32) { SSD_GNB_LAUNCH(GN_F32) } else if (dtype == GN_F16) { SSD_GNB_LAUNCH(GN_F16) } else { SSD_GNB_LAUNCH(GN_BF16) }
#undef SSD_GNB_LAUNCH
    SSD_CHECK_LAUNCH("group_norm_nhwc_backward");
    return SSDNERF_OK;
}

extern "C" int ssdnerf_group_norm_nhwc_backward(const void* x, const void* dy, int dtype, uint32_t B, uint32_t HW, uint32_t C, uint32_t G, const float* gamma,
                                                const float* beta, const float* scale_shift, uint32_t scale_shift_stride, float eps, int act,
                                                const void* fwd_sums, void* bwd_workspace, int bwd_workspace_is_zero, void* dx, void* stream) {
    return ssdnerf_group_norm_nhwc_backward_cat(x, nullptr, C, dy, dtype, B, HW, C, G, gamma, beta, scale_shift, scale_shift_stride, eps, act, fwd_sums, nullptr, bwd_workspace,
                                                bwd_workspace_is_zero, dx, nullptr, stream);
}

extern "C" int ssdnerf_bias_residual_nhwc(const void* x, int dtype, uint32_t B, uint32_t HW, uint32_t C, const float* bias, const void* residual, void* y,
                                          void* gn_sums, uint32_t gn_groups, void* stream) {
    if (B == 0 || HW == 0 || C == 0) return SSDNERF_OK;
    SSD_REQUIRE(x && y, "bias_residual_nhwc: null pointer");
    SSD_REQUIRE(dtype == GN_F32 || dtype == GN_F16 || dtype == GN_BF16, "bias_residual_nhwc: dtype must be 0 (f32), 1 (f16) or 2 (bf16)");
    const uint32_t V = dtype == GN_F32 ? 4 : 8;
    SSD_REQUIRE(C % V == 0 && C / V <= GN_TPB && C <= GN_MAX_C, "bias_residual_nhwc: channel count must be a multiple of the 16-byte vector and <= 1024 (f32) / 2048 (16-bit)");
    SSD_REQUIRE(!gn_sums || (gn_groups > 0 && C % gn_groups == 0), "bias_residual_nhwc: channels must be divisible by groups");
    const uint32_t rows = gn_rows_per_block(B, HW, GN_FWD_APPLY_BLOCKS, GN_FWD_APPLY_ROWS);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(HW / rows, B), block(GN_TPB);
#define SSD_BR_LAUNCH(DT) hipLaunchKernelGGL(k_bias_residual<DT>, grid, block, 0, st, x, bias, residual, HW, C, rows, y, (double*)gn_sums, gn_groups ? gn_groups : 1u);
    if (dtype == GN_F32) { SSD_BR_LAUNCH(GN_F32) } else if (dtype == GN_F16) { SSD_BR_LAUNCH(GN_F16) } else { SSD_BR_LAUNCH(GN_BF16) }
#undef SSD_BR_LAUNCH
    SSD_CHECK_LAUNCH("bias_residual_nhwc");
    return SSDNERF_OK;
}

// y = x in the pre-split layout of ssdnerf_group_norm_nhwc's act | 2 (fp32 [pixels][C], C % 32 == 0; y: the same number of bytes): an operand for
// ssdnerf_conv2d_nhwc_f32x2_presplit that no norm produced.  Two passes over the tensor (17 us at 128 x 128 x 128 x 8) against the 60 us the two-group
// kernel's on-the-fly split costs per large layer.
extern "C" int ssdnerf_split_f32_nhwc(const void* x, void* y, uint64_t pixels, uint32_t C, uint64_t x_stride, void* stream) {
    if (pixels == 0 || C == 0) return SSDNERF_OK;
    SSD_REQUIRE(x && y && x != y, "split_f32_nhwc: null pointer / in place");
    SSD_REQUIRE(C % 32 == 0, "split_f32_nhwc: C %% 32 == 0");
    if (x_stride == 0) x_stride = C;
    SSD_REQUIRE(x_stride >= C && x_stride % 4 == 0 && ((uintptr_t)x & 15) == 0, "split_f32_nhwc: the source's pixel stride must be >= C and, like its base, a multiple of 16 bytes");
    const size_t n_vec = (size_t)pixels * (C / 4);
    const size_t blocks = (n_vec + 255) / 256;
    hipLaunchKernelGGL(k_split_f32, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, (const float*)x, y, n_vec, C / 4, (size_t)x_stride);
    SSD_CHECK_LAUNCH("split_f32_nhwc");
    return SSDNERF_OK;
}
