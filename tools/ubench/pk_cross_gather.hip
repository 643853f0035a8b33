// tools/ubench/pk_cross_gather.hip -- second stand-alone probe of the round-6 finding (see pk_cross_hazard.hip): the triplane gather of k_shade_mfma by itself.
// Every wave, in a loop: a pseudo-random subset of lanes ("on", as in the shading kernel: the blend runs under a partial EXEC mask) computes a sample position, requests
// its twelve texels (dwordx4 + dwordx2 each, 32-bit offsets from an SGPR base), blends them with the PACKED chain the compiler forms from the kernel's source (op_sel
// broadcasts of the four weights) and, from the same registers, with pinned plain v_mul / v_fma; lanes where the two disagree are counted by lane quarter and feature.
//   usage: pk_cross_gather WAVES_PER_SIMD ROUNDS [FILL]   FILL (bits): 1 LDS reads, 2 v_exp / v_rcp, 4 packed fp32, 8 MFMA between gathers (the shading kernel's MLP mix); 15 all
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__
DEV float pmul(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float pfma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
DEV void coord(float u, uint32_t& i0, uint32_t& i1, float& w0, float& w1) {
    float ix = ((u + 1.0f) * 128.0f - 1.0f) * 0.5f;
    ix = fminf(127.0f, fmaxf(ix, 0.0f));
    const float fl = floorf(ix);
    i0 = (uint32_t)fl; i1 = min(i0 + 1u, 127u); w1 = ix - fl; w0 = (fl + 1.0f) - ix;
}
struct Tap {
    float t00[6], t01[6], t10[6], t11[6], w00, w01, w10, w11;
    DEV void load6(const float* base, uint32_t off, float v[6]) {
        const char* q = reinterpret_cast<const char*>(base) + off;
        const float4 a = *reinterpret_cast<const float4*>(q); const float2 b = *reinterpret_cast<const float2*>(q + 16);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y;
    }
    DEV void issue(const float* planes, int p, float u, float v) {
        uint32_t x0, x1, y0, y1; float wx0, wx1, wy0, wy1;
        coord(u, x0, x1, wx0, wx1); coord(v, y0, y1, wy0, wy1);
        const uint32_t row0 = ((uint32_t)p * 128u + y0) * 128u, row1 = ((uint32_t)p * 128u + y1) * 128u;
        load6(planes, (row0 + x0) * 32u, t00); load6(planes, (row0 + x1) * 32u, t01); load6(planes, (row1 + x0) * 32u, t10); load6(planes, (row1 + x1) * 32u, t11);
        w00 = wx0 * wy0; w01 = wx1 * wy0; w10 = wx0 * wy1; w11 = wx1 * wy1;
    }
    DEV void blend_packed(int p, float f[18]) const {
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
            const f2 a00 = {t00[c], t00[c + 1]}, a01 = {t01[c], t01[c + 1]}, a10 = {t10[c], t10[c + 1]}, a11 = {t11[c], t11[c + 1]};
            f2 r = a00 * f2{w00, w00};
            r = __builtin_elementwise_fma(a01, f2{w01, w01}, r); r = __builtin_elementwise_fma(a10, f2{w10, w10}, r); r = __builtin_elementwise_fma(a11, f2{w11, w11}, r);
            f[c * 3 + p] = r.x; f[(c + 1) * 3 + p] = r.y;
        }
    }
    DEV void blend_plain(int p, float f[18]) const {
#pragma unroll
        for (int c = 0; c < 6; ++c) { float r = pmul(t00[c], w00); r = pfma(t01[c], w01, r); r = pfma(t10[c], w10, r); r = pfma(t11[c], w11, r); f[c * 3 + p] = r; }
    }
};

template <int FILL>
__global__ void __launch_bounds__(256, 2) k(uint32_t* __restrict__ bad, const float* __restrict__ planes, int rounds) {
    const uint32_t lane = threadIdx.x & 63, wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t rng = (wave_id * 64u + lane) * 2654435761u + 777u;
    float tpar = 0.01f * (float)(lane + 1), acc_out = 0.f;
    float e[8]; for (int i = 0; i < 8; ++i) e[i] = 0.1f * (float)(i + 1) + 0.001f * (float)lane;
    floatx16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    uint32_t nb[4] = {0, 0, 0, 0};
    __shared__ uint4 lds_buf[4 * 64 * 3];
    for (int i = threadIdx.x; i < 4 * 64 * 3; i += 256) lds_buf[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(lds_buf + (threadIdx.x >> 6) * 192) + lane * 16u;   // (shared -> LDS byte address: low 32 bits)
    const f2 pk1 = {0.999f, 1.001f};
    const float dx = 0.31f + 0.001f * (float)lane, dy = -0.22f + 0.002f * (float)(wave_id & 15), dz = 0.17f;
    for (int r = 0; r < rounds; ++r) {
        rng = rng * 1664525u + 1013904223u;
        const bool on = ((rng >> 20) & 15u) != 0u;                     // ~94 % of the lanes shade, as in the kernel
        tpar += 0.0135f; if (tpar > 1.6f) tpar -= 3.2f;
        float f[18], g[18];
        if (on) {
            const float sx = __builtin_amdgcn_fmed3f(__builtin_fmaf(tpar, dx, 0.1f), -1.f, 1.f), sy = __builtin_amdgcn_fmed3f(__builtin_fmaf(tpar, dy, -0.05f), -1.f, 1.f),
                        sz = __builtin_amdgcn_fmed3f(__builtin_fmaf(tpar, dz, 0.02f), -1.f, 1.f);
            Tap tap[3];
            tap[0].issue(planes, 0, sx, sy); tap[1].issue(planes, 1, sx, sz); tap[2].issue(planes, 2, sy, sz);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 3; ++p) tap[p].blend_packed(p, f);
#pragma unroll
            for (int p = 0; p < 3; ++p) tap[p].blend_plain(p, g);
            bool any = false;
#pragma unroll
            for (int i = 0; i < 18; ++i) any |= __float_as_uint(f[i]) != __float_as_uint(g[i]);
            if (any) { ++nb[0]; if (__float_as_uint(f[0]) != __float_as_uint(g[0]) || __float_as_uint(f[1]) != __float_as_uint(g[1]) || __float_as_uint(f[2]) != __float_as_uint(g[2])) ++nb[1]; }
            acc_out += f[0] + f[17] + g[5];
        }
        // the partner's instruction mix (the MLP of the shading kernel), pinned: FILL bit 0 LDS reads that return into VGPRs, 1 v_exp / v_rcp, 2 packed fp32 (uncrossed), 3 MFMA
        rng = rng * 1664525u + 1013904223u;
        const int reps = 1 + (int)__builtin_amdgcn_readfirstlane((rng >> 27) & 3);
        for (int rep = 0; rep < reps; ++rep) {
            if (FILL & 1) {
                uint4 q0, q1, q2;
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(lds_addr) : "memory");
                e[0] += __uint_as_float(q0.x & 0u) + __uint_as_float(q1.y & 0u) + __uint_as_float(q2.z & 0u);
            }
            if (FILL & 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_exp_f32 %0, %0" : "+v"(e[q]));
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_add_f32 %0, 1.0, %0\n\ts_nop 0\n\tv_rcp_f32 %0, %0" : "+v"(e[q]));
            }
            if (FILL & 4) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) { f2 pq = {e[q], e[q + 1]}; asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(pq) : "v"(pk1)); e[q] = pq.x; e[q + 1] = pq.y; }
            }
            if (FILL & 8) {
                bf16x8 x; for (int q = 0; q < 8; ++q) x[q] = (__bf16)1.0f;
#pragma unroll
                for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc, 0, 0, 0);
            }
        }
    }
    if (nb[0]) { atomicAdd(bad + (lane >> 4), nb[0]); atomicAdd(bad + 4, nb[1]); }
    float s = acc_out; for (int i = 0; i < 8; ++i) s += e[i]; for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) bad[8] = 1;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2, rounds = argc > 2 ? atoi(argv[2]) : 20000, fill = argc > 3 ? atoi(argv[3]) : 0;
    uint32_t* bad; float* planes;
    hipMalloc(&bad, 64); hipMemset(bad, 0, 64);
    const size_t n = (size_t)3 * 128 * 128 * 8;
    hipMalloc(&planes, n * 4);
    float* h = (float*)malloc(n * 4);
    uint32_t s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(int)(s >> 8 & 0xffff) * (4.0f / 65536.0f) - 2.0f; }
    hipMemcpy(planes, h, n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    dim3 g(256 * wps), b(256);
    if (fill == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, bad, planes, rounds);
    if (fill == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, bad, planes, rounds);
    if (fill == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, bad, planes, rounds);
    if (fill == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, bad, planes, rounds);
    if (fill == 8) hipLaunchKernelGGL(k<8>, g, b, 0, 0, bad, planes, rounds);
    if (fill == 15) hipLaunchKernelGGL(k<15>, g, b, 0, 0, bad, planes, rounds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t hb[16]; hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost);
    printf("waves/SIMD %d fill %d: %d rounds x %d waves: lanes whose packed blend differs from the plain one, by lane quarter [%u %u %u %u]; of them in features 0-2: %u  (%.0f ms)\n",
           wps, fill, rounds, 1024 * wps, hb[0], hb[1], hb[2], hb[3], hb[4], ms);
    return 0;
}
