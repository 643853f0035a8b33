// tools/ubench/mfma_valu_2wave.hip -- do MFMAs of ONE wave and VALU work of ANOTHER wave of the same SIMD overlap on gfx950?
// A block of 512 threads puts two waves on every SIMD of its CU (wave w -> SIMD w % 4): waves 0-3 issue only bf16 MFMAs, waves 4-7 only VALU
// work (plain FMAs / transcendentals / packed FMAs).  Printed: time with only the MFMA waves working, only the VALU waves, and both.
// both ~ max(a, b): the pipes overlap across waves;  both ~ a + b: an executing MFMA blocks the SIMD's VALU issue.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VKIND>   // 0 v_fma, 1 v_exp, 2 v_pk_fma
__global__ void __launch_bounds__(512) k(float* out, int iters, int do_mfma, int do_valu, int dependent) {
    const int role = threadIdx.x >> 8;             // 0: MFMA waves, 1: VALU waves
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    float a = threadIdx.x * 0.001f + 1.0f, b = 0.5f;
    bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (__bf16)a;
    float v[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; p[i] = f32x2{a + i, a - i}; }
    if (role == 0 && do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {          // 16 MFMAs per iteration: 4 independent accumulators, or ONE chain (dependent)
                if (dependent) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[0], 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[q], 0, 0, 0);
                }
            }
        }
    }
    if (role == 1 && do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {      // 128 VALU instructions per iteration, 8 independent chains
                    if (VKIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
                    if (VKIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                    if (VKIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                }
        }
    }
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VKIND> float run(float* out, int do_mfma, int do_valu, int dependent) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<VKIND>), dim3(256), dim3(512), 0, 0, out, iters, do_mfma, do_valu, dependent);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<VKIND>), dim3(256), dim3(512), 0, 0, out, iters, do_mfma, do_valu, dependent);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;                      // ns per iteration (16 MFMAs | 128 VALU)
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const char* names[3] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32"};
    for (int dep = 0; dep < 2; ++dep) {
        printf("%s MFMAs: ns per iteration [16 MFMA only | 128 VALU only | both on one SIMD from two waves]\n", dep ? "dependent (one accumulator chain)" : "independent");
        printf("  %-13s %7.1f %7.1f %7.1f\n", names[0], run<0>(out, 1, 0, dep), run<0>(out, 0, 1, dep), run<0>(out, 1, 1, dep));
        printf("  %-13s %7.1f %7.1f %7.1f\n", names[1], run<1>(out, 1, 0, dep), run<1>(out, 0, 1, dep), run<1>(out, 1, 1, dep));
        printf("  %-13s %7.1f %7.1f %7.1f\n", names[2], run<2>(out, 1, 0, dep), run<2>(out, 0, 1, dep), run<2>(out, 1, 1, dep));
    }
    return 0;
}
