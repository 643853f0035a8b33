// tools/ubench/mfma_valu_overlap.hip -- does VALU work hide under MFMA on gfx950?  For f32-input MFMA (32x32x2) and bf16 MFMA (32x32x16):
// time per loop iteration of (a) MFMAs only, (b) VALU FMAs only, (c) both interleaved in one wave, at 1, 2 and 3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int KIND>   // MODE 0 mfma only, 1 valu only, 2 both, 3 both with transcendental VALU; KIND 0 f32 mfma, 1 bf16 mfma
__global__ void k(float* out, int iters, long long* cycles) {
    f32x16 acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 0.001f + 1.0f, b = 0.5f;
    bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (__bf16)a;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE != 1) {
                if (KIND == 0) { acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0); }
                else { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc1, 0, 0, 0); }
            }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));      // 32 plain FMAs per u
            }
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));                                     // 8 transcendentals per u
            }
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int MODE, int KIND> double run(int waves_per_simd, float* out, long long* cyc) {
    const int iters = 2000;
    // one block of 256 threads = one wave per SIMD on its CU; waves_per_simd blocks per CU need LDS-free, register-light kernels: launch CUs * w blocks
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 8.0);     // ns per u-step (2 MFMAs and/or 32 FMAs / 8 exps) per resident wave set
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 3 * 256 * 4); hipMalloc(&cyc, 8);
    for (int w = 1; w <= 3; ++w) {
        printf("waves/SIMD %d  [ns per step: 2 MFMA | 32 v_fma | both | 2 MFMA + 8 v_exp]\n", w);
        printf("  f32  mfma 32x32x2 : %7.1f %7.1f %7.1f %7.1f\n", run<0, 0>(w, out, cyc), run<1, 0>(w, out, cyc), run<2, 0>(w, out, cyc), run<3, 0>(w, out, cyc));
        printf("  bf16 mfma 32x32x16: %7.1f %7.1f %7.1f %7.1f\n", run<0, 1>(w, out, cyc), run<1, 1>(w, out, cyc), run<2, 1>(w, out, cyc), run<3, 1>(w, out, cyc));
    }
    return 0;
}
