// tools/ubench/trans_rate.hip -- issue cost of transcendental VALU instructions on gfx950 and whether plain VALU work hides under them.
// Per loop step: (a) 8 independent v_exp_f32, (b) 8 independent v_rcp_f32, (c) 32 v_fma_f32, (d) 8 v_exp + 32 v_fma interleaved in one wave,
// (e) 8 v_exp + 16 v_pk_fma_f32; at 1, 2 and 3 waves per SIMD.  If (d) ~ max(a, c) the transcendental unit runs beside the FMA lanes and
// the SiLU heads of k_shade_mfma can be scheduled to hide their packed arithmetic under the exp/rcp stream; if (d) ~ a + c it cannot.
// Build: hipcc --offload-arch=gfx950 -O3 trans_rate.hip -o trans_rate ; run on the GPU box (prints ns per step and cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters) {
    float a = threadIdx.x * 0.001f + 1.0f, b = 0.999f;
    float v[8], w[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i * 0.01f; w[i] = b + i * 0.01f; p[i] = f32x2{a + i, b + i}; }
    const f32x2 pb = {b, b}, pa = {1e-3f, 1e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0 || MODE == 3 || MODE == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                    if (MODE == 3) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w[(i + r) & 7]) : "v"(b), "v"(a));
                    }
                    if (MODE == 4) {
#pragma unroll
                        for (int r = 0; r < 2; ++r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + r) & 7]) : "v"(pb), "v"(pa));
                    }
                }
            }
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            }
            if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(b), "v"(a));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> double run(int waves_per_simd, float* out) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 8.0);     // ns per step for the resident wave set (waves_per_simd waves share each SIMD)
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 3 * 256 * 4);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz; ns per step (and cycles per instruction per wave, assuming the SIMD is shared evenly)\n", ghz);
    for (int w = 1; w <= 3; ++w) {
        const double a = run<0>(w, out), b = run<1>(w, out), c = run<2>(w, out), d = run<3>(w, out), e = run<4>(w, out);
        printf("waves/SIMD %d: 8 v_exp %6.1f (%.1f cyc each) | 8 v_rcp %6.1f (%.1f) | 32 v_fma %6.1f (%.1f) | 8 exp + 32 fma %6.1f (sum %.1f, max %.1f) | 8 exp + 16 pk_fma %6.1f\n",
               w, a, a * ghz / (8.0 * w), b, b * ghz / (8.0 * w), c, c * ghz / (32.0 * w), d, a + c, a > c ? a : c, e);
    }
    return 0;
}
