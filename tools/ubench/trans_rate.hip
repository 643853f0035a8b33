// tools/ubench/trans_rate.hip -- issue cost of transcendental VALU instructions on gfx950 and whether plain VALU work hides under them.
// Per loop step: 8 independent transcendentals (v_exp_f32 or v_rcp_f32), each followed by R plain instructions (v_fma_f32, v_pk_fma_f32 or
// v_pk_mul_f32) on other registers; R = 0 .. 4, and the plain instructions alone; at 1, 2 and 3 waves per SIMD.
// If time(8 trans + 8 R plain) ~ max(time(8 trans), time(8 R plain)) the transcendental unit runs beside the FMA lanes and the SiLU heads of
// k_shade_mfma should interleave their packed arithmetic with the exp / rcp stream; if it is the sum they cannot hide.
// Build: hipcc --offload-arch=gfx950 -O3 trans_rate.hip -o trans_rate ; run on the GPU box.  r06: profiles/r06/a_trans_rate.txt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// TK: 0 none, 1 v_exp, 2 v_rcp, 3 alternating exp / rcp.  PK: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_mul_f32, 3 v_pk_add_f32.  R plain per transcendental slot.
template <int TK, int PK, int R>
__global__ void k(float* out, int iters) {
    float a = threadIdx.x * 0.001f + 1.0f, b = 0.999f;
    float v[8], w[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i * 0.01f; w[i] = b + i * 0.01f; p[i] = f32x2{a + i, b + i}; }
    const f32x2 pb = {b, b}, pa = {1e-3f, 1e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (TK == 1 || (TK == 3 && (i & 1) == 0)) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (TK == 2 || (TK == 3 && (i & 1) == 1)) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (PK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w[(i + r) & 7]) : "v"(b), "v"(a));
                    if (PK == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + r) & 7]) : "v"(pb), "v"(pa));
                    if (PK == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i + r) & 7]) : "v"(pb));
                    if (PK == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i + r) & 7]) : "v"(pa));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int TK, int PK, int R> double run(int waves_per_simd, float* out) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<TK, PK, R>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<TK, PK, R>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 8.0);     // ns per step (8 transcendental slots) for the resident wave set (waves_per_simd waves share each SIMD)
}

template <int PK> void table(const char* name, int w, double ghz, float* out) {
    const double t0 = run<1, PK, 0>(w, out), r0 = run<2, PK, 0>(w, out), x0 = run<3, PK, 0>(w, out);
    const double p1 = run<0, PK, 1>(w, out), p2 = run<0, PK, 2>(w, out), p3 = run<0, PK, 3>(w, out), p4 = run<0, PK, 4>(w, out);
    const double m1 = run<1, PK, 1>(w, out), m2 = run<1, PK, 2>(w, out), m3 = run<1, PK, 3>(w, out), m4 = run<1, PK, 4>(w, out);
    const double y2 = run<3, PK, 2>(w, out);
    const double c = ghz / (8.0 * w);      // ns per step -> SIMD cycles per slot of ONE wave's instruction (the SIMD is shared by w waves)
    printf("waves/SIMD %d, plain = %s\n", w, name);
    printf("  8 v_exp alone %7.1f ns (%.1f cyc each) | 8 v_rcp %7.1f (%.1f) | exp/rcp alternating %7.1f (%.1f)\n", t0, t0 * c, r0, r0 * c, x0, x0 * c);
    printf("  plain alone, R per slot:   R=1 %7.1f (%.1f cyc each)  R=2 %7.1f (%.1f)  R=3 %7.1f (%.1f)  R=4 %7.1f (%.1f)\n", p1, p1 * c, p2, p2 * c / 2, p3, p3 * c / 3, p4, p4 * c / 4);
    printf("  v_exp + R plain per slot:  R=1 %7.1f (sum %.1f)  R=2 %7.1f (sum %.1f)  R=3 %7.1f (sum %.1f)  R=4 %7.1f (sum %.1f)   [exp/rcp + 2: %7.1f]\n",
           m1, t0 + p1, m2, t0 + p2, m3, t0 + p3, m4, t0 + p4, y2);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 3 * 256 * 4);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("clock %.2f GHz; ns per step of 8 slots for the resident wave set; cycles = SIMD cycles per instruction\n", ghz);
    for (int w = 1; w <= 3; ++w) {
        table<0>("v_fma_f32", w, ghz, out);
        table<1>("v_pk_fma_f32", w, ghz, out);
        table<2>("v_pk_mul_f32", w, ghz, out);
    }
    return 0;
}
