// tools/ubench/swap_swap_hazard.hip -- do two waves that share a SIMD disturb each other's v_permlane32_swap?
// r06 hypothesis: the failures of k_shade_mfma that need two waves per SIMD (profiles/r05/m_*, zz_*; r06/c_loud.txt) come from v_permlane32_swap_b32 itself --
// r05's probe of the swap -> MFMA pair kept the PARTNER wave busy with other instructions (matrix, SiLU mix) and never with swaps of its own.
// Here every wave runs bursts of swaps on registers whose contents name (wave, lane, register, round), with pseudo-random gaps of plain VALU work between the bursts
// so that the bursts of the two waves of a SIMD overlap at every offset; after each burst every lane checks what it received.
//   arg 1: waves per SIMD (1 or 2: blocks per CU); arg 2: rounds; arg 3: mode (0 swaps only, 1 + v_exp / packed noise between swaps, 2 + MFMA in the gaps)
// Build: hipcc --offload-arch=gfx950 -O3 swap_swap_hazard.hip -o swap_swap_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(uint32_t* __restrict__ bad, uint32_t* __restrict__ first, int rounds) {
    const uint32_t lane = threadIdx.x & 63, wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t half = lane >> 5;
    uint32_t rng = wave_id * 2654435761u + 12345u;
    uint32_t nbad = 0;
    float noise = (float)lane * 1e-3f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk = {1.0f, 1.0f};
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int r = 0; r < rounds; ++r) {
        uint32_t a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[i] = (wave_id << 20) ^ ((uint32_t)r << 10) ^ (lane << 4) ^ (uint32_t)i;                  // "A of (wave, round, lane, i)"
            b[i] = a[i] ^ 0x80000000u;                                                                 // "B of ..."
        }
        // a wave-uniform pseudo-random gap so that the partner's burst lands anywhere relative to this one
        rng = rng * 1664525u + 1013904223u;
        const int gap = __builtin_amdgcn_readfirstlane((rng >> 24) & 31);
        for (int g = 0; g < gap; ++g) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(noise));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            if (MODE == 1) { asm volatile("v_exp_f32 %0, %0" : "+v"(noise)); asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(pk)); }
            if (MODE == 2) {
                bf16x8 x; for (int q = 0; q < 8; ++q) x[q] = (__bf16)1.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc, 0, 0, 0);
            }
        }
        asm volatile("s_nop 3");
        __builtin_amdgcn_sched_barrier(0);
        // expected: a' = [own a (lanes < 32) ; b of lane - 32], b' = [a of lane + 32 ; own b]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t own_a = (wave_id << 20) ^ ((uint32_t)r << 10) ^ (lane << 4) ^ (uint32_t)i, own_b = own_a ^ 0x80000000u;
            const uint32_t partner = lane ^ 32u;
            const uint32_t p_a = (wave_id << 20) ^ ((uint32_t)r << 10) ^ (partner << 4) ^ (uint32_t)i, p_b = p_a ^ 0x80000000u;
            const uint32_t want_a = half == 0 ? own_a : p_b, want_b = half == 0 ? p_a : own_b;
            if (a[i] != want_a || b[i] != want_b) {
                ++nbad;
                if (atomicAdd(bad + 1, 1u) < 16u) {                                                     // keep the first few: (wave, round, lane, i, got a, want a, got b, want b)
                    uint32_t* d = first + 8 * atomicAdd(bad + 2, 1u);
                    d[0] = wave_id; d[1] = (uint32_t)r; d[2] = lane; d[3] = (uint32_t)i; d[4] = a[i]; d[5] = want_a; d[6] = b[i]; d[7] = want_b;
                }
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (noise == 123.456f || acc[0] == 77.f || pk.x == 5.f) bad[3] = 1;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2, rounds = argc > 2 ? atoi(argv[2]) : 200000, mode = argc > 3 ? atoi(argv[3]) : 0;
    uint32_t *bad, *first;
    hipMalloc(&bad, 16); hipMalloc(&first, 8 * 4 * 64);
    hipMemset(bad, 0, 16); hipMemset(first, 0, 8 * 4 * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, bad, first, rounds);
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, bad, first, rounds);
    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), 0, 0, bad, first, rounds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[4], f[8 * 16];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(f, first, sizeof(f), hipMemcpyDeviceToHost);
    printf("waves/SIMD %d, mode %d, %d rounds x 8 swaps x %d waves: %u wrong (lane, register) results  (%.0f ms)\n", wps, mode, rounds, 1024 * wps, h[0], ms);
    for (uint32_t i = 0; i < (h[2] < 16 ? h[2] : 16); ++i)
        printf("  wave %u round %u lane %u reg %u: a %08x (want %08x)  b %08x (want %08x)\n", f[8 * i], f[8 * i + 1], f[8 * i + 2], f[8 * i + 3], f[8 * i + 4], f[8 * i + 5], f[8 * i + 6], f[8 * i + 7]);
    return 0;
}
