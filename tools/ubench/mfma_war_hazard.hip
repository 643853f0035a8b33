// tools/ubench/mfma_war_hazard.hip -- does a matrix instruction that has to QUEUE behind the other wave's matrix work still own its source registers?
//
// Hypothesis for the run-to-run differences of k_shade_mfma: they appear only with two waves per SIMD, move with code placement, and the
// "transcendental -> use distance" explains them badly (profiles/r05/m_*).  A write-after-read hazard would fit: v_mfma reads its A / B operands when it STARTS on the
// matrix pipe; if the pipe is busy with the co-resident wave's MFMAs, the instruction waits -- while its own wave goes on issuing VALU instructions that may
// overwrite those operand registers (the toolchain's WAR table covers SrcC only).
// Test: waves 0-3 of a 512-thread block (one per SIMD) run   B <- pattern;  v_mfma acc, A, B, 0;  [k x s_nop 0];  B <- poison;  ...wait...;  keep acc
// and the same with a long wait in front of the poison; any bitwise difference between the two accumulators is the hazard.  Waves 4-7 (the co-resident wave of
// every SIMD) either idle or issue matrix instructions back to back.
//     hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_war_hazard.hip -o .variants/ub/mfma_war_hazard && .variants/ub/mfma_war_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float floatx16 __attribute__((ext_vector_type(16)));
#define STR2(x) #x
#define STR(x) STR2(x)
// v[120:123] = A (bf16 ones), v[124:127] = B, v[128:143] = fast accumulator, v[144:159] = reference accumulator
#define LOADB "v_mov_b32 v124, %1\nv_add_u32 v125, 0x00010001, v124\nv_add_u32 v126, 0x00020002, v124\nv_add_u32 v127, 0x00030003, v124\ns_nop 4\n"
#define POISON "v_mov_b32 v124, 0x7fc07fc0\nv_mov_b32 v125, 0x7fc07fc0\nv_mov_b32 v126, 0x7fc07fc0\nv_mov_b32 v127, 0x7fc07fc0\n"
#define ZERO(b) "v_mov_b32 v" STR(b) ", 0\n"
#define CLOB "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143", \
             "v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","memory"

template <int K, int STRESS>
__global__ void __launch_bounds__(512, 1) k_war(unsigned* bad, int iters) {
    __shared__ float pad[20480];                                            // 80 KiB: one block per CU -> exactly two waves per SIMD
    if (threadIdx.x == 0) pad[0] = 0.f;
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {               // waves 4-7: the partner
        if (STRESS) {
            floatx16 a0, a1;
            for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 z;
            for (int i = 0; i < 8; ++i) z[i] = (__bf16)1.0f;
            for (int it = 0; it < iters * 6; ++it) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, z, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, z, a1, 0, 0, 0);
            }
            asm volatile("" :: "v"(a0), "v"(a1));
        }
        return;
    }
    unsigned nbad = 0;
    const unsigned lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const unsigned b = 0x3f803f80u + ((lane * 7u + (unsigned)it * 13u) & 0x3fu) * 0x00010001u;      // bf16 pairs near 1.0
        unsigned diff;
        asm volatile(
            "v_mov_b32 v120, 0x3f803f80\nv_mov_b32 v121, 0x3f803f80\nv_mov_b32 v122, 0x3f803f80\nv_mov_b32 v123, 0x3f803f80\n"
            LOADB
            "v_mfma_f32_32x32x16_bf16 v[128:143], v[120:123], v[124:127], 0\n"
#if 1
            ".rept " STR(KK) "\ns_nop 0\n.endr\n"
#endif
            POISON
            "s_nop 15\ns_nop 15\ns_nop 15\n"
            LOADB
            "v_mfma_f32_32x32x16_bf16 v[144:159], v[120:123], v[124:127], 0\n"
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
            POISON
            "s_nop 15\ns_nop 15\n"
            "v_xor_b32 %0, v128, v144\n"
            "v_xor_b32 v129, v129, v145\nv_or_b32 %0, %0, v129\nv_xor_b32 v130, v130, v146\nv_or_b32 %0, %0, v130\nv_xor_b32 v131, v131, v147\nv_or_b32 %0, %0, v131\n"
            "v_xor_b32 v132, v132, v148\nv_or_b32 %0, %0, v132\nv_xor_b32 v133, v133, v149\nv_or_b32 %0, %0, v133\nv_xor_b32 v134, v134, v150\nv_or_b32 %0, %0, v134\n"
            "v_xor_b32 v135, v135, v151\nv_or_b32 %0, %0, v135\nv_xor_b32 v136, v136, v152\nv_or_b32 %0, %0, v136\nv_xor_b32 v137, v137, v153\nv_or_b32 %0, %0, v137\n"
            "v_xor_b32 v138, v138, v154\nv_or_b32 %0, %0, v138\nv_xor_b32 v139, v139, v155\nv_or_b32 %0, %0, v139\nv_xor_b32 v140, v140, v156\nv_or_b32 %0, %0, v140\n"
            "v_xor_b32 v141, v141, v157\nv_or_b32 %0, %0, v141\nv_xor_b32 v142, v142, v158\nv_or_b32 %0, %0, v142\nv_xor_b32 v143, v143, v159\nv_or_b32 %0, %0, v143\n"
            : "=v"(diff) : "v"(b) : CLOB);
        nbad += diff != 0;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int K, int STRESS> static void run(unsigned* bad, int iters) {
    (void)hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((k_war<K, STRESS>), dim3(256), dim3(512), 0, 0, bad, iters);
    unsigned h = 0;
    (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("slots between the MFMA and the overwrite of its B operand: %d, partner wave %-22s lanes x iterations with a wrong product: %u of %llu\n", KK, STRESS ? "streams MFMAs:" : "idle:", h,
           256ull * 256 * iters);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned* bad;
    (void)hipMalloc(&bad, 4);
    run<0, 0>(bad, iters);
    run<0, 1>(bad, iters);
    return 0;
}
