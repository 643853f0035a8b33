// tools/ubench/lds_trans_hazard.hip -- stand-alone probe for the run-to-run differences of k_shade_mfma (profiles/r05/m_* and r_*).
//
// r05 evidence from the full kernel: arrangements of the SiLU "heads" that differ only in WHERE the group's LDS weight reads sit relative to its burst of
// v_exp_f32 (in front of it: -DSM_W_EARLY=1) or in how many pairs a group holds (-DSM_QB={0,2,4,7,10,13,16}) give renders that differ from run to run on
// groups of rays, at EVERY transcendental -> use distance the post-pass enforces for the failing grouping up to 6 (and none at 7+), while the shipped
// arrangement is clean at every distance including the compiler's own.  So the distance rule is not the whole story; the suspects are the units that
// write VGPRs asynchronously beside the VALU: LDS return data, the quarter-rate transcendental unit, the matrix unit.
//
// This probe runs, with two waves per SIMD (80 KiB of LDS per 256-thread block), the heads' instruction pattern in ONE inline-asm block per variant:
//     [ds_read_b128 x R]  v_exp_f32 x 6  [ds_read_b128 x R]  v_pk_add x 3  v_rcp_f32 x 6  v_pk_mul x 3  s_waitcnt lgkmcnt(0)  v_pk_fma x 3 (+ optional MFMA)
// on per-lane inputs, and the SAME arithmetic again with `s_nop 7` behind every instruction; both are deterministic functions of the inputs, so any
// lane where they differ bit for bit is a hardware-level hazard.  Variants: LDS reads in front of / behind the exp burst, 0..3 wait states between the
// last transcendental and its first reader, with / without a matrix instruction in the group.
//     hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_trans_hazard.hip -o gpurun_out/lds_trans_hazard && gpurun_out/lds_trans_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float floatx16 __attribute__((ext_vector_type(16)));
#define STR2(x) #x
#define STR(x) STR2(x)
// v[100:105] inputs, v[106:111] exp, v[112:123] weights (LDS), v[124:125] accumulator; %0 = LDS address VGPR
#define EXPS "v_exp_f32 v106, v100\nv_exp_f32 v107, v101\nv_exp_f32 v108, v102\nv_exp_f32 v109, v103\nv_exp_f32 v110, v104\nv_exp_f32 v111, v105\n"
#define LDSR "ds_read_b128 v[112:115], %2\nds_read_b128 v[116:119], %2 offset:16\nds_read_b128 v[120:123], %2 offset:32\n"
#define ADDS "v_pk_add_f32 v[106:107], v[106:107], 1.0 op_sel_hi:[1,0]\nv_pk_add_f32 v[108:109], v[108:109], 1.0 op_sel_hi:[1,0]\nv_pk_add_f32 v[110:111], v[110:111], 1.0 op_sel_hi:[1,0]\n"
#define RCPS "v_rcp_f32 v106, v106\nv_rcp_f32 v107, v107\nv_rcp_f32 v108, v108\nv_rcp_f32 v109, v109\nv_rcp_f32 v110, v110\nv_rcp_f32 v111, v111\n"
#define MULS "v_pk_mul_f32 v[106:107], v[100:101], v[106:107]\nv_pk_mul_f32 v[108:109], v[102:103], v[108:109]\nv_pk_mul_f32 v[110:111], v[104:105], v[110:111]\n"
#define FMAS "v_pk_fma_f32 v[124:125], v[112:113], v[106:107], v[124:125]\nv_pk_fma_f32 v[124:125], v[116:117], v[108:109], v[124:125]\nv_pk_fma_f32 v[124:125], v[120:121], v[110:111], v[124:125]\n"
#define NOPS(k) "s_nop " STR(k) "\n"
#define SETUP "v_mov_b32 v100, %3\nv_add_f32 v101, 0.25, v100\nv_add_f32 v102, 0.5, v100\nv_add_f32 v103, -0.75, v100\nv_add_f32 v104, 1.5, v100\nv_add_f32 v105, -2.25, v100\nv_mov_b32 v124, 0\nv_mov_b32 v125, 0\ns_nop 7\n"
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","memory"

template <int ORDER, int K, int MF>
__global__ void __launch_bounds__(256, 2) k_probe(unsigned* bad, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[20480];              // 80 KiB: two blocks per CU -> two waves per SIMD
    for (int i = threadIdx.x; i < 20480; i += 256) lds[i] = 0.001f * (float)((i * 37) % 1000) - 0.4f;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    unsigned nbad = 0;
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 za;
    for (int i = 0; i < 8; ++i) za[i] = (__bf16)0.0f;
    for (int it = 0; it < iters; ++it) {
        const float x = in[(blockIdx.x * 256 + threadIdx.x + it * 7919) & 0xffff];
        const unsigned addr = (((lane >> 5) * 3 + (unsigned)(it & 15) * 6) * 16) & 0xffff;     // broadcast within a lane half, like the output weights
        float f0, f1, s0, s1;
        if (MF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(za, za, acc, 0, 0, 0);
        if (ORDER == 0)       // LDS reads behind the exp burst (the shipped arrangement)
            asm volatile(SETUP EXPS LDSR ADDS RCPS NOPS(0) MULS "s_waitcnt lgkmcnt(0)\n" FMAS "s_nop 7\nv_mov_b32 %0, v124\nv_mov_b32 %1, v125\n"
                         : "=v"(f0), "=v"(f1) : "v"(addr), "v"(x) : CLOB);
        else                  // LDS reads in front of it (-DSM_W_EARLY=1)
            asm volatile(SETUP LDSR EXPS ADDS RCPS NOPS(0) MULS "s_waitcnt lgkmcnt(0)\n" FMAS "s_nop 7\nv_mov_b32 %0, v124\nv_mov_b32 %1, v125\n"
                         : "=v"(f0), "=v"(f1) : "v"(addr), "v"(x) : CLOB);
        // the same arithmetic, every producer far from its consumer
        asm volatile(SETUP LDSR "s_waitcnt lgkmcnt(0)\ns_nop 7\n"
                     "v_exp_f32 v106, v100\ns_nop 7\nv_exp_f32 v107, v101\ns_nop 7\nv_exp_f32 v108, v102\ns_nop 7\nv_exp_f32 v109, v103\ns_nop 7\nv_exp_f32 v110, v104\ns_nop 7\nv_exp_f32 v111, v105\ns_nop 7\ns_nop 7\n"
                     ADDS "s_nop 7\n"
                     "v_rcp_f32 v106, v106\ns_nop 7\nv_rcp_f32 v107, v107\ns_nop 7\nv_rcp_f32 v108, v108\ns_nop 7\nv_rcp_f32 v109, v109\ns_nop 7\nv_rcp_f32 v110, v110\ns_nop 7\nv_rcp_f32 v111, v111\ns_nop 7\ns_nop 7\n"
                     MULS "s_nop 7\n"
                     "v_pk_fma_f32 v[124:125], v[112:113], v[106:107], v[124:125]\ns_nop 7\nv_pk_fma_f32 v[124:125], v[116:117], v[108:109], v[124:125]\ns_nop 7\nv_pk_fma_f32 v[124:125], v[120:121], v[110:111], v[124:125]\n"
                     "s_nop 7\nv_mov_b32 %0, v124\nv_mov_b32 %1, v125\n"
                     : "=v"(s0), "=v"(s1) : "v"(addr), "v"(x) : CLOB);
        nbad += (__float_as_uint(f0) != __float_as_uint(s0)) || (__float_as_uint(f1) != __float_as_uint(s1));
    }
    if (MF) asm volatile("" :: "v"(acc));
    if (nbad) atomicAdd(bad, nbad);
}

template <int ORDER, int K, int MF> static void run(const char* what, unsigned* bad, const float* in, int iters) {
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((k_probe<ORDER, K, MF>), dim3(512), dim3(256), 0, 0, bad, in, iters);
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("%-72s lanes x iterations that differ from the padded sequence: %u of %llu\n", what, h, 512ull * 256 * iters);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned* bad; float* in;
    hipMalloc(&bad, 4); hipMalloc(&in, 65536 * 4);
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = -4.0f + 8.0f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    run<0, 0, 0>("LDS reads behind the exp burst", bad, in, iters);
    run<1, 0, 0>("LDS reads in front of the exp burst", bad, in, iters);
    run<0, 0, 1>("LDS reads behind the exp burst, one MFMA per group", bad, in, iters);
    run<1, 0, 1>("LDS reads in front of the exp burst, one MFMA per group", bad, in, iters);
    return 0;
}
