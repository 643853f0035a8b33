#!/usr/bin/env python
"""tools/ubench/gen_trans_use_hazard.py > tools/ubench/trans_use_hazard.hip

Round 3 wait-state micro-benchmark, second subject: a transcendental VALU instruction (v_exp_f32 / v_rcp_f32, quarter rate: 16 lanes per pass) whose
result is read k issue slots later by an ordinary VALU instruction (v_pk_add_f32 / v_pk_mul_f32 / v_add_f32).  The toolchain's hazard recogniser
(VALUTransUseHazard) pads this pair to ONE wait state; the bisect of the irreproducible shading kernel (profiles/r03/hazard.txt) ends on exactly
such pairs -- `v_exp_f32; v_exp_f32; s_nop 0; v_pk_add_f32` -- and on their placement modulo 64 bytes.

Waves 0-3 of a 512-thread block run, inside ONE asm block (nothing is scheduled or padded by the compiler):
    x = small integers per lane and iteration -> poison the destinations -> [A x s_nop 0 in front of the loop: placement phase] ->
    v_exp_f32 e0, x0 ; v_exp_f32 e1, x1 ; k x s_nop 0 ; CONSUMER(e0, e1) ; accumulate ; loop
and check sum_it (2^x0 + 1) + (2^x1 + 1) exactly (powers of two: exact).  Waves 4-7 -- the co-resident wave of every SIMD -- run a stream of one
instruction class (none / v_exp / v_rcp / MFMA / packed FMA / the mix of the shading kernel): the trans unit is shared by the SIMD's waves.
Printed: wrong lanes per (consumer, partner stream, k = 0..3) for the 16 placement phases."""

POISON = "v_mov_b32 v202, v187\\nv_mov_b32 v203, v187\\n"
VALUES = ("s_mul_i32 s41, s40, 5\\nv_add_u32 v191, s41, v190\\nv_and_b32 v191, 7, v191\\nv_cvt_f32_u32 v191, v191\\n"
          "s_mul_i32 s41, s40, 3\\nv_add_u32 v192, s41, v190\\nv_and_b32 v192, 7, v192\\nv_cvt_f32_u32 v192, v192\\n")
CONSUMERS = {
    # name: (trans pair, consumer; result in v[204:205])
    "exp_pkadd": ("v_exp_f32 v202, v191\\nv_exp_f32 v203, v192\\n", "v_pk_add_f32 v[204:205], v[202:203], 1.0 op_sel_hi:[1,0]\\n"),
    "exp_add": ("v_exp_f32 v202, v191\\nv_exp_f32 v203, v192\\n", "v_add_f32 v205, 1.0, v203\\nv_add_f32 v204, 1.0, v202\\n"),
    "rcp_pkmul": ("v_rcp_f32 v202, v193\\nv_rcp_f32 v203, v194\\n", "v_pk_mul_f32 v[204:205], v[202:203], v[196:197]\\n"),
}
STRESS = {
    "none": "",
    "exp": "".join(f"v_exp_f32 v{100 + i}, v{100 + i}\\n" for i in range(8)) * 8,
    "rcp": "".join(f"v_rcp_f32 v{100 + i}, v{100 + i}\\n" for i in range(8)) * 8,
    "mfma": "v_mfma_f32_32x32x16_bf16 v[132:147], v[124:127], v[124:127], v[132:147]\\nv_mfma_f32_32x32x16_bf16 v[148:163], v[124:127], v[124:127], v[148:163]\\n" * 16,
    "pkfma": "".join(f"v_pk_fma_f32 v[{100 + 2 * i}:{101 + 2 * i}], v[{100 + 2 * i}:{101 + 2 * i}], v[120:121], v[122:123]\\n" for i in range(8)) * 8,
    "mix": ("v_exp_f32 v100, v100\\nv_rcp_f32 v101, v101\\nv_pk_mul_f32 v[102:103], v[102:103], v[120:121]\\nv_pk_fma_f32 v[104:105], v[104:105], v[120:121], v[122:123]\\n"
            "v_mfma_f32_32x32x16_bf16 v[132:147], v[124:127], v[124:127], v[132:147]\\nv_exp_f32 v106, v106\\nv_permlane32_swap_b32 v109, v110\\n"
            "v_rcp_f32 v111, v111\\nds_read_b128 v[112:115], v119\\nv_mfma_f32_32x32x16_bf16 v[148:163], v[124:127], v[124:127], v[148:163]\\n") * 6,
}
CLOBBERS = ", ".join(f'"v{r}"' for r in range(180, 212)) + ', "s40", "s41", "vcc", "scc", "memory"'
STRESS_CLOBBERS = ", ".join(f'"v{r}"' for r in range(100, 164)) + ', "s40", "scc", "memory"'


def kernel(name, cons, k, phase, stress):
    trans, use = CONSUMERS[cons]
    # x for rcp: v193 = 2^x0, v194 = 2^x1 (exact reciprocals), v196 / v197 = the same (so the product is exactly 1)
    pre = ("v_mbcnt_lo_u32_b32 v188, -1, 0\\nv_mbcnt_hi_u32_b32 v188, -1, v188\\nv_mul_u32_u24 v190, 3, v188\\nv_mov_b32 v187, 0x7fc00000\\n"
           "v_mov_b32 v208, 0\\nv_mov_b32 v209, 0\\ns_mov_b32 s40, %2\\ns_nop 4\\n" + "s_nop 0\\n" * phase)
    loop = ("L_%=:\\n" + VALUES + "v_exp_f32 v193, v191\\nv_exp_f32 v194, v192\\ns_nop 4\\nv_mov_b32 v196, v193\\nv_mov_b32 v197, v194\\n" + POISON + "s_nop 4\\n"
            + trans + "s_nop 0\\n" * k + use + "s_nop 4\\nv_add_f32 v208, v208, v204\\nv_add_f32 v209, v209, v205\\n"
            "s_sub_u32 s40, s40, 1\\ns_cmp_lg_u32 s40, 0\\ns_cbranch_scc1 L_%=\\n" "s_nop 4\\nv_mov_b32 %0, v208\\nv_mov_b32 %1, v209\\n")
    sbody = ("v_mov_b32 v120, 1.0\\nv_mov_b32 v121, 0\\nv_mov_b32 v122, 0\\nv_mov_b32 v123, 0\\nv_mov_b32 v119, 0\\n"
             + "".join(f"v_mov_b32 v{r}, 1.0\\n" for r in range(100, 119)) + "".join(f"v_mov_b32 v{r}, 0\\n" for r in range(124, 164))
             + "s_mul_i32 s40, %0, 4\\nS_%=:\\n" + STRESS[stress] + "s_sub_u32 s40, s40, 1\\ns_cmp_lg_u32 s40, 0\\ns_cbranch_scc1 S_%=\\n")
    mode = {"exp_pkadd": 0, "exp_add": 0, "rcp_pkmul": 1}[cons]
    return (f"__global__ void __launch_bounds__(512) {name}(unsigned* bad, int iters) {{\n    __shared__ float pad[2048];\n"
            f"    if (threadIdx.x == 0) pad[0] = 0.f;\n    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {{\n"
            + (f"        asm volatile(\"{sbody}\"\n            :: \"s\"(iters) : {STRESS_CLOBBERS});\n" if stress != "none" else "")
            + f"        return;\n    }}\n    float o0, o1;\n"
            f"    asm volatile(\"{pre + loop}\"\n        : \"=v\"(o0), \"=v\"(o1) : \"s\"(iters) : {CLOBBERS});\n    tu_check(o0, o1, bad, iters, {mode});\n}}\n")


def main():
    print("// tools/ubench/trans_use_hazard.hip -- GENERATED by tools/ubench/gen_trans_use_hazard.py (see its docstring); do not edit by hand.")
    print("// hipcc --offload-arch=gfx950 -O2 tools/ubench/trans_use_hazard.hip -o .variants/trans_use_hazard && .variants/trans_use_hazard")
    print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n")
    print("__device__ void tu_check(float o0, float o1, unsigned* bad, int iters, int mode) {\n"
          "    const int lane = threadIdx.x & 63;\n    float e0 = 0.f, e1 = 0.f;\n"
          "    for (int it = iters; it >= 1; --it) {\n"
          "        const unsigned x0 = (unsigned)(it * 5 + lane * 3) & 7u, x1 = (unsigned)(it * 3 + lane * 3) & 7u;\n"
          "        if (mode == 0) { e0 += (float)(1u << x0) + 1.f; e1 += (float)(1u << x1) + 1.f; } else { e0 += 1.f; e1 += 1.f; }\n    }\n"
          "    if (!(o0 == e0) || !(o1 == e1)) atomicAdd(bad, 1u);\n}\n")
    rows = []
    for cons in CONSUMERS:
        for st in STRESS:
            for k in range(0, 4):
                names = []
                for ph in range(16):
                    n = f"tu_{cons}_{st}_{k}_{ph}"
                    print(kernel(n, cons, k, ph, st))
                    names.append(n)
                rows.append((f"{cons:10s} k = {k}, partner wave: {st}", names))
    print("typedef void (*tu_fn)(unsigned*, int);\nstruct Row { const char* what; tu_fn fn[16]; };\nstatic const Row rows[] = {")
    for what, names in rows:
        print(f'    {{"{what}", {{{", ".join(names)}}}}},')
    print("};\n")
    print("int main(int argc, char** argv) {\n"
          "    const int iters = argc > 1 ? atoi(argv[1]) : 2000;\n    unsigned* bad; (void)hipMalloc(&bad, 4);\n"
          "    printf(\"wrong lanes (of 65536 test lanes) per placement phase 0..15 (4-byte steps of the loop's start address); %d iterations per lane\\n\", iters);\n"
          "    for (const Row& r : rows) {\n        printf(\"%-46s\", r.what);\n"
          "        for (int p = 0; p < 16; ++p) {\n"
          "            (void)hipMemset(bad, 0, 4);\n            hipLaunchKernelGGL(r.fn[p], dim3(256), dim3(512), 0, 0, bad, iters);\n"
          "            unsigned h = 0; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);\n"
          "            if (hipGetLastError() != hipSuccess) { printf(\" ERR\"); continue; }\n            printf(\" %5u\", h);\n        }\n"
          "        printf(\"\\n\");\n    }\n    return 0;\n}")


if __name__ == "__main__":
    main()
