// ssdnerf_amd/csrc/shade_mfma.hip -- stage B of the fused renderer (the default shading kernel).
//
// Same contract as k_shade_queue (render_queue.hip): persistent waves shade the per-scene hit queues written by stage A
// (k_ray_cull + k_survivor_march).  What differs, all of it driven by the r01 / r02 profiles (k_shade_queue was ISSUE-bound: ~5600 VALU
// instructions per 64-sample iteration at ~50 % lane utilisation):
//
// 1. The two wide layers of the tiny MLP run on the bf16 matrix cores in fp32-class arithmetic:
//        h      = W1 . [f ; 1]        64 x (18+1)
//        h_col  = h + Wd' . SH'(d)    64 x 16, accumulated IN PLACE on top of h
//    every fp32 operand is split exactly into three bf16 terms (x = hi + mid + lo, 8 + 8 + 8 significand bits) and the six products whose
//    weight is >= 2^-16 accumulate in fp32 (dropped terms <= 2^-24 relative: the accuracy class of the fp32 chain).  v_mfma_f32_32x32x16_bf16
//    runs beside the VALU stream, which the f32-input MFMA does not (it executes on the VALU's own FMA lanes: tools/ubench/mfma_valu_overlap.hip,
//    2 MFMA + 32 v_fma = 126 ns = 59 + 70 with f32 MFMA, 77 ns with bf16 MFMA; the f32-MFMA forms of r01, 9.4-9.7 ms, are gone).
//    One lane = one sample = one MFMA "column": lane l supplies the B operand of column l with ONE v_permlane32_swap per packed feature
//    pair for both 32-sample tiles, and after a second swap+add every lane ends up with the four outputs (sigma, r, g, b) of ITS OWN sample.
//    Weights live in registers as pre-split A operands (loaded once per wave).  The bias b1 rides as a constant-1 input row; the direction
//    layer's bias is folded into its first column -- SH_0 is the constant 0.2820948, so Wd'[:,0] = fl32(Wd[:,0] * SH_0 + bd) (one rounding,
//    computed in fp64 in the prologue) against the input 1.0 replaces (Wd[:,0], bd) against (SH_0, 1): six MFMAs per tile and 24 registers
//    less, and one rounding instead of two in that term.
//    The 64->{1,3} output layer and the 128 SiLUs per sample stay on the VALU (4 outputs cannot fill an MFMA tile); output weights sit in
//    1 KiB of LDS and are read as broadcast ds_read_b128.
//
// 2. A lane never searches more than SM_SEARCH_PROBES empty voxels for its next sample.  A ray that needs a longer search
//    (typically: it left the object and must cross the rest of the box) is parked - state and all - in a wave-local
//    LDS pool; when 64 of them have collected (or nothing else is left) the wave runs a MARCH PASS in which every lane
//    marches one parked ray to its next hit (-> "ready" pool, picked up by the next refill) or to the end of the box
//    (-> finished).  Shading iterations therefore stay full, and marching runs at full lane utilisation too.
//
// 3. Rays are prepared 64 at a time by the whole wave (queue entry -> ray from the arrays or from the view's camera (RaySrc) -> near/far, tail
//    bound, first sample -> LDS stage); idle lanes refill from LDS by ballot/mbcnt rank.
//
// Results are bit-identical in the integer outputs and within fp32 rounding of k_shade_queue for the floats (the MFMA
// accumulates the same products in a different, fixed order); tests/test_render_gpu.py checks both against the oracle.
#include "decode_core.h"
#include <type_traits>
#include <utility>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// SiLU of two values at once: the multiply / add / multiply around the two transcendentals are packed fp32 ops (v_pk_mul_f32 / v_pk_add_f32
// process two values per instruction at the rate of a plain op: tools/ubench/mfma_valu_2wave.hip, 128 v_pk_fma = 278 ns = 128 v_fma).
SSD_DEV floatx2 sm_silu2(floatx2 h) {
    const floatx2 a = h * floatx2{-1.4426950408889634f, -1.4426950408889634f};
    floatx2 e;
    e.x = __builtin_amdgcn_exp2f(a.x);
    e.y = __builtin_amdgcn_exp2f(a.y);
    const floatx2 d = e + floatx2{1.0f, 1.0f};
    floatx2 rcp;
    rcp.x = __builtin_amdgcn_rcpf(d.x);
    rcp.y = __builtin_amdgcn_rcpf(d.y);
    return h * rcp;
}
// (r03, measured and dropped: the heads' packed fp32 ops pinned as inline assembly.  Written as vector arithmetic, ROCm 7.2's backend UN-packs the packed
// ops it finds in the shadow of an MFMA into two plain ones -- plain VALU ops execute beside the matrix pipe, packed ones do not: 62 of the 467 VALU
// instructions of the MFMA block are such halves.  Forcing them to stay packed, i.e. fewer issue slots, was 7 % SLOWER: profiles/r03/h_shade_valu_diet.txt.)
// SM_UNPACK (r05 experiment): 1 = the heads' packed fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma) as two plain instructions each -- packed fp32 ops beside matrix
// instructions (this wave's or, with two waves per SIMD, the other's) cost more than two plain ones on gfx950 (MI355X_MICROARCH.md: + 22 cycles per v_pk_fma against
// two v_fma beside MFMAs); the compiler already unpacks the ones it sees in an MFMA's shadow.  Inline asm pins the plain forms (the SLP vectoriser would re-pack them).
#ifndef SM_UNPACK
#define SM_UNPACK 0
#endif
SSD_DEV float sm_plain_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
SSD_DEV float sm_plain_mul(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
SSD_DEV float sm_plain_add1(float a) { float r; asm("v_add_f32 %0, 1.0, %1" : "=v"(r) : "v"(a)); return r; }
SSD_DEV floatx2 sm_fma2(floatx2 w, floatx2 v, floatx2 acc) {
#if SM_UNPACK
    return floatx2{sm_plain_fma(w.x, v.x, acc.x), sm_plain_fma(w.y, v.y, acc.y)};
#else
    return __builtin_elementwise_fma(w, v, acc);
#endif
}

static constexpr unsigned SM_TPB = 256;
#ifndef SM_SLICE_RAYS
#define SM_SLICE_RAYS 64                           // r02 A/B (bench scene, shade kernel): 512: 7.39, 256: 6.75, 128: 6.68, 64: 6.56 ms -- the kernel ENDS when the last
                                                   // wave has consumed its last ticket and drained; a 512-ray ticket is ~2 ms of work for a wave
#endif
static constexpr unsigned SM_SLICE = SM_SLICE_RAYS;   // hit-queue entries per ticket
#ifndef SM_PROBES
#define SM_PROBES 2                                // r02 A/B (bench scene, shade kernel): 1: 7.58, 2: 7.29, 3: 7.51, 4: 7.58, 8: 8.31 ms
#endif
static constexpr unsigned SM_SEARCH_PROBES = SM_PROBES;   // in-lane search budget after each sample
// Kernel forms (template parameter MODE):
//   0: any power-of-two grid, any plane size, per-scene cone angle.
//   1: the hot-path geometry as compile-time constants -- 64^3 grid, 128 x 128 planes, bound 1, 256 steps (configs/paper_cfgs/ssdnerf_*.py) --
//      so that march / plane constants are immediates instead of ~25 SGPRs (the generic form spills 81 SGPRs to VGPR lanes and pays a
//      v_readlane per use, ~130 in the composite / search section alone; this form still spills 44: pointers, camera source, loop state).
//   2: MODE 1 with dt_gamma == 0 (the uncond render of cached triplanes): the march step is a constant.
// The MLP schedule is the tile-interleaved one (r01 "variant 6"): both 32-sample tiles hold their accumulators and every MFMA group is followed, in
// program order, by SiLU pairs of the other tile, so a wave overlaps its own matrix-pipe time (7.72 vs 7.88 ms for tile-after-tile, r02 A/B).
struct SmGeo {
    static constexpr int WPS = 2;
    static constexpr unsigned POOL = 128;                       // entries per wave-local pool (two pools per wave)
    static constexpr unsigned MARCH_W = POOL / 2;               // rays advanced by one march pass
    static constexpr unsigned STAGE = 64;                       // prepared rays per wave
    static constexpr unsigned SHF = 1536;                       // floats of per-ray SH operands per wave (three bf16 terms x 64 rays x 16)
    static constexpr unsigned LDS_FLOATS = 512 + (SM_TPB / 64) * (2 * POOL * 8 + SHF + STAGE * 16);
};

typedef __bf16 sm_bf16x8 __attribute__((ext_vector_type(8)));
// x == hi + mid + lo exactly, each term a bf16 value (returned as fp32 bit patterns whose low 16 bits are zero)
SSD_DEV void sm_split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hi);                  // exact: the low 16 significand bits
    mid = __float_as_uint(r1) & 0xffff0000u;
    lo = __float_as_uint(r1 - __uint_as_float(mid));           // exact, at most 8 significant bits
}
SSD_DEV uint32_t sm_pack2(uint32_t even, uint32_t odd) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }   // {bf16(even), bf16(odd)}
SSD_DEV sm_bf16x8 sm_op(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint4 u = make_uint4(a, b, c, d);
    return *reinterpret_cast<const sm_bf16x8*>(&u);
}
SSD_DEV void sm_swap_u(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
template <class F, int... I> SSD_DEV void sm_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> SSD_DEV void sm_static_for(F&& f) { sm_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }
#ifndef SM_HEAD_GROUP
#define SM_HEAD_GROUP 4                            // SiLU pairs per stage-ordered group of the colour head that runs without MFMAs beside it (16 % SM_HEAD_GROUP == 0)
#endif
#ifndef SM_GATHER_BY_PLANE
#define SM_GATHER_BY_PLANE 0                       // 1: gather one plane at a time (24 texel registers in flight instead of 72; three dependent rounds, the r01-r04 form);
                                                   // 0 (r05): all twelve texels in ONE round -- 5.05 -> 4.94 ms with two waves per SIMD, 6.72 -> 6.53 with one
#endif
// SM_EVENT_MIN (r05): lane bookkeeping -- the stores of a finished ray, the pool entry of a parked one, the refill of an idle lane and the SH
// operands of a fresh ray -- is ~300 VALU instructions plus ~200 scalar / branch instructions that the WHOLE wave issues whenever ONE lane needs
// them, and with 24 samples per hitting ray 2.6 lanes of 64 need them in every iteration (r04 census: 844 of the 1 219 VALU instructions per
// iteration are gather + split + MLP, the rest is this).  A lane that finishes or must park now just raises a flag and sits out; the wave runs
// the bookkeeping as ONE event when SM_EVENT_MIN lanes are waiting (or nothing is left to shade).  Per-ray arithmetic and its order are
// untouched -- every output is bit-identical for any value; 1 = an event in every iteration (the r01-r04 behaviour).
#ifndef SM_W_EARLY
#define SM_W_EARLY 1                               // heads: the group's output-weight reads in front of its transcendentals instead of behind them: their LDS latency runs under the
                                                   // v_exp burst (shade kernel 4.73 -> 4.65 ms, profiles/r06/h_ab_heads.txt; r05 had measured the same gain and could not ship it: this is
                                                   // one of the arrangements in which the crossed packed instructions of the blend failed loudly -- see "Run-to-run reproducibility" below)
#endif
#ifndef SM_SWAP_F32
#define SM_SWAP_F32 1                              // (r06) 1: the lane-half exchange on the fp32 features in front of the split (8 swaps) instead of on the split terms behind it (12):
                                                   // bit-identical, shade kernel 4.75 -> 4.72 ms (profiles/r06/f_ab_split_vs_packed.txt)
#endif
#ifndef SM_BLEND_VARIANT
#define SM_BLEND_VARIANT 0
#endif
#ifndef SM_BLEND_SCALAR
#define SM_BLEND_SCALAR 0
#endif
#ifndef SM_KEEP_ADDR
#define SM_KEEP_ADDR 0
#endif
#ifndef SM_GATHER_DRAIN
#define SM_GATHER_DRAIN 0                          // (r06 experiments) bit 0: s_waitcnt vmcnt(0) in front of the texel requests; bit 1: ... behind them, in front of the blend
#endif
#ifndef SM_NO_BOUNDARY_COUNT
#define SM_NO_BOUNDARY_COUNT 0
#endif
#ifndef SM_GATHER_FIRST
#define SM_GATHER_FIRST 1                          // (r05) texel requests in front of the search arithmetic, blend behind it
#endif
#ifndef SM_SKIP_UNROLL
#define SM_SKIP_UNROLL 0                           // (r05) sm_skip: this many predicated steps before the loop (0: the plain do-while)
#endif
#ifndef SM_EVENT_MIN
#define SM_EVENT_MIN 4                             // r05 A/B (bench scene, shade kernel, with SM_SEARCH_AHEAD and the one-round gather): 1: 5.06, 3: 4.94-4.97, 4: 4.94-4.96, 5: 4.97, 6: 5.01 ms
#endif

// Run-to-run reproducibility (r02 symptom; r06 cause).  With two waves per SIMD a render repeated on the same inputs differed on groups of neighbouring rays -- up to 16,
// one quarter-wave: lanes 48-63 -- by 1e-8 .. 5e-2, about one render in 3 000 in the shipped arrangement and hundreds to 250 000 rays per render in arrangements of the
// SiLU heads that differ from it only in grouping.  r02 blamed a VALU -> MFMA operand hazard, r03 the transcendental -> use pair (and padded it at build level), r05 the
// v_permlane32_swap -> MFMA pair (padded too); every "fix" was a change of timing.  r06 followed the error with per-ray hashes of every stage (SM_DEBUG_TRACE below,
// tools/trace_check.py): march parameters, texels, bilinear weights and the MLP (evaluated twice per sample, SM_DEBUG_DUAL) always agree between runs -- the FEATURES do
// not: the bilinear blend's packed chain  v_pk_mul_f32 r, a00, w op_sel_hi:[1,0] ; v_pk_fma_f32 r, a01, w, r op_sel:[0,1,0] ; ...  occasionally returns, in the low
// half of lanes 48-63, the chain WITHOUT one product term (SM_DEBUG_BLEND2, tools/blend_check.py: the same registers blended again, packed and plain).  The trigger is the
// packed fp32 instruction whose op_sel / op_sel_hi read across the halves of a VGPR pair (the compiler's broadcast of one weight): the same chain with the broadcast in a
// register pair (SM_BLEND_VARIANT 1) or as plain v_mul / v_fma (SM_BLEND_SCALAR) never fails, in any arrangement.  The fix is in the build, for the whole library:
// asm_postpass.unpack_cross_half splits every such instruction into two plain ones and the build refuses a code object that still holds one.  The r03 / r05 padding
// rules are off.  Stand-alone probes of the instruction (tools/ubench/pk_cross_hazard.hip, pk_cross_gather.hip) are clean: the failure needs this kernel around it.

// SM_ADDR32 (r05): the gather of decode_core.h with the texel address as ONE 32-bit byte offset from the scene's (wave-uniform) plane base, so that
// the loads take the base from SGPRs (saddr form) instead of a 64-bit add per corner; element-wise the same arithmetic, bit-identical features.
// The host refuses plane sets of 4 GiB and more per scene (sm_shade).
#ifndef SM_ADDR32
#define SM_ADDR32 1
#endif
template <typename PT> struct SmTexel;
template <> struct SmTexel<float> {
    static SSD_DEV void load6(const float* __restrict__ base, uint32_t byte_off, float v[6]) {
        const char* q = reinterpret_cast<const char*>(base) + byte_off;
        const float4 a = *reinterpret_cast<const float4*>(q);
        const float2 b = *reinterpret_cast<const float2*>(q + 16);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y;
    }
};
template <> struct SmTexel<__half> {
    static SSD_DEV void load6(const __half* __restrict__ base, uint32_t byte_off, float v[6]) {
        const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + byte_off);
        const __half2* h = reinterpret_cast<const __half2*>(&raw);
        const float2 a = __half22float2(h[0]), b = __half22float2(h[1]), c = __half22float2(h[2]);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
    }
};
// one plane: the four texels of the bilinear footprint (issue) and their blend (consume), separately, so that a caller can put work between them
template <typename PT> struct SmPlaneTap {
    float t00[6], t01[6], t10[6], t11[6];
    float w00, w01, w10, w11;
    SSD_DEV void issue(const PT* __restrict__ planes, const PlaneGeom& g, int p, float u, float v) {
        uint32_t x0, x1, y0, y1;
        float wx0, wx1, wy0, wy1;
        ssd_grid_coord(u, g.Wf, g.Wp, x0, x1, wx0, wx1);
        ssd_grid_coord(v, g.Hf, g.Hp, y0, y1, wy0, wy1);
        constexpr uint32_t TEXEL = 8u * (uint32_t)sizeof(PT);
        const uint32_t row0 = ((uint32_t)p * g.Hp + y0) * g.Wp, row1 = ((uint32_t)p * g.Hp + y1) * g.Wp;
        SmTexel<PT>::load6(planes, (row0 + x0) * TEXEL, t00);
        SmTexel<PT>::load6(planes, (row0 + x1) * TEXEL, t01);
        SmTexel<PT>::load6(planes, (row1 + x0) * TEXEL, t10);
        SmTexel<PT>::load6(planes, (row1 + x1) * TEXEL, t11);
        w00 = wx0 * wy0; w01 = wx1 * wy0; w10 = wx0 * wy1; w11 = wx1 * wy1;
#if SM_KEEP_ADDR
        o00 = (row0 + x0) * TEXEL; o01 = (row0 + x1) * TEXEL; o10 = (row1 + x0) * TEXEL; o11 = (row1 + x1) * TEXEL;
#endif
    }
#if SM_KEEP_ADDR
    uint32_t o00, o01, o10, o11;
    SSD_DEV void keep() const { asm volatile("" :: "v"(o00), "v"(o01), "v"(o10), "v"(o11)); }     // the offsets stay live: no load writes its data over its own address register
#endif
    SSD_DEV void blend(int p, float f[18]) const {             // decode_core.h ssd_gather18's chain, on channel pairs
#if SM_BLEND_SCALAR
        // (r06 experiment) the same chain per channel as plain v_mul / v_fma (pinned: the SLP vectoriser would re-pack them): element-wise the same arithmetic
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float r = sm_plain_mul(t00[c], w00);
            r = sm_plain_fma(t01[c], w01, r); r = sm_plain_fma(t10[c], w10, r); r = sm_plain_fma(t11[c], w11, r);
            f[c * 3 + p] = r;
        }
        return;
#endif
        typedef float ssd_f2 __attribute__((ext_vector_type(2)));
#if SM_BLEND_VARIANT == 1
        {   // (r06 experiment) packed, but the weights as explicit register pairs: no op_sel broadcast on the packed instructions
            ssd_f2 W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};
            asm volatile("" : "+v"(W00), "+v"(W01), "+v"(W10), "+v"(W11));
#pragma unroll
            for (int c = 0; c < 6; c += 2) {
                const ssd_f2 a00 = {t00[c], t00[c + 1]}, a01 = {t01[c], t01[c + 1]}, a10 = {t10[c], t10[c + 1]}, a11 = {t11[c], t11[c + 1]};
                ssd_f2 r = a00 * W00;
                r = __builtin_elementwise_fma(a01, W01, r); r = __builtin_elementwise_fma(a10, W10, r); r = __builtin_elementwise_fma(a11, W11, r);
                f[c * 3 + p] = r.x; f[(c + 1) * 3 + p] = r.y;
            }
            return;
        }
#elif SM_BLEND_VARIANT == 2
        {   // (r06 experiment) packed with op_sel, the three channel-pair chains advanced side by side: dependent packed instructions three slots apart
            ssd_f2 r[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) r[q] = ssd_f2{t00[2 * q], t00[2 * q + 1]} * ssd_f2{w00, w00};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 3; ++q) r[q] = __builtin_elementwise_fma(ssd_f2{t01[2 * q], t01[2 * q + 1]}, ssd_f2{w01, w01}, r[q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 3; ++q) r[q] = __builtin_elementwise_fma(ssd_f2{t10[2 * q], t10[2 * q + 1]}, ssd_f2{w10, w10}, r[q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 3; ++q) r[q] = __builtin_elementwise_fma(ssd_f2{t11[2 * q], t11[2 * q + 1]}, ssd_f2{w11, w11}, r[q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 3; ++q) { f[2 * q * 3 + p] = r[q].x; f[(2 * q + 1) * 3 + p] = r[q].y; }
            return;
        }
#endif
#pragma unroll
        for (int c = 0; c < 6; c += 2) {
            const ssd_f2 a00 = {t00[c], t00[c + 1]}, a01 = {t01[c], t01[c + 1]}, a10 = {t10[c], t10[c + 1]}, a11 = {t11[c], t11[c + 1]};
            ssd_f2 r = a00 * ssd_f2{w00, w00};
            r = __builtin_elementwise_fma(a01, ssd_f2{w01, w01}, r);
            r = __builtin_elementwise_fma(a10, ssd_f2{w10, w10}, r);
            r = __builtin_elementwise_fma(a11, ssd_f2{w11, w11}, r);
            f[c * 3 + p] = r.x;
            f[(c + 1) * 3 + p] = r.y;
        }
    }
};
template <typename PT, bool PLANE_BY_PLANE>
SSD_DEV void sm_gather18(const PT* __restrict__ planes, const PlaneGeom& g, float x, float y, float z, float f[18]) {
    const float us[3] = {x, x, y};
    const float vs[3] = {y, z, z};
    if (PLANE_BY_PLANE) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            SmPlaneTap<PT> tap;
            tap.issue(planes, g, p, us[p], vs[p]);
            tap.blend(p, f);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        SmPlaneTap<PT> tap[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) tap[p].issue(planes, g, p, us[p], vs[p]);
#pragma unroll
        for (int p = 0; p < 3; ++p) tap[p].blend(p, f);
    }
}

struct FastMarchB {
    float bound, dt_gamma, dt_min, dt_max, mip_bound, rb, half_H, two_rH, Hm1f;
    uint32_t H, log2H;
};
struct ShadeCfg {
    FastMarchB m;
    PlaneGeom g;
    float aabb[6];
    float min_near, T_thresh, bg, sat;
    uint32_t N, S, cap;
    uint64_t plane_stride;
    uint32_t bitfield_stride;
    const float* dt_gammas;
    uint8_t* image_u8;          // [S][N][3] or null: quantised copy of the image (k_quantize_u8's rounding), written with it
    const uint64_t* blocks64;   // [S][(H/4)^3] or null: the bitfield block-major, one u64 per 4^3 cells (k_bitfield_blocks64): the march pass skips empty blocks
    const uint32_t* order;      // [S][order_stride] or null: ticket i of a scene is slice order[i] of its queue (k_ticket_order, render_queue.hip)
    uint32_t order_stride;
#ifdef SM_DEBUG_TRACE
    uint32_t* dbg_trace;        // [S][N][8]: per-ray XOR hashes of what the kernel computed for the ray (tools/trace_check.py)
#endif
};

struct ProbeB { float x, y, z, dt; int nx, ny, nz; bool occ; };

// One probe of the reference's march (common.h ssd_probe, cascade 0, power-of-two grid), trimmed to the instructions that can change a result:
//   * position clamp as ONE v_med3_f32 (== min(hi, max(lo, v)) for the finite values that occur);
//   * the cell index needs only its upper clamp: v = p * rb + 1 > -1 always, and (int) truncates (-1, 0) to 0 like the clamp at 0 would;
//   * DTG0 (dt_gamma == 0, the uncond render): clamp(t * 0, dt_min, dt_max) == dt_min, so the step is a constant.
template <bool DTG0>
SSD_DEV float sm_dt(const FastMarchB& m, float t) { return DTG0 ? m.dt_min : ssd_clamp(t * m.dt_gamma, m.dt_min, m.dt_max); }

// `do { t += dt(t); } while (t < tt);` -- SM_RUN_TO_CLOSED (r05): for the constant step the closed form of common.h (ssd_run_to_const), bit-identical
#ifndef SM_RUN_TO_CLOSED
#define SM_RUN_TO_CLOSED 0                         // 0: loops (default); 1: closed form in the march pass's block exits (up to ~16 steps): no measurable change (4.78 ms either
                                                   // way); 2: in the cell skips too: +1.2 % (a cell is 1 - 4 steps).  Bit-identical in all three (profiles/r05/q_*)
#endif
template <bool DTG0>
SSD_DEV float sm_run_to(const FastMarchB& m, float t, float tt) {
    if (DTG0 && SM_RUN_TO_CLOSED) return ssd_run_to_const(m.dt_min, t, tt);
    do { t += sm_dt<DTG0>(m, t); } while (t < tt);
    return t;
}

template <bool DTG0>
SSD_DEV ProbeB sm_probe(const FastMarchB& m, const uint8_t* __restrict__ lin_bits, const RayGeom& r, float t) {
    ProbeB p;
    p.x = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dx, r.ox), -m.bound, m.bound);
    p.y = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dy, r.oy), -m.bound, m.bound);
    p.z = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dz, r.oz), -m.bound, m.bound);
    p.dt = sm_dt<DTG0>(m, t);
    p.nx = (int)fminf(ssd_fma(p.x, m.rb, 1.0f) * m.half_H, m.Hm1f);
    p.ny = (int)fminf(ssd_fma(p.y, m.rb, 1.0f) * m.half_H, m.Hm1f);
    p.nz = (int)fminf(ssd_fma(p.z, m.rb, 1.0f) * m.half_H, m.Hm1f);
    const uint32_t idx = ((((uint32_t)p.nz << m.log2H) + (uint32_t)p.ny) << m.log2H) + (uint32_t)p.nx;
    p.occ = (lin_bits[idx >> 3] >> (idx & 7u)) & 1u;
    return p;
}
template <bool DTG0>
SSD_DEV float sm_skip(const FastMarchB& m, const RayGeom& r, const ProbeB& p, float sgx, float sgy, float sgz, float t) {
    const float tx = ssd_fma(ssd_fma((float)p.nx + sgx, m.two_rH, -1.0f), m.mip_bound, -p.x) * r.rdx;
    const float ty = ssd_fma(ssd_fma((float)p.ny + sgy, m.two_rH, -1.0f), m.mip_bound, -p.y) * r.rdy;
    const float tz = ssd_fma(ssd_fma((float)p.nz + sgz, m.two_rH, -1.0f), m.mip_bound, -p.z) * r.rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    // `do t += dt(t); while (t < tt)` -- the same additions on the same values in the same order, but the first steps as selects (a cell is 2.3 minimum
    // steps wide, its diagonal 4: a divergent loop paid a branch round trip per step per wave); the loop itself stays for whatever is left
    if (DTG0 && SM_RUN_TO_CLOSED > 1) return ssd_run_to_const(m.dt_min, t, tt);      // (a cell is 1 - 4 steps: the loop is cheaper here, r05 A/B)
    t += sm_dt<DTG0>(m, t);
#pragma unroll
    for (int i = 0; i < SM_SKIP_UNROLL; ++i) t = t < tt ? t + sm_dt<DTG0>(m, t) : t;
    while (t < tt) t += sm_dt<DTG0>(m, t);
    return t;
}

// SM_SEARCH_AHEAD (r05).  The successor of a sample does not depend on the sample's MLP -- only on the march: t1 = t + dt is probed; if that cell is
// empty the DDA skip gives t2 (a function of the empty cell's INDEX, not of the loaded bit), then t3.  So both probe addresses and both skips are
// straight-line arithmetic on t, and both occupancy bytes can be requested BEFORE the sample's gather and MLP; after compositing, the lane only
// selects: the same far / cap / probe-budget tests in the same order on the same values as the loop they replace (bit-identical sample
// sequences), without its two dependent load rounds and its divergent control flow (r05 section timing: composite + search was 2.8 k of a lone
// wave's 14.3 k cycles per iteration).
#ifndef SM_SEARCH_AHEAD
#define SM_SEARCH_AHEAD 1
#endif
#ifndef SM_AHEAD_PROBES
#define SM_AHEAD_PROBES 2                          // (r06) probes of the straight-line successor search: 2 = t1, then t2 behind one DDA skip (the r05 form); 1 = t1 only -- a sample whose successor
                                                   // cell is empty parks at t1 and the march pass runs the skip it would have run here (same parameter sequence: bit-identical), ~90 VALU
                                                   // instructions and two divergent stepping loops less per iteration against more parked rays (profiles/r06/c_*)
#endif
#ifndef SM_MARCH_BLOCKS
#define SM_MARCH_BLOCKS 1                          // the march pass skips empty 4^3 / 2^3 blocks (r05 A/B below)
#endif
#ifndef SM_MARCH_TRIPS
#define SM_MARCH_TRIPS 0                           // probes per lane and march pass (0: unbounded).  r05 A/B: 3: 4.82, 4: 4.79, 6: 4.80, 8: 4.81, 12: 4.80, unbounded: 4.80 ms --
                                                   // a pass costs ~18 k cycles whatever its probe count (53 k passes of 6 trips against 47 k unbounded): off
#endif
#ifndef SM_DRAIN_MARCH
#define SM_DRAIN_MARCH 0                           // r05, measured and off: march parked rays as soon as a quarter of the lanes idle once the queue is used up -- 46 k -> 77 k
#endif                                             // march passes per launch, 4.78 -> 4.93 ms: a pass costs ~18 k cycles whatever its size
// sm_probe without the load: position, step, cell, and the cell's bit index
template <bool DTG0>
SSD_DEV ProbeB sm_probe_addr(const FastMarchB& m, const RayGeom& r, float t, uint32_t& idx) {
    ProbeB p;
    p.x = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dx, r.ox), -m.bound, m.bound);
    p.y = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dy, r.oy), -m.bound, m.bound);
    p.z = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dz, r.oz), -m.bound, m.bound);
    p.dt = sm_dt<DTG0>(m, t);
    p.nx = (int)fminf(ssd_fma(p.x, m.rb, 1.0f) * m.half_H, m.Hm1f);
    p.ny = (int)fminf(ssd_fma(p.y, m.rb, 1.0f) * m.half_H, m.Hm1f);
    p.nz = (int)fminf(ssd_fma(p.z, m.rb, 1.0f) * m.half_H, m.Hm1f);
    idx = ((((uint32_t)p.nz << m.log2H) + (uint32_t)p.ny) << m.log2H) + (uint32_t)p.nx;
    p.occ = false;
    return p;
}
// the position / step part only (what a sample's gather and composite need): the same expressions as sm_probe's
template <bool DTG0>
SSD_DEV void sm_sample_at(const FastMarchB& m, const RayGeom& r, float t, float& x, float& y, float& z, float& dt) {
    x = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dx, r.ox), -m.bound, m.bound);
    y = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dy, r.oy), -m.bound, m.bound);
    z = __builtin_amdgcn_fmed3f(ssd_fma(t, r.dz, r.oz), -m.bound, m.bound);
    dt = sm_dt<DTG0>(m, t);
}

// exit parameter of the SZ^3-cell block around probe p, `e` inside the exit face (render_queue.hip rq_block_exit, k_survivor_march's block skip)
template <int SZ>
SSD_DEV float sm_block_exit(const FastMarchB& m, const RayGeom& r, const ProbeB& p, float sgx, float sgy, float sgz, float ex, float ey, float ez, float t) {
    const float tx = (ssd_fma(ssd_fma((float)(p.nx & ~(SZ - 1)) + (float)SZ * sgx, m.two_rH, -1.0f), m.mip_bound, -p.x) - ex) * r.rdx;
    const float ty = (ssd_fma(ssd_fma((float)(p.ny & ~(SZ - 1)) + (float)SZ * sgy, m.two_rH, -1.0f), m.mip_bound, -p.y) - ey) * r.rdy;
    const float tz = (ssd_fma(ssd_fma((float)(p.nz & ~(SZ - 1)) + (float)SZ * sgz, m.two_rH, -1.0f), m.mip_bound, -p.z) - ez) * r.rdz;
    return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
}

// v_permlane32_swap: lanes 32..63 of `a` trade places with lanes 0..31 of `b`.
SSD_DEV void sm_swap(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// SM_PRESCALE (r03): silu(h) = h / (1 + 2^(-log2(e) h)).  The hidden layers' weights and biases are multiplied by -log2(e) when they are split into
// matrix-core operands (in fp64, one rounding to fp32 -- the same rounding class as the split itself), so that the accumulators hold
// h' = -log2(e) h and feed v_exp directly; the output layer's weights carry the inverse factor, w' = -ln(2) w, because w silu(h) = w' h' / (1 + 2^h').
// Saves the 64 packed multiplies per 64 samples that scaled the 128 pre-activations (the kernel is VALU-issue bound: DESIGN.md section 5).
SSD_DEV float sm_prescale(double w) { return (float)(w * -1.4426950408889634074); }
SSD_DEV float sm_outscale(float w) { return (float)((double)w * -0.69314718055994530942); }

template <typename PT, int MODE, int DIRP>
__global__ void __launch_bounds__(SM_TPB, SmGeo::WPS) k_shade_mfma(ShadeCfg c, RaySrc src, const PT* __restrict__ planes,
                                                           const float* __restrict__ P, const uint8_t* __restrict__ lin_bits,
                                                           const uint2* __restrict__ queue, uint32_t* __restrict__ queue_count,
                                                           float* __restrict__ image, float* __restrict__ depth, float* __restrict__ weights_sum,
                                                           int32_t* __restrict__ sample_counts, int32_t* __restrict__ overflow_flag) {
    // LDS: [0,1 KiB) output-layer weights per accumulator slot; then per wave two pools of SM_POOL x 8 dwords.
    // wout2: 32 entries x 8 floats; per (mt, pair p of adjacent accumulator registers, half): {ws_a, ws_b, wr_a, wr_b, wg_a, wg_b, wb_a, wb_b}
    // sh   : per wave 64 rays x 16 SH' values as three bf16 terms in MFMA B-operand form, [term][sample][k 0-7 | k 8-15] (SH'_0 = 1, see the header)
    // stage: per wave 64 PREPARED rays x 16 dwords {ray, t, far, dt | o | d | 1/d | sample xyz}: queue entries and ray geometry are fetched
    //        64 at a time by the whole wave (coalesced, full lane utilisation) instead of one lane at a time inside the divergent refill
    constexpr unsigned SM_POOL = SmGeo::POOL, SM_MARCH_W = SmGeo::MARCH_W, SM_STAGE = SmGeo::STAGE, SM_SHF = SmGeo::SHF;
    constexpr bool DTG0 = MODE == 2;
    if constexpr (MODE != 0) {       // the hot-path geometry as literals (the host dispatches here only when the arguments say exactly this)
        c.m.bound = 1.0f; c.m.mip_bound = 1.0f; c.m.rb = 1.0f;
        c.m.dt_min = 2.0f * SSD_SQRT3 / 256.0f; c.m.dt_max = 2.0f * SSD_SQRT3 / 64.0f;        // ssd_make_march_cfg's expressions, evaluated at compile time
        c.m.half_H = 32.0f; c.m.two_rH = 2.0f / 64.0f; c.m.Hm1f = 63.0f; c.m.H = 64u; c.m.log2H = 6u;
        c.g.Hp = c.g.Wp = 128u; c.g.Hf = c.g.Wf = 128.0f;
        c.aabb[0] = c.aabb[1] = c.aabb[2] = -1.0f; c.aabb[3] = c.aabb[4] = c.aabb[5] = 1.0f;
        c.cap = 256u;
        c.plane_stride = (uint64_t)3 * 128 * 128 * 8;
        c.bitfield_stride = 64u * 64u * 64u / 8u;
    }
    __shared__ __attribute__((aligned(16))) float lds[SmGeo::LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- output weights: slot (mt, reg, half) <-> hidden row mt*32 + (reg&3) + 8*(reg>>2) + 4*half (32x32 MFMA C/D layout) ----
    if (threadIdx.x < 32) {                  // 2 row tiles x 8 pairs x 2 halves (r02 fix: 64 threads here read rows 64..127, 1.7 KB past the block)
        const int mt = threadIdx.x >> 4, pr_ = (threadIdx.x >> 1) & 7, hf = threadIdx.x & 1;       // slot = (mt*8 + pair)*2 + half
        float v[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int reg = 2 * pr_ + e;
            const int row = mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hf;
            const float* rec = P + row * 24;                         // rec[19] = w_sigma, rec[20..22] = Wc[0..2][row]
            // (x -ln 2: the hidden layers arrive scaled by -log2(e), see SM_PRESCALE below)
            v[0 + e] = sm_outscale(rec[19]); v[2 + e] = sm_outscale(rec[20]); v[4 + e] = sm_outscale(rec[21]); v[6 + e] = sm_outscale(rec[22]);
        }
        reinterpret_cast<float4*>(lds)[threadIdx.x * 2 + 0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(lds)[threadIdx.x * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
    const float4* wout2 = reinterpret_cast<const float4*>(lds);     // index ((mt*8 + pair)*2 + half)*2 + {0: sigma|r, 1: g|b}
    uint32_t* pool_search = reinterpret_cast<uint32_t*>(lds + 512) + wave * 2 * SM_POOL * 8;
    uint32_t* pool_ready = pool_search + SM_POOL * 8;
    uint4* sh_lds = reinterpret_cast<uint4*>(lds + 512 + (SM_TPB / 64) * 2 * SM_POOL * 8 + wave * SM_SHF);   // [term][sample][2 x 16 B]
    float4* stage = reinterpret_cast<float4*>(lds + 512 + (SM_TPB / 64) * (2 * SM_POOL * 8 + SM_SHF) + wave * SM_STAGE * 16);   // [slot][4]

    // ---- persistent grid: every wave pulls SM_SLICE-ray slices (64) of a scene's hit queue with one atomic ticket per slice; the queue holds
    // the rays that may still take many samples first (render_queue.hip, k_survivor_march), so the launch ends on short rays.  Waves of
    // XCD x (workgroups are dispatched round-robin over the 8 XCDs, b % 8) start on scene x so that the scene's 1.5 MiB of
    // planes stay in that XCD's L2, and move on to the next scene when theirs has no slices left (work stealing: a wrong
    // placement guess only costs L2 misses).  Counters (hits per scene, slice tickets): common.h, ssd_counter. ----
    const uint32_t start_scene = (blockIdx.x & 7u) % c.S;
    // the order table is used only if stage A says it built one for THIS workspace (word 1 of scene 0's ticket line, k_ticket_order): the two launches read the
    // SSDNERF_TICKET_ORDER switch separately, and an order pointer into a table nobody wrote would send tickets to arbitrary slices (r05 advisor)
    const bool use_order = c.order != nullptr && queue_count[ssd_counter(SSD_CNT_TICKETS, c.S, 0) + 1] != 0u;
    const PT* planes_base = planes;
    const uint8_t* bits_base = lin_bits;
    const uint2* queue_base = queue;
    const float dt_gamma_default = c.m.dt_gamma;
    const float cell_world = c.m.two_rH * c.m.mip_bound;
    // ---- pre-split A operands.  wa1[mt][ks][term]: lane l holds the 8 bf16 terms of [W1 | b1 | 0][mt*32 + (l&31)][16 ks + 8 (l>>5) + e];
    // wa2[mt][term]: the same for Wd' (16 columns, one k-step; column 0 = fl32(Wd[:,0] * SH_0 + bd), the folded bias) ----
    // SM_K1_PACK (r03): layer 1 has 19 rows (18 features + the bias row), three more than a 16-deep k-step.  r01 / r02 ran a second k-step per
    // product -- 24 of the 72 MFMAs and three operand registers of zeros per product for rows 16 .. 18.  Now those rows of ALL six products share
    // ONE k-step: lane half 0 (k 0-7) carries [x_hi | x_mid | x_lo | x_hi](features 16, 17) against [W_hi | W_hi | W_hi | W_mid], lane half 1
    // (k 8-15) [x_hi | x_mid | (1, 1) | (1, 0)] against [W_lo | W_mid | (b_hi, b_mid) | (b_lo, 0)]: the same six products and the bias, 4 MFMAs
    // per 64 samples instead of 24, and the operand is four registers of data (no zero fill).  wa1x[mt] is that A operand.
#ifndef SM_K1_PACK
#define SM_K1_PACK 1
#endif
    sm_bf16x8 wa1[2][SM_K1_PACK ? 1 : 2][3], wa2[2][3];
#if SM_K1_PACK
    sm_bf16x8 wa1x[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 32 + (lane & 31);
        uint32_t wt[3][3];                                                   // [feature 16, feature 17, bias][term]
#pragma unroll
        for (int q = 0; q < 3; ++q) sm_split3(sm_prescale((double)P[row * 24 + 16 + q]), wt[q][0], wt[q][1], wt[q][2]);
        const uint32_t w_hi = sm_pack2(wt[0][0], wt[1][0]), w_mid = sm_pack2(wt[0][1], wt[1][1]), w_lo = sm_pack2(wt[0][2], wt[1][2]);
        wa1x[mt] = half == 0 ? sm_op(w_hi, w_hi, w_hi, w_mid) : sm_op(w_lo, w_mid, sm_pack2(wt[2][0], wt[2][1]), sm_pack2(wt[2][2], 0u));
    }
#endif
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < (SM_K1_PACK ? 1 : 2); ++ks) {
            uint32_t t1[3][8], t2[3][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * ks + 8 * half + e;
                const float w1 = k < 19 ? sm_prescale((double)P[row * 24 + k]) : 0.0f;                                   // W1 | b1 | 0
                sm_split3(w1, t1[0][e], t1[1][e], t1[2][e]);
                if (ks == 0) {
                    double w2d = (double)P[MLP_OFF_WD + row * 16 + k];
                    if (k == 0) w2d = w2d * (double)shb::C0 + (double)P[MLP_OFF_BD + row];
                    const float w2 = sm_prescale(w2d);
                    sm_split3(w2, t2[0][e], t2[1][e], t2[2][e]);
                }
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                wa1[mt][ks][t] = sm_op(sm_pack2(t1[t][0], t1[t][1]), sm_pack2(t1[t][2], t1[t][3]), sm_pack2(t1[t][4], t1[t][5]), sm_pack2(t1[t][6], t1[t][7]));
                if (ks == 0) wa2[mt][t] = sm_op(sm_pack2(t2[t][0], t2[t][1]), sm_pack2(t2[t][2], t2[t][3]), sm_pack2(t2[t][4], t2[t][5]), sm_pack2(t2[t][6], t2[t][7]));
            }
        }
    }
    const float b_sigma = P[MLP_OFF_TAIL + 0], bc0 = P[MLP_OFF_TAIL + 1], bc1 = P[MLP_OFF_TAIL + 2], bc2 = P[MLP_OFF_TAIL + 3];
    const float sat_k = ssd_fma(c.sat, 2.0f, 1.0f);

#ifdef SM_DEBUG_ITERS
  uint32_t dbg_iters = 0, dbg_live = 0;
  const uint64_t dbg_t0 = wall_clock64();
#endif
  // SM_DEBUG_SECTIONS (tools/shade_sections.py): shader cycles a wave spends in each section of the loop body (its own stalls and the other wave's
  // turns included), summed over waves into spare words of scene 0's boundary-counter line: 0 event, 1 gather, 2 split + MLP, 3 composite + search
#ifdef SM_DEBUG_SECTIONS
  uint64_t dbg_sec[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t dbg_marches = 0, dbg_stages = 0, dbg_poolrefills = 0;
  uint32_t dbg_events = 0, dbg_loops = 0;
  uint64_t dbg_last = clock64();
#define SM_SEC(i) do { __builtin_amdgcn_sched_barrier(0); const uint64_t n_ = clock64(); dbg_sec[i] += n_ - dbg_last; dbg_last = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SM_SEC(i) do { } while (0)
#endif
#if defined(SM_DEBUG_SECTIONS) && defined(SM_DEBUG_MLP_PHASES)
#define SM_SEC_MLP(i) SM_SEC(i)
#else
#define SM_SEC_MLP(i) do { } while (0)
#endif
#if defined(SM_DEBUG_SECTIONS) && defined(SM_DEBUG_MARCH) && !defined(SM_DEBUG_MLP_PHASES)
#define SM_SEC_MARCH(i) SM_SEC(i)          // 10: everything since the last mark up to the pass, 11: pool read + ray + bounds, 12: the probe loop, 13: stores + pool writes
#else
#define SM_SEC_MARCH(i) do { } while (0)
#endif
  for (uint32_t sk = 0; sk < c.S; ++sk) {
    const uint32_t scene = (start_scene + sk) % c.S;
    const uint32_t count = queue_count[ssd_counter(SSD_CNT_HITS, c.S, scene)];
    const uint32_t n_slices = (count + SM_SLICE - 1) / SM_SLICE;
    const uint64_t ray0 = (uint64_t)scene * c.N;
    planes = planes_base + scene * c.plane_stride;
    lin_bits = bits_base + (uint64_t)scene * c.bitfield_stride;
    const uint64_t* blocks64 = c.blocks64 ? c.blocks64 + (uint64_t)scene * (c.bitfield_stride >> 3) : nullptr;
    queue = queue_base + ray0;
    c.m.dt_gamma = DTG0 ? 0.0f : (c.dt_gammas ? c.dt_gammas[scene] : dt_gamma_default);
    uint32_t next = 0, end = 0;
    bool scene_done = n_slices == 0;

    // ---- lane state ----
    int ray = -1;
    RayGeom r = {};
    float sgx = 0.f, sgy = 0.f, sgz = 0.f;
    float t = 0.f, far_ = 0.f, ws = 0.f, dep = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    uint32_t cnt = 0;
    float sx = 0.f, sy = 0.f, sz = 0.f, sdt = 0.f;
    uint32_t sp_head = 0, sp_count = 0, rp_head = 0, rp_count = 0;    // wave-uniform pool cursors
    uint32_t st_head = 0, st_count = 0;                                // staged (prepared) rays
    bool fresh = false;                                                // this lane took a new ray in the refill above: its SH operands are due

    const bool packing = c.N <= SSD_RAY_ID_MASK + 1u;           // queue / pool ray words carry the tail bound in their upper 8 bits
    const uint32_t id_mask = packing ? SSD_RAY_ID_MASK : 0xffffffffu;
    auto write_out = [&](uint32_t rid, float ws_, float dep_, float cr_, float cg_, float cb_, uint32_t cnt_) {
        const uint64_t gi = ray0 + (rid & id_mask);
        const float bgk = c.bg * (1.0f - ws_);
        const float o0 = cr_ + bgk, o1 = cg_ + bgk, o2 = cb_ + bgk;
        image[3 * gi + 0] = o0;
        image[3 * gi + 1] = o1;
        image[3 * gi + 2] = o2;
        if (c.image_u8) { c.image_u8[3 * gi + 0] = ssd_quant_u8(o0); c.image_u8[3 * gi + 1] = ssd_quant_u8(o1); c.image_u8[3 * gi + 2] = ssd_quant_u8(o2); }
        depth[gi] = dep_;
        weights_sum[gi] = ws_;
        if (sample_counts) sample_counts[gi] = (int32_t)cnt_;
    };
    // SM_DEBUG_TRACE (r06, tools/trace_check.py): per ray, order-independent XOR hashes of 0 the march parameters t of its samples, 1 the gathered features, 2 the MLP
    // outputs + step, 3 the SH operands read for it, 4 the ray constants at every (re)load into a lane, 5 the composited state behind every sample -- flushed with
    // atomicXor whenever the ray leaves a lane.  Two renders of one scene must agree on every hash of every ray; the first kind that differs names the stage.
#ifdef SM_DEBUG_TRACE
    uint32_t th[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto th_mix = [](uint32_t x, uint32_t k) { uint32_t m = (x + k * 0x9E3779B9u) * 0x85EBCA6Bu; return m ^ (m >> 15); };
    auto th_flush = [&](uint32_t rid) {
        uint32_t* d = c.dbg_trace + (ray0 + (rid & id_mask)) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) { if (th[q] != 0u) atomicXor(d + q, th[q]); th[q] = 0u; }
    };
#define SM_TH_FLUSH(rid) th_flush(rid)
#else
#define SM_TH_FLUSH(rid) do { } while (0)
#endif
    auto set_signs = [&]() {
        sgx = ssd_fma(0.5f, ssd_sign1(r.dx), 0.5f); sgy = ssd_fma(0.5f, ssd_sign1(r.dy), 0.5f); sgz = ssd_fma(0.5f, ssd_sign1(r.dz), 0.5f);
    };
    auto store_sh = [&]() {    // SH'(d) of this lane's ray as three bf16 terms per value, in MFMA B-operand form: [term][sample][k 0-7 | k 8-15]
        float sh[16];
        shb::eval<4, false>(r.dx, r.dy, r.dz, sh, nullptr, nullptr, nullptr);
        sh[0] = 1.0f;                                            // SH_0 is constant: its weight column carries Wd[:,0] * SH_0 + bd
        uint32_t tm[3][16];
#pragma unroll
        for (int k = 0; k < 16; ++k) sm_split3(sh[k], tm[0][k], tm[1][k], tm[2][k]);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            uint4* dst = sh_lds + tt * 128 + lane * 2;
            dst[0] = make_uint4(sm_pack2(tm[tt][0], tm[tt][1]), sm_pack2(tm[tt][2], tm[tt][3]), sm_pack2(tm[tt][4], tm[tt][5]), sm_pack2(tm[tt][6], tm[tt][7]));
            dst[1] = make_uint4(sm_pack2(tm[tt][8], tm[tt][9]), sm_pack2(tm[tt][10], tm[tt][11]), sm_pack2(tm[tt][12], tm[tt][13]), sm_pack2(tm[tt][14], tm[tt][15]));
        }
    };

    uint32_t wait = 0;                                                 // 0: shading; 1: finished, outputs not stored yet; 2: must park (SM_EVENT_MIN)
    for (;;) {
      // ================= event: store finished rays, park, refill, march pass (every iteration when SM_EVENT_MIN == 1) =================
      bool event = true;
      if (SM_EVENT_MIN > 1) {
          const uint64_t actv = __ballot(ray >= 0 && wait == 0);
          const bool sources = rp_count != 0 || st_count != 0 || next < end || !scene_done;
          const uint32_t waiting = (uint32_t)__popcll(__ballot(wait != 0)) + (sources ? (uint32_t)__popcll(__ballot(ray < 0)) : 0u);
          event = actv == 0 || waiting >= (uint32_t)SM_EVENT_MIN ||
                  (SM_DRAIN_MARCH && !sources && sp_count != 0 && (uint32_t)__popcll(actv) <= 48u);       // draining with parked rays: see the march-pass trigger below
      }
      if (event) {
        SM_SEC(0);
        if (wait == 1) { SM_TH_FLUSH((uint32_t)ray); write_out((uint32_t)ray, ws, dep, cr, cg, cb, cnt); ray = -1; wait = 0; }      // the ONE store sequence of the loop body
        SM_SEC(4);
        // ---- park rays that need a long search (state -> LDS search pool) ----
        {
            const uint64_t pm = __ballot(wait == 2);
            if (pm != 0) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                const uint32_t room = SM_POOL - sp_count;
                if (wait == 2 && rank < room) {
                    uint32_t* e = pool_search + ((sp_head + sp_count + rank) % SM_POOL) * 8;
                    *reinterpret_cast<uint4*>(e) = make_uint4((uint32_t)ray, __float_as_uint(t), __float_as_uint(ws), __float_as_uint(dep));
                    *reinterpret_cast<uint4*>(e + 4) = make_uint4(__float_as_uint(cr), __float_as_uint(cg), __float_as_uint(cb), cnt);
                    SM_TH_FLUSH((uint32_t)ray);
                    ray = -1;
                    wait = 0;
                }
                sp_count += min((uint32_t)__popcll(pm), room);
                if (wait == 2) {   // pool full (cannot happen with the march-pass trigger below, kept as a guard): finish the search in the lane
                    for (;;) {
                        if (!(t < far_)) { SM_TH_FLUSH((uint32_t)ray); write_out((uint32_t)ray, ws, dep, cr, cg, cb, cnt); ray = -1; break; }
                        const ProbeB p = sm_probe<DTG0>(c.m, lin_bits, r, t);
                        if (p.occ) { sx = p.x; sy = p.y; sz = p.z; sdt = p.dt; break; }
                        t = sm_skip<DTG0>(c.m, r, p, sgx, sgy, sgz, t);
                    }
                    wait = 0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        SM_SEC(5);
        // ---- refill idle lanes: parked-and-found rays first, then the global hit queue ----
        {
            const uint64_t idle = __ballot(ray < 0);
            if (idle != 0 && rp_count != 0) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t take = min((uint32_t)__popcll(idle), rp_count);
                if (ray < 0 && rank < take) {
                    const uint32_t* e = pool_ready + ((rp_head + rank) % SM_POOL) * 8;
                    const uint4 e0 = *reinterpret_cast<const uint4*>(e), e1 = *reinterpret_cast<const uint4*>(e + 4);
                    ray = (int)e0.x; t = __uint_as_float(e0.y); ws = __uint_as_float(e0.z); dep = __uint_as_float(e0.w);
                    cr = __uint_as_float(e1.x); cg = __uint_as_float(e1.y); cb = __uint_as_float(e1.z); cnt = e1.w;
                    r = ssd_fetch_ray(src, scene, c.N, e0.x & id_mask);
                    float near_;
                    ssd_near_far(c.aabb, r, c.min_near, near_, far_);
                    far_ = ssd_tail_far(r, cell_world, near_, far_, e0.x, packing);
                    set_signs();
#if !SM_SEARCH_AHEAD                                                               // (SM_SEARCH_AHEAD: position and step are recomputed from t where they are used)
                    const ProbeB p = sm_probe<DTG0>(c.m, lin_bits, r, t);      // t points at an occupied probe
                    sx = p.x; sy = p.y; sz = p.z; sdt = p.dt;
#endif
                    fresh = true;
                }
                rp_head = (rp_head + take) % SM_POOL;
                rp_count -= take;
            }
        }
        SM_SEC(6);
        // ---- from the staged rays (LDS); when the stage runs dry the whole wave prepares the next 64 queue entries ----
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const uint64_t idle = __ballot(ray < 0);
            if (idle == 0) break;
            if (st_count == 0) {
                if (next >= end && !scene_done) {            // current slice used up: take a ticket for the next one
                    uint32_t sl = 0;
                    if (lane == 0) sl = atomicAdd(queue_count + ssd_counter(SSD_CNT_TICKETS, c.S, scene), 1u);
                    sl = __builtin_amdgcn_readfirstlane(sl);
                    if (sl < n_slices) {
                        if (use_order) sl = c.order[(uint64_t)scene * c.order_stride + sl];
                        next = sl * SM_SLICE; end = min(next + SM_SLICE, count);
                    }
                    else scene_done = true;
                }
                if (next >= end) break;
                const uint32_t n = min(end - next, SM_STAGE);
                if ((uint32_t)lane < n) {                    // stage fill: one queue entry per lane, all lanes busy
                    const uint2 e = queue[next + lane];
                    const RayGeom q = ssd_fetch_ray(src, scene, c.N, e.x & id_mask);
                    float qn, qf;
                    ssd_near_far(c.aabb, q, c.min_near, qn, qf);
                    qf = ssd_tail_far(q, cell_world, qn, qf, e.x, packing);
                    const float qt = __uint_as_float(e.y);
#if SM_SEARCH_AHEAD
                    ProbeB p; p.x = p.y = p.z = p.dt = 0.f;                     // (unused: recomputed from t at the top of the shading section)
#else
                    const ProbeB p = sm_probe<DTG0>(c.m, lin_bits, q, qt);      // the queued t is an occupied probe by construction
#endif
                    float4* dst = stage + lane * 4;
                    dst[0] = make_float4(__uint_as_float(e.x), qt, qf, p.dt);
                    dst[1] = make_float4(q.ox, q.oy, q.oz, q.dx);
                    dst[2] = make_float4(q.dy, q.dz, q.rdx, q.rdy);
                    dst[3] = make_float4(q.rdz, p.x, p.y, p.z);
                }
                next = __builtin_amdgcn_readfirstlane(next + n);
                st_head = 0; st_count = n;
#ifdef SM_DEBUG_SECTIONS
                ++dbg_stages;
#endif
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t take = min((uint32_t)__popcll(idle), st_count);
            if (ray < 0 && rank < take) {
                const float4* sp = stage + (st_head + rank) * 4;
                const float4 g0 = sp[0], g1 = sp[1], g2 = sp[2], g3 = sp[3];
                ray = (int)__float_as_uint(g0.x); t = g0.y; far_ = g0.z; sdt = g0.w;
                r.ox = g1.x; r.oy = g1.y; r.oz = g1.z; r.dx = g1.w; r.dy = g2.x; r.dz = g2.y; r.rdx = g2.z; r.rdy = g2.w; r.rdz = g3.x;
                sx = g3.y; sy = g3.z; sz = g3.w;
                set_signs();
                ws = dep = cr = cg = cb = 0.f; cnt = 0;
                fresh = true;
            }
            st_head += take; st_count -= take;
        }
        SM_SEC(7);
#ifdef SM_DEBUG_TRACE
        if (fresh) {
            uint32_t hh = th_mix((uint32_t)ray, 1u) ^ th_mix(__float_as_uint(t), 2u) ^ th_mix(__float_as_uint(far_), 3u) ^ th_mix(__float_as_uint(r.ox), 4u) ^ th_mix(__float_as_uint(r.oy), 5u) ^
                          th_mix(__float_as_uint(r.oz), 6u) ^ th_mix(__float_as_uint(r.dx), 7u) ^ th_mix(__float_as_uint(r.dy), 8u) ^ th_mix(__float_as_uint(r.dz), 9u) ^
                          th_mix(__float_as_uint(r.rdx), 10u) ^ th_mix(__float_as_uint(r.rdy), 11u) ^ th_mix(__float_as_uint(r.rdz), 12u) ^ th_mix(__float_as_uint(ws), 13u) ^ th_mix(cnt, 14u);
            th[4] ^= th_mix(hh, cnt);
        }
#endif
        if (fresh) { store_sh(); fresh = false; }               // (one copy of the SH evaluation + operand split for both refill sources)
        const uint64_t live = __ballot(ray >= 0);
        SM_SEC(8);

        // ================= march pass: every lane takes one parked ray to its next hit or to the end of the box =================
        // (r05) ... or, once the scene's queue is used up, as soon as a quarter of the lanes idle: parked rays that wait for `live == 0` would start their
        // remaining samples only after everything else has drained, one drain after the other at the very end of the launch
        const bool draining = SM_DRAIN_MARCH && scene_done && next >= end && st_count == 0 && rp_count == 0 && sp_count != 0 && (uint32_t)__popcll(live) <= 48u;
        if ((sp_count >= SM_MARCH_W || draining || (live == 0 && sp_count != 0)) && rp_count <= SM_POOL - SM_MARCH_W) {
            SM_SEC_MARCH(10);
            const uint32_t n = min(sp_count, SM_MARCH_W);
            bool found = false, again = false, mine = (uint32_t)lane < n;
            uint4 e0 = make_uint4(0, 0, 0, 0), e1 = make_uint4(0, 0, 0, 0);
            if (mine) {
                const uint32_t* e = pool_search + ((sp_head + lane) % SM_POOL) * 8;
                e0 = *reinterpret_cast<const uint4*>(e); e1 = *reinterpret_cast<const uint4*>(e + 4);
                const RayGeom q = ssd_fetch_ray(src, scene, c.N, e0.x & id_mask);
                float qn, qf;
                ssd_near_far(c.aabb, q, c.min_near, qn, qf);
                qf = ssd_tail_far(q, cell_world, qn, qf, e0.x, packing);
                const float qx = ssd_fma(0.5f, ssd_sign1(q.dx), 0.5f), qy = ssd_fma(0.5f, ssd_sign1(q.dy), 0.5f), qz = ssd_fma(0.5f, ssd_sign1(q.dz), 0.5f);
                float qt = __uint_as_float(e0.y);
                SM_SEC_MARCH(11);
                // SM_MARCH_TRIPS (r05): a pass lasts as long as its slowest lane, and a few rays (no tail bound: a gap between two parts of the object, the whole
                // box behind it) need dozens of dependent probes where the others need three -- a lane gives up after this many and its ray goes back to
                // the search pool with the parameter it reached (0: no limit, the r01 - r04 form; the sequence of parameters does not depend on where
                // a pass stops, so nothing a ray computes changes)
                uint32_t trips = 0;
                if (SM_MARCH_BLOCKS && blocks64 != nullptr) {
                    // BLOCK SKIP (k_survivor_march's, r03; here since r05): the march's parameter sequence does not depend on the cells, and the next sample is the
                    // first member whose cell is occupied.  While the 4^3 block (or its 2^3 sub-block) around the position holds no occupied cell, the members
                    // whose positions are still inside it -- monotone per axis, `blk_eps` inside the exit face covers rounding -- are run through without
                    // probes: one 8-byte load per block instead of one byte load per cell (a parked ray crosses ~30 empty cells to leave the box).
                    const float blk_eps = c.m.two_rH * c.m.mip_bound * (1.0f / 4096.0f);
                    const float ex = blk_eps * ssd_sign1(q.dx), ey = blk_eps * ssd_sign1(q.dy), ez = blk_eps * ssd_sign1(q.dz);
                    const uint32_t lb = c.m.log2H - 2;
                    while (qt < qf) {
                        if (SM_MARCH_TRIPS != 0 && trips == (uint32_t)SM_MARCH_TRIPS) { again = true; break; }
                        ++trips;
                        uint32_t idx;
                        const ProbeB p = sm_probe_addr<DTG0>(c.m, q, qt, idx);
                        const uint32_t bi = (((((uint32_t)p.nz >> 2) << lb) + ((uint32_t)p.ny >> 2)) << lb) + ((uint32_t)p.nx >> 2);
                        const uint64_t w = blocks64[bi];
                        float tt;
                        if (w == 0) tt = sm_block_exit<4>(c.m, q, p, qx, qy, qz, ex, ey, ez, qt);
                        else {
                            const uint64_t sub = w >> ((((uint32_t)p.nz & 2u) << 4) | (((uint32_t)p.ny & 2u) << 2) | ((uint32_t)p.nx & 2u));
                            if ((sub & 0x00330033ull) == 0) tt = sm_block_exit<2>(c.m, q, p, qx, qy, qz, ex, ey, ez, qt);
                            else {
                                if ((sub >> ((((uint32_t)p.nz & 1u) << 4) | (((uint32_t)p.ny & 1u) << 2) | ((uint32_t)p.nx & 1u))) & 1ull) { found = true; break; }
                                qt = sm_skip<DTG0>(c.m, q, p, qx, qy, qz, qt);
                                continue;
                            }
                        }
                        qt = sm_run_to<DTG0>(c.m, qt, tt);
                    }
                } else
                while (qt < qf) {
                    if (SM_MARCH_TRIPS != 0 && trips == (uint32_t)SM_MARCH_TRIPS) { again = true; break; }
                    ++trips;
                    const ProbeB p = sm_probe<DTG0>(c.m, lin_bits, q, qt);
                    if (p.occ) { found = true; break; }
                    qt = sm_skip<DTG0>(c.m, q, p, qx, qy, qz, qt);
                }
                SM_SEC_MARCH(12);
                if (found || again) e0.y = __float_as_uint(qt);
                else write_out(e0.x, __uint_as_float(e0.z), __uint_as_float(e0.w), __uint_as_float(e1.x), __uint_as_float(e1.y), __uint_as_float(e1.z), e1.w);
            }
            const uint64_t fm = __ballot(found);
            if (found) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                uint32_t* e = pool_ready + ((rp_head + rp_count + rank) % SM_POOL) * 8;
                *reinterpret_cast<uint4*>(e) = e0; *reinterpret_cast<uint4*>(e + 4) = e1;
            }
            rp_count += (uint32_t)__popcll(fm);
            sp_head = (sp_head + n) % SM_POOL;
            sp_count -= n;
            const uint64_t am = __ballot(again);                           // back to the search pool, behind what is already parked (n slots were just freed)
            if (again) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                uint32_t* e = pool_search + ((sp_head + sp_count + rank) % SM_POOL) * 8;
                *reinterpret_cast<uint4*>(e) = e0; *reinterpret_cast<uint4*>(e + 4) = e1;
            }
            sp_count += (uint32_t)__popcll(am);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            SM_SEC_MARCH(13);
#ifdef SM_DEBUG_SECTIONS
            ++dbg_marches;
#endif
            SM_SEC(9);
            continue;   // refill from the ready pool before shading
        }
        if (live == 0) {
            if (scene_done && next >= end && st_count == 0 && sp_count == 0 && rp_count == 0) break;      // this scene is finished for this wave: go steal from the next one
            continue;
        }
#ifdef SM_DEBUG_SECTIONS
        ++dbg_events;
#endif
      }   // event
      const bool on = ray >= 0 && wait == 0;                          // this lane shades a sample in this iteration
#ifdef SM_DEBUG_ITERS
        ++dbg_iters; dbg_live += (uint32_t)__popcll(__ballot(on));
#endif
#ifdef SM_DEBUG_SECTIONS
      ++dbg_loops;
#endif
      SM_SEC(0);

        // ================= shade: gather -> MFMA layers -> output layer -> composite =================
        float f[18];
#ifdef SM_DEBUG_TRACE
        uint32_t th_tex = 0u, th_w = 0u;
#endif
#if SM_SEARCH_AHEAD
        // the successor search of this sample, as far as it can go without the MLP: t1, t2, t3 and the two occupancy bytes (requested here, read after compositing)
        float sa_t2 = 0.f, sa_t3 = 0.f;
        uint32_t sa_b1 = 0, sa_b2 = 0, sa_sh = 0;
#if SM_GATHER_FIRST && SM_ADDR32
        // (r05) one block, in this order: the twelve texel requests, THEN the search arithmetic (two probe addresses, two DDA skips: ~100 VALU
        // instructions that need nothing from memory) under their latency, then the blend
#if SM_GATHER_DRAIN & 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (r06 experiment) nothing older in flight when the texel requests start
#endif
        if (on) {
            sm_sample_at<DTG0>(c.m, r, t, sx, sy, sz, sdt);
            SmPlaneTap<PT> tap[3];
            tap[0].issue(planes, c.g, 0, sx, sy);
            tap[1].issue(planes, c.g, 1, sx, sz);
            tap[2].issue(planes, c.g, 2, sy, sz);
            __builtin_amdgcn_sched_barrier(0);
            uint32_t i1, i2;
            const float t1 = t + sdt;
            const ProbeB q1 = sm_probe_addr<DTG0>(c.m, r, t1, i1);
#if SM_AHEAD_PROBES == 1
            i2 = 0; sa_t2 = sa_t3 = t1;
            sa_b1 = lin_bits[i1 >> 3];
            sa_sh = i1 & 7u;
#else
            sa_t2 = sm_skip<DTG0>(c.m, r, q1, sgx, sgy, sgz, t1);
            const ProbeB q2 = sm_probe_addr<DTG0>(c.m, r, sa_t2, i2);
            sa_t3 = sm_skip<DTG0>(c.m, r, q2, sgx, sgy, sgz, sa_t2);
            sa_b1 = lin_bits[i1 >> 3];
            sa_b2 = lin_bits[i2 >> 3];
            sa_sh = (i1 & 7u) | ((i2 & 7u) << 3);
#endif
            __builtin_amdgcn_sched_barrier(0);
#if SM_GATHER_DRAIN & 2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (r06 experiment) every request back before the first texel is used (no counted partial waits)
            __builtin_amdgcn_sched_barrier(0);
#endif
#if SM_GATHER_DRAIN & 4
            asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // ... and 16 idle issue slots behind the wait
            __builtin_amdgcn_sched_barrier(0);
#endif
#if SM_KEEP_ADDR
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) tap[pl].keep();
#endif
#ifdef SM_DEBUG_TRACE
            th_tex = 0u; th_w = 0u;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    th_tex = (th_tex << 3 | th_tex >> 29) ^ __float_as_uint(tap[pl].t00[q]) ^ (__float_as_uint(tap[pl].t01[q]) * 3u) ^ (__float_as_uint(tap[pl].t10[q]) * 5u) ^ (__float_as_uint(tap[pl].t11[q]) * 7u);
                th_w = (th_w << 5 | th_w >> 27) ^ __float_as_uint(tap[pl].w00) ^ (__float_as_uint(tap[pl].w01) * 3u) ^ (__float_as_uint(tap[pl].w10) * 5u) ^ (__float_as_uint(tap[pl].w11) * 7u);
            }
#endif
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) tap[pl].blend(pl, f);
#if SM_KEEP_ADDR
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) tap[pl].keep();
#endif
#ifdef SM_DEBUG_BLEND2
            {   // (r06) the blend a second time from the same registers (made opaque to the optimiser), and a third time as scalar FMAs: words 8 / 9 of scene 0's boundary line count
                // lanes whose packed evaluations disagree / whose scalar evaluation disagrees with the first packed one; 10: lanes checked; 12..29: per feature
                float f2[18], f3[18];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    SmPlaneTap<PT> q = tap[pl];
#pragma unroll
                    for (int e = 0; e < 6; ++e) asm volatile("" : "+v"(q.t00[e]), "+v"(q.t01[e]), "+v"(q.t10[e]), "+v"(q.t11[e]));
                    asm volatile("" : "+v"(q.w00), "+v"(q.w01), "+v"(q.w10), "+v"(q.w11));
                    q.blend(pl, f2);
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        float rr = q.t00[e] * q.w00;
                        rr = __builtin_fmaf(q.t01[e], q.w01, rr); rr = __builtin_fmaf(q.t10[e], q.w10, rr); rr = __builtin_fmaf(q.t11[e], q.w11, rr);
                        asm volatile("" : "+v"(rr));
                        f3[e * 3 + pl] = rr;
                    }
                }
                uint32_t* d = queue_count + ssd_counter(SSD_CNT_BOUNDARY, c.S, 0);
                bool b2 = false, b3 = false;
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    const bool x2 = __float_as_uint(f2[i]) != __float_as_uint(f[i]), x3 = __float_as_uint(f3[i]) != __float_as_uint(f[i]);
                    b2 |= x2; b3 |= x3;
                    if (x2 | x3) atomicAdd(d + 12 + i, 1u);
                }
                if ((b2 | b3) && c.S >= 8) {                                   // the first seven cases in full: words 8..27 of the boundary lines of scenes 1..7
                    const uint32_t k = atomicAdd(d + 11, 1u);
                    if (k < 7u) {
                        int fi = 0;
#pragma unroll
                        for (int i = 17; i >= 0; --i) if (__float_as_uint(f2[i]) != __float_as_uint(f[i]) || __float_as_uint(f3[i]) != __float_as_uint(f[i])) fi = i;
                        uint32_t* e = queue_count + ssd_counter(SSD_CNT_BOUNDARY, c.S, 1 + k) + 8;
                        e[0] = (uint32_t)lane | (uint32_t)fi << 8 | (b2 ? 1u << 16 : 0u) | (b3 ? 1u << 17 : 0u);
                        float fv = 0.f, f2v = 0.f, f3v = 0.f;
#pragma unroll
                        for (int i = 0; i < 18; ++i) if (i == fi) { fv = f[i]; f2v = f2[i]; f3v = f3[i]; }
                        e[1] = __float_as_uint(fv); e[2] = __float_as_uint(f2v); e[3] = __float_as_uint(f3v);
                        const int pl = fi % 3, ch = fi / 3;
#pragma unroll
                        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                            for (int cc = 0; cc < 6; ++cc) if (pp == pl && cc == ch) {
                                e[4] = __float_as_uint(tap[pp].t00[cc]); e[5] = __float_as_uint(tap[pp].t01[cc]); e[6] = __float_as_uint(tap[pp].t10[cc]); e[7] = __float_as_uint(tap[pp].t11[cc]);
                                e[8] = __float_as_uint(tap[pp].w00); e[9] = __float_as_uint(tap[pp].w01); e[10] = __float_as_uint(tap[pp].w10); e[11] = __float_as_uint(tap[pp].w11);
                                e[12] = __float_as_uint(tap[pp].t00[cc ^ 1]); e[13] = __float_as_uint(tap[pp].t01[cc ^ 1]);
                            }
                    }
                }
                if (b2) atomicAdd(d + 8, 1u);
                if (b3) atomicAdd(d + 9, 1u);
                if (lane == 0) atomicAdd(d + 10, 1u);
                if (b2 | b3) atomicOr(d + 30 + (lane >> 5), 1u << (lane & 31));
            }
#endif
        } else {
#pragma unroll
            for (int i = 0; i < 18; ++i) f[i] = 0.f;
        }
#else
        if (on) {
            sm_sample_at<DTG0>(c.m, r, t, sx, sy, sz, sdt);
            uint32_t i1, i2;
            const float t1 = t + sdt;
            const ProbeB q1 = sm_probe_addr<DTG0>(c.m, r, t1, i1);
            sa_t2 = sm_skip<DTG0>(c.m, r, q1, sgx, sgy, sgz, t1);
            const ProbeB q2 = sm_probe_addr<DTG0>(c.m, r, sa_t2, i2);
            sa_t3 = sm_skip<DTG0>(c.m, r, q2, sgx, sgy, sgz, sa_t2);
            sa_b1 = lin_bits[i1 >> 3];
            sa_b2 = lin_bits[i2 >> 3];
            sa_sh = (i1 & 7u) | ((i2 & 7u) << 3);
        }
#endif
#endif
#if !(SM_SEARCH_AHEAD && SM_GATHER_FIRST && SM_ADDR32)
#if SM_ADDR32
        if (on) sm_gather18<PT, SM_GATHER_BY_PLANE != 0>(planes, c.g, sx, sy, sz, f);
#else
        if (on) ssd_gather18<PT, SM_GATHER_BY_PLANE != 0>(planes, c.g, sx, sy, sz, f);
#endif
        else {
#pragma unroll
            for (int i = 0; i < 18; ++i) f[i] = 0.f;
        }
#endif
        SM_SEC(1);
#if defined(SM_X_VALU) || defined(SM_X_TRANS) || defined(SM_X_SLEEP) || defined(SM_X_LDS)
        // sensitivity probes (tools/ab_shade.sh, profiles/r05): extra work of ONE kind per iteration, results unchanged
        {
            float xd = f[0];
#ifdef SM_X_VALU
#pragma unroll
            for (int i = 0; i < SM_X_VALU; ++i) asm volatile("v_add_f32 %0, %1, %1" : "=v"(xd) : "v"(f[1]));
#endif
#ifdef SM_X_TRANS
            {
                float xe[8];
#pragma unroll
                for (int i = 0; i < SM_X_TRANS; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(xe[i & 7]) : "v"(f[1 + (i & 7)]));
                asm volatile("" :: "v"(xe[0]), "v"(xe[1]), "v"(xe[2]), "v"(xe[3]), "v"(xe[4]), "v"(xe[5]), "v"(xe[6]), "v"(xe[7]));
            }
#endif
#ifdef SM_X_SLEEP
            __builtin_amdgcn_s_sleep(SM_X_SLEEP);
#endif
#ifdef SM_X_LDS
            uint32_t xa = (uint32_t)lane * 16u;
#pragma unroll
            for (int i = 0; i < SM_X_LDS; ++i) { uint32_t xv; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(xv) : "v"(xa) : "memory"); xa = (xa + (xv & 0u)) ; }
#endif
            asm volatile("" :: "v"(xd));
        }
#endif
#ifdef SM_DEBUG_TRACE
        uint32_t th_f = 0u;
#pragma unroll
        for (int i = 0; i < 18; ++i) th_f = (th_f << 3 | th_f >> 29) ^ __float_as_uint(f[i]);
        uint32_t th_sh = 0u;
#pragma unroll
        for (int tt_ = 0; tt_ < 3; ++tt_) {
            const uint4 q0 = sh_lds[tt_ * 128 + lane * 2], q1 = sh_lds[tt_ * 128 + lane * 2 + 1];
            th_sh = (th_sh << 5 | th_sh >> 27) ^ q0.x ^ (q0.y * 3u) ^ (q0.z * 5u) ^ (q0.w * 7u) ^ (q1.x * 11u) ^ (q1.y * 13u) ^ (q1.z * 17u) ^ (q1.w * 19u);
        }
#endif
        // Split every feature into three bf16 terms, pack feature pairs, and trade halves so that T[t][0..3] is tile 0's k-step-0 operand
        // (features 0-15 of the samples of lanes 0-31) and T[t][4..7] tile 1's; T[t][8] / Z[t] carry features 16, 17 for k-step 1 (their other
        // k slots are the bias row and zeros).
        // SM_DEBUG_DUAL (r06, tools/dual_check.py): the MLP of an iteration is evaluated TWICE by the same instructions on the same inputs (a loop that is not
        // unrolled); per-lane hashes of six stages -- 0 matrix operands (split + lane-half exchange), 1 accumulators behind layer 1, 2 density pre-activations,
        // 3 accumulators behind the direction term, 4 colour pre-activations, 5 the four outputs -- are compared between the two evaluations, and every lane
        // whose hashes differ is counted per stage and per FIRST differing stage in spare counter words (scene 0's boundary line, words 8 ..).  Localises the
        // run-to-run differences that need two waves per SIMD to a part of the MLP.
#ifdef SM_DEBUG_DUAL
        float dd_keep[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) dd_keep[i] = f[i];
        uint32_t dd_first[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        float sigma = 0.f, sr = 0.f, sg = 0.f, sb_ = 0.f;
#pragma unroll 1
        for (int dd_rep = 0; dd_rep < 2; ++dd_rep) {
#pragma unroll
        for (int i = 0; i < 18; ++i) f[i] = dd_keep[i];
        uint32_t dd_cur[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#define SM_DD(stage, bits) do { dd_cur[stage] = (dd_cur[stage] << 1 | dd_cur[stage] >> 31) ^ (uint32_t)(bits); } while (0)
#define SM_DD_ACC(stage, nt) do { _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) SM_DD(stage, __float_as_uint(acc[nt][mt_][i_])); } } while (0)
#else
#define SM_DD(stage, bits) do { } while (0)
#define SM_DD_ACC(stage, nt) do { } while (0)
#endif
        uint32_t T[3][9];
#if SM_K1_PACK
        uint32_t K1[2][4];                                                   // k-step 1 operands of tile 0 / tile 1 (see SM_K1_PACK)
#else
        uint32_t Z[3] = {0u, 0u, 0u};
#endif
#if SM_SWAP_F32
        // (r06) the half exchange on the fp32 FEATURES, in front of the split: f[k] <-> f[8 + k] puts feature k (lane half 0) / 8 + k (half 1) of tile 0's samples into
        // f[k] and of tile 1's into f[8 + k] -- the very values whose split terms the swaps further down used to exchange, so T[.][0..7] hold the same bits -- with 8
        // swaps instead of 12 (written while the swap -> MFMA pair was the suspect of the reproducibility hunt, kept because it is 0.7 % faster).  Features 16, 17 (one
        // k-step for all six products, SM_K1_PACK) keep their swaps of split terms.
#pragma unroll
        for (int k = 0; k < 8; ++k) sm_swap(f[k], f[8 + k]);
#endif
#pragma unroll
        for (int p2 = 0; p2 < 9; ++p2) {
            uint32_t h0, m0, l0, h1, m1, l1;
            sm_split3(f[2 * p2], h0, m0, l0);
            sm_split3(f[2 * p2 + 1], h1, m1, l1);
            T[0][p2] = sm_pack2(h0, h1); T[1][p2] = sm_pack2(m0, m1); T[2][p2] = sm_pack2(l0, l1);
        }
#if !SM_SWAP_F32
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sm_swap_u(T[tt][k], T[tt][4 + k]);
#if !SM_K1_PACK
            sm_swap_u(T[tt][8], Z[tt]);
#endif
        }
#endif
#if SM_K1_PACK
        // v_permlane32_swap(a, b): a = [a.lo ; b.lo], b = [a.hi ; b.hi] (lane halves).  With a = b = X: a = X of samples 0-31 in both halves (tile 0),
        // b = X of samples 32-63 (tile 1); with b = a constant: the constant lands in the upper half (k 8-15) of both tiles' operands
        K1[0][0] = T[0][8]; K1[1][0] = T[0][8]; sm_swap_u(K1[0][0], K1[1][0]);
        K1[0][1] = T[1][8]; K1[1][1] = T[1][8]; sm_swap_u(K1[0][1], K1[1][1]);
        K1[0][2] = T[2][8]; K1[1][2] = 0x3F803F80u; sm_swap_u(K1[0][2], K1[1][2]);     // (1.0, 1.0) x (b_hi, b_mid)
        K1[0][3] = T[0][8]; K1[1][3] = 0x00003F80u; sm_swap_u(K1[0][3], K1[1][3]);     // (1.0, 0)   x (b_lo, 0)
#endif
#if defined(SM_DEBUG_DUAL) && SM_K1_PACK
#pragma unroll
        for (int tt_ = 0; tt_ < 3; ++tt_) {
#pragma unroll
            for (int k_ = 0; k_ < 8; ++k_) SM_DD(0, T[tt_][k_]);
        }
#pragma unroll
        for (int k_ = 0; k_ < 4; ++k_) { SM_DD(0, K1[0][k_]); SM_DD(0, K1[1][k_]); }
#endif
        const uint32_t bias_pair = half == 0 ? 0x00003F80u : 0u;          // {bf16(1.0), 0}: the bias row of layer 1's k-step 1 (k = 18), lane half 0 only
        constexpr int TI[6] = {2, 1, 0, 1, 0, 0}, TJ[6] = {0, 1, 2, 0, 1, 0};   // products (weight term i) x (input term j), i + j <= 2, smallest first
        float ps0, ps1, pr0, pr1, pg0, pg1, pb0, pb1;            // per tile: this lane half's share of (sigma, r, g, b) pre-activations
        // ---- tile-interleaved schedule:
        //   A: layer 1, tile 0 (24 MFMA)   B: layer 1, tile 1 (24) || density head 0   C: dir term 0 (12) || density head 1
        //   D: dir term 1 (12) || colour head 0   E: colour head 1
        floatx16 acc[2][2];                                              // [tile][mt]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.0f;
        auto layer1 = [&](int nt, int pr_i) {                            // 4 MFMA: products (weight term i) x (feature term j) of both row tiles
            const int i = TI[pr_i], j = TJ[pr_i];
            const sm_bf16x8 b0 = sm_op(T[j][4 * nt], T[j][4 * nt + 1], T[j][4 * nt + 2], T[j][4 * nt + 3]);
#if SM_K1_PACK
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1[mt][0][i], b0, acc[nt][mt], 0, 0, 0);
            if (pr_i == 5) {                                             // rows 16 .. 18 of all six products, after the last (largest) product of rows 0 .. 15
                const sm_bf16x8 b1 = sm_op(K1[nt][0], K1[nt][1], K1[nt][2], K1[nt][3]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1x[mt], b1, acc[nt][mt], 0, 0, 0);
            }
#else
            const sm_bf16x8 b1 = sm_op(nt == 0 ? T[j][8] : Z[j], j == 0 ? bias_pair : 0u, 0u, 0u);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1[mt][0][i], b0, acc[nt][mt], 0, 0, 0);
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1[mt][1][i], b1, acc[nt][mt], 0, 0, 0);
            }
#endif
        };
        sm_bf16x8 sb[3];
        auto load_sh = [&](int nt) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) sb[tt] = *reinterpret_cast<const sm_bf16x8*>(sh_lds + tt * 128 + (nt * 32 + (lane & 31)) * 2 + half);
        };
        auto dir_term = [&](int nt, int pr_i) {                          // 2 MFMA: h += Wd' SH'(d)
            const int i = TI[pr_i], j = TJ[pr_i];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2[mt][i], sb[j], acc[nt][mt], 0, 0, 0);
        };
        floatx2 ps_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}};
        floatx2 pr_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}}, pg_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}}, pb_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}};
        // SiLU + output layer of pairs [Q0, Q0 + N) of tile NT, STAGE BY STAGE (r03): all scalings, all v_exp, the LDS weight reads, all adds, all
        // v_rcp, all products.  A pair-by-pair chain (exp, exp -> add -> rcp, rcp -> mul) puts every transcendental directly in front of its
        // consumer, where the toolchain must pad (one `s_nop` per pair and stage: 108 of the 777 instructions of this block in r02) and where
        // gfx950 needs MORE than that pad with two waves per SIMD (asm_postpass.py); staged, the consumers sit N instructions behind their
        // producers, the quarter-rate unit runs N * 2 ops back to back under the ordinary ones, and the post-pass has next to nothing to add.
        auto heads = [&](auto ntc, auto colour_c, auto q0c, auto nc) {
            constexpr int nt = decltype(ntc)::value, Q0 = decltype(q0c)::value, N = decltype(nc)::value;
            constexpr bool COLOUR = decltype(colour_c)::value;
            floatx2 h[N], u[N], v[N];
            float4 w0[N], w1[N];
#if SM_W_EARLY
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int q = Q0 + i, mt = q >> 3, p2 = q & 7;
                w0[i] = wout2[((mt * 8 + p2) * 2 + half) * 2];
                if (COLOUR) w1[i] = wout2[((mt * 8 + p2) * 2 + half) * 2 + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int q = Q0 + i, mt = q >> 3, p2 = q & 7;
                h[i] = floatx2{acc[nt][mt][2 * p2], acc[nt][mt][2 * p2 + 1]};              // = -log2(e) x the pre-activation (SM_PRESCALE)
            }
#pragma unroll
            for (int i = 0; i < N; ++i) { v[i].x = __builtin_amdgcn_exp2f(h[i].x); v[i].y = __builtin_amdgcn_exp2f(h[i].y); }
#if !SM_W_EARLY
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int q = Q0 + i, mt = q >> 3, p2 = q & 7;
                w0[i] = wout2[((mt * 8 + p2) * 2 + half) * 2];
                if (COLOUR) w1[i] = wout2[((mt * 8 + p2) * 2 + half) * 2 + 1];
            }
#endif
#pragma unroll
            for (int i = 0; i < N; ++i) {
#if SM_UNPACK
                u[i] = floatx2{sm_plain_add1(v[i].x), sm_plain_add1(v[i].y)};
#else
                u[i] = v[i] + floatx2{1.0f, 1.0f};
#endif
            }
#pragma unroll
            for (int i = 0; i < N; ++i) { v[i].x = __builtin_amdgcn_rcpf(u[i].x); v[i].y = __builtin_amdgcn_rcpf(u[i].y); }
#pragma unroll
            for (int i = 0; i < N; ++i) {
#if SM_UNPACK
                h[i] = floatx2{sm_plain_mul(h[i].x, v[i].x), sm_plain_mul(h[i].y, v[i].y)};
#else
                h[i] = h[i] * v[i];
#endif
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (COLOUR) {
                    pr_[nt] = sm_fma2(floatx2{w0[i].z, w0[i].w}, h[i], pr_[nt]);
                    pg_[nt] = sm_fma2(floatx2{w1[i].x, w1[i].y}, h[i], pg_[nt]);
                    pb_[nt] = sm_fma2(floatx2{w1[i].z, w1[i].w}, h[i], pb_[nt]);
                } else {
                    ps_[nt] = sm_fma2(floatx2{w0[i].x, w0[i].y}, h[i], ps_[nt]);
                }
            }
        };
#ifndef SM_QB
#define SM_QB {0, 3, 6, 9, 12, 14, 16}
#endif
        constexpr int QB[7] = SM_QB;                                     // 16 SiLU pairs spread over the 6 MFMA groups of a phase
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        SM_SEC_MLP(10);
        // ---- A
#pragma unroll
        for (int g = 0; g < 6; ++g) layer1(0, g);
        SM_SEC_MLP(11);
        SM_DD_ACC(1, 0);
        // ---- B: layer 1 of tile 1 || density head of tile 0;  C: direction term of tile 0 || density head of tile 1;  D: direction term of tile 1 || colour head of tile 0
        sm_static_for<6>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            layer1(1, g);
            heads(I0{}, std::false_type{}, std::integral_constant<int, QB[g]>{}, std::integral_constant<int, QB[g + 1] - QB[g]>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        // DIRP (template; r03, the r02 verdict's item 3b): 6 = all split products of the direction layer (the fp32 class of layer 1); 3 = only
        // hi*hi, hi*mid, mid*hi -- the direction term, an additive correction of the colour pre-activation, then carries 16 significand bits per
        // factor; 12 MFMAs and a third of its LDS operand reads fewer per 64 samples (kernel -5 %).  Measured on the bench scenes: the image
        // moves by <= 1.6e-6 (mean 3e-8; profiles/r03/r_dir_products.txt), every parity test holds with its tolerance unchanged.  The host picks
        // 3 unless the caller sets SSDNERF_SHADE_FULL_DIR_PRODUCTS in `planes_dtype`.
        auto dir_group = [&](int nt, int g) {
            if (DIRP == 6) dir_term(nt, g);
            else if (g & 1) dir_term(nt, 3 + g / 2);                         // products (1,0), (0,1), (0,0) behind the groups 1, 3, 5
        };
        SM_SEC_MLP(12);
        SM_DD_ACC(1, 1);
        SM_DD(2, __float_as_uint(ps_[0].x)); SM_DD(2, __float_as_uint(ps_[0].y));
        load_sh(0);
        sm_static_for<6>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            dir_group(0, g);
            heads(I1{}, std::false_type{}, std::integral_constant<int, QB[g]>{}, std::integral_constant<int, QB[g + 1] - QB[g]>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        SM_SEC_MLP(13);
        SM_DD_ACC(3, 0);
        SM_DD(2, __float_as_uint(ps_[1].x)); SM_DD(2, __float_as_uint(ps_[1].y));
        load_sh(1);
        sm_static_for<6>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            dir_group(1, g);
            heads(I0{}, std::true_type{}, std::integral_constant<int, QB[g]>{}, std::integral_constant<int, QB[g + 1] - QB[g]>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        SM_SEC_MLP(14);
        SM_DD_ACC(3, 1);
        SM_DD(4, __float_as_uint(pr_[0].x)); SM_DD(4, __float_as_uint(pr_[0].y)); SM_DD(4, __float_as_uint(pg_[0].x)); SM_DD(4, __float_as_uint(pg_[0].y));
        SM_DD(4, __float_as_uint(pb_[0].x)); SM_DD(4, __float_as_uint(pb_[0].y));
#ifdef SM_X_MFMA
        {   // sensitivity probe: SM_X_MFMA extra matrix instructions on their own accumulators, beside the last head (which has none of its own)
            floatx16 xa0, xa1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { xa0[i] = 0.f; xa1[i] = 0.f; }
#pragma unroll
            for (int i = 0; i < SM_X_MFMA / 2; ++i) {
                xa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2[0][0], sb[0], xa0, 0, 0, 0);
                xa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2[1][0], sb[1], xa1, 0, 0, 0);
            }
            asm volatile("" :: "v"(xa0), "v"(xa1));
        }
#endif
        // ---- E: colour head of tile 1, SM_HEAD_GROUP pairs at a time
        sm_static_for<16 / SM_HEAD_GROUP>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            heads(I1{}, std::true_type{}, std::integral_constant<int, g * SM_HEAD_GROUP>{}, std::integral_constant<int, SM_HEAD_GROUP>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        SM_SEC_MLP(15);
        SM_DD(4, __float_as_uint(pr_[1].x)); SM_DD(4, __float_as_uint(pr_[1].y)); SM_DD(4, __float_as_uint(pg_[1].x)); SM_DD(4, __float_as_uint(pg_[1].y));
        SM_DD(4, __float_as_uint(pb_[1].x)); SM_DD(4, __float_as_uint(pb_[1].y));
        ps0 = ps_[0].x + ps_[0].y; ps1 = ps_[1].x + ps_[1].y; pr0 = pr_[0].x + pr_[0].y; pr1 = pr_[1].x + pr_[1].y;
        pg0 = pg_[0].x + pg_[0].y; pg1 = pg_[1].x + pg_[1].y; pb0 = pb_[0].x + pb_[0].y; pb1 = pb_[1].x + pb_[1].y;
        // cross-half reduction: after the swap, (x0 + x1) on lane l is the total for sample l
        sm_swap(ps0, ps1); sm_swap(pr0, pr1); sm_swap(pg0, pg1); sm_swap(pb0, pb1);
#ifdef SM_DEBUG_DUAL
        sigma = ssd_exp(ps0 + ps1 + b_sigma);
        sr = ssd_fma(ssd_sigmoid(pr0 + pr1 + bc0), sat_k, -c.sat);
        sg = ssd_fma(ssd_sigmoid(pg0 + pg1 + bc1), sat_k, -c.sat);
        sb_ = ssd_fma(ssd_sigmoid(pb0 + pb1 + bc2), sat_k, -c.sat);
        SM_DD(5, __float_as_uint(sigma)); SM_DD(5, __float_as_uint(sr)); SM_DD(5, __float_as_uint(sg)); SM_DD(5, __float_as_uint(sb_));
        if (dd_rep == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) dd_first[q] = dd_cur[q];
        } else {
            uint32_t dd_mask = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) dd_mask |= (dd_cur[q] != dd_first[q] ? 1u : 0u) << q;
            if (dd_mask != 0) {                                                  // words 8..13: lanes per differing stage; 14: lanes; 16..21: lanes per FIRST differing stage; 22: lanes that were shading
                uint32_t* d = queue_count + ssd_counter(SSD_CNT_BOUNDARY, c.S, 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) if (dd_mask >> q & 1u) atomicAdd(d + 8 + q, 1u);
                atomicAdd(d + 14, 1u);
                atomicAdd(d + 16 + (__builtin_ctz(dd_mask)), 1u);
                if (on) atomicAdd(d + 22, 1u);
                atomicOr(d + 24 + (lane >> 5), 1u << (lane & 31));              // words 24, 25: which lanes ever differed
            }
        }
        }   // dd_rep
#else
        const float sigma = ssd_exp(ps0 + ps1 + b_sigma);
        const float sr = ssd_fma(ssd_sigmoid(pr0 + pr1 + bc0), sat_k, -c.sat);
        const float sg = ssd_fma(ssd_sigmoid(pg0 + pg1 + bc1), sat_k, -c.sat);
        const float sb_ = ssd_fma(ssd_sigmoid(pb0 + pb1 + bc2), sat_k, -c.sat);
#endif

        SM_SEC(2);
        if (on) {
            const float alpha = 1.0f - __expf(-sigma * sdt);
            const float Tr = 1.0f - ws;
            const float w = alpha * Tr;
            // diagnostic: termination tests that land within float noise of the threshold (the only rays whose sample count may differ from
            // the reference's, whose exp is CUDA's __expf: DESIGN.md "arithmetic contract"); a few % of the hitting rays, once each
#if !SM_NO_BOUNDARY_COUNT
            if (fabsf(Tr - c.T_thresh) < 2e-6f) atomicAdd(queue_count + ssd_counter(SSD_CNT_BOUNDARY, c.S, scene), 1u);
#endif
#ifdef SM_DEBUG_TRACE
            th[0] ^= th_mix(__float_as_uint(t), cnt);
            th[1] ^= th_mix(th_f, cnt);
            th[2] ^= th_mix(__float_as_uint(sigma) ^ (__float_as_uint(sr) * 3u) ^ (__float_as_uint(sg) * 5u) ^ (__float_as_uint(sb_) * 7u) ^ (__float_as_uint(sdt) * 11u), cnt);
            th[3] ^= th_mix(th_sh, cnt);
            th[6] ^= th_mix(th_tex, cnt);
            th[7] ^= th_mix(th_w, cnt);
#endif
            ws += w;
            dep = ssd_fma(w, t, dep);
            cr = ssd_fma(w, sr, cr);
            cg = ssd_fma(w, sg, cg);
            cb = ssd_fma(w, sb_, cb);
#ifdef SM_DEBUG_TRACE
            th[5] ^= th_mix(__float_as_uint(ws) ^ (__float_as_uint(dep) * 3u) ^ (__float_as_uint(cr) * 5u) ^ (__float_as_uint(cg) * 7u) ^ (__float_as_uint(cb) * 11u), cnt);
#endif
            ++cnt;
#if SM_SEARCH_AHEAD
            // the loop's tests in the loop's order -- far, cap, probe 1, far, probe 2, far, park -- as predicates and selects (the if / else-if chain
            // compiled to nine branches; r05 section timing: ~1 k cycles for ~50 VALU instructions)
            const float t1 = t + sdt;
            const bool go = !(Tr < c.T_thresh) & (t1 < far_);                      // still marching after this sample
            const bool capped = cnt >= c.cap;
            const bool o1 = ((sa_b1 >> (sa_sh & 7u)) & 1u) != 0, o2 = ((sa_b2 >> (sa_sh >> 3)) & 1u) != 0;
            const bool look = go & !capped;
            const bool take1 = look & o1;
#if SM_AHEAD_PROBES == 1
            const bool take2 = false;
            const bool park = look & !o1;                                          // one empty probe: park AT it (t1 < far_ holds: `go`); the march pass probes t1 again and skips from there
#else
            const bool look2 = look & !o1 & (sa_t2 < far_);
            const bool take2 = look2 & o2;
            const bool park = look2 & !o2 & (sa_t3 < far_);                        // two empty probes: park at t3 (SM_SEARCH_PROBES == 2 in this form)
#endif
            if (go & capped) { if (overflow_flag) atomicAdd(overflow_flag, 1); }
            t = take1 ? t1 : take2 ? sa_t2 : sa_t3;
            wait = (take1 | take2) ? 0u : park ? 2u : 1u;
#else
            t += sdt;
            if (Tr < c.T_thresh) wait = 1;
            else {
                uint32_t probes = 0;
                for (;;) {
                    if (!(t < far_)) { wait = 1; break; }
                    if (cnt >= c.cap) {
                        if (overflow_flag) atomicAdd(overflow_flag, 1);
                        wait = 1; break;
                    }
                    if (probes == SM_SEARCH_PROBES) { wait = 2; break; }
                    const ProbeB p = sm_probe<DTG0>(c.m, lin_bits, r, t);
                    if (p.occ) { sx = p.x; sy = p.y; sz = p.z; sdt = p.dt; break; }
                    t = sm_skip<DTG0>(c.m, r, p, sgx, sgy, sgz, t);
                    ++probes;
                }
            }
#endif
        }
        SM_SEC(3);
    }
  }   // scene loop
#ifdef SM_DEBUG_SECTIONS
  if (lane == 0) {   // words 10..29: cycles / 16 per section (u64 x 10); 4: events, 5: loop iterations, 6: waves, 7: march passes, 8: stage fills
      uint32_t* d = queue_count + ssd_counter(SSD_CNT_BOUNDARY, c.S, 0);
#pragma unroll
      for (int i = 0; i < 10; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(d + 10 + 2 * i), (unsigned long long)(dbg_sec[i] >> 4));   // (units of 16 cycles)
      // MLP phases (SM_DEBUG_MLP_PHASES): 10 split, 11 A, 12 B, 13 C, 14 D, 15 E -> the ticket line of scene 0, words 2..13 (the tickets are taken: word 0 only)
      uint32_t* d2 = queue_count + ssd_counter(SSD_CNT_TICKETS, c.S, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(d2 + 2 + 2 * i), (unsigned long long)(dbg_sec[10 + i] >> 4));
      atomicAdd(d + 4, dbg_events); atomicAdd(d + 5, dbg_loops); atomicAdd(d + 6, 1u); atomicAdd(d + 7, dbg_marches); atomicAdd(d + 8, dbg_stages);
  }
#endif
#ifdef SM_DEBUG_ITERS
  if (lane == 0) {   // words 1..3 of scene 0's boundary-counter line: sum of wave iterations, max over waves, sum of live lanes
      uint32_t* d = queue_count + ssd_counter(SSD_CNT_BOUNDARY, c.S, 0);
      atomicAdd(d + 1, dbg_iters); atomicMax(d + 2, dbg_iters); atomicAdd(d + 3, dbg_live >> 6);
      const uint64_t t1 = wall_clock64();                       // 100 MHz: start (min), end (max), sum of ends over waves (relative to a coarse origin)
      atomicMax(reinterpret_cast<unsigned long long*>(d + 4), ~(unsigned long long)dbg_t0);     // (counters start at zero: keep the complement)
      atomicMax(reinterpret_cast<unsigned long long*>(d + 6), (unsigned long long)t1);
      atomicAdd(reinterpret_cast<unsigned long long*>(d + 8), (unsigned long long)(t1 & 0xffffffffull));
      // histogram of the waves' own durations in bins of 0.1 ms (words 1..31 of the boundary lines of scenes 1 and 2: 62 bins) and, beside it, of their iteration counts per bin
      if (c.S >= 5) {
          const uint32_t bin = min((uint32_t)((t1 - dbg_t0) / 10000ull), 61u);
          atomicAdd(queue_count + ssd_counter(SSD_CNT_BOUNDARY, c.S, 1 + bin / 31) + 1 + bin % 31, 1u);
          atomicAdd(queue_count + ssd_counter(SSD_CNT_BOUNDARY, c.S, 3 + bin / 31) + 1 + bin % 31, dbg_iters);
      }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
extern "C" size_t ssdnerf_render_queue_workspace(uint32_t S, uint32_t N, uint32_t grid_size);   // render_queue.hip (same workspace layout)

static int sm_shade(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params, uint32_t grid_size, const RaySrc& src,
                    uint32_t S, uint32_t N, float bound, float min_near, float dt_gamma, const float* dt_gammas, uint32_t max_steps, float T_thresh,
                    float bg_color, float sigmoid_saturation, float* image, float* depth, float* weights_sum, int32_t* sample_counts,
                    int32_t* overflow_flag, uint8_t* image_u8, void* workspace, size_t workspace_bytes, void* stream) {
    SSD_REQUIRE(planes && mlp_params && image && depth && weights_sum && workspace, "render_shade_queue_mfma: null pointer");
    const bool full_dir = (planes_dtype & SSDNERF_SHADE_FULL_DIR_PRODUCTS) != 0;
    planes_dtype &= 0xff;
    SSD_REQUIRE(planes_dtype == 0 || planes_dtype == 1, "render_shade_queue_mfma: unsupported plane dtype");
    SSD_REQUIRE(grid_size >= 8 && grid_size <= 512 && (grid_size & (grid_size - 1)) == 0, "render_shade_queue_mfma: grid_size must be a power of two in [8, 512]");
    SSD_REQUIRE((uint64_t)3 * Hp * Wp * 8 * (planes_dtype == 0 ? 4 : 2) < (1ull << 32), "render_shade_queue_mfma: a scene's planes must stay below 4 GiB (32-bit texel offsets)");
    if (workspace_bytes < ssdnerf_render_queue_workspace(S, N, grid_size))
        return ssdnerf_fail(SSDNERF_E_WORKSPACE, "render_shade_queue_mfma: workspace too small");
    ShadeCfg c;
    const MarchCfg mc = ssd_make_march_cfg(bound, dt_gamma, max_steps, 1, grid_size, nullptr);
    c.m.bound = bound; c.m.dt_gamma = dt_gamma; c.m.dt_min = mc.dt_min; c.m.dt_max = mc.dt_max;
    c.m.mip_bound = fminf(1.0f, bound); c.m.rb = 1.0f / c.m.mip_bound;
    c.m.half_H = 0.5f * (float)grid_size; c.m.two_rH = 2.0f / (float)grid_size; c.m.Hm1f = (float)(grid_size - 1);
    c.m.H = grid_size; c.m.log2H = (uint32_t)__builtin_ctz(grid_size);
    c.g = ssd_plane_geom(Hp, Wp);
    c.aabb[0] = c.aabb[1] = c.aabb[2] = -bound; c.aabb[3] = c.aabb[4] = c.aabb[5] = bound;
    c.min_near = min_near; c.T_thresh = T_thresh; c.bg = bg_color; c.sat = sigmoid_saturation;
    c.N = N; c.S = S; c.cap = max_steps;
    c.plane_stride = (uint64_t)3 * Hp * Wp * 8;
    c.bitfield_stride = (grid_size * grid_size * grid_size) / 8;
    c.dt_gammas = dt_gammas;
    c.image_u8 = image_u8;
    const RenderWs w = ssd_render_ws(workspace, S, N, grid_size);
    const char* to_env = getenv("SSDNERF_TICKET_ORDER");            // must say what render_first_hit said for this render (render_queue.hip, rq_first_hit)
    const bool ticket_order = !(to_env && to_env[0] == '0');
    static_assert(SM_SLICE == 64, "k_ticket_order sorts 64-entry slices");
    c.blocks64 = (grid_size >= 16 && getenv("SSDNERF_NO_COARSE") == nullptr) ? w.blocks64 : nullptr;      // (what rq_first_hit built: same condition)
    c.order = ticket_order ? w.order : nullptr;
    c.order_stride = w.order_stride;
#ifdef SM_DEBUG_TRACE
    {
        const char* tp = getenv("SSDNERF_DEBUG_TRACE_PTR");                 // a zeroed device buffer of S * N * 8 words (tools/trace_check.py)
        SSD_REQUIRE(tp != nullptr, "render_shade_queue_mfma: a SM_DEBUG_TRACE build needs SSDNERF_DEBUG_TRACE_PTR");
        c.dbg_trace = reinterpret_cast<uint32_t*>(strtoull(tp, nullptr, 0));
    }
#endif
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    // residency: 2 workgroups x 4 waves per CU, persistent.
    static int blocks_per_cu = 0;                                  // SSDNERF_SHADE_BLOCKS_PER_CU=1: one wave per SIMD (occupancy experiments only)
    static int force_generic = -1;                                 // SSDNERF_SHADE_GENERIC=1: never take the specialised forms (bit-identity test)
    if (blocks_per_cu == 0) {
        const char* e = getenv("SSDNERF_SHADE_BLOCKS_PER_CU");
        blocks_per_cu = (e && e[0] == '1') ? 1 : 2;
        force_generic = getenv("SSDNERF_SHADE_GENERIC") != nullptr;
    }
    // MODE 1 / 2: the cars / chairs / tables geometry exactly (the constants of csrc above); anything else runs the generic form
    const bool hot = !force_generic && grid_size == 64 && Hp == 128 && Wp == 128 && bound == 1.0f && max_steps == 256;
    const int mode = !hot ? 0 : ((dt_gammas == nullptr && dt_gamma == 0.0f) ? 2 : 1);
    dim3 g((unsigned)n_cu * (unsigned)blocks_per_cu), b(SM_TPB);
    hipStream_t s = (hipStream_t)stream;
#define SM_LAUNCH_D(PT, M, D) hipLaunchKernelGGL((k_shade_mfma<PT, M, D>), g, b, 0, s, c, src, (const PT*)planes, mlp_params, (const uint8_t*)w.lin_bits, (const uint2*)w.queue, w.counters, image, depth, weights_sum, sample_counts, overflow_flag)
#define SM_LAUNCH(PT, M) do { if (full_dir) SM_LAUNCH_D(PT, M, 6); else SM_LAUNCH_D(PT, M, 3); } while (0)
    if (planes_dtype == 0) { if (mode == 2) SM_LAUNCH(float, 2); else if (mode == 1) SM_LAUNCH(float, 1); else SM_LAUNCH(float, 0); }
    else { if (mode == 2) SM_LAUNCH(__half, 2); else if (mode == 1) SM_LAUNCH(__half, 1); else SM_LAUNCH(__half, 0); }
#undef SM_LAUNCH
#undef SM_LAUNCH_D
    SSD_CHECK_LAUNCH("render_shade_queue_mfma");
    return SSDNERF_OK;
}

extern "C" int ssdnerf_render_shade_queue_mfma(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                               uint32_t grid_size, const float* rays_o, const float* rays_d, uint32_t S, uint32_t N, float bound,
                                               float min_near, float dt_gamma, const float* dt_gammas, uint32_t max_steps, float T_thresh,
                                               float bg_color, float sigmoid_saturation, float* image, float* depth, float* weights_sum,
                                               int32_t* sample_counts, int32_t* overflow_flag, void* workspace, size_t workspace_bytes,
                                               void* stream) {
    if (N == 0 || S == 0) return SSDNERF_OK;
    SSD_REQUIRE(rays_o && rays_d, "render_shade_queue_mfma: null ray arrays");
    return sm_shade(planes, planes_dtype, Hp, Wp, mlp_params, grid_size, ssd_ray_src_arrays(rays_o, rays_d), S, N, bound, min_near, dt_gamma, dt_gammas,
                    max_steps, T_thresh, bg_color, sigmoid_saturation, image, depth, weights_sum, sample_counts, overflow_flag, nullptr, workspace, workspace_bytes,
                    stream);
}

extern "C" int ssdnerf_render_shade_queue_mfma_cams(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                                    uint32_t grid_size, const float* c2w, const float* intrinsics, uint32_t S, uint32_t V, uint32_t h,
                                                    uint32_t w, float bound, float min_near, float dt_gamma, const float* dt_gammas,
                                                    uint32_t max_steps, float T_thresh, float bg_color, float sigmoid_saturation, float* image,
                                                    float* depth, float* weights_sum, int32_t* sample_counts, int32_t* overflow_flag,
                                                    uint8_t* image_u8, void* workspace, size_t workspace_bytes, void* stream) {
    const uint64_t N64 = (uint64_t)V * h * w;
    if (N64 == 0 || S == 0) return SSDNERF_OK;
    SSD_REQUIRE(c2w && intrinsics && N64 <= 0xffffffffull && V <= 65535, "render_shade_queue_mfma_cams: bad camera arguments");
    return sm_shade(planes, planes_dtype, Hp, Wp, mlp_params, grid_size, ssd_ray_src_cams(c2w, intrinsics, V, h, w), S, (uint32_t)N64, bound, min_near,
                    dt_gamma, dt_gammas, max_steps, T_thresh, bg_color, sigmoid_saturation, image, depth, weights_sum, sample_counts, overflow_flag, image_u8,
                    workspace, workspace_bytes, stream);
}
