// ssdnerf_amd/csrc/shade_mfma.hip -- stage B of the fused renderer, MFMA form (the default shading kernel).
//
// Same contract as k_shade_queue (render_queue.hip): persistent waves shade the per-scene hit queues written by
// k_first_hit.  Two changes, both driven by the r01 profiles (k_shade_queue was ISSUE-bound: ~5600 VALU instructions
// per 64-sample iteration at ~50 % lane utilisation):
//
// (The numbered points describe the f32-MFMA form, variants 2 and 3; the default, variant 4, keeps the structure and runs the same two layers on
//  the bf16 matrix cores with exact three-term operand splitting -- see the variant list above SmGeo below.)
//
// 1. The two wide layers of the tiny MLP run on the matrix pipe in exact fp32:
//        h      = W1 . [f ; 1]        64 x (18+1)   -> v_mfma_f32_32x32x2_f32, 10 k-steps x (2 M-tiles x 2 N-tiles)
//        h_col  = h + Wd . [SH(d); 1] 64 x (16+1)   ->  9 k-steps, accumulated IN PLACE on top of h
//    (biases ride along as a constant-1 input row, so the accumulator starts from the inline constant 0).
//    One lane = one sample = one MFMA "column": lane l supplies the B operand of column l with ONE v_permlane32_swap per
//    k-step for both 32-sample tiles, and after a second swap+add every lane ends up with the four outputs (sigma, r, g, b)
//    of ITS OWN sample.  Weights live in 38 VGPRs as A operands (loaded once per wave) - no per-iteration scalar loads, and
//    the per-ray LDS cache of the direction term disappears (it is recomputed on the otherwise idle matrix pipe).
//    The 64->{1,3} output layer and the 128 SiLUs per sample stay on the VALU (4 outputs cannot fill an MFMA tile);
//    output weights sit in 1 KiB of LDS and are read as broadcast ds_read_b128.
//    f32-input MFMA is a k-ordered fp32 FMA chain (MI355X guide), i.e. the same arithmetic class as the VALU kernel.
//
// 2. A lane never searches more than SEARCH_PROBES empty voxels for its next sample.  A ray that needs a longer search
//    (typically: it left the object and must cross the rest of the box) is parked - state and all - in a wave-local
//    LDS pool; when 64 of them have collected (or nothing else is left) the wave runs a MARCH PASS in which every lane
//    marches one parked ray to its next hit (-> "ready" pool, picked up by the next refill) or to the end of the box
//    (-> finished).  Shading iterations therefore stay full, and marching runs at full lane utilisation too.
//
// Results are bit-identical in the integer outputs and within fp32 rounding of k_shade_queue for the floats (the MFMA
// accumulates the same products in a different, fixed order); tests/test_render_gpu.py checks both against the oracle.
#include "decode_core.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// SiLU of two values at once.  Default: the multiply / add / multiply around the two transcendentals are packed fp32 ops
// (v_pk_mul_f32 / v_pk_add_f32 process two values per instruction).  -DSM_SCALAR_VALU issues plain v_mul/v_add/v_fma instead
// (inline asm, so that the SLP vectoriser cannot re-pack them): packed f32 ops issued beside MFMAs cost extra cycles on gfx950
// (MI355X guide, "price of one filler beside MFMAs"), and phases B-D of the shading loop run VALU work under MFMAs on purpose.
#ifdef SM_SCALAR_VALU
SSD_DEV float sm_vmul(float a, float b) { float d; asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
SSD_DEV float sm_vadd(float a, float b) { float d; asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
SSD_DEV float sm_vfma(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
SSD_DEV floatx2 sm_silu2(floatx2 h) {
    floatx2 o;
    o.x = sm_vmul(h.x, __builtin_amdgcn_rcpf(sm_vadd(__builtin_amdgcn_exp2f(sm_vmul(h.x, -1.4426950408889634f)), 1.0f)));
    o.y = sm_vmul(h.y, __builtin_amdgcn_rcpf(sm_vadd(__builtin_amdgcn_exp2f(sm_vmul(h.y, -1.4426950408889634f)), 1.0f)));
    return o;
}
SSD_DEV floatx2 sm_fma2(floatx2 w, floatx2 v, floatx2 acc) { return floatx2{sm_vfma(w.x, v.x, acc.x), sm_vfma(w.y, v.y, acc.y)}; }
#else
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
SSD_DEV floatx2 sm_fma2(floatx2 w, floatx2 v, floatx2 acc) { return __builtin_elementwise_fma(w, v, acc); }
#endif

static constexpr unsigned SM_TPB = 256;
static constexpr unsigned SM_SLICE = 512;          // hit-queue entries per shading wave
static constexpr unsigned SM_SEARCH_PROBES = 4;    // in-lane search budget after each sample
// Variants of the same kernel (template parameter VAR):
//   VAR 2: weights as MFMA A operands in 38 VGPRs, both 32-sample tiles in flight (64 accumulator registers), 253 VGPRs, 66 KiB LDS per block.
//   VAR 3: weights as A operands read from LDS right before each MFMA, the two tiles shaded one after the other (32 accumulator registers),
//          B operands built in place; <= 168 VGPRs and <= 53 KiB LDS per block, so THREE waves share a SIMD: the r01 counters showed the
//          2-wave form waiting (SQ_WAIT_INST_ANY ~ 43 % of wave time) with both pipes half idle -- a third wave is what fills those gaps.
//   VAR 4: the two wide layers on the REAL matrix cores (v_mfma_f32_32x32x16_bf16) with every fp32 operand split exactly into three
//          bf16 terms (x = hi + mid + lo, 8 + 8 + 8 significand bits) and the six products whose weight is >= 2^-16 accumulated in
//          fp32: the same accuracy class as the fp32 chain (dropped terms <= 2^-24 relative), but -- unlike the f32-input MFMA, which
//          executes on the VALU's own FMA lanes and therefore cannot overlap with VALU work at all (tools/ubench/mfma_valu_overlap.hip:
//          2 MFMA + 32 v_fma take 126 ns = 59 + 70 with f32 MFMA, 77 ns with bf16 MFMA) -- it runs beside the SiLU / gather / marching VALU
//          stream.  Two waves per SIMD, weights as 96 VGPRs of pre-split A operands, tiles shaded in sequence, SH operands pre-split in LDS.
template <int VAR> struct SmGeo {
    static constexpr int WPS = VAR == 3 ? 3 : 2;
    static constexpr unsigned POOL = VAR == 3 ? 64 : 128;      // entries per wave-local pool (two pools per wave)
    static constexpr unsigned MARCH_W = POOL / 2;              // rays advanced by one march pass
    static constexpr unsigned STAGE = VAR == 3 ? 32 : 64;      // prepared rays per wave
    static constexpr unsigned WLDS = VAR == 3 ? (2 * 10 + 2 * 9) * 64 : 0;   // floats of A-operand weights in LDS
    static constexpr unsigned SHF = (VAR == 4 || VAR == 6) ? 1536 : 1024;    // floats of per-ray SH operands per wave (VAR 4: three bf16 terms x 64 rays x 16)
    static constexpr unsigned LDS_FLOATS = 512 + WLDS + (SM_TPB / 64) * (2 * POOL * 8 + SHF + STAGE * 16);
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
#ifndef SM_DEFAULT_VARIANT
#define SM_DEFAULT_VARIANT 4                       // measured r01 (bench scene / fog scene): variant 2 9.66 / 29.2 ms, 3 9.36 / 28.1 ms, 4 8.10 / 26.0 ms
#endif
#ifndef SM_HEAD_GROUP
#define SM_HEAD_GROUP 3                            // variant 4: SiLU pairs per scheduling group minus one (3 = four pairs)
#endif
#ifndef SM_REFILL_MIN
#define SM_REFILL_MIN 1                            // refill only when this many lanes are idle (the divergent refill code then runs every few iterations instead of every iteration)
#endif

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
};

struct ProbeB { float x, y, z, dt; int nx, ny, nz; bool occ; };

SSD_DEV ProbeB sm_probe(const FastMarchB& m, const uint8_t* __restrict__ lin_bits, const RayGeom& r, float t) {
    ProbeB p;
    p.x = ssd_clamp(ssd_fma(t, r.dx, r.ox), -m.bound, m.bound);
    p.y = ssd_clamp(ssd_fma(t, r.dy, r.oy), -m.bound, m.bound);
    p.z = ssd_clamp(ssd_fma(t, r.dz, r.oz), -m.bound, m.bound);
    p.dt = ssd_clamp(t * m.dt_gamma, m.dt_min, m.dt_max);
    p.nx = (int)ssd_clamp(ssd_fma(p.x, m.rb, 1.0f) * m.half_H, 0.0f, m.Hm1f);
    p.ny = (int)ssd_clamp(ssd_fma(p.y, m.rb, 1.0f) * m.half_H, 0.0f, m.Hm1f);
    p.nz = (int)ssd_clamp(ssd_fma(p.z, m.rb, 1.0f) * m.half_H, 0.0f, m.Hm1f);
    const uint32_t idx = (((uint32_t)p.nz << m.log2H) + (uint32_t)p.ny << m.log2H) + (uint32_t)p.nx;
    p.occ = (lin_bits[idx >> 3] >> (idx & 7u)) & 1u;
    return p;
}
SSD_DEV float sm_skip(const FastMarchB& m, const RayGeom& r, const ProbeB& p, float sgx, float sgy, float sgz, float t) {
    const float tx = ssd_fma(ssd_fma((float)p.nx + sgx, m.two_rH, -1.0f), m.mip_bound, -p.x) * r.rdx;
    const float ty = ssd_fma(ssd_fma((float)p.ny + sgy, m.two_rH, -1.0f), m.mip_bound, -p.y) * r.rdy;
    const float tz = ssd_fma(ssd_fma((float)p.nz + sgz, m.two_rH, -1.0f), m.mip_bound, -p.z) * r.rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do { t += ssd_clamp(t * m.dt_gamma, m.dt_min, m.dt_max); } while (t < tt);
    return t;
}

// Tail bound of a queued ray (common.h): past near + (j_last + 1) coarse steps no occupied cell can be met; + 0.5 step of slack for the
// difference between k_first_hit's accumulated test parameters and this product form.
SSD_DEV float sm_tail_far(const FastMarchB& m, const RayGeom& q, float near_, float far_, uint32_t packed, bool packing) {
    const uint32_t jl = packed >> 24;
    if (!packing || jl >= SSD_TAIL_NONE) return far_;
    const float len = sqrtf(ssd_fma(q.dx, q.dx, ssd_fma(q.dy, q.dy, q.dz * q.dz)));
    const float step_t = (SSD_COARSE_STEP * m.two_rH * m.mip_bound) / fmaxf(len, 1e-20f);
    return fminf(far_, ssd_fma((float)jl + 1.5f, step_t, near_));
}

// v_permlane32_swap: lanes 32..63 of `a` trade places with lanes 0..31 of `b`.
SSD_DEV void sm_swap(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

template <typename PT, int VAR>
__global__ void __launch_bounds__(SM_TPB, SmGeo<VAR>::WPS) k_shade_mfma(ShadeCfg c, uint32_t slices_per_scene, const PT* __restrict__ planes,
                                                           const float* __restrict__ P, const uint8_t* __restrict__ lin_bits,
                                                           const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                           const uint2* __restrict__ queue, uint32_t* __restrict__ queue_count,
                                                           float* __restrict__ image, float* __restrict__ depth, float* __restrict__ weights_sum,
                                                           int32_t* __restrict__ sample_counts, int32_t* __restrict__ overflow_flag) {
    // LDS: [0,1 KiB) output-layer weights per accumulator slot; then per wave two pools of SM_POOL x 8 dwords.
    // wout2: 64 entries x 8 floats; per (mt, pair p of adjacent accumulator registers, half): {ws_a, ws_b, wr_a, wr_b, wg_a, wg_b, wb_a, wb_b}
    // sh  : per wave 64 lanes x 16 floats as [k/4][lane][k%4] (lane-contiguous 16-byte slots: conflict-free ds_read_b128)
    // stage: per wave 64 PREPARED rays x 16 dwords {ray, t, far, dt | o | d | 1/d | sample xyz}: queue entries and ray geometry are fetched from
    //        HBM 64 at a time by the whole wave (coalesced, full lane utilisation) instead of one lane at a time inside the divergent refill
    constexpr int WPS = SmGeo<VAR>::WPS;
    constexpr unsigned SM_POOL = SmGeo<VAR>::POOL, SM_MARCH_W = SmGeo<VAR>::MARCH_W, SM_STAGE = SmGeo<VAR>::STAGE, SM_WLDS = SmGeo<VAR>::WLDS, SM_SHF = SmGeo<VAR>::SHF;
    __shared__ __attribute__((aligned(16))) float lds[SmGeo<VAR>::LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- output weights: slot (mt, reg, half) <-> hidden row mt*32 + (reg&3) + 8*(reg>>2) + 4*half (32x32 MFMA C/D layout) ----
    if (threadIdx.x < 64) {
        const int mt = threadIdx.x >> 4, pr_ = (threadIdx.x >> 1) & 7, hf = threadIdx.x & 1;       // slot = (mt*8 + pair)*2 + half
        float v[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int reg = 2 * pr_ + e;
            const int row = mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hf;
            const float* rec = P + row * 24;                         // rec[19] = w_sigma, rec[20..22] = Wc[0..2][row]
            v[0 + e] = rec[19]; v[2 + e] = rec[20]; v[4 + e] = rec[21]; v[6 + e] = rec[22];
        }
        reinterpret_cast<float4*>(lds)[threadIdx.x * 2 + 0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(lds)[threadIdx.x * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
    const float4* wout2 = reinterpret_cast<const float4*>(lds);     // index ((mt*8 + pair)*2 + half)*2 + {0: sigma|r, 1: g|b}
    float* w_lds = lds + 512;                                       // WPS 3: A operands [layer-1: mt][s][lane] then [dir: mt][s][lane]
    uint32_t* pool_search = reinterpret_cast<uint32_t*>(lds + 512 + SM_WLDS) + wave * 2 * SM_POOL * 8;
    uint32_t* pool_ready = pool_search + SM_POOL * 8;
    float4* sh_lds = reinterpret_cast<float4*>(lds + 512 + SM_WLDS + (SM_TPB / 64) * 2 * SM_POOL * 8 + wave * SM_SHF);   // [kq][lane]  (VAR 4: [term][sample][2 x 16 B])
    float4* stage = reinterpret_cast<float4*>(lds + 512 + SM_WLDS + (SM_TPB / 64) * (2 * SM_POOL * 8 + SM_SHF) + wave * SM_STAGE * 16);   // [slot][4]

    // ---- persistent grid: every wave pulls 512-ray slices of a scene's hit queue with one atomic ticket per slice.  Waves of
    // XCD x (workgroups are dispatched round-robin over the 8 XCDs, b % 8) start on scene x so that the scene's 1.5 MiB of
    // planes stay in that XCD's L2, and move on to the next scene when theirs has no slices left (work stealing: a wrong
    // placement guess only costs L2 misses).  queue_count[0..S) = hits per scene, queue_count[S..2S) = slice tickets. ----
    uint32_t* tickets = queue_count + c.S;
    const uint32_t start_scene = (blockIdx.x & 7u) % c.S;
    const PT* planes_base = planes;
    const uint8_t* bits_base = lin_bits;
    const uint2* queue_base = queue;
    const float dt_gamma_default = c.m.dt_gamma;
    // ---- A operands: lane l holds W[mt*32 + (l&31)][2s + (l>>5)] for every k-step s ----
    // VAR 4: pre-split A operands.  wa1[mt][ks][term] / wa2[mt][ks][term]: lane l holds the 8 bf16 terms of W[mt*32 + (l&31)][16 ks + 8 (l>>5) + e]
    sm_bf16x8 wa1[2][2][3], wa2[2][2][3];
    if constexpr (VAR == 4 || VAR == 6) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = mt * 32 + (lane & 31);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint32_t t1[3][8], t2[3][8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * ks + 8 * half + e;
                    const float w1 = k < 19 ? P[row * 24 + k] : 0.0f;                                                        // W1 | b1 | 0
                    const float w2 = k < 16 ? P[MLP_OFF_WD + row * 16 + k] : (k == 16 ? P[MLP_OFF_BD + row] : 0.0f);           // Wd | bd | 0
                    sm_split3(w1, t1[0][e], t1[1][e], t1[2][e]);
                    sm_split3(w2, t2[0][e], t2[1][e], t2[2][e]);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    wa1[mt][ks][t] = sm_op(sm_pack2(t1[t][0], t1[t][1]), sm_pack2(t1[t][2], t1[t][3]), sm_pack2(t1[t][4], t1[t][5]), sm_pack2(t1[t][6], t1[t][7]));
                    wa2[mt][ks][t] = sm_op(sm_pack2(t2[t][0], t2[t][1]), sm_pack2(t2[t][2], t2[t][3]), sm_pack2(t2[t][4], t2[t][5]), sm_pack2(t2[t][6], t2[t][7]));
                }
            }
        }
    }
    float a1[2][10], a2[2][9];
    if (VAR == 2 || (VAR == 3 && wave == 0)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = mt * 32 + (lane & 31);
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                const int k = 2 * s + half;                       // W1 | b1 | 0
                a1[mt][s] = k < 19 ? P[row * 24 + k] : 0.0f;      // rec[row][0..17] = W1, rec[row][18] = b1
            }
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const int k = 2 * s + half;                       // Wd | bd | 0
                a2[mt][s] = k < 16 ? P[MLP_OFF_WD + row * 16 + k] : (k == 16 ? P[MLP_OFF_BD + row] : 0.0f);
            }
        }
    }
    if constexpr (VAR == 3) {                                 // park the A operands in LDS (one copy per block), lane-contiguous: conflict-free ds_read_b32
        if (wave == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int s = 0; s < 10; ++s) w_lds[(mt * 10 + s) * 64 + lane] = a1[mt][s];
#pragma unroll
                for (int s = 0; s < 9; ++s) w_lds[1280 + (mt * 9 + s) * 64 + lane] = a2[mt][s];
            }
        }
        __syncthreads();
    }
    const float b_const = half == 0 ? 1.0f : 0.0f;            // B operand of the (1, 0) bias/pad k-step, both tiles
    const float b_sigma = P[MLP_OFF_TAIL + 0], bc0 = P[MLP_OFF_TAIL + 1], bc1 = P[MLP_OFF_TAIL + 2], bc2 = P[MLP_OFF_TAIL + 3];
    const float sat_k = ssd_fma(c.sat, 2.0f, 1.0f);

  for (uint32_t sk = 0; sk < c.S; ++sk) {
    const uint32_t scene = (start_scene + sk) % c.S;
    const uint32_t count = queue_count[scene];
    const uint32_t n_slices = (count + SM_SLICE - 1) / SM_SLICE;
    const uint64_t ray0 = (uint64_t)scene * c.N;
    planes = planes_base + scene * c.plane_stride;
    lin_bits = bits_base + (uint64_t)scene * c.bitfield_stride;
    queue = queue_base + ray0;
    c.m.dt_gamma = c.dt_gammas ? c.dt_gammas[scene] : dt_gamma_default;
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

    const bool packing = c.N <= SSD_RAY_ID_MASK + 1u;           // queue / pool ray words carry the tail bound in their upper 8 bits
    const uint32_t id_mask = packing ? SSD_RAY_ID_MASK : 0xffffffffu;
    auto write_out = [&](uint32_t rid, float ws_, float dep_, float cr_, float cg_, float cb_, uint32_t cnt_) {
        const uint64_t gi = ray0 + (rid & id_mask);
        const float bgk = c.bg * (1.0f - ws_);
        image[3 * gi + 0] = cr_ + bgk;
        image[3 * gi + 1] = cg_ + bgk;
        image[3 * gi + 2] = cb_ + bgk;
        depth[gi] = dep_;
        weights_sum[gi] = ws_;
        if (sample_counts) sample_counts[gi] = (int32_t)cnt_;
    };
    auto load_geometry = [&](uint32_t rid) {
        const uint64_t gi = ray0 + (rid & id_mask);
        r = ssd_load_ray(rays_o + 3 * gi, rays_d + 3 * gi);
        float near_;
        ssd_near_far(c.aabb, r, c.min_near, near_, far_);
        far_ = sm_tail_far(c.m, r, near_, far_, rid, packing);
        sgx = ssd_fma(0.5f, ssd_sign1(r.dx), 0.5f); sgy = ssd_fma(0.5f, ssd_sign1(r.dy), 0.5f); sgz = ssd_fma(0.5f, ssd_sign1(r.dz), 0.5f);
    };
    auto begin_ray = [&]() {   // geometry is loaded and t points at an occupied probe
        const ProbeB p = sm_probe(c.m, lin_bits, r, t);
        sx = p.x; sy = p.y; sz = p.z; sdt = p.dt;
        float sh[16];
        shb::eval<4, false>(r.dx, r.dy, r.dz, sh, nullptr, nullptr, nullptr);
        if constexpr (VAR == 4 || VAR == 6) {          // three bf16 terms per value, in MFMA B-operand form: [term][sample][k 0-7 | k 8-15]
            uint32_t tm[3][16];
#pragma unroll
            for (int k = 0; k < 16; ++k) sm_split3(sh[k], tm[0][k], tm[1][k], tm[2][k]);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                uint4* dst = reinterpret_cast<uint4*>(sh_lds) + t * 128 + lane * 2;
                dst[0] = make_uint4(sm_pack2(tm[t][0], tm[t][1]), sm_pack2(tm[t][2], tm[t][3]), sm_pack2(tm[t][4], tm[t][5]), sm_pack2(tm[t][6], tm[t][7]));
                dst[1] = make_uint4(sm_pack2(tm[t][8], tm[t][9]), sm_pack2(tm[t][10], tm[t][11]), sm_pack2(tm[t][12], tm[t][13]), sm_pack2(tm[t][14], tm[t][15]));
            }
        } else if constexpr (VAR == 3) {               // operand form [k][sample]: the dir-term MFMAs read their B operands straight from LDS
#pragma unroll
            for (int k = 0; k < 16; ++k) reinterpret_cast<float*>(sh_lds)[k * 64 + lane] = sh[k];
        } else {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) sh_lds[kq * 64 + lane] = make_float4(sh[4 * kq], sh[4 * kq + 1], sh[4 * kq + 2], sh[4 * kq + 3]);
        }
    };

    for (;;) {
        // ================= refill idle lanes: parked-and-found rays first, then the global hit queue =================
        const uint64_t idle_now = __ballot(ray < 0);
        const bool do_refill = (uint32_t)__popcll(idle_now) >= SM_REFILL_MIN || idle_now == ~0ull;
        if (do_refill) {
            const uint64_t idle = idle_now;
            if (idle != 0 && rp_count != 0) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t take = min((uint32_t)__popcll(idle), rp_count);
                if (ray < 0 && rank < take) {
                    const uint32_t* e = pool_ready + ((rp_head + rank) % SM_POOL) * 8;
                    const uint4 e0 = *reinterpret_cast<const uint4*>(e), e1 = *reinterpret_cast<const uint4*>(e + 4);
                    ray = (int)e0.x; t = __uint_as_float(e0.y); ws = __uint_as_float(e0.z); dep = __uint_as_float(e0.w);
                    cr = __uint_as_float(e1.x); cg = __uint_as_float(e1.y); cb = __uint_as_float(e1.z); cnt = e1.w;
                    load_geometry(e0.x);
                    begin_ray();
                }
                rp_head = (rp_head + take) % SM_POOL;
                rp_count -= take;
            }
        }
        // ---- from the staged rays (LDS); when the stage runs dry the whole wave prepares the next 64 queue entries ----
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const uint64_t idle = __ballot(ray < 0);
            if (idle == 0) break;
            if (st_count == 0) {
                if (next >= end && !scene_done) {            // current slice used up: take a ticket for the next one
                    uint32_t sl = 0;
                    if (lane == 0) sl = atomicAdd(tickets + scene, 1u);
                    sl = __builtin_amdgcn_readfirstlane(sl);
                    if (sl < n_slices) { next = sl * SM_SLICE; end = min(next + SM_SLICE, count); }
                    else scene_done = true;
                }
                if (next >= end) break;
                const uint32_t n = min(end - next, SM_STAGE);
                if ((uint32_t)lane < n) {                    // stage fill: one queue entry per lane, all lanes busy
                    const uint2 e = queue[next + lane];
                    const uint64_t gi = ray0 + (e.x & id_mask);
                    const RayGeom q = ssd_load_ray(rays_o + 3 * gi, rays_d + 3 * gi);
                    float qn, qf;
                    ssd_near_far(c.aabb, q, c.min_near, qn, qf);
                    qf = sm_tail_far(c.m, q, qn, qf, e.x, packing);
                    const float qt = __uint_as_float(e.y);
                    const ProbeB p = sm_probe(c.m, lin_bits, q, qt);      // the queued t is an occupied probe by construction
                    float4* dst = stage + lane * 4;
                    dst[0] = make_float4(__uint_as_float(e.x), qt, qf, p.dt);
                    dst[1] = make_float4(q.ox, q.oy, q.oz, q.dx);
                    dst[2] = make_float4(q.dy, q.dz, q.rdx, q.rdy);
                    dst[3] = make_float4(q.rdz, p.x, p.y, p.z);
                }
                next = __builtin_amdgcn_readfirstlane(next + n);
                st_head = 0; st_count = n;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t take = min((uint32_t)__popcll(idle), st_count);
            if (ray < 0 && rank < take) {
                const float4* src = stage + (st_head + rank) * 4;
                const float4 g0 = src[0], g1 = src[1], g2 = src[2], g3 = src[3];
                ray = (int)__float_as_uint(g0.x); t = g0.y; far_ = g0.z; sdt = g0.w;
                r.ox = g1.x; r.oy = g1.y; r.oz = g1.z; r.dx = g1.w; r.dy = g2.x; r.dz = g2.y; r.rdx = g2.z; r.rdy = g2.w; r.rdz = g3.x;
                sx = g3.y; sy = g3.z; sz = g3.w;
                sgx = ssd_fma(0.5f, ssd_sign1(r.dx), 0.5f); sgy = ssd_fma(0.5f, ssd_sign1(r.dy), 0.5f); sgz = ssd_fma(0.5f, ssd_sign1(r.dz), 0.5f);
                ws = dep = cr = cg = cb = 0.f; cnt = 0;
                float sh[16];
                shb::eval<4, false>(r.dx, r.dy, r.dz, sh, nullptr, nullptr, nullptr);
                if constexpr (VAR == 4 || VAR == 6) {
                    uint32_t tm[3][16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) sm_split3(sh[k], tm[0][k], tm[1][k], tm[2][k]);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        uint4* dst = reinterpret_cast<uint4*>(sh_lds) + t * 128 + lane * 2;
                        dst[0] = make_uint4(sm_pack2(tm[t][0], tm[t][1]), sm_pack2(tm[t][2], tm[t][3]), sm_pack2(tm[t][4], tm[t][5]), sm_pack2(tm[t][6], tm[t][7]));
                        dst[1] = make_uint4(sm_pack2(tm[t][8], tm[t][9]), sm_pack2(tm[t][10], tm[t][11]), sm_pack2(tm[t][12], tm[t][13]), sm_pack2(tm[t][14], tm[t][15]));
                    }
                } else if constexpr (VAR == 3) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) reinterpret_cast<float*>(sh_lds)[k * 64 + lane] = sh[k];
                } else {
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) sh_lds[kq * 64 + lane] = make_float4(sh[4 * kq], sh[4 * kq + 1], sh[4 * kq + 2], sh[4 * kq + 3]);
                }
            }
            st_head += take; st_count -= take;
        }
        const uint64_t live = __ballot(ray >= 0);

        // ================= march pass: every lane takes one parked ray to its next hit or to the end of the box =================
        if ((sp_count >= SM_MARCH_W || (live == 0 && sp_count != 0)) && rp_count <= SM_POOL - SM_MARCH_W) {
            const uint32_t n = min(sp_count, SM_MARCH_W);
            bool found = false, mine = (uint32_t)lane < n;
            uint4 e0 = make_uint4(0, 0, 0, 0), e1 = make_uint4(0, 0, 0, 0);
            if (mine) {
                const uint32_t* e = pool_search + ((sp_head + lane) % SM_POOL) * 8;
                e0 = *reinterpret_cast<const uint4*>(e); e1 = *reinterpret_cast<const uint4*>(e + 4);
                const uint64_t gi = ray0 + (e0.x & id_mask);
                const RayGeom q = ssd_load_ray(rays_o + 3 * gi, rays_d + 3 * gi);
                float qn, qf;
                ssd_near_far(c.aabb, q, c.min_near, qn, qf);
                qf = sm_tail_far(c.m, q, qn, qf, e0.x, packing);
                const float qx = ssd_fma(0.5f, ssd_sign1(q.dx), 0.5f), qy = ssd_fma(0.5f, ssd_sign1(q.dy), 0.5f), qz = ssd_fma(0.5f, ssd_sign1(q.dz), 0.5f);
                float qt = __uint_as_float(e0.y);
                while (qt < qf) {
                    const ProbeB p = sm_probe(c.m, lin_bits, q, qt);
                    if (p.occ) { found = true; break; }
                    qt = sm_skip(c.m, q, p, qx, qy, qz, qt);
                }
                if (found) e0.y = __float_as_uint(qt);
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            continue;   // refill from the ready pool before shading
        }
        if (live == 0) {
            if (scene_done && next >= end && st_count == 0 && sp_count == 0 && rp_count == 0) break;      // this scene is finished for this wave: go steal from the next one
            continue;
        }

        // ================= shade: gather -> MFMA layers -> output layer -> composite =================
        float f[18];
        if (ray >= 0) ssd_gather18<PT, VAR != 2>(planes, c.g, sx, sy, sz, f);
        else {
#pragma unroll
            for (int i = 0; i < 18; ++i) f[i] = 0.f;
        }
        // Software-pipelined by sample tile so that (almost) every MFMA has independent VALU work to hide under, inside this
        // wave: the f32 MFMA occupies the matrix pipe for 64 cycles, and an in-order wave that issues MFMAs back to back just
        // waits (r01 counters: SQ_WAIT_INST_ANY = 2.4x the MFMA time, both waves of a SIMD colliding in their MFMA bursts).
        //   A: layer-1 tile 0            (20 MFMA)
        //   B: layer-1 tile 1            (20 MFMA)  ||  density head of tile 0   (16 SiLU pairs)
        //   C: direction term tile 0     (18 MFMA)  ||  density head of tile 1
        //   D: direction term tile 1     (18 MFMA)  ||  colour head of tile 0
        //   E: colour head of tile 1
        float ps0, ps1, pr0, pr1, pg0, pg1, pb0, pb1;            // per tile: this lane half's share of (sigma, r, g, b) pre-activations
        if constexpr (VAR == 4) {
            // ---- bf16 x 3 on the matrix cores.  Split every feature into three bf16 terms, pack feature pairs, and trade halves so that
            // T[t][0..3] is tile 0's k-step-0 operand (features 0-15 of the samples of lanes 0-31) and T[t][4..7] tile 1's; T[t][8] / Z[t]
            // carry features 16, 17 for k-step 1 (their other k slots are the bias row and zeros)
            uint32_t T[3][9], Z[3] = {0u, 0u, 0u};
#pragma unroll
            for (int p2 = 0; p2 < 9; ++p2) {
                uint32_t h0, m0, l0, h1, m1, l1;
                sm_split3(f[2 * p2], h0, m0, l0);
                sm_split3(f[2 * p2 + 1], h1, m1, l1);
                T[0][p2] = sm_pack2(h0, h1); T[1][p2] = sm_pack2(m0, m1); T[2][p2] = sm_pack2(l0, l1);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sm_swap_u(T[t][k], T[t][4 + k]);
                sm_swap_u(T[t][8], Z[t]);
            }
            const uint32_t bias_pair = half == 0 ? 0x00003F80u : 0u;          // {bf16(1.0), 0}: the bias row of k-step 1 (k = 18 resp. 16), lane half 0 only
            const sm_bf16x8 b_bias = sm_op(bias_pair, 0u, 0u, 0u);            // dir layer, k-step 1: [1, 0, ...]
            float res[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                floatx16 acc[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
                sm_bf16x8 b0[3], b1[3];                                      // B operands of k-step 0 / 1, terms hi, mid, lo
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    b0[t] = sm_op(T[t][4 * nt], T[t][4 * nt + 1], T[t][4 * nt + 2], T[t][4 * nt + 3]);
                    b1[t] = sm_op(nt == 0 ? T[t][8] : Z[t], t == 0 ? bias_pair : 0u, 0u, 0u);
                }
                // h = W1 [f; 1]: products (weight term i) x (feature term j), i + j <= 2, smallest first
#pragma unroll
                for (int pr_i = 0; pr_i < 6; ++pr_i) {
                    constexpr int TI[6] = {2, 1, 0, 1, 0, 0}, TJ[6] = {0, 1, 2, 0, 1, 0};
                    const int i = TI[pr_i], j = TJ[pr_i];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1[mt][0][i], b0[j], acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1[mt][1][i], b1[j], acc[mt], 0, 0, 0);
                    }
                }
                floatx2 ps = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) {                              // density head on silu(h)
                    const int mt = q >> 3, p2 = q & 7;
                    const float4 w = wout2[((mt * 8 + p2) * 2 + half) * 2];
                    ps = sm_fma2(floatx2{w.x, w.y}, sm_silu2(floatx2{acc[mt][2 * p2], acc[mt][2 * p2 + 1]}), ps);
                    if ((q & SM_HEAD_GROUP) == SM_HEAD_GROUP) __builtin_amdgcn_sched_barrier(0);    // four pairs at a time (see VAR 3)
                }
                // h += Wd [SH(d); 1]: SH operands pre-split per ray in LDS, [term][sample][k 0-7 | k 8-15]
                sm_bf16x8 sb[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) sb[t] = *reinterpret_cast<const sm_bf16x8*>(reinterpret_cast<const uint4*>(sh_lds) + t * 128 + (nt * 32 + (lane & 31)) * 2 + half);
#pragma unroll
                for (int pr_i = 0; pr_i < 6; ++pr_i) {
                    constexpr int TI[6] = {2, 1, 0, 1, 0, 0}, TJ[6] = {0, 1, 2, 0, 1, 0};
                    const int i = TI[pr_i], j = TJ[pr_i];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2[mt][0][i], sb[j], acc[mt], 0, 0, 0);
                        if (j == 0) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2[mt][1][i], b_bias, acc[mt], 0, 0, 0);   // + bd (exactly: 1.0 has one term)
                    }
                }
                floatx2 pr = {0.f, 0.f}, pg = {0.f, 0.f}, pb = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) {                              // colour head on silu(h + hd)
                    const int mt = q >> 3, p2 = q & 7;
                    const float4 w0 = wout2[((mt * 8 + p2) * 2 + half) * 2], w1 = wout2[((mt * 8 + p2) * 2 + half) * 2 + 1];
                    const floatx2 cc = sm_silu2(floatx2{acc[mt][2 * p2], acc[mt][2 * p2 + 1]});
                    pr = sm_fma2(floatx2{w0.z, w0.w}, cc, pr);
                    pg = sm_fma2(floatx2{w1.x, w1.y}, cc, pg);
                    pb = sm_fma2(floatx2{w1.z, w1.w}, cc, pb);
                    if ((q & SM_HEAD_GROUP) == SM_HEAD_GROUP) __builtin_amdgcn_sched_barrier(0);
                }
                res[nt][0] = ps.x + ps.y; res[nt][1] = pr.x + pr.y; res[nt][2] = pg.x + pg.y; res[nt][3] = pb.x + pb.y;
            }
            ps0 = res[0][0]; ps1 = res[1][0]; pr0 = res[0][1]; pr1 = res[1][1]; pg0 = res[0][2]; pg1 = res[1][2]; pb0 = res[0][3]; pb1 = res[1][3];
        } else if constexpr (VAR == 6) {
            // ---- variant 4's operands (bf16 x 3 on the matrix cores, same products in the same order per accumulator: bit-identical
            // results) in variant 2's schedule: both sample tiles hold their accumulators (64 registers instead of 32), and every MFMA
            // group has SiLU pairs of the OTHER tile behind it in program order, so the wave keeps issuing VALU work while its own MFMAs
            // run (variant 4 leaves that to the second wave of the SIMD).  Needs the register headroom of SSD_GATHER_PAIRS=1.
            //   A: layer 1, tile 0 (24 MFMA)   B: layer 1, tile 1 (24) || density head 0   C: dir term 0 (18) || density head 1
            //   D: dir term 1 (18) || colour head 0   E: colour head 1
            uint32_t T[3][9], Z[3] = {0u, 0u, 0u};
#pragma unroll
            for (int p2 = 0; p2 < 9; ++p2) {
                uint32_t h0, m0, l0, h1, m1, l1;
                sm_split3(f[2 * p2], h0, m0, l0);
                sm_split3(f[2 * p2 + 1], h1, m1, l1);
                T[0][p2] = sm_pack2(h0, h1); T[1][p2] = sm_pack2(m0, m1); T[2][p2] = sm_pack2(l0, l1);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sm_swap_u(T[t][k], T[t][4 + k]);
                sm_swap_u(T[t][8], Z[t]);
            }
            const uint32_t bias_pair = half == 0 ? 0x00003F80u : 0u;
            const sm_bf16x8 b_bias = sm_op(bias_pair, 0u, 0u, 0u);
            floatx16 acc[2][2];                                              // [tile][mt]
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.0f;
            constexpr int TI[6] = {2, 1, 0, 1, 0, 0}, TJ[6] = {0, 1, 2, 0, 1, 0};
            auto layer1 = [&](int nt, int pr_i) {                            // 4 MFMA: products (weight term i) x (feature term j) of both row tiles
                const int i = TI[pr_i], j = TJ[pr_i];
                const sm_bf16x8 b0 = sm_op(T[j][4 * nt], T[j][4 * nt + 1], T[j][4 * nt + 2], T[j][4 * nt + 3]);
                const sm_bf16x8 b1 = sm_op(nt == 0 ? T[j][8] : Z[j], j == 0 ? bias_pair : 0u, 0u, 0u);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1[mt][0][i], b0, acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1[mt][1][i], b1, acc[nt][mt], 0, 0, 0);
                }
            };
            sm_bf16x8 sb[3];
            auto load_sh = [&](int nt) {
#pragma unroll
                for (int t = 0; t < 3; ++t) sb[t] = *reinterpret_cast<const sm_bf16x8*>(reinterpret_cast<const uint4*>(sh_lds) + t * 128 + (nt * 32 + (lane & 31)) * 2 + half);
            };
            auto dir_term = [&](int nt, int pr_i) {                          // 2 or 4 MFMA: h += Wd [SH(d); 1]
                const int i = TI[pr_i], j = TJ[pr_i];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2[mt][0][i], sb[j], acc[nt][mt], 0, 0, 0);
                    if (j == 0) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2[mt][1][i], b_bias, acc[nt][mt], 0, 0, 0);
                }
            };
            floatx2 ps_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}};
            floatx2 pr_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}}, pg_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}}, pb_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}};
            auto density_pair = [&](int nt, int q) {
                const int mt = q >> 3, p2 = q & 7;
                const float4 w = wout2[((mt * 8 + p2) * 2 + half) * 2];
                ps_[nt] = sm_fma2(floatx2{w.x, w.y}, sm_silu2(floatx2{acc[nt][mt][2 * p2], acc[nt][mt][2 * p2 + 1]}), ps_[nt]);
            };
            auto colour_pair = [&](int nt, int q) {
                const int mt = q >> 3, p2 = q & 7;
                const float4 w0 = wout2[((mt * 8 + p2) * 2 + half) * 2], w1 = wout2[((mt * 8 + p2) * 2 + half) * 2 + 1];
                const floatx2 cc = sm_silu2(floatx2{acc[nt][mt][2 * p2], acc[nt][mt][2 * p2 + 1]});
                pr_[nt] = sm_fma2(floatx2{w0.z, w0.w}, cc, pr_[nt]);
                pg_[nt] = sm_fma2(floatx2{w1.x, w1.y}, cc, pg_[nt]);
                pb_[nt] = sm_fma2(floatx2{w1.z, w1.w}, cc, pb_[nt]);
            };
            constexpr int QB[7] = {0, 3, 6, 9, 12, 14, 16};                 // 16 SiLU pairs spread over the 6 MFMA groups of a phase
            // ---- A
#pragma unroll
            for (int g = 0; g < 6; ++g) layer1(0, g);
            // ---- B
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                layer1(1, g);
#pragma unroll
                for (int q = QB[g]; q < QB[g + 1]; ++q) density_pair(0, q);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- C
            load_sh(0);
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                dir_term(0, g);
#pragma unroll
                for (int q = QB[g]; q < QB[g + 1]; ++q) density_pair(1, q);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- D
            load_sh(1);
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                dir_term(1, g);
#pragma unroll
                for (int q = QB[g]; q < QB[g + 1]; ++q) colour_pair(0, q);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- E
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                colour_pair(1, q);
                if ((q & SM_HEAD_GROUP) == SM_HEAD_GROUP) __builtin_amdgcn_sched_barrier(0);
            }
            ps0 = ps_[0].x + ps_[0].y; ps1 = ps_[1].x + ps_[1].y; pr0 = pr_[0].x + pr_[0].y; pr1 = pr_[1].x + pr_[1].y;
            pg0 = pg_[0].x + pg_[0].y; pg1 = pg_[1].x + pg_[1].y; pb0 = pb_[0].x + pb_[0].y; pb1 = pb_[1].x + pb_[1].y;
        } else if constexpr (VAR == 3) {
            // ---- three waves per SIMD: the tiles are shaded one after the other, A operands come from LDS, B operands are built in place ----
            // after the swaps f[2s] feeds tile 0 (the samples of lanes 0-31) and f[2s+1] tile 1, k-step s; the SH operands sit in LDS in that form
#pragma unroll
            for (int s = 0; s < 9; ++s) sm_swap(f[2 * s], f[2 * s + 1]);
            const float* sh_op = reinterpret_cast<const float*>(sh_lds) + half * 64 + (lane & 31);   // + (2s)*64 + nt*32: SH_{2s+half} of sample nt*32 + (lane & 31)
            float res[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // the A operands are the same for both tiles; the laundered lane offset keeps the compiler from loading them once and holding
                // all 38 of them in registers across the two tiles (which is exactly the register budget this variant exists to avoid)
                uint32_t wl = (uint32_t)lane;
                asm volatile("" : "+v"(wl));
                const float* wA = w_lds + wl;
                floatx16 acc[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
#pragma unroll
                for (int s = 0; s < 10; ++s) {                              // h = W1 [f; 1]
                    const float bop = s < 9 ? f[2 * s + nt] : b_const;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[(0 * 10 + s) * 64], bop, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[(1 * 10 + s) * 64], bop, acc[1], 0, 0, 0);
                }
                floatx2 ps = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) {                              // density head on silu(h)
                    const int mt = q >> 3, p2 = q & 7;
                    const float4 w = wout2[((mt * 8 + p2) * 2 + half) * 2];
                    ps = sm_fma2(floatx2{w.x, w.y}, sm_silu2(floatx2{acc[mt][2 * p2], acc[mt][2 * p2 + 1]}), ps);
                    if (q & 1) __builtin_amdgcn_sched_barrier(0);           // two pairs at a time: the scheduler otherwise batches all 32 exp/rcp and spills their results
                }
#pragma unroll
                for (int s = 0; s < 9; ++s) {                               // h += Wd [SH(d); 1], in place
                    const float bop = s < 8 ? sh_op[2 * s * 64 + nt * 32] : b_const;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[1280 + (0 * 9 + s) * 64], bop, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[1280 + (1 * 9 + s) * 64], bop, acc[1], 0, 0, 0);
                }
                floatx2 pr = {0.f, 0.f}, pg = {0.f, 0.f}, pb = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) {                              // colour head on silu(h + hd)
                    const int mt = q >> 3, p2 = q & 7;
                    const float4 w0 = wout2[((mt * 8 + p2) * 2 + half) * 2], w1 = wout2[((mt * 8 + p2) * 2 + half) * 2 + 1];
                    const floatx2 cc = sm_silu2(floatx2{acc[mt][2 * p2], acc[mt][2 * p2 + 1]});
                    pr = sm_fma2(floatx2{w0.z, w0.w}, cc, pr);
                    pg = sm_fma2(floatx2{w1.x, w1.y}, cc, pg);
                    pb = sm_fma2(floatx2{w1.z, w1.w}, cc, pb);
                    if (q & 1) __builtin_amdgcn_sched_barrier(0);
                }
                res[nt][0] = ps.x + ps.y; res[nt][1] = pr.x + pr.y; res[nt][2] = pg.x + pg.y; res[nt][3] = pb.x + pb.y;
            }
            ps0 = res[0][0]; ps1 = res[1][0]; pr0 = res[0][1]; pr1 = res[1][1]; pg0 = res[0][2]; pg1 = res[1][2]; pb0 = res[0][3]; pb1 = res[1][3];
        } else {
            floatx16 acc[2][2];
    #pragma unroll
            for (int mt = 0; mt < 2; ++mt)
    #pragma unroll
                for (int nt = 0; nt < 2; ++nt)
    #pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;
            float fb0[10], fb1[10];                 // B operands of layer 1 for tile 0 / tile 1
    #pragma unroll
            for (int s = 0; s < 9; ++s) {
                fb0[s] = f[2 * s]; fb1[s] = f[2 * s + 1];
                sm_swap(fb0[s], fb1[s]);
            }
            fb0[9] = fb1[9] = b_const;
            floatx2 ps_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}};
            floatx2 pr_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}}, pg_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}}, pb_[2] = {floatx2{0.f, 0.f}, floatx2{0.f, 0.f}};
            auto density_pair = [&](int nt, int q) {     // q in [0,16): accumulator pair (mt = q / 8, registers 2*(q%8), 2*(q%8)+1)
                const int mt = q >> 3, p2 = q & 7;
                const float4 w = wout2[((mt * 8 + p2) * 2 + half) * 2];
                const floatx2 wS = {w.x, w.y};
                ps_[nt] = sm_fma2(wS, sm_silu2(floatx2{acc[mt][nt][2 * p2], acc[mt][nt][2 * p2 + 1]}), ps_[nt]);
            };
            auto colour_pair = [&](int nt, int q) {
                const int mt = q >> 3, p2 = q & 7;
                const float4 w0 = wout2[((mt * 8 + p2) * 2 + half) * 2], w1 = wout2[((mt * 8 + p2) * 2 + half) * 2 + 1];
                const floatx2 wR = {w0.z, w0.w}, wG = {w1.x, w1.y}, wB = {w1.z, w1.w};
                const floatx2 cc = sm_silu2(floatx2{acc[mt][nt][2 * p2], acc[mt][nt][2 * p2 + 1]});
                pr_[nt] = sm_fma2(wR, cc, pr_[nt]);
                pg_[nt] = sm_fma2(wG, cc, pg_[nt]);
                pb_[nt] = sm_fma2(wB, cc, pb_[nt]);
            };
            // ---- A
    #pragma unroll
            for (int s = 0; s < 10; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0][s], fb0[s], acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1][s], fb0[s], acc[1][0], 0, 0, 0);
            }
            // SH operands (per-ray constants parked in LDS)
            float4 shq[4];
    #pragma unroll
            for (int kq = 0; kq < 4; ++kq) shq[kq] = sh_lds[kq * 64 + lane];
            const float* sh = reinterpret_cast<const float*>(shq);
            float sb0[9], sb1[9];
    #pragma unroll
            for (int s = 0; s < 8; ++s) {
                sb0[s] = sh[2 * s]; sb1[s] = sh[2 * s + 1];
                sm_swap(sb0[s], sb1[s]);
            }
            sb0[8] = sb1[8] = b_const;
            // ---- B
    #pragma unroll
            for (int s = 0; s < 10; ++s) {
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0][s], fb1[s], acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1][s], fb1[s], acc[1][1], 0, 0, 0);
                if (s < 8) { density_pair(0, 2 * s); density_pair(0, 2 * s + 1); }
            }
            // ---- C
    #pragma unroll
            for (int s = 0; s < 9; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][s], sb0[s], acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][s], sb0[s], acc[1][0], 0, 0, 0);
                if (s < 8) { density_pair(1, 2 * s); density_pair(1, 2 * s + 1); }
            }
            // ---- D
    #pragma unroll
            for (int s = 0; s < 9; ++s) {
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][s], sb1[s], acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][s], sb1[s], acc[1][1], 0, 0, 0);
                if (s < 8) { colour_pair(0, 2 * s); colour_pair(0, 2 * s + 1); }
            }
            // ---- E
    #pragma unroll
            for (int q = 0; q < 16; ++q) colour_pair(1, q);
            ps0 = ps_[0].x + ps_[0].y; ps1 = ps_[1].x + ps_[1].y;
            pr0 = pr_[0].x + pr_[0].y; pr1 = pr_[1].x + pr_[1].y; pg0 = pg_[0].x + pg_[0].y; pg1 = pg_[1].x + pg_[1].y;
            pb0 = pb_[0].x + pb_[0].y; pb1 = pb_[1].x + pb_[1].y;
        }
        // cross-half reduction: after the swap, (x0 + x1) on lane l is the total for sample l
        sm_swap(ps0, ps1); sm_swap(pr0, pr1); sm_swap(pg0, pg1); sm_swap(pb0, pb1);
        const float sigma = ssd_exp(ps0 + ps1 + b_sigma);
        const float sr = ssd_fma(ssd_sigmoid(pr0 + pr1 + bc0), sat_k, -c.sat);
        const float sg = ssd_fma(ssd_sigmoid(pg0 + pg1 + bc1), sat_k, -c.sat);
        const float sb = ssd_fma(ssd_sigmoid(pb0 + pb1 + bc2), sat_k, -c.sat);

        bool park = false;
        if (ray >= 0) {
            const float alpha = 1.0f - __expf(-sigma * sdt);
            const float T = 1.0f - ws;
            const float w = alpha * T;
            ws += w;
            dep = ssd_fma(w, t, dep);
            cr = ssd_fma(w, sr, cr);
            cg = ssd_fma(w, sg, cg);
            cb = ssd_fma(w, sb, cb);
            t += sdt;
            ++cnt;
            if (T < c.T_thresh) {
                write_out((uint32_t)ray, ws, dep, cr, cg, cb, cnt); ray = -1;
            } else {
                uint32_t probes = 0;
                for (;;) {
                    if (!(t < far_)) { write_out((uint32_t)ray, ws, dep, cr, cg, cb, cnt); ray = -1; break; }
                    if (cnt >= c.cap) {
                        if (overflow_flag) atomicAdd(overflow_flag, 1);
                        write_out((uint32_t)ray, ws, dep, cr, cg, cb, cnt); ray = -1; break;
                    }
                    if (probes == SM_SEARCH_PROBES) { park = true; break; }
                    const ProbeB p = sm_probe(c.m, lin_bits, r, t);
                    if (p.occ) { sx = p.x; sy = p.y; sz = p.z; sdt = p.dt; break; }
                    t = sm_skip(c.m, r, p, sgx, sgy, sgz, t);
                    ++probes;
                }
            }
        }
        // ---- park rays that need a long search (state -> LDS search pool) ----
        const uint64_t pm = __ballot(park);
        if (pm != 0) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
            const uint32_t room = SM_POOL - sp_count;
            if (park && rank < room) {
                uint32_t* e = pool_search + ((sp_head + sp_count + rank) % SM_POOL) * 8;
                *reinterpret_cast<uint4*>(e) = make_uint4((uint32_t)ray, __float_as_uint(t), __float_as_uint(ws), __float_as_uint(dep));
                *reinterpret_cast<uint4*>(e + 4) = make_uint4(__float_as_uint(cr), __float_as_uint(cg), __float_as_uint(cb), cnt);
                ray = -1;
                park = false;
            }
            sp_count += min((uint32_t)__popcll(pm), room);
            if (park) {   // pool full (cannot happen with the trigger above, kept as a guard): finish the search in the lane
                for (;;) {
                    if (!(t < far_)) { write_out((uint32_t)ray, ws, dep, cr, cg, cb, cnt); ray = -1; break; }
                    const ProbeB p = sm_probe(c.m, lin_bits, r, t);
                    if (p.occ) { sx = p.x; sy = p.y; sz = p.z; sdt = p.dt; break; }
                    t = sm_skip(c.m, r, p, sgx, sgy, sgz, t);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
  }   // scene loop
}

// ------------------------------------------------------------------------------------------------
size_t ssdnerf_render_queue_workspace(uint32_t S, uint32_t N, uint32_t grid_size);   // render_queue.hip (same workspace layout)

extern "C" int ssdnerf_render_shade_queue_mfma(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                               uint32_t grid_size, const float* rays_o, const float* rays_d, uint32_t S, uint32_t N, float bound,
                                               float min_near, float dt_gamma, const float* dt_gammas, uint32_t max_steps, float T_thresh,
                                               float bg_color, float sigmoid_saturation, float* image, float* depth, float* weights_sum,
                                               int32_t* sample_counts, int32_t* overflow_flag, void* workspace, size_t workspace_bytes,
                                               void* stream) {
    if (N == 0 || S == 0) return SSDNERF_OK;
    SSD_REQUIRE(planes && mlp_params && rays_o && rays_d && image && depth && weights_sum && workspace, "render_shade_queue_mfma: null pointer");
    SSD_REQUIRE(planes_dtype == 0 || planes_dtype == 1, "render_shade_queue_mfma: unsupported plane dtype");
    SSD_REQUIRE(grid_size >= 8 && grid_size <= 512 && (grid_size & (grid_size - 1)) == 0, "render_shade_queue_mfma: grid_size must be a power of two in [8, 512]");
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
    // workspace carve (must match render_queue.hip)
    const size_t counters = ((size_t)S * 8 + 255) / 256 * 256;
    const size_t bits = ((size_t)S * c.bitfield_stride + 255) / 256 * 256;
    uint32_t* q_count = (uint32_t*)workspace;
    const uint8_t* lin_bits = (const uint8_t*)workspace + counters;
    const size_t hc = grid_size / 2, coarse = ((size_t)S * (hc * hc * hc / 8) + 255) / 256 * 256;   // k_first_hit's coarse occupancy
    const uint2* queue = (const uint2*)((const uint8_t*)workspace + counters + bits + coarse);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const uint32_t slices = 0;   // (kept in the kernel signature; slices are ticketed dynamically)
    // residency: WPS workgroups x 4 waves per CU, persistent.  SSDNERF_SHADE_WPS=2|3 picks the variant (default below).
    static int var = 0;                                  // 2: f32 MFMA, 2 waves/SIMD; 3: f32 MFMA, 3 waves/SIMD; 4: bf16 x 3 on the matrix cores;
                                                         // 6: variant 4's arithmetic in variant 2's interleaved schedule (experimental: not yet run on hardware)
    if (var == 0) {
        const char* e = getenv("SSDNERF_SHADE_VARIANT");
        var = (e && ((e[0] >= '2' && e[0] <= '4') || e[0] == '6')) ? e[0] - '0' : SM_DEFAULT_VARIANT;
    }
    dim3 g((unsigned)n_cu * (var == 3 ? 3u : 2u)), b(SM_TPB);
    hipStream_t s = (hipStream_t)stream;
#define SM_LAUNCH(PT, V) hipLaunchKernelGGL((k_shade_mfma<PT, V>), g, b, 0, s, c, slices, (const PT*)planes, mlp_params, lin_bits, rays_o, rays_d, queue, q_count, image, depth, weights_sum, sample_counts, overflow_flag)
    if (planes_dtype == 0) { if (var == 4) SM_LAUNCH(float, 4); else if (var == 6) SM_LAUNCH(float, 6); else if (var == 3) SM_LAUNCH(float, 3); else SM_LAUNCH(float, 2); }
    else { if (var == 4) SM_LAUNCH(__half, 4); else if (var == 6) SM_LAUNCH(__half, 6); else if (var == 3) SM_LAUNCH(__half, 3); else SM_LAUNCH(__half, 2); }
#undef SM_LAUNCH
    SSD_CHECK_LAUNCH("render_shade_queue_mfma");
    return SSDNERF_OK;
}
