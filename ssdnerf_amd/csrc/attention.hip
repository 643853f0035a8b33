// ssdnerf_amd/csrc/attention.hip -- the denoising UNet's self-attention (MultiHeadAttentionMod.QKVAttention,
// lib/models/architecture/ddpm/modules.py:12-48 + mmgen's QKVAttention) on the bf16 matrix cores.
//
// Input is the qkv projection of the channel-last activation, [B][T][3C] with the reference's per-head channel order
// [head][q | k | v][ch]; output is [B][T][C], channel = head*ch + i (what the proj GEMM consumes).  Cars layout: 16 sites per forward,
// T = 1024 (ch 64), 256 and 64 (ch 128), B*heads = 32 independent problems each; tiled layout: T = 768 / 192 / 48, ch = 40 / 80.
//
// Flash-style, one pass over the keys, nothing T x T ever leaves the registers:
//   * a wave owns 32 queries; per block of 32 keys it computes  S^T = K Q^T  with v_mfma_f32_32x32x16_bf16 -- transposed on
//     purpose: in the MFMA's C layout a lane then holds ONE query (column) and 16 keys (registers), so the running max / sum of
//     the online softmax are per-lane scalars (one cross-half shuffle per block) and rescaling O is a per-lane multiply;
//   * P = exp2((S - m) * scale*log2e) is rounded to bf16 in registers and is *already* the B operand of  O^T += V^T P : the
//     C-layout's key order per lane half ({0-3, 8-11} / {4-7, 12-15} of every 16) is simply adopted as the k-slot order, and the
//     V tile is written to LDS transposed in that same order, so no permute or LDS round trip of P is needed;
//   * K and V of a key block go through LDS once per block of four waves, already as operand terms (r04; K used to be loaded -- and, fp32, split -- by
//     every wave into its own registers): K row-major (its A fragments are one ds_read_b128 each), V transposed; global -> registers at the head of a
//     key block, registers -> the other LDS buffer behind its products, one barrier per key block;
//   * softmax statistics in fp32, O accumulated in fp32, divided by the row sum at the end.
//
// Two element types share the kernel (template F32):
//   bf16 : operands as stored; P rounded to bf16 (the reference's `.type(weight.dtype)` under autocast).
//   fp32 : the fp32 configs (no autocast).  q, k, v and P are each split into a bf16 pair (hi = bf16(x), lo = bf16(x - hi)) and every product
//          is hi*hi + hi*lo + lo*hi accumulated in fp32 -- >= 16 significand bits per factor, the arithmetic class of the fp32 convolutions
//          (k_conv_igemm_f32x2); three MFMAs where the bf16 form has one, still < 3 % of the UNet's matrix work.
// Shapes: any T >= 1 (keys past T are masked to -inf / zero V rows, queries past T are not stored) and any head width ch that is a
// multiple of 8 up to 128 (the kernel is instantiated for the padded widths 32 / 64 / 96 / 128; channels past ch read as zeros).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int AT_ROW = 80;                 // bytes per LDS row of V^T (32 keys * 2 B + 16 B pad: conflict-free ds_read_b128)

// {bf16(lo), bf16(hi)}, round to nearest even: gfx950's v_cvt_pk_bf16_f32, one instruction per pair (r03; the integer form was five per element --
// the operand splits of the fp32-class kernel were ~90 VALU instructions per 8 values, now 24)
SSD_DEV uint32_t at_pack_bf16(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 r = __builtin_convertvector(f2{lo, hi}, b2);
    return *reinterpret_cast<const uint32_t*>(&r);
}
// the pair's lo terms: bf16 of what the hi terms left over
SSD_DEV uint32_t at_pack_bf16_rest(float lo, float hi, uint32_t packed_hi) {
    return at_pack_bf16(lo - __uint_as_float(packed_hi << 16), hi - __uint_as_float(packed_hi & 0xffff0000u));
}

// 8 fp32 values -> bf16 hi and lo operand vectors (x ~= hi + lo to >= 16 significand bits)
SSD_DEV void at_split8(const float4& a, const float4& b, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = at_pack_bf16(v[2 * e], v[2 * e + 1]);
        l[e] = at_pack_bf16_rest(v[2 * e], v[2 * e + 1], h[e]);
    }
    const uint4 uh = make_uint4(h[0], h[1], h[2], h[3]);
    const uint4 ul = make_uint4(l[0], l[1], l[2], l[3]);
    hi = *reinterpret_cast<const bf16x8*>(&uh);
    lo = *reinterpret_cast<const bf16x8*>(&ul);
}

// position of key kk (0..31) inside a V^T row: per 16 keys, lane half 0 owns {0-3, 8-11}, half 1 owns {4-7, 12-15} (MFMA C layout)
SSD_DEV uint32_t at_key_pos(uint32_t kk) {
    const uint32_t w = kk & 15u;
    return (kk & 16u) + ((w >> 2) & 1u) * 8u + (w & 3u) + 4u * (w >> 3);
}

// WPB: waves (= 32-query tiles) per block; 4 is what at_launch uses (see there), 2 / 1 exist for A/B runs.
// KG (r04): key groups.  With KG = 2 a block has 2 x WPB waves: wave w and wave w + WPB own the SAME 32 queries and walk the even / the odd key blocks, each with
// its own running max / sum / O, merged once at the end through LDS -- two waves per SIMD that stall independently instead of one that exposes every latency
// (the cars shapes give 8 scenes x 4 heads x T / 32 query tiles = 1024 waves at T = 1024: ONE per SIMD; the loop was 5 k cycles per key block for 0.8 k of MFMA).
template <int CHP, bool F32, int WPB = 4, int KG = 1>
__global__ __launch_bounds__(64 * WPB * KG) void k_attn_fwd(const unsigned char* __restrict__ qkv, unsigned char* __restrict__ out, uint32_t T, uint32_t heads,
                                                  uint32_t ch, float scale_log2e, float* __restrict__ lse2) {
    constexpr int KS = CHP / 16;           // k-steps of the QK^T product
    constexpr int CT = CHP / 32;           // 32-channel tiles of O
    constexpr int G = F32 ? 4 : 8;         // channels per staged V item and row (16 bytes)
    constexpr int VTOT = 16 * CHP / G;     // V items per key block: (pair of adjacent keys, G-channel group)
    constexpr int NTH = 64 * WPB;          // threads per key group (the staging of a key block is theirs)
    constexpr int VCH = (VTOT + NTH - 1) / NTH; // ... per thread
    constexpr int ES = F32 ? 4 : 2;        // bytes per stored element
    constexpr int NT = F32 ? 2 : 1;        // operand terms (hi, lo)
    constexpr int KROW = CHP * 2 + 16;     // bytes per LDS row of K (one key's CHP bf16 terms + 16 B pad: the 32 rows of a ds_read_b128 fall on different banks)
    constexpr int KTOT = 32 * CHP / 8;     // K items per key block: (key, 8-channel chunk)
    constexpr int KCH = (KTOT + NTH - 1) / NTH;
    constexpr int VT_BYTES = 2 * KG * NT * CHP * AT_ROW, KT_BYTES = 2 * KG * NT * 32 * KROW;
    constexpr int MERGE_BYTES = (KG - 1) * WPB * 64 * (CT * 16 + 2) * 4;              // the other key groups' (O, max, sum) per lane, after the loop
    constexpr int LDS_BYTES = VT_BYTES + KT_BYTES > MERGE_BYTES ? VT_BYTES + KT_BYTES : MERGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[LDS_BYTES];           // (ONE LDS object)
    auto vt = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + ((buf * KG + g) * NT + tm) * (CHP * AT_ROW); };
    auto kt = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + VT_BYTES + ((buf * KG + g) * NT + tm) * (32 * KROW); };

    const uint32_t lane = threadIdx.x & 63, wave_all = threadIdx.x >> 6, kg = wave_all / WPB, wave = wave_all % WPB, tid = threadIdx.x - kg * NTH;   // tid: inside the key group
    const uint32_t hf = lane >> 5, l31 = lane & 31;
    const uint32_t b = blockIdx.y / heads, h = blockIdx.y % heads, C = heads * ch;
    const size_t row_bytes = (size_t)3 * C * ES;
    const unsigned char* base = qkv + (size_t)b * T * row_bytes + (size_t)h * 3 * ch * ES;      // q of this head; k at +ch, v at +2ch elements
    const uint32_t q0 = blockIdx.x * (32 * WPB) + wave * 32;
    const bool active = q0 < T;                          // wave-uniform
    const uint32_t nchunk = ch / 8;                      // valid 8-channel chunks per row
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // 8 channels [c8*8, c8*8+8) of row `row` of q (which = 0), k (1) or v (2) as operand terms; zeros outside the tensor
    auto load8 = [&](uint32_t row, uint32_t which, uint32_t c8, bf16x8& hi, bf16x8& lo) {
        hi = zero8; lo = zero8;
        if (row < T && c8 < nchunk) {
            const unsigned char* p = base + (size_t)row * row_bytes + ((size_t)which * ch + c8 * 8) * ES;
            if constexpr (F32) at_split8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 16), hi, lo);
            else hi = *reinterpret_cast<const bf16x8*>(p);
        }
    };

    bf16x8 qf[NT][KS];
    if (active) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 hi, lo;
            load8(q0 + l31, 0, 2 * s + hf, hi, lo);
            qf[0][s] = hi;
            if constexpr (F32) qf[1][s] = lo;
        }
    }

    // V staging (r04: as the backward kernels stage their operands -- the first form wrote eight 2-byte values per 8-channel chunk with every lane of a wave
    // on the same two LDS banks, and THAT, not the products, was most of a key block's time: T = 256 / ch 128 ran 5.7 us per key block).  Item id ->
    // (pair of adjacent keys 2 rp, 2 rp + 1 -- adjacent positions of a V^T row as well --, G channels): every LDS store is one dword {key 2 rp, key 2 rp + 1}
    // of one channel, and the lanes of a wave half cover 16 key pairs x 2 channel groups = 32 different banks (a row is 80 bytes: 20 dwords).
    uint4 va[VCH], vb[VCH];                              // the items' 16 bytes of the two rows, as loaded
    auto v_item = [&](uint32_t id, uint32_t& rp, uint32_t& cg) { rp = (id >> 1) & 15u; cg = ((id >> 5) << 1) | (id & 1u); };
    auto v_load = [&](uint32_t kb) {
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            uint32_t rp, cg;
            v_item(tid + NTH * i, rp, cg);
            const uint32_t ra = kb * 32 + 2 * rp;
            va[i] = make_uint4(0u, 0u, 0u, 0u); vb[i] = va[i];
            if (tid + NTH * i < (uint32_t)VTOT && cg * G < ch) {
                const unsigned char* p = base + (size_t)ra * row_bytes + ((size_t)2 * ch + cg * G) * ES;
                if (ra < T) va[i] = *reinterpret_cast<const uint4*>(p);
                if (ra + 1 < T) vb[i] = *reinterpret_cast<const uint4*>(p + row_bytes);
            }
        }
    };
    auto v_store = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            uint32_t rp, cg;
            v_item(tid + NTH * i, rp, cg);
            if (tid + NTH * i >= (uint32_t)VTOT) continue;
            const uint32_t off = (cg * G) * AT_ROW + at_key_pos(2 * rp) * 2;
            const uint32_t wa[4] = {va[i].x, va[i].y, va[i].z, va[i].w}, wb[4] = {vb[i].x, vb[i].y, vb[i].z, vb[i].w};
            if constexpr (F32) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float fa = __uint_as_float(wa[c]), fb = __uint_as_float(wb[c]);
                    const uint32_t wh = at_pack_bf16(fa, fb);
                    *reinterpret_cast<uint32_t*>(vt(buf, kg, 0) + off + c * AT_ROW) = wh;
                    *reinterpret_cast<uint32_t*>(vt(buf, kg, 1) + off + c * AT_ROW) = at_pack_bf16_rest(fa, fb, wh);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {                    // bf16 pairs as stored: channel c of the two keys -> one dword
                    const uint32_t a = wa[c >> 1], b = wb[c >> 1];
                    *reinterpret_cast<uint32_t*>(vt(buf, kg, 0) + off + c * AT_ROW) = (c & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
                }
            }
        }
    };
    // K staging (r04): the block's waves share a key block, so its K goes through LDS ONCE, already as the operand terms -- the first forms had every wave
    // load the 32 x ch tile into registers and (fp32) split it into the bf16 pair itself: four times the loads and splits, and 64-128 registers for the
    // prefetched and the current raw fragments (the ch = 128 fp32 form moved 416 values per key block between VGPRs and AccVGPRs).  Item id -> (key, 8 channels),
    // consecutive lanes = consecutive chunks of a row (coalesced); rows of KROW bytes, read back as the MFMA A fragments with one ds_read_b128 each.
    uint4 ka[KCH], kb2[F32 ? KCH : 1];
    auto k_load = [&](uint32_t kb) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const uint32_t id = tid + NTH * i, key = id / (CHP / 8), c8 = id % (CHP / 8), row = kb * 32 + key;
            ka[i] = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (F32) kb2[i] = ka[i];
            if (id < (uint32_t)KTOT && row < T && c8 < nchunk) {
                const unsigned char* p = base + (size_t)row * row_bytes + ((size_t)ch + c8 * 8) * ES;
                ka[i] = *reinterpret_cast<const uint4*>(p);
                if constexpr (F32) kb2[i] = *reinterpret_cast<const uint4*>(p + 16);
            }
        }
    };
    auto k_store = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const uint32_t id = tid + NTH * i, key = id / (CHP / 8), c8 = id % (CHP / 8);
            if (id >= (uint32_t)KTOT) continue;
            const uint32_t off = key * KROW + c8 * 16;
            if constexpr (F32) {
                bf16x8 hi, lo;
                at_split8(*reinterpret_cast<const float4*>(&ka[i]), *reinterpret_cast<const float4*>(&kb2[i]), hi, lo);
                *reinterpret_cast<bf16x8*>(kt(buf, kg, 0) + off) = hi;
                *reinterpret_cast<bf16x8*>(kt(buf, kg, 1) + off) = lo;
            } else {
                *reinterpret_cast<uint4*>(kt(buf, kg, 0) + off) = ka[i];
            }
        }
    };

    f32x16 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[c][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const uint32_t nkb = (T + 31) / 32, nsb = (nkb + KG - 1) / KG;          // key blocks; rounds of KG key blocks (group kg takes block sb * KG + kg)
    v_load(kg);
    k_load(kg);
    v_store(0);
    k_store(0);
    __syncthreads();
    for (uint32_t sb = 0; sb < nsb; ++sb) {
        const uint32_t buf = sb & 1, kb = sb * KG + kg;
        if (sb + 1 < nsb) {                                                  // the next round's K and V: global -> registers now, -> LDS after the products
            v_load(kb + KG);                                                 // (a block past the last one reads as zeros and is not multiplied)
            k_load(kb + KG);
        }
        if (active && kb < nkb) {
            f32x16 sacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {                                   // S^T[key][query]
                const uint32_t koff = l31 * KROW + (2 * s + hf) * 16;
                const bf16x8 khi = *reinterpret_cast<const bf16x8*>(kt(buf, kg, 0) + koff);
                if constexpr (F32) {
                    const bf16x8 klo = *reinterpret_cast<const bf16x8*>(kt(buf, kg, 1) + koff);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(klo, qf[0][s], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(khi, qf[1][s], sacc, 0, 0, 0);
                }
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(khi, qf[0][s], sacc, 0, 0, 0);
            }
            if ((kb + 1) * 32 > T) {                                         // last, partial key block: keys past T never win the softmax
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hf;       // C layout: register e of lane half hf
                    if (key >= T) sacc[e] = -1e30f;
                }
            }
            float mx = sacc[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sacc[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            float p[16], psum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], scale_log2e, -m_new)); psum += p[e]; }
            l_run = __builtin_fmaf(l_run, alpha, psum);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[c][e] *= alpha;
            bf16x8 pb[NT][2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint32_t wh[4], wl[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    wh[k] = at_pack_bf16(p[8 * s + 2 * k], p[8 * s + 2 * k + 1]);
                    if constexpr (F32) wl[k] = at_pack_bf16_rest(p[8 * s + 2 * k], p[8 * s + 2 * k + 1], wh[k]);
                }
                const uint4 u = make_uint4(wh[0], wh[1], wh[2], wh[3]);
                pb[0][s] = *reinterpret_cast<const bf16x8*>(&u);
                if constexpr (F32) {
                    const uint4 ul = make_uint4(wl[0], wl[1], wl[2], wl[3]);
                    pb[1][s] = *reinterpret_cast<const bf16x8*>(&ul);
                }
            }
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s) {                                // O^T[channel][query] += V^T[channel][8 keys of this half] P
                    const size_t off = (size_t)(c * 32 + l31) * AT_ROW + s * 32 + hf * 16;
                    const bf16x8 vh = *reinterpret_cast<const bf16x8*>(vt(buf, kg, 0) + off);
                    if constexpr (F32) {
                        const bf16x8 vl = *reinterpret_cast<const bf16x8*>(vt(buf, kg, 1) + off);
                        o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, pb[0][s], o[c], 0, 0, 0);
                        o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pb[1][s], o[c], 0, 0, 0);
                    }
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pb[0][s], o[c], 0, 0, 0);
                }
        }
        if (sb + 1 < nsb) { v_store(buf ^ 1); k_store(buf ^ 1); }
        __syncthreads();
    }
    if constexpr (KG > 1) {                                                  // merge the key groups' partial softmaxes: groups 1.. -> LDS -> group 0 (lane for lane: same query, same half)
        constexpr int MS = CT * 16 + 2;
        float* mg0 = reinterpret_cast<float*>(lds_all) + (wave * 64 + lane) * MS;
        if (kg > 0) {
            float* mg = mg0 + (kg - 1) * (WPB * 64 * MS);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4*>(mg + c * 16 + e) = make_float4(o[c][e], o[c][e + 1], o[c][e + 2], o[c][e + 3]);
            mg[CT * 16] = m_run; mg[CT * 16 + 1] = l_run;
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float* mg = mg0 + (g - 1) * (WPB * 64 * MS);
            const float m1 = mg[CT * 16], l1 = mg[CT * 16 + 1];
            const float m_new = fmaxf(m_run, m1), a0 = __builtin_amdgcn_exp2f(m_run - m_new), a1 = __builtin_amdgcn_exp2f(m1 - m_new);
            m_run = m_new;
            l_run = __builtin_fmaf(l_run, a0, l1 * a1);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(mg + c * 16 + e);
                    o[c][e] = __builtin_fmaf(o[c][e], a0, v.x * a1); o[c][e + 1] = __builtin_fmaf(o[c][e + 1], a0, v.y * a1);
                    o[c][e + 2] = __builtin_fmaf(o[c][e + 2], a0, v.z * a1); o[c][e + 3] = __builtin_fmaf(o[c][e + 3], a0, v.w * a1);
                }
        }
    }
    if (!active || q0 + l31 >= T) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (lse2 && hf == 0) lse2[(size_t)blockIdx.y * T + q0 + l31] = m_run + __builtin_amdgcn_logf(l_tot);      // log2-domain log-sum-exp of the row (backward kernels)
    const float inv_l = 1.0f / l_tot;
    unsigned char* op = out + ((size_t)(b * T + q0 + l31) * C + h * ch) * ES;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                        // registers 4g..4g+3 = channels c*32 + 8g + 4hf + {0..3}
            const uint32_t c0 = c * 32 + 8 * g + 4 * hf;
            if (c0 >= ch) continue;                                          // (ch % 8 == 0: a group of 4 is inside or outside as a whole)
            if constexpr (F32) {
                *reinterpret_cast<float4*>(op + c0 * 4) = make_float4(o[c][4 * g] * inv_l, o[c][4 * g + 1] * inv_l, o[c][4 * g + 2] * inv_l, o[c][4 * g + 3] * inv_l);
            } else {
                const uint32_t w0 = at_pack_bf16(o[c][4 * g] * inv_l, o[c][4 * g + 1] * inv_l);
                const uint32_t w1 = at_pack_bf16(o[c][4 * g + 2] * inv_l, o[c][4 * g + 3] * inv_l);
                *reinterpret_cast<uint2*>(op + c0 * 2) = make_uint2(w0, w1);
            }
        }
}

// ================================================================================================================================
// Backward of the fp32-class attention (r03): the gradient path -- rendering guidance, the fine-tuning prior -- needs d(attention) / d(qkv) with
// frozen weights; r02 ran the library's fp32 kernels there (3.1 ms of a 36 ms guided step).  Same arithmetic class as the forward (bf16 pair
// splits, hi*hi + hi*lo + lo*hi), same layouts, nothing T x T leaves the registers; the forward saves the rows' log-sum-exp (log2 domain).
//   P = exp2(S c - lse),  dP = dO V^T,  D_i = sum_c dO_ic O_ic,  dS = P o (dP - D),  dQ = s dS K,  dK = s dS^T Q,  dV = P^T dO     (c = s log2 e)
//   k_attn_bwd_D    D per (sample, head, row)
//   k_attn_bwd_dq   a wave owns 32 queries and walks the key blocks exactly like the forward: S^T = K Q^T and dP^T = V dO^T (a lane = one query, so lse
//                   and D are per-lane scalars), dS^T is rounded to the bf16 pair in registers and is the B operand of  dQ^T += K^T dS^T  (K^T through
//                   LDS in the C-layout key order, as the forward's V^T)
//   k_attn_bwd_dkv  a wave owns 32 keys and walks the query blocks: S = Q K^T and dP = dO V^T (a lane = one key, 16 queries in its registers; their lse
//                   and D come from LDS), P and dS are the B operands of  dV^T += dO^T P  and  dK^T += Q^T dS  (Q^T and dO^T through LDS).
// r04: the A-operand ROWS of the walked axis (K, V for dQ; Q, dO for dK / dV) also go through LDS, once per block and already split (AtRowStage), and the 64-wide
// heads run two key / query groups per block (template KG): profiles/r04/z_attention_backward_lds_key_groups.txt.
SSD_DEV void at_load_split(const float* p, bool ok, bf16x8& hi, bf16x8& lo) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (ok) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    at_split8(a, b, hi, lo);
}
// 16 fp32 values of an MFMA C tile (this lane's column, its 16 rows) -> the B operands of the two 16-deep k-steps over those rows, as a bf16 pair
SSD_DEV void at_c_to_b(const float (&v)[16], bf16x8 (&hi)[2], bf16x8 (&lo)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint32_t wh[4], wl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wh[k] = at_pack_bf16(v[8 * s + 2 * k], v[8 * s + 2 * k + 1]);
            wl[k] = at_pack_bf16_rest(v[8 * s + 2 * k], v[8 * s + 2 * k + 1], wh[k]);
        }
        const uint4 uh = make_uint4(wh[0], wh[1], wh[2], wh[3]), ul = make_uint4(wl[0], wl[1], wl[2], wl[3]);
        hi[s] = *reinterpret_cast<const bf16x8*>(&uh);
        lo[s] = *reinterpret_cast<const bf16x8*>(&ul);
    }
}
// rows [r0, r0 + 32) x ch channels of an fp32 tensor (row stride `rstride` floats, rows >= T read as zeros) -> LDS, TRANSPOSED, as the bf16 pair:
// dst_hi / dst_lo [channel][AT_ROW bytes], the row index at its C-layout position (at_key_pos).  All 256 threads.  A thread takes TWO adjacent rows
// (2 j, 2 j + 1: adjacent positions as well) and four channels, so that every LDS store is a whole dword {row 2 j, row 2 j + 1} of one channel
// (16 two-byte stores per thread in the first version of this function: the staging, not the products, was most of a key block's time).
template <int CHP> struct AtStage {                                              // one tile's items of this thread, between the global loads and the LDS stores
    static constexpr int TOT = 16 * CHP / 4, N = (TOT + 255) / 256;               // (row pair, 4-channel group) items; per thread
    float4 a[N], b[N];
};
// load half: issued at the head of a block's iteration, so that the round trip runs under the block's products ...
template <int CHP>
SSD_DEV void at_stage_load(AtStage<CHP>& st, const float* __restrict__ src, size_t rstride, uint32_t r0, uint32_t T, uint32_t ch, uint32_t tid) {
#pragma unroll
    for (int i = 0; i < AtStage<CHP>::N; ++i) {
        const uint32_t id = tid + 256 * i, rp = id / (CHP / 4), cg = id % (CHP / 4), ra = r0 + 2 * rp;
        st.a[i] = make_float4(0.f, 0.f, 0.f, 0.f); st.b[i] = st.a[i];
        if (id < (uint32_t)AtStage<CHP>::TOT && cg * 4 < ch) {
            if (ra < T) st.a[i] = *reinterpret_cast<const float4*>(src + (size_t)ra * rstride + cg * 4);
            if (ra + 1 < T) st.b[i] = *reinterpret_cast<const float4*>(src + (size_t)(ra + 1) * rstride + cg * 4);
        }
    }
}
// ... store half: split and write, at the iteration's end
template <int CHP>
SSD_DEV void at_stage_store(const AtStage<CHP>& st, unsigned char* dst_hi, unsigned char* dst_lo, uint32_t tid) {
#pragma unroll
    for (int i = 0; i < AtStage<CHP>::N; ++i) {
        const uint32_t id = tid + 256 * i, rp = id / (CHP / 4), cg = id % (CHP / 4);
        if (id >= (uint32_t)AtStage<CHP>::TOT) break;
        const float va[4] = {st.a[i].x, st.a[i].y, st.a[i].z, st.a[i].w}, vb[4] = {st.b[i].x, st.b[i].y, st.b[i].z, st.b[i].w};
        const uint32_t off = (cg * 4) * AT_ROW + at_key_pos(2 * rp) * 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t wh = at_pack_bf16(va[c], vb[c]);
            *reinterpret_cast<uint32_t*>(dst_hi + off + c * AT_ROW) = wh;
            *reinterpret_cast<uint32_t*>(dst_lo + off + c * AT_ROW) = at_pack_bf16_rest(va[c], vb[c], wh);
        }
    }
}
struct AtRaw8 { float4 a, b; };                                                  // 8 fp32 channels of one row, as loaded (split at their use)
SSD_DEV AtRaw8 at_load_raw(const float* p, bool ok) {
    AtRaw8 r;
    r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a;
    if (ok) { r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); }
    return r;
}
SSD_DEV f32x16 at_mfma3(const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b_hi, const bf16x8& b_lo, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc, 0, 0, 0);
}

__global__ __launch_bounds__(256) void k_attn_bwd_D(const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ Dv, uint32_t B, uint32_t T,
                                                    uint32_t heads, uint32_t ch) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;                         // (b, t, h), h fastest: a wave reads consecutive channels of consecutive heads
    if (i >= B * T * heads) return;
    const uint32_t h = i % heads, bt = i / heads, b = bt / T, t = bt % T;
    const float* o = out + (size_t)bt * heads * ch + h * ch;
    const float* g = dout + (size_t)bt * heads * ch + h * ch;
    float acc = 0.f;
    for (uint32_t c = 0; c < ch; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(o + c), d = *reinterpret_cast<const float4*>(g + c);
        acc = __builtin_fmaf(a.x, d.x, acc); acc = __builtin_fmaf(a.y, d.y, acc); acc = __builtin_fmaf(a.z, d.z, acc); acc = __builtin_fmaf(a.w, d.w, acc);
    }
    Dv[((size_t)b * heads + h) * T + t] = acc;
}

// rows [r0, r0 + 32) x ch channels of an fp32 tensor -> LDS ROW-MAJOR as the bf16 pair (hi | lo buffers, rows of CHP * 2 + 16 bytes: an MFMA A fragment = one
// ds_read_b128, the 32 rows on different banks): the block's waves share the tile, so it is loaded and split ONCE (r04; the backward kernels used to fetch these rows
// into every wave's registers -- raw fp32, one block ahead, 128 registers at 64-wide heads, and at 128-wide heads not ahead at all).  NTH threads; load / store halves
// as at_stage_load / at_stage_store.
template <int CHP, int NTH> struct AtRowStage {
    static constexpr int TOT = 32 * CHP / 8, N = (TOT + NTH - 1) / NTH;            // (row, 8-channel chunk) items; per thread
    float4 a[N], b[N];
};
template <int CHP, int NTH>
SSD_DEV void at_row_load(AtRowStage<CHP, NTH>& st, const float* __restrict__ src, size_t rstride, uint32_t r0, uint32_t T, uint32_t ch, uint32_t tid) {
#pragma unroll
    for (int i = 0; i < AtRowStage<CHP, NTH>::N; ++i) {
        const uint32_t id = tid + NTH * i, row = id / (CHP / 8), c8 = id % (CHP / 8);
        st.a[i] = make_float4(0.f, 0.f, 0.f, 0.f); st.b[i] = st.a[i];
        if (id < (uint32_t)AtRowStage<CHP, NTH>::TOT && r0 + row < T && c8 * 8 < ch) {
            const float* p = src + (size_t)(r0 + row) * rstride + c8 * 8;
            st.a[i] = *reinterpret_cast<const float4*>(p); st.b[i] = *reinterpret_cast<const float4*>(p + 4);
        }
    }
}
template <int CHP, int NTH>
SSD_DEV void at_row_store(const AtRowStage<CHP, NTH>& st, unsigned char* dst_hi, unsigned char* dst_lo, uint32_t tid) {
    constexpr int KROW = CHP * 2 + 16;
#pragma unroll
    for (int i = 0; i < AtRowStage<CHP, NTH>::N; ++i) {
        const uint32_t id = tid + NTH * i, row = id / (CHP / 8), c8 = id % (CHP / 8);
        if (id >= (uint32_t)AtRowStage<CHP, NTH>::TOT) break;
        bf16x8 hi, lo;
        at_split8(st.a[i], st.b[i], hi, lo);
        *reinterpret_cast<bf16x8*>(dst_hi + row * KROW + c8 * 16) = hi;
        *reinterpret_cast<bf16x8*>(dst_lo + row * KROW + c8 * 16) = lo;
    }
}

// KG (r04, as in the forward): with KG = 2 a block has eight waves: wave w and wave w + 4 own the SAME 32 queries (dq) / keys (dkv) and walk the even / the odd blocks
// of the other axis; their partial gradients are plain sums, merged through LDS at the end.  Two waves per SIMD instead of one; 64-wide heads only (the 128-wide
// forms need more than 256 registers).
template <int CHP, int KG>
__global__ __launch_bounds__(256 * KG) void k_attn_bwd_dq(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse2,
                                                          const float* __restrict__ Dv, float* __restrict__ dqkv, uint32_t T, uint32_t heads, uint32_t ch,
                                                          float scale, float scale_log2e) {
    constexpr int KS = CHP / 16, CT = CHP / 32, KROW = CHP * 2 + 16;
    constexpr int TR_BYTES = 2 * KG * 2 * CHP * AT_ROW, ROW_BYTES = 2 * KG * 2 * 32 * KROW;          // K^T; K and V row-major
    constexpr int MERGE_BYTES = (KG - 1) * 4 * 64 * CT * 16 * 4;
    constexpr int LDS_BYTES = TR_BYTES + 2 * ROW_BYTES > MERGE_BYTES ? TR_BYTES + 2 * ROW_BYTES : MERGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[LDS_BYTES];                         // (ONE LDS object)
    auto kt = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + ((buf * KG + g) * 2 + tm) * (CHP * AT_ROW); };            // K^T of a key block, hi | lo
    auto kr = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + TR_BYTES + ((buf * KG + g) * 2 + tm) * (32 * KROW); };    // K rows
    auto vr = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + TR_BYTES + ROW_BYTES + ((buf * KG + g) * 2 + tm) * (32 * KROW); };
    const uint32_t lane = threadIdx.x & 63, wave_all = threadIdx.x >> 6, kg = wave_all / 4, wave = wave_all % 4, tid = threadIdx.x - kg * 256;   // tid: inside the key group
    const uint32_t hf = lane >> 5, l31 = lane & 31;
    const uint32_t b = blockIdx.y / heads, h = blockIdx.y % heads, C = heads * ch, nchunk = ch / 8;
    const size_t rs = (size_t)3 * C;
    const float* base = qkv + (size_t)b * T * rs + (size_t)h * 3 * ch;             // q of this head; k at + ch, v at + 2 ch
    const float* dbase = dout + (size_t)b * T * C + (size_t)h * ch;
    const uint32_t q0 = blockIdx.x * 128 + wave * 32, q = q0 + l31;
    const bool active = q0 < T, q_ok = q < T;
    bf16x8 qh[KS], ql[KS], gh[KS], gl[KS];                                          // Q and dO of this lane's query: B operands (column = query)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const uint32_t c8 = 2 * s + hf;
        at_load_split(base + (size_t)q * rs + c8 * 8, active && q_ok && c8 < nchunk, qh[s], ql[s]);
        at_load_split(dbase + (size_t)q * C + c8 * 8, active && q_ok && c8 < nchunk, gh[s], gl[s]);
    }
    const float lse_q = (active && q_ok) ? lse2[(size_t)blockIdx.y * T + q] : 1e30f;
    const float D_q = (active && q_ok) ? Dv[(size_t)blockIdx.y * T + q] : 0.f;
    f32x16 dq[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[c][e] = 0.f;
    const uint32_t nkb = (T + 31) / 32, nsb = (nkb + KG - 1) / KG;                 // key blocks; rounds of KG key blocks (group kg takes block sb * KG + kg)
    AtStage<CHP> st;
    AtRowStage<CHP, 256> sk, sv;
    auto stage_load = [&](uint32_t kb) {                                           // (a block past the last one reads as zeros and is not multiplied)
        at_stage_load<CHP>(st, base + ch, rs, kb * 32, T, ch, tid);
        at_row_load<CHP, 256>(sk, base + ch, rs, kb * 32, T, ch, tid);
        at_row_load<CHP, 256>(sv, base + 2 * ch, rs, kb * 32, T, ch, tid);
    };
    auto stage_store = [&](uint32_t buf) {
        at_stage_store<CHP>(st, kt(buf, kg, 0), kt(buf, kg, 1), tid);
        at_row_store<CHP, 256>(sk, kr(buf, kg, 0), kr(buf, kg, 1), tid);
        at_row_store<CHP, 256>(sv, vr(buf, kg, 0), vr(buf, kg, 1), tid);
    };
    stage_load(kg);
    stage_store(0);
    __syncthreads();
    for (uint32_t sb = 0; sb < nsb; ++sb) {
        const uint32_t buf = sb & 1, kb = sb * KG + kg;
        if (sb + 1 < nsb) stage_load(kb + KG);
        if (active && kb < nkb) {
            f32x16 sacc, dpacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; dpacc[e] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const uint32_t roff = l31 * KROW + (2 * s + hf) * 16;               // the row this lane supplies to the A operands: key kb * 32 + l31
                const bf16x8 k_h = *reinterpret_cast<const bf16x8*>(kr(buf, kg, 0) + roff), k_l = *reinterpret_cast<const bf16x8*>(kr(buf, kg, 1) + roff);
                sacc = at_mfma3(k_h, k_l, qh[s], ql[s], sacc);                      // K: S^T[key][query]
                const bf16x8 v_h = *reinterpret_cast<const bf16x8*>(vr(buf, kg, 0) + roff), v_l = *reinterpret_cast<const bf16x8*>(vr(buf, kg, 1) + roff);
                dpacc = at_mfma3(v_h, v_l, gh[s], gl[s], dpacc);                    // V: dP^T[key][query]
            }
            float ds[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t k_e = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hf;     // C layout: register e of lane half hf
                const float p = k_e < T ? __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], scale_log2e, -lse_q)) : 0.f;
                ds[e] = p * (dpacc[e] - D_q);
            }
            bf16x8 dsh[2], dsl[2];
            at_c_to_b(ds, dsh, dsl);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s) {                                       // dQ^T[channel][query] += K^T[channel][8 keys of this half] dS^T
                    const size_t off = (size_t)(c * 32 + l31) * AT_ROW + s * 32 + hf * 16;
                    const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kt(buf, kg, 0) + off), kl = *reinterpret_cast<const bf16x8*>(kt(buf, kg, 1) + off);
                    dq[c] = at_mfma3(kh, kl, dsh[s], dsl[s], dq[c]);
                }
        }
        if (sb + 1 < nsb) stage_store(buf ^ 1);
        __syncthreads();
    }
    if constexpr (KG > 1) {                                                         // the key groups' partial dQ: groups 1.. -> LDS -> group 0 (lane for lane)
        float* mg0 = reinterpret_cast<float*>(lds_all) + (wave * 64 + lane) * (CT * 16);
        if (kg > 0) {
            float* mg = mg0 + (kg - 1) * (4 * 64 * CT * 16);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4*>(mg + c * 16 + e) = make_float4(dq[c][e], dq[c][e + 1], dq[c][e + 2], dq[c][e + 3]);
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(mg0 + (g - 1) * (4 * 64 * CT * 16) + c * 16 + e);
                    dq[c][e] += v.x; dq[c][e + 1] += v.y; dq[c][e + 2] += v.z; dq[c][e + 3] += v.w;
                }
    }
    if (!active || !q_ok) return;
    float* op = dqkv + (size_t)(b * T + q) * rs + (size_t)h * 3 * ch;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t c0 = c * 32 + 8 * g + 4 * hf;
            if (c0 >= ch) continue;
            *reinterpret_cast<float4*>(op + c0) = make_float4(dq[c][4 * g] * scale, dq[c][4 * g + 1] * scale, dq[c][4 * g + 2] * scale, dq[c][4 * g + 3] * scale);
        }
}

template <int CHP, int KG>
__global__ __launch_bounds__(256 * KG) void k_attn_bwd_dkv(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse2,
                                                           const float* __restrict__ Dv, float* __restrict__ dqkv, uint32_t T, uint32_t heads, uint32_t ch,
                                                           float scale, float scale_log2e) {
    constexpr int KS = CHP / 16, CT = CHP / 32, KROW = CHP * 2 + 16;
    constexpr int TR_BYTES = 2 * KG * 2 * CHP * AT_ROW, ROW_BYTES = 2 * KG * 2 * 32 * KROW, SC_BYTES = 2 * KG * 32 * 4;      // Q^T, dO^T; Q, dO rows; lse, D
    constexpr int MERGE_BYTES = (KG - 1) * 4 * 64 * 2 * CT * 16 * 4;
    constexpr int STAGE_BYTES = 2 * TR_BYTES + 2 * ROW_BYTES + 2 * SC_BYTES;
    constexpr int LDS_BYTES = STAGE_BYTES > MERGE_BYTES ? STAGE_BYTES : MERGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[LDS_BYTES];                         // (ONE LDS object)
    auto qt = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + ((buf * KG + g) * 2 + tm) * (CHP * AT_ROW); };                          // Q^T of a query block
    auto gt = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + TR_BYTES + ((buf * KG + g) * 2 + tm) * (CHP * AT_ROW); };               // dO^T
    auto qr = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + 2 * TR_BYTES + ((buf * KG + g) * 2 + tm) * (32 * KROW); };              // Q rows
    auto gr = [&](uint32_t buf, uint32_t g, uint32_t tm) { return lds_all + 2 * TR_BYTES + ROW_BYTES + ((buf * KG + g) * 2 + tm) * (32 * KROW); };  // dO rows
    auto lse_s = [&](uint32_t buf, uint32_t g) { return reinterpret_cast<float*>(lds_all + 2 * TR_BYTES + 2 * ROW_BYTES) + (buf * KG + g) * 32; };
    auto d_s = [&](uint32_t buf, uint32_t g) { return reinterpret_cast<float*>(lds_all + 2 * TR_BYTES + 2 * ROW_BYTES + SC_BYTES) + (buf * KG + g) * 32; };
    const uint32_t lane = threadIdx.x & 63, wave_all = threadIdx.x >> 6, kg = wave_all / 4, wave = wave_all % 4, tid = threadIdx.x - kg * 256;   // tid: inside the query group
    const uint32_t hf = lane >> 5, l31 = lane & 31;
    const uint32_t b = blockIdx.y / heads, h = blockIdx.y % heads, C = heads * ch, nchunk = ch / 8;
    const size_t rs = (size_t)3 * C;
    const float* base = qkv + (size_t)b * T * rs + (size_t)h * 3 * ch;
    const float* dbase = dout + (size_t)b * T * C + (size_t)h * ch;
    const uint32_t k0 = blockIdx.x * 128 + wave * 32, key = k0 + l31;
    const bool active = k0 < T, k_ok = key < T;
    bf16x8 kh[KS], kl[KS], vh[KS], vl[KS];                                          // K and V of this lane's key: B operands (column = key)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const uint32_t c8 = 2 * s + hf;
        at_load_split(base + (size_t)key * rs + ch + c8 * 8, active && k_ok && c8 < nchunk, kh[s], kl[s]);
        at_load_split(base + (size_t)key * rs + 2 * ch + c8 * 8, active && k_ok && c8 < nchunk, vh[s], vl[s]);
    }
    f32x16 dk[CT], dv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[c][e] = 0.f; dv[c][e] = 0.f; }
    const uint32_t nqb = (T + 31) / 32, nsb = (nqb + KG - 1) / KG;                 // query blocks; rounds of KG of them (group kg takes block sb * KG + kg)
    AtStage<CHP> sq, sg;
    AtRowStage<CHP, 256> rq, rg;
    float s_lse = 0.f, s_d = 0.f;
    auto stage_load = [&](uint32_t qb) {                                           // (a block past the last one: zeros, lse = +huge -> P = 0; it is not multiplied anyway)
        at_stage_load<CHP>(sq, base, rs, qb * 32, T, ch, tid);
        at_stage_load<CHP>(sg, dbase, (size_t)C, qb * 32, T, ch, tid);
        at_row_load<CHP, 256>(rq, base, rs, qb * 32, T, ch, tid);
        at_row_load<CHP, 256>(rg, dbase, (size_t)C, qb * 32, T, ch, tid);
        if (tid < 32) {
            const uint32_t qi = qb * 32 + tid;
            s_lse = qi < T ? lse2[(size_t)blockIdx.y * T + qi] : 1e30f;             // rows past T: P = exp2(-huge) = 0
            s_d = qi < T ? Dv[(size_t)blockIdx.y * T + qi] : 0.f;
        }
    };
    auto stage_store = [&](uint32_t buf) {
        at_stage_store<CHP>(sq, qt(buf, kg, 0), qt(buf, kg, 1), tid);
        at_stage_store<CHP>(sg, gt(buf, kg, 0), gt(buf, kg, 1), tid);
        at_row_store<CHP, 256>(rq, qr(buf, kg, 0), qr(buf, kg, 1), tid);
        at_row_store<CHP, 256>(rg, gr(buf, kg, 0), gr(buf, kg, 1), tid);
        if (tid < 32) { lse_s(buf, kg)[tid] = s_lse; d_s(buf, kg)[tid] = s_d; }
    };
    stage_load(kg);
    stage_store(0);
    __syncthreads();
    for (uint32_t sb = 0; sb < nsb; ++sb) {
        const uint32_t buf = sb & 1, qb = sb * KG + kg;
        if (sb + 1 < nsb) stage_load(qb + KG);
        if (active && qb < nqb) {
            f32x16 sacc, dpacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; dpacc[e] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const uint32_t roff = l31 * KROW + (2 * s + hf) * 16;               // the row this lane supplies to the A operands: query qb * 32 + l31
                const bf16x8 q_h = *reinterpret_cast<const bf16x8*>(qr(buf, kg, 0) + roff), q_l = *reinterpret_cast<const bf16x8*>(qr(buf, kg, 1) + roff);
                sacc = at_mfma3(q_h, q_l, kh[s], kl[s], sacc);                      // Q: S[query][key]
                const bf16x8 g_h = *reinterpret_cast<const bf16x8*>(gr(buf, kg, 0) + roff), g_l = *reinterpret_cast<const bf16x8*>(gr(buf, kg, 1) + roff);
                dpacc = at_mfma3(g_h, g_l, vh[s], vl[s], dpacc);                    // dO: dP[query][key]
            }
            float p[16], ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {                                           // registers 4 g .. 4 g + 3 = queries 8 g + 4 hf + {0 .. 3} of the block
                const float4 lq = *reinterpret_cast<const float4*>(lse_s(buf, kg) + 8 * g + 4 * hf), dq4 = *reinterpret_cast<const float4*>(d_s(buf, kg) + 8 * g + 4 * hf);
                const float lv[4] = {lq.x, lq.y, lq.z, lq.w}, dvv[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 4 * g + r;
                    p[e] = k_ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], scale_log2e, -lv[r])) : 0.f;
                    ds[e] = p[e] * (dpacc[e] - dvv[r]);
                }
            }
            bf16x8 ph[2], pl[2], dsh[2], dsl[2];
            at_c_to_b(p, ph, pl);
            at_c_to_b(ds, dsh, dsl);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const size_t off = (size_t)(c * 32 + l31) * AT_ROW + s * 32 + hf * 16;
                    const bf16x8 g_h = *reinterpret_cast<const bf16x8*>(gt(buf, kg, 0) + off), g_l = *reinterpret_cast<const bf16x8*>(gt(buf, kg, 1) + off);
                    dv[c] = at_mfma3(g_h, g_l, ph[s], pl[s], dv[c]);                 // dV^T[channel][key] += dO^T[channel][8 queries of this half] P
                    const bf16x8 q_h = *reinterpret_cast<const bf16x8*>(qt(buf, kg, 0) + off), q_l = *reinterpret_cast<const bf16x8*>(qt(buf, kg, 1) + off);
                    dk[c] = at_mfma3(q_h, q_l, dsh[s], dsl[s], dk[c]);               // dK^T[channel][key] += Q^T[channel][8 queries of this half] dS
                }
        }
        if (sb + 1 < nsb) stage_store(buf ^ 1);
        __syncthreads();
    }
    if constexpr (KG > 1) {                                                         // the query groups' partial dK, dV: groups 1.. -> LDS -> group 0 (lane for lane)
        constexpr int MS = 2 * CT * 16;
        float* mg0 = reinterpret_cast<float*>(lds_all) + (wave * 64 + lane) * MS;
        if (kg > 0) {
            float* mg = mg0 + (kg - 1) * (4 * 64 * MS);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    *reinterpret_cast<float4*>(mg + c * 16 + e) = make_float4(dk[c][e], dk[c][e + 1], dk[c][e + 2], dk[c][e + 3]);
                    *reinterpret_cast<float4*>(mg + CT * 16 + c * 16 + e) = make_float4(dv[c][e], dv[c][e + 1], dv[c][e + 2], dv[c][e + 3]);
                }
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float* mg = mg0 + (g - 1) * (4 * 64 * MS);
                    const float4 a = *reinterpret_cast<const float4*>(mg + c * 16 + e), v = *reinterpret_cast<const float4*>(mg + CT * 16 + c * 16 + e);
                    dk[c][e] += a.x; dk[c][e + 1] += a.y; dk[c][e + 2] += a.z; dk[c][e + 3] += a.w;
                    dv[c][e] += v.x; dv[c][e + 1] += v.y; dv[c][e + 2] += v.z; dv[c][e + 3] += v.w;
                }
    }
    if (!active || !k_ok) return;
    float* op = dqkv + (size_t)(b * T + key) * rs + (size_t)h * 3 * ch;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t c0 = c * 32 + 8 * g + 4 * hf;
            if (c0 >= ch) continue;
            *reinterpret_cast<float4*>(op + ch + c0) = make_float4(dk[c][4 * g] * scale, dk[c][4 * g + 1] * scale, dk[c][4 * g + 2] * scale, dk[c][4 * g + 3] * scale);
            *reinterpret_cast<float4*>(op + 2 * ch + c0) = make_float4(dv[c][4 * g], dv[c][4 * g + 1], dv[c][4 * g + 2], dv[c][4 * g + 3]);
        }
}

template <bool F32>
int at_launch(const char* who, const void* qkv, void* out, uint32_t B, uint32_t T, uint32_t heads, uint32_t ch, void* stream, float* lse2 = nullptr) {
    if (B == 0 || T == 0) return SSDNERF_OK;
    SSD_REQUIRE(qkv && out, "%s: null pointer", who);
    SSD_REQUIRE(ch >= 8 && ch <= 128 && ch % 8 == 0, "%s: head width must be a multiple of 8 in [8, 128]", who);
    SSD_REQUIRE(heads > 0 && (uint64_t)B * heads <= 65535, "%s: B*heads <= 65535", who);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)ch);       // softmax(q.k / sqrt(ch)) == softmax((q s)(k s)), s = ch^-1/4
    // waves per block: four.  Fewer (so that the short sequences' 32-64 blocks become 128-256) measured SLOWER at every shape -- the block's K / V staging is
    // shared work, and halving the waves that share it costs more than the idle CUs (profiles/r04/n_attention_*.txt); SSDNERF_ATTN_WPB=2|1 keeps the A/B.
    static const int forced = getenv("SSDNERF_ATTN_WPB") ? atoi(getenv("SSDNERF_ATTN_WPB")) : 0;
    const int wpb = (forced == 1 || forced == 2) ? forced : 4;
    const dim3 grid((T + 32 * wpb - 1) / (32 * wpb), B * heads), block(64 * wpb);
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* in = (const unsigned char*)qkv;
    unsigned char* o = (unsigned char*)out;
    // key groups: two (8 waves per block, two per SIMD) when the grid alone leaves the chip at one wave per SIMD or less; SSDNERF_ATTN_KG=1|2 forces.
    // r04, 8 scenes x 4 heads (profiles/r04/q_attention_key_groups.txt): T = 1024 fp32 73.5 -> 49.4 us, bf16 37.7 -> 25.0; T = 256 32.3 -> 25.5 / 15.5 -> 14.9;
    // T = 64 unchanged (two key blocks: nothing to split); FOUR groups (16 waves, 128 registers each) 53.7 / 29.2 us at T = 1024: slower than two, removed.
    static const int forced_kg = getenv("SSDNERF_ATTN_KG") ? atoi(getenv("SSDNERF_ATTN_KG")) : 0;
    const bool kg2 = wpb == 4 && (forced_kg == 2 || (forced_kg != 1 && (uint64_t)grid.x * grid.y <= 512 && T >= 128));
#define AT_FWD(CHP)                                                                                                                       \
    if (kg2) hipLaunchKernelGGL((k_attn_fwd<CHP, F32, 4, 2>), grid, dim3(512), 0, st, in, o, T, heads, ch, scale_log2e, lse2);            \
    else if (wpb == 4) hipLaunchKernelGGL((k_attn_fwd<CHP, F32, 4>), grid, block, 0, st, in, o, T, heads, ch, scale_log2e, lse2);         \
    else if (wpb == 2) hipLaunchKernelGGL((k_attn_fwd<CHP, F32, 2>), grid, block, 0, st, in, o, T, heads, ch, scale_log2e, lse2);         \
    else hipLaunchKernelGGL((k_attn_fwd<CHP, F32, 1>), grid, block, 0, st, in, o, T, heads, ch, scale_log2e, lse2);
    if (ch <= 32) { AT_FWD(32) } else if (ch <= 64) { AT_FWD(64) } else if (ch <= 96) { AT_FWD(96) } else { AT_FWD(128) }
#undef AT_FWD
    SSD_CHECK_LAUNCH(who);
    return SSDNERF_OK;
}

}  // namespace

extern "C" int ssdnerf_attention_qkv_bf16(const void* qkv, void* out, uint32_t B, uint32_t T, uint32_t heads, uint32_t ch, void* stream) {
    return at_launch<false>("attention_qkv_bf16", qkv, out, B, T, heads, ch, stream);
}

extern "C" int ssdnerf_attention_qkv_f32(const void* qkv, void* out, uint32_t B, uint32_t T, uint32_t heads, uint32_t ch, void* stream) {
    return at_launch<true>("attention_qkv_f32", qkv, out, B, T, heads, ch, stream);
}

// Forward that also saves the rows' log-sum-exp (log2 domain) for ssdnerf_attention_qkv_f32_backward: lse2 fp32 [B][heads][T].
extern "C" int ssdnerf_attention_qkv_f32_lse(const void* qkv, void* out, void* lse2, uint32_t B, uint32_t T, uint32_t heads, uint32_t ch, void* stream) {
    SSD_REQUIRE(lse2, "attention_qkv_f32_lse: null pointer");
    return at_launch<true>("attention_qkv_f32_lse", qkv, out, B, T, heads, ch, stream, (float*)lse2);
}

extern "C" int ssdnerf_attention_qkv_f32_backward(const void* qkv, const void* out, const void* dout, const void* lse2, void* dqkv, void* workspace, uint32_t B, uint32_t T,
                                                  uint32_t heads, uint32_t ch, void* stream) {
    if (B == 0 || T == 0) return SSDNERF_OK;
    SSD_REQUIRE(qkv && out && dout && lse2 && dqkv && workspace, "attention_qkv_f32_backward: null pointer");
    SSD_REQUIRE(ch >= 8 && ch <= 128 && ch % 8 == 0, "attention_qkv_f32_backward: head width must be a multiple of 8 in [8, 128]");
    SSD_REQUIRE(heads > 0 && (uint64_t)B * heads <= 65535, "attention_qkv_f32_backward: B*heads <= 65535");
    const float scale = 1.0f / sqrtf((float)ch), scale_log2e = 1.4426950408889634f * scale;
    hipStream_t st = (hipStream_t)stream;
    const float* q = (const float*)qkv;
    const float* g = (const float*)dout;
    const float* l = (const float*)lse2;
    float* Dv = (float*)workspace;                                            // B * heads * T floats
    float* d = (float*)dqkv;
    hipLaunchKernelGGL(k_attn_bwd_D, dim3((B * T * heads + 255) / 256), dim3(256), 0, st, (const float*)out, g, Dv, B, T, heads, ch);
    const dim3 grid((T + 127) / 128, B * heads);
    // two key / query groups (eight waves per block, two per SIMD) for the 64-wide-or-narrower heads when the grid leaves the chip at <= one wave per SIMD;
    // SSDNERF_ATTN_BWD_KG=1|2 forces (A/B runs)
    static const int forced_kg = getenv("SSDNERF_ATTN_BWD_KG") ? atoi(getenv("SSDNERF_ATTN_BWD_KG")) : 0;
    const bool kg2 = ch <= 64 && (forced_kg == 2 || (forced_kg != 1 && (uint64_t)grid.x * grid.y <= 512 && T >= 128));
#define AT_BWD_KG(CHP, KG)                                                                                                                    \
    hipLaunchKernelGGL((k_attn_bwd_dq<CHP, KG>), grid, dim3(256 * KG), 0, st, q, g, l, (const float*)Dv, d, T, heads, ch, scale, scale_log2e);  \
    hipLaunchKernelGGL((k_attn_bwd_dkv<CHP, KG>), grid, dim3(256 * KG), 0, st, q, g, l, (const float*)Dv, d, T, heads, ch, scale, scale_log2e);
    if (ch <= 32) { if (kg2) { AT_BWD_KG(32, 2) } else { AT_BWD_KG(32, 1) } }
    else if (ch <= 64) { if (kg2) { AT_BWD_KG(64, 2) } else { AT_BWD_KG(64, 1) } }
    else if (ch <= 96) { AT_BWD_KG(96, 1) } else { AT_BWD_KG(128, 1) }
#undef AT_BWD_KG
    SSD_CHECK_LAUNCH("attention_qkv_f32_backward");
    return SSDNERF_OK;
}
