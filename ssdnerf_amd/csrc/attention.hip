// ssdnerf_amd/csrc/attention.hip -- the denoising UNet's self-attention (MultiHeadAttentionMod.QKVAttention,
// lib/models/architecture/ddpm/modules.py:12-48 + mmgen's QKVAttention) on the bf16 matrix cores.
//
// Input is the qkv projection of the channel-last activation, [B][T][3C] bf16 with the reference's per-head channel order
// [head][q | k | v][ch]; output is [B][T][C] bf16, channel = head*ch + i (what the proj GEMM consumes).  16 sites per forward:
// T = 1024 (ch 64), 256 and 64 (ch 128), B*heads = 32 independent problems each.
//
// Flash-style, one pass over the keys, nothing T x T ever leaves the registers:
//   * a wave owns 32 queries; per block of 32 keys it computes  S^T = K Q^T  with v_mfma_f32_32x32x16_bf16 -- transposed on
//     purpose: in the MFMA's C layout a lane then holds ONE query (column) and 16 keys (registers), so the running max / sum of
//     the online softmax are per-lane scalars (one cross-half shuffle per block) and rescaling O is a per-lane multiply;
//   * P = exp2((S - m) * scale*log2e) is rounded to bf16 in registers and is *already* the B operand of  O^T += V^T P : the
//     C-layout's key order per lane half ({0-3, 8-11} / {4-7, 12-15} of every 16) is simply adopted as the k-slot order, and the
//     V tile is written to LDS transposed in that same order, so no permute or LDS round trip of P is needed;
//   * K fragments come straight from L2 into registers (each lane's 16 bytes are contiguous in the qkv row), V goes through LDS
//     because it needs the transpose; both are prefetched one key block ahead; one barrier per key block;
//   * softmax statistics in fp32, O accumulated in fp32, divided by the row sum at the end.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int AT_ROW = 80;                 // bytes per LDS row of V^T (32 keys * 2 B + 16 B pad: conflict-free ds_read_b128)

SSD_DEV uint32_t at_bf16_rne(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// position of key kk (0..31) inside a V^T row: per 16 keys, lane half 0 owns {0-3, 8-11}, half 1 owns {4-7, 12-15} (MFMA C layout)
SSD_DEV uint32_t at_key_pos(uint32_t kk) {
    const uint32_t w = kk & 15u;
    return (kk & 16u) + ((w >> 2) & 1u) * 8u + (w & 3u) + 4u * (w >> 3);
}

template <int CH>
__global__ __launch_bounds__(256) void k_attn_fwd_bf16(const unsigned char* __restrict__ qkv, unsigned char* __restrict__ out, uint32_t T, uint32_t heads,
                                                       float scale_log2e) {
    constexpr int KS = CH / 16;            // k-steps of the QK^T product
    constexpr int CT = CH / 32;            // 32-channel tiles of O
    constexpr int VCH = (32 * CH / 8) / 256;   // 16-byte V chunks per thread per key block
    __shared__ __attribute__((aligned(16))) unsigned char vt[2][CH * AT_ROW];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const uint32_t b = blockIdx.y / heads, h = blockIdx.y % heads, C = heads * CH;
    const size_t row_bytes = (size_t)3 * C * 2;
    const unsigned char* base = qkv + (size_t)b * T * row_bytes + (size_t)h * 3 * CH * 2;       // q of this head; k at +CH, v at +2CH elements
    const uint32_t q0 = blockIdx.x * 128 + wave * 32;
    const bool active = q0 < T;

    bf16x8 qf[KS];
    if (active) {
        const unsigned char* qp = base + (size_t)(q0 + l31) * row_bytes + hf * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + s * 32);
    }

    // V staging: chunk id = tid + 256*i -> (key, 8-channel chunk)
    uint4 vreg[VCH];
    auto v_load = [&](uint32_t kb) {
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const uint32_t id = tid + 256 * i, key = id / (CH / 8), cc = id % (CH / 8);
            vreg[i] = *reinterpret_cast<const uint4*>(base + (size_t)(kb * 32 + key) * row_bytes + (2 * CH + cc * 8) * 2);
        }
    };
    auto v_store = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const uint32_t id = tid + 256 * i, key = id / (CH / 8), cc = id % (CH / 8);
            const uint32_t w[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
            unsigned char* dst = vt[buf] + (cc * 8) * AT_ROW + at_key_pos(key) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<uint16_t*>(dst + e * AT_ROW) = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
        }
    };
    bf16x8 kf[KS];
    auto k_load = [&](uint32_t kb) {
        const unsigned char* kp = base + (size_t)(kb * 32 + l31) * row_bytes + (CH + hf * 8) * 2;
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(kp + s * 32);
    };

    f32x16 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[c][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const uint32_t nkb = T / 32;
    v_load(0);
    v_store(0);
    if (active) k_load(0);
    __syncthreads();
    for (uint32_t kb = 0; kb < nkb; ++kb) {
        const uint32_t buf = kb & 1;
        bf16x8 kcur[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) kcur[s] = kf[s];
        if (kb + 1 < nkb) {                                                  // prefetch the next key block (K -> registers, V -> registers)
            v_load(kb + 1);
            if (active) k_load(kb + 1);
        }
        if (active) {
            f32x16 sacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcur[s], qf[s], sacc, 0, 0, 0);   // S^T[key][query]
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
            bf16x8 pb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = at_bf16_rne(p[8 * s + 2 * k]) | (at_bf16_rne(p[8 * s + 2 * k + 1]) << 16);
                const uint4 u = make_uint4(w[0], w[1], w[2], w[3]);
                pb[s] = *reinterpret_cast<const bf16x8*>(&u);
            }
            const unsigned char* vrow = vt[buf] + l31 * AT_ROW + hf * 16;
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vrow + c * 32 * AT_ROW + s * 32);   // V^T[channel][8 keys of this half]
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[s], o[c], 0, 0, 0);              // O^T[channel][query]
                }
        }
        if (kb + 1 < nkb) v_store(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
    const float inv_l = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    unsigned char* op = out + ((size_t)(b * T + q0 + l31) * C + h * CH) * 2;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                        // registers 4g..4g+3 = channels c*32 + 8g + 4hf + {0..3}
            const uint32_t w0 = at_bf16_rne(o[c][4 * g] * inv_l) | (at_bf16_rne(o[c][4 * g + 1] * inv_l) << 16);
            const uint32_t w1 = at_bf16_rne(o[c][4 * g + 2] * inv_l) | (at_bf16_rne(o[c][4 * g + 3] * inv_l) << 16);
            *reinterpret_cast<uint2*>(op + (c * 32 + 8 * g + 4 * hf) * 2) = make_uint2(w0, w1);
        }
}

}  // namespace

extern "C" int ssdnerf_attention_qkv_bf16(const void* qkv, void* out, uint32_t B, uint32_t T, uint32_t heads, uint32_t ch, void* stream) {
    if (B == 0 || T == 0) return SSDNERF_OK;
    SSD_REQUIRE(qkv && out, "attention_qkv_bf16: null pointer");
    SSD_REQUIRE(ch == 64 || ch == 128, "attention_qkv_bf16: head width must be 64 or 128");
    SSD_REQUIRE(T % 32 == 0 && heads > 0 && B * heads <= 65535, "attention_qkv_bf16: T must be a multiple of 32, B*heads <= 65535");
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)ch);       // softmax(q.k / sqrt(ch)) == softmax((q s)(k s)), s = ch^-1/4
    const dim3 grid((T + 127) / 128, B * heads), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (ch == 64) hipLaunchKernelGGL(k_attn_fwd_bf16<64>, grid, block, 0, st, (const unsigned char*)qkv, (unsigned char*)out, T, heads, scale_log2e);
    else hipLaunchKernelGGL(k_attn_fwd_bf16<128>, grid, block, 0, st, (const unsigned char*)qkv, (unsigned char*)out, T, heads, scale_log2e);
    SSD_CHECK_LAUNCH("attention_qkv_bf16");
    return SSDNERF_OK;
}
