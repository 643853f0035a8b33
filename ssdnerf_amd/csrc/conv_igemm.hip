// ssdnerf_amd/csrc/conv_igemm.hip -- the denoising UNet's convolutions as an implicit GEMM on the bf16 matrix cores.
//
// Reference: every 3x3 / 1x1 / strided / post-upsample convolution of DenoisingUnetMod's residual, down- and up-sampling
// blocks (lib/models/architecture/ddpm/modules.py:51-129; denoising.py:106-187 lays them out), 211.8 of the UNet's
// 218 GFLOP per scene per DDIM step.  The reference runs them through cuDNN on NCHW tensors.
//
// Here, for channel-last bf16 activations x[B][H][W][Cin] and weights w[Cout][kh][kw][Cin] (torch's channels_last weight
// memory, consumed in place):
//     y[m][co] = sum_{tap, ci} x[pixel(m, tap)][ci] * w[co][tap][ci]   (+ bias[co]) (+ residual[m][co]),   m = (b, yo, xo)
// is a GEMM with M = B*Ho*Wo, N = Cout, K = taps*Cin whose A rows are *gathered*: K-tile (tap, ci0) of output pixel m is
// the 128 contiguous bytes x[pixel(m, tap)][ci0 .. ci0+64), or zeros where the tap falls into the padding.
//
//   * block = 4 waves (2 x 2) with a 64x64 / 64x128 / 128x128 tile, or 8 waves (4 x 2) with a 256x128 tile (large layers: the
//     bigger the tile, the fewer L2->LDS bytes per FLOP); K-tile 64, `v_mfma_f32_32x32x16_bf16`, fp32 accumulators;
//   * both operands go HBM/L2 -> LDS with `global_load_lds_dwordx4` (no VGPR staging, no ds_write pass) through a 2-stage
//     (128x128 tile, two blocks per CU) to 4-stage (64x64 tile) ring with counted `s_waitcnt vmcnt(N)` and one raw barrier
//     per K-tile; padding taps read a 16-byte zero page instead of being predicated;
//   * the LDS image of a tile row is its eight 16-byte chunks XOR-permuted by ((row >> 1) & 7) -- applied on the SOURCE
//     address of the DMA (its LDS side is lane-linear) and again on the ds_read_b128 address -- which makes every one of
//     ds_read_b128's 16-lane groups hit 16 distinct 16-byte bank slots (MI355X_MICROARCH.md, LDS table);
//   * epilogue through LDS: accumulators -> fp32 tile -> rows of 8 channels per lane, + bias + residual in fp32, one
//     rounding to bf16, 16-byte stores (a residual block's `conv_2 + bias + skip` costs no extra pass); optionally the
//     per-(sample, group) sum / sum-of-squares of the *output* are accumulated for the GroupNorm that follows
//     (fp64 atomics into the arena `ssdnerf_group_norm_nhwc` reads), which removes that norm's statistics pass;
//   * stride 2 (DenoisingDownsampleMod) and "nearest-2x-upsample then 3x3" (DenoisingUpsampleMod, the upsampled tensor is
//     never materialised) are index maps of the same gather;
//   * consecutive M-tiles go to the same XCD so that the halo rows neighbouring tiles share are L2 hits.
//
// Bound: MFMA (dense bf16 peak ~2.5 PFLOP/s).  Arithmetic: bf16 products, fp32 accumulation in MFMA order.
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __attribute__((aligned(16))) const uint32_t g_conv_zero_page[4] = {0, 0, 0, 0};

struct ConvArgs {
    const unsigned char* x;      // bf16 [B][H][W][Cin1]   (Cin1 = Cin unless x2 is given)
    const unsigned char* x2;     // bf16 [B][H][W][Cin - Cin1] or null: the input is the never-materialised concatenation [x | x2]
    uint32_t Cin1;
    const unsigned char* w;      // bf16 [Cout][taps][Cin]
    const float* bias;           // fp32 [Cout] or null
    const unsigned char* res;    // bf16 [M][Cout] or null
    unsigned char* y;            // bf16 [M][Cout]
    double* gn_sums;             // fp64 [B][G][2] or null
    float* splitk_ws;            // fp32 [M][Cout], zero on entry (split-K partial sums), or null
    uint32_t splits;             // K is cut into `splits` ranges of K-tiles, one block each
    uint32_t B, H, W, Cin, Cout; // input geometry
    uint32_t Ho, Wo, M;          // output geometry, M = B*Ho*Wo
    uint32_t ksize, stride, pad, upsample;
    uint32_t G;                  // GroupNorm groups of the output (for gn_sums)
    uint32_t m_tiles, n_tiles;
    uint32_t* tickets;           // r05, split-K fold: one arrival counter per output tile (all zero on entry, left all zero), or null: the finishing pass is a kernel of its own
};

// Split-K fold (r05).  A split layer's blocks add their partial sums to the fp32 scratch; until r04 a second kernel (k_conv_splitk_finish / k_conv_f32_finish: 45 launches
// per UNet forward, 0.2 - 0.3 ms of a 4.3 / 8.3 ms step) read the sums back, added bias / residual, took the GroupNorm statistics and wrote the output.  With
// `tickets`, the block that ARRIVES LAST at a tile does that itself: every block fences its atomics and takes a ticket; the holder of the last one reads the tile's sums
// back into its accumulators with agent-scope loads (the other blocks' atomics were performed at the memory side; a plain load could hit a stale line of this XCD's L2),
// zeroes scratch and ticket, and runs the kernel's ordinary epilogue.  Same sums (the order of the fp32 atomics was already arbitrary), one launch less per layer.
// Host side: cv_fold_tickets() -- needs the scratch (not the output) as accumulator, room for the tickets behind it, and, for the epilogue's statistics, tiles inside one sample.
constexpr size_t CV_TICKET_BYTES = 16384;
static uint32_t* cv_fold_tickets(const ConvArgs& a, void* splitk_ws, size_t splitk_ws_bytes, uint32_t bm, bool want_stats) {
    // OFF unless SSDNERF_CONV_FOLD=1 (r05, measured on the cars UNet at 8 scenes: fp32 step 8.13 -> 8.87 ms, bf16 4.24 -> 4.50 ms WITH the fold): the finishing
    // kernel spreads a layer's epilogue over the whole chip, the fold leaves it to the one block per tile that arrives last -- 64 uncached loads per lane in series
    // with that tile's tail -- and these layers are latency-bound already.  Kept as an opt-in; results equal the finishing pass's (tests/test_unet_fast_gpu.py run with it).
    static const bool on = [] { const char* e = getenv("SSDNERF_CONV_FOLD"); return e != nullptr && e[0] != '\0' && e[0] != '0'; }();      // (=0 and empty mean off)
    if (!on || a.splits <= 1 || splitk_ws == nullptr || (void*)a.splitk_ws != splitk_ws) return nullptr;
    if (splitk_ws_bytes < (size_t)a.M * a.Cout * 4 + CV_TICKET_BYTES || (size_t)a.m_tiles * a.n_tiles * 4 > CV_TICKET_BYTES) return nullptr;
    if (want_stats && (a.Ho * a.Wo) % bm != 0) return nullptr;
    return reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(splitk_ws) + splitk_ws_bytes - CV_TICKET_BYTES);
}
// device side, behind a block's atomics: true for the block that arrived last -- its accumulators then hold the tile's complete sums and scratch / ticket are zero again
template <int TM, int TN>
SSD_DEV bool cv_fold_last(const ConvArgs& a, f32x16 (&acc)[TM][TN], float* ws, uint32_t* flag_lds, uint32_t tile, uint32_t m0, uint32_t n0, uint32_t row_off, uint32_t col_off, bool PT) {
    // this block's atomics are acknowledged before its ticket is taken: `s_waitcnt vmcnt(0)` per wave (a workgroup-scope release), then the barrier.  NOT
    // __threadfence(): an agent-scope release writes the L2's dirty lines back (buffer_wbl2) -- measured: the UNet step 8.1 -> 11.5 ms with it.  The float atomics
    // and the ticket are device-scope read-modify-writes resolved at one coherence point; the sums are read back with agent-scope loads that bypass this XCD's L2.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) *flag_lds = atomicAdd(a.tickets + tile, 1u);
    __syncthreads();
    const bool last = *flag_lds == a.splits - 1;
    __syncthreads();                                                         // (the flag's word belongs to the epilogue's scratch)
    if (!last) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t m = m0 + row_off + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const uint32_t col = n0 + col_off + j * 32 + (lane & 31);
                float v = 0.f;
                if (m < a.M && (!PT || col < a.Cout)) {
                    float* p = ws + (size_t)m * a.Cout + col;
                    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *p = 0.f;
                }
                acc[i][j][e] = v;
            }
    if (threadIdx.x == 0) a.tickets[tile] = 0u;
    return true;
}

constexpr int CV_BK = 64;                  // bf16 elements per K-tile = 128 bytes per tile row
constexpr int CV_ROWB = CV_BK * 2;

SSD_DEV void cv_glds16(const void* gptr, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// {bf16(lo), bf16(hi)}, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (r03; the integer form cost five instructions per element)
SSD_DEV uint32_t cv_pack_bf16(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 r = __builtin_convertvector(f2{lo, hi}, b2);
    return *reinterpret_cast<const uint32_t*>(&r);
}

// LDS-DMA through a BUFFER descriptor (r03): `buffer_load_dwordx4 voffset, rsrc, soffset offen lds`.  Against the flat form used above
// (global_load_lds with a 64-bit address per lane) it takes ONE 32-bit offset register per lane and a scalar offset -- the channel tile / tap of a
// K-tile is a scalar add -- and a lane whose offset lies outside the buffer (>= num_records) reads zeros: padding taps are an offset of 2^31, no zero
// page, no 64-bit selects.  The r02 loop spent ~190 VALU + SALU instructions per K-tile and wave on addresses beside 16 MFMAs (PMC: active-issue
// 1000 cycles per K-tile against 512 MFMA cycles); this form needs ~40.
// Epilogue of the two-group kernel, straight from the accumulators (r03).  The K loop multiplies TRANSPOSED -- weights as the MFMA's A operand, pixels
// as B -- so that in the 32 x 32 C layout a lane holds ONE output pixel (lane & 31) and four runs of four consecutive output channels
// (e = 4 q + r  ->  channel 8 q + 4 (lane >> 5) + r): bias (+ residual) are added in fp32, the 16 values are rounded once to bf16 and packed, one
// v_permlane32_swap per packed dword pairs the two lane halves so that every lane holds 8 consecutive channels, and the tile leaves as 16-byte stores
// (programming guide T21).  No LDS round trip, no block barrier: the LDS-staged epilogue above cost 17 - 25 us of a 56 us layer once the K loop no
// longer hid it (profiles/r03/d_pp_epilogue_split.txt).  GroupNorm sums of the rounded values: per lane over its pixels, DPP-reduced over the 32
// lanes of each half, one LDS atomic per 4-channel run and wave, then the same per-group fp64 atomics as cv_epilogue_bf16.
SSD_DEV float cv_half_wave_sum(float v) {                                       // sum over the 32 lanes of each wave half, valid in lanes 16-31 / 48-63
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
    return v;
}

template <int TM, int TN, bool F32 = false>
SSD_DEV void cv_epilogue_direct(const ConvArgs& a, f32x16 (&acc)[TN][TM], unsigned char* lds, uint32_t m0, uint32_t n0, uint32_t wm, uint32_t wn) {
    constexpr int BN = 128;
    const uint32_t tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    float* red = reinterpret_cast<float*>(lds);                              // [BN / 4 runs][sum, sumsq]: the caller hands in 512 bytes BEHIND the stage ring (the
                                                                             // persistent kernel's zero rows at the head of A stage 0 must survive this epilogue)
    if (a.gn_sums) {
        if (tid < BN / 4 * 2) red[tid] = 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const uint32_t cb = n0 + wn * 32 * TN + j * 32;                      // first channel of this 32-channel tile
        float4 bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = a.bias ? *reinterpret_cast<const float4*>(a.bias + cb + 8 * q + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint32_t m = m0 + wm * 32 * TM + i * 32 + (lane & 31);
            const bool ok = m < a.M;
            const size_t row = (size_t)(ok ? m : 0) * a.Cout;
            uint2 pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float f[4] = {acc[j][i][4 * q] + bv[q].x, acc[j][i][4 * q + 1] + bv[q].y, acc[j][i][4 * q + 2] + bv[q].z, acc[j][i][4 * q + 3] + bv[q].w};
                if (F32) {                                                   // fp32 activations (the f32x2 form): a lane's run of four channels is one 16-byte store
                    const size_t o = row + cb + 8 * q + 4 * half;
                    if (a.res) {
                        const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.res) + o);
                        f[0] += rv.x; f[1] += rv.y; f[2] += rv.z; f[3] += rv.w;
                    }
                    if (ok) *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + o) = make_float4(f[0], f[1], f[2], f[3]);
                    if (a.gn_sums && ok) {
                        gs[q] += (f[0] + f[1]) + (f[2] + f[3]);
                        gq[q] = __builtin_fmaf(f[0], f[0], gq[q]); gq[q] = __builtin_fmaf(f[1], f[1], gq[q]);
                        gq[q] = __builtin_fmaf(f[2], f[2], gq[q]); gq[q] = __builtin_fmaf(f[3], f[3], gq[q]);
                    }
                    continue;
                }
                if (a.res) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(a.res + (row + cb + 8 * q + 4 * half) * 2);
                    f[0] += __uint_as_float(rv.x << 16); f[1] += __uint_as_float(rv.x & 0xffff0000u);
                    f[2] += __uint_as_float(rv.y << 16); f[3] += __uint_as_float(rv.y & 0xffff0000u);
                }
                pk[q] = make_uint2(cv_pack_bf16(f[0], f[1]), cv_pack_bf16(f[2], f[3]));
                if (a.gn_sums && ok) {
                    const float r0 = __uint_as_float(pk[q].x << 16), r1 = __uint_as_float(pk[q].x & 0xffff0000u);
                    const float r2 = __uint_as_float(pk[q].y << 16), r3 = __uint_as_float(pk[q].y & 0xffff0000u);
                    gs[q] += (r0 + r1) + (r2 + r3);
                    gq[q] = __builtin_fmaf(r0, r0, gq[q]); gq[q] = __builtin_fmaf(r1, r1, gq[q]);
                    gq[q] = __builtin_fmaf(r2, r2, gq[q]); gq[q] = __builtin_fmaf(r3, r3, gq[q]);
                }
            }
#pragma unroll
            for (int qp = 0; qp < (F32 ? 0 : 2); ++qp) {                     // lower half ends up with channels 16 qp .. + 7, upper half with 16 qp + 8 .. + 15 of ITS pixel
                auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * qp].x, pk[2 * qp + 1].x, false, false);
                auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * qp].y, pk[2 * qp + 1].y, false, false);
                if (ok) *reinterpret_cast<uint4*>(a.y + (row + cb + 16 * qp + 8 * half) * 2) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
        }
        if (a.gn_sums) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s1 = cv_half_wave_sum(gs[q]), s2 = cv_half_wave_sum(gq[q]);
                if ((lane & 31) == 31) {                                     // lanes 31 and 63 hold their half's totals: run index = (channel - n0) / 4
                    const uint32_t run = (cb - n0 + 8 * q + 4 * half) / 4;
                    atomicAdd(&red[run * 2], s1);
                    atomicAdd(&red[run * 2 + 1], s2);
                }
            }
        }
    }
    if (a.gn_sums) {                                                         // host guarantees: the tile lies in ONE sample, groups are multiples of 4 channels
        __syncthreads();
        const uint32_t cpg = a.Cout / a.G, hpg = cpg / 4, g0 = n0 / cpg, ng = (n0 + BN - 1) / cpg - g0 + 1;
        if (tid < ng && m0 < a.M) {
            const uint32_t lo = max((g0 + tid) * hpg, n0 / 4) - n0 / 4, hi = min((g0 + tid + 1) * hpg, (n0 + BN) / 4) - n0 / 4;
            float ss = 0.f, qq = 0.f;
            for (uint32_t i = lo; i < hi; ++i) { ss += red[i * 2]; qq += red[i * 2 + 1]; }
            double* dst = a.gn_sums + ((size_t)(m0 / (a.Ho * a.Wo)) * a.G + g0 + tid) * 2;
            atomicAdd(dst, (double)ss);
            atomicAdd(dst + 1, (double)qq);
        }
    }
}

typedef __attribute__((address_space(3))) void* cv_lds_ptr;
SSD_DEV __amdgpu_buffer_rsrc_t cv_rsrc(const void* base, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (uint32_t)bytes, 0x00020000);
}
// (a plain function on purpose: with the builtin written directly inside the five-parameter kernel template, this toolchain's HOST pass marks the
// instantiation invalid without a diagnostic and emits no launch stub -- an undefined `__device_stub__` at load time)
SSD_DEV void cv_dma16(__amdgpu_buffer_rsrc_t r, unsigned char* dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (cv_lds_ptr)dst, 16, voff, soff, 0, 0);
}
static constexpr uint32_t CV_OOB = 0x80000000u;                              // voffset of a padding / out-of-range row (tensors are < 2^31 bytes: host check)


template <int N> SSD_DEV void cv_wait_tiles_and_barrier() {
    // this wave's DMA of the tile about to be read has landed (N younger loads may stay in flight), its own LDS reads of the previous
    // tile have returned; after the barrier that holds for every wave, so the buffer of the previous tile may be refilled
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// bf16 epilogue shared by the bf16 kernels: accumulators -> fp32 tile in LDS (EP_ROWS rows per pass) -> (+bias, +residual) -> bf16 rows (+ GroupNorm sums)
template <int TM, int TN, int WM, int WN, bool PT = false>
SSD_DEV void cv_epilogue_bf16(const ConvArgs& a, f32x16 (&acc)[TM][TN], unsigned char* lds, uint32_t m0, uint32_t n0) {
    constexpr int NT = 64 * WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int EP_ROWS = BM < 128 ? BM : 128, EPI = EP_ROWS * BN * 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    float* tile_f = reinterpret_cast<float*>(lds);
    constexpr int CPR = BN / 8;                                              // 8-channel chunks per row; NT % CPR == 0, so a thread keeps its chunk column
    float* red = reinterpret_cast<float*>(lds + EPI);                        // [CPR][2 halves][sum, sumsq] block partials for the GroupNorm statistics
    if (a.gn_sums && tid < CPR * 4) red[tid] = 0.f;
    const uint32_t cc = tid % CPR, co = n0 + cc * 8;
    const bool c_ok = !PT || co < a.Cout;                                    // (PT: Cout is a multiple of 8 and need not fill the last N tile)
    float bias_v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias_v[k] = (a.bias && c_ok) ? a.bias[co + k] : 0.f;
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < BM / EP_ROWS; ++pass) {
        if (pass) __syncthreads();                                           // the previous pass has been read out
        if ((wm * 32 * TM) / EP_ROWS == (uint32_t)pass) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t row = (wm * 32 * TM) % EP_ROWS + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        const uint32_t col = wn * 32 * TN + j * 32 + (lane & 31);
                        tile_f[row * BN + col] = acc[i][j][e];
                    }
        }
        __syncthreads();
#pragma unroll 2
        for (uint32_t row = tid / CPR; row < (uint32_t)EP_ROWS; row += NT / CPR) {
            const uint32_t m = m0 + pass * EP_ROWS + row;
            if (m >= a.M || !c_ok) break;
            const float4 v0 = *reinterpret_cast<const float4*>(tile_f + row * BN + cc * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(tile_f + row * BN + cc * 8 + 4);
            float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] += bias_v[k];
            const size_t o = ((size_t)m * a.Cout + co) * 2;
            if (a.res) {
                const uint4 rv = *reinterpret_cast<const uint4*>(a.res + o);
                const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { f[2 * k] += __uint_as_float(rw[k] << 16); f[2 * k + 1] += __uint_as_float(rw[k] & 0xffff0000u); }
            }
            uint32_t pk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pk[k] = cv_pack_bf16(f[2 * k], f[2 * k + 1]);
            *reinterpret_cast<uint4*>(a.y + o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            if (a.gn_sums) {                                                 // statistics of what the next norm will read (the rounded values)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(pk[k] << 16), hi = __uint_as_float(pk[k] & 0xffff0000u);
                    gs[k >> 1] += lo + hi;
                    gq[k >> 1] = __builtin_fmaf(lo, lo, gq[k >> 1]);
                    gq[k >> 1] = __builtin_fmaf(hi, hi, gq[k >> 1]);
                }
            }
        }
    }
    if (a.gn_sums) {                                                         // host guarantees: the tile lies in ONE sample, groups are multiples of 4 channels
#pragma unroll
        for (int h = 0; h < 2; ++h) { atomicAdd(&red[(cc * 2 + h) * 2], gs[h]); atomicAdd(&red[(cc * 2 + h) * 2 + 1], gq[h]); }
        __syncthreads();
        const uint32_t n1 = PT ? min(n0 + BN, a.Cout) : n0 + BN;              // end of the tile's channels
        const uint32_t cpg = a.Cout / a.G, hpg = cpg / 4, g0 = n0 / cpg, ng = (n1 - 1) / cpg - g0 + 1;   // groups this tile touches (BN and cpg are multiples of 4)
        if (tid < ng && m0 < a.M) {
            // half chunks of group g0 + tid inside this tile: global half-chunk index range [g*hpg, (g+1)*hpg) minus the tile's first, n0/4
            const uint32_t lo = max((g0 + tid) * hpg, n0 / 4) - n0 / 4, hi = min((g0 + tid + 1) * hpg, n1 / 4) - n0 / 4;
            float ss = 0.f, qq = 0.f;
            for (uint32_t i = lo; i < hi; ++i) { ss += red[i * 2]; qq += red[i * 2 + 1]; }
            double* dst = a.gn_sums + ((size_t)(m0 / (a.Ho * a.Wo)) * a.G + g0 + tid) * 2;
            atomicAdd(dst, (double)ss);
            atomicAdd(dst + 1, (double)qq);
        }
    }
}

// TM x TN MFMA tiles (32 x 32) per wave, WM x WN waves per block, NS staging buffers.  PT ("partial tiles", r03): channel counts are multiples of
// 8 instead of 64 / the N tile -- the last K-tile of a tensor and the last N tile may be partly empty (the tiled UNet's 80 / 160 / 240 / 480-channel
// layers, the 18 -> 24-channel stem and head).  A separate instantiation: the masks cost the small layers of the cars UNet (4 MFMAs per K-tile and
// wave on the 64 x 64 tile) up to 30 % when they were unconditional (profiles/r03/f_partial_tiles_ab.txt).
template <int TM, int TN, bool PT = false> SSD_DEV void cv_epilogue_f32(const ConvArgs& a, f32x16 (&acc)[TM][TN], unsigned char* lds, uint32_t m0, uint32_t n0);

// PS (r04): the fp32-class form on PRE-SPLIT activations (what ssdnerf_group_norm_nhwc(..., act | 2) writes: per pixel and 32-channel block 32 bf16 hi
// terms, then the 32 lo terms -- the bytes of the fp32 tensor).  A K-tile is then 32 channels: its 128-byte A row is [hi | lo] straight from memory, its B
// row is fetched as [w_hi | w_lo] (w_lo lies directly behind w_hi), and a k-step pair multiplies lo * hi + hi * lo + hi * hi.  Same DMA ring, no
// register pass over the operands (k_conv_igemm_f32x2 loads fp32, splits in registers and writes LDS by hand -- per K-tile, per block).  fp32 output
// through cv_epilogue_f32 (bias, residual, GroupNorm sums) or fp32 split-K atomics + k_conv_f32_finish.
template <int TM, int TN, int WM, int WN, int NS, bool PT, bool PS = false>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_igemm_bf16(const ConvArgs a) {
    static_assert(!PS || (WM == 2 && WN == 2 && !PT), "the pre-split form shares cv_epilogue_f32's 2 x 2 wave grid and takes whole tiles only");
    constexpr int KCH = PS ? 32 : CV_BK;                                     // channels per K-tile
    constexpr int XB = PS ? 4 : 2;                                           // activation bytes per channel
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int A_INST = BM / 8 / NW, B_INST = BN / 8 / NW;                // global_load_lds instructions per wave per K-tile (8 rows each)
    constexpr int STAGE = (BM + BN) * CV_ROWB;
    constexpr int EP_ROWS = BM < 128 ? BM : 128;                             // the epilogue goes through LDS in passes of EP_ROWS rows
    constexpr int EPI = EP_ROWS * BN * 4;
    constexpr int LPT = A_INST + B_INST;                                     // DMA instructions per wave per K-tile
    constexpr int LDS_BYTES = ((NS * STAGE > EPI) ? NS * STAGE : EPI) + 512;   // + block partials of the fused GroupNorm statistics (ONE LDS object)
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0 && EP_ROWS % (32 * TM) == 0, "tile / wave grid mismatch");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: LDS-DMA destinations are wave-uniform)
    const uint32_t wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile order: block b runs on XCD b % 8; give each XCD a contiguous range of tiles (M-major) ----------------
    const uint32_t n_blocks = a.m_tiles * a.n_tiles * a.splits;
    uint32_t tile, split;
    {
        const uint32_t xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = n_blocks >> 3, r = n_blocks & 7;
        const uint32_t lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile = lin / a.splits; split = lin % a.splits;                       // a tile's K ranges run side by side on one XCD
    }
    const uint32_t m0 = (tile / a.n_tiles) * BM, n0 = (tile % a.n_tiles) * BN;

    // ---- loader geometry: instruction i of this wave covers tile rows (wave*INST + i)*8 + (lane >> 3) ------------------------------
    const uint32_t taps = a.ksize * a.ksize;
    const uint32_t Hv = a.upsample ? a.H * 2 : a.H, Wv = a.upsample ? a.W * 2 : a.W;   // the (virtual) image the taps move over
    int32_t a_y0[A_INST], a_x0[A_INST];
    uint32_t a_img[A_INST], a_sc[A_INST];
    bool a_ok[A_INST];
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const uint32_t r = (wave * A_INST + i) * 8 + (lane >> 3);
        const uint32_t m = m0 + r;
        a_ok[i] = m < a.M;
        const uint32_t mm = a_ok[i] ? m : 0;
        const uint32_t b = mm / (a.Ho * a.Wo), rem = mm % (a.Ho * a.Wo);
        a_y0[i] = (int32_t)((rem / a.Wo) * a.stride) - (int32_t)a.pad;
        a_x0[i] = (int32_t)((rem % a.Wo) * a.stride) - (int32_t)a.pad;
        a_img[i] = b * a.H * a.W;
        a_sc[i] = (lane & 7) ^ ((r >> 1) & 7);                               // source-side swizzle: the 8-channel chunk of the K-tile this lane fetches
    }
    uint32_t b_off[B_INST], b_sc[B_INST];
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        const uint32_t r = (wave * B_INST + i) * 8 + (lane >> 3);
        b_sc[i] = (lane & 7) ^ ((r >> 1) & 7);
        if (PS) b_off[i] = ((n0 + r) * taps * a.Cin) * 2 + (b_sc[i] & 3) * 16 + (b_sc[i] >> 2) * (a.Cout * taps * a.Cin * 2);   // chunks 0-3: 32 hi terms, 4-7: the lo terms
        else b_off[i] = (!PT || n0 + r < a.Cout) ? ((n0 + r) * taps * a.Cin) * 2 + b_sc[i] * 16 : CV_OOB;   // (PT: rows past Cout read zeros)
    }

    // r03: LDS-DMA through buffer descriptors (see cv_rsrc): one 32-bit offset per piece and tap, the channel tile as the scalar offset, padding taps
    // as an out-of-range offset that reads zeros -- the 64-bit source selects of r02 were most of the ~190 address instructions per K-tile
    const __amdgpu_buffer_rsrc_t rs_x = cv_rsrc(a.x, (uint64_t)a.B * a.H * a.W * a.Cin1 * XB);
    const __amdgpu_buffer_rsrc_t rs_x2 = cv_rsrc(a.x2 ? a.x2 : a.x, a.x2 ? (uint64_t)a.B * a.H * a.W * (a.Cin - a.Cin1) * 2 : 0);
    const __amdgpu_buffer_rsrc_t rs_w = cv_rsrc(a.w, (uint64_t)a.Cout * taps * a.Cin * 2 * (PS ? 2 : 1));
    uint32_t a_voff[A_INST], a_voff2[A_INST];
    const uint32_t Cin2 = a.Cin - a.Cin1;
    auto set_tap = [&](uint32_t tap) {
        const int32_t kh = (int32_t)(tap / a.ksize), kw = (int32_t)(tap % a.ksize);
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
            const int32_t yv = a_y0[i] + kh, xv = a_x0[i] + kw;
            const bool ok = a_ok[i] && yv >= 0 && xv >= 0 && yv < (int32_t)Hv && xv < (int32_t)Wv;
            const uint32_t yi = a.upsample ? (uint32_t)yv >> 1 : (uint32_t)yv, xi = a.upsample ? (uint32_t)xv >> 1 : (uint32_t)xv;
            const uint32_t pix = a_img[i] + yi * a.W + xi;
            a_voff[i] = ok ? pix * a.Cin1 * XB + a_sc[i] * 16 : CV_OOB;
            a_voff2[i] = ok ? pix * Cin2 * 2 + a_sc[i] * 16 : CV_OOB;
        }
    };

    // K-tiles of a tap: kc1 over the first tensor's channels, then kc2 over the second's; a tensor's LAST tile may hold fewer than 64 channels
    // (channel counts are multiples of 8, e.g. the 80 / 160 / 240 / 480-channel layers of the tiled UNet): its missing 16-byte chunks read zeros on
    // both operands, and the multiply loop runs only the 16-deep k-steps that hold channels
    const uint32_t kc1 = (a.Cin1 + KCH - 1) / KCH, kc = kc1 + (Cin2 + CV_BK - 1) / CV_BK;        // (PS: one tensor, whole 32-channel tiles)
    auto tile_channels = [&](uint32_t ci) {                                  // channels in K-tile ci of a tap
        if (!PT) return (uint32_t)CV_BK;
        const uint32_t left = ci >= kc1 ? Cin2 - (ci - kc1) * CV_BK : a.Cin1 - ci * CV_BK;
        return left < (uint32_t)CV_BK ? left : (uint32_t)CV_BK;
    };
    auto issue = [&](uint32_t tap, uint32_t ci, uint32_t buf) {
        unsigned char* sa = lds + buf * STAGE;
        unsigned char* sb = sa + BM * CV_ROWB;
        const bool second = ci >= kc1;                                       // K-tiles never straddle the two tensors
        const uint32_t cb = (second ? ci - kc1 : ci) * KCH, coff = cb * XB, nchunk = tile_channels(ci) >> 3;
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
            unsigned char* dst = sa + (wave * A_INST + i) * 1024;
            if (second) cv_dma16(rs_x2, dst, (!PT || a_sc[i] < nchunk) ? a_voff2[i] : CV_OOB, coff);
            else cv_dma16(rs_x, dst, (!PT || a_sc[i] < nchunk) ? a_voff[i] : CV_OOB, coff);
        }
#pragma unroll
        for (int i = 0; i < B_INST; ++i)
            cv_dma16(rs_w, sb + (wave * B_INST + i) * 1024, (!PT || b_sc[i] < nchunk) ? b_off[i] : CV_OOB, (tap * a.Cin + (second ? a.Cin1 : 0u) + cb) * 2);
    };

    // ---- reader geometry ---------------------------------------------------------------------------------------------------
    const uint32_t rd_row = lane & 31;
    const uint32_t rd_c0 = ((lane >> 5) ^ ((lane >> 1) & 7)) * 16;           // chunk of k-step 0; k-step s is ^ (s * 32)
    const uint32_t a_rd = (wm * 32 * TM + rd_row) * CV_ROWB + rd_c0;
    const uint32_t b_rd = BM * CV_ROWB + (wn * 32 * TN + rd_row) * CV_ROWB + rd_c0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const uint32_t KT_all = taps * kc;
    const uint32_t kt_begin = (uint32_t)((uint64_t)split * KT_all / a.splits), KT = (uint32_t)((uint64_t)(split + 1) * KT_all / a.splits) - kt_begin;
    // NS-stage pipeline, one barrier per K-tile: tiles kt+1 .. kt+NS-2 stay in flight while tile kt is multiplied
    uint32_t tap = kt_begin / kc, ci = kt_begin % kc, issued = 0;
    uint32_t ci_mul = ci;                                                    // the multiply loop's own position inside the tap
    set_tap(tap);
    auto issue_next = [&]() {
        if (issued) { if (++ci == kc) { ci = 0; ++tap; set_tap(tap); } }
        issue(tap, ci, issued % NS);
        ++issued;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (issued < KT) issue_next();
    for (uint32_t kt = 0; kt < KT; ++kt) {
        const uint32_t in_flight = issued - (kt + 1);                        // tiles younger than kt already issued (0 .. NS-2)
        if (NS >= 4 && in_flight >= 2) cv_wait_tiles_and_barrier<2 * LPT>();
        else if (NS >= 3 && in_flight == 1) cv_wait_tiles_and_barrier<1 * LPT>();
        else cv_wait_tiles_and_barrier<0>();
        if (issued < KT) issue_next();                                       // refills the buffer tile kt-1 was read from
        const unsigned char* st = lds + (kt % NS) * STAGE;
        const uint32_t ksteps = (tile_channels(ci_mul) + 15) >> 4;           // 16-deep k-steps that hold channels (4 except in a tensor's last tile)
        if (PT && ++ci_mul == kc) ci_mul = 0;
        if (!PT || ksteps == 4) {
            // all of the K-tile's fragments are requested before the first MFMA: the LDS latency is paid once per tile, not per k-step
            bf16x8 fa[4][TM], fb[4][TN];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(st + ((a_rd + i * 32 * CV_ROWB) ^ (s * 32)));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(st + ((b_rd + j * 32 * CV_ROWB) ^ (s * 32)));
            }
            __builtin_amdgcn_sched_barrier(0);                               // keep the reads clustered ahead of the MFMAs (the scheduler would re-serialise them)
            if (PS) {                                                        // k-steps 0, 1: the hi terms of channels 0-15, 16-31; 2, 3: their lo terms
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 + s][i], fb[s][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[2 + s][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
                        }
            } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
            }
        } else {
            for (uint32_t s = 0; s < ksteps; ++s) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(st + ((a_rd + i * 32 * CV_ROWB) ^ (s * 32)));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(st + ((b_rd + j * 32 * CV_ROWB) ^ (s * 32)));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                                         // every wave is done with the staging buffers (the epilogue reuses them)

    // ---- split-K: add this block's partial sums to the fp32 workspace; k_conv_splitk_finish turns it into the output ------------------
    if (a.splits > 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t m = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const uint32_t col = n0 + wn * 32 * TN + j * 32 + (lane & 31);
                    if (m < a.M && (!PT || col < a.Cout)) unsafeAtomicAdd(a.splitk_ws + (size_t)m * a.Cout + col, acc[i][j][e]);
                }
        if (a.tickets == nullptr) return;
        if (!cv_fold_last<TM, TN>(a, acc, a.splitk_ws, reinterpret_cast<uint32_t*>(lds + LDS_BYTES - 4), tile, m0, n0, wm * 32 * TM, wn * 32 * TN, PT)) return;
    }

    if constexpr (PS) cv_epilogue_f32<TM, TN, false>(a, acc, lds, m0, n0);
    else cv_epilogue_bf16<TM, TN, WM, WN, PT>(a, acc, lds, m0, n0);
}

// fp32 epilogue shared by the fp32-class kernels: accumulators -> fp32 tile in LDS -> (+bias, +residual) -> fp32 rows (+ GroupNorm sums)
template <int TM, int TN, bool PT>
SSD_DEV void cv_epilogue_f32(const ConvArgs& a, f32x16 (&acc)[TM][TN], unsigned char* lds, uint32_t m0, uint32_t n0) {
    constexpr int BM = 64 * TM, BN = 64 * TN, EPI = BM * BN * 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    float* tile_f = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                tile_f[(wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * BN + wn * 32 * TN + j * 32 + (lane & 31)] = acc[i][j][e];
    constexpr int CPR = BN / 8;
    float* red = reinterpret_cast<float*>(lds + EPI);
    if (a.gn_sums && tid < CPR * 4) red[tid] = 0.f;
    __syncthreads();
    const uint32_t cc = tid % CPR, co = n0 + cc * 8;
    const bool c_ok = !PT || co < a.Cout;
    float bias_v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias_v[k] = (a.bias && c_ok) ? a.bias[co + k] : 0.f;
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
    float* yo = reinterpret_cast<float*>(a.y);
    const float* ro = reinterpret_cast<const float*>(a.res);
#pragma unroll 2
    for (uint32_t row = tid / CPR; row < (uint32_t)BM; row += 256 / CPR) {
        const uint32_t m = m0 + row;
        if (m >= a.M || !c_ok) break;
        const float4 v0 = *reinterpret_cast<const float4*>(tile_f + row * BN + cc * 8), v1 = *reinterpret_cast<const float4*>(tile_f + row * BN + cc * 8 + 4);
        float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] += bias_v[k];
        const size_t o = (size_t)m * a.Cout + co;
        if (ro) {
            const float4 r0 = *reinterpret_cast<const float4*>(ro + o), r1 = *reinterpret_cast<const float4*>(ro + o + 4);
            f[0] += r0.x; f[1] += r0.y; f[2] += r0.z; f[3] += r0.w; f[4] += r1.x; f[5] += r1.y; f[6] += r1.z; f[7] += r1.w;
        }
        *reinterpret_cast<float4*>(yo + o) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(yo + o + 4) = make_float4(f[4], f[5], f[6], f[7]);
        if (a.gn_sums) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { gs[k >> 2] += f[k]; gq[k >> 2] = __builtin_fmaf(f[k], f[k], gq[k >> 2]); }
        }
    }
    if (a.gn_sums) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { atomicAdd(&red[(cc * 2 + h) * 2], gs[h]); atomicAdd(&red[(cc * 2 + h) * 2 + 1], gq[h]); }
        __syncthreads();
        const uint32_t n1 = PT ? min(n0 + BN, a.Cout) : n0 + BN;
        const uint32_t cpg = a.Cout / a.G, hpg = cpg / 4, g0 = n0 / cpg, ng = (n1 - 1) / cpg - g0 + 1;
        if (tid < ng && m0 < a.M) {
            const uint32_t lo = max((g0 + tid) * hpg, n0 / 4) - n0 / 4, hi = min((g0 + tid + 1) * hpg, n1 / 4) - n0 / 4;
            float ss = 0.f, qq = 0.f;
            for (uint32_t i = lo; i < hi; ++i) { ss += red[i * 2]; qq += red[i * 2 + 1]; }
            double* dst = a.gn_sums + ((size_t)(m0 / (a.Ho * a.Wo)) * a.G + g0 + tid) * 2;
            atomicAdd(dst, (double)ss);
            atomicAdd(dst + 1, (double)qq);
        }
    }
}

// 3x3 / stride 1 variant of the bf16 kernel with the A tile loaded once per (kh, channel tile) and the three kw taps served from it by
// shifted LDS reads (zero rows in front of every image row of the tile and after the last; see k_conv3x3_f32x2_rows).  The A rows still
// arrive by global_load_lds: an instruction's 8 consecutive tile rows lie in one image row (W is a multiple of 8), so they are 8 consecutive
// LDS rows too.  128 x 128 tile, K-tile 64, two buffers per operand, no split-K.
__global__ __launch_bounds__(256) void k_conv3x3_bf16_rows(const ConvArgs a) {
    constexpr int TM = 2, TN = 2, BM = 128, BN = 128;
    constexpr int A_INST = BM / 32, B_INST = BN / 32;
    constexpr int A_ROWS = BM + 4 + 1, A_BUF = A_ROWS * CV_ROWB, B_BUF = BN * CV_ROWB;
    constexpr int EPI = BM * BN * 4;
    constexpr int LDS_BYTES = ((2 * A_BUF + 2 * B_BUF > EPI) ? 2 * A_BUF + 2 * B_BUF : EPI) + 512;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    unsigned char* const abuf = lds;
    unsigned char* const bbuf = lds + 2 * A_BUF;

    const uint32_t n_blocks = a.m_tiles * a.n_tiles;
    uint32_t tile;
    {
        const uint32_t xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = n_blocks >> 3, r = n_blocks & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const uint32_t m0 = (tile / a.n_tiles) * BM, n0 = (tile % a.n_tiles) * BN;
    const uint32_t W = a.W, Cin2 = a.Cin - a.Cin1;
    const uint32_t n_seg = W >= (uint32_t)BM ? 1u : (uint32_t)BM / W, seg_w = W >= (uint32_t)BM ? (uint32_t)BM : W;
    for (uint32_t i = tid; i < 2 * (n_seg + 1) * (CV_ROWB / 16); i += 256) {   // zero rows of both A buffers
        const uint32_t chunk = i % (CV_ROWB / 16), row = (i / (CV_ROWB / 16)) % (n_seg + 1), buf = i / ((CV_ROWB / 16) * (n_seg + 1));
        *reinterpret_cast<uint4*>(abuf + buf * A_BUF + row * (seg_w + 1) * CV_ROWB + chunk * 16) = make_uint4(0, 0, 0, 0);
    }
    // ---- A loader: instruction i of this wave covers tile rows (wave*A_INST + i)*8 + (lane >> 3) = LDS rows rho0 + (lane >> 3) -------------
    int32_t a_y0[A_INST];
    uint32_t a_pix0[A_INST], a_chunk[A_INST], a_lds[A_INST];
    bool a_ok[A_INST];
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const uint32_t r8 = (wave * A_INST + i) * 8, r = r8 + (lane >> 3), m = m0 + r;
        a_ok[i] = m < a.M;
        const uint32_t mm = a_ok[i] ? m : 0;
        const uint32_t b = mm / (a.H * W), rem = mm % (a.H * W);
        a_y0[i] = (int32_t)(rem / W) - 1;
        a_pix0[i] = b * a.H * W + (rem % W);
        const uint32_t rho = r + r / seg_w + 1;
        a_chunk[i] = ((lane & 7) ^ ((rho >> 1) & 7)) * 16;                       // source-side swizzle keyed by the LDS row
        a_lds[i] = (r8 + r8 / seg_w + 1) * CV_ROWB;                              // LDS byte offset of the instruction's first row (wave-uniform)
    }
    uint32_t b_off[B_INST];
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        const uint32_t r = (wave * B_INST + i) * 8 + (lane >> 3);
        b_off[i] = ((n0 + r) * 9 * a.Cin) * 2 + ((lane & 7) ^ ((r >> 1) & 7)) * 16;
    }
    uint64_t a_src[A_INST], a_src2[A_INST];
    bool a_zero[A_INST];
    auto set_kh = [&](uint32_t kh) {
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
            const int32_t yv = a_y0[i] + (int32_t)kh;
            const bool ok = a_ok[i] && yv >= 0 && yv < (int32_t)a.H;
            const uint64_t pix = (uint64_t)a_pix0[i] + (uint64_t)(ok ? yv : 0) * W;
            a_zero[i] = !ok;
            a_src[i] = ok ? (uint64_t)a.x + pix * a.Cin1 * 2 + a_chunk[i] : (uint64_t)g_conv_zero_page;
            a_src2[i] = (ok && a.x2) ? (uint64_t)a.x2 + pix * Cin2 * 2 + a_chunk[i] : (uint64_t)g_conv_zero_page;
        }
    };
    auto a_issue = [&](uint32_t ci0, uint32_t buf) {
        const bool second = ci0 >= a.Cin1;
        const uint64_t coff = (uint64_t)(second ? ci0 - a.Cin1 : ci0) * 2;
#pragma unroll
        for (int i = 0; i < A_INST; ++i)
            cv_glds16((const void*)((second ? a_src2[i] : a_src[i]) + (a_zero[i] ? 0 : coff)), abuf + buf * A_BUF + __builtin_amdgcn_readfirstlane(a_lds[i]));
    };
    auto b_issue = [&](uint32_t tap, uint32_t ci0, uint32_t buf) {
#pragma unroll
        for (int i = 0; i < B_INST; ++i)
            cv_glds16(a.w + b_off[i] + (uint64_t)(tap * a.Cin + ci0) * 2, bbuf + buf * B_BUF + (wave * B_INST + i) * 1024);
    };
    // ---- reader geometry ---------------------------------------------------------------------------------------------------------------
    uint32_t a_rd[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const uint32_t t0 = wm * 32 * TM + i * 32;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const uint32_t rho = t0 + (lane & 31) + t0 / seg_w + kw;
            a_rd[i][kw] = rho * CV_ROWB + (((lane >> 5) ^ ((rho >> 1) & 7)) * 16);
        }
    }
    const uint32_t b_rd = (wn * 32 * TN + (lane & 31)) * CV_ROWB + (((lane >> 5) ^ ((lane >> 1) & 7)) * 16);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const uint32_t kc = a.Cin / CV_BK, G = 3 * kc, KT = 3 * G;
    uint32_t kh = 0, ci = 0;
    set_kh(0);
    a_issue(0, 0);
    b_issue(0, 0, 0);
    __syncthreads();
    for (uint32_t kt = 0, g = 0, kw = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) {
            const uint32_t kw1 = kw == 2 ? 0 : kw + 1;
            uint32_t kh1 = kh, ci1 = ci;
            if (kw == 2) { if (++ci1 == kc) { ci1 = 0; ++kh1; } }
            b_issue(kh1 * 3 + kw1, ci1 * CV_BK, (kt + 1) & 1);
        }
        if (kw == 0 && g + 1 < G) {                                          // the next group's A tile; its buffer was last read two barriers ago
            uint32_t kh1 = kh, ci1 = ci;
            if (++ci1 == kc) { ci1 = 0; ++kh1; set_kh(kh1); }
            a_issue(ci1 * CV_BK, (g + 1) & 1);
        }
        const unsigned char* sa = abuf + (g & 1) * A_BUF;
        const unsigned char* sb = bbuf + (kt & 1) * B_BUF;
        bf16x8 fa[4][TM], fb[4][TN];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const uint32_t off = kw == 0 ? a_rd[i][0] : kw == 1 ? a_rd[i][1] : a_rd[i][2];
                fa[s][i] = *reinterpret_cast<const bf16x8*>(sa + (off ^ (s * 32)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(sb + ((b_rd + j * 32 * CV_ROWB) ^ (s * 32)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
        __syncthreads();
        if (++kw == 3) { kw = 0; ++g; if (++ci == kc) { ci = 0; ++kh; } }
    }
    cv_epilogue_bf16<TM, TN, 2, 2>(a, acc, lds, m0, n0);
}

// ================================================================================================================================
// fp32 activations, fp32-class products on the bf16 matrix cores ("bf16 x 2"): every fp32 operand is split into two bf16 terms
// (x ~ hi + lo, 16 significand bits, relative error <= 2^-16; TF32 -- what the reference's cuDNN convolutions use on Ampere -- has 11)
// and hi*hi + hi*lo + lo*hi are accumulated in fp32.  For configs that run the UNet in fp32 (every paper config but the fp16 one).
//   * x fp32 [B][H][W][Cin]: the A tile goes global -> registers -> (split) -> LDS, one K-tile ahead of the MFMAs (the loads of tile kt+1
//     are issued before the MFMAs of tile kt and written to LDS after them); weights are pre-split on the host into two bf16 tensors and
//     go L2 -> LDS by global_load_lds like the bf16 kernel's;
//   * K-tile 32 (64-byte rows, chunks XOR-swizzled by (row >> 2) & 3: conflict-free ds_read_b128), 24 MFMAs per K-tile and wave for the
//     128 x 128 tile -- three times the bf16 kernel's MFMA work per byte of L2 traffic, which is what that DMA-issue-bound kernel had spare;
//   * epilogue as in the bf16 kernel, fp32 out (+ bias, + fp32 residual, + GroupNorm sums).
template <int TM, int TN, bool PT>
__global__ __launch_bounds__(256) void k_conv_igemm_f32x2(const ConvArgs a, const unsigned char* __restrict__ w_lo) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, ROWB = BK * 2;          // 64-byte bf16 rows
    constexpr int A_PIECES = BM * (BK / 4) / 256;                              // 16-byte fp32 pieces (4 channels) per thread per K-tile
    constexpr int B_INST = BN / 16 / 4;                                        // DMA instructions (16 rows x 64 B) per wave per K-tile and term
    constexpr int STAGE = 2 * (BM + BN) * ROWB;                                // [A hi | A lo | B hi | B lo]
    constexpr int EPI = BM * BN * 4;
    constexpr int LDS_BYTES = ((2 * STAGE > EPI) ? 2 * STAGE : EPI) + 512;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;

    const uint32_t n_blocks = a.m_tiles * a.n_tiles * a.splits;
    uint32_t tile, split;
    {
        const uint32_t xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = n_blocks >> 3, r = n_blocks & 7;
        const uint32_t lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile = lin / a.splits; split = lin % a.splits;
    }
    const uint32_t m0 = (tile / a.n_tiles) * BM, n0 = (tile % a.n_tiles) * BN;
    const uint32_t taps = a.ksize * a.ksize;
    const uint32_t Hv = a.upsample ? a.H * 2 : a.H, Wv = a.upsample ? a.W * 2 : a.W;
    const uint32_t Cin2 = a.Cin - a.Cin1;

    // ---- A loader: piece p = tid + 256 i -> tile row p / 8, channels 4 (p % 8) .. + 3 of the K-tile ------------------------------------
    int32_t a_y0[A_PIECES], a_x0[A_PIECES];
    uint32_t a_img[A_PIECES], a_dst[A_PIECES];
    bool a_ok[A_PIECES];
    const uint32_t c4 = tid & 7;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const uint32_t r = (tid >> 3) + 32 * i, m = m0 + r;
        a_ok[i] = m < a.M;
        const uint32_t mm = a_ok[i] ? m : 0;
        const uint32_t b = mm / (a.Ho * a.Wo), rem = mm % (a.Ho * a.Wo);
        a_y0[i] = (int32_t)((rem / a.Wo) * a.stride) - (int32_t)a.pad;
        a_x0[i] = (int32_t)((rem % a.Wo) * a.stride) - (int32_t)a.pad;
        a_img[i] = b * a.H * a.W;
        a_dst[i] = r * ROWB + (((c4 >> 1) ^ ((r >> 2) & 3)) * 16) + (c4 & 1) * 8;         // swizzled 8-byte slot of this piece's 4 bf16
    }
    const float* a_src[A_PIECES];
    const float* a_src2[A_PIECES];
    auto set_tap = [&](uint32_t tap) {
        const int32_t kh = (int32_t)(tap / a.ksize), kw = (int32_t)(tap % a.ksize);
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const int32_t yv = a_y0[i] + kh, xv = a_x0[i] + kw;
            const bool ok = a_ok[i] && yv >= 0 && xv >= 0 && yv < (int32_t)Hv && xv < (int32_t)Wv;
            const uint32_t yi = a.upsample ? (uint32_t)yv >> 1 : (uint32_t)yv, xi = a.upsample ? (uint32_t)xv >> 1 : (uint32_t)xv;
            const uint64_t pix = (uint64_t)(a_img[i] + yi * a.W + xi);
            a_src[i] = ok ? reinterpret_cast<const float*>(a.x) + pix * a.Cin1 + c4 * 4 : nullptr;
            a_src2[i] = (ok && a.x2) ? reinterpret_cast<const float*>(a.x2) + pix * Cin2 + c4 * 4 : nullptr;
        }
    };
    // K-tiles of a tap: kc1 over the first tensor's channels, then the second's; a tensor's last tile may hold fewer than 32 channels (channel counts
    // are multiples of 8): its missing pieces are zeros on both operands and the multiply loop runs only the k-steps that hold channels
    const uint32_t kc1 = (a.Cin1 + BK - 1) / BK, kc = kc1 + (Cin2 + BK - 1) / BK;
    auto tile_channels = [&](uint32_t ci) {
        if (!PT) return (uint32_t)BK;
        const uint32_t left = ci >= kc1 ? Cin2 - (ci - kc1) * BK : a.Cin1 - ci * BK;
        return left < (uint32_t)BK ? left : (uint32_t)BK;
    };
    float4 a_reg[A_PIECES];
    auto a_load = [&](uint32_t ci) {
        const bool second = ci >= kc1;
        const uint32_t coff = (second ? ci - kc1 : ci) * BK;
        const bool c_ok = !PT || c4 * 4 < tile_channels(ci);
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const float* p = second ? a_src2[i] : a_src[i];
            a_reg[i] = (p && c_ok) ? *reinterpret_cast<const float4*>(p + coff) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto a_store = [&](uint32_t buf) {                                          // split and park: hi = truncation to bf16, lo = truncation of the exact remainder
        unsigned char* sa = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const float v[4] = {a_reg[i].x, a_reg[i].y, a_reg[i].z, a_reg[i].w};
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hi[k] = __float_as_uint(v[k]) & 0xffff0000u;
                lo[k] = __float_as_uint(v[k] - __uint_as_float(hi[k]));
            }
            *reinterpret_cast<uint2*>(sa + a_dst[i]) = make_uint2(__builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u), __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u));
            *reinterpret_cast<uint2*>(sa + BM * ROWB + a_dst[i]) = make_uint2(__builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u), __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u));
        }
    };
    // ---- B loader: DMA instruction i of this wave covers rows (wave*B_INST + i)*16 + (lane >> 2), 16-byte chunk lane & 3 ----------------
    uint32_t b_off[B_INST], b_sc[B_INST];
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);               // (LDS-DMA destinations are wave-uniform)
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        const uint32_t r = (wave_u * B_INST + i) * 16 + (lane >> 2);
        b_sc[i] = (lane & 3) ^ ((r >> 2) & 3);
        b_off[i] = (!PT || n0 + r < a.Cout) ? ((n0 + r) * taps * a.Cin) * 2 + b_sc[i] * 16 : CV_OOB;   // PT: rows past Cout (last N tile) read zeros
    }
    const __amdgpu_buffer_rsrc_t rs_wh = cv_rsrc(a.w, (uint64_t)a.Cout * taps * a.Cin * 2), rs_wl = cv_rsrc(w_lo, (uint64_t)a.Cout * taps * a.Cin * 2);
    auto b_issue = [&](uint32_t tap, uint32_t ci, uint32_t buf) {
        unsigned char* sb = lds + buf * STAGE + 2 * BM * ROWB;
        const bool second = ci >= kc1;
        const uint32_t koff = (tap * a.Cin + (second ? a.Cin1 + (ci - kc1) * BK : ci * BK)) * 2, nchunk = tile_channels(ci) >> 3;
#pragma unroll
        for (int i = 0; i < B_INST; ++i) {
            const uint32_t v = (!PT || b_sc[i] < nchunk) ? b_off[i] : CV_OOB;
            cv_dma16(rs_wh, sb + (wave_u * B_INST + i) * 1024, v, koff);
            cv_dma16(rs_wl, sb + BN * ROWB + (wave_u * B_INST + i) * 1024, v, koff);
        }
    };
    // ---- reader geometry ---------------------------------------------------------------------------------------------------------------
    const uint32_t rd_row = lane & 31;
    const uint32_t rd_c0 = ((lane >> 5) ^ ((lane >> 2) & 3)) * 16;             // chunk of k-step 0 (rows of a 32-row tile: (row >> 2) & 3 == (lane >> 2) & 3); k-step 1 is ^ 32
    const uint32_t a_rd = (wm * 32 * TM + rd_row) * ROWB + rd_c0;
    const uint32_t b_rd = 2 * BM * ROWB + (wn * 32 * TN + rd_row) * ROWB + rd_c0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const uint32_t KT_all = taps * kc;
    const uint32_t kt_begin = (uint32_t)((uint64_t)split * KT_all / a.splits), KT = (uint32_t)((uint64_t)(split + 1) * KT_all / a.splits) - kt_begin;
    uint32_t tap = kt_begin / kc, ci = kt_begin % kc;
    set_tap(tap);
    a_load(ci);
    b_issue(tap, ci, 0);
    a_store(0);
    __syncthreads();
    for (uint32_t kt = 0; kt < KT; ++kt) {
        const uint32_t buf = kt & 1;
        const bool more = kt + 1 < KT;
        const uint32_t ksteps = (tile_channels(ci) + 15) >> 4;               // of the tile multiplied now (ci still names it)
        if (more) {
            if (++ci == kc) { ci = 0; ++tap; set_tap(tap); }
            a_load(ci);                                                      // tile kt+1: fp32 -> registers (in flight under the MFMAs below)
            b_issue(tap, ci, buf ^ 1);
        }
        const unsigned char* st = lds + buf * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (PT && s2 >= (int)ksteps) break;
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(st + ((a_rd + i * 32 * ROWB) ^ (s2 * 32)));
                al[i] = *reinterpret_cast<const bf16x8*>(st + BM * ROWB + ((a_rd + i * 32 * ROWB) ^ (s2 * 32)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(st + ((b_rd + j * 32 * ROWB) ^ (s2 * 32)));
                bl[j] = *reinterpret_cast<const bf16x8*>(st + BN * ROWB + ((b_rd + j * 32 * ROWB) ^ (s2 * 32)));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) a_store(buf ^ 1);                                          // the other buffer was last read in iteration kt-1 (barrier below it)
        __syncthreads();
    }

    // ---- split-K: partial sums are added into the caller's all-zero fp32 scratch (or, without one, straight into the pre-zeroed output);
    // k_conv_f32_finish adds bias / residual / statistics ------
    if (a.splits > 1) {
        float* yo = a.splitk_ws ? a.splitk_ws : reinterpret_cast<float*>(a.y);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t m = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const uint32_t col = n0 + wn * 32 * TN + j * 32 + (lane & 31);
                    if (m < a.M && (!PT || col < a.Cout)) unsafeAtomicAdd(yo + (size_t)m * a.Cout + col, acc[i][j][e]);
                }
        if (a.tickets == nullptr) return;
        if (!cv_fold_last<TM, TN>(a, acc, yo, reinterpret_cast<uint32_t*>(lds + LDS_BYTES - 4), tile, m0, n0, wm * 32 * TM, wn * 32 * TN, PT)) return;
    }

    cv_epilogue_f32<TM, TN, PT>(a, acc, lds, m0, n0);
}

// 3x3 / stride 1 variant of k_conv_igemm_f32x2 that loads the A tile ONCE per (kh, channel tile) and serves the three kw taps from it:
// the 128 output pixels of a tile are whole image rows (W in {32, 64, 128}), so the tap kw just reads the tile shifted by kw - 1 pixels,
// and the pixels it shifts in at a row's ends are padding -- a zero row kept in LDS in front of every image row and after the last
// (pixel x of image row r of the tile sits in LDS row 1 + r (W + 1) + x).  A loads per K-tile drop to a third (the kernel is bound by the
// number of memory instructions it has to issue, not by their bytes).  128 x 128 tile, no split-K.
__global__ __launch_bounds__(256) void k_conv3x3_f32x2_rows(const ConvArgs a, const unsigned char* __restrict__ w_lo) {
    constexpr int TM = 2, TN = 2, BM = 128, BN = 128, BK = 32, ROWB = BK * 2;
    constexpr int A_PIECES = BM * (BK / 4) / 256, B_INST = BN / 16 / 4;
    constexpr int A_ROWS = BM + 4 + 1;                                         // up to four image rows per tile (W = 32) + their zero rows
    constexpr int A_BUF = 2 * A_ROWS * ROWB;                                   // [hi | lo]
    constexpr int B_BUF = 2 * BN * ROWB;                                       // [hi | lo]
    constexpr int EPI = BM * BN * 4;
    constexpr int LDS_BYTES = ((2 * A_BUF + 2 * B_BUF > EPI) ? 2 * A_BUF + 2 * B_BUF : EPI) + 512;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    unsigned char* const abuf = lds;                                           // two A buffers, then two B buffers
    unsigned char* const bbuf = lds + 2 * A_BUF;

    const uint32_t n_blocks = a.m_tiles * a.n_tiles;
    uint32_t tile;
    {
        const uint32_t xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = n_blocks >> 3, r = n_blocks & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const uint32_t m0 = (tile / a.n_tiles) * BM, n0 = (tile % a.n_tiles) * BN;
    const uint32_t W = a.W, Cin2 = a.Cin - a.Cin1;
    const uint32_t n_seg = W >= (uint32_t)BM ? 1u : (uint32_t)BM / W, seg_w = W >= (uint32_t)BM ? (uint32_t)BM : W;

    // zero rows (never overwritten): LDS rows s * (seg_w + 1), s = 0 .. n_seg, of both terms of both A buffers
    for (uint32_t i = tid; i < 2 * 2 * (n_seg + 1) * (ROWB / 16); i += 256) {
        const uint32_t chunk = i % (ROWB / 16), row = (i / (ROWB / 16)) % (n_seg + 1), which = i / ((ROWB / 16) * (n_seg + 1));   // which: buffer * 2 + term
        *reinterpret_cast<uint4*>(abuf + (which >> 1) * A_BUF + (which & 1) * A_ROWS * ROWB + row * (seg_w + 1) * ROWB + chunk * 16) = make_uint4(0, 0, 0, 0);
    }

    // ---- A loader: piece p = tid + 256 i -> tile row p / 8 (an output pixel; its own x, the kw shift happens at read time) -----------------
    int32_t a_y0[A_PIECES];
    uint32_t a_pix0[A_PIECES], a_dst[A_PIECES];
    bool a_ok[A_PIECES];
    const uint32_t c4 = tid & 7;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const uint32_t r = (tid >> 3) + 32 * i, m = m0 + r;
        a_ok[i] = m < a.M;
        const uint32_t mm = a_ok[i] ? m : 0;
        const uint32_t b = mm / (a.H * W), rem = mm % (a.H * W);
        a_y0[i] = (int32_t)(rem / W) - 1;
        a_pix0[i] = b * a.H * W + (rem % W);                                    // + y * W
        const uint32_t rho = r + r / seg_w + 1;                                 // LDS row of this pixel
        a_dst[i] = rho * ROWB + (((c4 >> 1) ^ ((rho >> 2) & 3)) * 16) + (c4 & 1) * 8;
    }
    const float* a_src[A_PIECES];
    const float* a_src2[A_PIECES];
    auto set_kh = [&](uint32_t kh) {
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const int32_t yv = a_y0[i] + (int32_t)kh;
            const bool ok = a_ok[i] && yv >= 0 && yv < (int32_t)a.H;
            const uint64_t pix = (uint64_t)a_pix0[i] + (uint64_t)(ok ? yv : 0) * W;
            a_src[i] = ok ? reinterpret_cast<const float*>(a.x) + pix * a.Cin1 + c4 * 4 : nullptr;
            a_src2[i] = (ok && a.x2) ? reinterpret_cast<const float*>(a.x2) + pix * Cin2 + c4 * 4 : nullptr;
        }
    };
    float4 a_reg[A_PIECES];
    auto a_load = [&](uint32_t ci0) {
        const bool second = ci0 >= a.Cin1;
        const uint32_t coff = second ? ci0 - a.Cin1 : ci0;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const float* p = second ? a_src2[i] : a_src[i];
            a_reg[i] = p ? *reinterpret_cast<const float4*>(p + coff) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto a_store = [&](uint32_t buf) {
        unsigned char* sa = abuf + buf * A_BUF;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const float v[4] = {a_reg[i].x, a_reg[i].y, a_reg[i].z, a_reg[i].w};
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hi[k] = __float_as_uint(v[k]) & 0xffff0000u;
                lo[k] = __float_as_uint(v[k] - __uint_as_float(hi[k]));
            }
            *reinterpret_cast<uint2*>(sa + a_dst[i]) = make_uint2(__builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u), __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u));
            *reinterpret_cast<uint2*>(sa + A_ROWS * ROWB + a_dst[i]) = make_uint2(__builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u), __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u));
        }
    };
    uint32_t b_off[B_INST];
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        const uint32_t r = (wave * B_INST + i) * 16 + (lane >> 2);
        b_off[i] = ((n0 + r) * 9 * a.Cin) * 2 + ((lane & 3) ^ ((r >> 2) & 3)) * 16;
    }
    auto b_issue = [&](uint32_t tap, uint32_t ci0, uint32_t buf) {
        unsigned char* sb = bbuf + buf * B_BUF;
        const uint64_t koff = (uint64_t)(tap * a.Cin + ci0) * 2;
#pragma unroll
        for (int i = 0; i < B_INST; ++i) {
            cv_glds16(a.w + b_off[i] + koff, sb + (wave * B_INST + i) * 1024);
            cv_glds16(w_lo + b_off[i] + koff, sb + BN * ROWB + (wave * B_INST + i) * 1024);
        }
    };
    // ---- reader geometry: MFMA tile i of this wave covers tile rows t0 = wm*64 + i*32 .. +31, all in image row (segment) t0 / seg_w -------
    uint32_t a_rd[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const uint32_t t0 = wm * 32 * TM + i * 32;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const uint32_t rho = t0 + (lane & 31) + t0 / seg_w + kw;            // = 1 + seg (seg_w + 1) + x + (kw - 1)
            a_rd[i][kw] = rho * ROWB + (((lane >> 5) ^ ((rho >> 2) & 3)) * 16);
        }
    }
    const uint32_t b_rd = (wn * 32 * TN + (lane & 31)) * ROWB + (((lane >> 5) ^ ((lane >> 2) & 3)) * 16);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const uint32_t kc = a.Cin / BK, G = 3 * kc, KT = 3 * G;                     // groups (kh, ci tile); K-tiles (group, kw)
    uint32_t kh = 0, ci = 0;
    set_kh(0);
    a_load(0);
    b_issue(0, 0, 0);
    a_store(0);
    __syncthreads();
    for (uint32_t kt = 0, g = 0, kw = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) {
            const uint32_t kw1 = kw == 2 ? 0 : kw + 1;
            uint32_t kh1 = kh, ci1 = ci;
            if (kw == 2) { if (++ci1 == kc) { ci1 = 0; ++kh1; } }
            b_issue(kh1 * 3 + kw1, ci1 * BK, (kt + 1) & 1);
        }
        const bool next_group = kw == 0 && g + 1 < G;
        if (next_group) {                                                    // the next group's A tile: fp32 -> registers now, split -> LDS after the MFMAs
            uint32_t kh1 = kh, ci1 = ci;
            if (++ci1 == kc) { ci1 = 0; ++kh1; set_kh(kh1); }
            a_load(ci1 * BK);
        }
        const unsigned char* sa = abuf + (g & 1) * A_BUF;
        const unsigned char* sb = bbuf + (kt & 1) * B_BUF;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const uint32_t off = kw == 0 ? a_rd[i][0] : kw == 1 ? a_rd[i][1] : a_rd[i][2];
                ah[i] = *reinterpret_cast<const bf16x8*>(sa + (off ^ (s2 * 32)));
                al[i] = *reinterpret_cast<const bf16x8*>(sa + A_ROWS * ROWB + (off ^ (s2 * 32)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(sb + ((b_rd + j * 32 * ROWB) ^ (s2 * 32)));
                bl[j] = *reinterpret_cast<const bf16x8*>(sb + BN * ROWB + ((b_rd + j * 32 * ROWB) ^ (s2 * 32)));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (next_group) a_store((g + 1) & 1);                                // last read two barriers ago (group g-1, kw = 2)
        __syncthreads();
        if (++kw == 3) { kw = 0; ++g; if (++ci == kc) { ci = 0; ++kh; } }
    }
    cv_epilogue_f32<TM, TN>(a, acc, lds, m0, n0);
}

// fp32 split-K finish: y = (ws or y) + bias + residual, plus the GroupNorm sums of the result; ws (the shared scratch the partial sums went to, r03: saves
// the zero fill of y in front of every split layer -- 46 fill kernels per forward) goes back to zero.  Same thread layout as k_conv_splitk_finish.
__global__ __launch_bounds__(256) void k_conv_f32_finish(float* __restrict__ y, float* __restrict__ ws, const float* __restrict__ bias, const float* __restrict__ res, uint32_t HW,
                                                         uint32_t cpr, uint32_t rows_per_block, double* __restrict__ gn_sums, uint32_t G) {
    __shared__ float red[512];
    const uint32_t tid = threadIdx.x, cc = tid % cpr, rstep = 256 / cpr, b = blockIdx.y;
    const uint32_t Cout = cpr * 8, co = cc * 8;
    if (gn_sums) { for (uint32_t i = tid; i < cpr * 4; i += 256) red[i] = 0.f; __syncthreads(); }
    float bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[k] = bias ? bias[co + k] : 0.f;
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
    const uint32_t row_end = (tid / cpr < rstep) ? min((blockIdx.x + 1) * rows_per_block, HW) : 0u;
    for (uint32_t row = blockIdx.x * rows_per_block + tid / cpr; row < row_end; row += rstep) {
        const size_t o = (((size_t)b * HW + row) * cpr + cc) * 8;
        float* src = ws ? ws : y;
        float4 v0 = *reinterpret_cast<const float4*>(src + o), v1 = *reinterpret_cast<const float4*>(src + o + 4);
        if (ws) { *reinterpret_cast<float4*>(ws + o) = make_float4(0.f, 0.f, 0.f, 0.f); *reinterpret_cast<float4*>(ws + o + 4) = make_float4(0.f, 0.f, 0.f, 0.f); }
        float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] += bv[k];
        if (res) {
            const float4 r0 = *reinterpret_cast<const float4*>(res + o), r1 = *reinterpret_cast<const float4*>(res + o + 4);
            f[0] += r0.x; f[1] += r0.y; f[2] += r0.z; f[3] += r0.w; f[4] += r1.x; f[5] += r1.y; f[6] += r1.z; f[7] += r1.w;
        }
        *reinterpret_cast<float4*>(y + o) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(y + o + 4) = make_float4(f[4], f[5], f[6], f[7]);
        if (gn_sums) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { gs[k >> 2] += f[k]; gq[k >> 2] = __builtin_fmaf(f[k], f[k], gq[k >> 2]); }
        }
    }
    if (gn_sums) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { atomicAdd(&red[(cc * 2 + h) * 2], gs[h]); atomicAdd(&red[(cc * 2 + h) * 2 + 1], gq[h]); }
        __syncthreads();
        const uint32_t hpg = (Cout / G) / 4;
        for (uint32_t g = tid; g < G; g += 256) {
            float ss = 0.f, qq = 0.f;
            for (uint32_t i = g * hpg; i < (g + 1) * hpg; ++i) { ss += red[i * 2]; qq += red[i * 2 + 1]; }
            double* dst = gn_sums + ((size_t)b * G + g) * 2;
            atomicAdd(dst, (double)ss);
            atomicAdd(dst + 1, (double)qq);
        }
    }
}

// y = bf16(ws + bias + residual), ws goes back to zero for the next split-K convolution, and (optionally) the GroupNorm sums of y are
// accumulated for the norm that follows.  grid (row slabs, B); 256 threads = (256 / cpr) rows x cpr 8-channel chunks.
__global__ __launch_bounds__(256) void k_conv_splitk_finish(float* __restrict__ ws, const float* __restrict__ bias, const unsigned char* __restrict__ res,
                                                            unsigned char* __restrict__ y, uint32_t HW, uint32_t cpr, uint32_t rows_per_block,
                                                            double* __restrict__ gn_sums, uint32_t G) {
    __shared__ float red[512];                                               // [cpr][2 halves][sum, sumsq], cpr <= 64... sized for Cout <= 1024
    const uint32_t tid = threadIdx.x, cc = tid % cpr, rstep = 256 / cpr, b = blockIdx.y;      // threads past rstep * cpr idle (cpr not a divisor of 256)
    const uint32_t Cout = cpr * 8, co = cc * 8;
    if (gn_sums) { for (uint32_t i = tid; i < cpr * 4; i += 256) red[i] = 0.f; __syncthreads(); }
    float bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[k] = bias ? bias[co + k] : 0.f;
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
    const uint32_t row_end = (tid / cpr < rstep) ? min((blockIdx.x + 1) * rows_per_block, HW) : 0u;
    for (uint32_t row = blockIdx.x * rows_per_block + tid / cpr; row < row_end; row += rstep) {
        const size_t q = ((size_t)b * HW + row) * cpr + cc;
        float4* p = reinterpret_cast<float4*>(ws + q * 8);
        const float4 v0 = p[0], v1 = p[1];
        p[0] = make_float4(0.f, 0.f, 0.f, 0.f); p[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] += bv[k];
        if (res) {
            const uint4 rv = *reinterpret_cast<const uint4*>(res + q * 16);
            const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { f[2 * k] += __uint_as_float(rw[k] << 16); f[2 * k + 1] += __uint_as_float(rw[k] & 0xffff0000u); }
        }
        uint32_t pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = cv_pack_bf16(f[2 * k], f[2 * k + 1]);
        *reinterpret_cast<uint4*>(y + q * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        if (gn_sums) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = __uint_as_float(pk[k] << 16), hi = __uint_as_float(pk[k] & 0xffff0000u);
                gs[k >> 1] += lo + hi;
                gq[k >> 1] = __builtin_fmaf(lo, lo, gq[k >> 1]);
                gq[k >> 1] = __builtin_fmaf(hi, hi, gq[k >> 1]);
            }
        }
    }
    if (gn_sums) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { atomicAdd(&red[(cc * 2 + h) * 2], gs[h]); atomicAdd(&red[(cc * 2 + h) * 2 + 1], gq[h]); }
        __syncthreads();
        const uint32_t hpg = (Cout / G) / 4;                                 // 4-channel half chunks per group: one pair of atomics per group and block
        for (uint32_t g = tid; g < G; g += 256) {
            float ss = 0.f, qq = 0.f;
            for (uint32_t i = g * hpg; i < (g + 1) * hpg; ++i) { ss += red[i * 2]; qq += red[i * 2 + 1]; }
            double* dst = gn_sums + ((size_t)b * G + g) * 2;
            atomicAdd(dst, (double)ss);
            atomicAdd(dst + 1, (double)qq);
        }
    }
}

// ================================================================================================================================
// Round 3: the large layers (M >= 16 384 output pixels, 63 % of the UNet's convolution time) on a TWO-GROUP ("ping-pong") schedule.
// What bounded the 128 x 128 kernels above (r01 / r02 profiles, and the programming guide's "128^2 tile + two barriers per K-step ~900 TFLOP/s
// ceiling"): every wave runs load -> wait -> barrier -> 16 MFMA in lockstep, so the matrix pipe idles while the waves issue their LDS-DMA
// instructions and wait for them.  Here a block has 8 waves in two groups of four (one wave of each group per SIMD) that run the SAME phase
// sequence ONE BARRIER APART: while group 0 multiplies, group 1 reads its fragments and issues DMA, and vice versa, so each SIMD's matrix
// pipe always has one wave in an MFMA segment; `s_setprio` lets that wave win the issue port.
//   * tile 256 x 128, 8 waves as 4 (M) x 2 (N), 64 x 64 per wave; K-tile 64 in TWO phases of two 16-deep k-steps (8 MFMA 32x32x16 = 256
//     matrix-pipe cycles per phase and wave);
//   * per phase and wave: 8 ds_read_b128 (fragments of the phase), its share of the DMA of the K-tile TWO ahead, `s_waitcnt vmcnt(N)
//     lgkmcnt(0)` + s_barrier, then the MFMAs, s_barrier;
//   * three LDS stages (generic form: 3 x 48 KiB): a stage is re-filled in the K-tile after its last read; the reads are retired (lgkmcnt(0))
//     BEFORE the barrier the other group's first DMA into that stage waits behind, and a stage is read only after every wave has waited for
//     its own DMA pieces of it (counted vmcnt: only the pieces of the youngest K-tile stay in flight) and passed a barrier (the argument is
//     written out at pp_phase below);
//   * ROWS form (3 x 3, stride 1, tile = whole image rows): the A tile is loaded once per (kh, channel tile) and the three kw taps read it
//     shifted by a row (zero rows between image rows, as k_conv3x3_bf16_rows): 3.3 instead of 6 DMA instructions per wave and K-tile;
//   * epilogue straight from the accumulators (cv_epilogue_direct: transposed product, permlane32_swap, 16-byte stores).
// One block per CU (144.5 / 114.8 KiB of LDS).
SSD_DEV void pp_wait_vm_lgkm_barrier(uint32_t n) {          // n = DMA instructions of this wave that may stay in flight (younger than what must have landed)
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
}
SSD_DEV void pp_wait_vm_barrier(uint32_t n) {               // the same without the LDS wait (F32 form: that wait sits in front of the operand split)
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); break;
    }
}
SSD_DEV void pp_wait_lgkm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
SSD_DEV void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

// F32 (r03): the same kernel for fp32 activations with fp32-class products (the "f32x2" arithmetic of k_conv_igemm_f32x2: hi * hi + hi * lo + lo * hi
// on bf16 pairs).  The tile rows stay 128 bytes, so ring, DMA schedule and barriers are unchanged: an A row is 32 fp32 channels of a pixel (split
// into the bf16 pair when the fragment is read: 24 VALU instructions per fragment, under the other group's MFMA segment), a B row the 32 hi terms
// followed by the 32 lo terms of an output channel (w_lo must lie directly behind w_hi in memory: one buffer descriptor serves both); a K-tile is
// 32 channels, each of its two phases one 16-deep k-step of 12 MFMAs -- 1.5 x the matrix work of the bf16 form per byte moved.
// PS (r04, F32 only): the activations arrive PRE-SPLIT -- per pixel and block of 32 channels 128 bytes = [32 hi terms | 32 lo terms] in bf16, written
// by the GroupNorm pass that produced them (groupnorm.hip, GnVec<GN_F32>::store_split; same bytes per element as fp32, so ring, DMA schedule and
// addressing are those of the F32 form).  An A row then has exactly the layout of a B row, the fragments are read as bf16x8 like the weights', and the
// split leaves the K loop: in the F32 form every wave split every pixel fragment it read -- 48 VALU instructions per phase in the load segment, the
// same pixels again in the wave of the other channel half and (ROWS) for each of the three kw taps, six times in all.  Same products, same order:
// bit-identical results.
template <bool ROWS, bool F32 = false, bool PS = false>
__global__ __launch_bounds__(512) void k_conv_pp_bf16(const ConvArgs a) {
    static_assert(!PS || F32, "pre-split activations are a form of the fp32-class kernel");
    constexpr int BM = 256, BN = 128, TM = 2, TN = 2;
    constexpr uint32_t KCH = F32 ? 32 : 64, AB = F32 ? 4 : 2;                 // channels per K-tile, bytes per activation element
    constexpr int A_ROWS = ROWS ? BM + 8 + 1 : BM;                            // ROWS: up to eight image rows per tile (W = 32) + their zero rows
    constexpr int A_BUF = A_ROWS * CV_ROWB, B_BUF = BN * CV_ROWB;
    constexpr int NA = ROWS ? 2 : 3, NB = 3;                                 // stages per operand
    constexpr int EPI = 128 * BN * 4;                                        // cv_epilogue_bf16 goes through LDS in passes of 128 rows
    constexpr int RING = NA * A_BUF + NB * B_BUF;
    constexpr int LDS_BYTES = (RING > EPI ? RING : EPI) + 512;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    unsigned char* const abuf = lds;
    unsigned char* const bbuf = lds + NA * A_BUF;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;     // (scalar: LDS-DMA destinations are wave-uniform)
    const uint32_t group = wave >> 2;                                        // waves 0-3 / 4-7: one wave of each group per SIMD

    const uint32_t n_blocks = a.m_tiles * a.n_tiles;
    const uint32_t taps = a.ksize * a.ksize, kc = a.Cin / KCH, KT = taps * kc, Cin2 = a.Cin - a.Cin1;
    const uint32_t W = a.W;
    const uint32_t seg_w = ROWS ? (W >= (uint32_t)BM ? (uint32_t)BM : W) : 1u, n_seg = ROWS ? (uint32_t)BM / seg_w : 0u;
    if (ROWS) {
        for (uint32_t i = tid; i < NA * (n_seg + 1) * (CV_ROWB / 16); i += 512) {      // zero rows of the A stages (never overwritten)
            const uint32_t chunk = i % (CV_ROWB / 16), row = (i / (CV_ROWB / 16)) % (n_seg + 1), buf = i / ((CV_ROWB / 16) * (n_seg + 1));
            *reinterpret_cast<uint4*>(abuf + buf * A_BUF + row * (seg_w + 1) * CV_ROWB + chunk * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    const __amdgpu_buffer_rsrc_t rs_x = cv_rsrc(a.x, (uint64_t)a.B * a.H * a.W * a.Cin1 * AB);
    const __amdgpu_buffer_rsrc_t rs_x2 = cv_rsrc(a.x2 ? a.x2 : a.x, a.x2 ? (uint64_t)a.B * a.H * a.W * Cin2 * AB : 0);
    const uint32_t w_term_bytes = a.Cout * taps * a.Cin * 2;                  // one bf16 weight tensor (F32: hi, then lo directly behind it)
    const __amdgpu_buffer_rsrc_t rs_w = cv_rsrc(a.w, (uint64_t)w_term_bytes * (F32 ? 2 : 1));

    // ---- PERSISTENT over tiles: block b takes tiles b, b + gridDim, ... (gridDim = min(tiles, CUs), a multiple of 8).  The stores of a tile's
    // epilogue are posted writes: they drain while the next tile's K loop runs, instead of ending every round of blocks with a chip-wide burst of
    // 64 KiB per CU that nothing overlaps (measured: 12 us per round on the 128 x 128 layers).
    const uint32_t Hv = a.upsample ? a.H * 2 : a.H, Wv = a.upsample ? a.W * 2 : a.W;
  for (uint32_t vb = blockIdx.x; vb < n_blocks; vb += gridDim.x) {
    uint32_t tile;
    {
        const uint32_t xcd = vb & 7, idx = vb >> 3, q = n_blocks >> 3, r = n_blocks & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const uint32_t m0 = (tile / a.n_tiles) * BM, n0 = (tile % a.n_tiles) * BN;
#ifdef CV_PP_TIMING                                                              // (instrumented build: shader-clock stamps per tile into the split-K workspace pointer)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.splitk_ws) + (size_t)vb * 4;
    if (tid == 0) stamps[0] = __builtin_amdgcn_s_memtime();
#endif
    // ---- loader geometry: this wave's DMA pieces.  A piece i (0..3) = tile rows (wave * 4 + i) * 8 + (lane >> 3); B piece i (0..1) = rows (wave * 2 + i) * 8 + (lane >> 3)
    int32_t a_y0[4], a_x0[4];
    uint32_t a_img[4], a_chunk[4], a_lds[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t r8 = (wave * 4 + i) * 8, r = r8 + (lane >> 3), m = m0 + r;
        a_ok[i] = m < a.M;
        const uint32_t mm = a_ok[i] ? m : 0;
        if (ROWS) {
            const uint32_t b = mm / (a.H * W), rem = mm % (a.H * W);
            a_y0[i] = (int32_t)(rem / W) - 1;
            a_x0[i] = 0;
            a_img[i] = b * a.H * W + (rem % W);                                  // + y * W
            const uint32_t rho = r + r / seg_w + 1;
            a_chunk[i] = ((lane & 7) ^ ((rho >> 1) & 7)) * 16;                   // source-side swizzle keyed by the LDS row
            a_lds[i] = (r8 + r8 / seg_w + 1) * CV_ROWB;                          // (scalar)
        } else {
            const uint32_t b = mm / (a.Ho * a.Wo), rem = mm % (a.Ho * a.Wo);
            a_y0[i] = (int32_t)((rem / a.Wo) * a.stride) - (int32_t)a.pad;
            a_x0[i] = (int32_t)((rem % a.Wo) * a.stride) - (int32_t)a.pad;
            a_img[i] = b * a.H * a.W;
            a_chunk[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
            a_lds[i] = r8 * CV_ROWB;
        }
    }
    uint32_t b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t r = (wave * 2 + i) * 8 + (lane >> 3), sc = (lane & 7) ^ ((r >> 1) & 7);     // sc: the 16-byte chunk of the row this lane fetches
        b_voff[i] = F32 ? ((n0 + r) * taps * a.Cin) * 2 + (sc & 3) * 16 + (sc >> 2) * w_term_bytes      // chunks 0-3: 32 hi terms, 4-7: the 32 lo terms
                        : ((n0 + r) * taps * a.Cin) * 2 + sc * 16;
    }
    uint32_t a_voff[4], a_voff2[4];                                          // per tap: byte offset of this lane's 16 bytes in x / x2 at channel 0, or CV_OOB
    auto set_tap = [&](uint32_t tap) {                                          // generic: tap = kh * ksize + kw;  ROWS: tap = kh (the kw shift happens at read time)
        const int32_t kh = ROWS ? (int32_t)tap : (int32_t)(tap / a.ksize), kw = ROWS ? 0 : (int32_t)(tap % a.ksize);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int32_t yv = a_y0[i] + kh, xv = a_x0[i] + kw;
            const bool ok = a_ok[i] && yv >= 0 && xv >= 0 && yv < (int32_t)Hv && (ROWS || xv < (int32_t)Wv);
            uint32_t pix;
            if (ROWS) pix = a_img[i] + (uint32_t)yv * W;
            else {
                const uint32_t yi = a.upsample ? (uint32_t)yv >> 1 : (uint32_t)yv, xi = a.upsample ? (uint32_t)xv >> 1 : (uint32_t)xv;
                pix = a_img[i] + yi * a.W + xi;
            }
            a_voff[i] = ok ? pix * a.Cin1 * AB + a_chunk[i] : CV_OOB;
            a_voff2[i] = ok ? pix * Cin2 * AB + a_chunk[i] : CV_OOB;
        }
    };
    auto issue_a = [&](int i, uint32_t ci0, uint32_t buf) {                      // one A piece of the tap set by set_tap; ci0 is scalar
        unsigned char* dst = abuf + buf * A_BUF + a_lds[i];
        if (ci0 >= a.Cin1) cv_dma16(rs_x2, dst, a_voff2[i], (ci0 - a.Cin1) * AB);
        else cv_dma16(rs_x, dst, a_voff[i], ci0 * AB);
    };
    auto issue_b = [&](int i, uint32_t tap, uint32_t ci0, uint32_t buf) {
        cv_dma16(rs_w, bbuf + buf * B_BUF + (wave * 2 + i) * 1024, b_voff[i], (tap * a.Cin + ci0) * 2);
    };

    // ---- reader geometry ---------------------------------------------------------------------------------------------------------------
    uint32_t a_rd[TM][ROWS ? 3 : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const uint32_t t0 = wm * 64 + i * 32;
        if (ROWS) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const uint32_t rho = t0 + (lane & 31) + t0 / seg_w + kw;          // = 1 + seg (seg_w + 1) + x + (kw - 1)
                a_rd[i][ROWS ? kw : 0] = rho * CV_ROWB + ((((lane >> 5) << (F32 && !PS ? 1 : 0)) ^ ((rho >> 1) & 7)) * 16);
            }
        } else {
            a_rd[i][0] = (t0 + (lane & 31)) * CV_ROWB + ((((lane >> 5) << (F32 && !PS ? 1 : 0)) ^ ((lane >> 1) & 7)) * 16);
        }
    }
    // (F32: a lane's 8 channels of k-step P are the TWO chunks 4 P + 2 half + {0, 1} of the fp32 row: a_rd ^ (P * 64) ^ (e * 16))
    const uint32_t b_rd = (wn * 64 + (lane & 31)) * CV_ROWB + (((lane >> 5) ^ ((lane >> 1) & 7)) * 16);

    f32x16 acc[TN][TM];                                                          // TRANSPOSED: [channel tile][pixel tile], rows of a tile = output channels (cv_epilogue_direct)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    // ---- the K loop ------------------------------------------------------------------------------------------------------------------------
    // K-tile t: generic (tap, ci) = (t / kc, t % kc); ROWS: group g = t / 3 = kh * kc + ci, kw = t % 3, B tap = kh * 3 + kw.
    // DMA schedule of one wave: phase q = 2 t + p issues B piece p of K-tile t + 2 and
    //   generic: A pieces 2 p, 2 p + 1 of K-tile t + 2;        ROWS: A piece (q % 6) of group g + 1 when q % 6 < 4.
    // pp_phase safety (P = number of DMA instructions this wave issued in phases 2 t and 2 t + 1):
    //   RAW  the wait of phase 2 t + 1, `vmcnt(P)`, retires every piece issued BEFORE phase 2 t: all of K-tile t + 1 (and, ROWS, of the A group read
    //        next).  It sits before that phase's first barrier; a wave reads K-tile t + 1 only after the phase's second barrier, which the other
    //        group reaches only after ITS wait (the groups are one barrier apart), so every wave's pieces have landed.
    //   WAR  the stage K-tile t + 2 goes to was last read in phase 2 t - 1, and those reads are retired (`lgkmcnt(0)`) before that phase's first
    //        barrier; the first DMA into it is issued in phase 2 t, i.e. after the second barrier of phase 2 t - 1, which the other group cannot
    //        pass before its own first barrier of phase 2 t - 1.
    uint32_t l_tap = 0, l_ci = 0, l_kt = 0;                                      // loader iterator (generic: the K-tile being issued; ROWS: its B side)
    uint32_t g_kh = 0, g_ci = 0, g_next = 0;                                     // ROWS: the A group being issued
    auto advance_l = [&]() { ++l_kt; if (ROWS) { if (l_kt % 3 == 0) { if (++l_ci == kc) { l_ci = 0; ++l_tap; } } } else { if (++l_ci == kc) { l_ci = 0; ++l_tap; } } };
    // prologue: K-tiles 0 and 1 (ROWS: A group 0, B of K-tiles 0 and 1)
    uint32_t in_flight_young = 0;
    if (ROWS) {
        set_tap(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_a(i, 0, 0);
        g_next = 1; g_ci = 1 % kc; g_kh = 1 / kc;                                // next group to issue
        issue_b(0, 0, 0, 0); issue_b(1, 0, 0, 0);
        if (KT > 1) { issue_b(0, 1, 0, 1); issue_b(1, 1, 0, 1); in_flight_young = 2; }
        l_kt = 2; l_tap = 0; l_ci = 0;                                           // B iterator: K-tile 2 = (kh 0, ci 0, kw 2); l_tap counts kh * kc + ci groups here
    } else {
        set_tap(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_a(i, 0, 0);
        issue_b(0, 0, 0, 0); issue_b(1, 0, 0, 0);
        advance_l();
        if (KT > 1) {
            if (l_ci == 0) set_tap(l_tap);
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_a(i, l_ci * KCH, 1);
            issue_b(0, l_tap, l_ci * KCH, 1); issue_b(1, l_tap, l_ci * KCH, 1);
            advance_l();
            in_flight_young = 6;
        }
        if (l_kt < KT && l_ci == 0) set_tap(l_tap);
    }
    pp_wait_vm_lgkm_barrier(in_flight_young);                                    // K-tile 0 (and the zero rows) in place for every wave
#ifdef CV_PP_TIMING
    if (tid == 0) stamps[1] = __builtin_amdgcn_s_memtime();
#endif

    // Fragments are read ONE PHASE AHEAD, in the compute segment (r03): a phase's load segment was twice as long as its MFMA segment -- eight
    // ds_read_b128 round trips, then the DMA issue, in one wave -- so the two groups alternated at ~50 % matrix-pipe use.  Now the reads of phase
    // q + 1 are issued at the head of phase q's compute segment and fly under its MFMAs; the load segment is DMA issue, the waits and (F32) the
    // operand split.  Two register sets, indexed by the phase's k-half.  What changes in the safety argument (pp_phase above):
    //   RAW  K-tile t + 1 is first read in the compute segment of phase 2 t + 1, so its DMA wait moves ONE phase up: phase 2 t waits
    //        `vmcnt(pieces issued in phase 2 t)` before its first barrier, which retires every older piece (all of K-tile t + 1; ROWS: the next A
    //        group too); the other group does the same one barrier later, and phase 2 t + 1's compute segment lies behind both.
    //   WAR  a stage's last reads now sit in the compute segment of the phase BEFORE the old one -- earlier, not later; they are retired by the
    //        `lgkmcnt(0)` of the following load segment, i.e. still before the barrier the first DMA into that stage waits behind.
    bf16x8 fa[2][2][TM], fb[2][2][TN];                                           // [register set = k-half P][bf16: k-step of the phase | F32: hi, lo][tile]
    float4 xa[F32 ? TM : 1][2];                                                  // F32: the NEXT phase's raw fp32 pixels (split in its load segment)
    auto load_frags = [&](auto pc, auto kwc, const unsigned char* sa, const unsigned char* sb) {
        constexpr int P = decltype(pc)::value, KW = decltype(kwc)::value;
        if (F32) {
            if (PS) {                                                        // pre-split pixels: hi / lo chunks exactly like the weights'
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[P][s][i] = *reinterpret_cast<const bf16x8*>(sa + (a_rd[i][ROWS ? KW : 0] ^ ((P + 2 * s) * 32)));
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 2; ++e) xa[F32 ? i : 0][e] = *reinterpret_cast<const float4*>(sa + (a_rd[i][ROWS ? KW : 0] ^ (P * 64) ^ (e * 16)));
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)                                      // s = 0: the hi terms (row chunks 0-3), 1: the lo terms (chunks 4-7)
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[P][s][j] = *reinterpret_cast<const bf16x8*>(sb + ((b_rd + j * 32 * CV_ROWB) ^ ((P + 2 * s) * 32)));
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[P][s][i] = *reinterpret_cast<const bf16x8*>(sa + (a_rd[i][ROWS ? KW : 0] ^ ((2 * P + s) * 32)));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[P][s][j] = *reinterpret_cast<const bf16x8*>(sb + ((b_rd + j * 32 * CV_ROWB) ^ ((2 * P + s) * 32)));
            }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    load_frags(P0{}, K0{}, abuf, bbuf);                                          // phase 0's fragments (K-tile 0 is in place: the barrier above)
    if (group == 1) pp_barrier();                                                // the stagger: group 1 runs one barrier behind

    // one phase: P = k-half of the K-tile (compile time), KW = tap column (ROWS; compile time: the K-tiles of a group are unrolled);
    // (KWN, sa_n, sb_n) = where the NEXT phase reads (has_next: there is one)
    auto phase = [&](auto pc, auto kwc, auto kwnc, const unsigned char* sa_n, const unsigned char* sb_n, bool has_next) {
        constexpr int P = decltype(pc)::value, KW = decltype(kwc)::value;
        // ---- load segment: this wave's share of the DMA two K-tiles ahead
        uint32_t issued = 0;
        if (ROWS) {
            if (l_kt < KT) {                                                     // B piece P of K-tile l_kt = t + 2: tap = kh * 3 + kw of group l_kt / 3
                const uint32_t g = l_kt / 3, bkh = g / kc, bci = g % kc;
                issue_b(P, bkh * 3 + l_kt % 3, bci * KCH, l_kt % NB);
                ++issued;
            }
            constexpr int Q6 = 2 * KW + P;                                       // phase within the group: A piece Q6 of the NEXT group in its first four phases
            if (Q6 < 4 && g_next < 3 * kc) {
                if (Q6 == 0) set_tap(g_kh);
                issue_a(Q6 < 4 ? Q6 : 0, g_ci * KCH, g_next & 1);
                ++issued;
                if (Q6 == 3) { ++g_next; if (++g_ci == kc) { g_ci = 0; ++g_kh; } }
            }
            if (P == 1) ++l_kt;
        } else {
            if (l_kt < KT) {
                issue_a(2 * P, l_ci * KCH, l_kt % NA); issue_a(2 * P + 1, l_ci * KCH, l_kt % NA);
                issue_b(P, l_tap, l_ci * KCH, l_kt % NB);
                issued = 3;
                if (P == 1) { advance_l(); if (l_kt < KT && l_ci == 0) set_tap(l_tap); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (F32 && PS) {                                                     // nothing to split: the load segment is DMA issue and the waits, as in the bf16 form
            if (P == 0) pp_wait_vm_lgkm_barrier(issued);
            else pp_wait_lgkm_barrier();
        } else if (F32) {
            // this phase's fragments (read in the previous compute segment) are in: split the pixels' fp32 values -- hi = truncation to bf16,
            // lo = truncation of the exact remainder -- here, under the other group's MFMA segment
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v[8] = {xa[F32 ? i : 0][0].x, xa[F32 ? i : 0][0].y, xa[F32 ? i : 0][0].z, xa[F32 ? i : 0][0].w,
                                    xa[F32 ? i : 0][1].x, xa[F32 ? i : 0][1].y, xa[F32 ? i : 0][1].z, xa[F32 ? i : 0][1].w};
                uint32_t hw[4], lw[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float r0 = v[2 * k] - __uint_as_float(__float_as_uint(v[2 * k]) & 0xffff0000u);
                    const float r1 = v[2 * k + 1] - __uint_as_float(__float_as_uint(v[2 * k + 1]) & 0xffff0000u);
                    hw[k] = __builtin_amdgcn_perm(__float_as_uint(v[2 * k + 1]), __float_as_uint(v[2 * k]), 0x07060302u);
                    lw[k] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                }
                const uint4 uh = make_uint4(hw[0], hw[1], hw[2], hw[3]), ul = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                fa[P][0][i] = *reinterpret_cast<const bf16x8*>(&uh);
                fa[P][1][i] = *reinterpret_cast<const bf16x8*>(&ul);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (P == 0) pp_wait_vm_barrier(issued);
            else pp_barrier();
        } else if (P == 0) pp_wait_vm_lgkm_barrier(issued);
        else pp_wait_lgkm_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- compute segment: the next phase's fragment reads first, then this phase's MFMAs over them
        __builtin_amdgcn_s_setprio(1);
        if (has_next) {
            if (P == 0) load_frags(P1{}, kwc, sa_n, sb_n);
            else load_frags(P0{}, kwnc, sa_n, sb_n);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (F32) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[P][1][j], fa[P][0][i], acc[j][i], 0, 0, 0);      // w_lo x_hi
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[P][0][j], fa[P][1][i], acc[j][i], 0, 0, 0);      // w_hi x_lo
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[P][0][j], fa[P][0][i], acc[j][i], 0, 0, 0);      // w_hi x_hi
                }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[P][s][j], fa[P][s][i], acc[j][i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if (ROWS) {
        const uint32_t G = 3 * kc;
        for (uint32_t g = 0, t = 0; g < G; ++g, t += 3) {                        // group (kh, ci): three K-tiles (kw) from one A stage
            const unsigned char* sa = abuf + (g & 1) * A_BUF;
            const unsigned char* sb0 = bbuf + (t % NB) * B_BUF;
            const unsigned char* sb1 = bbuf + ((t + 1) % NB) * B_BUF;
            const unsigned char* sb2 = bbuf + ((t + 2) % NB) * B_BUF;
            phase(P0{}, K0{}, K0{}, sa, sb0, true);
            phase(P1{}, K0{}, K1{}, sa, sb1, true);
            phase(P0{}, K1{}, K1{}, sa, sb1, true);
            phase(P1{}, K1{}, K2{}, sa, sb2, true);
            phase(P0{}, K2{}, K2{}, sa, sb2, true);
            phase(P1{}, K2{}, K0{}, abuf + ((g + 1) & 1) * A_BUF, bbuf + ((t + 3) % NB) * B_BUF, g + 1 < G);
        }
    } else {
        for (uint32_t t = 0; t < KT; ++t) {
            const unsigned char* sa = abuf + (t % NA) * A_BUF;
            const unsigned char* sb = bbuf + (t % NB) * B_BUF;
            phase(P0{}, K0{}, K0{}, sa, sb, true);
            phase(P1{}, K0{}, K0{}, abuf + ((t + 1) % NA) * A_BUF, bbuf + ((t + 1) % NB) * B_BUF, t + 1 < KT);
        }
    }
    if (group == 0) pp_barrier();                                                // even out the stagger
    __syncthreads();                                                             // every wave is done with the stages (the epilogue reuses them)
#ifdef CV_PP_TIMING
    if (tid == 0) stamps[2] = __builtin_amdgcn_s_memtime();
#endif
    cv_epilogue_direct<TM, TN, F32>(a, acc, lds + RING, m0, n0, wm, wn);
#ifdef CV_PP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // (stamp 3 = this wave's stores acknowledged)
    if (tid == 0) stamps[3] = __builtin_amdgcn_s_memtime();
#endif
    if (a.gn_sums) __syncthreads();                                              // (the next tile's epilogue zeroes the statistics scratch again)
  }
}

#ifndef CV_NS_SMALL
#define CV_NS_SMALL 4                              // DMA ring depth of the 64 x 64 tile (66 KB of LDS: two blocks per CU).  r04 A/B, UNet step fp32 / bf16: 4: 8.23 / 4.29 ms,
                                                   // 3 (three blocks per CU): 8.22 / 4.31, 2 (four): 8.40 / 4.69 -- occupancy is not what these layers lack
#endif
template <int TM, int TN, int WM, int WN, int NS>
int cv_launch(ConvArgs& a, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    a.m_tiles = (a.M + BM - 1) / BM;
    a.n_tiles = (a.Cout + BN - 1) / BN;
    const bool partial = a.Cin1 % CV_BK != 0 || (a.Cin - a.Cin1) % CV_BK != 0 || a.Cout % BN != 0;
    if (partial) hipLaunchKernelGGL((k_conv_igemm_bf16<TM, TN, WM, WN, NS, true>), dim3(a.m_tiles * a.n_tiles * a.splits), dim3(64 * WM * WN), 0, st, a);
    else hipLaunchKernelGGL((k_conv_igemm_bf16<TM, TN, WM, WN, NS, false>), dim3(a.m_tiles * a.n_tiles * a.splits), dim3(64 * WM * WN), 0, st, a);
    return 0;
}

}  // namespace

// tile choice: the largest tile that still gives every CU (256) a block; tile_hint 1/2/3/4 forces 128x128 / 64x128 / 64x64 / 256x128
// (the bigger the tile the fewer L2->LDS bytes per FLOP, which is what bounds the large layers).
// split-K: a layer with too few output tiles to occupy the chip (each block then pays a full memory round trip per K-tile with
// nothing to hide it) is cut along K into `splits` blocks per tile that accumulate into the caller's zeroed fp32 workspace.
static void cv_plan(uint32_t M, uint32_t Cin, uint32_t Cout, uint32_t ksize, int tile_hint, bool may_split, int splits_hint, int* choice_out, uint32_t* splits_out) {
    int choice = tile_hint;
    if (choice < 1 || choice > 4) {
        // r03 sweep of every low-resolution layer over (tile, splits) (profiles/r03/e_small_layer_sweep.jsonl): 128 x 128 tiles once they fill the chip
        // 1.5 times over, 64 x 128 only for very wide outputs, else 64 x 64 -- more, shorter blocks beat fewer, longer ones below 32 x 32.
        // (256 x 128 on 8 lock-step waves, hint 4, measures no faster than 128 x 128 at 2 blocks / CU; the two-group 256 x 128 kernel is hints 5 / 6.)
        const uint64_t t128 = (uint64_t)((M + 127) / 128) * (Cout / 128);
        const uint64_t t64 = (uint64_t)((M + 63) / 64) * (Cout / 128);
        choice = (Cout % 128 != 0) ? 3 : (t128 >= 384) ? 1 : (t64 >= 512) ? 2 : 3;
    }
    const uint32_t bm = choice == 4 ? 256 : choice == 1 ? 128 : 64, bn = choice == 3 ? 64 : 128;
    const uint32_t tiles = ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
    const uint32_t KT = ksize * ksize * ((Cin + 63) / 64);
    uint32_t splits = 1;
    if (may_split) {
        if (splits_hint > 0) splits = (uint32_t)splits_hint;
        else if (tiles < 384 && (uint64_t)M * Cout <= (1u << 20)) {
            // enough blocks for ~1.5 per CU, at least 12 K-tiles each; a 2-way split of a short K (< 64 K-tiles) costs more in fp32 atomics and
            // the finishing pass than the half-empty chip it avoids (same sweep: 1 x 1 layers and 256 -> 512 @ 16 x 16 are fastest unsplit)
            splits = (384 + tiles / 2) / tiles;
            if (splits > KT / 12) splits = KT / 12;
            if (splits == 2 && KT < 64) splits = 1;
            if (splits < 1) splits = 1;
        }
        if (splits > 16) splits = 16;
        if (splits > KT / 2) splits = KT / 2 ? KT / 2 : 1;                   // at least two K-tiles per block
    }
    *choice_out = choice; *splits_out = splits;
}

// Returns tile choice (1..3) | splits << 8 for a layer with M = B*Ho*Wo output pixels: lets the host know whether the epilogue
// can carry GroupNorm statistics (only unsplit layers) before it decides what to ask for.
extern "C" int ssdnerf_conv2d_nhwc_bf16_plan(uint32_t M, uint32_t Cin, uint32_t Cout, uint32_t ksize, int tile_hint, int may_split, int splits_hint) {
    int choice; uint32_t splits;
    cv_plan(M, Cin, Cout, ksize, tile_hint, may_split != 0, splits_hint, &choice, &splits);
    return choice | (int)(splits << 8);
}

// tile (1 = 128x128, 3 = 64x64) | splits << 8 of the fp32 kernel for M output pixels
extern "C" int ssdnerf_conv2d_nhwc_f32x2_plan(uint32_t M, uint32_t Cin, uint32_t Cout, uint32_t ksize, int tile_hint, int splits_hint) {
    int choice = tile_hint;
    if (choice != 1 && choice != 3) choice = (Cout % 128 == 0 && (uint64_t)((M + 127) / 128) * (Cout / 128) >= 384) ? 1 : 3;
    const uint32_t bm = choice == 1 ? 128 : 64, tiles = ((M + bm - 1) / bm) * ((Cout + bm - 1) / bm), KT = ksize * ksize * ((Cin + 31) / 32);
    uint32_t splits = 1;
    if (splits_hint > 0) splits = (uint32_t)splits_hint;
    else if (tiles < 512 && (uint64_t)M * Cout <= (1u << 21) && Cout <= 1024) splits = (1024 + tiles - 1) / tiles;
    if (splits > 16) splits = 16;
    if (splits > KT / 2) splits = KT / 2 ? KT / 2 : 1;
    return choice | (int)(splits << 8);
}

extern "C" int ssdnerf_conv2d_nhwc_f32x2(const void* x, const void* x2, uint32_t Cin1, const void* w_hi, const void* w_lo, const float* bias, const void* residual,
                                         void* y, uint32_t B, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t ksize, uint32_t stride, uint32_t upsample,
                                         void* gn_sums, uint32_t gn_groups, int tile_hint, int splits_hint, int y_is_zero, void* splitk_ws, size_t splitk_ws_bytes,
                                         void* stream) {
    if (B == 0 || H == 0 || W == 0) return SSDNERF_OK;
    SSD_REQUIRE(x && w_hi && w_lo && y, "conv2d_nhwc_f32x2: null pointer");
    SSD_REQUIRE(ssdnerf_conv2d_nhwc_bf16_supported(Cin, Cout, ksize, stride, upsample),
                "conv2d_nhwc_f32x2: needs Cin %% 8 == 0, Cout %% 8 == 0, ksize 1|3, stride 1|2 (no stride with upsample)");
    if (!x2) Cin1 = Cin;
    SSD_REQUIRE(Cin1 <= Cin && Cin1 % 8 == 0 && (x2 || Cin1 == Cin), "conv2d_nhwc_f32x2: the first input's channel count must be a multiple of 8 and <= Cin");
    SSD_REQUIRE((uint64_t)Cout * ksize * ksize * Cin * 2 < (1ull << 31), "conv2d_nhwc_f32x2: weight tensor too large");
    SSD_REQUIRE(!gn_sums || (gn_groups > 0 && Cout % gn_groups == 0 && (Cout / gn_groups) % 4 == 0), "conv2d_nhwc_f32x2: fused GroupNorm statistics need groups of a multiple of 4 channels");
    ConvArgs a;
    a.tickets = nullptr;
    a.x2 = (const unsigned char*)x2; a.Cin1 = Cin1;
    a.x = (const unsigned char*)x; a.w = (const unsigned char*)w_hi; a.bias = bias; a.res = (const unsigned char*)residual; a.y = (unsigned char*)y;
    a.gn_sums = (double*)gn_sums; a.G = gn_groups ? gn_groups : 1;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.ksize = ksize; a.stride = stride; a.pad = ksize / 2; a.upsample = upsample;
    const uint32_t Hv = upsample ? 2 * H : H, Wv = upsample ? 2 * W : W;
    a.Ho = (Hv + 2 * a.pad - ksize) / stride + 1;
    a.Wo = (Wv + 2 * a.pad - ksize) / stride + 1;
    SSD_REQUIRE((uint64_t)B * a.Ho * a.Wo < (1ull << 31), "conv2d_nhwc_f32x2: tensor too large");
    a.M = B * a.Ho * a.Wo;
    a.splitk_ws = (splitk_ws && splitk_ws_bytes >= (size_t)a.M * Cout * 4) ? (float*)splitk_ws : nullptr;
    const int plan = ssdnerf_conv2d_nhwc_f32x2_plan(a.M, Cin, Cout, ksize, tile_hint, splits_hint);
    const int choice = plan & 0xff;
    const uint32_t splits = (uint32_t)plan >> 8, bm = choice == 1 ? 128 : 64;
    if (choice == 1) SSD_REQUIRE(Cout % 128 == 0, "conv2d_nhwc_f32x2: 128-wide tiles need Cout %% 128 == 0");
    a.splits = splits;
    SSD_REQUIRE(!gn_sums || splits > 1 || (a.Ho * a.Wo) % bm == 0, "conv2d_nhwc_f32x2: fused GroupNorm statistics need Ho*Wo to be a multiple of the M tile");
    hipStream_t st = (hipStream_t)stream;
    double* stats = a.gn_sums;
    a.tickets = nullptr;
    a.m_tiles = (a.M + bm - 1) / bm; a.n_tiles = (Cout + bm - 1) / bm;
    if (splits > 1) {
        a.tickets = cv_fold_tickets(a, splitk_ws, splitk_ws_bytes, bm, stats != nullptr);
        if (!a.tickets) a.gn_sums = nullptr;                                 // (folded: the last block's epilogue takes the statistics)
        if (!a.splitk_ws && !y_is_zero && hipMemsetAsync(y, 0, (size_t)a.M * Cout * 4, st) != hipSuccess) return ssdnerf_fail(SSDNERF_E_LAUNCH, "conv2d_nhwc_f32x2: memset failed");
    } else {
        a.splitk_ws = nullptr;
    }
    // r03: the two-group 256 x 128 kernel (k_conv_pp_bf16<ROWS, F32 = true>) takes the layers its bf16 form takes (tile_hint 5 / 6 force it);
    // it needs the lo weights directly behind the hi ones (one buffer descriptor serves both terms)
    static const bool pp_auto = getenv("SSDNERF_CONV_NO_TWO_GROUP") == nullptr;
    const bool pp_ok = Cin % 32 == 0 && Cin1 % 32 == 0 && Cout % 128 == 0 && (const unsigned char*)w_lo == (const unsigned char*)w_hi + (size_t)Cout * ksize * ksize * Cin * 2
                       && (uint64_t)B * H * W * Cin * 4 < (1ull << 31) && (!gn_sums || (a.Ho * a.Wo) % 256 == 0);
    // (measured, profiles/r03/m_two_group_fp32.txt: the row-reuse form equals k_conv3x3_f32x2_rows on the 128 x 128 level -- 164 vs 166 us per layer --;
    // the generic form is SLOWER than k_conv_igemm_f32x2<2, 2> on the upsampling layer, 510 vs 447 us, so only the former is picked automatically)
    if (tile_hint == 0 && pp_auto && pp_ok && splits == 1 && ksize == 3 && stride == 1 && a.M >= 32768 && !upsample && (W == 128 || W == 64) && (H * W) % 256 == 0)
        tile_hint = 6;
    if (tile_hint == 5 || tile_hint == 6) {
        SSD_REQUIRE(pp_ok, "conv2d_nhwc_f32x2: the 256 x 128 kernel needs Cin %% 32 == 0, Cout %% 128 == 0 and w_lo directly behind w_hi");
        a.splits = 1; a.splitk_ws = nullptr; a.gn_sums = stats;
        a.m_tiles = (a.M + 255) / 256; a.n_tiles = Cout / 128;
        const bool pp_rows = tile_hint == 6 && ksize == 3 && stride == 1 && !upsample && (W == 128 || W == 64 || W == 32) && (H * W) % 256 == 0;
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
            if (n_cu < 8) n_cu = 256;
            n_cu &= ~7;
        }
        const uint32_t tiles = a.m_tiles * a.n_tiles, grid = tiles < (uint32_t)n_cu ? tiles : (uint32_t)n_cu;
        if (pp_rows) hipLaunchKernelGGL((k_conv_pp_bf16<true, true>), dim3(grid), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_conv_pp_bf16<false, true>), dim3(grid), dim3(512), 0, st, a);
        SSD_CHECK_LAUNCH("conv2d_nhwc_f32x2");
        return SSDNERF_OK;
    }
    a.m_tiles = (a.M + bm - 1) / bm; a.n_tiles = (Cout + bm - 1) / bm;
    // 3x3 / stride 1 layers whose 128-pixel tiles are whole image rows take the row-reuse kernel (A loaded once per kh, shared by the three kw taps)
    static const bool rows_ok = getenv("SSDNERF_CONV_NO_ROW_REUSE") == nullptr;
    const bool rows = rows_ok && Cin % 64 == 0 && Cin1 % 64 == 0 && choice == 1 && splits == 1 && ksize == 3 && stride == 1 && !upsample && (W == 128 || W == 64 || W == 32) && (H * W) % 128 == 0;
    if (rows) hipLaunchKernelGGL(k_conv3x3_f32x2_rows, dim3(a.m_tiles * a.n_tiles), dim3(256), 0, st, a, (const unsigned char*)w_lo);
    else {
        const bool partial = Cin1 % 32 != 0 || (Cin - Cin1) % 32 != 0 || Cout % bm != 0;
        const dim3 grid(a.m_tiles * a.n_tiles * splits);
        if (choice == 1 && partial) hipLaunchKernelGGL((k_conv_igemm_f32x2<2, 2, true>), grid, dim3(256), 0, st, a, (const unsigned char*)w_lo);
        else if (choice == 1) hipLaunchKernelGGL((k_conv_igemm_f32x2<2, 2, false>), grid, dim3(256), 0, st, a, (const unsigned char*)w_lo);
        else if (partial) hipLaunchKernelGGL((k_conv_igemm_f32x2<1, 1, true>), grid, dim3(256), 0, st, a, (const unsigned char*)w_lo);
        else hipLaunchKernelGGL((k_conv_igemm_f32x2<1, 1, false>), grid, dim3(256), 0, st, a, (const unsigned char*)w_lo);
    }
    if (splits > 1 && !a.tickets) {
        const uint32_t HWo = a.Ho * a.Wo, cpr = Cout / 8, rstep = 256 / cpr ? 256 / cpr : 1;
        uint32_t rows = HWo;
        while (rows > rstep && rows % 2 == 0 && (uint64_t)B * (HWo / rows) < 1024) rows /= 2;
        hipLaunchKernelGGL(k_conv_f32_finish, dim3((HWo + rows - 1) / rows, B), dim3(256), 0, st, (float*)y, a.splitk_ws, bias, (const float*)residual, HWo, cpr, rows, stats, a.G);
    }
    SSD_CHECK_LAUNCH("conv2d_nhwc_f32x2");
    return SSDNERF_OK;
}

// tile (1 = 128 x 128, 2 = 64 x 128, 3 = 64 x 64) | splits << 8 of the generic kernel's PS form.  Its own rule, from the r04 sweep of the low-resolution
// layers over (tile, splits) (profiles/r04/k_ps_small_layer_sweep.jsonl): without the register pass of k_conv_igemm_f32x2 a block is short enough that
// ~512 blocks (two per CU) are the target, not 1024, and a K of fewer than 64 tiles (every 1 x 1 layer) is fastest unsplit -- the fp32 atomics and the
// finishing pass cost more than the half-empty chip.
static int cv_ps_plan(uint32_t M, uint32_t Cin, uint32_t Cout, uint32_t ksize, int tile_hint, int splits_hint) {
    int choice = tile_hint;
    if (choice < 1 || choice > 3) choice = (Cout % 128 == 0 && (uint64_t)((M + 127) / 128) * (Cout / 128) >= 384) ? 1 : 3;
    const uint32_t bm = choice == 1 ? 128 : 64, bn = choice == 3 ? 64 : 128, tiles = ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn), KT = ksize * ksize * (Cin / 32);
    uint32_t splits = 1;
    if (splits_hint > 0) splits = (uint32_t)splits_hint;
    else if (tiles < 384 && KT >= 64 && (uint64_t)M * Cout <= (1u << 21) && Cout <= 1024) {
        splits = (512 + tiles / 2) / tiles;
        if (splits > 8) splits = 8;
        while (splits > 1 && KT / splits < 16) --splits;
    }
    if (splits > 16) splits = 16;
    if (splits > KT / 2) splits = KT / 2 ? KT / 2 : 1;
    return choice | (int)(splits << 8);
}

// Convolutions on PRE-SPLIT activations: x = what ssdnerf_group_norm_nhwc(..., act | 2) wrote.  Two kernels take them:
//   1  the two-group row kernel (k_conv_pp_bf16<true, true, true>) on the layers its fp32 form takes on its own (3 x 3, W in {64, 128}, >= 32768 output pixels,
//      Cin % 32 == 0, Cout % 128 == 0) -- bit-identical to the on-the-fly split;
//   2  (r04) the generic DMA-ring kernel's PS form (k_conv_igemm_bf16<..., PS = true>) on every other stride-1 layer with Cin % 32 == 0, Cout % 64 == 0
//      (tile and split-K from ssdnerf_conv2d_nhwc_f32x2_plan): same fp32-class products as k_conv_igemm_f32x2, summed in another order.
// _supported returns which (0 = neither), so that the host asks its GroupNorm for the split output exactly then.
extern "C" int ssdnerf_conv2d_nhwc_f32x2_presplit_supported(uint32_t B, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t ksize, int with_gn_sums) {
    static const bool pp_auto = getenv("SSDNERF_CONV_NO_TWO_GROUP") == nullptr && getenv("SSDNERF_CONV_NO_PRESPLIT") == nullptr;
    static const bool small_auto = getenv("SSDNERF_CONV_NO_PRESPLIT") == nullptr && getenv("SSDNERF_CONV_NO_PRESPLIT_SMALL") == nullptr;
    const uint64_t M = (uint64_t)B * H * W;
    if (B == 0 || H == 0 || W == 0 || (ksize != 1 && ksize != 3) || Cin % 32 != 0 || M >= (1ull << 31) || M * Cin * 4 >= (1ull << 31)
        || (uint64_t)Cout * ksize * ksize * Cin * 4 >= (1ull << 31)) return 0;
    if (pp_auto && ksize == 3 && Cout % 128 == 0 && (W == 128 || W == 64) && (H * W) % 256 == 0 && M >= 32768) return 1;
    if (!small_auto || Cout % 64 != 0 || M * Cout * 4 >= (1ull << 31)) return 0;
    const int plan = cv_ps_plan((uint32_t)M, Cin, Cout, ksize, 0, 0);
    const uint32_t bm = (plan & 0xff) == 1 ? 128 : 64, splits = (uint32_t)plan >> 8;
    if (with_gn_sums && splits == 1 && (H * W) % bm != 0) return 0;
    return 2;
}

extern "C" int ssdnerf_conv2d_nhwc_f32x2_presplit(const void* x_split, const void* w_hi, const void* w_lo, const float* bias, const void* residual, void* y, uint32_t B,
                                                  uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t ksize, void* gn_sums, uint32_t gn_groups,
                                                  int tile_hint, int splits_hint, void* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if (B == 0 || H == 0 || W == 0) return SSDNERF_OK;
    SSD_REQUIRE(x_split && w_hi && w_lo && y, "conv2d_nhwc_f32x2_presplit: null pointer");
    int kind = ssdnerf_conv2d_nhwc_f32x2_presplit_supported(B, H, W, Cin, Cout, ksize, gn_sums != nullptr);
    if (tile_hint >= 1 && tile_hint <= 3 && ssdnerf_conv2d_nhwc_f32x2_presplit_supported(B, H, W, Cin, tile_hint == 3 ? 64 : 128, ksize, 0) && Cout % (tile_hint == 3 ? 64 : 128) == 0)
        kind = 2;                                                            // (sweeps: force the generic kernel's PS form with this tile)
    SSD_REQUIRE(kind != 0, "conv2d_nhwc_f32x2_presplit: layer not taken by a pre-split kernel (ask ssdnerf_conv2d_nhwc_f32x2_presplit_supported first)");
    SSD_REQUIRE((const unsigned char*)w_lo == (const unsigned char*)w_hi + (size_t)Cout * ksize * ksize * Cin * 2, "conv2d_nhwc_f32x2_presplit: w_lo must lie directly behind w_hi");
    SSD_REQUIRE(!gn_sums || (gn_groups > 0 && Cout % gn_groups == 0 && (Cout / gn_groups) % 4 == 0), "conv2d_nhwc_f32x2_presplit: fused GroupNorm statistics need groups of a multiple of 4 channels");
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a;
    a.tickets = nullptr;
    a.x2 = nullptr; a.Cin1 = Cin;
    a.x = (const unsigned char*)x_split; a.w = (const unsigned char*)w_hi; a.bias = bias; a.res = (const unsigned char*)residual; a.y = (unsigned char*)y;
    a.gn_sums = (double*)gn_sums; a.G = gn_groups ? gn_groups : 1;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.ksize = ksize; a.stride = 1; a.pad = ksize / 2; a.upsample = 0;
    a.Ho = H; a.Wo = W; a.M = B * H * W;
    a.splits = 1; a.splitk_ws = nullptr; a.tickets = nullptr;
    if (kind == 2) {
        const int plan = cv_ps_plan(a.M, Cin, Cout, ksize, tile_hint, splits_hint);
        const int choice = plan & 0xff;
        const uint32_t bm = choice == 1 ? 128 : 64, bn = choice == 3 ? 64 : 128;
        a.splits = (uint32_t)plan >> 8;
        SSD_REQUIRE(!gn_sums || a.splits > 1 || (H * W) % bm == 0, "conv2d_nhwc_f32x2_presplit: fused GroupNorm statistics need H*W to be a multiple of the M tile");
        a.m_tiles = (a.M + bm - 1) / bm; a.n_tiles = Cout / bn;
        float* ws = (splitk_ws && splitk_ws_bytes >= (size_t)a.M * Cout * 4) ? (float*)splitk_ws : nullptr;
        double* stats = a.gn_sums;
        if (a.splits > 1) {
            if (!ws && hipMemsetAsync(y, 0, (size_t)a.M * Cout * 4, st) != hipSuccess) return ssdnerf_fail(SSDNERF_E_LAUNCH, "conv2d_nhwc_f32x2_presplit: memset failed");
            a.splitk_ws = ws ? ws : (float*)y;
            a.tickets = cv_fold_tickets(a, splitk_ws, splitk_ws_bytes, bm, stats != nullptr);
            if (!a.tickets) a.gn_sums = nullptr;
        }
        const dim3 grid(a.m_tiles * a.n_tiles * a.splits);
        if (choice == 1) hipLaunchKernelGGL((k_conv_igemm_bf16<2, 2, 2, 2, 2, false, true>), grid, dim3(256), 0, st, a);
        else if (choice == 2) hipLaunchKernelGGL((k_conv_igemm_bf16<1, 2, 2, 2, 3, false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_conv_igemm_bf16<1, 1, 2, 2, CV_NS_SMALL, false, true>), grid, dim3(256), 0, st, a);
        if (a.splits > 1 && !a.tickets) {
            const uint32_t HWo = H * W, cpr = Cout / 8, rstep = 256 / cpr ? 256 / cpr : 1;
            uint32_t rows = HWo;
            while (rows > rstep && rows % 2 == 0 && (uint64_t)B * (HWo / rows) < 1024) rows /= 2;
            hipLaunchKernelGGL(k_conv_f32_finish, dim3((HWo + rows - 1) / rows, B), dim3(256), 0, st, (float*)y, ws, bias, (const float*)residual, HWo, cpr, rows, stats, a.G);
        }
        SSD_CHECK_LAUNCH("conv2d_nhwc_f32x2_presplit");
        return SSDNERF_OK;
    }
    a.m_tiles = (a.M + 255) / 256; a.n_tiles = Cout / 128;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu < 8) n_cu = 256;
        n_cu &= ~7;
    }
    const uint32_t tiles = a.m_tiles * a.n_tiles, grid = tiles < (uint32_t)n_cu ? tiles : (uint32_t)n_cu;
    hipLaunchKernelGGL((k_conv_pp_bf16<true, true, true>), dim3(grid), dim3(512), 0, st, a);
    SSD_CHECK_LAUNCH("conv2d_nhwc_f32x2_presplit");
    return SSDNERF_OK;
}

// (r03: channel counts are multiples of 8 -- one 16-byte bf16 chunk --, no longer of 64: a tensor's last K-tile and the last N tile may be partial)
extern "C" int ssdnerf_conv2d_nhwc_bf16_supported(uint32_t Cin, uint32_t Cout, uint32_t ksize, uint32_t stride, uint32_t upsample) {
    return Cin > 0 && Cout > 0 && (Cin % 8 == 0) && (Cout % 8 == 0) && (ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && !(upsample && stride != 1);
}

extern "C" int ssdnerf_conv2d_nhwc_bf16(const void* x, const void* x2, uint32_t Cin1, const void* w, const float* bias, const void* residual, void* y, uint32_t B,
                                        uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t ksize, uint32_t stride, uint32_t upsample, void* gn_sums, uint32_t gn_groups,
                                        int tile_hint, void* splitk_ws, size_t splitk_ws_bytes, int splits_hint, void* stream) {
    if (B == 0 || H == 0 || W == 0) return SSDNERF_OK;
    SSD_REQUIRE(x && w && y, "conv2d_nhwc_bf16: null pointer");
    SSD_REQUIRE(ssdnerf_conv2d_nhwc_bf16_supported(Cin, Cout, ksize, stride, upsample),
                "conv2d_nhwc_bf16: needs Cin %% 8 == 0, Cout %% 8 == 0, ksize 1|3, stride 1|2 (no stride with upsample)");
    SSD_REQUIRE(!gn_sums || (gn_groups > 0 && Cout % gn_groups == 0 && (Cout / gn_groups) % 4 == 0),
                "conv2d_nhwc_bf16: fused GroupNorm statistics need groups of a multiple of 4 channels");
    if (!x2) Cin1 = Cin;
    SSD_REQUIRE(Cin1 <= Cin && Cin1 % 8 == 0 && (x2 || Cin1 == Cin), "conv2d_nhwc_bf16: the first input's channel count must be a multiple of 8 and <= Cin");
    SSD_REQUIRE((uint64_t)Cout * ksize * ksize * Cin * 2 < (1ull << 31), "conv2d_nhwc_bf16: weight tensor too large");
    const bool c64 = Cin % 64 == 0 && Cin1 % 64 == 0;                         // what the row-reuse and two-group kernels take
    ConvArgs a;
    a.tickets = nullptr;
    a.x2 = (const unsigned char*)x2; a.Cin1 = Cin1;
    a.x = (const unsigned char*)x; a.w = (const unsigned char*)w; a.bias = bias; a.res = (const unsigned char*)residual; a.y = (unsigned char*)y;
    a.gn_sums = (double*)gn_sums; a.G = gn_groups ? gn_groups : 1;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.ksize = ksize; a.stride = stride; a.pad = ksize / 2; a.upsample = upsample;
    const uint32_t Hv = upsample ? 2 * H : H, Wv = upsample ? 2 * W : W;
    a.Ho = (Hv + 2 * a.pad - ksize) / stride + 1;
    a.Wo = (Wv + 2 * a.pad - ksize) / stride + 1;
    SSD_REQUIRE((uint64_t)B * a.Ho * a.Wo < (1ull << 31) && (uint64_t)B * H * W * Cin * 2 < (1ull << 31), "conv2d_nhwc_bf16: tensor too large (32-bit buffer offsets)");
    a.M = B * a.Ho * a.Wo;
    hipStream_t st = (hipStream_t)stream;
    // tile_hint 0: the two-group kernel takes the layers it measures faster on (r03, profiles/r03/e_bench_conv_two_group.jsonl): 3 x 3 / stride 1 layers
    // whose 256-pixel tiles are whole image rows, from 64 x 64 upwards (row-reuse form), and the upsampling convolution of the 128 x 128 level
    static const bool pp_auto = getenv("SSDNERF_CONV_NO_TWO_GROUP") == nullptr;
    if (tile_hint == 0 && pp_auto && c64 && Cout % 128 == 0 && ksize == 3 && stride == 1 && a.M >= 32768 && (!gn_sums || (a.Ho * a.Wo) % 256 == 0)
        && ((!upsample && (W == 128 || W == 64) && (H * W) % 256 == 0) || (upsample && a.M >= 131072)))
        tile_hint = 6;
    if (tile_hint == 5 || tile_hint == 6) {                                  // two-group ("ping-pong") 256 x 128 kernel; 6: the row-reuse form where it applies
        SSD_REQUIRE(c64 && Cout % 128 == 0, "conv2d_nhwc_bf16: the 256 x 128 kernel needs Cin %% 64 == 0 and Cout %% 128 == 0");
        SSD_REQUIRE(!gn_sums || (a.Ho * a.Wo) % 256 == 0, "conv2d_nhwc_bf16: fused GroupNorm statistics need Ho*Wo to be a multiple of the M tile");
        a.splits = 1;
#ifdef CV_PP_TIMING
        a.splitk_ws = (float*)splitk_ws;                                     // instrumented build: the caller's scratch receives the time stamps
#else
        a.splitk_ws = nullptr;
#endif
        a.m_tiles = (a.M + 255) / 256; a.n_tiles = Cout / 128;
        const bool pp_rows = tile_hint == 6 && ksize == 3 && stride == 1 && !upsample && (W == 128 || W == 64 || W == 32) && (H * W) % 256 == 0;
        static int n_cu = 0;                                                 // one block per CU (LDS), persistent over the tiles
        if (n_cu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
            if (n_cu < 8) n_cu = 256;
            n_cu &= ~7;
        }
        const uint32_t tiles = a.m_tiles * a.n_tiles, grid = tiles < (uint32_t)n_cu ? tiles : (uint32_t)n_cu;
        if (pp_rows) hipLaunchKernelGGL((k_conv_pp_bf16<true>), dim3(grid), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_conv_pp_bf16<false>), dim3(grid), dim3(512), 0, st, a);
        SSD_CHECK_LAUNCH("conv2d_nhwc_bf16");
        return SSDNERF_OK;
    }
    int choice; uint32_t splits;
    cv_plan(a.M, Cin, Cout, ksize, tile_hint, splitk_ws && splitk_ws_bytes >= (size_t)a.M * Cout * 4 && Cout <= 1024, splits_hint, &choice, &splits);
    if (choice != 3) SSD_REQUIRE(Cout % 128 == 0, "conv2d_nhwc_bf16: 128-wide tiles need Cout %% 128 == 0");
    SSD_REQUIRE(!gn_sums || splits > 1 || (a.Ho * a.Wo) % (choice == 4 ? 256 : choice == 1 ? 128 : 64) == 0, "conv2d_nhwc_bf16: fused GroupNorm statistics need Ho*Wo to be a multiple of the M tile");
    a.splits = splits; a.splitk_ws = (float*)splitk_ws;
    double* stats = a.gn_sums;
    a.tickets = nullptr;
    if (splits > 1) {
        const uint32_t bm = choice == 4 ? 256 : choice == 1 ? 128 : 64, bn = choice == 3 ? 64 : 128;
        a.m_tiles = (a.M + bm - 1) / bm; a.n_tiles = (Cout + bn - 1) / bn;     // (what cv_launch sets; the fold needs the tile count now)
        a.tickets = cv_fold_tickets(a, splitk_ws, splitk_ws_bytes, bm, stats != nullptr);
        if (!a.tickets) a.gn_sums = nullptr;                                 // a split layer's statistics are taken by the finishing pass, or -- folded -- by the last block's epilogue
    }
    static const bool rows_ok = getenv("SSDNERF_CONV_NO_ROW_REUSE") == nullptr;
    const bool rows = rows_ok && c64 && choice == 1 && splits == 1 && ksize == 3 && stride == 1 && !upsample && (W == 128 || W == 64 || W == 32) && (H * W) % 128 == 0;
    if (rows) {
        a.m_tiles = (a.M + 127) / 128; a.n_tiles = Cout / 128;
        hipLaunchKernelGGL(k_conv3x3_bf16_rows, dim3(a.m_tiles * a.n_tiles), dim3(256), 0, st, a);
    } else if (choice == 4) cv_launch<2, 2, 4, 2, 3>(a, st); else if (choice == 1) cv_launch<2, 2, 2, 2, 2>(a, st); else if (choice == 2) cv_launch<1, 2, 2, 2, 3>(a, st); else cv_launch<1, 1, 2, 2, CV_NS_SMALL>(a, st);
    if (splits > 1 && !a.tickets) {
        const uint32_t HWo = a.Ho * a.Wo, cpr = Cout / 8, rstep = 256 / cpr ? 256 / cpr : 1;
        uint32_t rows = HWo;                                                 // rows per block: >= one pass of the rows in flight, ~1024 blocks
        while (rows > rstep && rows % 2 == 0 && (uint64_t)B * (HWo / rows) < 1024) rows /= 2;
        hipLaunchKernelGGL(k_conv_splitk_finish, dim3((HWo + rows - 1) / rows, B), dim3(256), 0, st, a.splitk_ws, bias, a.res, a.y, HWo, cpr, rows, stats, a.G);
    }
    SSD_CHECK_LAUNCH("conv2d_nhwc_bf16");
    return SSDNERF_OK;
}
