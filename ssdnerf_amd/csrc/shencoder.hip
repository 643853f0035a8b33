// ssdnerf_amd/csrc/shencoder.hip -- Part 1 of the C ABI: the two operators of the reference's
// `_shencoder` pybind module (lib/ops/shencoder/src/bindings.cpp:5-8).  One lane per direction; outputs
// are staged through LDS so that each wave writes its 64 x C^2 block as contiguous 256-byte rows
// instead of 64 interleaved strided streams.
#include "sh_basis.h"

static constexpr unsigned SH_TPB = 128;  // 128 x (64+1) floats of LDS staging at degree 8

template <int C, bool GRAD>
__global__ void __launch_bounds__(SH_TPB) k_sh_forward(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B, uint32_t D,
                                                        float* __restrict__ dy_dx) {
    constexpr int C2 = C * C;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    float y[C2], gx[GRAD ? C2 : 1], gy[GRAD ? C2 : 1], gz[GRAD ? C2 : 1];
    float vx = 0.f, vy = 0.f, vz = 1.f;
    if (b < B) { vx = inputs[(uint64_t)b * D]; vy = inputs[(uint64_t)b * D + 1]; vz = inputs[(uint64_t)b * D + 2]; }
    shb::eval<C, GRAD>(vx, vy, vz, y, gx, gy, gz);
    // transpose through LDS: lane-major registers -> row-contiguous global stores
    __shared__ float tile[SH_TPB * (C2 + 1)];
    const uint32_t block_first = blockIdx.x * blockDim.x;
    const uint32_t rows = min((uint32_t)SH_TPB, B > block_first ? B - block_first : 0u);
    auto flush = [&](const float* v, float* dst /* [rows, stride] block base */, uint32_t stride) {
#pragma unroll
        for (int i = 0; i < C2; ++i) tile[threadIdx.x * (C2 + 1) + i] = v[i];
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < rows * C2; e += SH_TPB) {
            const uint32_t r = e / C2, c = e - r * C2;
            dst[(uint64_t)r * stride + c] = tile[r * (C2 + 1) + c];
        }
        __syncthreads();
    };
    flush(y, outputs + (uint64_t)block_first * C2, C2);
    if (GRAD) {
        float* base = dy_dx + (uint64_t)block_first * 3 * C2;
        flush(gx, base, 3 * C2);
        flush(gy, base + C2, 3 * C2);
        flush(gz, base + 2 * C2, 3 * C2);
    }
}

template <int C>
static void launch_sh(const float* inputs, float* outputs, uint32_t B, uint32_t D, bool grad, float* dy_dx, hipStream_t s) {
    dim3 g(ssd_blocks(B, SH_TPB)), b(SH_TPB);
    if (grad) hipLaunchKernelGGL((k_sh_forward<C, true>), g, b, 0, s, inputs, outputs, B, D, dy_dx);
    else hipLaunchKernelGGL((k_sh_forward<C, false>), g, b, 0, s, inputs, outputs, B, D, dy_dx);
}

extern "C" int ssdnerf_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs,
                                         float* dy_dx, void* stream) {
    if (B == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(inputs && outputs, "sh_encode_forward: null pointer");
    SSD_REQUIRE(D == 3, "sh_encode_forward: input dim must be 3 (got %u)", D);
    SSD_REQUIRE(C >= 1 && C <= 8, "sh_encode_forward: degree must be in [1, 8] (got %u)", C);
    SSD_REQUIRE(!calc_grad_inputs || dy_dx, "sh_encode_forward: dy_dx is null but calc_grad_inputs is set");
    hipStream_t s = (hipStream_t)stream;
    const bool g = calc_grad_inputs != 0;
    switch (C) {
        case 1: launch_sh<1>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 2: launch_sh<2>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 3: launch_sh<3>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 4: launch_sh<4>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 5: launch_sh<5>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 6: launch_sh<6>(inputs, outputs, B, D, g, dy_dx, s); break;
        case 7: launch_sh<7>(inputs, outputs, B, D, g, dy_dx, s); break;
        default: launch_sh<8>(inputs, outputs, B, D, g, dy_dx, s); break;
    }
    SSD_CHECK_LAUNCH("sh_encode_forward");
    return SSDNERF_OK;
}

// grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]: one lane per (b,d) like the reference, the
// C^2-long dot product read as contiguous rows.
__global__ void k_sh_backward(const float* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2, const float* __restrict__ dy_dx,
                              float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / D;
    if (b >= B) return;
    const float* g = grad + (uint64_t)b * C2;
    const float* j = dy_dx + (uint64_t)t * C2;
    float acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ++ch) acc = ssd_fma(g[ch], j[ch], acc);
    grad_inputs[t] = acc;
}

extern "C" int ssdnerf_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx,
                                          float* grad_inputs, void* stream) {
    (void)inputs;
    if (B == 0) return SSDNERF_OK;  // empty input: nothing to do (pointers may be null)
    SSD_REQUIRE(grad && dy_dx && grad_inputs, "sh_encode_backward: null pointer");
    SSD_REQUIRE(D == 3, "sh_encode_backward: input dim must be 3 (got %u)", D);
    SSD_REQUIRE(C >= 1 && C <= 8, "sh_encode_backward: degree must be in [1, 8] (got %u)", C);
    hipLaunchKernelGGL(k_sh_backward, dim3(ssd_blocks((uint64_t)B * D, SH_TPB)), dim3(SH_TPB), 0, (hipStream_t)stream, grad, B, D, C * C, dy_dx, grad_inputs);
    SSD_CHECK_LAUNCH("sh_encode_backward");
    return SSDNERF_OK;
}
