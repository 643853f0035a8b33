// tools/ubench/atomic_scope.hip -- fp32 atomic-add throughput on MI355X by memory scope, and which XCD a workgroup runs on.
//   agent scope     : what atomicAdd() emits; with 8 XCDs (8 L2s) the read-modify-write has to happen at a point all XCDs share
//   workgroup scope : the RMW may complete in the issuing XCD's L2 -- only correct if every contributor to an address runs on ONE XCD
// Prints: M atomics / s per scope for 2^22 atomics into a 1.5 MiB image (the decode backward's gradient image), the histogram of
// HW_REG_XCC_ID over workgroups (is blockIdx % 8 the XCD?), and whether the workgroup-scope sum is exact when the work is pinned per XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)); }

template <int SCOPE>   // 0 agent, 1 workgroup, 2 wavefront
__global__ void k_atomics(float* img, uint32_t words, uint32_t per_thread, uint32_t pin_xcd /* 0xffffffff: no pinning */, uint32_t* xcc_hist) {
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(xcc_hist + (blockIdx.x & 7) * 8 + (x & 7), 1u);
    if (pin_xcd != 0xffffffffu && x != pin_xcd) return;
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t i = 0; i < per_thread; ++i) {
        s = s * 1664525u + 1013904223u;
        float* p = img + (s >> 8) % words;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (SCOPE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

template <int SCOPE> void run(const char* name, float* img, uint32_t words, uint32_t pin, uint32_t* hist) {
    const uint32_t blocks = 4096, tpb = 256, per_thread = 16;
    hipMemset(img, 0, words * 4);
    hipLaunchKernelGGL((k_atomics<SCOPE>), dim3(blocks), dim3(tpb), 0, 0, img, words, per_thread, pin, (uint32_t*)nullptr);
    hipDeviceSynchronize();
    hipMemset(img, 0, words * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_atomics<SCOPE>), dim3(blocks), dim3(tpb), 0, 0, img, words, per_thread, pin, hist);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(words);
    hipMemcpy(h.data(), img, words * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (float v : h) sum += v;
    const double issued = pin == 0xffffffffu ? (double)blocks * tpb * per_thread : -1;
    printf("%-28s %8.3f ms  sum %.0f%s\n", name, ms, sum, issued > 0 ? (sum == issued ? "  (exact)" : "  (LOST UPDATES)") : "  (pinned: one XCD's share)");
    if (issued > 0) printf("    -> %.0f M atomics/s\n", issued / ms / 1e3);
}

int main() {
    const uint32_t words = 3 * 128 * 128 * 8;
    float* img; uint32_t* hist;
    hipMalloc(&img, words * 4); hipMalloc(&hist, 64 * 4); hipMemset(hist, 0, 64 * 4);
    run<0>("agent scope", img, words, 0xffffffffu, hist);
    run<1>("workgroup scope (unpinned)", img, words, 0xffffffffu, nullptr);
    run<2>("wavefront scope (unpinned)", img, words, 0xffffffffu, nullptr);
    run<1>("workgroup scope, XCD 3 only", img, words, 3, nullptr);
    run<0>("agent scope, XCD 3 only", img, words, 3, nullptr);
    std::vector<uint32_t> h(64);
    hipMemcpy(h.data(), hist, 64 * 4, hipMemcpyDeviceToHost);
    printf("workgroups by (blockIdx %% 8) -> XCC_ID histogram:\n");
    for (int b = 0; b < 8; ++b) { printf("  b%%8=%d:", b); for (int x = 0; x < 8; ++x) printf(" %5u", h[b * 8 + x]); printf("\n"); }
    return 0;
}
