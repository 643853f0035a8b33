// oracle/ref_shim/cpu_cuda_shim.h -- TEST INFRASTRUCTURE (oracle), not product code.
//
// A minimal CPU stand-in for the CUDA / ATen surface that the reference's two
// extension sources (lib/ops/raymarching/src/raymarching.cu and
// lib/ops/shencoder/src/shencoder.cu) touch, so that those files can be compiled
// *where they lie* under /root/reference by g++ and executed on the host as the
// strongest available pin for oracle/ (see oracle/build_ref.sh).  Nothing here is
// derived from the reference: it only defines the vocabulary (__global__,
// threadIdx, atomicAdd, at::Tensor::data_ptr<T>() ...) with serial CPU semantics.
//
// Thread model: every "kernel launch" is executed as a serial double loop over
// (block, thread).  The reference kernels use no shared memory and no barriers
// (SURVEY.md 2a), so serial execution is a valid schedule of the CUDA program.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline

struct cpu_dim3 { unsigned x = 0, y = 0, z = 0; };
extern thread_local cpu_dim3 threadIdx, blockIdx, blockDim, gridDim;

template <class F>
static inline void cpu_launch_impl(unsigned grid, unsigned block, F&& body) {
    gridDim.x = grid; blockDim.x = block;
    for (unsigned b = 0; b < grid; ++b) {
        blockIdx.x = b;
        for (unsigned t = 0; t < block; ++t) { threadIdx.x = t; body(); }
    }
}
// `kernel<<<grid, block>>>(args...)` is rewritten by build_ref.sh (a sed on the
// byte stream, never stored) into CPU_LAUNCH(grid, block, kernel, args...).
#define CPU_LAUNCH(G, B, K, ...) cpu_launch_impl((unsigned)(G), (unsigned)(B), [&] { K(__VA_ARGS__); })

// returns the old value, like CUDA's atomicAdd; serial execution makes it trivially atomic.
static inline int atomicAdd(int* addr, int val) { int old = *addr; *addr = old + val; return old; }

// CUDA's __expf is ex2.approx(x*log2e); the closest host statement is expf (documented tolerance).
static inline float cpu_fast_expf(float x) { return expf(x); }
#define __expf(x) cpu_fast_expf(x)   /* glibc already declares a symbol called __expf */
using std::max;
using std::min;

struct __half { uint16_t bits; };
namespace c10 { struct Half { uint16_t bits; }; }

namespace at {
enum class ScalarType { Float, Half, Double, Int, Byte };
struct Device { bool cuda = true; bool is_cuda() const { return cuda; } };
struct Tensor {
    void* ptr = nullptr;
    ScalarType st = ScalarType::Float;
    Tensor() = default;
    Tensor(void* p, ScalarType s) : ptr(p), st(s) {}
    template <class T> T* data_ptr() const { return static_cast<T*>(ptr); }
    ScalarType scalar_type() const { return st; }
    Device device() const { return Device{}; }
    bool is_contiguous() const { return true; }
};
}  // namespace at
namespace torch { using at::Tensor; }

#define TORCH_CHECK(cond, ...) do { if (!(cond)) throw std::runtime_error("TORCH_CHECK failed: " #cond); } while (0)
// The reference python wrappers force fp32 (custom_fwd(cast_inputs=float32)); only that arm is instantiated.
#define AT_DISPATCH_FLOATING_TYPES_AND_HALF(TYPE, NAME, ...) do { using scalar_t = float; (void)(TYPE); __VA_ARGS__(); } while (0)
