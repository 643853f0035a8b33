// oracle/ref_glue.cpp -- TEST INFRASTRUCTURE (oracle), not product code.
//
// C-callable entry points (ctypes) into the REFERENCE's own host functions
// (declared in /root/reference/lib/ops/raymarching/src/raymarching.h:7-18 and
// /root/reference/lib/ops/shencoder/src/shencoder.h:9-12), compiled for the CPU
// through oracle/ref_shim/.  Built only by oracle/build_ref.sh, output only into
// oracle/_ref/ (git-ignored).  All pointers are host pointers; float = fp32.
#include "ref_shim/cpu_cuda_shim.h"
#include "raymarching.h"   // found via -I/root/reference/lib/ops/raymarching/src
#include "shencoder.h"     // found via -I/root/reference/lib/ops/shencoder/src

thread_local cpu_dim3 threadIdx, blockIdx, blockDim, gridDim;

using at::Tensor;
using at::ScalarType;
static inline Tensor F(const void* p) { return Tensor(const_cast<void*>(p), ScalarType::Float); }
static inline Tensor I(const void* p) { return Tensor(const_cast<void*>(p), ScalarType::Int); }
static inline Tensor B(const void* p) { return Tensor(const_cast<void*>(p), ScalarType::Byte); }

extern "C" {
void ref_near_far_from_aabb(const float* o, const float* d, const float* aabb, uint32_t N, float min_near, float* nears, float* fars) {
    near_far_from_aabb(F(o), F(d), F(aabb), N, min_near, F(nears), F(fars));
}
void ref_sph_from_ray(const float* o, const float* d, float radius, uint32_t N, float* coords) {
    sph_from_ray(F(o), F(d), radius, N, F(coords));
}
void ref_morton3D(const int* coords, uint32_t N, int* indices) { morton3D(I(coords), N, I(indices)); }
void ref_morton3D_invert(const int* indices, uint32_t N, int* coords) { morton3D_invert(I(indices), N, I(coords)); }
void ref_packbits(const float* grid, uint32_t N, float thresh, uint8_t* bits) { packbits(F(grid), N, thresh, B(bits)); }
void ref_march_rays_train(const float* o, const float* d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                          uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars,
                          float* xyzs, float* dirs, float* deltas, int* rays, int* counter, const float* noises) {
    march_rays_train(F(o), F(d), B(grid), bound, dt_gamma, max_steps, N, C, H, M, F(nears), F(fars),
                     F(xyzs), F(dirs), F(deltas), I(rays), I(counter), F(noises));
}
void ref_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int* rays,
                                      uint32_t M, uint32_t N, float T_thresh, float* ws, float* depth, float* image) {
    composite_rays_train_forward(F(sigmas), F(rgbs), F(deltas), I(rays), M, N, T_thresh, F(ws), F(depth), F(image));
}
void ref_composite_rays_train_backward(const float* g_ws, const float* g_img, const float* sigmas, const float* rgbs,
                                       const float* deltas, const int* rays, const float* ws, const float* image,
                                       uint32_t M, uint32_t N, float T_thresh, float* g_sigmas, float* g_rgbs) {
    composite_rays_train_backward(F(g_ws), F(g_img), F(sigmas), F(rgbs), F(deltas), I(rays), F(ws), F(image), M, N, T_thresh,
                                  F(g_sigmas), F(g_rgbs));
}
void ref_march_rays(uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t, const float* o, const float* d,
                    float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                    const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises) {
    march_rays(n_alive, n_step, I(rays_alive), F(rays_t), F(o), F(d), bound, dt_gamma, max_steps, C, H, B(grid), F(nears), F(fars),
               F(xyzs), F(dirs), F(deltas), F(noises));
}
void ref_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int* rays_alive, float* rays_t, const float* sigmas,
                        const float* rgbs, const float* deltas, float* ws, float* depth, float* image) {
    composite_rays(n_alive, n_step, T_thresh, I(rays_alive), F(rays_t), F(sigmas), F(rgbs), F(deltas), F(ws), F(depth), F(image));
}
void ref_sh_encode_forward(const float* inputs, float* outputs, uint32_t Bn, uint32_t D, uint32_t C, int calc_grad, float* dy_dx) {
    sh_encode_forward(F(inputs), F(outputs), Bn, D, C, calc_grad != 0, F(dy_dx));
}
void ref_sh_encode_backward(const float* grad, const float* inputs, uint32_t Bn, uint32_t D, uint32_t C, const float* dy_dx, float* grad_inputs) {
    sh_encode_backward(F(grad), F(inputs), Bn, D, C, F(dy_dx), F(grad_inputs));
}
}
