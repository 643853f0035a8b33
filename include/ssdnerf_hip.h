/* include/ssdnerf_hip.h -- C ABI of libssdnerf_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for SSDNeRF's volumetric-rendering hot path.  Every entry point takes plain
 * DEVICE pointers, sizes and a HIP stream handle (void* == hipStream_t, NULL = default stream);
 * no torch / ATen types cross this boundary.  Ownership follows the reference: the CALLER allocates
 * every output; the library allocates nothing, holds no state and is re-entrant per stream.
 *
 * Return value: 0 on success, otherwise a negative SSDNERF_E_* code; ssdnerf_last_error() returns a
 * thread-local human-readable message (the reference surfaces C++ exceptions as Python RuntimeError;
 * the Python mirror in ssdnerf_amd/_cabi.py raises RuntimeError on any non-zero status).
 * Unlike the reference (raymarching.cu launches on the legacy default stream and checks nothing),
 * arguments are validated and hipGetLastError() is checked after every launch.
 *
 * Part 1 replaces, one for one, the 10 + 2 functions the reference binds with pybind11:
 *     lib/ops/raymarching/src/bindings.cpp:5-18   (signatures: raymarching.h:7-18)
 *     lib/ops/shencoder/src/bindings.cpp:5-8      (signatures: shencoder.h:9-12)
 * Part 2 is the fused fast path that sits behind TriPlaneDecoder.point_decode /
 * VolumeRenderer.forward / BaseNeRF.update_extra_state (same results, fewer HBM round trips).
 *
 * All floating-point tensors are fp32 unless a dtype argument says otherwise (the reference's Python
 * wrappers force fp32 with custom_fwd(cast_inputs=torch.float32)).
 */
#ifndef SSDNERF_HIP_H
#define SSDNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDNERF_OK 0
#define SSDNERF_E_INVALID (-1)   /* bad argument (null pointer, size out of range, unsupported degree ...) */
#define SSDNERF_E_LAUNCH (-2)    /* HIP reported an error at or after kernel launch */
#define SSDNERF_E_WORKSPACE (-3) /* caller-provided workspace too small */

#define SSDNERF_DTYPE_F32 0
#define SSDNERF_DTYPE_F16 1

const char* ssdnerf_last_error(void);
/* ABI version: bumped whenever a signature below changes. */
int ssdnerf_abi_version(void);

/* ============================ Part 1: reference operator surface ============================ */

/* replaces near_far_from_aabb (raymarching.h:7; kernel raymarching.cu:91-145).
 * rays_o, rays_d [N,3]; aabb [6] = (xmin,ymin,zmin,xmax,ymax,zmax); nears, fars [N].
 * A miss writes FLT_MAX to both. */
int ssdnerf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                               float* nears, float* fars, void* stream);

/* replaces sph_from_ray (raymarching.h:8; raymarching.cu:162-198). coords [N,2]. */
int ssdnerf_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream);

/* replaces morton3D / morton3D_invert (raymarching.h:9-10; raymarching.cu:214-254). coords [N,3] int32. */
int ssdnerf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream);
int ssdnerf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream);

/* replaces packbits (raymarching.h:11; raymarching.cu:267-289). grid [8*N] of grid_dtype (the reference is
 * templated on float/half); bitfield [N]; bit i of byte n = grid[8n+i] > density_thresh. */
int ssdnerf_packbits(const void* grid, int grid_dtype, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream);

/* replaces march_rays_train (raymarching.h:13; raymarching.cu:311-482).
 * rays [N,3] int32 = (ray id, point offset, point count); counter [2] int32 is ACCUMULATED like the
 * reference's atomicAdd: counter[0] += total points, counter[1] += N.  xyzs/dirs [M,3], deltas [M,2] = (dt, t).
 * Difference by design: slots are assigned by an exclusive prefix sum in ray order (deterministic) instead of
 * the reference's arrival-order atomicAdd; every ordering the reference can produce is a permutation of this
 * one and compositing is order-independent.  Rays whose samples would overflow M are dropped exactly like
 * raymarching.cu:415-416.  workspace: ssdnerf_march_rays_train_workspace(N) bytes. */
size_t ssdnerf_march_rays_train_workspace(uint32_t N);
int ssdnerf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                             const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                             const float* noises, void* workspace, size_t workspace_bytes, void* stream);

/* replaces composite_rays_train_forward / _backward (raymarching.h:14-15; raymarching.cu:502-698).
 * forward writes weights_sum/depth [N], image [N,3] at index rays[n,0]; backward writes grad_sigmas [M],
 * grad_rgbs [M,3] (caller zero-initialises, as the reference's Python does); grad wrt depth is not propagated.
 * M == 0 (no ray of the batch took a sample): the sample arrays may be null; forward still writes the N per-ray zeros, backward is a no-op. */
int ssdnerf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                         uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth,
                                         float* image, void* stream);
int ssdnerf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                          const float* rgbs, const float* deltas, const int32_t* rays,
                                          const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                          float T_thresh, float* grad_sigmas, float* grad_rgbs, void* stream);

/* replaces march_rays (raymarching.h:17; raymarching.cu:705-812): alive ray n writes <= n_step samples into
 * slots [n*n_step, (n+1)*n_step); the caller zero-initialises xyzs/dirs/deltas (unused slots keep dt == 0). */
int ssdnerf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                       const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                       uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                       float* dirs, float* deltas, const float* noises, void* stream);

/* replaces composite_rays (raymarching.h:18; raymarching.cu:825-913): in place on rays_alive (dead -> -1),
 * rays_t, weights_sum, depth, image. */
int ssdnerf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                           const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                           float* image, void* stream);

/* replaces sh_encode_forward / sh_encode_backward (shencoder.h:9,12; shencoder.cu:27-383).
 * inputs [B,D] with D == 3; C = degree in [1,8]; outputs [B,C*C]; dy_dx [B,3*C*C] when calc_grad_inputs.
 * backward ACCUMULATES into grad_inputs [B,3] like the reference. */
int ssdnerf_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                              int calc_grad_inputs, float* dy_dx, void* stream);
int ssdnerf_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                               const float* dy_dx, float* grad_inputs, void* stream);

/* ================================ Part 2: fused fast path =================================== */

/* Number of floats of the packed tiny-MLP parameter block consumed by the fused kernels (layout in
 * DESIGN.md "Decoder parameter block"; built from the reference's state-dict by ssdnerf_amd/decoder.py). */
#define SSDNERF_MLP_HIDDEN 64
#define SSDNERF_MLP_FEATS 18
#define SSDNERF_MLP_SH 16
#define SSDNERF_MLP_REC 24 /* per hidden unit: W1[18], b1, w_sigma, Wc[3], pad */
#define SSDNERF_MLP_PARAM_FLOATS (64 * 24 + 64 * 16 + 64 + 4)

/* Triplane repack: code (S,3,Cch,Hp,Wp) NCHW (the reference's layout, triplane_decoder.py:125) of dtype
 * code_dtype -> planes (S,3,Hp,Wp,8) channel-last, zero-padded to 8 channels, fp32 (32 B per texel) or
 * fp16 (16 B per texel).  Cch <= 8.  One bilinear corner then is one (two) 16-byte load(s). */
int ssdnerf_triplane_pack(const void* code, int code_dtype, uint32_t S, uint32_t Cch, uint32_t Hp, uint32_t Wp,
                          void* planes, int planes_dtype, void* stream);

/* Fused TriPlaneDecoder.point_decode (triplane_decoder.py:119-179) for ONE scene's packed planes:
 * bilinear/border/align_corners=False gather of 3 planes -> 18 features -> base 18->64 -> sigma = exp(.)
 * and rgb = sigmoid(color(silu(h + dir(SH4(d)))))*(1+2*sat)-sat.  xyzs, dirs [P,3]; sigmas [P]; rgbs [P,3]
 * (rgbs and dirs may be NULL for density-only == point_density_decode). */
int ssdnerf_point_decode(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                         const float* xyzs, const float* dirs, uint32_t P, float sigmoid_saturation, float* sigmas,
                         float* rgbs, void* stream);

/* The train-branch march of VolumeRenderer.forward for ALL S scenes of a batch (base_volume_renderer.py:59-77 calls march_rays_train once
 * per scene, each call ending in a device->host read of the sample count, raymarching.py:268-274).  Two calls around ONE host read:
 *   _count: rays_o/rays_d (S,N,3), grids (S, C*H^3/8) bitfields, nears/fars/noises (S,N); dt_gammas (S, device) or NULL (-> dt_gamma for all);
 *           scene_offsets [S+1] int32 <- first sample of every scene in the packed arrays, scene_offsets[S] = total samples M.
 *   _write: (same inputs, SAME workspace, after the caller has read M and allocated) xyzs/dirs [M,3], deltas [M,2] = (dt, t);
 *           rays [S*N,3] int32 = (n, offset, count) with ray index n and offset GLOBAL over the batch -- the form the reference's
 *           batch_composite_rays_train builds by rebasing (raymarching.py:349-395), so ssdnerf_composite_rays_train_* take them as they are.
 * Per-ray arithmetic and deterministic slot order as ssdnerf_march_rays_train; samples are packed WITHOUT the per-scene 128-row padding
 * (padding rows are never composited).  workspace: ssdnerf_march_rays_train_batch_workspace(S, N) bytes. */
size_t ssdnerf_march_rays_train_batch_workspace(uint32_t S, uint32_t N);
int ssdnerf_march_rays_train_batch_count(const float* rays_o, const float* rays_d, const uint8_t* grids, float bound, float dt_gamma,
                                         const float* dt_gammas, uint32_t max_steps, uint32_t S, uint32_t N, uint32_t C, uint32_t H,
                                         const float* nears, const float* fars, const float* noises, int32_t* scene_offsets,
                                         void* workspace, size_t workspace_bytes, void* stream);
int ssdnerf_march_rays_train_batch_write(const float* rays_o, const float* rays_d, const uint8_t* grids, float bound, float dt_gamma,
                                         const float* dt_gammas, uint32_t max_steps, uint32_t S, uint32_t N, uint32_t C, uint32_t H,
                                         uint32_t M, const float* nears, const float* fars, const float* noises, float* xyzs,
                                         float* dirs, float* deltas, int32_t* rays, void* workspace, size_t workspace_bytes, void* stream);

/* Gradient of ssdnerf_point_decode w.r.t. the scene codes with the decoder frozen -- what autograd does through grid_sample and the
 * four nn.Linear layers (triplane_decoder.py:136-179, TruncExp.backward lib/ops/activation.py:15-20) when the rendering loss is
 * differentiated w.r.t. the scene code (guidance: diffusion_nerf.py:282-294; fine-tuning: base_nerf.py:446-470) -- for ALL scenes of a
 * batch: planes (S,3,Hp,Wp,8); the samples of scene s are rows [offsets[s], offsets[s+1]) of xyzs / dirs / grad_sigmas / grad_rgbs
 * (offsets: S+1 uint32 in DEVICE memory, offsets[S] == total).  grad_sigmas [total] (may be NULL), grad_rgbs [total,3] (NULL together
 * with dirs: density head only) -> grad_code (S,3,6,Hp,Wp) fp32, the code's own NCHW layout, OVERWRITTEN (no zero-fill needed).
 * Nothing is saved by the forward; the hidden units are recomputed.  Samples with an all-zero upstream gradient are skipped.  The
 * scatter is a binned reduction in LDS (no global atomics); additions inside a 32x32-texel tile are unordered, so results are
 * reproducible to fp32 rounding.  workspace: ssdnerf_point_decode_backward_workspace bytes (108 B per sample + the per-split tile images).
 * planes_dtype | SSDNERF_DECODE_BWD_FEAT_MFMA (r06, opt-in, colour + density gradients only): the per-sample feature gradients through three matrix-core products per
 * 64 samples (bf16-pair operands, fp32 accumulation: the arithmetic class of the UNet's fp32-class kernels) instead of fp32 fused multiply-adds on the vector ALUs. */
#define SSDNERF_DECODE_BWD_FEAT_MFMA 0x100
size_t ssdnerf_point_decode_backward_workspace(uint32_t S, uint32_t total, uint32_t Hp, uint32_t Wp);
int ssdnerf_point_decode_backward(const void* planes, int planes_dtype, uint32_t S, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                  const float* xyzs, const float* dirs, const uint32_t* offsets, uint32_t total,
                                  float sigmoid_saturation, const float* grad_sigmas, const float* grad_rgbs, float* grad_code,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* Fused eval-branch render of VolumeRenderer.forward (base_volume_renderer.py:79-123) + the background blend
 * of BaseNeRF.render (base_nerf.py:522-523) for ONE scene: AABB -> bitfield-guided march -> gather -> MLP ->
 * composite entirely on chip.  rays_o/rays_d [N,3] in, image [N,3] (already blended with bg_color), depth [N],
 * weights_sum [N] out; sample_counts [N] int32 (optional, may be NULL) receives the number of samples each ray
 * took - the integer parity contract.  overflow_flag [1] int32 (optional) is incremented by every ray that reached
 * the reference loop's sample cap (>= max_steps samples), where the reference's result depends on its global
 * n_step schedule; callers re-render those scenes through the stepwise ops (never happens for bound=1 scenes
 * unless a ray crosses 256 occupied steps). */
int ssdnerf_render_rays_fused(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                              const uint8_t* bitfield, uint32_t grid_size, const float* rays_o, const float* rays_d,
                              uint32_t N, float bound, float min_near, float dt_gamma, uint32_t max_steps, float T_thresh,
                              float bg_color, float sigmoid_saturation, float* image, float* depth, float* weights_sum,
                              int32_t* sample_counts, int32_t* overflow_flag, void* stream);

/* Same, for S scenes in ONE launch (one tail instead of S): planes (S,3,Hp,Wp,8), bitfield (S, H^3/8), rays_o/rays_d (S,N,3),
 * outputs (S,N,3) / (S,N) dense.  dt_gammas: optional DEVICE array [S] of per-scene cone angles (the reference pulls each
 * one to the host with .item(), base_volume_renderer.py:112); NULL -> dt_gamma for every scene. */
int ssdnerf_render_rays_fused_batch(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                    const uint8_t* bitfield, uint32_t grid_size, const float* rays_o, const float* rays_d,
                                    uint32_t S, uint32_t N, float bound, float min_near, float dt_gamma, const float* dt_gammas,
                                    uint32_t max_steps, float T_thresh, float bg_color, float sigmoid_saturation, float* image,
                                    float* depth, float* weights_sum, int32_t* sample_counts, int32_t* overflow_flag,
                                    void* stream);

/* The same render as two stages (the fast path for power-of-two grids; csrc/render_queue.hip):
 *   first_hit   : a conservative empty-space pre-test finishes the rays that cannot meet an occupied cell (background written)
 *                 and lists the others; the listed rays are marched, densely, to their first occupied sample; rays without one
 *                 are finished too, the others are appended to a per-scene hit queue inside `workspace`;
 *   shade_queue : persistent waves shade the queued rays (gather + MLP + composite + onward march).
 * Both take the SAME workspace (ssdnerf_render_queue_workspace(S, N, grid_size) bytes, caller-owned) and must be issued
 * in this order on one stream.  Results are bit-identical to ssdnerf_render_rays_fused_batch, and from run to run (soaked over 10^5 renders of 33 M rays since r06:
 * the build removes the one instruction kind that made rounds 1 - 5 differ about once in 3 000 renders -- ssdnerf_amd/asm_postpass.py).
 * r05: first_hit also leaves, per hit-queue entry, a bound of the ray's remaining march steps and -- k_ticket_order -- the order in which the
 * MFMA shading kernel takes the queue's 64-entry slices (longest first; + 1.06 B per ray of workspace).  The order never enters a ray's result.
 * Environment SSDNERF_TICKET_ORDER=0 (read per call, by both stages) restores the two-class queue of earlier rounds. */
size_t ssdnerf_render_queue_workspace(uint32_t S, uint32_t N, uint32_t grid_size);
int ssdnerf_render_first_hit(const uint8_t* bitfield, uint32_t grid_size, const float* rays_o, const float* rays_d, uint32_t S,
                             uint32_t N, float bound, float min_near, float dt_gamma, const float* dt_gammas, uint32_t max_steps,
                             float bg_color, float* image, float* depth, float* weights_sum, int32_t* sample_counts,
                             void* workspace, size_t workspace_bytes, void* stream);
int ssdnerf_render_shade_queue(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                               uint32_t grid_size, const float* rays_o, const float* rays_d, uint32_t S, uint32_t N, float bound,
                               float min_near, float dt_gamma, const float* dt_gammas, uint32_t max_steps, float T_thresh,
                               float bg_color, float sigmoid_saturation, float* image, float* depth, float* weights_sum,
                               int32_t* sample_counts, int32_t* overflow_flag, void* workspace, size_t workspace_bytes,
                               void* stream);

/* Stage B with the 18->64 and 16->64 MLP layers on the bf16 matrix cores in fp32-class arithmetic (every operand split exactly into
 * three bf16 terms) and wave-local LDS pools that turn long empty-space searches into full-width march passes
 * (csrc/shade_mfma.hip).  Same arguments and workspace as ssdnerf_render_shade_queue; integer outputs identical, floats within
 * fp32 rounding.
 * Precision of the direction term (the additive Wd.SH(d) correction of the colour pre-activation, triplane_decoder.py:170-173): OR
 * SSDNERF_SHADE_FULL_DIR_PRODUCTS into `planes_dtype` for all six split products (the fp32 class of the other layers -- what the Python host
 * layer passes by default since r04); without the flag three of the six are formed (hi*hi, hi*mid, mid*hi: 16 significand bits per factor; the
 * image moves by <= 1.6e-6 on the bench scenes -- an error that scales with |Wd.SH(d)| |Wc|, so it is an opt-in, TriPlaneDecoder.shade_dir_products
 * = 3 -- and the kernel is 5 % faster).  Densities, sample counts and depth do not depend on it. */
#define SSDNERF_SHADE_FULL_DIR_PRODUCTS 0x100
int ssdnerf_render_shade_queue_mfma(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                    uint32_t grid_size, const float* rays_o, const float* rays_d, uint32_t S, uint32_t N,
                                    float bound, float min_near, float dt_gamma, const float* dt_gammas, uint32_t max_steps,
                                    float T_thresh, float bg_color, float sigmoid_saturation, float* image, float* depth,
                                    float* weights_sum, int32_t* sample_counts, int32_t* overflow_flag, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* The same two stages fed with CAMERAS instead of ray arrays: c2w [S][V][4][4] row-major, intrinsics [S][V][4] = {fx, fy, cx, cy};
 * ray n = pixel (n % (h*w)) of view n / (h*w), N = V*h*w rays per scene, generated in the kernels by the arithmetic of
 * ssdnerf_cam_rays (get_cam_rays, lib/core/utils/nerf_utils.py:17-61; a generated ray is bit-identical to the stored one), so
 * BaseNeRF.render (base_nerf.py:494-533) never materialises its (S,V,h,w,3) ray arrays: 80 B per view instead of 24 B per ray.
 * Workspace: ssdnerf_render_queue_workspace(S, V*h*w, grid_size).  Outputs as above, (S, V*h*w, ...).  image_u8 (S, V*h*w, 3) uint8, may be
 * NULL: the image ALSO quantised the way eval_and_viz does before views are gathered and written (base_nerf.py:551-553: clamp to [0,1], x 255,
 * round half to even -- ssdnerf_quantize_u8's arithmetic), stored by the same kernels that store the float image (both calls must get it). */
/* Flag for the `grid_size` argument of the two ssdnerf_render_first_hit* entry points (OR it in; the grid size proper is the low 16 bits).
 * SSDNERF_FIRST_HIT_SMALL_BLOCKS (r06): the cull kernel in blocks of 256 rays (3.6 KB of LDS) instead of 2048 (21 KB), so that stage A of the NEXT render can run on a
 * second stream BESIDE the shading kernel of the current one, whose two workgroups hold 148 of a CU's 160 KB of LDS: with its own workspace and output tensors the call
 * is independent of the render in flight (the host layer's `next_batch`, ssdnerf_amd/nerf.py).  Same outputs bit for bit; slower when nothing runs beside it.
 * Grids finer than 64^3 (coarse bitfield above 512 B) take the default form. */
#define SSDNERF_FIRST_HIT_SMALL_BLOCKS 0x10000u
int ssdnerf_render_first_hit_cams(const uint8_t* bitfield, uint32_t grid_size, const float* c2w, const float* intrinsics, uint32_t S,
                                  uint32_t V, uint32_t h, uint32_t w, float bound, float min_near, float dt_gamma,
                                  const float* dt_gammas, uint32_t max_steps, float bg_color, float* image, float* depth,
                                  float* weights_sum, int32_t* sample_counts, uint8_t* image_u8, void* workspace, size_t workspace_bytes,
                                  void* stream);
int ssdnerf_render_shade_queue_mfma_cams(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                         uint32_t grid_size, const float* c2w, const float* intrinsics, uint32_t S, uint32_t V,
                                         uint32_t h, uint32_t w, float bound, float min_near, float dt_gamma, const float* dt_gammas,
                                         uint32_t max_steps, float T_thresh, float bg_color, float sigmoid_saturation, float* image,
                                         float* depth, float* weights_sum, int32_t* sample_counts, int32_t* overflow_flag,
                                         uint8_t* image_u8, void* workspace, size_t workspace_bytes, void* stream);

/* Fused full-refresh branch of BaseNeRF.update_extra_state (base_nerf.py:328-351,377-387) for S scenes:
 * for every cell of the H^3 grid (x-major order like custom_meshgrid) decode sigma at the jittered cell centre
 * (jitter [H^3,3] uniform [0,1) shared by all scenes as in the reference, or NULL for no jitter) and fold it
 * into density_grid [S,H^3] (Morton order, dtype grid_dtype): g = (g>=0) ? max(g*decay, sigma) : g.
 * mean_out [1] fp32 (optional) receives mean(max(g,0)) over all S*H^3 cells (caller zero-initialises; it is
 * accumulated with atomics), which the reference turns into the packbits threshold. */
int ssdnerf_density_grid_update(const void* planes, int planes_dtype, uint32_t Hp, uint32_t Wp, const float* mlp_params,
                                uint32_t S, uint32_t grid_size, float bound, const float* jitter, float decay,
                                void* density_grid, int grid_dtype, float* mean_out, void* stream);

/* packbits with the threshold min(*mean, density_thresh) read ON DEVICE (removes the host sync of
 * base_nerf.py:386-387). */
int ssdnerf_packbits_dev_thresh(const void* grid, int grid_dtype, uint32_t N, const float* mean, float density_thresh,
                                uint8_t* bitfield, void* stream);

/* One DDIM step of the latent in V-parameterisation, eta = 0, no guidance (gaussian_diffusion.py:213,235,281-283):
 *   x0 = clamp(sqrt_ab * x_t - sqrt_1mab * v, lo, hi);  eps = (x_t - sqrt_ab * x0) / sqrt_1mab;
 *   x_prev = sqrt_ab_prev * x0 + dir_coef * eps,   dir_coef = sqrt(1 - ab_prev).      n elements (multiple of 4). */
int ssdnerf_ddim_step_v(const float* x_t, const float* v, uint64_t n, float sqrt_ab, float sqrt_1mab, float sqrt_ab_prev,
                        float dir_coef, float clip_lo, float clip_hi, float* x0_out, float* xprev_out, void* stream);

/* Camera rays of n_views pinhole views (get_cam_rays, lib/core/utils/nerf_utils.py:17-61): c2w [n_views][4][4] row-major,
 * intrinsics [n_views][4] = {fx, fy, cx, cy}; rays_o, rays_d [n_views][h][w][3]: pixel-centre (x + 0.5) directions rotated by
 * c2w[:3,:3] and L2-normalised, origins c2w[:3,3] broadcast. */
int ssdnerf_cam_rays(const float* c2w, const float* intrinsics, uint32_t n_views, uint32_t h, uint32_t w, float* rays_o,
                     float* rays_d, void* stream);

/* y[i] = uint8(round_half_even(clamp(x[i], 0, 1) * 255)): the output quantisation of eval_and_viz (base_nerf.py:551-553). */
int ssdnerf_quantize_u8(const float* x, uint64_t n, uint8_t* y, void* stream);

/* ---- Part 3: denoising-UNet glue (lib/models/architecture/ddpm/modules.py:12-129, denoising.py:178-187) ------------------
 * Activations are channel-last: x, y are [B][HW][C] of dtype 0 = fp32, 1 = fp16, 2 = bf16.
 *
 * y = act( GroupNorm_G(X + pre_bias) * gamma + beta  [ * (1 + scale[b]) + shift[b] ] ),  act bit 0: SiLU; bit 1 (r04, fp32 only): write y PRE-SPLIT for ssdnerf_conv2d_nhwc_f32x2_presplit (see there),
 * where X = x, or -- with x2 != NULL -- the channel concatenation [x (C1 channels) | x2 (C - C1 channels)] of two tensors, which
 * is never materialised (the decoder half's `torch.cat([h, skip], dim=1)`, denoising.py:209-213).
 * pre_bias (nullable, fp32 [C]) is the bias of the convolution that produced x, folded in here so the producer needs no
 * pass of its own.  scale_shift (nullable) is the NormWithEmbedding projection of the time embedding, fp32, row b at
 * scale_shift + b * scale_shift_stride holding [scale[C] | shift[C]] (use_scale_shift_norm=True, modules.py:97-104);
 * gamma/beta fp32 [C]; statistics in fp32/fp64 for every dtype.
 * workspace: ssdnerf_group_norm_workspace(B, G) bytes, 8-byte aligned, fp64 [B][G][sum, sum of squares].
 * workspace_state: 0 = the call zeroes it and computes the statistics; 1 = already zero (a caller running many norms zeroes
 * one arena once); 2 = already holds the statistics of x (written by ssdnerf_conv2d_nhwc_bf16's gn_sums),
 * only the normalisation pass runs (pre_bias must be NULL).  y may alias x. */
size_t ssdnerf_group_norm_workspace(uint32_t B, uint32_t G);
int ssdnerf_group_norm_nhwc(const void* x, const void* x2, uint32_t C1, int dtype, uint32_t B, uint32_t HW, uint32_t C, uint32_t G, const float* pre_bias,
                            const float* gamma, const float* beta, const float* scale_shift, uint32_t scale_shift_stride,
                            float eps, int act, void* workspace, int workspace_state, void* y, void* stream);
/* The normalisation pass alone, from statistics kept per RUN of 4 consecutive channels: runs1 fp64 [B][C1/4][sum, sum of squares] for x, runs2
 * [B][(C-C1)/4][2] for x2 (NULL without x2) -- what ssdnerf_conv2d_nhwc_* write with gn_groups = Cout / 4.  A producer then does not need to
 * know how its consumer groups the channels, and the decoder's concatenated inputs (groups of 12 / 24 channels straddling the two tensors,
 * denoising.py:209-213) need no statistics pass of their own.  (C / G) % 4 == 0. */
int ssdnerf_group_norm_nhwc_runs(const void* x, const void* x2, uint32_t C1, int dtype, uint32_t B, uint32_t HW, uint32_t C, uint32_t G, const float* gamma,
                                 const float* beta, const float* scale_shift, uint32_t scale_shift_stride, float eps, int act, const void* runs1,
                                 const void* runs2, void* y, void* stream);

/* d/dx of ssdnerf_group_norm_nhwc (single source x, no pre_bias) for frozen gamma / beta and a scale/shift that does not depend on x --
 * the gradient rendering guidance and the fine-tuning prior push through every norm of the UNet (what autograd assembles from
 * native_group_norm_backward, silu_backward and the scale/shift mul/add; modules.py:51-110).  x, dy, dx:
 * [B][HW][C] of `dtype`; fwd_sums: the forward's workspace (sum, sum of squares per sample and group); bwd_workspace:
 * ssdnerf_group_norm_backward_workspace(B, G) bytes (r06: several copies of the sums, so that the statistics pass's atomics do not queue on one
 * address), zero-filled by the call unless bwd_workspace_is_zero.  Two passes, nothing else saved.
 * act: bit 0 = the forward applied SiLU; bit 1 (r04, fp32, C % 32 == 0): write dx PRE-SPLIT (the layout of ssdnerf_group_norm_nhwc's act | 2) for the
 * backward-data convolution that consumes it (ssdnerf_conv2d_nhwc_f32x2_presplit on the transposed weights); bit 2 (r06, (C / G) % 4 == 0): fwd_sums holds the
 * forward's statistics per RUN of 4 channels, [B][C / 4][2] -- what ssdnerf_group_norm_nhwc_runs read -- instead of per group.
 * Arithmetic: csrc/gn_bwd_math.h (plain C, also built by gcc for tests/test_groupnorm_backward_cpu.py). */
size_t ssdnerf_group_norm_backward_workspace(uint32_t B, uint32_t G);

/* r06: the same for a norm over the channel concatenation [x | x2] (C1 channels from x) that ssdnerf_group_norm_nhwc normalised without building it -- the skip
 * connections of the UNet's decoder half (denoising.py:209-213 `torch.cat([h, hs.pop()], dim=1)`): the gradient leaves as two dense tensors, dx [B][HW][C1] and dx2
 * [B][HW][C - C1] (autograd's cat hands back channel slices of one tensor, which every consumer copied dense first).  x2 == NULL: the single-source call above.
 * fwd_sums2 (act & 4 only): x2's run-level statistics, fwd_sums then holds x's.  No pre-split dx (act & 2) for two sources. */
int ssdnerf_group_norm_nhwc_backward_cat(const void* x, const void* x2, uint32_t C1, const void* dy, int dtype, uint32_t B, uint32_t HW, uint32_t C, uint32_t G,
                                         const float* gamma, const float* beta, const float* scale_shift, uint32_t scale_shift_stride, float eps, int act,
                                         const void* fwd_sums, const void* fwd_sums2, void* bwd_workspace, int bwd_workspace_is_zero, void* dx, void* dx2,
                                         void* stream);
int ssdnerf_group_norm_nhwc_backward(const void* x, const void* dy, int dtype, uint32_t B, uint32_t HW, uint32_t C, uint32_t G,
                                     const float* gamma, const float* beta, const float* scale_shift, uint32_t scale_shift_stride,
                                     float eps, int act, const void* fwd_sums, void* bwd_workspace, int bwd_workspace_is_zero,
                                     void* dx, void* stream);

/* y = x + bias[c] + residual over [B][HW][C] channel-last data (bias fp32 [C] nullable, residual nullable, y may alias x):
 * the epilogue of a bias-less convolution (modules.py:51-110) or the `h + x` closing an attention block (modules.py:47).
 * gn_sums (nullable, fp64 [B][gn_groups][2], pre-zeroed) receives the GroupNorm sums of y for the norm that follows. */
int ssdnerf_bias_residual_nhwc(const void* x, int dtype, uint32_t B, uint32_t HW, uint32_t C, const float* bias,
                               const void* residual, void* y, void* gn_sums, uint32_t gn_groups, void* stream);

/* The UNet's convolutions (modules.py:51-129; denoising.py:106-187) as an implicit GEMM on the bf16 matrix cores:
 *   y[b][yo][xo][co] = sum_{kh,kw,ci} X[b][yo*stride+kh-pad][xo*stride+kw-pad][ci] * w[co][kh][kw][ci]  (+ bias[co]) (+ residual[b][yo][xo][co])
 * x bf16 [B][H][W][Cin] channel-last -- or, with x2 != NULL, the never-materialised channel concatenation of x [..][Cin1] and
 * x2 [..][Cin - Cin1] (both channel counts multiples of 8); w bf16 [Cout][ksize][ksize][Cin] (= torch channels_last weight memory); pad = ksize/2;
 * X = x, or with upsample != 0 the nearest-neighbour 2x upsampling of x, never materialised (DenoisingUpsampleMod);
 * stride 2 = DenoisingDownsampleMod.  bias fp32 [Cout] (nullable), residual / y bf16 [B][Ho][Wo][Cout] (residual nullable,
 * may alias y).  fp32 accumulation; bias and residual are added in fp32 before the single rounding to bf16.
 * gn_sums (nullable): fp64 [B][gn_groups][2], pre-zeroed by the caller; receives sum and sum of squares of the rounded output
 * per (sample, channel group) -- the statistics pass of the GroupNorm that follows (ssdnerf_group_norm_nhwc, workspace_state 2).
 * Needs (Cout / gn_groups) % 4 == 0 and, for layers that are not cut along K, Ho*Wo a multiple of the M tile (256 / 128 / 64).
 * tile_hint: 0 = choose by problem size, 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 256x128 block tile, 5 / 6 = the two-group 256x128 kernel.
 * splitk_ws (nullable): fp32 scratch of splitk_ws_bytes >= B*Ho*Wo*Cout*4 that is ALL ZERO on entry and is left all zero on
 * return; when given, layers with too few output tiles to fill the chip are cut along K (splits_hint: 0 =
 * choose, n = force n ranges) and reduced through it.
 * ssdnerf_conv2d_nhwc_bf16_supported() tells whether a layer fits (Cin % 8 == 0, Cout % 8 == 0 -- one 16-byte bf16 chunk; r02: 64 --, ksize 1|3,
 * stride 1|2).  tile_hint 5 / 6 force the two-group 256 x 128 kernel (6: its row-reuse form where it applies; Cin % 64 == 0, Cout % 128 == 0). */
int ssdnerf_conv2d_nhwc_bf16_supported(uint32_t Cin, uint32_t Cout, uint32_t ksize, uint32_t stride, uint32_t upsample);
/* The decomposition ssdnerf_conv2d_nhwc_bf16 will use for M = B*Ho*Wo output pixels: tile choice (1..3) | splits << 8. */
int ssdnerf_conv2d_nhwc_bf16_plan(uint32_t M, uint32_t Cin, uint32_t Cout, uint32_t ksize, int tile_hint, int may_split,
                                  int splits_hint);
int ssdnerf_conv2d_nhwc_bf16(const void* x, const void* x2, uint32_t Cin1, const void* w, const float* bias,
                             const void* residual, void* y, uint32_t B, uint32_t H, uint32_t W, uint32_t Cin,
                             uint32_t Cout, uint32_t ksize, uint32_t stride,
                             uint32_t upsample, void* gn_sums, uint32_t gn_groups, int tile_hint, void* splitk_ws,
                             size_t splitk_ws_bytes, int splits_hint, void* stream);

/* The same convolution for fp32 activations with fp32-class products on the bf16 matrix cores: x (and x2), residual, y are fp32
 * channel-last; the weights are passed pre-split, w_hi = bf16(w), w_lo = bf16(w - w_hi), each [Cout][ksize][ksize][Cin]; activations are
 * split in the kernel; hi*hi + hi*lo + lo*hi accumulate in fp32 (>= 16 significand bits per product; TF32, the reference's cuDNN default on
 * Ampere, keeps 11).  tile_hint 0 = choose, 1 = 128x128, 3 = 64x64; layers with too few tiles are cut along K (splits_hint 0 = choose) and
 * reduced through splitk_ws (as for the bf16 form: all zero on entry, left all zero) or, without it, straight into the output, which the call
 * zeroes first unless y_is_zero != 0 (ssdnerf_conv2d_nhwc_f32x2_plan tells a caller in advance: tile | splits << 8).  Residual must not alias
 * y.  tile_hint 5 / 6: the two-group kernel's fp32 form (needs Cin % 32 == 0, Cout % 128 == 0 and w_lo directly behind w_hi in memory).
 * Other arguments as ssdnerf_conv2d_nhwc_bf16. */
int ssdnerf_conv2d_nhwc_f32x2_plan(uint32_t M, uint32_t Cin, uint32_t Cout, uint32_t ksize, int tile_hint, int splits_hint);
int ssdnerf_conv2d_nhwc_f32x2(const void* x, const void* x2, uint32_t Cin1, const void* w_hi, const void* w_lo, const float* bias,
                              const void* residual, void* y, uint32_t B, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout,
                              uint32_t ksize, uint32_t stride, uint32_t upsample, void* gn_sums, uint32_t gn_groups, int tile_hint,
                              int splits_hint, int y_is_zero, void* splitk_ws, size_t splitk_ws_bytes, void* stream);

/* r04: the fp32 configs' convolutions on PRE-SPLIT activations.  ssdnerf_group_norm_nhwc / _runs with bit 1 of `act` set (act | 2; fp32, C % 32 == 0)
 * write their result not as fp32 but as its bf16 pair split in the layout the convolution kernels' K-tiles take -- per pixel and block of 32 channels
 * 128 bytes = [32 hi terms | 32 lo terms], hi = truncation to bf16, lo = truncation of the exact remainder: the same bytes per element and the same
 * arithmetic as the split ssdnerf_conv2d_nhwc_f32x2 does on the fly -- and ssdnerf_conv2d_nhwc_f32x2_presplit convolves such a tensor (ksize 1 | 3,
 * stride 1, pad ksize / 2; w_lo directly behind w_hi; bias / residual / y / gn_sums / splitk_ws as in ssdnerf_conv2d_nhwc_f32x2).
 * _supported: 1 = the two-group row kernel takes the layer (large 3 x 3 layers; bit-identical to the on-the-fly split), 2 = the generic kernel's PS form
 * does (any other stride-1 layer with Cin % 32 == 0, Cout % 64 == 0; same products, another summation order), 0 = neither: ask the producing norm for the
 * split output only when it is non-zero.  tile_hint 1 / 2 / 3 (128 x 128, 64 x 128, 64 x 64) and splits_hint > 0 force the generic form's plan (sweeps); 0 = auto. */
int ssdnerf_conv2d_nhwc_f32x2_presplit_supported(uint32_t B, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t ksize, int with_gn_sums);
/* x fp32 [pixels][C] (C % 32 == 0) -> y, the same bytes in the pre-split layout: for an operand no norm produced (unet._ConvF32x2Fn.backward).
 * x_stride: floats from pixel to pixel in the source (0 = C: dense; > C: a channel slice of a wider channel-last tensor, read in place). */
int ssdnerf_split_f32_nhwc(const void* x, void* y, uint64_t pixels, uint32_t C, uint64_t x_stride, void* stream);
int ssdnerf_conv2d_nhwc_f32x2_presplit(const void* x_split, const void* w_hi, const void* w_lo, const float* bias, const void* residual, void* y, uint32_t B,
                                       uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t ksize, void* gn_sums, uint32_t gn_groups,
                                       int tile_hint, int splits_hint, void* splitk_ws, size_t splitk_ws_bytes, void* stream);

/* Self-attention of MultiHeadAttentionMod (modules.py:12-48; mmgen QKVAttention) over the qkv projection of a channel-last
 * activation: qkv bf16 [B][T][3*heads*ch] with the reference's channel order [head][q | k | v][ch], out bf16 [B][T][heads*ch]
 * (channel = head*ch + i):  out = softmax(q k^T / sqrt(ch)) v  per (sample, head), softmax statistics and accumulation in fp32,
 * probabilities rounded to bf16 before the PV product (as the reference's `.type(weight.dtype)`).  Any T >= 1; ch a multiple of 8 in
 * [8, 128] (keys / channels past the tensor are masked). */
int ssdnerf_attention_qkv_bf16(const void* qkv, void* out, uint32_t B, uint32_t T, uint32_t heads, uint32_t ch, void* stream);
/* The same for the fp32 configs (the UNet runs without autocast in every paper config but the fp16 one): qkv, out fp32; q, k, v and the
 * probabilities are each split into a bf16 pair and every product is hi*hi + hi*lo + lo*hi in fp32 on the matrix cores (the arithmetic
 * class of ssdnerf_conv2d_nhwc_f32x2), softmax in fp32.
 * TOLERANCE CLASS, stated because the name says f32: this is NOT IEEE fp32 arithmetic.  Each factor keeps 16 significand bits (the dropped
 * lo*lo term and the split's own remainder are both 2^-16 relative), so a logit or an output element carries a relative error of order 1e-5,
 * two decimal digits better than TF32 and two worse than fp32; the UNet's output differs from the fp32 module by 7e-6 relative at the
 * benchmarked shape (bench.py, ddim.fp32.rel_err_vs_eager).  Used by the inference executor and -- since r03, through the _lse / _backward entry
 * points below -- by the gradient path (guidance, fine-tuning) as well: the reference computes those gradients in IEEE fp32, this path in the
 * class above (tests: test_attention_backward_kernels_match_autograd, 5e-5 vs fp64 autograd).  SSDNERF_UNET_GRAD_ATT_KERNEL=0 keeps the library's
 * true-fp32 scaled_dot_product_attention for the gradient path (tests/test_recons_gpu.py covers both), and B * heads > 65 535 falls back to it. */
int ssdnerf_attention_qkv_f32(const void* qkv, void* out, uint32_t B, uint32_t T, uint32_t heads, uint32_t ch, void* stream);
/* Forward that also saves what the backward needs, and the backward itself (r03: the gradient path of rendering guidance / fine-tuning, which r02 ran on
 * the library's fp32 attention): lse2 fp32 [B][heads][T] = the rows' log-sum-exp in the log2 domain; dout fp32 [B][T][heads*ch] = d loss / d out;
 * dqkv fp32 [B][T][3*heads*ch] receives d loss / d qkv in qkv's layout; workspace: B*heads*T floats (D = rowsum(dout o out)).  Same arithmetic class
 * as the forward (see TOLERANCE CLASS above): S and dP are recomputed per 32 x 32 tile, nothing T x T is stored. */
int ssdnerf_attention_qkv_f32_lse(const void* qkv, void* out, void* lse2, uint32_t B, uint32_t T, uint32_t heads, uint32_t ch, void* stream);
int ssdnerf_attention_qkv_f32_backward(const void* qkv, const void* out, const void* dout, const void* lse2, void* dqkv, void* workspace, uint32_t B,
                                       uint32_t T, uint32_t heads, uint32_t ch, void* stream);

/* ---- iso-surface extraction (replaces the PyMCubes call of lib/core/utils/nerf_utils.py:103-105) -----------------
 * Marching cubes over volume fp32 [nx][ny][nz] (x-major, PyMCubes' indexing) at `iso` in two kernels around two prefix sums the caller runs:
 *   _count : per lattice point p = (x*ny + y)*nz + z:  point_mask[p] = which of its owned edges (+x: bit 0, +y: bit 1, +z: bit 2) cross the
 *            iso-value, point_verts[p] = their number, cell_tris[p] = triangles of the cell whose minimum corner is p (0 on the last layers).
 *            tri_count: uint8 [256] triangles per corner-sign case (corner i INSIDE when its value > iso; ssdnerf_amd/mesh.py triangle_table).
 *   _emit  : tri_offsets / vert_offsets = INCLUSIVE prefix sums of cell_tris / point_verts; tri_edges int8 [256][15] = cube-edge ids of every
 *            case's triangles.  Writes vertices fp32 [V][3] in index coordinates (one per crossing lattice edge, linearly interpolated:
 *            coordinate + (iso - a) / (b - a) along the edge) and triangles int32 [T][3] (indices into vertices; counter-clockwise seen from the
 *            low-density side).  The mesh is watertight by construction of the table. */
int ssdnerf_marching_cubes_count(const float* volume, uint32_t nx, uint32_t ny, uint32_t nz, float iso, const uint8_t* tri_count,
                                 int32_t* cell_tris, int32_t* point_verts, uint8_t* point_mask, void* stream);
int ssdnerf_marching_cubes_emit(const float* volume, uint32_t nx, uint32_t ny, uint32_t nz, float iso, const uint8_t* tri_count,
                                const int8_t* tri_edges, const int32_t* tri_offsets, const int32_t* vert_offsets, const uint8_t* point_mask,
                                float* vertices, int32_t* triangles, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSDNERF_HIP_H */
