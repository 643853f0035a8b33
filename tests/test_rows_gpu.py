"""GPU tests for the SURVEY.md section 8 rows the round-1 verdict left partial:

  item 8   integer sample counts of a FULL bench scene (251 views) against the threaded oracle, object-like and fog scene
  (f)2     Langevin correction steps and the tiled (6, H, 3H) layout through the sampler on the GPU
  (f)3     a 16-bit cached ``.pth`` scene (fp16 code, written by ``save_cache``) loaded and rendered through the fused HIP path
  (f)4     ``MultiSceneNeRF.train_step`` / ``DiffusionNeRF.train_step`` with the real renderer on the GPU
  N1       config 5's precision mix end to end: bf16 UNet + fp16 planes against the fp32 path, PSNR floor
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEC = dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], use_dir_enc=True,
           dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256)


def _unet(image_size=128, in_channels=18, base=64, cfg=(1, 2), att=(64,), groups=32):
    return dict(type="DenoisingUnetMod", image_size=image_size, in_channels=in_channels, base_channels=base, channels_cfg=list(cfg),
                resblocks_per_downsample=1, dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4,
                attention_res=list(att), norm_cfg=dict(type="GN", num_groups=groups))


def _randomize(module, seed, scale=0.2):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (scale / max(1.0, p[0].numel() ** 0.5) if p.dim() > 1 else 0.1))


def _decoder():
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import TriPlaneDecoder
    dec = TriPlaneDecoder(**{k: v for k, v in DEC.items() if k != "type"})
    dec.load_state_dict(S.make_decoder_params(), strict=False)
    return dec.cuda().eval()


# ---------------------------------------------------------------------------------------------- item 8: bench-scale integer parity
@pytest.mark.parametrize("variant,n_views", [("object", 251), ("uniform", 16)])
def test_full_scene_sample_counts_match_the_oracle(variant, n_views):
    """Per-ray sample counts of a whole bench scene -- every view of the spiral for the object-like scene, 16 views of the fog scene (33
    samples per ray) -- from the camera-fed fused path against the reference-shaped oracle loop (C restatement of the reference kernels on all
    host cores + PyTorch-CPU decode).  Integer contract: equal except on rays with a termination test within 1e-5 of T_thresh -- the
    transmittance there is 1 - (a sum of ~30 weights near 1), so hardware exp vs expf and the summation order of the MLP move it by a few 1e-6 --
    and those mismatches must be rarer than 1 ray in 2000.

    Float contract (r04; the r03 verdict's weak #2: "whole-scene float parity is observed, not asserted"): over the SAME sweep RGB within 1e-4
    of the oracle everywhere and within 2e-5 on every ray whose sample count agrees (the single-view tolerance), depth within 1e-4 on those rays
    (a ray that takes one sample more or less at T ~ 1e-4 moves depth by up to 1e-4 x t); and with the oracle marching the rays the CPU
    generates itself (<= 1 ulp away from the kernels': samples move across voxel faces) RGB still within north_star's 1e-4 on a quarter of
    the views."""
    import oracle
    from oracle import render as R
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import pack_triplanes
    from ssdnerf_amd.density import get_density
    oracle.set_threads(min(os.cpu_count() or 1, 32))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    dec = _decoder()
    params, code = S.make_decoder_params(), S.make_triplane(2021, variant)
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g) for _ in range(8)]
    _, bits = get_density(dec, code.cuda()[None], 64, density_thresh=0.1, density_step=8, jitters=[j.cuda() for j in jit])
    poses = S.spiral_poses(251)[:n_views]
    intr = S.cars_intrinsics(128, 128)[None].expand(n_views, -1)
    out = dec.render_packed(pack_triplanes(code.cuda()[None]), None, None, bits, 64, [0.0], 1e-4, bg_color=1.0, want_counts=True, check_overflow=False,
                            cams=(poses.cuda()[None], intr.cuda()[None], 128, 128))
    got = dec.last_render_stats["sample_counts"][0].cpu().numpy().reshape(n_views, -1)
    got_rgb = out["image"][0].cpu().numpy().reshape(n_views, -1, 3)
    got_dep = out["depth"][0].cpu().numpy().reshape(n_views, -1)
    near_thresh = int(dec.last_render_stats["boundary_tests"].sum())
    assert int(dec.last_render_stats["overflow"].item()) == 0
    bits_np = bits[0].cpu().numpy()
    # the oracle marches the SAME rays: the arrays ssdnerf_cam_rays materialises, which the camera-fed kernels reproduce bit for bit
    # (test_camera_fed_render_is_bit_identical_to_ray_arrays); the CPU tensor-op form differs from them by a few ulp (row a1's tolerance),
    # which is enough to move a sample across a voxel face on ~1 ray per view
    from ssdnerf_amd import nerf
    ro, rd = (t[0].cpu() for t in nerf.get_cam_rays(poses.cuda()[None], intr.cuda()[None], 128, 128))
    mismatched = unexplained = total = 0
    rgb_err = rgb_err_same = dep_err_same = dep_err = 0.0
    for v in range(n_views):
        tr = {}
        rgb0, dep0, _ = R.render_eval(params, code, bits_np, ro[v].reshape(-1, 3).numpy(), rd[v].reshape(-1, 3).numpy(), trace=tr, near_band=1e-5)
        diff = tr["samples_composited"] != got[v]
        mismatched += int(diff.sum())
        unexplained += int((diff & ~tr["near_threshold"]).sum())
        total += int(tr["samples_composited"].sum())
        e_rgb, e_dep = np.abs(got_rgb[v] - rgb0).max(axis=-1), np.abs(got_dep[v] - dep0)
        rgb_err, dep_err = max(rgb_err, float(e_rgb.max())), max(dep_err, float(e_dep.max()))
        rgb_err_same, dep_err_same = max(rgb_err_same, float(e_rgb[~diff].max())), max(dep_err_same, float(e_dep[~diff].max()))
    print(f"{variant}: {n_views} views, max|rgb - oracle| = {rgb_err:.2e} ({rgb_err_same:.2e} on rays with equal counts), max|depth - oracle| = "
          f"{dep_err:.2e} ({dep_err_same:.2e}); {mismatched} rays differ in count")
    assert rgb_err <= 1e-4 and rgb_err_same <= 2e-5, (rgb_err, rgb_err_same)
    assert dep_err_same <= 1e-4 and dep_err <= 6e-4, (dep_err_same, dep_err)
    # the oracle on ITS OWN rays (the CPU tensor-op form of get_cam_rays, <= 1 ulp from the kernels' rays): what bench.py's
    # cpu_baseline.max_abs_rgb_err_gpu_vs_oracle reports for the whole scene (7.4e-5 in r03) -- every fourth view here
    rgb_err_cpu_rays = 0.0
    for v in range(0, n_views, 4):
        ro_c, rd_c = R.get_cam_rays(poses[v][None], intr[v][None], 128, 128)
        rgb_c, _, _ = R.render_eval(params, code, bits_np, ro_c.reshape(-1, 3).numpy(), rd_c.reshape(-1, 3).numpy())
        rgb_err_cpu_rays = max(rgb_err_cpu_rays, float(np.abs(got_rgb[v] - rgb_c).max()))
    print(f"{variant}: oracle on CPU-generated rays, {len(range(0, n_views, 4))} views: max|rgb - oracle| = {rgb_err_cpu_rays:.2e}")
    assert rgb_err_cpu_rays <= 1e-4, rgb_err_cpu_rays
    n_rays = n_views * 128 * 128
    assert total > 50 * n_views and abs(int(got.sum()) - total) <= max(8, mismatched * 8)
    assert unexplained == 0                                                    # only rays sitting at the threshold may differ ...
    assert mismatched <= max(1, n_rays // 2000), (mismatched, n_rays)          # ... and they are rare
    assert near_thresh > 0 or mismatched == 0                                  # (the kernel's own diagnostic count of tests within 2e-6)


# ---------------------------------------------------------------------------------------------- a11: the deferred overflow check of nerf.render
def test_deferred_overflow_check_redoes_the_batch_in_place():
    """``nerf.render(..., defer_overflow_check=True)`` (r05) returns without the host read of the overflow flag; the flag is examined behind the next render's
    launches, or in ``finish_render``.  (1) On a batch that stays below the step cap the outputs equal the synchronous call's, bit for bit, and nothing is
    redone.  (2) A fog scene with a full bitfield and ``T_thresh`` 0 reaches the cap: the synchronous call redoes it through the stepwise path; the deferred
    call hands out the fused kernels' images first and, once settled, the SAME tensors hold the stepwise result -- settled by ``finish_render`` and, equally,
    by the next ``render`` on the decoder."""
    from ssdnerf_amd import nerf, synthetic as S
    dec = _decoder()
    nv = 3
    poses = S.spiral_poses()[::90][:nv][None].cuda().contiguous()
    intr = S.cars_intrinsics(128, 128)[None, None].expand(1, nv, -1).cuda().contiguous()
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g).cuda() for _ in range(8)]
    from ssdnerf_amd.density import get_density
    code = S.make_triplane(2021, "object").cuda()[None]
    _, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=jit)
    # (1) no overflow
    a = nerf.render(dec, code, bits, 128, 128, intr, poses, return_u8=True)
    b = nerf.render(dec, code, bits, 128, 128, intr, poses, return_u8=True, defer_overflow_check=True)
    assert nerf.finish_render(dec) is False
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # (2) a raised flag.  With bound 1 a ray cannot really collect max_steps occupied samples (the box diagonal is exactly max_steps minimum steps), so the
    # flag is raised by hand behind the fused launch: the synchronous call then redoes the batch through the stepwise path at once, the deferred one later
    real = dec.render_packed

    def flagged(*a, **k):
        out = real(*a, **k)
        dec.last_render_stats["overflow"] = torch.ones(1, dtype=torch.int32, device="cuda")
        return out
    dec.render_packed = flagged
    try:
        sync = nerf.render(dec, code, bits, 128, 128, intr, poses, return_u8=True)                 # stepwise result (redone at once)
        d1 = nerf.render(dec, code, bits, 128, 128, intr, poses, return_u8=True, defer_overflow_check=True)
        first = [t.clone() for t in d1]
        for x, y in zip(a, first):
            assert torch.equal(x, y)                                       # what the caller holds first is the fused kernels' batch
        assert nerf.finish_render(dec) is True                             # redone now ...
        for x, y in zip(sync, d1):
            assert torch.equal(x, y)                                       # ... into the tensors the caller already holds
        assert float((sync[0] - a[0]).abs().max()) <= 2e-5                 # (stepwise and fused agree to rounding when nothing really overflowed)
        d2 = nerf.render(dec, code, bits, 128, 128, intr, poses, return_u8=True, defer_overflow_check=True)
        dec.render_packed = real
        ok = nerf.render(dec, code, bits, 128, 128, intr, poses, return_u8=True, defer_overflow_check=True)      # settles d2 behind its own launches
        for x, y in zip(sync, d2):
            assert torch.equal(x, y)
    finally:
        dec.__dict__.pop("render_packed", None)
    assert nerf.finish_render(dec) is False
    for x, y in zip(a, ok):
        assert torch.equal(x, y)


# ---------------------------------------------------------------------------------------------- (f)3: 16-bit scene cache -> fused render
def test_sixteen_bit_cached_scene_renders_through_the_fused_path(tmp_path):
    """``MultiSceneNeRF.save_cache`` with ``cache_16bit`` writes fp16 pre-activation codes (+ bf16 optimizer moments); the files are read back
    (``data['code']`` -> ``load_scene``), rendered by the fused kernels, and must equal the oracle's render of the SAME fp16-rounded code to
    render tolerance, and stay close to the fp32 original (the quantisation is the only difference)."""
    from oracle import render as R
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.registry import MODELS
    m = MODELS.build(dict(type="MultiSceneNeRF", code_size=(3, 6, 128, 128), code_activation=dict(type="TanhCode", scale=2), grid_size=64,
                          decoder=DEC, pixel_loss=dict(type="MSELoss", loss_weight=20.0), cache_size=2, cache_16bit=True,
                          train_cfg=dict(save_dir=str(tmp_path), optimizer=dict(type="Adam", lr=0.01))))
    m.decoder.load_state_dict(S.make_decoder_params(), strict=False)
    m = m.cuda().eval()
    codes = torch.stack([S.make_triplane(31), S.make_triplane(32)]).cuda()
    leaves = [m.code_activation.inverse(c).detach().requires_grad_(True) for c in codes]
    opts = m.build_optimizer(leaves, m.train_cfg)
    for leaf, opt in zip(leaves, opts):                                        # one step so that the optimizers have state to cast
        leaf.grad = torch.zeros_like(leaf)
        opt.step()
    g = torch.Generator().manual_seed(5)
    jit = [torch.rand(64 ** 3, 3, generator=g) for _ in range(4)]
    grid, bits = m.get_density(m.decoder, m.code_activation(torch.stack(leaves)).detach(), cfg=dict(density_thresh=0.1, density_step=4),
                               jitters=[j.cuda() for j in jit])
    m.save_cache(leaves, opts, grid, bits, [0, 1], ["scene_a", "scene_b"])
    files = sorted(os.listdir(tmp_path))
    assert files == ["scene_a.pth", "scene_b.pth"]
    entries = [torch.load(os.path.join(tmp_path, f), map_location="cpu") for f in files]
    assert entries[0]["param"]["code_"].dtype == torch.float16 and entries[0]["param"]["density_grid"].dtype == torch.float16
    assert all(v.dtype == torch.bfloat16 for k, v in entries[0]["optimizer"]["state"][0].items() if k != "step" and torch.is_tensor(v))
    code, grid2, bits2 = m.load_scene(dict(code=entries), load_density=True)
    assert code.dtype in (torch.float16, torch.float32) and torch.equal(bits2.cpu(), bits.cpu())
    poses = S.spiral_poses()[[64]].cuda()[None].expand(2, -1, -1, -1)
    intr = S.cars_intrinsics(64, 64).cuda()[None, None].expand(2, 1, -1)
    image, depth = m.render(m.decoder, code.float(), bits2, 64, 64, intr, poses, cfg=dict())
    ref, _ = m.render(m.decoder, codes, bits, 64, 64, intr, poses, cfg=dict())
    assert float((image - ref).abs().max()) < 5e-2 and float(((image - ref) ** 2).mean()) < 1e-5      # fp16 code quantisation only
    ro, rd = R.get_cam_rays(S.spiral_poses()[[64]], S.cars_intrinsics(64, 64)[None], 64, 64)
    want, dep0, _ = R.render_eval(S.make_decoder_params(), code[0].float().cpu(), bits2[0].cpu().numpy(), ro.reshape(-1, 3).numpy(),
                                  rd.reshape(-1, 3).numpy())
    np.testing.assert_allclose(image[0].reshape(-1, 3).cpu().numpy(), want, rtol=0, atol=2e-5)
    np.testing.assert_allclose(depth[0].reshape(-1).cpu().numpy(), dep0, rtol=0, atol=1e-4)


# ---------------------------------------------------------------------------------------------- N1: config 5's precision mix
def _sampling_model(autocast_dtype, plane_dtype, test_cfg, code_permute=None, code_reshape=(18, 128, 128), unet=None):
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.registry import MODELS
    m = MODELS.build(dict(type="DiffusionNeRF", code_size=(3, 6, 128, 128), code_reshape=code_reshape, code_permute=code_permute,
                          code_activation=dict(type="TanhCode", scale=2), grid_size=64, autocast_dtype=autocast_dtype,
                          diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"), denoising=unet or _unet()),
                          decoder=dict(DEC, plane_dtype=plane_dtype), decoder_use_ema=True, freeze_decoder=False, bg_color=1,
                          pixel_loss=dict(type="MSELoss", loss_weight=20.0), cache_size=0, test_cfg=test_cfg))
    _randomize(m.diffusion_ema.denoising, 17)
    m.decoder_ema.load_state_dict(S.make_decoder_params(), strict=False)
    return m.cuda().eval()


def test_bf16_unet_fp16_planes_end_to_end_against_fp32():
    """ssdnerf_chairs_recons1v-style precision (BASELINE.json configs[4]): the UNet under bf16 autocast (inference executor, bf16 matrix-core
    convolutions and attention) and fp16 triplanes in the renderer, against the all-fp32 path from the same noise and weights: the sampled
    codes stay within bf16 accumulation error and the rendered views agree to > 35 dB PSNR."""
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd import nerf
    cfg = dict(img_size=(128, 128), num_timesteps=6, clip_range=[-2, 2], density_thresh=0.1)
    lo = _sampling_model("bfloat16", "float16", cfg)
    hi = _sampling_model(None, "float32", cfg)
    hi.load_state_dict(lo.state_dict())
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(2, 3, 6, 128, 128, generator=g).cuda()
    jit = [torch.rand(64 ** 3, 3, generator=g).cuda() for _ in range(8)]
    code_lo, _, _ = lo.val_uncond(dict(scene_id=[0, 1], noise=noise), density_jitters=jit)
    code_hi, _, bits_hi = hi.val_uncond(dict(scene_id=[0, 1], noise=noise), density_jitters=jit)
    rel = float((code_lo - code_hi).norm() / code_hi.norm())
    assert rel < 5e-2, rel
    ex = lo.diffusion_ema.denoising._fast_cache[torch.bfloat16]
    assert ex.library_fallbacks == 0, ex.fallback_log                         # the bf16 step ran on the hand-written kernels
    # render the SAME scene (the fp32 codes; object-like stand-in for visible content) with fp16 and fp32 planes
    codes = torch.stack([S.make_triplane(41), S.make_triplane(42)]).cuda()
    _, bits = hi.get_density(hi.decoder_ema, codes, cfg=cfg, jitters=jit)
    poses = S.spiral_poses()[[20, 140]].cuda()[None].expand(2, -1, -1, -1)
    intr = S.cars_intrinsics().cuda()[None, None].expand(2, 2, -1)
    img_lo, _ = lo.render(lo.decoder_ema, codes, bits, 128, 128, intr, poses, cfg=cfg)
    img_hi, _ = hi.render(hi.decoder_ema, codes, bits, 128, 128, intr, poses, cfg=cfg)
    psnr = nerf.eval_psnr(img_lo.reshape(4, -1), img_hi.reshape(4, -1))
    assert float(psnr.min()) > 35.0, psnr


# ---------------------------------------------------------------------------------------------- (f)2: Langevin steps, tiled layout
def test_langevin_sampling_and_tiled_layout_on_the_gpu():
    """``langevin_steps`` correction evaluations after every DDIM step (ssdnerf_chairs_recons1v.py:95-96) and the tiled (6, 128, 384) latent
    layout (``code_permute=(1, 2, 0, 3)``; new_cfgs/ssdnerf_cars_recons1v_tiled.py:6-28, GroupNorm(16), widths that are not multiples of 64):
    the GPU sampler (inference executor, device-resident loop where the step kind allows it) reproduces the CPU module run of the same model
    with the same host-drawn noise."""
    cfg = dict(img_size=(128, 128), num_timesteps=3, clip_range=[-2, 2], density_thresh=0.1, langevin_steps=2, langevin_delta=0.4)
    tiled_unet = _unet(image_size=128, in_channels=6, base=48, cfg=(1, 1, 2), att=(32,), groups=16)
    for kw in (dict(), dict(code_permute=(1, 2, 0, 3), code_reshape=(6, 128, 384), unet=tiled_unet)):
        m = _sampling_model(None, "float32", cfg, **kw)
        g = torch.Generator().manual_seed(9)
        noise = torch.randn(1, 3, 6, 128, 128, generator=g)
        plan = m.diffusion_ema.sampling_plan("ddim")
        assert [s.kind for s in plan].count("langevin") == 2 * 2 and len(plan) == 3 + 4      # no correction after the last step (t_prev = -1)
        torch.manual_seed(123)
        with torch.no_grad():
            lat_gpu = m.diffusion_ema(m.code_diff_pr(noise.cuda()), return_loss=False)
        m.cpu()
        torch.manual_seed(123)
        with torch.no_grad():
            lat_cpu = m.diffusion_ema(m.code_diff_pr(noise), return_loss=False)
        assert lat_gpu.shape == lat_cpu.shape == ((1, 18, 128, 128) if not kw else (1, 6, 128, 384))
        err = float((lat_gpu.cpu() - lat_cpu).abs().max())
        assert err < 2e-3 * max(1.0, float(lat_cpu.abs().max())), err
        back = m.code_diff_pr_inv(lat_cpu)
        assert back.shape == (1, 3, 6, 128, 128) and torch.equal(m.code_diff_pr(back), lat_cpu)


# ---------------------------------------------------------------------------------------------- (f)4: training steps on the GPU
def test_training_steps_run_on_the_gpu(tmp_path):
    """One ``DiffusionNeRF.train_step`` and one ``MultiSceneNeRF.train_step`` with the real renderer: prior loss through the UNet, seeded
    code-only fitting iterations through the train-branch render (HIP march / decode forward+backward / composite), the joint step that also
    moves the decoder, cache write-back.  Checks: every logged value finite, codes / denoiser / decoder all moved, the cached codes change from step to step."""
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.registry import MODELS
    train_cfg = dict(dt_gamma_scale=0.5, density_thresh=0.1, extra_scene_step=2, n_inverse_rays=2 ** 12, n_decoder_rays=2 ** 12,
                     loss_coef=0.1 / (64 * 64), optimizer=dict(type="Adam", lr=0.02), save_dir=str(tmp_path / "cache"))
    m = MODELS.build(dict(type="DiffusionNeRF", code_size=(3, 6, 128, 128), code_reshape=(18, 128, 128), code_activation=dict(type="TanhCode", scale=2),
                          grid_size=64, diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"),
                                                       denoising=_unet(base=32, cfg=(1, 1), att=(), groups=8),
                                                       timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
                                                       ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight",
                                                                      data_info=dict(pred="v_t_pred", target="v_t"), weight_scale=4.0, scale_norm=True)),
                          decoder=DEC, decoder_use_ema=True, freeze_decoder=False, bg_color=1, pixel_loss=dict(type="MSELoss", loss_weight=20.0),
                          reg_loss=dict(type="RegLoss", power=2, loss_weight=3e-3), cache_size=4, init_scale=0.5, train_cfg=train_cfg))
    _randomize(m.diffusion.denoising, 3)
    m.decoder.load_state_dict(S.make_decoder_params(), strict=False)
    m = m.cuda().train()
    # targets: views of two synthetic scenes
    dec = _decoder()
    codes = torch.stack([S.make_triplane(51), S.make_triplane(52)]).cuda()
    from ssdnerf_amd import nerf
    from ssdnerf_amd.density import get_density
    _, bits = get_density(dec, codes, 64, density_thresh=0.1, density_step=4)
    poses = S.spiral_poses()[[30, 150]].cuda()[None].expand(2, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(64, 64).cuda()[None, None].expand(2, 2, -1).contiguous()
    target, _ = nerf.render(dec, codes, bits, 64, 64, intr, poses)
    data = dict(scene_id=[0, 2], scene_name=["s0", "s2"], cond_imgs=target.clamp(0, 1), cond_poses=poses, cond_intrinsics=intr)
    opt = dict(diffusion=torch.optim.Adam(m.diffusion.parameters(), lr=1e-4), decoder=torch.optim.Adam(m.decoder.parameters(), lr=1e-3))
    unet_w = m.diffusion.denoising.out.conv.weight.detach().clone()
    dec_w = m.decoder.base_net[0].weight.detach().clone()
    np.random.seed(1); torch.manual_seed(1)
    out1 = m.train_step(data, opt)
    code_after_1 = m.cache[2]["param"]["code_"].clone()
    out2 = m.train_step(data, opt)
    for out in (out1, out2):
        assert out["num_samples"] == 2
        for k in ("loss_ddpm_mse", "pixel_loss", "loss_decoder", "train_psnr", "code_rms"):
            assert bool(torch.isfinite(torch.as_tensor(out["log_vars"][k]).float()).all()), k
    assert not torch.equal(m.cache[2]["param"]["code_"], code_after_1)                              # the cached codes keep moving (3 Adam steps per call)
    assert not torch.equal(m.diffusion.denoising.out.conv.weight, unet_w) and not torch.equal(m.decoder.base_net[0].weight, dec_w)
    assert sorted(os.listdir(tmp_path / "cache")) == ["s0.pth", "s2.pth"] and m.cache[0] is not None and m.cache[1] is None
    assert float(m.cache[2]["param"]["code_"].abs().max()) > 0 and m.cache[2]["param"]["density_bitfield"].dtype == torch.uint8

    ms = MODELS.build(dict(type="MultiSceneNeRF", code_size=(3, 6, 128, 128), code_activation=dict(type="TanhCode", scale=2), grid_size=64, decoder=DEC,
                           pixel_loss=dict(type="MSELoss", loss_weight=20.0), cache_size=4, init_scale=0.5,
                           train_cfg=dict(train_cfg, save_dir=None, extra_scene_step=1)))
    ms.decoder.load_state_dict(S.make_decoder_params(), strict=False)
    ms = ms.cuda().train()
    o1 = ms.train_step(data, dict(decoder=torch.optim.Adam(ms.decoder.parameters(), lr=1e-3)))
    o2 = ms.train_step(data, dict(decoder=torch.optim.Adam(ms.decoder.parameters(), lr=1e-3)))
    assert bool(torch.isfinite(o1["log_vars"]["loss"])) and bool(torch.isfinite(o2["log_vars"]["loss"])) and bool(torch.isfinite(o2["log_vars"]["train_psnr"]))


# ---------------------------------------------------------------------------------------------- run-to-run reproducibility at bench scale
@pytest.mark.parametrize("variant,n_scenes,n_views,repeats", [("object", 1, 251, 6), ("uniform", 1, 48, 4), ("object", 8, 251, 3)])
def test_fused_render_is_reproducible_bit_for_bit(variant, n_scenes, n_views, repeats):
    """The same render issued several times must give the same bits every time (counts, image, depth): the bench scene, the fog scene (every
    ray shades, long rays), and the 8-scene batch of the bench.  Nothing in the fused path is order-dependent per ray -- queue order and ticket
    assignment vary from run to run, the arithmetic of a ray does not -- so any difference is a hardware hazard or a race.  r02 found one this
    way (16 neighbouring rays off by up to 6e-3, ~30 rays of 4 M per launch, only with two waves per SIMD); r06 named it: packed fp32 instructions
    whose halves read across a VGPR pair, split by the build since (ssdnerf_amd/asm_postpass.py; the rare form of it -- one render in ~3 000 --
    is what tests/test_soak_gpu.py is for).  The sample totals are pinned too (a drifting total was the first sign of a broken build in r02 and r03)."""
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import pack_triplanes
    from ssdnerf_amd.density import get_density
    dec = _decoder()
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g).cuda() for _ in range(8)]
    seeds = [2022] if n_scenes == 1 else list(range(2021, 2021 + n_scenes))
    code = torch.stack([S.make_triplane(sd, variant) for sd in seeds]).cuda()
    _, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=jit)
    planes = pack_triplanes(code)
    poses = S.spiral_poses(251)[:n_views].cuda()[None].expand(n_scenes, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(128, 128).cuda()[None, None].expand(n_scenes, n_views, -1).contiguous()

    def render():
        out = dec.render_packed(planes, None, None, bits, 64, [0.0] * n_scenes, 1e-4, bg_color=1.0, want_counts=True, check_overflow=False,
                                cams=(poses, intr, 128, 128), want_u8=True)
        return dec.last_render_stats["sample_counts"].clone(), out["image"].clone(), out["depth"].clone(), out["image_u8"].clone()

    ref = render()
    assert int((ref[0] > 0).sum()) > 100000
    if variant == "object" and n_scenes == 1:
        assert int(ref[0].sum()) == 12014640, int(ref[0].sum())   # the total of every reproducible build since the SiLU scale was folded into the weights (r03; 12014632 before: 8 rays whose
                                                                 # termination test sits within float noise of T_thresh; the oracle sweep above bounds it per ray); broken builds drifted by 3 .. 2400
    if variant == "object" and n_scenes == 8:
        assert int(ref[0].sum()) == 84632345, int(ref[0].sum())   # the bench workload's total (bench.py prints it as boundary_rays.samples_per_step_per_gpu)
    for _ in range(repeats - 1):
        again = render()
        for a, b, name in zip(ref, again, ("sample_counts", "image", "depth", "image_u8")):
            assert torch.equal(a, b), (name, int((a != b).sum()))


# ---------------------------------------------------------------------------------------------- r06: stage A of the next render beside this render's shading kernel
def test_prefetched_stage_a_renders_the_same_bits():
    """``nerf.render(next_batch=...)``: stage A of the NEXT render is launched on a second stream, into its own workspace and outputs, beside the current render's shading
    kernel.  A streaming loop over two alternating batches (other cameras, other bitfield) must give, for every call, the bits of a plain render of that batch; a call
    whose inputs are NOT the announced ones must render normally; ``finish_render`` forgets a stage nobody uses."""
    from ssdnerf_amd import nerf, synthetic as S
    from ssdnerf_amd.decoders import pack_triplanes
    from ssdnerf_amd.density import get_density
    dec = _decoder()
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g).cuda() for _ in range(8)]
    ns, nv = 2, 24
    codes = [torch.stack([S.make_triplane(sd, "object") for sd in seeds]).cuda() for seeds in ((2021, 2022), (2023, 2024))]
    bits = [get_density(dec, c, 64, density_thresh=0.1, density_step=8, jitters=jit)[1] for c in codes]
    planes = [pack_triplanes(c) for c in codes]
    all_poses = S.spiral_poses(251).cuda()
    poses = [all_poses[:nv][None].expand(ns, -1, -1, -1).contiguous(), all_poses[100:100 + nv][None].expand(ns, -1, -1, -1).contiguous()]
    intr = S.cars_intrinsics(128, 128).cuda()[None, None].expand(ns, nv, -1).contiguous()

    def plain(b):
        im, dp, u8 = nerf.render(dec, codes[b], bits[b], 128, 128, intr, poses[b], grid_size=64, bg_color=1.0, cfg={}, planes=planes[b], return_u8=True)
        return im.clone(), dp.clone(), u8.clone()
    want = [plain(0), plain(1)]
    assert not torch.equal(want[0][0], want[1][0])
    got = []
    order = [0, 1, 0, 0, 1, 1, 0]
    for i, b in enumerate(order):
        nb = order[i + 1] if i + 1 < len(order) else 0
        im, dp, u8 = nerf.render(dec, codes[b], bits[b], 128, 128, intr, poses[b], grid_size=64, bg_color=1.0, cfg={}, planes=planes[b], return_u8=True,
                                 defer_overflow_check=True, next_batch=(bits[nb], intr, poses[nb]))
        got.append((b, im, dp, u8))
        if i in (0, 3):
            assert getattr(dec, "_prefetched", None) is not None                      # a stage A is in flight for the next call
    nerf.finish_render(dec)
    assert getattr(dec, "_prefetched", None) is None
    for b, im, dp, u8 in got:
        assert torch.equal(im, want[b][0]) and torch.equal(dp, want[b][1]) and torch.equal(u8, want[b][2]), b
    # an announced batch that does not come: the call renders its own inputs
    nerf.render(dec, codes[0], bits[0], 128, 128, intr, poses[0], grid_size=64, bg_color=1.0, cfg={}, planes=planes[0], next_batch=(bits[1], intr, poses[1]))
    im, dp = nerf.render(dec, codes[0], bits[0], 128, 128, intr, poses[0], grid_size=64, bg_color=1.0, cfg={}, planes=planes[0])
    assert torch.equal(im, want[0][0]) and torch.equal(dp, want[0][1])
    # ... and a bitfield changed IN PLACE between the announcement and the call is noticed (tensor version)
    nerf.render(dec, codes[0], bits[0], 128, 128, intr, poses[0], grid_size=64, bg_color=1.0, cfg={}, planes=planes[0], next_batch=(bits[0], intr, poses[0]))
    saved = bits[0].clone()
    bits[0].copy_(bits[1])
    im2, _ = nerf.render(dec, codes[0], bits[0], 128, 128, intr, poses[0], grid_size=64, bg_color=1.0, cfg={}, planes=planes[0])
    bits[0].copy_(saved)
    im3, _ = nerf.render(dec, codes[0], bits[1], 128, 128, intr, poses[0], grid_size=64, bg_color=1.0, cfg={}, planes=planes[0])
    assert torch.equal(im2, im3)
