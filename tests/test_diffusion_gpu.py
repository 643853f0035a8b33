"""GPU parity of the DDIM / guidance rows against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny_unet_cfg(image_size=16):
    return dict(type="DenoisingUnetMod", image_size=image_size, in_channels=18, base_channels=32, channels_cfg=[1, 2], resblocks_per_downsample=1,
                dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[image_size // 2],
                norm_cfg=dict(type="GN", num_groups=8))


def _randomize(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.2 / max(1.0, p[0].numel() ** 0.5) if p.dim() > 1 else 0.1))


@pytest.fixture(scope="module")
def diffusion():
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.diffusion import GaussianDiffusion
    d = GaussianDiffusion(denoising=_tiny_unet_cfg(), betas_cfg=dict(type="linear"), num_timesteps=1000, denoising_mean_mode="V",
                          test_cfg=dict(num_timesteps=10, clip_range=[-2, 2]))
    _randomize(d, 5)
    return d.eval()


def test_unet_gpu_matches_cpu_oracle(diffusion):
    from oracle import diffusion as OD
    g = torch.Generator().manual_seed(1)
    x, t = torch.randn(2, 18, 16, 16, generator=g), torch.tensor([999, 3])
    sd = {k: v.clone() for k, v in diffusion.denoising.state_dict().items()}
    y0 = OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1, num_heads=4,
                         attention_res=(8,), norm_groups=8)
    d = diffusion.cuda()
    with torch.no_grad():
        y = d.denoising(x.cuda(), t.cuda())
    np.testing.assert_allclose(y.cpu().numpy(), y0.numpy(), rtol=1e-3, atol=1e-4)
    diffusion.cpu()


def test_fused_ddim_step_equals_reference_ordered_eager(diffusion):
    from oracle import diffusion as OD
    d = diffusion.cuda()
    g = torch.Generator().manual_seed(2)
    noise = torch.randn(2, 18, 16, 16, generator=g)
    with torch.no_grad():
        d.use_fused_step = True
        fused = d.ddim_sample(noise.cuda(), save_intermediates=True)
        d.use_fused_step = False
        eager = d.ddim_sample(noise.cuda(), save_intermediates=True)
        d.use_fused_step = True
    for a, b in zip(fused, eager):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=3e-6)
    assert torch.equal(fused[-1], fused[-2])                                   # last step: x_prev == x0_pred
    sd = {k: v.cpu() for k, v in d.denoising.state_dict().items()}
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(8,), norm_groups=8)
    want = OD.ddim_sample(den, noise, OD.schedule_tables(1000, "linear"), 10, clip_range=(-2, 2))
    np.testing.assert_allclose(fused[-1].cpu().numpy(), want.numpy(), rtol=0, atol=2e-4)
    diffusion.cpu()


@pytest.fixture(scope="module")
def model():
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODELS
    from ssdnerf_amd import synthetic as S
    cfg = dict(type="DiffusionNeRF", code_size=(3, 6, 128, 128), code_reshape=(18, 128, 128), code_activation=dict(type="TanhCode", scale=2),
               grid_size=64,
               diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"),
                              denoising=dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=32, channels_cfg=[1, 1, 2],
                                             resblocks_per_downsample=1, dropout=0.0, use_scale_shift_norm=True, num_heads=4, attention_res=[32],
                                             norm_cfg=dict(type="GN", num_groups=8))),
               decoder=dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3],
                            use_dir_enc=True, dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001,
                            max_steps=256),
               decoder_use_ema=True, freeze_decoder=False, bg_color=1, pixel_loss=dict(type="MSELoss", loss_weight=20.0),
               reg_loss=dict(type="RegLoss", power=2, loss_weight=3e-3), cache_size=0,
               test_cfg=dict(img_size=(128, 128), num_timesteps=4, clip_range=[-2, 2], density_thresh=0.1, n_inverse_rays=2 ** 14,
                             loss_coef=0.1 / (128 * 128), guidance_gain=1.0, dt_gamma_scale=0.5, cond_mode="guide"))
    m = MODELS.build(cfg)
    _randomize(m.diffusion_ema, 9)
    m.decoder_ema.load_state_dict(S.make_decoder_params(), strict=False)
    return m.cuda().eval()


def test_val_uncond_end_to_end(model):
    """noise -> DDIM (fused step) -> code -> 8 density refreshes -> render one view; vs the oracle DDIM on the CPU."""
    from oracle import diffusion as OD
    from ssdnerf_amd import synthetic as S
    g = torch.Generator().manual_seed(4)
    noise = torch.randn(2, 3, 6, 128, 128, generator=g)
    jit = [torch.rand(64 ** 3, 3, generator=g) for _ in range(8)]
    code, grid, bits = model.val_uncond(dict(scene_id=[0, 1], noise=noise.cuda()), density_jitters=[j.cuda() for j in jit])
    assert code.shape == (2, 3, 6, 128, 128) and grid.dtype == torch.float16 and bits.shape == (2, 64 ** 3 // 8)
    sd = {k: v.cpu() for k, v in model.diffusion_ema.denoising.state_dict().items()}
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=128, base_channels=32, channels_cfg=(1, 1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(32,), norm_groups=8)
    want = OD.ddim_sample(den, noise.reshape(2, 18, 128, 128), OD.schedule_tables(1000, "linear"), 4, clip_range=(-2, 2)).reshape(2, 3, 6, 128, 128)
    np.testing.assert_allclose(code.cpu().numpy(), want.numpy(), rtol=0, atol=5e-4)
    poses = S.spiral_poses()[[40]].cuda()[None].expand(2, -1, -1, -1)
    intr = S.cars_intrinsics().cuda()[None, None].expand(2, 1, -1)
    image, depth = model.render(model.decoder_ema, code, bits, 128, 128, intr, poses, cfg=model.test_cfg)
    assert image.shape == (2, 1, 128, 128, 3) and bool(torch.isfinite(image).all()) and float(image.min()) >= -0.002
    out = model.val_step(dict(scene_id=[0, 1], noise=noise.cuda(), test_poses=poses, test_intrinsics=intr), density_jitters=[j.cuda() for j in jit])
    assert out["pred_imgs"].shape == (2, 1, 3, 128, 128)


def test_guidance_loss_and_gradient_match_oracle(model):
    """One evaluation of the guidance closure (train-branch render + loss) : integer march record bit-exact, loss and d(loss)/d(code)
    within fp32 tolerance of the CPU oracle with autograd."""
    from oracle import guidance as OG, render as R
    from ssdnerf_amd import synthetic as S
    params = S.make_decoder_params()
    code_cpu = S.make_triplane(31).requires_grad_(True)
    g = torch.Generator().manual_seed(8)
    jit = [torch.rand(64 ** 3, 3, generator=g).numpy() for _ in range(2)]
    _, bits, _ = R.get_density(params, code_cpu.detach(), jit, density_thresh=0.1, dtype=np.float32)
    ro, rd = R.get_cam_rays(S.spiral_poses()[64][None], S.cars_intrinsics()[None], 128, 128)
    ro, rd = ro.reshape(-1, 3).numpy(), rd.reshape(-1, 3).numpy()
    target = torch.rand(128 * 128, 3, generator=g)
    noises = torch.rand(128 * 128, generator=g).numpy()
    dt_gamma = 0.5 / 131.25
    loss0, rec = OG.guidance_loss(params, code_cpu, bits, ro, rd, target, noises, dt_gamma)
    (g0,) = torch.autograd.grad(loss0, code_cpu)

    dec = model.decoder_ema
    code = code_cpu.detach().cuda()[None].requires_grad_(True)
    dec.train(True)
    try:
        for p in dec.parameters():
            p.requires_grad_(False)
        dec.injected_noises = torch.from_numpy(noises).cuda()[None]
        _, loss, _ = model.loss(dec, code, torch.from_numpy(bits).cuda()[None], target.cuda()[None], torch.from_numpy(ro).cuda()[None],
                                torch.from_numpy(rd).cuda()[None], torch.tensor([dt_gamma]).cuda(), scale_num_ray=128 * 128, cfg=model.test_cfg)
        (g1,) = torch.autograd.grad(loss, code)
    finally:
        dec.injected_noises = None
        dec.train(False)
    assert abs(float(loss.detach()) - float(loss0.detach())) <= 2e-5 * max(1.0, abs(float(loss0.detach())))
    g1 = g1[0].cpu()
    denom = float(g0.abs().max())
    assert denom > 0
    assert float((g1 - g0).abs().max()) <= 2e-4 * denom
    assert rec["num_points"] > 5000


def test_val_guide_runs_the_guidance_closure_every_step(model):
    from ssdnerf_amd import synthetic as S
    g = torch.Generator().manual_seed(12)
    noise = torch.randn(1, 3, 6, 128, 128, generator=g).cuda()
    poses = S.spiral_poses()[[64]].cuda()[None]
    intr = S.cars_intrinsics().cuda()[None, None]
    cond = torch.rand(1, 1, 128, 128, 3, generator=g).cuda()
    data = dict(cond_imgs=cond, cond_intrinsics=intr, cond_poses=poses, noise=noise)
    calls = []
    orig = model.loss
    model.loss = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        code_g, grid, bits = model.val_guide(data)
    finally:
        model.loss = orig
    assert len(calls) == model.test_cfg["num_timesteps"]                      # one train-branch render + backward per DDIM step
    assert code_g.shape == (1, 3, 6, 128, 128) and bool(torch.isfinite(code_g).all()) and not code_g.requires_grad
    assert grid.dtype == torch.float32 and float(grid.max()) > 0              # the density grid was refreshed from x0_pred (fp32, diffusion_nerf.py:278)
    assert all(p.requires_grad for p in model.diffusion_ema.parameters())      # requires_grad flags restored
    # (the quantitative check of the guidance term is test_guidance_loss_and_gradient_match_oracle; with random UNet weights the
    #  predicted x0 holds no occupied voxels, so the rendering loss has nothing to push against here)


# ---------------------------------------------------------------------------------------------- fine-tuning
@pytest.fixture(scope="module")
def recons_model():
    """the recons1v configuration (configs/paper_cfgs/ssdnerf_cars_recons1v.py:5-110) with a small UNet and a short schedule"""
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODELS
    from ssdnerf_amd import synthetic as S
    cfg = dict(type="DiffusionNeRF", code_size=(3, 6, 128, 128), code_reshape=(18, 128, 128), code_activation=dict(type="TanhCode", scale=2),
               grid_size=64,
               diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"),
                              denoising=dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=32, channels_cfg=[1, 1, 2],
                                             resblocks_per_downsample=1, dropout=0.0, use_scale_shift_norm=True, num_heads=4, attention_res=[32],
                                             norm_cfg=dict(type="GN", num_groups=8)),
                              timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
                              ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight",
                                             log_cfgs=dict(type="quartile", prefix_name="loss_mse", total_timesteps=1000),
                                             data_info=dict(pred="v_t_pred", target="v_t"), weight_scale=4.0, scale_norm=True)),
               decoder=dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3],
                            use_dir_enc=True, dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001,
                            max_steps=256),
               decoder_use_ema=True, freeze_decoder=False, bg_color=1, pixel_loss=dict(type="MSELoss", loss_weight=20.0),
               reg_loss=dict(type="RegLoss", power=2, loss_weight=3e-3), cache_size=0,
               test_cfg=dict(img_size=(128, 128), num_timesteps=2, clip_range=[-2, 2], density_thresh=0.1, dt_gamma_scale=0.5,
                             n_inverse_rays=2 ** 14, override_cfg={"diffusion_ema.ddpm_loss.weight_scale": 1.0}, loss_coef=0.1 / (128 * 128),
                             guidance_gain=3.2 * (2 ** 14), cond_mode="guide_optim", n_inverse_steps=2, extra_scene_step=1,
                             optimizer=dict(type="Adam", lr=0.005, weight_decay=0.0), lr_scheduler=dict(type="ExponentialLR", gamma=0.998)))
    m = MODELS.build(cfg)
    _randomize(m.diffusion_ema.denoising, 9)
    m.diffusion_ema.ddpm_loss.norm_factor.fill_(0.8)
    m.decoder_ema.load_state_dict(S.make_decoder_params(), strict=False)
    return m.cuda().eval()


def test_val_optim_matches_oracle(recons_model):
    """2 outer x 2 inner fine-tuning iterations of one scene (prior loss through the UNet, train-branch render, grid refresh with decay,
    SGD + exponential LR so that the update is linear in the gradients) against the CPU restatement with every random draw injected."""
    from oracle import diffusion as OD, guidance as OG, render as R
    from ssdnerf_amd import synthetic as S
    m = recons_model
    params = S.make_decoder_params()
    g = torch.Generator().manual_seed(33)
    code0 = S.make_triplane(31)
    code0_ = m.code_activation.inverse(code0).contiguous()
    jit0 = [torch.rand(64 ** 3, 3, generator=g).numpy() for _ in range(2)]
    grid0, bits0, _ = R.get_density(params, code0, jit0, density_thresh=0.1, dtype=np.float32)
    ro, rd = R.get_cam_rays(S.spiral_poses()[64][None], S.cars_intrinsics()[None], 128, 128)
    ro, rd = ro.reshape(-1, 3).numpy(), rd.reshape(-1, 3).numpy()
    target = torch.rand(128 * 128, 3, generator=g)
    ts = [torch.tensor([700]), torch.tensor([150])]
    ns = [torch.randn(1, 18, 128, 128, generator=g) for _ in range(2)]
    marches = [torch.rand(128 * 128, generator=g) for _ in range(4)]
    jits = [torch.rand(64 ** 3, 3, generator=g) for _ in range(2)]
    opt = dict(type="SGD", lr=100.0)

    sd = {k: v.cpu() for k, v in m.diffusion_ema.denoising.state_dict().items()}
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=128, base_channels=32, channels_cfg=(1, 1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(32,), norm_groups=8)
    tables = OD.schedule_tables(1000, "linear")
    w, _ = OD.snr_timestep_weights(tables, 0.5, "V")
    act = lambda c: c.tanh() * 2
    grid_cpu = grid0.copy()
    want, bits_want, losses_want = OG.finetune_code(
        params, den, code0_.clone(), act, tables, w, grid_cpu, ro, rd, target, 0.5 / 131.25, ts, ns, [n.numpy() for n in marches],
        [j.numpy() for j in jits], n_outer=2, n_inner=2, optimizer=opt, lr_gamma=0.998, weight_scale=1.0, norm_factor=0.8, density_thresh=0.1)

    saved = dict(m.test_cfg)
    m.test_cfg.update(optimizer=opt)
    losses = []
    orig = m.loss
    m.loss = lambda *a, **k: (lambda r: (losses.append(r[1].detach()), r)[1])(orig(*a, **k))
    try:
        data = dict(cond_imgs=target.reshape(1, 1, 128, 128, 3).cuda(), cond_intrinsics=S.cars_intrinsics().cuda()[None, None],
                    cond_poses=S.spiral_poses()[[64]].cuda()[None])
        code, grid, bits = m.val_optim(data, code_=code0_.clone().cuda()[None].requires_grad_(True), density_grid=torch.from_numpy(grid0).cuda()[None],
                                       density_bitfield=torch.from_numpy(bits0).cuda()[None], prior_timesteps=ts,
                                       prior_noises=[n.cuda() for n in ns], march_noises=[n.cuda()[None] for n in marches],
                                       density_jitters=[j.cuda() for j in jits])
    finally:
        m.loss = orig
        m.test_cfg.clear(); m.test_cfg.update(saved)
    moved = float((want - code0).abs().max())
    err = float((code[0].cpu() - want).abs().max())
    lo = [float(v) for v in losses]
    print(f"val_optim parity: max|code - oracle| = {err:.3e}, max|update| = {moved:.3e}, losses {lo} vs {losses_want}")
    assert moved > 1e-3, "the fine-tuning steps must move the code measurably for this comparison to mean anything"
    assert err <= 1e-3 * moved
    assert len(lo) == 4 and all(abs(a - b) <= 2e-4 * abs(b) for a, b in zip(lo, losses_want))
    assert int((np.unpackbits(bits[0].cpu().numpy()) != np.unpackbits(bits_want)).sum()) <= 8      # threshold-edge cells only
    np.testing.assert_allclose(grid[0].cpu().numpy(), grid_cpu, rtol=2e-4, atol=1e-5)


def test_guide_optim_end_to_end(recons_model):
    """cond_mode='guide_optim' through val_step: guided DDIM, then fine-tuning with Adam + ExponentialLR as configured, then a render."""
    from ssdnerf_amd import synthetic as S
    m = recons_model
    g = torch.Generator().manual_seed(12)
    noise = torch.randn(2, 3, 6, 128, 128, generator=g).cuda()
    poses = S.spiral_poses()[[64]].cuda()[None].expand(2, -1, -1, -1)
    intr = S.cars_intrinsics().cuda()[None, None].expand(2, 1, -1)
    cond = torch.rand(2, 1, 128, 128, 3, generator=g).cuda()
    steps = []
    orig = m.inverse_code
    m.inverse_code = lambda *a, **k: (steps.append(k["cfg"]["n_inverse_steps"]), orig(*a, **k))[1]
    try:
        np.random.seed(0)
        out = m.val_step(dict(cond_imgs=cond, cond_intrinsics=intr, cond_poses=poses, noise=noise, test_poses=poses, test_intrinsics=intr))
    finally:
        m.inverse_code = orig
    assert steps == [2, 2]                                                       # n_inverse_steps outer calls of extra_scene_step + 1 iterations
    assert out["code"].shape == (2, 3, 6, 128, 128) and bool(torch.isfinite(out["code"]).all()) and not out["code"].requires_grad
    assert float(out["code"].abs().max()) <= 2.0 and out["pred_imgs"].shape == (2, 1, 3, 128, 128)
    assert m.diffusion_ema.ddpm_loss.weight_scale == 1.0 and not m.decoder_ema.training
    assert all(p.requires_grad for p in m.diffusion_ema.denoising.parameters())


# ---------------------------------------------------------------------------------------------- the training step
def test_diffusion_nerf_train_step_matches_oracle(tmp_path):
    """``DiffusionNeRF.train_step`` (lib/models/autodecoders/diffusion_nerf.py:66-189) through the REAL renderer on the GPU against the same step
    written out on the CPU with the oracle's renderer (C march / composite, PyTorch-CPU decode with autograd): prior loss at drawn (t, noise) ->
    its gradient on the pre-activation code seeds ``extra_scene_step`` code-only rendering iterations (grid refresh with decay, jittered march)
    -> the joint iteration that steps the decoder as well.  Every random draw is injected on both sides (initial code, timestep and noise through
    the host generators, march jitter and grid jitter through the model's public hooks); SGD everywhere, so updates are linear in the
    gradients.  Compared: the cached pre-activation code, the stepped decoder weights, the logged losses, the density grid."""
    import ssdnerf_amd  # noqa: F401
    from oracle import diffusion as OD, guidance as OG, render as R
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.registry import MODELS
    E, lr_c, lr_d = 1, 200.0, 0.05
    hw = 64
    m = MODELS.build(dict(
        type="DiffusionNeRF", code_size=(3, 6, 128, 128), code_reshape=(18, 128, 128), code_activation=dict(type="TanhCode", scale=2), grid_size=64,
        diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"),
                       denoising=dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=32, channels_cfg=[1, 1, 2],
                                      resblocks_per_downsample=1, dropout=0.0, use_scale_shift_norm=True, num_heads=4, attention_res=[32],
                                      norm_cfg=dict(type="GN", num_groups=8)),
                       timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
                       ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight", data_info=dict(pred="v_t_pred", target="v_t"),
                                      weight_scale=4.0, scale_norm=True)),
        decoder=dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3],
                     use_dir_enc=True, dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256),
        decoder_use_ema=True, freeze_decoder=False, bg_color=1, pixel_loss=dict(type="MSELoss", loss_weight=20.0),
        reg_loss=dict(type="RegLoss", power=2, loss_weight=3e-3), cache_size=4,
        train_cfg=dict(dt_gamma_scale=0.5, density_thresh=0.1, extra_scene_step=E, n_inverse_rays=hw * hw, n_decoder_rays=hw * hw,
                       loss_coef=0.1 / (hw * hw), optimizer=dict(type="SGD", lr=lr_c))))
    _randomize(m.diffusion.denoising, 9)
    params0 = S.make_decoder_params()
    m.decoder.load_state_dict(params0, strict=False)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(41)
    code0_ = m.code_activation.inverse(S.make_triplane(37)).contiguous()                 # the "fresh" pre-activation code of the scene
    target = torch.rand(hw * hw, 3, generator=g)
    marches = [torch.rand(hw * hw, generator=g) for _ in range(E + 1)]
    jits = [torch.rand(64 ** 3, 3, generator=g) for _ in range(E + 1)]                  # one refresh in the code-only iterations (k = 0), one in the joint step
    pose, intr = S.spiral_poses()[[64]], S.cars_intrinsics(hw, hw)

    # ---- the product, with the draws routed in through its public hooks
    m.get_init_code_ = lambda num_scenes, device=None: code0_.clone().to(device).requires_grad_(True)
    orig_loss, orig_update = m.loss, m.update_extra_state
    it_m, it_j, losses = iter(marches), iter(jits), []

    def loss_with_injected_jitter(decoder, *a, **k):
        decoder.injected_noises = next(it_m).cuda()[None]
        try:
            out = orig_loss(decoder, *a, **k)
        finally:
            decoder.injected_noises = None
        losses.append(out[1].detach())
        return out

    m.loss = loss_with_injected_jitter
    m.update_extra_state = lambda decoder, code, grid, bits, it, **k: orig_update(decoder, code, grid, bits, it, **dict(k, jitter=next(it_j).cuda()))
    data = dict(scene_id=[1], scene_name=["a"], cond_imgs=target.reshape(1, 1, hw, hw, 3).cuda(), cond_poses=pose.cuda()[None],
                cond_intrinsics=intr.cuda()[None, None])
    opt = dict(diffusion=torch.optim.SGD(m.diffusion.parameters(), lr=1e-3), decoder=torch.optim.SGD(m.decoder.parameters(), lr=lr_d))
    sd = {k: v.detach().cpu().clone() for k, v in m.diffusion.denoising.state_dict().items()}
    norm0 = float(m.diffusion.ddpm_loss.norm_factor)
    unet_before = m.diffusion.denoising.out.conv.weight.detach().clone()
    np.random.seed(13); torch.manual_seed(13)
    state = torch.random.get_rng_state()
    t_exp = m.diffusion.sampler(1)                                                       # the draws train_step is about to make, in its order
    n_exp = torch.randn(1, 18, 128, 128)
    np.random.seed(13); torch.random.set_rng_state(state)
    out = m.train_step(data, opt)
    assert len(losses) == E + 1 and not torch.equal(m.diffusion.denoising.out.conv.weight, unet_before)
    got_code_ = m.cache[1]["param"]["code_"].float().cpu()
    got_grid = m.cache[1]["param"]["density_grid"].float().cpu().numpy()

    # ---- the same step on the CPU
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=128, base_channels=32, channels_cfg=(1, 1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(32,), norm_groups=8)
    tables = OD.schedule_tables(1000, "linear")
    w, _ = OD.snr_timestep_weights(tables, 0.5, "V")
    ro, rd = R.get_cam_rays(pose, intr[None], hw, hw)
    ro, rd = ro.reshape(-1, 3).numpy(), rd.reshape(-1, 3).numpy()
    dt_gamma = 0.5 / float(intr[:2].mean())
    c_ = code0_.clone().requires_grad_(True)
    x0 = (c_.tanh() * 2).reshape(1, 18, 128, 128)
    norm1 = 0.999 * norm0 + 0.001 * float(x0.detach().square().mean())                   # training mode: the running norm moves before it divides
    prior = OD.prior_loss_v(den, x0, t_exp, n_exp, tables, w, weight_scale=4.0, norm_factor=norm1)
    (pg,) = torch.autograd.grad(prior, c_)
    params = {k: v.clone() for k, v in params0.items()}
    grid = np.zeros(64 ** 3, dtype=np.float16)                                           # a fresh scene's grid is fp16 zeros (get_init_density_grid)
    kw = dict(loss_weight=20.0, loss_coef=0.1 / (hw * hw), reg_weight=3e-3, reg_power=2, scale_num_ray=hw * hw)
    want_losses = []
    for i in range(E):                                                                   # code-only iterations, decoder frozen
        code = c_.tanh() * 2
        bits, _ = R.update_extra_state(params, code.detach(), grid, jits[i].numpy(), density_thresh=0.1, decay=0.9)
        loss, _ = OG.guidance_loss(params, code, bits, ro, rd, target, marches[i].numpy(), dt_gamma, **kw)
        (rg,) = torch.autograd.grad(loss, c_)
        with torch.no_grad():
            c_ -= lr_c * (pg + rg)
        want_losses.append(float(loss.detach()))
    leaf_params = {k: v.clone().requires_grad_(True) for k, v in params.items()}         # the joint iteration: decoder and code
    code = c_.tanh() * 2
    bits, _ = R.update_extra_state(params, code.detach(), grid, jits[E].numpy(), density_thresh=0.1, decay=0.9)
    loss, _ = OG.guidance_loss(leaf_params, code, bits, ro, rd, target, marches[E].numpy(), dt_gamma, **kw)
    grads = torch.autograd.grad(loss, [c_] + list(leaf_params.values()))
    want_losses.append(float(loss.detach()))
    with torch.no_grad():
        c_ -= lr_c * (pg + grads[0])
    want_dec = {k: v.detach() - lr_d * gk for (k, v), gk in zip(leaf_params.items(), grads[1:])}

    moved = float((c_.detach() - code0_).abs().max())
    err = float((got_code_ - c_.detach()).abs().max())
    lo = [float(v) for v in losses]
    print(f"train_step parity: max|code_ - oracle| = {err:.3e}, max|update| = {moved:.3e}, losses {lo} vs {want_losses}")
    assert moved > 1e-3 and err <= 2e-3 * moved
    assert all(abs(a - b) <= 3e-4 * abs(b) for a, b in zip(lo, want_losses))
    sd_dec = m.decoder.state_dict()
    for k, v in want_dec.items():
        step = float((v - params0[k]).abs().max())
        assert float((sd_dec[k].cpu() - v).abs().max()) <= 2e-3 * step + 1e-7, k        # every decoder tensor moved by the oracle's gradient
    assert float(out["log_vars"]["loss_decoder"]) == pytest.approx(want_losses[-1], rel=3e-4)
    np.testing.assert_allclose(got_grid, grid.astype(np.float32), rtol=2e-3, atol=1e-4)
