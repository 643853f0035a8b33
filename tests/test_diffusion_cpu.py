"""CPU tests of the DDIM sampler and UNet module against the oracle restatements (oracle/diffusion.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion as OD


def _tiny_unet_cfg():
    return dict(type="DenoisingUnetMod", image_size=16, in_channels=18, base_channels=32, channels_cfg=[1, 2], resblocks_per_downsample=1,
                dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[8],
                norm_cfg=dict(type="GN", num_groups=8))


def _randomize(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():                      # also un-zero conv_2 / proj / out so errors cannot hide
            p.copy_(torch.randn(p.shape, generator=g) * (0.2 / max(1.0, p[0].numel() ** 0.5) if p.dim() > 1 else 0.1))


@pytest.fixture(scope="module")
def diffusion():
    import ssdnerf_amd.unet  # noqa: F401
    from ssdnerf_amd.diffusion import GaussianDiffusion
    d = GaussianDiffusion(denoising=_tiny_unet_cfg(), betas_cfg=dict(type="linear"), num_timesteps=1000, denoising_mean_mode="V",
                          test_cfg=dict(num_timesteps=10, clip_range=[-2, 2]))
    _randomize(d, 5)
    return d.eval()


def test_schedule_tables_and_timesteps(diffusion):
    t = OD.schedule_tables(1000, "linear")
    for k in ("betas", "alphas_bar", "alphas_bar_prev", "sqrt_alphas_bar", "sqrt_one_minus_alphas_bar", "tilde_betas_t"):
        np.testing.assert_array_equal(getattr(diffusion, k), t[k])
    assert diffusion.alphas_bar_prev[0] == 1.0
    ts50 = diffusion.ddim_timesteps(50).tolist()
    assert ts50[:3] == [999, 979, 959] and ts50[-1] == 19 and len(ts50) == 50
    ts75 = diffusion.ddim_timesteps(75).tolist()
    assert ts75[:4] == [999, 985, 972, 959] and ts75[-1] == 12 and len(ts75) == 75
    assert ts75 == OD.ddim_timesteps(1000, 75)
    from ssdnerf_amd.diffusion import GaussianDiffusion
    c = GaussianDiffusion(denoising=_tiny_unet_cfg(), betas_cfg=dict(type="cosine"), num_timesteps=100)
    np.testing.assert_allclose(c.betas, OD.schedule_tables(100, "cosine")["betas"], rtol=1e-12)


def test_unet_module_matches_functional_restatement(diffusion):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 18, 16, 16, generator=g)
    t = torch.tensor([999, 3])
    with torch.no_grad():
        y = diffusion.denoising(x, t)
        y0 = OD.unet_forward(diffusion.denoising.state_dict(), x, t, image_size=16, base_channels=32, channels_cfg=(1, 2),
                             resblocks_per_downsample=1, num_heads=4, attention_res=(8,), norm_groups=8)
    assert y.shape == x.shape and float(y.abs().mean()) > 1e-3
    np.testing.assert_allclose(y.numpy(), y0.numpy(), rtol=1e-4, atol=2e-5)


def test_ddim_sample_matches_oracle(diffusion):
    g = torch.Generator().manual_seed(2)
    noise = torch.randn(2, 18, 16, 16, generator=g)
    sd = diffusion.denoising.state_dict()
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(8,), norm_groups=8)
    want, trace = OD.ddim_sample(den, noise, OD.schedule_tables(1000, "linear"), 10, clip_range=(-2, 2), return_all=True)
    with torch.no_grad():
        got = diffusion(noise.clone(), return_loss=False)
        both = diffusion.ddim_sample(noise.clone(), save_intermediates=True)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=2e-5)
    assert len(both) == 20
    assert torch.equal(both[-1], both[-2])                      # last step: alpha_bar_prev = 1 -> x_prev == x0_pred exactly
    assert float(got.abs().max()) <= 2.0


def test_ddim_guidance_matches_oracle(diffusion):
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(1, 18, 16, 16, generator=g)
    target = torch.randn(1, 18, 16, 16, generator=g)
    guide = lambda x0: ((x0 - target) ** 2).mean() * 5.0
    sd = diffusion.denoising.state_dict()
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(8,), norm_groups=8)
    want = OD.ddim_sample(den, noise, OD.schedule_tables(1000, "linear"), 10, clip_range=(-2, 2), grad_guide_fn=guide, guidance_gain=2.0)
    diffusion.test_cfg["guidance_gain"] = 2.0
    try:
        for p in diffusion.parameters():
            p.requires_grad_(False)
        got = diffusion(noise.clone(), return_loss=False, grad_guide_fn=guide)
    finally:
        diffusion.test_cfg.pop("guidance_gain")
    np.testing.assert_allclose(got.detach().numpy(), want.numpy(), rtol=0, atol=5e-5)
    plain = diffusion(noise.clone(), return_loss=False)
    assert float((got - plain).abs().max()) > 1e-3               # guidance actually moved the sample


def test_code_layout_roundtrip_and_losses():
    from ssdnerf_amd.models import RegLoss, MSELoss, TanhCode, NormalizedTanhCode
    x = torch.randn(2, 3, 6, 8, 8)
    from ssdnerf_amd.models import DiffusionNeRF
    dn = DiffusionNeRF.__new__(DiffusionNeRF)
    torch.nn.Module.__init__(dn)
    dn.code_size, dn.code_reshape, dn.code_permute = (3, 6, 8, 8), (18, 8, 8), None
    dn.code_reshape_inv, dn.code_permute_inv = dn.code_size, None
    y = dn.code_diff_pr(x)
    assert y.shape == (2, 18, 8, 8) and torch.equal(dn.code_diff_pr_inv(y), x)
    dn.code_permute, dn.code_reshape = (1, 2, 0, 3), (6, 8, 24)                  # the tiled layout of new_cfgs/*_tiled.py
    dn.code_reshape_inv = [dn.code_size[a] for a in dn.code_permute]
    dn.code_permute_inv = [dn.code_permute.index(a) for a in range(4)]
    y = dn.code_diff_pr(x)
    assert y.shape == (2, 6, 8, 24) and torch.equal(dn.code_diff_pr_inv(y), x)
    assert abs(float(RegLoss(power=2, loss_weight=3e-3)(x)) - 3e-3 * float((x ** 2).mean())) < 1e-9
    assert abs(float(MSELoss(loss_weight=20.0)(x, x * 0)) - 20 * float((x ** 2).mean())) < 1e-5
    tc = TanhCode(scale=2)
    c = tc(x)
    assert float(c.abs().max()) <= 2 and torch.allclose(tc(tc.inverse(c)), c, atol=1e-4)
    nt = NormalizedTanhCode(std=0.5, clip_range=2)
    assert torch.allclose(nt(nt.inverse(nt(x))), nt(x), atol=1e-4)


# ---------------------------------------------------------------------------------------------- fine-tuning
def _recons_diffusion(weight_scale=4.0):
    """the recons configs' sampler / prior-loss entries (configs/paper_cfgs/ssdnerf_cars_recons1v.py:28-39) on the tiny UNet"""
    import ssdnerf_amd.unet  # noqa: F401
    from ssdnerf_amd.diffusion import GaussianDiffusion
    d = GaussianDiffusion(denoising=_tiny_unet_cfg(), betas_cfg=dict(type="linear"), num_timesteps=1000, denoising_mean_mode="V",
                          timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
                          ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight",
                                         log_cfgs=dict(type="quartile", prefix_name="loss_mse", total_timesteps=1000),
                                         data_info=dict(pred="v_t_pred", target="v_t"), weight_scale=weight_scale, scale_norm=True),
                          test_cfg=dict(num_timesteps=10, clip_range=[-2, 2]))
    _randomize(d.denoising, 5)
    return d.eval()


def test_snr_sampler_weights_and_draws():
    d = _recons_diffusion()
    t = OD.schedule_tables(1000, "linear")
    w, prob = OD.snr_timestep_weights(t, power=0.5, mode="V")
    np.testing.assert_array_equal(d.sampler.weight.numpy(), w)
    np.testing.assert_allclose(w, np.sqrt(t["alphas_bar"] * (1 - t["alphas_bar"])).astype(np.float32), rtol=2e-7)   # sqrt(SNR) * (1 - abar)
    np.testing.assert_allclose(d.sampler.prob, prob, rtol=1e-15)
    np.random.seed(3)
    a = d.sampler(8)
    np.random.seed(3)
    b = torch.from_numpy(np.random.choice(1000, size=(8,), p=prob)).long()       # host-side draw, as mmgen's sampler
    assert a.dtype == torch.long and torch.equal(a, b)
    from ssdnerf_amd.diffusion import SNRWeightedTimeStepSampler
    s2 = SNRWeightedTimeStepSampler(1000, t["sqrt_alphas_bar"], t["sqrt_one_minus_alphas_bar"], "EPS", power=1, min=0.5, max=4, prob_power=0.5)
    assert abs(sum(s2.prob) - 1) < 1e-12 and s2.weight.shape == (1000,)


def test_prior_loss_and_its_gradient_match_oracle():
    d = _recons_diffusion(weight_scale=4.0)
    d.ddpm_loss.norm_factor.fill_(1.7)                                           # as if loaded from a trained checkpoint
    g = torch.Generator().manual_seed(11)
    x0 = (torch.randn(3, 18, 16, 16, generator=g) * 0.7).requires_grad_(True)
    noise = torch.randn(3, 18, 16, 16, generator=g)
    ts = torch.tensor([17, 480, 995])
    loss, log_vars = d(x0, return_loss=True, timesteps=ts, noise=noise, cfg=d.test_cfg)
    (g1,) = torch.autograd.grad(loss, x0)
    x0b = x0.detach().clone().requires_grad_(True)
    sd = d.denoising.state_dict()
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(8,), norm_groups=8)
    w, _ = OD.snr_timestep_weights(OD.schedule_tables(1000, "linear"), 0.5, "V")
    want = OD.prior_loss_v(den, x0b, ts, noise, OD.schedule_tables(1000, "linear"), w, weight_scale=4.0, norm_factor=1.7)
    (g0,) = torch.autograd.grad(want, x0b)
    assert abs(float(loss.detach()) - float(want.detach())) <= 1e-5 * abs(float(want.detach()))
    assert float((g1 - g0).abs().max()) <= 2e-4 * float(g0.abs().max())
    assert float(d.ddpm_loss.norm_factor) == pytest.approx(1.7)                  # eval mode: the running norm is frozen
    assert set(log_vars) == {"loss_mse_quartile_0", "loss_mse_quartile_1", "loss_mse_quartile_2", "loss_mse_quartile_3", "loss_ddpm_mse"}
    assert float(log_vars["loss_mse_quartile_2"]) == 0.0 and float(log_vars["loss_mse_quartile_0"]) > 0   # no sample fell in [500, 750)
    # seeded default draws: timesteps from np.random on the host, noise from the CPU generator
    np.random.seed(5); torch.manual_seed(5)
    l1, _ = d(x0.detach(), return_loss=True, cfg=d.test_cfg)
    np.random.seed(5); torch.manual_seed(5)
    l2, _ = d(x0.detach(), return_loss=True, cfg=d.test_cfg)
    assert float(l1.detach()) == float(l2.detach())
    # training mode moves the running norm: EMA of mean(x_0^2)
    d.train()
    d(x0.detach(), return_loss=True, timesteps=ts, noise=noise, cfg=d.test_cfg)
    assert float(d.ddpm_loss.norm_factor) == pytest.approx(0.999 * 1.7 + 0.001 * float(x0.detach().square().mean()), rel=1e-6)
    d.eval()


def _finetune_model(test_cfg):
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODELS
    cfg = dict(type="DiffusionNeRF", code_size=(3, 6, 16, 16), code_reshape=(18, 16, 16), code_activation=dict(type="TanhCode", scale=2),
               grid_size=64,
               diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"), denoising=_tiny_unet_cfg(),
                              timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
                              ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight", data_info=dict(pred="v_t_pred", target="v_t"),
                                             weight_scale=4.0, scale_norm=True)),
               decoder=dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3],
                            use_dir_enc=True, dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001,
                            max_steps=256),
               decoder_use_ema=True, freeze_decoder=False, bg_color=1, pixel_loss=dict(type="MSELoss", loss_weight=20.0),
               reg_loss=dict(type="RegLoss", power=2, loss_weight=3e-3), cache_size=0, test_cfg=test_cfg)
    m = MODELS.build(cfg)
    _randomize(m.diffusion_ema.denoising, 9)
    return m


def test_override_cfg_switches_with_train_and_eval():
    m = _finetune_model(dict(override_cfg={"diffusion_ema.ddpm_loss.weight_scale": 1.0}))
    assert m.diffusion_ema.ddpm_loss.weight_scale == 4.0 and m.train_cfg_backup == {"diffusion_ema.ddpm_loss.weight_scale": 4.0}
    m.eval()
    assert m.diffusion_ema.ddpm_loss.weight_scale == 1.0
    m.train()
    assert m.diffusion_ema.ddpm_loss.weight_scale == 4.0
    m.eval(); m.eval()                                                          # a second eval() must not overwrite the backup
    m.train()
    assert m.diffusion_ema.ddpm_loss.weight_scale == 4.0


def test_val_optim_gradient_seeding_optimizer_and_schedule(monkeypatch):
    """Host logic of val_optim / inverse_code with the renderer replaced by a differentiable stand-in (the train-branch render needs the
    GPU library): the prior gradient seeds every inner step, the rendering gradient accumulates onto it, one optimizer and one LR
    schedule run across all outer x inner steps, and the grid is refreshed once per inverse_code call."""
    m = _finetune_model(dict(n_inverse_steps=3, extra_scene_step=2, n_inverse_rays=2 ** 14, density_thresh=0.1, dt_gamma_scale=0.5,
                             optimizer=dict(type="SGD", lr=0.05), lr_scheduler=dict(type="ExponentialLR", gamma=0.9),
                             override_cfg={"diffusion_ema.ddpm_loss.weight_scale": 1.0})).eval()
    g = torch.Generator().manual_seed(21)
    cond = torch.rand(1, 1, 8, 8, 3, generator=g)
    poses, intr = torch.eye(4)[None, None], torch.tensor([[[8.0, 8.0, 4.0, 4.0]]])
    target_vec = torch.randn(3 * 6 * 16 * 16, generator=g)
    refreshes, seen = [], []

    def fake_update(decoder, code, grid, bits, iter_density, density_thresh=0.01, decay=0.9, S=128, jitter=None):
        refreshes.append((iter_density, density_thresh, decay, jitter))

    def fake_loss(decoder, code, bits, target_rgbs, rays_o, rays_d, dt_gamma=0.0, return_decoder_loss=False, scale_num_ray=1.0, cfg=dict(),
                  perturb=True, **kw):
        assert decoder.training and target_rgbs.shape == (1, 64, 3) and scale_num_ray == 64
        seen.append(decoder.injected_noises)
        loss = (code.reshape(-1) * target_vec).sum() * 0.01 + code.square().mean()
        return target_rgbs, loss, dict(pixel_loss=loss)

    monkeypatch.setattr(m, "update_extra_state", fake_update)
    monkeypatch.setattr(m, "loss", fake_loss)
    code0_ = (torch.randn(1, 3, 6, 16, 16, generator=g) * 0.3)
    ts = [torch.tensor([900]), torch.tensor([500]), torch.tensor([40])]
    ns = [torch.randn(1, 18, 16, 16, generator=g) for _ in range(3)]
    marks = [torch.full((1, 64), float(i)) for i in range(9)]
    code, grid, bits = m.val_optim(dict(cond_imgs=cond, cond_intrinsics=intr, cond_poses=poses), code_=code0_.clone().requires_grad_(True),
                                   prior_timesteps=ts, prior_noises=ns, march_noises=marks, density_jitters=["j0", "j1", "j2"])
    assert [r[3] for r in refreshes] == ["j0", "j1", "j2"] and all(r[:3] == (0, 0.1, 0.9) for r in refreshes)
    assert [float(s[0, 0]) for s in seen] == [float(i) for i in range(9)]
    assert grid.dtype == torch.float16 and bits.dtype == torch.uint8 and not code.requires_grad
    assert not m.decoder_ema.training and all(p.requires_grad for p in m.diffusion_ema.parameters())

    # the same loop written out by hand
    sd = m.diffusion_ema.denoising.state_dict()
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(8,), norm_groups=8)
    tables = OD.schedule_tables(1000, "linear")
    w, _ = OD.snr_timestep_weights(tables, 0.5, "V")
    c_ = code0_.clone().requires_grad_(True)
    lr = 0.05
    for k in range(3):
        prior = OD.prior_loss_v(den, (c_.tanh() * 2).reshape(1, 18, 16, 16), ts[k], ns[k], tables, w, weight_scale=1.0, norm_factor=1.0)
        (pg,) = torch.autograd.grad(prior, c_)
        for i in range(3):
            cc = c_.tanh() * 2
            (rg,) = torch.autograd.grad((cc.reshape(-1) * target_vec).sum() * 0.01 + cc.square().mean(), c_)
            with torch.no_grad():
                c_ -= lr * (pg + rg)
            lr *= 0.9
    np.testing.assert_allclose(code.numpy(), (c_.tanh() * 2).detach().numpy(), rtol=0, atol=2e-6)

    # extra_scene_step = 0: one loss_decoder backward + step per outer iteration, prior gradient still included
    m.test_cfg.update(extra_scene_step=0, n_inverse_steps=2)
    del seen[:]
    code1, _, _ = m.val_optim(dict(cond_imgs=cond, cond_intrinsics=intr, cond_poses=poses), code_=code0_.clone().requires_grad_(True),
                              prior_timesteps=ts, prior_noises=ns, march_noises=marks)
    assert len(seen) == 2
    c_ = code0_.clone().requires_grad_(True)
    lr = 0.05
    for k in range(2):
        prior = OD.prior_loss_v(den, (c_.tanh() * 2).reshape(1, 18, 16, 16), ts[k], ns[k], tables, w, weight_scale=1.0, norm_factor=1.0)
        cc = c_.tanh() * 2
        (gsum,) = torch.autograd.grad(prior + (cc.reshape(-1) * target_vec).sum() * 0.01 + cc.square().mean(), c_)
        with torch.no_grad():
            c_ -= lr * gsum
        lr *= 0.9
    np.testing.assert_allclose(code1.numpy(), (c_.tanh() * 2).detach().numpy(), rtol=0, atol=2e-6)


def test_val_uncond_refines_the_sample_under_the_prior(monkeypatch):
    """``n_inverse_steps`` on a batch without conditioning views (set by every recons config; reference val_uncond,
    lib/models/autodecoders/diffusion_nerf.py:212-229): the sampled code is polished by n optimizer steps on its pre-activation under the
    diffusion loss alone.  Checked against the same loop written out with the oracle's prior loss."""
    m = _finetune_model(dict(num_timesteps=3, clip_range=[-2, 2], n_inverse_steps=2, density_thresh=0.1,
                             optimizer=dict(type="SGD", lr=50.0), lr_scheduler=dict(type="ExponentialLR", gamma=0.9),
                             override_cfg={"diffusion_ema.ddpm_loss.weight_scale": 1.0})).eval()
    monkeypatch.setattr(m, "get_density", lambda decoder, code, cfg=dict(), jitters=None: ("grid", "bits"))   # (the grid refresh needs the GPU library)
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(1, 3, 6, 16, 16, generator=g) * 0.7
    ts = [torch.tensor([700]), torch.tensor([120])]
    ns = [torch.randn(1, 18, 16, 16, generator=g) for _ in range(2)]
    code, grid, bits = m.val_uncond(dict(scene_id=[0], noise=noise), prior_timesteps=ts, prior_noises=ns)
    assert (grid, bits) == ("grid", "bits") and not code.requires_grad and all(p.requires_grad for p in m.diffusion_ema.parameters())
    m.test_cfg["n_inverse_steps"] = 0
    sampled, _, _ = m.val_uncond(dict(scene_id=[0], noise=noise))
    assert float((sampled - code).abs().max()) > 1e-3                          # the refinement moved the code
    sd = m.diffusion_ema.denoising.state_dict()
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(8,), norm_groups=8)
    tables = OD.schedule_tables(1000, "linear")
    w, _ = OD.snr_timestep_weights(tables, 0.5, "V")
    c_ = m.code_activation.inverse(sampled).clone().requires_grad_(True)
    lr = 50.0
    for k in range(2):
        prior = OD.prior_loss_v(den, (c_.tanh() * 2).reshape(1, 18, 16, 16), ts[k], ns[k], tables, w, weight_scale=1.0, norm_factor=1.0)
        (pg,) = torch.autograd.grad(prior, c_)
        with torch.no_grad():
            c_ -= lr * pg
        lr *= 0.9
    np.testing.assert_allclose(code.numpy(), (c_.tanh() * 2).detach().numpy(), rtol=0, atol=3e-6)


def test_init_code_optimizer_and_scheduler_builders():
    m = _finetune_model(dict(optimizer=dict(type="Adam", lr=0.005, weight_decay=0.0), lr_scheduler=dict(type="ExponentialLR", gamma=0.998)))
    c = m.get_init_code_(2, device="cpu")
    assert c.shape == (2, 3, 6, 16, 16) and c.requires_grad and c.is_leaf and float(c.abs().max()) <= m.init_scale
    opt = m.build_optimizer(c, m.test_cfg)
    sch = m.build_scheduler(opt, m.test_cfg)
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["lr"] == 0.005 and isinstance(sch, torch.optim.lr_scheduler.ExponentialLR)
    assert m.build_scheduler(opt, dict()) is None
    opts = m.build_optimizer([m.get_init_code_(None), m.get_init_code_(None)], m.test_cfg)
    assert isinstance(opts, list) and len(opts) == 2 and len(m.build_scheduler(opts, m.test_cfg)) == 2


# ---------------------------------------------------------------------------------------------- Langevin correction
def test_langevin_correction_steps_match_oracle(diffusion):
    """``langevin_steps`` / ``langevin_delta`` of ssdnerf_chairs_recons1v (configs/paper_cfgs/ssdnerf_chairs_recons1v.py:95-96): the
    correction steps run at t_prev after every DDIM step with 0 < t_prev < 1000, i.e. not after the last one; noise is drawn on the host."""
    g = torch.Generator().manual_seed(6)
    noise = torch.randn(2, 18, 16, 16, generator=g)
    saved = dict(diffusion.test_cfg)
    diffusion.test_cfg.update(num_timesteps=4, langevin_steps=2, langevin_delta=0.4)
    sd = diffusion.denoising.state_dict()
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(8,), norm_groups=8)
    try:
        torch.manual_seed(77)
        with torch.no_grad():
            got = diffusion(noise, return_loss=False)
        torch.manual_seed(77)
        zs = [torch.randn(2, 18, 16, 16) for _ in range(6)]                    # 3 DDIM steps with t_prev >= 0, 2 corrections each
        want = OD.ddim_sample(den, noise, OD.schedule_tables(1000, "linear"), 4, clip_range=(-2, 2), langevin_steps=2, langevin_delta=0.4,
                              langevin_noises=zs)
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=3e-4)
        plain = OD.ddim_sample(den, noise, OD.schedule_tables(1000, "linear"), 4, clip_range=(-2, 2))
        assert float((want - plain).abs().max()) > 1e-2                        # the corrections do change the sample
        # save_intermediates: per DDIM step (x0 of that step, the latent AFTER the corrections that follow it) -- gaussian_diffusion.py:313-326;
        # rebuilt here from the single-step entry points in the reference loop's order with the same host draws
        torch.manual_seed(77)
        with torch.no_grad():
            traj = diffusion.ddim_sample(noise, save_intermediates=True)
        assert len(traj) == 8 and float((traj[-1] - got).abs().max()) == 0.0
        torch.manual_seed(77)
        ts, x = [999, 749, 499, 249], noise
        with torch.no_grad():
            for k, t in enumerate(ts):
                t_prev = ts[k + 1] if k + 1 < len(ts) else -1
                x, x0 = diffusion.p_sample_ddim(x, t, t_prev, cfg=diffusion.test_cfg)
                if 0 < t_prev < 1000:
                    for _ in range(2):
                        x = diffusion.p_sample_langevin(x, t_prev, cfg=diffusion.test_cfg)
                np.testing.assert_allclose(traj[2 * k].numpy(), x0.numpy(), rtol=0, atol=2e-5)
                np.testing.assert_allclose(traj[2 * k + 1].numpy(), x.numpy(), rtol=0, atol=2e-5)
        # guided variant: the guidance closure is evaluated in the correction steps too (1 + 2 calls per step, 1 for the last)
        calls = []

        def guide(x0):
            calls.append(1)
            return (x0 ** 2).mean() * 5.0

        diffusion.test_cfg.update(guidance_gain=2.0)
        torch.manual_seed(78)
        got_g = diffusion(noise, return_loss=False, grad_guide_fn=guide)
        assert len(calls) == 3 * 3 + 1
        torch.manual_seed(78)
        zs = [torch.randn(2, 18, 16, 16) for _ in range(6)]
        want_g = OD.ddim_sample(den, noise, OD.schedule_tables(1000, "linear"), 4, clip_range=(-2, 2), grad_guide_fn=guide, guidance_gain=2.0,
                                langevin_steps=2, langevin_delta=0.4, langevin_noises=zs)
        np.testing.assert_allclose(got_g.detach().numpy(), want_g.numpy(), rtol=0, atol=3e-4)
    finally:
        diffusion.test_cfg.clear(); diffusion.test_cfg.update(saved)


# ---------------------------------------------------------------------------------------------- training steps
def test_train_steps_compose_prior_gradient_inner_iterations_and_cache(monkeypatch, tmp_path):
    """DiffusionNeRF.train_step / MultiSceneNeRF.train_step (diffusion_nerf.py:66-189, multiscene_nerf.py:185-245) with the renderer replaced by
    a differentiable stand-in: denoiser stepped on the prior loss, its code gradient seeding extra_scene_step + 1 rendering iterations,
    decoder stepped once, cache written back -- against the same arithmetic written out by hand."""
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODELS
    E, lr_c = 2, 0.05
    cfg = dict(type="DiffusionNeRF", code_size=(3, 6, 16, 16), code_reshape=(18, 16, 16), code_activation=dict(type="TanhCode", scale=2), grid_size=16,
               diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(type="linear"), denoising=_tiny_unet_cfg(),
                              timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
                              ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight", data_info=dict(pred="v_t_pred", target="v_t"),
                                             weight_scale=4.0, scale_norm=True)),
               decoder=dict(type="TriPlaneDecoder", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64]),
               decoder_use_ema=True, freeze_decoder=False, bg_color=1, pixel_loss=dict(type="MSELoss", loss_weight=20.0), cache_size=4,
               train_cfg=dict(dt_gamma_scale=0.5, density_thresh=0.1, extra_scene_step=E, n_inverse_rays=2 ** 12, n_decoder_rays=2 ** 12,
                              loss_coef=0.1 / (128 * 128), optimizer=dict(type="SGD", lr=lr_c), save_dir=str(tmp_path / "cache")))
    m = MODELS.build(cfg)
    _randomize(m.diffusion.denoising, 9)
    m.train()
    g = torch.Generator().manual_seed(31)
    probe = torch.randn(3 * 6 * 16 * 16, generator=g)
    refreshes = []
    monkeypatch.setattr(m, "update_extra_state", lambda *a, **k: refreshes.append(1))

    def fake_loss(decoder, code, bits, target_rgbs, rays_o, rays_d, dt_gamma=0.0, return_decoder_loss=False, scale_num_ray=1.0, cfg=dict(),
                  perturb=True, **kw):
        assert decoder is m.decoder and decoder.training
        loss = (code.flatten(1) * probe).sum() * 0.01 + code.square().mean() + decoder.base_net[0].weight.sum() * 1e-3
        return target_rgbs * 0.5, loss, dict(pixel_loss=loss)

    monkeypatch.setattr(m, "loss", fake_loss)
    data = dict(scene_id=[1, 3], scene_name=["a", "b"], cond_imgs=torch.rand(2, 1, 8, 8, 3, generator=g), cond_poses=torch.eye(4).expand(2, 1, 4, 4),
                cond_intrinsics=torch.tensor([8.0, 8.0, 4.0, 4.0]).expand(2, 1, 4))
    opt = dict(diffusion=torch.optim.SGD(m.diffusion.parameters(), lr=0.1), decoder=torch.optim.SGD(m.decoder.parameters(), lr=0.1))
    torch.manual_seed(3)
    init = [m.get_init_code_(None) for _ in range(2)]
    torch.manual_seed(3)                                                      # load_cache draws the same initial codes
    unet_before = m.diffusion.denoising.out.conv.weight.detach().clone()
    dec_before = m.decoder.base_net[0].weight.detach().clone()
    sd = {k: v.clone() for k, v in m.diffusion.denoising.state_dict().items()}
    norm0 = float(m.diffusion.ddpm_loss.norm_factor)
    np.random.seed(11)
    # the draws train_step will make, in its order: init codes (above), timesteps (np.random), noise (torch CPU generator)
    state = torch.random.get_rng_state()
    [m.get_init_code_(None) for _ in range(2)]
    t_exp = m.diffusion.sampler(2)
    n_exp = torch.randn(2, 18, 16, 16)
    torch.random.set_rng_state(state); np.random.seed(11)
    out = m.train_step(data, opt)
    assert len(refreshes) == 1 + 1                                           # one refresh inside inverse_code (step 0 of E), one before the joint step
    assert not torch.equal(m.diffusion.denoising.out.conv.weight, unet_before)
    assert torch.allclose(m.decoder.base_net[0].weight, dec_before - 0.1 * 1e-3, atol=1e-7)        # decoder stepped exactly once
    for k in ("loss_ddpm_mse", "pixel_loss", "loss_decoder", "train_psnr", "code_rms"):
        assert k in out["log_vars"]
    assert out["num_samples"] == 2 and sorted(os.listdir(str(tmp_path / "cache"))) == ["a.pth", "b.pth"]
    assert m.cache[1] is not None and m.cache[0] is None and m.cache[1]["param"]["code_"].dtype == torch.float32
    # by hand
    den = lambda x, t: OD.unet_forward(sd, x, t, image_size=16, base_channels=32, channels_cfg=(1, 2), resblocks_per_downsample=1,
                                       num_heads=4, attention_res=(8,), norm_groups=8)
    tables = OD.schedule_tables(1000, "linear")
    w, _ = OD.snr_timestep_weights(tables, 0.5, "V")
    cs = [c.detach().clone().requires_grad_(True) for c in init]
    x0 = (torch.stack(cs).tanh() * 2).reshape(2, 18, 16, 16)
    norm1 = 0.999 * norm0 + 0.001 * float(x0.detach().square().mean())       # training mode: the running norm moves before it divides
    prior = OD.prior_loss_v(den, x0, t_exp, n_exp, tables, w, weight_scale=4.0, norm_factor=norm1)
    pg = torch.autograd.grad(prior, cs)
    for _ in range(E + 1):
        code = torch.stack(cs).tanh() * 2
        rg = torch.autograd.grad((code.flatten(1) * probe).sum() * 0.01 + code.square().mean(), cs)
        with torch.no_grad():
            for c, a, b in zip(cs, pg, rg):
                c -= lr_c * (a + b)
    for i, sid in enumerate((1, 3)):
        np.testing.assert_allclose(m.cache[sid]["param"]["code_"].numpy(), cs[i].detach().numpy(), rtol=0, atol=2e-6)

    # MultiSceneNeRF: no prior; E code-only iterations, then the joint one
    ms = MODELS.build(dict(type="MultiSceneNeRF", code_size=(3, 6, 16, 16), code_activation=dict(type="TanhCode", scale=2), grid_size=16,
                           decoder=cfg["decoder"], pixel_loss=cfg["pixel_loss"], cache_size=4,
                           train_cfg=dict(extra_scene_step=E, n_inverse_rays=2 ** 12, optimizer=dict(type="SGD", lr=lr_c)))).train()
    monkeypatch.setattr(ms, "update_extra_state", lambda *a, **k: None)
    monkeypatch.setattr(ms, "loss", lambda decoder, code, *a, **k: (a[1] * 0.5, (code.flatten(1) * probe).sum() * 0.01 + code.square().mean(), dict()))
    torch.manual_seed(5)
    init = [ms.get_init_code_(None) for _ in range(2)]
    torch.manual_seed(5)
    out = ms.train_step(data, dict(decoder=torch.optim.SGD(ms.decoder.parameters(), lr=0.1)))
    cs = [c.detach().clone().requires_grad_(True) for c in init]
    for _ in range(E + 1):
        code = torch.stack(cs).tanh() * 2
        rg = torch.autograd.grad((code.flatten(1) * probe).sum() * 0.01 + code.square().mean(), cs)
        with torch.no_grad():
            for c, b in zip(cs, rg):
                c -= lr_c * b
    np.testing.assert_allclose(ms.cache[3]["param"]["code_"].numpy(), cs[1].detach().numpy(), rtol=0, atol=2e-6)
    assert "loss" in out["log_vars"] and "train_psnr" in out["log_vars"]
