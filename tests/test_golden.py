"""Golden fixtures produced by EXECUTING THE REFERENCE'S PYTHON (tests/golden/make_golden.py: VolumeRenderer.forward,
TriPlaneDecoder.point_decode, get_cam_rays, GaussianDiffusion.ddim_sample ... imported from /root/reference, with the
reference's own CUDA kernels compiled for the CPU behind `_raymarching`/`_shencoder`).

CPU half: the oracle and the product's host-side code reproduce the fixtures.  GPU half: the HIP path does."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


@pytest.fixture(scope="module")
def scene():
    from oracle import render as R
    from ssdnerf_amd import synthetic as S
    params, code = S.make_decoder_params(), S.make_triplane()
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g).numpy() for _ in range(2)]
    _, bits, _ = R.get_density(params, code, jit, density_thresh=0.1)
    return dict(params=params, code=code, bits=bits)


# ------------------------------------------------------------------------------------------------ CPU
def test_cam_rays_match_reference():
    from oracle import render as R
    from ssdnerf_amd import nerf
    f = load("cam_rays_64.npz")
    pose, intr = torch.from_numpy(f["pose"]), torch.from_numpy(f["intrinsics"])
    for fn in (R.get_cam_rays, nerf.get_cam_rays):
        ro, rd = fn(pose, intr, 64, 64)
        np.testing.assert_allclose(ro.reshape(1, -1, 3).numpy(), f["rays_o"], rtol=0, atol=0)
        np.testing.assert_allclose(rd.reshape(1, -1, 3).numpy(), f["rays_d"], rtol=0, atol=3e-7)      # the 3x3 rotation is a BLAS call: a few ulp between hosts


def test_oracle_decode_matches_reference(scene):
    from oracle.decoder import point_decode
    f = load("point_decode.npz")
    sig, rgb = point_decode(scene["params"], scene["code"], torch.from_numpy(f["xyzs"]), torch.from_numpy(f["dirs"]))
    np.testing.assert_allclose(sig.numpy(), f["sigmas"], rtol=2e-6, atol=0)
    np.testing.assert_allclose(sig.numpy(), f["sigmas_density_only"], rtol=2e-6, atol=0)
    np.testing.assert_allclose(rgb.numpy(), f["rgbs"], rtol=0, atol=5e-7)


@pytest.mark.parametrize("tag", ["dtg0", "dtg"])
def test_oracle_eval_render_matches_reference_loop(scene, tag):
    from oracle import render as R
    f, rays = load(f"render_eval_64_{tag}.npz"), load("cam_rays_64.npz")
    tr = {}
    rgb, depth, ws = R.render_eval(scene["params"], scene["code"], scene["bits"], rays["rays_o"][0], rays["rays_d"][0], dt_gamma=float(f["dt_gamma"]),
                                   trace=tr)
    assert np.array_equal(np.array(tr["iterations"], np.int32), f["iterations"])          # (n_alive, n_step) per loop iteration: bit-exact
    np.testing.assert_allclose(ws, f["weights_sum"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(depth, f["depth"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(rgb, f["rgb_bg1"], rtol=0, atol=1e-6)
    assert float(f["weights_sum"].max()) > 0.9 and len(f["iterations"]) >= 4


def test_oracle_train_render_and_gradient_match_reference(scene):
    from oracle import guidance as OG
    f, rays = load("render_train_64.npz"), load("cam_rays_64.npz")
    sub = f["ray_subset"]
    ro, rd = rays["rays_o"][0][sub], rays["rays_d"][0][sub]
    code = scene["code"].clone().requires_grad_(True)
    loss, rec = OG.guidance_loss(scene["params"], code, scene["bits"], ro, rd, torch.from_numpy(f["target"][0]), np.zeros(len(sub), np.float32), 0.0038095,
                                 loss_weight=20.0, loss_coef=1e9, reg_weight=0.0)     # loss_coef -> scale 1: plain 20 * MSE * 3 / 3 ...
    # the fixture's loss is mean((rgb - target)^2) * 20 (no scale factor, no RegLoss): compare the rendered colours and the gradient shape
    np.testing.assert_allclose(rec["out_rgbs"].numpy(), f["image"][0] + (1 - f["weights_sum"][0][:, None]), rtol=0, atol=2e-6)
    want = float(((rec["out_rgbs"] - torch.from_numpy(f["target"][0])) ** 2).mean() * 20.0)
    assert abs(want - float(f["loss"])) < 1e-5
    (g,) = torch.autograd.grad(loss, code)
    g = g / 3.0                                                                          # guidance_loss multiplies the pixel loss by 3*scale
    np.testing.assert_allclose(g[:, :, ::16, ::16].numpy(), f["grad_code_sample"][...], rtol=2e-4, atol=1e-7 + 2e-4 * float(f["grad_code_absmax"]))


def _toy_diffusion(f):
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODULES
    from ssdnerf_amd.diffusion import GaussianDiffusion

    if "ToyDenoiser" not in MODULES:
        @MODULES.register_module()
        class ToyDenoiser(torch.nn.Module):
            def __init__(self, num_classes=0, num_timesteps=1000):
                super().__init__()
                self.conv = torch.nn.Conv2d(18, 18, 3, padding=1)

            def forward(self, x_t, t, concat_cond=None):
                return torch.tanh(self.conv(x_t)) * (1 + t.float().view(-1, 1, 1, 1) / 1000)
    d = GaussianDiffusion(denoising=dict(type="ToyDenoiser"), ddpm_loss=dict(type="DDPMMSELossMod"), betas_cfg=dict(type="linear"), num_timesteps=1000,
                          timestep_sampler=dict(type="SNRWeightedTimeStepSampler"), denoising_mean_mode="V", test_cfg=dict(num_timesteps=10, clip_range=[-2, 2]))
    with torch.no_grad():
        d.denoising.conv.weight.copy_(torch.from_numpy(f["conv_weight"]))
        d.denoising.conv.bias.copy_(torch.from_numpy(f["conv_bias"]))
    return d.eval()


def test_ddim_matches_reference_gaussian_diffusion():
    f = load("ddim.npz")
    d = _toy_diffusion(f)
    for k in ("betas", "alphas_bar", "alphas_bar_prev", "sqrt_alphas_bar", "sqrt_one_minus_alphas_bar", "tilde_betas_t"):
        np.testing.assert_array_equal(getattr(d, k), f[k])                                # float64 tables: bit-exact
    for n in (50, 75, 10):
        assert np.array_equal(d.ddim_timesteps(n).numpy(), f[f"timesteps_{n}"])
    noise = torch.from_numpy(f["noise"])
    with torch.no_grad():
        traj = d.ddim_sample(noise.clone(), save_intermediates=True)
    np.testing.assert_allclose(np.stack([t.numpy() for t in traj[0::2]]), f["x0_steps"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.stack([t.numpy() for t in traj[1::2]]), f["xt_steps"], rtol=0, atol=2e-6)
    target = torch.from_numpy(f["target"])
    d.test_cfg["guidance_gain"] = 2.0
    for p in d.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        guided = d.ddim_sample(noise.clone(), grad_guide_fn=lambda x0: ((x0 - target) ** 2).mean() * 5.0)
    np.testing.assert_allclose(guided.numpy(), f["guided_final"], rtol=0, atol=5e-6)
    # and the oracle's straight-line DDIM agrees with the reference trajectory too
    from oracle import diffusion as OD
    den = lambda x, t: d.denoising(x, t)
    want = OD.ddim_sample(den, noise, OD.schedule_tables(1000, "linear"), 10, clip_range=(-2, 2))
    np.testing.assert_allclose(want.numpy(), f["xt_steps"][-1], rtol=0, atol=5e-6)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def decoder(scene):
    from ssdnerf_amd.decoders import TriPlaneDecoder
    dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
    dec.load_state_dict(scene["params"], strict=False)
    return dec.cuda().eval()


@pytest.mark.gpu
def test_hip_decode_matches_reference_fixture(scene, decoder):
    f = load("point_decode.npz")
    with torch.no_grad():
        sig, rgb, _ = decoder.point_decode([torch.from_numpy(f["xyzs"]).cuda()], [torch.from_numpy(f["dirs"]).cuda()], scene["code"].cuda()[None])
    np.testing.assert_allclose(sig.cpu().numpy(), f["sigmas"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(rgb.cpu().numpy(), f["rgbs"], rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["dtg0", "dtg"])
@pytest.mark.parametrize("pipeline", ["queue_mfma", "queue", "single"])
def test_hip_fused_render_matches_reference_fixture(scene, decoder, tag, pipeline):
    from ssdnerf_amd.decoders import pack_triplanes
    f, rays = load(f"render_eval_64_{tag}.npz"), load("cam_rays_64.npz")
    planes = pack_triplanes(scene["code"].cuda()[None])
    decoder.fused_pipeline = pipeline
    try:
        out = decoder.render_packed(planes, torch.from_numpy(rays["rays_o"]).cuda(), torch.from_numpy(rays["rays_d"]).cuda(),
                                    torch.from_numpy(scene["bits"]).cuda()[None], 64, [float(f["dt_gamma"])], 1e-4, bg_color=1.0, want_counts=True)
    finally:
        decoder.fused_pipeline = "queue_mfma"
    np.testing.assert_allclose(out["image"][0].cpu().numpy(), f["rgb_bg1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["depth"][0].cpu().numpy(), f["depth"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["weights_sum"][0].cpu().numpy(), f["weights_sum"], rtol=0, atol=1e-5)
    # stepwise loop over the unfused HIP operators: the reference's own iteration history
    if pipeline == "queue_mfma":
        decoder.render_mode = "stepwise"
        try:
            with torch.no_grad():
                decoder(torch.from_numpy(rays["rays_o"]).cuda(), torch.from_numpy(rays["rays_d"]).cuda(), scene["code"].cuda()[None],
                        torch.from_numpy(scene["bits"]).cuda()[None], 64, dt_gamma=float(f["dt_gamma"]), perturb=False)
        finally:
            decoder.render_mode = "fused"
        hist = np.array(decoder.last_render_stats["iterations"][0], np.int32)
        assert hist.shape == f["iterations"].shape and np.array_equal(hist[:, 1], f["iterations"][:, 1])
        assert int(np.abs(hist[:, 0] - f["iterations"][:, 0]).max()) <= 2           # rays within float noise of T_thresh


@pytest.mark.gpu
def test_hip_train_branch_matches_reference_fixture(scene, decoder):
    f, rays = load("render_train_64.npz"), load("cam_rays_64.npz")
    sub = torch.from_numpy(f["ray_subset"]).long()
    ro, rd = torch.from_numpy(rays["rays_o"])[:, sub].cuda(), torch.from_numpy(rays["rays_d"])[:, sub].cuda()
    code = scene["code"].cuda()[None].requires_grad_(True)
    decoder.train(True)
    try:
        for p in decoder.parameters():
            p.requires_grad_(False)
        res = decoder(ro, rd, code, torch.from_numpy(scene["bits"]).cuda()[None], 64, dt_gamma=0.0038095, perturb=False)
        rgbs = res["image"] + 1.0 * (1 - res["weights_sum"].unsqueeze(-1))
        loss = ((rgbs - torch.from_numpy(f["target"]).cuda()) ** 2).mean() * 20.0
        (g,) = torch.autograd.grad(loss, code)
    finally:
        decoder.train(False)
    np.testing.assert_allclose(res["weights_sum"].detach().cpu().numpy(), f["weights_sum"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(res["image"].detach().cpu().numpy(), f["image"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(res["depth"].detach().cpu().numpy(), f["depth"], rtol=0, atol=1e-4)
    assert abs(float(loss) - float(f["loss"])) < 1e-4
    np.testing.assert_allclose(g[0, :, :, ::16, ::16].cpu().numpy(), f["grad_code_sample"], rtol=5e-4, atol=1e-7 + 5e-4 * float(f["grad_code_absmax"]))


@pytest.mark.gpu
def test_hip_ddim_fused_step_matches_reference_fixture():
    f = load("ddim.npz")
    d = _toy_diffusion(f).cuda()
    with torch.no_grad():
        traj = d.ddim_sample(torch.from_numpy(f["noise"]).cuda(), save_intermediates=True)
    np.testing.assert_allclose(np.stack([t.cpu().numpy() for t in traj[1::2]]), f["xt_steps"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(np.stack([t.cpu().numpy() for t in traj[0::2]]), f["x0_steps"], rtol=0, atol=1e-5)


# ------------------------------------------------------------------------------------------------ fixtures of make_golden_recons.py
def _toy_recons_diffusion(f):
    import torch.nn as nn
    from ssdnerf_amd.registry import MODULES
    from ssdnerf_amd.diffusion import GaussianDiffusion

    if "ToyDenoiser2" not in MODULES:
        @MODULES.register_module()
        class ToyDenoiser2(nn.Module):
            def __init__(self, num_classes=0, num_timesteps=1000):
                super().__init__()
                self.conv = nn.Conv2d(18, 18, 3, padding=1)

            def forward(self, x_t, t, concat_cond=None):
                return torch.tanh(self.conv(x_t)) * (1 + t.float().view(-1, 1, 1, 1) / 1000)

    d = GaussianDiffusion(denoising=dict(type="ToyDenoiser2"), betas_cfg=dict(type="linear"), num_timesteps=1000, denoising_mean_mode="V",
                          timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
                          ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight", data_info=dict(pred="v_t_pred", target="v_t"),
                                         weight_scale=4.0, scale_norm=True),
                          test_cfg=dict(num_timesteps=4, clip_range=[-2, 2], langevin_steps=2, langevin_delta=0.4))
    with torch.no_grad():
        d.denoising.conv.weight.copy_(torch.from_numpy(f["conv_weight"]))
        d.denoising.conv.bias.copy_(torch.from_numpy(f["conv_bias"]))
        d.ddpm_loss.norm_factor.fill_(1.7)
    return d.eval()


def test_prior_loss_sampler_and_langevin_match_reference_gaussian_diffusion():
    f = load("recons.npz")
    d = _toy_recons_diffusion(f)
    np.testing.assert_array_equal(d.sampler.weight.numpy(), f["snr_weight"])
    np.testing.assert_allclose(np.asarray(d.sampler.prob), f["snr_prob"], rtol=1e-15)
    # forward_train with the reference's host-side draws: same timesteps, same noise, same x_t, same loss and gradient
    x0 = torch.from_numpy(f["prior_x0"]).requires_grad_(True)
    np.random.seed(5); torch.manual_seed(5)
    t = d.sampler(3)
    noise = torch.randn(3, 18, 16, 16)
    assert np.array_equal(t.numpy(), f["prior_t"]) and np.array_equal(noise.numpy(), f["prior_noise"])
    x_t, _, _ = d.q_sample(x0.detach(), t, noise)
    np.testing.assert_allclose(x_t.numpy(), f["prior_x_t"], rtol=0, atol=1e-6)
    np.random.seed(5); torch.manual_seed(5)
    loss, _ = d(x0, return_loss=True, cfg=d.test_cfg)
    (gx,) = torch.autograd.grad(loss, x0)
    assert abs(float(loss.detach()) - float(f["prior_loss"])) <= 2e-6 * abs(float(f["prior_loss"]))
    np.testing.assert_allclose(gx.numpy(), f["prior_grad"], rtol=0, atol=1e-5 * float(np.abs(f["prior_grad"]).max()))
    # DDIM with 2 Langevin corrections per step, unguided and guided
    noise0, target = torch.from_numpy(f["noise"]), torch.from_numpy(f["target"])
    for p in d.parameters():
        p.requires_grad_(False)
    torch.manual_seed(77)
    with torch.no_grad():
        lv = d.ddim_sample(noise0.clone())
    np.testing.assert_allclose(lv.numpy(), f["langevin_final"], rtol=0, atol=5e-6)
    d.test_cfg["guidance_gain"] = 2.0
    torch.manual_seed(78)
    with torch.no_grad():
        lv_g = d.ddim_sample(noise0.clone(), grad_guide_fn=lambda x0_: ((x0_ - target) ** 2).mean() * 5.0)
    np.testing.assert_allclose(lv_g.numpy(), f["langevin_guided_final"], rtol=0, atol=1e-5)


def test_code_activations_match_reference_classes():
    from ssdnerf_amd.models import NormalizedTanhCode, TanhCode
    f = load("recons.npz")
    c_ = torch.from_numpy(f["act_in"])
    nt = NormalizedTanhCode(mean=0.0, std=0.5, clip_range=2).eval()
    y = nt(c_)
    np.testing.assert_allclose(y.numpy(), f["ntanh_eval"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(nt.inverse(y).numpy(), f["ntanh_inverse"], rtol=1e-5, atol=1e-5)
    nt.train()
    y2 = nt(c_, update_stats=True)                                           # running statistics move only in training mode
    np.testing.assert_allclose(nt.running_mean.numpy(), f["ntanh_running_mean"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(nt.running_var.numpy(), f["ntanh_running_var"], rtol=1e-6)
    np.testing.assert_allclose(y2.numpy(), f["ntanh_train"], rtol=0, atol=1e-6)
    t2 = TanhCode(scale=2)
    np.testing.assert_allclose(t2(c_).numpy(), f["tanh2"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(t2.inverse(t2(c_)).numpy(), f["tanh2_inverse"], rtol=1e-5, atol=1e-5)


def _same_tree(a, b, path=""):
    if isinstance(a, torch.Tensor):
        assert isinstance(b, torch.Tensor) and a.dtype == b.dtype and a.shape == b.shape, path
        assert torch.equal(a, b), path
    elif isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), (path, set(a), set(b))
        for k in a:
            _same_tree(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same_tree(x, y, f"{path}[{i}]")
    else:
        assert a == b, path


def test_scene_cache_casting_rules_match_reference_helpers():
    """fp16 code (clamped to +-65504, never inf) + bf16 optimizer moments, grids / bitfields / step untouched; in-place refresh of an existing
    entry; restoring a bf16 state into an fp32 optimizer keeps the LIVE learning rate -- bit for bit what the reference's helpers produce."""
    import copy
    from ssdnerf_amd import scene_cache as SC
    fx = torch.load(os.path.join(G, "scene_cache.pt"), map_location="cpu", weights_only=False)
    live, live2 = fx["live"], fx["live2"]
    c16 = SC.out_dict_to(live, device="cpu", code_dtype=torch.float16, optimizer_dtype=torch.bfloat16)
    _same_tree(c16, fx["cached16"])
    assert c16["param"]["code_"].dtype == torch.float16 and float(c16["param"]["code_"].max()) == 65504.0
    assert c16["param"]["density_grid"].dtype == torch.float16 and c16["param"]["density_bitfield"].dtype == torch.uint8
    st = next(iter(c16["optimizer"]["state"].values()))
    assert st["exp_avg"].dtype == torch.bfloat16 and st["exp_avg_sq"].dtype == torch.bfloat16 and st["step"].dtype == torch.float32
    _same_tree(SC.out_dict_to(live, device="cpu", code_dtype=torch.float32, optimizer_dtype=torch.float32), fx["cached32"])
    refreshed = copy.deepcopy(c16)
    keep = refreshed["param"]["code_"]
    for key, val in live2["param"].items():
        SC.load_tensor_to_dict(refreshed["param"], key, val, device="cpu", dtype=torch.float16)
    SC.optimizer_state_copy(live2["optimizer"], refreshed["optimizer"], device="cpu", dtype=torch.bfloat16)
    assert refreshed["param"]["code_"] is keep                                # refreshed in place
    _same_tree(refreshed, fx["refreshed16"])
    code_b = live2["param"]["code_"].clone().requires_grad_(True)
    opt = torch.optim.Adam([code_b], lr=0.5)
    SC.optimizer_set_state(opt, refreshed["optimizer"])
    s = opt.state[code_b]
    assert s["exp_avg"].dtype == torch.float32 and torch.equal(s["exp_avg"], fx["restored"]["exp_avg"]) and torch.equal(s["exp_avg_sq"], fx["restored"]["exp_avg_sq"])
    assert float(s["step"]) == float(fx["restored"]["step"]) == 5.0 and opt.param_groups[0]["lr"] == fx["restored"]["lr"] == 0.5
    opt.zero_grad(); (code_b ** 2).sum().backward(); opt.step()                 # and the optimizer keeps stepping from there
    assert float(opt.state[code_b]["step"]) == 6.0


def test_ddpm_ancestral_sampler_matches_reference():
    f = load("recons.npz")
    d = _toy_recons_diffusion(f)
    d.test_cfg.update(num_timesteps=5, langevin_steps=0)
    noise0 = torch.from_numpy(f["noise"])
    for mode in ("FIXED_LARGE", "FIXED_SMALL"):
        d.denoising_var_mode = mode
        d.sample_method = "ddpm"
        torch.manual_seed(91)
        with torch.no_grad():
            got = d(noise0.clone(), return_loss=False)
        np.testing.assert_allclose(got.numpy(), f[f"ddpm_{mode.lower()}"], rtol=0, atol=5e-6)
