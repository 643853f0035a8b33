"""GPU parity of the reconstruction path as a WHOLE (r03 verdict, missing #3 / #4; BASELINE.json configs[2] and configs[4]):

  * a multi-step rendering-guided DDIM trajectory (UNet backward -> renderer backward -> density refresh with decay, every step) against
    ``oracle/diffusion.py::ddim_sample`` + ``oracle/guidance.py::guidance_loss`` with CPU autograd (diffusion_nerf.py:241-311,
    gaussian_diffusion.py:193-227);
  * config 5's combination -- guided Langevin corrections under bf16 autocast with fp16 planes -- against the fp32 path (PSNR floor);
  * the direction term of the fused shading kernel with decoder weights far outside the Xavier range;
  * the true-fp32 attention of the gradient path (SSDNERF_UNET_GRAD_ATT_KERNEL=0) stays reachable and agrees with the fp32-class kernels.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEC = dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], use_dir_enc=True,
           dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256)
UNET = dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=32, channels_cfg=[1, 1, 2], resblocks_per_downsample=1, dropout=0.0,
            use_scale_shift_norm=True, num_heads=4, attention_res=[32], norm_cfg=dict(type="GN", num_groups=8))
UNET_ORACLE = dict(image_size=128, base_channels=32, channels_cfg=(1, 1, 2), resblocks_per_downsample=1, num_heads=4, attention_res=(32,), norm_groups=8)
# a LOW-noise schedule (alpha_bar_999 = 0.99): x_T = a code + b z is a noised object, so every guided step renders real geometry.  With the
# configs' schedule and random UNet weights the predicted x0 holds no occupied voxels and guidance would have nothing to push against.
BETAS = dict(type="linear", beta_0=1e-6, beta_T=2e-5)


def _randomize(module, seed, scale=0.2):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (scale / max(1.0, p[0].numel() ** 0.5) if p.dim() > 1 else 0.1))


def _model(test_cfg, autocast_dtype=None, plane_dtype="float32", betas=BETAS, unet=UNET):
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.registry import MODELS
    m = MODELS.build(dict(
        type="DiffusionNeRF", code_size=(3, 6, 128, 128), code_reshape=(18, 128, 128), code_activation=dict(type="TanhCode", scale=2), grid_size=64,
        autocast_dtype=autocast_dtype,
        diffusion=dict(type="GaussianDiffusion", num_timesteps=1000, betas_cfg=dict(betas), denoising=dict(unet), denoising_mean_mode="V",
                       timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
                       ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight", data_info=dict(pred="v_t_pred", target="v_t"),
                                      weight_scale=4.0, scale_norm=True)),
        decoder=dict(DEC, plane_dtype=plane_dtype), decoder_use_ema=True, freeze_decoder=False, bg_color=1,
        pixel_loss=dict(type="MSELoss", loss_weight=20.0), reg_loss=dict(type="RegLoss", power=2, loss_weight=3e-3), cache_size=0, test_cfg=test_cfg))
    _randomize(m.diffusion_ema.denoising, 9)
    m.decoder_ema.load_state_dict(S.make_decoder_params(), strict=False)
    return m.cuda().eval()


def _noised_objects(seeds, g, tables_a_b):
    from ssdnerf_amd import synthetic as S
    a, b = tables_a_b
    code = torch.stack([S.make_triplane(sd) for sd in seeds])                  # |code| <= 2; the tests clip x0 to [-3, 3], which stays inactive
    z = torch.randn(code.shape, generator=g).clamp(-3, 3)
    return (a * code + b * z).float()


# ---------------------------------------------------------------------------------------------- guided DDIM, several steps, against the oracle
def test_guided_ddim_trajectory_matches_oracle():
    """Three rendering-guided DDIM steps of two scenes (diffusion_nerf.py:241-311: per step x0 <- UNet(x_t), density grid refreshed from x0 with decay
    0.9, train-branch render of the 64x64 conditioning view with jitter, pixel + regularisation loss, its gradient w.r.t. x_t THROUGH the UNet,
    x0 <- x0 - grad * b^(2-2w) a^(2w-1) * gain, DDIM update) with every random draw injected, against the CPU restatement (oracle UNet + C march /
    composite + PyTorch-CPU decode, autograd end to end).  Compared: the final codes, and the guided trajectory must differ from the unguided one
    by orders of magnitude more than the product differs from the oracle."""
    from oracle import diffusion as OD, guidance as OG, render as R
    from ssdnerf_amd import synthetic as S
    n_steps, S_, hw = 3, 2, 64
    gain = 0.4 * (2 ** 14)                                                       # (the chairs config's lambda_gd; with the cars value 3.2 * 2^14 single entries move by > 1)
    # clip_range [-3, 3] instead of the configs' [-2, 2]: the synthetic object code saturates at +-2, and an entry that sits within rounding of an
    # ACTIVE clip has gradient 0 on one side of the comparison and 1 on the other; the clip arithmetic itself is pinned by the DDIM fixtures
    cfg = dict(img_size=(hw, hw), num_timesteps=n_steps, clip_range=[-3, 3], density_thresh=0.1, dt_gamma_scale=0.5, n_inverse_rays=2 ** 14,
               loss_coef=0.1 / (128 * 128), guidance_gain=gain, cond_mode="guide")
    m = _model(cfg)
    tables = OD.schedule_tables(1000, "linear", beta_0=BETAS["beta_0"], beta_T=BETAS["beta_T"])
    g = torch.Generator().manual_seed(77)
    x_T = _noised_objects([31, 32], g, (tables["sqrt_alphas_bar"][999], tables["sqrt_one_minus_alphas_bar"][999]))
    targets = torch.rand(S_, hw * hw, 3, generator=g)
    marches = [torch.rand(S_, hw * hw, generator=g) for _ in range(n_steps)]
    jits = [torch.rand(64 ** 3, 3, generator=g) for _ in range(n_steps)]
    pose, intr = S.spiral_poses()[[64]], S.cars_intrinsics(hw, hw)
    params = S.make_decoder_params()

    # ---- product
    data = dict(cond_imgs=targets.reshape(S_, 1, hw, hw, 3).cuda(), cond_intrinsics=intr.cuda()[None, None].expand(S_, 1, -1),
                cond_poses=pose.cuda()[None].expand(S_, -1, -1, -1), noise=x_T.cuda())
    with torch.enable_grad():
        code, grid, bits = m.val_guide(data, guide_noises=[n.cuda() for n in marches], density_jitters=[j.cuda() for j in jits])
    with torch.no_grad():
        plain = m.diffusion_ema(m.code_diff_pr(x_T.cuda()), return_loss=False)                   # the same trajectory without guidance

    # ---- oracle
    sd = {k: v.cpu() for k, v in m.diffusion_ema.denoising.state_dict().items()}
    den = lambda x, t: OD.unet_forward(sd, x, t, **UNET_ORACLE)
    ro, rd = R.get_cam_rays(pose, intr[None], hw, hw)
    ro, rd = ro.reshape(-1, 3).numpy(), rd.reshape(-1, 3).numpy()
    dt_gamma = 0.5 / float(intr[:2].mean())
    grids = [np.zeros(64 ** 3, np.float32) for _ in range(S_)]
    k = [0]
    points = []

    def guide(x0):
        codes = x0.reshape(S_, 3, 6, 128, 128)
        total = 0
        for s in range(S_):
            b_s, _ = R.update_extra_state(params, codes[s].detach(), grids[s], jits[k[0]].numpy(), density_thresh=0.1, decay=0.9)
            loss, rec = OG.guidance_loss(params, codes[s], b_s, ro, rd, targets[s], marches[k[0]][s].numpy(), dt_gamma, scale_num_ray=hw * hw)
            points.append(rec["num_points"])
            total = total + loss
        k[0] += 1
        return total                                                             # = (mean over scenes) x num_scenes, diffusion_nerf.py:294

    want = OD.ddim_sample(den, x_T.reshape(S_, 18, 128, 128), tables, n_steps, clip_range=(-3, 3), grad_guide_fn=guide, guidance_gain=gain)
    want_plain = OD.ddim_sample(den, x_T.reshape(S_, 18, 128, 128), tables, n_steps, clip_range=(-3, 3))
    want, want_plain = want.reshape(S_, 3, 6, 128, 128), want_plain.reshape(S_, 3, 6, 128, 128)

    assert k[0] == n_steps and min(points) > 2000, points                       # every step marched real geometry
    assert float(want.abs().max()) < 2.99                                       # the clip never engaged: gradients flow everywhere on both sides
    moved = float((want - want_plain).abs().max())
    err = float((code.cpu() - want).abs().max())
    err_plain = float((plain.reshape(S_, 3, 6, 128, 128).cpu() - want_plain).abs().max())
    rms_moved = float((want - want_plain).square().mean().sqrt())
    rms_err = float((code.cpu() - want).square().mean().sqrt())
    print(f"guided trajectory: max|code - oracle| = {err:.3e} (rms {rms_err:.3e}); guidance moved the result by {moved:.3e} (rms {rms_moved:.3e}); "
          f"unguided max|code - oracle| = {err_plain:.3e}; samples per guided render {points}")
    assert moved > 1e-2, "guidance must move the trajectory measurably for this comparison to mean anything"
    # measured on the MI355X (r04): err 1.4e-6 / rms 3.2e-8 for a guidance displacement of 1.7e-1 / rms 1.5e-3, unguided 7e-7
    assert err_plain <= 2e-5
    assert err <= 2e-4 * moved + 1e-5 and rms_err <= 2e-4 * rms_moved + 1e-6
    for s in range(S_):                                                          # the occupancy state the objective owns, after its last refresh
        np.testing.assert_allclose(grid[s].cpu().numpy(), grids[s], rtol=2e-3, atol=2e-5)


# ---------------------------------------------------------------------------------------------- config 5: Langevin x guidance under bf16 + fp16 planes
def test_config5_guided_langevin_bf16_fp16_against_fp32():
    """BASELINE.json configs[4] (ssdnerf_chairs_recons1v.py:79-97 with bf16 UNet + fp16 triplane cache): rendering-guided DDIM with Langevin
    correction steps (gaussian_diffusion.py:242-262, 318-323), snr_weight_power 0.25, then one fine-tuning iteration, then a render -- once in fp32,
    once with ``autocast_dtype='bfloat16'`` and fp16 planes, same injected draws.  Mixed-precision tolerance: PSNR of every rendered view against
    the fp32 run, and the codes' relative distance."""
    from oracle import diffusion as OD
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.nerf import eval_psnr
    hw, S_ = 64, 2
    cfg = dict(img_size=(hw, hw), num_timesteps=3, clip_range=[-2, 2], density_thresh=0.1, dt_gamma_scale=0.5, n_inverse_rays=2 ** 14,
               loss_coef=0.1 / (128 * 128), guidance_gain=0.4 * (2 ** 14), snr_weight_power=0.25, cond_mode="guide_optim", n_inverse_steps=1,
               extra_scene_step=1, optimizer=dict(type="Adam", lr=0.005, weight_decay=0.0), lr_scheduler=dict(type="ExponentialLR", gamma=0.998),
               langevin_steps=2, langevin_delta=0.4)
    tables = OD.schedule_tables(1000, "linear", beta_0=BETAS["beta_0"], beta_T=BETAS["beta_T"])
    g = torch.Generator().manual_seed(78)
    x_T = _noised_objects([33, 34], g, (tables["sqrt_alphas_bar"][999], tables["sqrt_one_minus_alphas_bar"][999]))
    targets = torch.rand(S_, 1, hw, hw, 3, generator=g)
    n_eval = 3 + 2 * 2                                                             # DDIM steps + Langevin corrections behind the first two
    marches = [torch.rand(S_, hw * hw, generator=g) for _ in range(n_eval)]
    jits = [torch.rand(64 ** 3, 3, generator=g) for _ in range(n_eval)]
    opt_marches = [torch.rand(S_, hw * hw, generator=g) for _ in range(2)]
    opt_jits = [torch.rand(64 ** 3, 3, generator=g) for _ in range(1)]
    pose, intr = S.spiral_poses()[[64, 100]], S.cars_intrinsics(hw, hw)
    out = {}
    # "mixed": the shipped form -- under autocast the guided / fine-tuning UNet calls run the fp32-class gradient kernels (unet.DenoisingUnetMod.
    # grad_path_fp32_under_autocast), so only the fp16 planes separate it from fp32.  "mixed_bf16_unet": the eager modules under bf16 autocast
    # (SSDNERF_UNET_GRAD_AUTOCAST=1, the reference's arithmetic for this config) -- the real mixed-precision tolerance check.
    # r06: "mixed" is now the NATIVE bf16 gradient path (unet._ConvBf16Fn; DenoisingUnetMod.grad_path_bf16_native), held to the tolerance of the reference arithmetic;
    # "mixed_fp32_class_unet" the r04 form (fp32-class kernels under autocast), which only the fp16 planes separate from fp32
    for name, ac, pd, eager, native in (("fp32", None, "float32", False, False), ("mixed", "bfloat16", "float16", False, True),
                                        ("mixed_fp32_class_unet", "bfloat16", "float16", False, False), ("mixed_bf16_unet", "bfloat16", "float16", True, False)):
        m = _model(dict(cfg), autocast_dtype=ac, plane_dtype=pd)
        m.diffusion_ema.denoising.grad_path_bf16_native = native
        assert m.diffusion_ema.denoising._bf16_grad_path_ok()
        if eager:
            m.diffusion_ema.denoising.grad_path_fp32_under_autocast = False
        assert len(m.diffusion_ema.sampling_plan("ddim")) == n_eval
        data = dict(cond_imgs=targets.cuda(), cond_intrinsics=intr.cuda()[None, None].expand(S_, 1, -1), cond_poses=pose[:1].cuda()[None].expand(S_, -1, -1, -1),
                    noise=x_T.cuda(), test_poses=pose.cuda()[None].expand(S_, -1, -1, -1), test_intrinsics=intr.cuda()[None, None].expand(S_, 2, -1))
        torch.manual_seed(5); np.random.seed(5)                                    # the Langevin z draws and the prior's (t, noise): host generators
        res = m.val_step(data, guide_noises=[n.cuda() for n in marches], guide_density_jitters=[j.cuda() for j in jits],
                         march_noises=[n.cuda() for n in opt_marches], optim_density_jitters=[j.cuda() for j in opt_jits])
        out[name] = (res["code"].float().cpu(), res["pred_imgs"].float().cpu())
        assert bool(torch.isfinite(res["code"]).all())
    # measured on the MI355X (r04): 6.1e-6 / 2.2e-5 relative, every view equal after the k/255 rounding (the PSNR formula's epsilon caps at 60 dB);
    # the schedule of this test is the low-noise one (x0 = a x_t - 0.1 v), so the bf16 UNet's 2e-3 reaches the code divided by ten
    for name, rel_max, psnr_min in (("mixed", 1e-3, 40.0), ("mixed_fp32_class_unet", 1e-4, 45.0), ("mixed_bf16_unet", 1e-3, 40.0)):
        rel = float((out[name][0] - out["fp32"][0]).norm() / out["fp32"][0].norm())
        psnr = eval_psnr(out[name][1].flatten(0, 1), out["fp32"][1].flatten(0, 1))
        print(f"config 5 {name} vs fp32: code rel distance {rel:.3e}, PSNR per view {[round(float(p), 1) for p in psnr]}")
        assert rel < rel_max and float(psnr.min()) > psnr_min, (name, rel, psnr)


# ---------------------------------------------------------------------------------------------- direction term, weights outside the Xavier range
@pytest.mark.parametrize("scale", [8.0, 16.0])
def test_direction_term_with_large_decoder_weights(scale):
    """r03 verdict weak #1: the error of a reduced-precision direction term scales with |Wd.SH(d)| |Wc|, and trained decoders are not bounded by
    Xavier.  With ``dir_net`` and ``color_net`` scaled x8 / x16 the DEFAULT kernel (all six split products) must still meet the single-view
    tolerance against the oracle; the opt-in three-product form is measured beside it (and is allowed to miss it)."""
    from oracle import render as R
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
    from ssdnerf_amd.density import get_density
    params = S.make_decoder_params()
    params["dir_net.0.weight"] = params["dir_net.0.weight"] * scale
    params["dir_net.0.bias"] = torch.linspace(-0.5, 0.5, 64) * scale
    params["color_net.0.weight"] = params["color_net.0.weight"] * scale
    dec = TriPlaneDecoder(**{k: v for k, v in DEC.items() if k != "type"})
    dec.load_state_dict(params, strict=False)
    dec = dec.cuda().eval()
    assert dec.shade_dir_products == 6
    code = S.make_triplane(2021)
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g) for _ in range(4)]
    _, bits = get_density(dec, code.cuda()[None], 64, density_thresh=0.1, density_step=4, jitters=[j.cuda() for j in jit])
    from ssdnerf_amd import nerf
    poses, intr = S.spiral_poses()[[64, 200]].cuda()[None], S.cars_intrinsics(128, 128).cuda()[None, None].expand(1, 2, -1)
    ro, rd = (t[0].cpu() for t in nerf.get_cam_rays(poses, intr, 128, 128))
    planes = pack_triplanes(code.cuda()[None])
    errs = {}
    for n in (6, 3):
        dec.shade_dir_products = n
        out = dec.render_packed(planes, None, None, bits, 64, [0.0], 1e-4, bg_color=1.0, check_overflow=False, cams=(poses, intr, 128, 128))
        rgb = out["image"][0].cpu().numpy().reshape(2, -1, 3)
        errs[n] = max(float(np.abs(rgb[v] - R.render_eval(params, code, bits[0].cpu().numpy(), ro[v].reshape(-1, 3).numpy(), rd[v].reshape(-1, 3).numpy())[0]).max())
                      for v in range(2))
    dec.shade_dir_products = 6
    print(f"decoder weights x{scale:g}: max|rgb - oracle| = {errs[6]:.2e} with six direction products, {errs[3]:.2e} with three")
    assert errs[6] <= 2e-5, errs


# ---------------------------------------------------------------------------------------------- gradient-path attention: library fp32 reachable, kernels agree
def test_gradient_path_attention_library_fp32_is_reachable_and_agrees(monkeypatch):
    """The gradient path's attention runs on the hand-written fp32-CLASS kernels by default (three bf16-pair products per matrix product);
    ``SSDNERF_UNET_GRAD_ATT_KERNEL=0`` (module flag ``unet.GRAD_ATT_KERNEL``) keeps the library's true-fp32 scaled_dot_product_attention reachable.
    Both at the bench's attention shapes (8 scenes x 4 heads; T = 1024 / 256 / 64, head width 64 / 128 / 128): outputs and input gradients agree to
    the stated tolerance (2e-4 of the largest entry: the three-product class), and a batch x heads product beyond the kernels' grid limit takes the
    library path instead of raising."""
    from ssdnerf_amd import unet as U
    g = torch.Generator().manual_seed(3)
    for c, hw in ((256, 32), (512, 16), (512, 8)):
        blk = U.MultiHeadAttentionMod(c, num_heads=4).cuda()
        _randomize(blk, 11, scale=1.0)
        blk.requires_grad_(False)
        x = torch.randn(8, c, hw, hw, generator=g).cuda()
        res = {}
        for flag in (True, False):
            monkeypatch.setattr(U, "GRAD_ATT_KERNEL", flag)
            xi = x.clone().requires_grad_(True)
            y = blk(xi)
            (gx,) = torch.autograd.grad((y * torch.cos(y.detach())).sum(), xi)
            res[flag] = (y.detach(), gx)
        for a, b, what in ((res[True][0], res[False][0], "output"), (res[True][1], res[False][1], "input gradient")):
            tol = 2e-4 * float(b.abs().max())
            assert float((a - b).abs().max()) <= tol, (c, hw, what, float((a - b).abs().max()), tol)
    monkeypatch.setattr(U, "GRAD_ATT_KERNEL", True)
    assert U.attention_kernel_ok(torch.empty(1, 64, 3 * 256, device="cuda"), heads=4)
    assert not U.attention_kernel_ok(torch.empty(20000, 4, 3 * 32, device="cuda"), heads=4)      # 80 000 (batch, head) pairs > 65 535 grid rows


def test_captured_gradient_path_matches_the_eager_one():
    """Input-gradient calls with frozen weights replay a captured forward + backward once a signature has been seen often enough
    (``DenoisingUnetMod._grad_graph_call``): same kernels in the same order, so outputs and input gradients equal the eager path's to the rounding of
    the kernels' atomics; new inputs go through the static buffers; a changed weight drops the graph (the result follows the new weights); a call that
    asks for weight gradients stays eager.  (Tolerance 3e-5 of the largest entry since r05: the stem, the head and the stride-2 layers run the fp32-class
    kernels in both arms, six more layers whose split-K atomics order differently between two runs: 2.8e-6 observed on a 0.127 gradient.)"""
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODULES
    net = MODULES.build(dict(UNET, base_channels=64, channels_cfg=[1, 2, 2], attention_res=[32, 64])).cuda().eval()
    _randomize(net, 5, scale=1.0)
    net.requires_grad_(False)
    net.grad_graph_after = 2
    g = torch.Generator().manual_seed(1)

    def call(x, t, graph):
        net.grad_graph = graph
        xi = x.clone().requires_grad_(True)
        y = net(xi, t)
        (gx,) = torch.autograd.grad((y * torch.sin(y.detach())).sum(), xi)
        return y.detach().clone(), gx.detach().clone()

    def captured():
        return [e["fn"] is not None for e in net.__dict__.get("_grad_graphs", {}).values()]

    xs = [torch.randn(2, 18, 128, 128, generator=g).cuda() for _ in range(6)]
    ts = [torch.tensor([100 + 37 * i, 900 - 11 * i], device="cuda") for i in range(6)]
    for i in range(6):
        y_g, gx_g = call(xs[i], ts[i], True)
        assert captured() == [i >= 2], (i, captured())                                  # calls 0, 1 eager; call 2 captures and replays
        y_e, gx_e = call(xs[i], ts[i], False)
        for a, b, what in ((y_g, y_e, "output"), (gx_g, gx_e, "input gradient")):
            assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()), (i, what, float((a - b).abs().max()), float(b.abs().max()))
    with torch.no_grad():
        net.out.conv.weight.mul_(1.5); net.out.conv.bias.add_(0.25)                     # in place: the version counters move
    y_g, gx_g = call(xs[0], ts[0], True)
    assert captured() == [False]                                                        # dropped; counted from zero again
    y_e, gx_e = call(xs[0], ts[0], False)
    assert float((y_g - y_e).abs().max()) <= 3e-5 * float(y_e.abs().max()) and float((gx_g - gx_e).abs().max()) <= 3e-5 * float(gx_e.abs().max())
    for i in range(1, 4):
        y_g, gx_g = call(xs[i], ts[i], True)
    assert captured() == [True]
    y_e, gx_e = call(xs[3], ts[3], False)
    assert float((y_g - y_e).abs().max()) <= 3e-5 * float(y_e.abs().max()) and float((gx_g - gx_e).abs().max()) <= 3e-5 * float(gx_e.abs().max())
    net.out.conv.weight.requires_grad_(True)                                            # a weight gradient is wanted: eager, and it arrives
    net.grad_graph = True
    xi = xs[4].clone().requires_grad_(True)
    y = net(xi, ts[4])
    gx, gw = torch.autograd.grad(y.square().sum(), (xi, net.out.conv.weight))
    assert gw is not None and float(gw.abs().max()) > 0


def test_captured_gradient_path_is_guarded_against_re_entry():
    """r04 advisor: the captured gradient path keeps the activations of ONE forward in static graph memory.  (1) Two forwards of one signature before
    their backwards -- loss(unet(x1)) + loss(unet(x2)) -- must still give the right gradients: the second forward takes the eager path while the
    first one's backward is outstanding.  (2) A second backward through one replayed forward (retain_graph) raises instead of accumulating onto
    stale norm workspaces.  (3) A forward whose graph the caller dropped without running it does not block the captured path for ever.
    (4) The returned input gradient is a copy, not the graph's static buffer."""
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd.registry import MODULES
    net = MODULES.build(dict(UNET, base_channels=64, channels_cfg=[1, 2, 2], attention_res=[32, 64])).cuda().eval()
    _randomize(net, 5, scale=1.0)
    net.requires_grad_(False)
    net.grad_graph_after = 1
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(2, 18, 128, 128, generator=g).cuda() for _ in range(4)]
    t = torch.tensor([500, 20], device="cuda")

    def eager(x):
        net.grad_graph = False
        xi = x.clone().requires_grad_(True)
        (gx,) = torch.autograd.grad(net(xi, t).square().sum(), xi)
        net.grad_graph = True
        return gx
    net.grad_graph = True
    for x in xs[:2]:                                                                   # call 0 eager, call 1 captures
        xi = x.clone().requires_grad_(True)
        torch.autograd.grad(net(xi, t).square().sum(), xi)
    fn = next(iter(net.__dict__["_grad_graphs"].values()))["fn"]
    assert fn is not None and not fn.busy()
    # (1) interleaved forwards
    x1, x2 = xs[2].clone().requires_grad_(True), xs[3].clone().requires_grad_(True)
    y1 = net(x1, t)
    assert fn.busy()
    y2 = net(x2, t)                                                                    # must not replay over y1's activations
    g1, g2 = torch.autograd.grad(y1.square().sum() + y2.square().sum(), (x1, x2))
    for got, x in ((g1, xs[2]), (g2, xs[3])):
        ref = eager(x)
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert not fn.busy()
    # (4) the gradient handed out is not the static buffer
    assert g1.data_ptr() != fn.gx.data_ptr() and g2.data_ptr() != fn.gx.data_ptr()
    # (2) double backward through one replay
    x3 = xs[0].clone().requires_grad_(True)
    loss = net(x3, t).square().sum()
    torch.autograd.grad(loss, x3, retain_graph=True)
    with pytest.raises(RuntimeError, match="ONE forward"):
        torch.autograd.grad(loss, x3)
    # (3) a dropped graph releases the path
    x4 = xs[1].clone().requires_grad_(True)
    y4 = net(x4, t)
    assert fn.busy()
    del y4
    x5 = xs[2].clone().requires_grad_(True)
    (g5,) = torch.autograd.grad(net(x5, t).square().sum(), x5)                         # replays (the pending forward's output is gone)
    assert fn.serial >= 4 and not fn.busy()
    ref = eager(xs[2])
    assert float((g5 - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_gradient_path_with_pre_split_dy_matches_the_on_the_fly_split(monkeypatch):
    """r04: the second norm of a residual block writes its dx PRE-SPLIT for the first convolution's backward-data kernel (one decision shared by the two
    autograd functions through a flag dict; ``SSDNERF_UNET_GRAD_SPLIT_DY=0`` / ``_Conv2d.grad_split_dy`` restores the on-the-fly split).  Same hi / lo
    terms and the same products: the UNet's input gradient agrees to fp32 rounding (relative L2 error <= 1e-5), on a net with large (two-group kernel) and small (generic kernel) layers,
    and the pre-split form is really taken."""
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd import unet as U, unet_fast as UF
    from ssdnerf_amd.registry import MODULES
    net = MODULES.build(dict(UNET, base_channels=128, channels_cfg=[1, 2, 2], attention_res=[32])).cuda().eval()
    _randomize(net, 7, scale=1.0)
    net.requires_grad_(False)
    net.grad_graph = False
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 18, 128, 128, generator=g).cuda()
    t = torch.tensor([700, 40], device="cuda")
    calls = {"split": 0, "pass": 0}
    real, real_pass = UF.group_norm_nhwc_backward, UF.split_f32_nhwc
    monkeypatch.setattr(UF, "group_norm_nhwc_backward", lambda *a, **k: (calls.__setitem__("split", calls["split"] + int(bool(k.get("split_out")))), real(*a, **k))[1])
    monkeypatch.setattr(UF, "split_f32_nhwc", lambda *a, **k: (calls.__setitem__("pass", calls["pass"] + 1), real_pass(*a, **k))[1])
    res = {}
    for on in (True, False):
        monkeypatch.setattr(U._Conv2d, "grad_split_dy", on)
        calls["split"] = calls["pass"] = 0
        xi = x.clone().requires_grad_(True)
        y = net(xi, t)
        (gx,) = torch.autograd.grad((y * torch.cos(y.detach())).sum(), xi)
        res[on] = (y.detach(), gx, calls["split"], calls["pass"])
    assert res[True][2] >= 6 and res[False][2] == 0, (res[True][2], res[False][2])      # every residual block whose conv_1 a pre-split kernel takes
    assert res[True][3] >= 2 and res[False][3] == 0, (res[True][3], res[False][3])      # the large second convolutions: one split pass in front of the pre-split kernel
    fscale = float(res[False][0].abs().max())                                            # (the forward is untouched: equal up to the order of its kernels' atomics)
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-5 * fscale
    err, scale = float((res[True][1] - res[False][1]).abs().max()), float(res[False][1].abs().max())
    # fp32 rounding through ~40 layers of backward: the small layers' pre-split kernel sums in another order, and the kernels' atomics differ run to run
    rel_l2 = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    # (r05: 2e-5 -- the stem, the head and the four stride-2 layers run the fp32-class kernels in both arms now, six more layers whose split-K atomics
    # order differently from run to run; measured 1.3e-5 between the arms, 1.0e-5 before)
    assert err <= 5e-5 * scale and rel_l2 <= 2e-5, (err, scale, rel_l2)


def test_host_noise_keeps_the_seeded_sequence_and_val_optim_its_draw_order():
    """The prior loss's noise (and the Langevin / ancestral noise) is a HOST draw, as in the reference (mmgen ``_get_noise_batch``): a seeded CPU generator
    reproduces a trajectory on any device.  r04 moved the draws to where the device has work queued (pinned buffer + asynchronous upload; the next
    fine-tuning iteration's draw behind this iteration's UNet launches): the VALUES and their ORDER must be those of plain ``torch.randn`` calls --
    checked on the helper itself and end to end: ``val_optim`` with implicit draws equals ``val_optim`` with the same seeded draws injected."""
    from ssdnerf_amd import diffusion as D, synthetic as S
    like = torch.empty(2, 18, 128, 128, device="cuda")
    torch.manual_seed(11)
    want = [torch.randn(2, 18, 128, 128) for _ in range(4)]
    torch.manual_seed(11)
    got = [D._host_noise(like) for _ in range(4)]                                # four draws: both pinned buffers are reused once
    torch.cuda.synchronize()
    assert all(torch.equal(a, b.cpu()) for a, b in zip(want, got))

    hw, S_, n_outer = 64, 2, 3
    cfg = dict(img_size=(hw, hw), num_timesteps=n_outer, clip_range=[-3, 3], density_thresh=0.1, dt_gamma_scale=0.5, n_inverse_rays=2 ** 12, loss_coef=0.1 / (128 * 128),
               cond_mode="optim", n_inverse_steps=n_outer, extra_scene_step=1, optimizer=dict(type="Adam", lr=0.005, weight_decay=0.0),
               lr_scheduler=dict(type="ExponentialLR", gamma=0.998))
    m = _model(cfg)
    g = torch.Generator().manual_seed(5)
    data = dict(cond_imgs=torch.rand(S_, 1, hw, hw, 3, generator=g).cuda(), cond_intrinsics=S.cars_intrinsics(hw, hw).cuda()[None, None].expand(S_, 1, -1),
                cond_poses=S.spiral_poses()[[64]].cuda()[None].expand(S_, -1, -1, -1))
    code0 = torch.stack([S.make_triplane(41 + i) for i in range(S_)]).cuda()

    def run(inject):
        torch.manual_seed(23); np.random.seed(23)
        noises = None
        if inject:
            noises = [torch.randn(S_, 18, 128, 128).cuda() for _ in range(n_outer)]      # the draws val_optim would make, in its order, from the same seed
            torch.manual_seed(23)                                                        # (the DEVICE generator back to the state the other run starts from)
        code_ = m.code_activation.inverse(code0).clone().requires_grad_(True)
        out, _, _ = m.val_optim(data, code_=code_, prior_noises=noises)
        return out

    a, b = run(False), run(True)
    moved = float((a - code0).abs().max())
    assert moved > 1e-3                                                                  # the optimisation did something
    assert float((a - b).abs().max()) <= 2e-4 * moved + 1e-6, (float((a - b).abs().max()), moved)
