#!/usr/bin/env python
"""tests/golden/make_golden.py -- generates the golden fixtures in this directory by EXECUTING THE REFERENCE'S OWN
PYTHON (run it in the build container, where /root/reference exists; the fixtures travel, the reference does not).

What runs for real (imported from /root/reference, unmodified):
    lib/ops/raymarching/raymarching.py     operator wrappers (allocation, align padding, autograd Functions)
    lib/ops/shencoder/sphere_harmonics.py  SHEncoder
    lib/ops/activation.py                  TruncExp
    lib/models/decoders/{base_volume_renderer,triplane_decoder}.py   VolumeRenderer.forward (both branches), point_decode
    lib/core/utils/nerf_utils.py           get_cam_rays
    lib/models/diffusions/gaussian_diffusion.py   schedules, pred_x_0 (+guidance), p_sample_ddim, ddim_sample

What is substituted, because it cannot exist in this container (no nvcc, no mmcv/mmgen, no GPU):
    _raymarching / _shencoder   the CUDA extensions -> CPU implementations backed by oracle/_ref when built (the reference's
                                own kernels compiled for the CPU) and by the C oracle otherwise (they agree bit for bit,
                                tests/test_oracle_vs_reference.py)
    mmcv / mmgen                the handful of helpers those files import (Registry, xavier_init, var_to_tensor ...),
                                restated below from SURVEY.md Appendix A
    Tensor.cuda()               identity (the wrappers force .cuda())
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REF", "/root/reference")
sys.path.insert(0, ROOT)


def _install_stubs(native="oracle"):
    """native="oracle": `_raymarching` / `_shencoder` backed by the CPU oracle (fixture generation, no GPU).
    native="dropin": the product's drop-in modules (ssdnerf_amd/dropin) are left to resolve those two names -- the reference's own wrappers
    and renderer then run on the MI355X library (tools/option_a_on_gpu.py); only the mmcv / mmgen helpers are stubbed."""
    if native == "dropin":
        ops, backend_name = None, "ssdnerf_amd/dropin over libssdnerf_hip.so"
        sys.path.insert(0, os.path.join(ROOT, "ssdnerf_amd", "dropin"))
    else:
        import oracle
        ops = oracle.ref_ops("fma") or oracle.ops()
        backend_name = "oracle/_ref (reference kernels on CPU)" if oracle.ref_ops("fma") else "C oracle"
        torch.Tensor.cuda = lambda self, *a, **k: self        # the wrappers call .cuda() unconditionally
    if not hasattr(np, "cumproduct"):
        np.cumproduct = np.cumprod                            # numpy >= 2 dropped the alias the reference uses (gaussian_diffusion.py:134)

    def npv(t):
        return t.detach().numpy()

    rm = types.ModuleType("_raymarching")

    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        a, b = ops.near_far_from_aabb(npv(rays_o), npv(rays_d), npv(aabb), min_near)
        nears.copy_(torch.from_numpy(a)); fars.copy_(torch.from_numpy(b))

    def morton3D(coords, N, indices):
        indices.copy_(torch.from_numpy(ops.morton3D(npv(coords))))

    def morton3D_invert(indices, N, coords):
        coords.copy_(torch.from_numpy(ops.morton3D_invert(npv(indices))))

    def packbits(grid, N, thresh, bitfield):
        bitfield.copy_(torch.from_numpy(ops.packbits(npv(grid.float()).reshape(-1), float(thresh))))

    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
        cnt = npv(counter).copy()
        x, d, dl, r, cnt = ops.march_rays_train(npv(rays_o), npv(rays_d), npv(grid), bound, dt_gamma, max_steps, C, H, npv(nears), npv(fars),
                                                npv(noises), M=M, counter=cnt)
        xyzs.copy_(torch.from_numpy(x)); dirs.copy_(torch.from_numpy(d)); deltas.copy_(torch.from_numpy(dl))
        rays.copy_(torch.from_numpy(r)); counter.copy_(torch.from_numpy(cnt))

    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        ws, dp, im = ops.composite_rays_train_forward(npv(sigmas), npv(rgbs), npv(deltas), npv(rays), T_thresh)
        weights_sum.copy_(torch.from_numpy(ws)); depth.copy_(torch.from_numpy(dp)); image.copy_(torch.from_numpy(im))

    def composite_rays_train_backward(gws, gimg, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs):
        gs, gc = ops.composite_rays_train_backward(npv(gws), npv(gimg), npv(sigmas), npv(rgbs), npv(deltas), npv(rays), npv(weights_sum),
                                                   npv(image), T_thresh)
        grad_sigmas.copy_(torch.from_numpy(gs)); grad_rgbs.copy_(torch.from_numpy(gc))

    LOG = {"march_rays": []}

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises):
        LOG["march_rays"].append((int(n_alive), int(n_step)))
        x, d, dl = ops.march_rays(n_alive, n_step, npv(rays_alive), npv(rays_t), npv(rays_o), npv(rays_d), bound, npv(grid), C, H, npv(nears),
                                  npv(fars), align=-1, dt_gamma=dt_gamma, max_steps=max_steps, noises=npv(noises))
        m = n_alive * n_step
        xyzs[:m].copy_(torch.from_numpy(x)); dirs[:m].copy_(torch.from_numpy(d)); deltas[:m].copy_(torch.from_numpy(dl))

    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        al, rt = npv(rays_alive).copy(), npv(rays_t).copy()
        ws, dp, im = npv(weights_sum).copy(), npv(depth).copy(), npv(image).copy()
        ops.composite_rays(n_alive, n_step, al, rt, npv(sigmas), npv(rgbs), npv(deltas), ws, dp, im, T_thresh)
        rays_alive.copy_(torch.from_numpy(al)); rays_t.copy_(torch.from_numpy(rt))
        weights_sum.copy_(torch.from_numpy(ws)); depth.copy_(torch.from_numpy(dp)); image.copy_(torch.from_numpy(im))

    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        coords.copy_(torch.from_numpy(ops.sph_from_ray(npv(rays_o), npv(rays_d), radius)))

    for f in (near_far_from_aabb, sph_from_ray, morton3D, morton3D_invert, packbits, march_rays_train, composite_rays_train_forward,
              composite_rays_train_backward, march_rays, composite_rays):
        setattr(rm, f.__name__, f)
    rm.LOG = LOG
    if native != "dropin":
        sys.modules["_raymarching"] = rm

    sh = types.ModuleType("_shencoder")

    def sh_encode_forward(inputs, outputs, B, D, C, calc_grad_inputs, dy_dx):
        r = ops.sh_encode_forward(npv(inputs), C, bool(calc_grad_inputs))
        if calc_grad_inputs:
            outputs.copy_(torch.from_numpy(r[0])); dy_dx.copy_(torch.from_numpy(r[1]))
        else:
            outputs.copy_(torch.from_numpy(r))

    def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
        grad_inputs.add_(torch.from_numpy(ops.sh_encode_backward(npv(grad), npv(inputs), C, npv(dy_dx))))

    sh.sh_encode_forward, sh.sh_encode_backward = sh_encode_forward, sh_encode_backward
    if native != "dropin":
        sys.modules["_shencoder"] = sh

    # ---- mmcv / mmgen: only what the imported files touch ----
    class Registry:
        def __init__(self):
            self.map = {}

        def register_module(self, name=None, force=False, module=None):
            def _r(cls):
                self.map[name or cls.__name__] = cls
                return cls
            return _r(module) if module is not None else _r

    MODULES = Registry()

    def build_module(cfg, default_args=None):
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        return MODULES.map[args.pop("type")](**args)

    def xavier_init(module, gain=1, bias=0, distribution="normal"):
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
        if module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def constant_init(module, val, bias=0):
        nn.init.constant_(module.weight, val)
        if module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def get_module_device(module):
        return next(module.parameters()).device

    def var_to_tensor(var, index, target_shape=None, device=None):
        v = torch.from_numpy(var)[index].float().to(device)
        while target_shape is not None and v.dim() < len(target_shape):
            v = v[..., None]
        return v

    def _get_noise_batch(noise, image_shape, num_timesteps=0, num_batches=0, timesteps_noise=False):
        return torch.randn((num_batches, *image_shape))

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("mmcv", ProgressBar=object)
    mod("mmcv.cnn", xavier_init=xavier_init, constant_init=constant_init)
    mod("mmgen"); mod("mmgen.models"); mod("mmgen.models.architectures"); mod("mmgen.models.diffusions")
    mod("mmgen.models.builder", MODULES=MODULES, MODELS=MODULES, build_module=build_module)
    mod("mmgen.models.architectures.common", get_module_device=get_module_device)
    mod("mmgen.models.diffusions.utils", var_to_tensor=var_to_tensor, _get_noise_batch=_get_noise_batch)
    mod("mcubes")
    # bare packages (skip the reference's __init__ files, which import the training stack, GUI, datasets ...)
    for pkg, path in (("lib", "lib"), ("lib.models", "lib/models"), ("lib.models.decoders", "lib/models/decoders"),
                      ("lib.models.diffusions", "lib/models/diffusions"), ("lib.core", "lib/core"), ("lib.core.utils", "lib/core/utils")):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[pkg] = m
    sys.path.insert(0, REF)
    return MODULES, LOG, backend_name


def main():
    assert os.path.isdir(REF), "the reference checkout is needed to (re)generate the fixtures"
    MODULES, LOG, backend_name = _install_stubs()
    import warnings
    warnings.filterwarnings("ignore")
    lib_ops = importlib.import_module("lib.ops")                       # executes the reference's lib/ops/__init__.py and wrappers
    tp = importlib.import_module("lib.models.decoders.triplane_decoder")
    nu = importlib.import_module("lib.core.utils.nerf_utils")
    gd = importlib.import_module("lib.models.diffusions.gaussian_diffusion")
    from ssdnerf_amd import synthetic as S
    from oracle import render as R

    torch.manual_seed(0)
    params, code = S.make_decoder_params(), S.make_triplane()
    dec = tp.TriPlaneDecoder(interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], use_dir_enc=True,
                             dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256)
    dec.load_state_dict(params, strict=False)
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g).numpy() for _ in range(2)]
    _, bits, _ = R.get_density(params, code, jit, density_thresh=0.1)
    out = {}

    # ---- 1. camera rays (reference get_cam_rays) : BASELINE config 1 = one 64x64 view, pose 64 ----
    poses, intr = S.spiral_poses()[[64]][None], S.cars_intrinsics(64, 64)[None, None]
    ro, rd = nu.get_cam_rays(poses, intr, 64, 64)
    ro, rd = ro.reshape(1, -1, 3).contiguous(), rd.reshape(1, -1, 3).contiguous()
    np.savez_compressed(os.path.join(HERE, "cam_rays_64.npz"), pose=poses.numpy(), intrinsics=intr.numpy(), rays_o=ro.numpy(), rays_d=rd.numpy())

    # ---- 2. decode ----
    xyz = torch.rand(4096, 3, generator=g) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1)
    with torch.no_grad():
        sig, rgb, n = dec.point_decode([xyz], [dirs], code[None])
        sig_d, _ = dec.point_density_decode([xyz], code[None])
    np.savez_compressed(os.path.join(HERE, "point_decode.npz"), xyzs=xyz.numpy(), dirs=dirs.numpy(), sigmas=sig.numpy(), rgbs=rgb.numpy(),
                        sigmas_density_only=sig_d.numpy())

    # ---- 3. eval-branch render (reference VolumeRenderer.forward, dec.eval()) ----
    dec.eval()
    for tag, dtg in (("dtg0", 0.0), ("dtg", 0.5 * 2 / (2 * 65.625))):
        LOG["march_rays"].clear()
        with torch.no_grad():
            res = dec(ro, rd, code[None], torch.from_numpy(bits)[None], 64, dt_gamma=torch.tensor([dtg], dtype=torch.float32), perturb=False)   # the reference calls .item() on it
        ws, dp, im = res["weights_sum"][0], res["depth"][0], res["image"][0]
        np.savez_compressed(os.path.join(HERE, f"render_eval_64_{tag}.npz"), dt_gamma=np.float32(dtg), weights_sum=ws.numpy(), depth=dp.numpy(),
                            image=im.numpy(), rgb_bg1=(im + 1.0 * (1 - ws.unsqueeze(-1))).numpy(), iterations=np.array(LOG["march_rays"], np.int32))

    # ---- 4. train-branch render + backward (perturb=False so no RNG is involved) ----
    dec.train()
    code_g = code.clone()[None].requires_grad_(True)
    sub = slice(0, 4096, 4)
    res = dec(ro[:, sub], rd[:, sub], code_g, torch.from_numpy(bits)[None], 64, dt_gamma=torch.tensor([0.0038095]), perturb=False)
    tgt = torch.rand(1, ro[:, sub].shape[1], 3, generator=g)
    rgbs = res["image"] + 1.0 * (1 - res["weights_sum"].unsqueeze(-1))
    loss = ((rgbs - tgt) ** 2).mean() * 20.0
    (gcode,) = torch.autograd.grad(loss, code_g)
    np.savez_compressed(os.path.join(HERE, "render_train_64.npz"), ray_subset=np.arange(0, 4096, 4), target=tgt.numpy(), weights_sum=res["weights_sum"].detach().numpy(),
                        depth=res["depth"].detach().numpy(), image=res["image"].detach().numpy(), loss=np.float32(loss.item()),
                        grad_code_absmax=np.float32(gcode.abs().max().item()), grad_code_sample=gcode[0, :, :, ::16, ::16].numpy())

    # ---- 5. DDIM: schedule tables, timesteps, a sampled trajectory with a toy denoiser (+ guidance) ----
    @MODULES.register_module()
    class ToyDenoiser(nn.Module):
        def __init__(self, num_classes=0, num_timesteps=1000):
            super().__init__()
            self.conv = nn.Conv2d(18, 18, 3, padding=1)
            gg = torch.Generator().manual_seed(3)
            with torch.no_grad():
                self.conv.weight.copy_(torch.randn(self.conv.weight.shape, generator=gg) * 0.05)
                self.conv.bias.copy_(torch.randn(18, generator=gg) * 0.1)

        def forward(self, x_t, t, concat_cond=None):
            return torch.tanh(self.conv(x_t)) * (1 + t.float().view(-1, 1, 1, 1) / 1000)

    @MODULES.register_module()
    class _Null(nn.Module):
        def __init__(self, **kw):
            super().__init__()

    MODULES.map["SNRWeightedTimeStepSampler"] = _Null
    MODULES.map["DDPMMSELossMod"] = _Null
    diff = gd.GaussianDiffusion(denoising=dict(type="ToyDenoiser"), ddpm_loss=dict(type="DDPMMSELossMod"), betas_cfg=dict(type="linear"),
                                num_timesteps=1000, timestep_sampler=dict(type="SNRWeightedTimeStepSampler"), denoising_mean_mode="V",
                                test_cfg=dict(num_timesteps=10, clip_range=[-2, 2]))
    noise = torch.randn(2, 18, 16, 16, generator=g)
    with torch.no_grad():
        traj = diff.ddim_sample(noise.clone(), save_intermediates=True)
    target = torch.randn(2, 18, 16, 16, generator=g)
    diff.test_cfg["guidance_gain"] = 2.0
    with torch.no_grad():                                                  # the reference samples under no_grad (diffusion_nerf.py:409) and
        guided = diff.ddim_sample(noise.clone(), grad_guide_fn=lambda x0: ((x0 - target) ** 2).mean() * 5.0)   # pred_x_0 re-enables grad itself
    ts = {}
    for n in (50, 75, 10):
        ts[n] = torch.arange(start=999, end=-1, step=-(1000 / n)).long().numpy()
    np.savez_compressed(os.path.join(HERE, "ddim.npz"), betas=diff.betas, alphas_bar=diff.alphas_bar, alphas_bar_prev=diff.alphas_bar_prev,
                        sqrt_alphas_bar=diff.sqrt_alphas_bar, sqrt_one_minus_alphas_bar=diff.sqrt_one_minus_alphas_bar,
                        tilde_betas_t=diff.tilde_betas_t, timesteps_50=ts[50], timesteps_75=ts[75], timesteps_10=ts[10],
                        noise=noise.numpy(), target=target.numpy(), conv_weight=diff.denoising.conv.weight.detach().numpy(),
                        conv_bias=diff.denoising.conv.bias.detach().numpy(), x0_steps=np.stack([t.numpy() for t in traj[0::2]]),
                        xt_steps=np.stack([t.numpy() for t in traj[1::2]]), guided_final=guided.detach().numpy())
    with open(os.path.join(HERE, "PROVENANCE.txt"), "w") as f:
        f.write("Generated by tests/golden/make_golden.py by executing the reference's Python modules (see its docstring).\n"
                f"Native kernels behind _raymarching/_shencoder: {backend_name}.\n"
                f"torch {torch.__version__}, numpy {np.__version__}.\n")
    print("fixtures written:", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")), "| backend:", backend_name)


if __name__ == "__main__":
    main()
