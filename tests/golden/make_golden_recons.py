#!/usr/bin/env python
"""tests/golden/make_golden_recons.py -- golden fixtures for SURVEY.md section 8(f) (fine-tuning, Langevin correction, scene cache),
generated like make_golden.py by EXECUTING THE REFERENCE'S OWN PYTHON in the build container (the fixture travels, /root/reference
does not).  Writes recons.npz and scene_cache.pt next to this file.

What runs for real (from /root/reference, unmodified):
    lib/models/diffusions/gaussian_diffusion.py   q_sample, loss, forward_train, p_sample_langevin, ddim_sample (with Langevin steps)
    lib/models/diffusions/sampler.py              SNRWeightedTimeStepSampler (weights, probabilities)
    lib/models/losses/ddpm_loss.py                DDPMLossMod.timestep_weight_rescale / forward, DDPMMSELossMod (0.5 * mse, norm_factor)
    lib/core/utils/misc.py                        optimizer_state_to, load_tensor_to_dict, optimizer_state_copy, optimizer_set_state
    lib/models/autodecoders/multiscene_nerf.py    out_dict_to
    lib/models/autodecoders/base_nerf.py          TanhCode, NormalizedTanhCode
  The last three files import the training stack (mmcv runners, lpips, datasets): only the named functions / classes are taken from
  them, by exec'ing their source segments (ast) in a namespace that holds what those segments use.

What is substituted (not installable here): mmcv / mmgen helpers as in make_golden.py, plus mmgen's ``UniformTimeStepSampler``
(np.random.choice over ``prob``) and ``DDPMLoss`` / ``mse_loss`` / ``reduce_loss`` restated from SURVEY.md Appendix A - so the prior-loss
value pins the reference's Mod layer on top of that restatement, not mmgen's own file."""
import ast
import importlib
import os
import sys
import types
from collections import abc as container_abcs, defaultdict
from functools import partial
from itertools import chain

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

REF = MG.REF


def _segments(path, names):
    """source of the named top-level functions / classes of a reference file"""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    out = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            out.append(ast.get_source_segment(src, node))
    assert len(out) == len(names), (path, names)
    return "\n\n".join(out)


def main():
    assert os.path.isdir(REF)
    MODULES, LOG, backend_name = MG._install_stubs()
    import warnings
    warnings.filterwarnings("ignore")

    # ---- mmgen pieces the two extra reference files import ----
    class UniformTimeStepSampler:
        def __init__(self, num_timesteps):
            self.num_timesteps = num_timesteps
            self.prob = [1 / self.num_timesteps for _ in range(self.num_timesteps)]

        def sample(self, batch_size):
            return torch.from_numpy(np.random.choice(self.num_timesteps, size=(batch_size,), p=self.prob)).long()

        def __call__(self, batch_size):
            return self.sample(batch_size)

    def reduce_loss(loss, reduction):
        if reduction == "flatmean":
            return loss.flatten(1).mean(dim=1)
        return {"none": lambda v: v, "mean": lambda v: v.mean(), "sum": lambda v: v.sum()}[reduction](loss)

    def mse_loss(pred, target, reduction="mean"):
        return reduce_loss(F.mse_loss(pred, target, reduction="none"), reduction)

    class DDPMLoss(nn.Module):
        def __init__(self, rescale_mode=None, rescale_cfg=None, sampler=None, weight=None, log_cfgs=None, reduction="mean", loss_name=None):
            super().__init__()
            self.reduction, self.sampler, self._loss_name = reduction, sampler, loss_name
            self.log_vars = dict()
            if not rescale_mode:
                self.rescale_fn = lambda loss, t: loss
            else:
                assert rescale_mode == "timestep_weight"
                if sampler is not None and hasattr(sampler, "weight"):
                    weight = sampler.weight
                self.rescale_fn = partial(self.timestep_weight_rescale, weight=weight)

        def collect_log(self, loss, timesteps):
            pass

    sys.modules["mmgen.models"].MODULES = MODULES
    sys.modules["mmgen.models.diffusions"].UniformTimeStepSampler = UniformTimeStepSampler
    MG_mod = lambda name, **a: sys.modules.setdefault(name, types.ModuleType(name)).__dict__.update(a)
    MG_mod("mmgen.models.losses"); MG_mod("mmgen.models.losses.ddpm_loss", DDPMLoss=DDPMLoss, mse_loss=mse_loss, reduce_loss=reduce_loss)
    for pkg, path in (("lib.models.losses", "lib/models/losses"),):
        m = types.ModuleType(pkg); m.__path__ = [os.path.join(REF, path)]; sys.modules[pkg] = m
    sys.modules["lib.core"].reduce_mean = lambda t: t

    gd = importlib.import_module("lib.models.diffusions.gaussian_diffusion")
    importlib.import_module("lib.models.diffusions.sampler")
    importlib.import_module("lib.models.losses.ddpm_loss")

    @MODULES.register_module()
    class ToyDenoiser2(nn.Module):
        def __init__(self, num_classes=0, num_timesteps=1000):
            super().__init__()
            self.conv = nn.Conv2d(18, 18, 3, padding=1)
            gg = torch.Generator().manual_seed(3)
            with torch.no_grad():
                self.conv.weight.copy_(torch.randn(self.conv.weight.shape, generator=gg) * 0.05)
                self.conv.bias.copy_(torch.randn(18, generator=gg) * 0.1)

        def forward(self, x_t, t, concat_cond=None):
            return torch.tanh(self.conv(x_t)) * (1 + t.float().view(-1, 1, 1, 1) / 1000)

    diff = gd.GaussianDiffusion(
        denoising=dict(type="ToyDenoiser2"), betas_cfg=dict(type="linear"), num_timesteps=1000, denoising_mean_mode="V",
        timestep_sampler=dict(type="SNRWeightedTimeStepSampler", power=0.5),
        ddpm_loss=dict(type="DDPMMSELossMod", rescale_mode="timestep_weight", data_info=dict(pred="v_t_pred", target="v_t"), weight_scale=4.0,
                       scale_norm=True),
        test_cfg=dict(num_timesteps=4, clip_range=[-2, 2], langevin_steps=2, langevin_delta=0.4))
    diff.eval()
    diff.ddpm_loss.norm_factor.fill_(1.7)
    g = torch.Generator().manual_seed(21)
    out = dict(conv_weight=diff.denoising.conv.weight.detach().numpy(), conv_bias=diff.denoising.conv.bias.detach().numpy(),
               snr_weight=diff.sampler.weight.numpy(), snr_prob=np.asarray(diff.sampler.prob))

    # ---- prior loss (forward_train) with seeded host draws, and its gradient w.r.t. x_0 ----
    x0 = (torch.randn(3, 18, 16, 16, generator=g) * 0.7).requires_grad_(True)
    np.random.seed(5); torch.manual_seed(5)
    loss, _ = diff(x0, return_loss=True, cfg=diff.test_cfg)
    (gx,) = torch.autograd.grad(loss, x0)
    np.random.seed(5); torch.manual_seed(5)
    t_drawn = diff.sampler(3)
    noise_drawn = torch.randn(3, 18, 16, 16)
    x_t, mean, std = diff.q_sample(x0.detach(), t_drawn, noise_drawn)
    out.update(prior_x0=x0.detach().numpy(), prior_t=t_drawn.numpy(), prior_noise=noise_drawn.numpy(), prior_x_t=x_t.numpy(),
               prior_loss=np.float32(loss.item()), prior_grad=gx.numpy())

    # ---- DDIM with Langevin correction steps (unguided and guided), noise drawn from the seeded global CPU generator ----
    noise = torch.randn(2, 18, 16, 16, generator=g)
    target = torch.randn(2, 18, 16, 16, generator=g)
    torch.manual_seed(77)
    with torch.no_grad():
        lv = diff.ddim_sample(noise.clone())
    diff.test_cfg["guidance_gain"] = 2.0
    torch.manual_seed(78)
    with torch.no_grad():
        lv_g = diff.ddim_sample(noise.clone(), grad_guide_fn=lambda x0_: ((x0_ - target) ** 2).mean() * 5.0)
    out.update(noise=noise.numpy(), target=target.numpy(), langevin_final=lv.numpy(), langevin_guided_final=lv_g.detach().numpy())

    # ---- DDPM ancestral sampler (sample_method='ddpm'), 5 steps, both variance modes ----
    diff.test_cfg.update(num_timesteps=5, langevin_steps=0)
    diff.test_cfg.pop("guidance_gain")
    for mode in ("FIXED_LARGE", "FIXED_SMALL"):
        diff.denoising_var_mode = mode
        torch.manual_seed(91)
        with torch.no_grad():
            out[f"ddpm_{mode.lower()}"] = diff.ddpm_sample(noise.clone()).numpy()

    # ---- code activations ----
    ns = dict(torch=torch, nn=nn, reduce_mean=lambda t: t, MODULES=MODULES)
    exec(_segments("lib/models/autodecoders/base_nerf.py", ["TanhCode", "NormalizedTanhCode"]), ns)
    c_ = torch.randn(2, 3, 6, 8, 8, generator=g) * 1.5
    nt = ns["NormalizedTanhCode"](mean=0.0, std=0.5, clip_range=2)
    y_eval = nt.eval()(c_)
    inv_eval = nt.inverse(y_eval)
    nt.train()
    y_train = nt(c_, update_stats=True)
    out.update(act_in=c_.numpy(), ntanh_eval=y_eval.numpy(), ntanh_inverse=inv_eval.numpy(), ntanh_train=y_train.numpy(),
               ntanh_running_mean=nt.running_mean.numpy().copy(), ntanh_running_var=nt.running_var.numpy().copy(),
               tanh2=ns["TanhCode"](scale=2)(c_).numpy(), tanh2_inverse=ns["TanhCode"](scale=2).inverse(ns["TanhCode"](scale=2)(c_)).numpy())
    np.savez_compressed(os.path.join(HERE, "recons.npz"), **out)

    # ---- scene-cache casting rules: a live Adam state through the reference's helpers ----
    ns = dict(torch=torch, chain=chain, defaultdict=defaultdict, container_abcs=container_abcs)
    exec(_segments("lib/core/utils/misc.py", ["optimizer_state_to", "load_tensor_to_dict", "optimizer_state_copy", "optimizer_set_state"]), ns)
    exec(_segments("lib/models/autodecoders/multiscene_nerf.py", ["out_dict_to"]), ns)
    code_ = (torch.randn(3, 6, 8, 8, generator=g) * 3).requires_grad_(True)
    code_.data[0, 0, 0, 0] = 1e6                                           # beyond fp16's range: must clamp to 65504, not overflow to inf
    opt = torch.optim.Adam([code_], lr=0.01)
    for _ in range(3):
        opt.zero_grad()
        ((code_ * 1e-3) ** 2).sum().backward()
        opt.step()
    import copy
    live = copy.deepcopy(dict(scene_id=7, scene_name="scene_7",       # deep copy: state_dict() aliases the live moments, which Adam updates in place
                              param=dict(code_=code_.data.clone(), density_grid=torch.rand(64, generator=g).half(),
                                         density_bitfield=torch.randint(0, 255, (8,), generator=g, dtype=torch.uint8)),
                              optimizer=opt.state_dict()))
    cached16 = ns["out_dict_to"](live, device="cpu", code_dtype=torch.float16, optimizer_dtype=torch.bfloat16)
    cached32 = ns["out_dict_to"](live, device="cpu", code_dtype=torch.float32, optimizer_dtype=torch.float32)
    # second save into an existing entry (in-place refresh), as save_cache does
    for _ in range(2):
        opt.zero_grad()
        ((code_ * 1e-3) ** 2).sum().backward()
        opt.step()
    live2 = copy.deepcopy(dict(live, param=dict(live["param"], code_=code_.data.clone()), optimizer=opt.state_dict()))
    refreshed16 = copy.deepcopy(cached16)
    for key, val in live2["param"].items():
        ns["load_tensor_to_dict"](refreshed16["param"], key, val, device="cpu", dtype=torch.float16)
    ns["optimizer_state_copy"](live2["optimizer"], refreshed16["optimizer"], device="cpu", dtype=torch.bfloat16)
    # restoring the bf16 state into a fresh fp32 optimizer
    code_b = code_.detach().clone().requires_grad_(True)
    opt_b = torch.optim.Adam([code_b], lr=0.5)
    ns["optimizer_set_state"](opt_b, refreshed16["optimizer"])
    st = opt_b.state[code_b]
    restored = dict(exp_avg=st["exp_avg"].clone(), exp_avg_sq=st["exp_avg_sq"].clone(), step=st["step"].clone() if torch.is_tensor(st["step"]) else st["step"],
                    lr=opt_b.param_groups[0]["lr"])
    torch.save(dict(live=live, cached16=cached16, cached32=cached32, live2=live2, refreshed16=refreshed16, restored=restored),
               os.path.join(HERE, "scene_cache.pt"))
    with open(os.path.join(HERE, "PROVENANCE.txt"), "a") as f:
        f.write("recons.npz, scene_cache.pt: tests/golden/make_golden_recons.py (reference gaussian_diffusion / sampler / ddpm_loss / misc / "
                f"multiscene_nerf / base_nerf segments executed); torch {torch.__version__}, numpy {np.__version__}.\n")
    print("written: recons.npz, scene_cache.pt")


if __name__ == "__main__":
    main()
