#!/usr/bin/env python
"""tools/recons_regime.py SCALE [...] -- which synthetic prior gives the bench's reconstruction batches something to reconstruct?  Builds bench.py's model, multiplies the
UNet's output convolution (and the attention output projections) by SCALE -- the V-prediction of a random network of unit-scale outputs drives every code into the
TanhCode's saturation and the scenes empty (r05); a SMALL prediction makes x0 ~ sqrt(alpha_bar) x_t, so the guidance gradient and the fine-tuning decide the scene -- and
runs one 'guide' batch (75 guided steps) and one 'guide_optim' batch (75 + 25): foreground share of the 250 test views, largest |code|, PSNR of the conditioning view."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from ssdnerf_amd import synthetic as S
dev = torch.device("cuda")
ns = 8
GAINS = [float(x) for x in os.environ.get("GAIN_SCALES", "1").split(",")]
for scale, gain_scale in [(float(x), gs) for x in (sys.argv[1:] or ["1.0", "0.05"]) for gs in GAINS]:
    model = B.build_model(dev)
    with torch.no_grad():
        unet = model.diffusion_ema.denoising
        for name, p in unet.named_parameters():
            if name.startswith("out.") and name.endswith(("weight", "bias")) and "conv" in name or ".proj." in name:
                p.mul_(scale)
    cfg = model.test_cfg
    codes = torch.stack([S.make_triplane(100 + i) for i in range(ns)]).to(dev)
    poses1 = S.spiral_poses()[[64]].to(dev)[None].expand(ns, -1, -1, -1).contiguous()
    intr1 = S.cars_intrinsics(128, 128).to(dev)[None, None].expand(ns, 1, -1).contiguous()
    with torch.no_grad():
        other = codes.roll(1, 0)
        target, _ = model.render(model.decoder_ema, other, model.get_density(model.decoder_ema, other, cfg=cfg)[1], 128, 128, intr1, poses1, cfg=cfg)
    data = dict(cond_imgs=target.clamp(0, 1), cond_intrinsics=intr1, cond_poses=poses1)
    nv = 250
    poses = S.spiral_poses(251)[:nv].to(dev)[None].expand(ns, -1, -1, -1).contiguous()
    intr = S.cars_intrinsics(128, 128).to(dev)[None, None].expand(ns, nv, -1).contiguous()
    g = torch.Generator().manual_seed(0)
    noise = torch.randn(ns, 3, 6, 128, 128, generator=g).to(dev)
    cond = data["cond_imgs"][:, 0].permute(0, 3, 1, 2)
    cfg["guidance_gain"] = 3.2 * (2 ** 14) * gain_scale
    print(f"== scale {scale}, guidance gain x {gain_scale}: target foreground {float((cond < 0.995).any(dim=1).float().mean()):.3f}")
    for mode in ("guide", "guide_optim"):
        cfg.update(num_timesteps=75, n_inverse_steps=25, extra_scene_step=3, cond_mode=mode)
        model.diffusion_ema.test_cfg.update(num_timesteps=75)
        torch.manual_seed(1234); np.random.seed(1234)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = model.val_step(dict(data, noise=noise, test_poses=poses, test_intrinsics=intr))
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        pred = res["pred_imgs"]
        mse = (pred[:, 64].float() - cond.float()).square().flatten(1).mean(-1)
        psnr = 10 * (-torch.log10(mse + 1e-6))
        print(f"  {mode:12s} {wall:6.2f} s  foreground {float((pred < 0.995).any(dim=2).float().mean()):.4f}  |code| max {float(res['code'].abs().max()):.4f} mean {float(res['code'].abs().mean()):.4f}"
              f"  PSNR(conditioning view) mean {float(psnr.mean()):.2f} min {float(psnr.min()):.2f}  finite {bool(torch.isfinite(res['code']).all())}")
    del model
    torch.cuda.empty_cache()
