#!/usr/bin/env python
"""tools/option_a_on_gpu.py -- INTEGRATION.md Option A executed on the MI355X: the REFERENCE's own operator wrappers
(lib/ops/raymarching/raymarching.py, lib/ops/shencoder/sphere_harmonics.py), its VolumeRenderer / TriPlaneDecoder and its host loop
(lib/models/decoders/base_volume_renderer.py:41-133) run unmodified, with `import _raymarching` / `import _shencoder` resolving to
ssdnerf_amd/dropin (ctypes over libssdnerf_hip.so).  Renders the fixtures' 64x64 view (eval branch, both dt_gamma values, and the train
branch with its backward) and compares with tests/golden/render_*_64*.npz, which the same reference code produced on the CPU with its own
kernels.  Needs a copy of the reference's Python: /root/reference in the build container, or REF=<dir> (tools/stage_reference.sh stages
lib/ into the git-ignored oracle/_ref/reference_py so that it travels to the GPU box; nothing of it is committed).  Skips without one."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)


def main():
    ref = os.environ.get("REF") or ("/root/reference" if os.path.isdir("/root/reference") else os.path.join(ROOT, "oracle", "_ref", "reference_py"))
    if not os.path.isdir(os.path.join(ref, "lib", "ops")):
        print(f"option A: no reference checkout at {ref}: skipped")
        return 0
    os.environ["REF"] = ref
    assert torch.cuda.is_available(), "option A runs on the MI355X"
    import make_golden as MG
    MG.REF = ref
    MODULES, _, backend = MG._install_stubs(native="dropin")
    import warnings
    warnings.filterwarnings("ignore")
    importlib.import_module("lib.ops")                                   # the reference's lib/ops/__init__.py: imports _raymarching / _shencoder by name
    import _raymarching, _shencoder
    assert "dropin" in _raymarching.__file__ and "dropin" in _shencoder.__file__, (_raymarching.__file__, _shencoder.__file__)
    tp = importlib.import_module("lib.models.decoders.triplane_decoder")
    from ssdnerf_amd import synthetic as S
    print(f"reference Python from {ref}; native backend: {backend} ({_raymarching.__file__})")
    params, code = S.make_decoder_params(), S.make_triplane()
    dec = tp.TriPlaneDecoder(interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], use_dir_enc=True,
                             dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256)
    dec.load_state_dict(params, strict=False)
    dec = dec.cuda()
    rays = np.load(os.path.join(GOLD, "cam_rays_64.npz"))
    ro, rd = torch.from_numpy(rays["rays_o"]).cuda(), torch.from_numpy(rays["rays_d"]).cuda()
    # the fixtures' occupancy: 2 jittered refreshes, seed 7 (tests/golden/make_golden.py) -- through the PRODUCT's density path
    from ssdnerf_amd.decoders import TriPlaneDecoder as OwnDecoder
    from ssdnerf_amd.density import get_density
    own = OwnDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
    own.load_state_dict(params, strict=False)
    own = own.cuda().eval()
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g).cuda() for _ in range(2)]
    _, bits = get_density(own, code.cuda()[None], 64, density_thresh=0.1, density_step=2, jitters=jit)
    worst = 0.0
    dec.eval()
    for tag in ("dtg0", "dtg"):
        f = np.load(os.path.join(GOLD, f"render_eval_64_{tag}.npz"))
        with torch.no_grad():
            res = dec(ro, rd, code.cuda()[None], bits, 64, dt_gamma=torch.tensor([float(f["dt_gamma"])], dtype=torch.float32), perturb=False)
        e_img = float(np.abs(res["image"][0].cpu().numpy() - f["image"]).max())
        e_ws = float(np.abs(res["weights_sum"][0].cpu().numpy() - f["weights_sum"]).max())
        e_dp = float(np.abs(res["depth"][0].cpu().numpy() - f["depth"]).max())
        print(f"eval branch, {tag}: max |image - fixture| {e_img:.2e}   |weights_sum| {e_ws:.2e}   |depth| {e_dp:.2e}")
        worst = max(worst, e_img, e_ws, e_dp / 4)
    f = np.load(os.path.join(GOLD, "render_train_64.npz"))
    dec.train()
    code_g = code.cuda()[None].clone().requires_grad_(True)
    sub = torch.from_numpy(f["ray_subset"]).cuda()
    res = dec(ro[:, sub], rd[:, sub], code_g, bits, 64, dt_gamma=torch.tensor([0.0038095]), perturb=False)
    rgbs = res["image"] + 1.0 * (1 - res["weights_sum"].unsqueeze(-1))
    loss = ((rgbs - torch.from_numpy(f["target"]).cuda()) ** 2).mean() * 20.0
    (gcode,) = torch.autograd.grad(loss, code_g)
    e_img = float(np.abs(res["image"].detach().cpu().numpy() - f["image"]).max())
    e_loss = abs(float(loss) - float(f["loss"])) / float(f["loss"])
    e_grad = float(np.abs(gcode[0, :, :, ::16, ::16].cpu().numpy() - f["grad_code_sample"]).max()) / float(f["grad_code_absmax"])
    print(f"train branch: max |image - fixture| {e_img:.2e}   loss rel. {e_loss:.2e}   d loss / d code (sampled) rel. to max {e_grad:.2e}")
    ok = worst < 5e-5 and e_img < 5e-5 and e_loss < 1e-4 and e_grad < 2e-3
    print("option A on the MI355X:", "PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
