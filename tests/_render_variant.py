"""Helper of tests/test_render_gpu.py::test_specialised_shading_kernels_are_bit_identical: renders a fixed workload with the kernel form chosen by
SSDNERF_SHADE_GENERIC (read once per process by the library, hence one process per form) and writes the outputs to the given .npz."""
import sys

import numpy as np
import torch

from ssdnerf_amd import nerf, synthetic as S
from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
from ssdnerf_amd.density import get_density

dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
dec.load_state_dict(S.make_decoder_params(), strict=False)
dec = dec.cuda().eval()
code = torch.stack([S.make_triplane(11), S.make_triplane(12, "uniform")]).cuda()
g = torch.Generator().manual_seed(7)
grid, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=4, jitters=[torch.rand(64 ** 3, 3, generator=g).cuda() for _ in range(4)])
poses = S.spiral_poses()[[3, 64, 180]].cuda()[None].expand(2, -1, -1, -1).contiguous()
intr = S.cars_intrinsics(128, 128).cuda()[None, None].expand(2, 3, -1).contiguous()
ro, rd = nerf.get_cam_rays(poses, intr, 128, 128)
planes = pack_triplanes(code, dec.plane_dtype)
out = {}
for tag, gammas in (("mixed", [0.0, 0.0038095]), ("zero", [0.0, 0.0])):      # per-scene cone angles (MODE 1) and the uncond render's dt_gamma == 0 (MODE 2)
    res = dec.render_packed(planes, ro.reshape(2, -1, 3), rd.reshape(2, -1, 3), bits, 64, gammas, 1e-4, bg_color=1.0, want_counts=True, check_overflow=False)
    torch.cuda.synchronize()
    out.update({f"{tag}_image": res["image"].cpu().numpy(), f"{tag}_depth": res["depth"].cpu().numpy(), f"{tag}_weights_sum": res["weights_sum"].cpu().numpy(),
                f"{tag}_counts": dec.last_render_stats["sample_counts"].cpu().numpy()})
np.savez(sys.argv[1], **out)
