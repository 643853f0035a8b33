"""oracle/decoder.py -- TEST INFRASTRUCTURE.  Triplane gather + tiny MLP on PyTorch-CPU, fp32.

Restates ``TriPlaneDecoder.point_decode`` for the configuration the north-star configs use
(reference: lib/models/decoders/triplane_decoder.py:104-117 ``xyz_transform``, :119-179
``point_decode``, lib/ops/activation.py:8-20 ``TruncExp``).  The gather arithmetic of the reference
lives in ATen's ``grid_sample(mode='bilinear', padding_mode='border', align_corners=False)``; the
oracle calls that very function on the CPU, so the bilinear weights are ATen's own.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as _ops


def sh_encode(dirs: torch.Tensor, degree: int = 4) -> torch.Tensor:
    """Degree-4 real SH of the (already normalised) ray direction -- via the C oracle."""
    out = _ops().sh_encode_forward(dirs.detach().cpu().numpy(), degree)
    return torch.from_numpy(out)


def gather_point_code(code: torch.Tensor, xyzs: torch.Tensor) -> torch.Tensor:
    """code (3, C, h, w), xyzs (P, 3) -> (P, 3*C) with feature index = c*3 + plane
    (triplane_decoder.py:136-160: planes (xy, xz, yz); grid x = first coordinate of the pair)."""
    xy, xz, yz = xyzs[..., :2], xyzs[..., ::2], xyzs[..., 1:]
    grid = torch.stack([xy, xz, yz], dim=0).unsqueeze(1)                    # (3, 1, P, 2)
    feat = F.grid_sample(code, grid, mode="bilinear", padding_mode="border", align_corners=False).squeeze(-2)  # (3, C, P)
    return feat.permute(2, 1, 0).reshape(xyzs.shape[0], -1)


def point_decode(params: Dict[str, torch.Tensor], code: torch.Tensor, xyzs: torch.Tensor, dirs: Optional[torch.Tensor],
                 density_only: bool = False, sigmoid_saturation: float = 0.001) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """sigma (P,), rgb (P, 3) for ONE scene.  ``params`` uses the reference's state-dict keys."""
    f = gather_point_code(code.float(), xyzs.float())
    base_x = F.linear(f, params["base_net.0.weight"], params["base_net.0.bias"])        # one layer: no activation inside
    base_act = F.silu(base_x)
    sigma = torch.exp(F.linear(base_act, params["density_net.0.weight"], params["density_net.0.bias"]).squeeze(-1))
    if density_only:
        return sigma, None
    sh = sh_encode(dirs)
    color_in = F.silu(base_x + F.linear(sh, params["dir_net.0.weight"], params["dir_net.0.bias"]))
    rgb = torch.sigmoid(F.linear(color_in, params["color_net.0.weight"], params["color_net.0.bias"]))
    if sigmoid_saturation > 0:
        rgb = rgb * (1 + sigmoid_saturation * 2) - sigmoid_saturation
    return sigma, rgb
