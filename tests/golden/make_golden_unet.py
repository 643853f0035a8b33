#!/usr/bin/env python
"""tests/golden/make_golden_unet.py -- pins SURVEY.md row a14 (the DDIM denoiser) by EXECUTING THE REFERENCE'S OWN UNet code in the build
container; writes unet_cars.npz and unet_tiled.npz next to this file (the fixtures travel, /root/reference does not).

What runs for real (imported from /root/reference, unmodified):
    lib/models/architecture/ddpm/denoising.py   DenoisingUnetMod.__init__ (all of the block wiring: levels, attention scales, skip bookkeeping,
                                                the norm-act-conv output head) and DenoisingUnetMod.forward (time embedding, skip concatenation order)
    lib/models/architecture/ddpm/modules.py     MultiHeadAttentionMod.__init__/forward (the grouped qkv reshape), DenoisingResBlockMod.__init__,
                                                DenoisingDownsampleMod.__init__, DenoisingUpsampleMod.__init__

What is substituted: the mmgen 0.7.2 / mmcv 1.6 parents those classes inherit their remaining methods from (not vendored, not installable:
no network) -- ``DenoisingUnet.init_weights``, ``DenoisingResBlock.forward/init_weights``, ``NormWithEmbedding``, ``TimeEmbedding``,
``EmbedSequential``, ``MultiHeadAttention.QKVAttention/init_weights``, ``DenoisingDownsample/Upsample.forward``, mmcv's ``ConvModule``,
``build_norm_layer``, ``build_activation_layer`` -- restated below from SURVEY.md Appendix A.  So the fixtures pin the reference's own layer on top
of that restatement; the restatement itself stays the one unverifiable link (said so in DESIGN.md).

The network weights are not stored: generator and tests fill the state-dict with unet_fill.fill_state_dict (seeded, sorted key order); the
fixture holds the key list, the shapes and a checksum of the fill, the input, the timesteps and the reference's output."""
import importlib
import math
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from unet_fill import UNET_CONFIGS, fill_state_dict, make_input  # noqa: E402

REF = MG.REF


def install_mmgen_unet_parents(MODULES, build_module):
    """mmgen.models.architectures.ddpm.{modules,denoising} + the mmcv bricks they use, restated."""

    def build_norm_layer(cfg, num_features, postfix=""):
        cfg = dict(cfg)
        typ = cfg.pop("type")
        assert typ == "GN"
        cfg.setdefault("eps", 1e-5)
        return "gn" + str(postfix), nn.GroupNorm(num_channels=num_features, **cfg)

    def build_activation_layer(cfg):
        cfg = dict(cfg)
        typ = cfg.pop("type")
        return {"SiLU": nn.SiLU, "ReLU": nn.ReLU}[typ](**cfg)

    def constant_init(module, val, bias=0):
        nn.init.constant_(module.weight, val)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    class ConvModule(nn.Module):
        """mmcv ConvModule restricted to what denoising.py:178-187 asks for: order=('norm','act','conv'), GN + SiLU, norm over the INPUT channels."""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto", conv_cfg=None,
                     norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, with_spectral_norm=False, padding_mode="zeros",
                     order=("conv", "norm", "act")):
            super().__init__()
            assert order == ("norm", "act", "conv") and norm_cfg is not None and act_cfg is not None
            self.order = order
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation, groups=groups,
                                  bias=bool(bias))
            self.norm_name, norm = build_norm_layer(norm_cfg, in_channels)        # norm before conv -> normalises the conv's input
            self.add_module(self.norm_name, norm)
            self.activate = build_activation_layer(act_cfg)

        def forward(self, x):
            for layer in self.order:
                x = {"conv": self.conv, "norm": getattr(self, self.norm_name), "act": self.activate}[layer](x)
            return x

    class EmbedSequential(nn.Sequential):
        def forward(self, x, y):
            for layer in self:
                x = layer(x, y) if isinstance(layer, DenoisingResBlock) else layer(x)
            return x

    class TimeEmbedding(nn.Module):
        def __init__(self, in_channels, embedding_channels, embedding_mode="sin", embedding_cfg=None, act_cfg=dict(type="SiLU", inplace=False)):
            super().__init__()
            self.blocks = nn.Sequential(nn.Linear(in_channels, embedding_channels), build_activation_layer(act_cfg),
                                        nn.Linear(embedding_channels, embedding_channels))
            cfg = dict(dim=in_channels)
            if embedding_cfg is not None:
                cfg.update(embedding_cfg)
            assert embedding_mode.upper() == "SIN"
            self.embedding_fn = partial(self.sinusodial_embedding, **cfg)

        @staticmethod
        def sinusodial_embedding(timesteps, dim, max_period=10000):
            half = dim // 2
            freqs = torch.exp(-np.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device=timesteps.device)
            args = timesteps[:, None].float() * freqs[None]
            emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
            if dim % 2:
                emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
            return emb

        def forward(self, t):
            return self.blocks(self.embedding_fn(t))

    @MODULES.register_module()
    class NormWithEmbedding(nn.Module):
        def __init__(self, in_channels, embedding_channels, norm_cfg=dict(type="GN", num_groups=32), act_cfg=dict(type="SiLU", inplace=False),
                     use_scale_shift=True):
            super().__init__()
            self.use_scale_shift = use_scale_shift
            _, self.norm = build_norm_layer(norm_cfg, in_channels)
            embedding_output = in_channels * 2 if use_scale_shift else in_channels
            self.embedding_layer = nn.Sequential(build_activation_layer(act_cfg), nn.Linear(embedding_channels, embedding_output))

        def forward(self, x, y):
            embedding = self.embedding_layer(y)[:, :, None, None]
            if self.use_scale_shift:
                scale, shift = torch.chunk(embedding, 2, dim=1)
                return self.norm(x) * (1 + scale) + shift
            return self.norm(x + embedding)

    class DenoisingResBlock(nn.Module):
        def forward_shortcut(self, x):
            return self.shortcut(x) if self.learnable_shortcut else x

        def forward(self, x, y):
            shortcut = self.forward_shortcut(x)
            x = self.conv_1(x)
            x = self.norm_with_embedding(x, y)
            x = self.conv_2(x)
            return x + shortcut

        def init_weights(self):
            constant_init(self.conv_2[-1], 0)

    class MultiHeadAttention(nn.Module):
        @staticmethod
        def QKVAttention(qkv):
            channel = qkv.shape[1] // 3
            q, k, v = torch.chunk(qkv, 3, dim=1)
            scale = 1 / np.sqrt(np.sqrt(channel))
            weight = torch.einsum("bct,bcs->bts", q * scale, k * scale)
            weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
            return torch.einsum("bts,bcs->bct", weight, v)

        def init_weights(self):
            constant_init(self.proj, 0)

    class DenoisingDownsample(nn.Module):
        def forward(self, x):
            return self.downsample(x)

    class DenoisingUpsample(nn.Module):
        def forward(self, x):
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            return self.conv(x) if getattr(self, "with_conv", False) else x

    class DenoisingUnet(nn.Module):
        def init_weights(self, pretrained=None):
            assert pretrained is None
            for n, m in self.named_modules():
                if isinstance(m, nn.Conv2d) and ("conv_2" in n or ("out" in n and "out_blocks" not in n)):
                    constant_init(m, 0)
                if isinstance(m, nn.Conv1d) and "proj" in n:
                    constant_init(m, 0)

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("mmcv.cnn.bricks", build_activation_layer=build_activation_layer, build_norm_layer=build_norm_layer)
    mod("mmcv.cnn.bricks.conv_module", ConvModule=ConvModule)
    mod("mmgen.models.architectures.ddpm")
    mod("mmgen.models.architectures.ddpm.modules", TimeEmbedding=TimeEmbedding, EmbedSequential=EmbedSequential, MultiHeadAttention=MultiHeadAttention,
        DenoisingResBlock=DenoisingResBlock, DenoisingDownsample=DenoisingDownsample, DenoisingUpsample=DenoisingUpsample,
        NormWithEmbedding=NormWithEmbedding)
    mod("mmgen.models.architectures.ddpm.denoising", DenoisingUnet=DenoisingUnet)
    for pkg, path in (("lib.models.architecture", "lib/models/architecture"), ("lib.models.architecture.ddpm", "lib/models/architecture/ddpm")):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[pkg] = m


def main():
    assert os.path.isdir(REF), "the reference checkout is needed to (re)generate the fixtures"
    MODULES, _, _ = MG._install_stubs()
    build_module = sys.modules["mmgen.models.builder"].build_module
    install_mmgen_unet_parents(MODULES, build_module)
    importlib.import_module("lib.models.architecture.ddpm.modules")         # the reference's files, executed
    den = importlib.import_module("lib.models.architecture.ddpm.denoising")
    torch.manual_seed(0)
    for name, spec in UNET_CONFIGS.items():
        net = den.DenoisingUnetMod(**spec["kwargs"]).eval()
        sd = net.state_dict()
        checksum = fill_state_dict(sd, spec["seed"])
        net.load_state_dict(sd)
        x = make_input(spec["x_shape"], spec["seed"])
        t = torch.tensor(spec["timesteps"], dtype=torch.long)
        with torch.no_grad():
            out = net(x, t)
            # intermediate probes (help to localise a mismatch): the time embedding and the activation after the encoder / the middle block
            emb = net.time_embedding(t.float() * (1000.0 / net.num_timesteps))
            h, hs = x, []
            for blk in net.in_blocks:
                h = blk(h, emb)
                hs.append(h)
            enc = h.clone()
            mid = net.mid_blocks(h, emb)
        keys = sorted(sd.keys())
        np.savez_compressed(os.path.join(HERE, f"unet_{name}.npz"), x=x.numpy(), t=t.numpy(), out=out.numpy(), time_embedding=emb.numpy(),
                            encoder_out=enc.numpy(), mid_out=mid.numpy(), keys=np.array(keys), shapes=np.array([str(tuple(sd[k].shape)) for k in keys]),
                            checksum=np.float64(checksum), n_params=np.int64(sum(p.numel() for p in net.parameters())),
                            skip_channels=np.array(net.in_channels_list, np.int64))
        print(f"unet_{name}.npz: {sum(p.numel() for p in net.parameters()) / 1e6:.2f} M parameters, |out| max {out.abs().max():.3f}, "
              f"{len(keys)} tensors, checksum {checksum:.6f}")


if __name__ == "__main__":
    main()
