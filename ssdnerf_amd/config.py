"""Minimal loader for the reference's mmcv-style python configs (``mmcv.Config.fromfile``; used at
tools/train.py:129 / tools/test.py of the reference): executes the file, resolves ``_base_`` inheritance with mmcv's
merge rules (child keys override recursively; ``_delete_=True`` replaces a dict), and exposes attribute access.
``build_model`` is what ``mmgen.models.build_model(cfg.model, train_cfg=..., test_cfg=...)`` does."""
from __future__ import annotations

import os
import runpy
from typing import Any, Dict

from .registry import MODELS


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            v = self[name]
        except KeyError as e:
            raise AttributeError(name) from e
        return v

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _merge(base: Dict[str, Any], child: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = v
    return out


def _load(path: str) -> Dict[str, Any]:
    path = os.path.abspath(path)
    ns = runpy.run_path(path)
    cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v) and type(v).__name__ != "module"}
    base = cfg.pop("_base_", None)
    if base is not None:
        bases = base if isinstance(base, (list, tuple)) else [base]
        merged: Dict[str, Any] = {}
        for b in bases:
            merged = _merge(merged, _load(os.path.join(os.path.dirname(path), b)))
        cfg = _merge(merged, cfg)
    return cfg


class Config(ConfigDict):
    @staticmethod
    def fromfile(path: str) -> "Config":
        c = Config(_wrap(_load(path)))
        c["filename"] = os.path.abspath(path)
        return c

    def merge_from_dict(self, options: Dict[str, Any]) -> None:
        """``--cfg-options a.b.c=v`` semantics."""
        for key, value in options.items():
            d = self
            parts = key.split(".")
            for p in parts[:-1]:
                d = d.setdefault(p, ConfigDict())
            d[parts[-1]] = _wrap(value)


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_plain(x) for x in v]
    return v


def build_model(cfg: Config):
    import ssdnerf_amd.decoders, ssdnerf_amd.diffusion, ssdnerf_amd.models, ssdnerf_amd.unet  # noqa: F401  (populate the registries)
    return MODELS.build(_plain(cfg["model"]), default_args=dict(train_cfg=_plain(cfg.get("train_cfg", {})), test_cfg=_plain(cfg.get("test_cfg", {}))))
