"""A small stand-in for the mmcv ``Registry`` / mmgen ``build_module`` surface the reference's configs rely on
(``@MODULES.register_module()``, ``build_module(cfg, default_args)``; e.g. lib/models/decoders/triplane_decoder.py:15,
lib/models/autodecoders/base_nerf.py:104-112).  mmcv / mmgen are not installable here (no network); only the
behaviour the hot-path types need is provided: register by class name, build from ``dict(type=..., **kwargs)``."""
from __future__ import annotations

import copy
from typing import Any, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._map: Dict[str, Any] = {}

    def register_module(self, name: Optional[str] = None, force: bool = False, module: Any = None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._map and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._map[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key: str):
        return self._map.get(key)

    def __contains__(self, key: str) -> bool:
        return key in self._map

    def build(self, cfg: Dict[str, Any], default_args: Optional[Dict[str, Any]] = None):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
        args = copy.deepcopy(dict(cfg))
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        typ = args.pop("type")
        cls = self._map.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**args)


MODULES = Registry("module")
MODELS = MODULES  # mmgen aliases the two (mmgen/models/builder.py)


def build_module(cfg, default_args=None):
    if isinstance(cfg, (list, tuple)):
        return [MODULES.build(c, default_args) for c in cfg]
    return MODULES.build(cfg, default_args)


def get_module_device(module):
    try:
        return next(module.parameters()).device
    except StopIteration:
        try:
            return next(module.buffers()).device
        except StopIteration as e:
            raise ValueError("module has neither parameters nor buffers") from e
