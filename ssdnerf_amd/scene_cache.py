"""Scene cache and its wire format.

A cached scene is the dict the reference passes between its training loop, its RAM / file cache and ``torch.save``
(reference: lib/models/autodecoders/multiscene_nerf.py:17-28 ``out_dict_to``, :74-183 ``load_cache`` / ``save_cache``;
casting helpers lib/core/utils/misc.py:43-128):

    {scene_id, scene_name,
     param:     {code_ (pre-activation, fp32 or fp16), density_grid (Morton, fp16), density_bitfield (uint8)},
     optimizer: state_dict() of the per-scene code optimizer (fp32, or bf16 in the 16-bit cache; 'step' keeps its dtype)}

``cache_16bit`` stores the code in fp16 and the optimizer moments in bf16, clamping to the target type's finite range first;
``density_grid`` / ``density_bitfield`` / ``step`` are never cast.  Test-time files written by ``BaseNeRF.save_scene`` carry the
ACTIVATED ``code`` instead of ``code_``; ``load_cache`` inverts the activation for those, ``load_scene`` uses them as they are."""
from __future__ import annotations

import atexit
import os
import queue
import threading
from collections import abc as container_abcs
from collections import defaultdict
from itertools import chain
from typing import Dict, List

import torch

_UNCAST_KEYS = ("density_grid", "density_bitfield", "step")


def _clamp_to(val: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    return val.clamp(min=torch.finfo(dtype).min, max=torch.finfo(dtype).max)


def optimizer_state_to(state_dict: Dict, device=None, dtype: torch.dtype = torch.float32) -> Dict:
    """Copy of an optimizer ``state_dict()`` on ``device`` with every tensor but ``step`` cast to ``dtype`` (clamped to its finite
    range when the dtype changes); ``param_groups`` is shared, not copied (misc.py:43-59)."""
    assert dtype.is_floating_point
    out = dict(state=dict(), param_groups=state_dict["param_groups"])
    for pid, st in state_dict["state"].items():
        o = dict()
        for key, val in st.items():
            if isinstance(val, torch.Tensor):
                if key != "step" and val.dtype != dtype:
                    val = _clamp_to(val, dtype)
                o[key] = val.to(device=device, dtype=None if key == "step" else dtype)
            else:
                o[key] = val
        out["state"][pid] = o
    return out


def load_tensor_to_dict(d: Dict, key: str, value, device=None, dtype: torch.dtype = torch.float32) -> None:
    """Store ``value`` under ``d[key]``: in place (``copy_``, keeping the existing tensor's dtype/device) when the key exists, else as a
    new tensor cast like ``optimizer_state_to`` does; grids, bitfields and step counters keep their dtype (misc.py:63-75)."""
    assert dtype.is_floating_point
    if isinstance(value, torch.Tensor):
        if key not in _UNCAST_KEYS and value.dtype != dtype:
            value = _clamp_to(value, dtype)
        if key in d:
            d[key].copy_(value)
        else:
            d[key] = value.to(device=device, dtype=None if key in _UNCAST_KEYS else dtype)
    else:
        d[key] = value


def optimizer_state_copy(d_src: Dict, d_dst: Dict, device=None, dtype: torch.dtype = torch.float32) -> None:
    """Refresh a cached optimizer state in place from a live ``state_dict()`` (misc.py:78-85)."""
    d_dst["param_groups"] = d_src["param_groups"]
    for pid, st in d_src["state"].items():
        if pid not in d_dst["state"]:
            d_dst["state"][pid] = dict()
        for key, val in st.items():
            load_tensor_to_dict(d_dst["state"][pid], key, val, device=device, dtype=dtype)


def optimizer_set_state(optimizer: torch.optim.Optimizer, state_dict: Dict) -> None:
    """``Optimizer.load_state_dict`` for the STATE only - the live ``param_groups`` (learning rate ...) stay as built from the current
    config (misc.py:88-128).  Moments are cast to the parameter's dtype and device, ``step`` is left alone."""
    groups, saved_groups = optimizer.param_groups, state_dict["param_groups"]
    if len(groups) != len(saved_groups):
        raise ValueError("loaded state dict has a different number of parameter groups")
    if any(len(g["params"]) != len(s["params"]) for g, s in zip(groups, saved_groups)):
        raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
    id_map = {old: p for old, p in zip(chain.from_iterable(g["params"] for g in saved_groups), chain.from_iterable(g["params"] for g in groups))}

    def cast(param, value, key=None):
        if isinstance(value, torch.Tensor):
            if key != "step":
                if param.is_floating_point():
                    value = value.to(param.dtype)
                value = value.to(param.device)
            return value
        if isinstance(value, dict):
            return {k: cast(param, v, key=k) for k, v in value.items()}
        if isinstance(value, container_abcs.Iterable) and not isinstance(value, str):
            return type(value)(cast(param, v) for v in value)
        return value

    state = defaultdict(dict)
    for k, v in state_dict["state"].items():
        if k in id_map:
            state[id_map[k]] = cast(id_map[k], v)
        else:
            state[k] = v
    optimizer.__setstate__({"state": state})


def out_dict_to(d: Dict, device=None, code_dtype: torch.dtype = torch.float32, optimizer_dtype: torch.dtype = torch.float32) -> Dict:
    """One cached scene on ``device`` in the cache's storage types (multiscene_nerf.py:17-28)."""
    assert code_dtype.is_floating_point and optimizer_dtype.is_floating_point
    return dict(scene_id=d["scene_id"], scene_name=d["scene_name"],
                param=dict(code_=_clamp_to(d["param"]["code_"], code_dtype).to(device=device, dtype=code_dtype),
                           density_grid=d["param"]["density_grid"].to(device=device),
                           density_bitfield=d["param"]["density_bitfield"].to(device=device)),
                optimizer=optimizer_state_to(d["optimizer"], device=device, dtype=optimizer_dtype))


class _FileWriters:
    """Background ``torch.save`` of cached scenes (the reference forks ``num_file_writers`` processes fed by size-1 queues,
    multiscene_nerf.py:55-72; threads do the same job here - the work is serialisation + file IO, which releases the GIL)."""

    def __init__(self, save_dir: str, n: int):
        self.save_dir = save_dir
        self.queues = [queue.Queue(maxsize=1) for _ in range(n)]
        self.threads = [threading.Thread(target=self._run, args=(q,), daemon=True) for q in self.queues]
        for t in self.threads:
            t.start()
        atexit.register(self.flush)          # daemon threads: make sure queued scenes reach the disk before the interpreter goes away

    def _run(self, q):
        while True:
            obj = q.get()
            try:
                if obj is None:
                    return
                torch.save(obj, os.path.join(self.save_dir, obj["scene_name"] + ".pth"))
            finally:
                q.task_done()

    def put(self, slot: int, obj: Dict):
        self.queues[slot % len(self.queues)].put(obj)

    def flush(self):
        for q in self.queues:
            q.join()


def read_scene_files(paths: List[str]) -> List[Dict]:
    """``data['code']`` of the reference's dataset (lib/datasets/shapenet_srn.py ``code_dir`` branch): the per-scene dicts, on the CPU."""
    return [torch.load(p, map_location="cpu") for p in paths]
