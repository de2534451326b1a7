"""CfgNode on PyYAML: attribute access, `_BASE_` inheritance, merge_from_file / merge_from_list, freeze, clone, dump.
Replaces the yacs/fvcore CfgNode the reference uses (d2z:config/config.py); same user-visible behaviour for the
`configs/fsod/*.yaml` overlays and `KEY VALUE` command-line opts (ref:fsod_train_net.py:76-89)."""
from __future__ import annotations

import ast
import copy
import os
from typing import Any, List

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute protocol
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def __setitem__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {k}, but CfgNode is immutable")
        super().__setitem__(k, v)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def _set_frozen(self, f):
        object.__setattr__(self, "_frozen", f)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(f)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        n = CfgNode()
        for k, v in self.items():
            dict.__setitem__(n, k, copy.deepcopy(v, memo))
        return n

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)) for k, v in self.items()}

    def dump(self, **kw):
        return yaml.safe_dump(self.to_dict(), **kw)

    # merging
    @staticmethod
    def load_yaml_with_base(filename: str) -> dict:
        with open(filename) as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = CfgNode.load_yaml_with_base(base)
            _merge_dict(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, filename: str, allow_unsafe: bool = True):
        self.merge_from_other_cfg(CfgNode.load_yaml_with_base(filename))

    def merge_from_other_cfg(self, other):
        _merge_into(other, self, [])

    def merge_from_list(self, opts: List[Any]):
        assert len(opts) % 2 == 0, f"opts must be KEY VALUE pairs, got {opts}"
        for full_key, v in zip(opts[0::2], opts[1::2]):
            node = self
            keys = full_key.split(".")
            for k in keys[:-1]:
                assert k in node, f"Non-existent key: {full_key}"
                node = node[k]
            assert keys[-1] in node, f"Non-existent key: {full_key}"
            node[keys[-1]] = _coerce(_decode(v), node[keys[-1]], full_key)


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key):
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return list(new)
    if old is None or new is None or type(new) is type(old):
        return new
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, str) and not isinstance(new, str):
        return str(new)
    raise ValueError(f"Type mismatch for {key}: {type(old).__name__} vs {type(new).__name__} ({new!r})")


def _merge_dict(src: dict, dst: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_dict(v, dst[k])
        else:
            dst[k] = v


def _merge_into(src, dst: CfgNode, path):
    for k, v in src.items():
        full = ".".join(path + [k])
        if k not in dst:
            raise KeyError(f"Non-existent config key: {full}")
        if isinstance(v, dict):
            assert isinstance(dst[k], CfgNode), full
            _merge_into(v, dst[k], path + [k])
        else:
            dst[k] = _coerce(_decode(v), dst[k], full)


def get_cfg() -> CfgNode:
    from .defaults import DEFAULTS
    return CfgNode(copy.deepcopy(DEFAULTS))


def configurable(init_func=None, *, from_config=None):
    """Call `Cls(cfg, ...)` -> `Cls(**Cls.from_config(cfg, ...))`; explicit kwargs pass straight through."""
    import functools
    import inspect

    def wrap(f, fc):
        @functools.wraps(f)
        def wrapped(self, *args, **kwargs):
            cfg_like = (args and isinstance(args[0], CfgNode)) or isinstance(kwargs.get("cfg"), CfgNode)
            if cfg_like:
                conf = (fc or type(self).from_config)(*args, **kwargs)
                f(self, **conf)
            else:
                f(self, *args, **kwargs)
        return wrapped

    if init_func is not None:
        assert inspect.isfunction(init_func) and from_config is None
        return wrap(init_func, None)
    return lambda f: wrap(f, from_config)
