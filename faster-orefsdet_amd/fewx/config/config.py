"""fewx.config.get_cfg: detectron2-level defaults + few-shot / CenterNet / VoVNet keys (ref:fewx/config/config.py:5-99,
ref:fewx/config/defaults.py)."""
import copy

from detectron2.config import CfgNode, get_cfg as _d2_get_cfg

from .defaults import FEWX_DEFAULTS


def _overlay(dst: CfgNode, src: dict):
    for k, v in src.items():
        if isinstance(v, dict):
            if k not in dst:
                dst[k] = CfgNode()
            _overlay(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def get_cfg() -> CfgNode:
    cfg = _d2_get_cfg()
    _overlay(cfg, FEWX_DEFAULTS)
    return cfg
