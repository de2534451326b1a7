"""Config surface: our CfgNode + defaults + configs/fsod/finetune_vovnet.yaml must resolve to exactly the config the
reference logged for its 25-shot VoVNet run (tests/golden/vovnet_25shot_full_config.yaml, extracted from
ref:log/fsod_finetune_stone_vovnet_25_test_log.txt:117-544)."""
import os

import pytest
import yaml

from conftest import GOLDEN, PKG


def _cfg():
    from fewx.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(PKG, "configs", "fsod", "finetune_vovnet.yaml"))
    return cfg


def test_resolved_config_matches_reference_log():
    cfg = _cfg()
    # the logged run passed MODEL.WEIGHTS on the command line (ref:all.sh:23)
    cfg.merge_from_list(["MODEL.WEIGHTS", "./output/fsod/finetune_dir/vovnet_25shot/model_final.pth"])
    want = yaml.safe_load(open(os.path.join(GOLDEN, "vovnet_25shot_full_config.yaml")))
    got = cfg.to_dict()

    def diff(a, b, path=""):
        out = []
        for k in sorted(set(a) | set(b)):
            p = f"{path}.{k}" if path else k
            if k not in a or k not in b:
                out.append(f"{p}: only in {'golden' if k in a else 'ours'}")
            elif isinstance(a[k], dict) and isinstance(b[k], dict):
                out += diff(a[k], b[k], p)
            elif a[k] != b[k] and not (isinstance(a[k], (int, float)) and isinstance(b[k], (int, float)) and float(a[k]) == float(b[k])):
                out.append(f"{p}: golden {a[k]!r} != ours {b[k]!r}")
        return out

    d = diff(want, got)
    assert not d, "\n".join(d)


def test_freeze_merge_list_and_errors():
    cfg = _cfg()
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", "16", "MODEL.DEVICE", "cpu", "MODEL.CENTERNET.NMS_TH_TEST", 0.5])
    assert cfg.SOLVER.IMS_PER_BATCH == 16 and cfg.MODEL.DEVICE == "cpu" and cfg.MODEL.CENTERNET.NMS_TH_TEST == 0.5
    with pytest.raises(AssertionError):
        cfg.merge_from_list(["MODEL.NOPE", 1])
    with pytest.raises(ValueError):
        cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", "abc"])
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.SOLVER.BASE_LR = 1.0
    c2 = cfg.clone()
    assert c2.INPUT.FS.SUPPORT_SHOT == 24 and cfg.MODEL.META_ARCHITECTURE == "CenterNet2Detector"
