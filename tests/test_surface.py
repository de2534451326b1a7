"""Drop-in surface: every name the reference's entry script imports (ref:fsod_train_net.py:12-34) resolves against
faster-orefsdet_amd/, and the thin host-side pieces behave like the originals.  No GPU needed."""
import sys

import pytest
import torch

from conftest import PKG

sys.path.insert(0, PKG)

REF_SCRIPT_IMPORTS = [
    "from detectron2.checkpoint import DetectionCheckpointer",
    "from detectron2.config import get_cfg",
    "from detectron2.engine import DefaultTrainer, default_argument_parser, default_setup, launch",
    "from detectron2.data import build_batch_data_loader",
    "from fewx.config import get_cfg",
    "from fewx.data.dataset_mapper import DatasetMapperWithSupport",
    "from fewx.data.build import build_detection_train_loader, build_detection_test_loader",
    "from fewx.solver import build_optimizer",
    "from fewx.evaluation import COCOEvaluator",
    "import detectron2.utils.comm as comm",
    "from detectron2.utils.logger import setup_logger",
    "import fewx.modeling",
]


@pytest.mark.parametrize("stmt", REF_SCRIPT_IMPORTS)
def test_reference_script_imports_resolve(stmt):
    exec(stmt, {})


def test_registries_hold_the_reference_names():
    import fewx.modeling  # noqa: F401
    from detectron2.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY
    from fewx.modeling.fsod.fsod_roi_heads import ROI_HEADS_REGISTRY
    for reg, names in ((META_ARCH_REGISTRY, ("CenterNet2Detector", "FsodRCNN")), (BACKBONE_REGISTRY, ("build_fcos_vovnet_fpn_backbone",)),
                       (PROPOSAL_GENERATOR_REGISTRY, ("CenterNet", "FsodRPN")),
                       (ROI_HEADS_REGISTRY, ("CustomCascadeROIHeads", "FsodRes5ROIHeads", "CustomROIHeads"))):
        for n in names:
            assert reg.get(n) is not None


def test_inference_on_dataset_protocol():
    """d2z:evaluation/evaluator.py:101-221: eval mode inside, restored after, first min(5, n-1) batches not timed, evaluator hooks."""
    from detectron2.evaluation import DatasetEvaluator, inference_on_dataset

    class M(torch.nn.Module):
        def forward(self, batch):
            assert not self.training
            return [{"n": len(batch)}]

    class E(DatasetEvaluator):
        def reset(self):
            self.seen = 0

        def process(self, inputs, outputs):
            self.seen += outputs[0]["n"]

        def evaluate(self):
            return {"seen": self.seen}

    m = M().train()
    res = inference_on_dataset(m, [[{"image": i}] for i in range(9)], E())
    assert res == {"seen": 9} and m.training
    assert inference_on_dataset.last_timing["images"] == 4


def test_default_argument_parser_and_checkpointer(tmp_path):
    from detectron2.checkpoint import DetectionCheckpointer
    from detectron2.engine import default_argument_parser
    a = default_argument_parser().parse_args(["--config-file", "x.yaml", "--num-gpus", "8", "SOLVER.BASE_LR", "0.01"])
    assert a.config_file == "x.yaml" and a.num_gpus == 8 and a.opts == ["SOLVER.BASE_LR", "0.01"] and a.dist_url.startswith("tcp://127.0.0.1:")
    m = torch.nn.Linear(3, 2)
    ck = DetectionCheckpointer(m, str(tmp_path))
    assert not ck.has_checkpoint()
    ck.save("model_0000001", iteration=1)
    w = m.weight.detach().clone()
    with torch.no_grad():
        m.weight.zero_()
    out = ck.resume_or_load("", resume=True)
    assert out["iteration"] == 1 and torch.equal(m.weight, w)
    plain = str(tmp_path / "plain.pth")                           # reference checkpoints: {"model": state_dict}
    torch.save({"model": {"weight": torch.ones(2, 3), "bias": torch.zeros(2)}}, plain)
    DetectionCheckpointer(m).load(plain)
    assert torch.equal(m.weight, torch.ones(2, 3))


def test_samplers_shard_like_the_reference():
    from detectron2.data import InferenceSampler, TrainingSampler
    assert list(InferenceSampler(5)) == [0, 1, 2, 3, 4]
    it = iter(TrainingSampler(4, shuffle=True, seed=1))
    first = [next(it) for _ in range(8)]
    assert sorted(first[:4]) == [0, 1, 2, 3] and sorted(first[4:]) == [0, 1, 2, 3]


def test_label_and_sample_host_logic_matches_reference_run(golden):
    """The product's proposal labelling / sampling / box-delta host logic (fewx/modeling/fsod/train_forward.py) on the
    reference-run fixture (vendored detectron2 Matcher + subsample_labels + Box2BoxTransform): bit-exact indices."""
    import types
    import numpy as np
    from fewx.modeling.fsod import train_forward as TF
    g = golden("roi_train_pieces")
    gt, boxes = torch.from_numpy(g["gt"]), torch.from_numpy(g["boxes"])
    rh = types.SimpleNamespace(proposal_append_gt=True, iou_threshold=0.6, batch_size_per_image=128, positive_fraction=0.5)
    np.testing.assert_array_equal(TF.pairwise_iou(gt, boxes).numpy(), g["iou"])
    torch.manual_seed(int(g["seed"]))
    sampled, roi_boxes, roi_labels, roi_gt = TF.label_and_sample(rh, boxes[:-gt.shape[0]], gt, lambda n: torch.randperm(n))
    np.testing.assert_array_equal(sampled.numpy(), g["sampled"])
    np.testing.assert_array_equal(roi_labels.numpy(), g["labels"][g["sampled"]])
    np.testing.assert_array_equal(roi_gt.numpy(), g["gt"][g["matched_idx"][g["sampled"]]])
    fg = roi_labels == 0
    d = TF.get_deltas(roi_boxes[fg], roi_gt[fg], (10.0, 10.0, 5.0, 5.0))
    np.testing.assert_allclose(d.numpy(), g["deltas"], rtol=1e-6, atol=1e-6)
