"""Drop-in surface: every name the reference's entry script imports (ref:fsod_train_net.py:12-34) resolves against
faster-orefsdet_amd/, and the thin host-side pieces behave like the originals.  No GPU needed."""
import sys

import numpy as np
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


def _synth_support_df(seed=7, n_img=30, per_img=3):
    """Same stand-in dataframe as oracle/refrun/gen_golden.py::synth_support_df (the fixture's inputs are regenerated from the seed)."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    rows, aid = [], 1000
    for img in range(n_img):
        for k in range(per_img):
            x0, y0 = rng.uniform(10, 90, 2)
            rows.append({"id": aid, "image_id": 500 + img, "category_id": 1 + (img % 3), "file_path": f"support/{aid}.jpg",
                         "support_box": [float(x0), float(y0), float(x0 + rng.uniform(40, 120)), float(y0 + rng.uniform(40, 120))]})
            aid += 1
    return pd.DataFrame(rows)


def _synth_crop(path, format=None):
    import zlib
    rng = np.random.default_rng(zlib.crc32(path.encode()))            # the full path, as the reference passes it to read_image
    return rng.integers(0, 256, (240, 240, 3), dtype=np.uint8)


def test_generate_support_matches_reference_run(golden):
    """f3: DatasetMapperWithSupport.generate_support against the reference's own method (ref:fewx/data/dataset_mapper.py:198-269)
    executed on the same synthetic support dataframe: the drawn support annotations, their order, boxes, class flags, 1- and 2-way."""
    from fewx.config import get_cfg
    from fewx.data.dataset_mapper import DatasetMapperWithSupport
    g = golden("generate_support")
    df = _synth_support_df(int(g["seed"]))
    for tag in ("w1", "w2"):
        way, shot = (int(v) for v in g[f"{tag}_way_shot"])
        cfg = get_cfg()
        cfg.merge_from_list(["INPUT.FS.SUPPORT_WAY", way, "INPUT.FS.SUPPORT_SHOT", shot])
        mp = DatasetMapperWithSupport(cfg, is_train=True, support_df=df, read_image=_synth_crop)
        data, boxes, cls = mp.generate_support({"annotations": [{"id": int(i)} for i in g[f"{tag}_query_ids"]]})
        assert data.shape == (way * shot, 3, 240, 240) and data.dtype == np.float32
        np.testing.assert_array_equal(boxes, g[f"{tag}_boxes"])
        assert list(cls) == list(g[f"{tag}_cls"])
        np.testing.assert_array_equal(data[:, :, ::60, ::60], g[f"{tag}_pix"])


def test_dataset_mapper_call_produces_the_model_input_dict(tmp_path):
    """__call__ end to end on a synthetic image file: resize / flip, box transform, Instances, support tensors in the layout
    train_forward consumes; a missing support dataframe raises instead of inventing data."""
    from PIL import Image
    from fewx.config import get_cfg
    from fewx.data.dataset_mapper import DatasetMapperWithSupport
    cfg = get_cfg()
    cfg.merge_from_list(["INPUT.FS.SUPPORT_WAY", 1, "INPUT.FS.SUPPORT_SHOT", 4, "INPUT.MIN_SIZE_TRAIN", (320,), "INPUT.MAX_SIZE_TRAIN", 640])
    with pytest.raises(FileNotFoundError):
        DatasetMapperWithSupport(cfg, is_train=True)
    img = np.random.default_rng(0).integers(0, 256, (200, 300, 3), dtype=np.uint8)
    f = str(tmp_path / "q.png")
    Image.fromarray(img).save(f)
    df = _synth_support_df()
    q = df.iloc[4]

    def reader(path, format=None):
        if path == f:
            return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
        return _synth_crop(path)
    mp = DatasetMapperWithSupport(cfg, is_train=True, support_df=df, read_image=reader)
    d = mp({"file_name": f, "height": 200, "width": 300, "image_id": int(q["image_id"]),
            "annotations": [{"id": int(q["id"]), "bbox": [30.0, 40.0, 100.0, 80.0], "bbox_mode": 1, "category_id": 0},
                            {"id": int(q["id"]) + 1, "bbox": [295.0, 10.0, 0.0, 50.0], "bbox_mode": 1, "category_id": 0}]})
    assert tuple(d["image"].shape) == (3, 320, 480) and d["image"].dtype == torch.uint8
    assert tuple(d["support_images"].shape) == (4, 3, 240, 240) and d["support_bboxes"].shape == (4, 4)
    inst = d["instances"]
    assert len(inst) == 1 and inst.image_size == (320, 480)                  # the zero-width box is dropped (filter_empty_instances)
    b = inst.gt_boxes.tensor[0]
    assert abs(float(b[2] - b[0]) - 160.0) < 1e-3 and abs(float(b[3] - b[1]) - 128.0) < 1e-3   # 100 x 80 box scaled by 1.6


def test_coco_json_registration_and_loader(tmp_path):
    """ref:fewx/data/datasets/builtin.py + register_coco.py: the reference's split names are registered lazily; a COCO json loads into
    the dataset-dict layout the mapper consumes (annotation ids kept: generate_support keys on them)."""
    import json
    import fewx.data  # noqa: F401
    from detectron2.data import DatasetCatalog, MetadataCatalog
    from fewx.data.datasets import register_coco_instances
    for k in ("coco_2017_train_stone", "coco_2017_val_stone", "coco_2017_train_nonvoc", "coco_2017_train_voc_10_shot"):
        assert k in DatasetCatalog
    with pytest.raises(FileNotFoundError):
        DatasetCatalog.get("coco_2017_val_stone")                      # registered lazily; the dataset itself is not shipped
    js = {"images": [{"id": 7, "file_name": "a.png", "height": 300, "width": 300}, {"id": 3, "file_name": "b.png", "height": 10, "width": 20}],
          "categories": [{"id": 5, "name": "ore"}],
          "annotations": [{"id": 11, "image_id": 7, "category_id": 5, "bbox": [1, 2, 30, 40], "iscrowd": 0, "area": 1200},
                          {"id": 12, "image_id": 7, "category_id": 5, "bbox": [5, 5, 10, 10], "iscrowd": 1, "area": 100}]}
    f = tmp_path / "inst.json"
    f.write_text(json.dumps(js))
    register_coco_instances("ore_test_split", {}, str(f), str(tmp_path / "img"))
    d = DatasetCatalog.get("ore_test_split")
    assert [r["image_id"] for r in d] == [3, 7] and d[0]["annotations"] == []
    a = d[1]["annotations"]
    assert d[1]["file_name"].endswith("img/a.png") and [x["id"] for x in a] == [11, 12] and a[0]["category_id"] == 0 and a[0]["bbox_mode"] == 1
    assert MetadataCatalog.get("ore_test_split").thing_classes == ["ore"] and MetadataCatalog.get("ore_test_split").evaluator_type == "coco"
    DatasetCatalog.remove("ore_test_split")


def test_per_category_split_matches_reference_run(golden):
    """ref:fewx/data/build.py:27-106 executed on a synthetic registered dataset (oracle/refrun/gen_golden.py::gen_dataset_split):
    the product's fsod_get_detection_dataset_dicts returns the same records, in the same order, for a 'train' name (split per
    category, crowd-only records dropped, segmentation / keypoints / image_id gone) and for any other name (pass-through)."""
    import copy
    import json
    from detectron2.data import DatasetCatalog
    from fewx.data.build import fsod_get_detection_dataset_dicts
    g = golden("dataset_split")
    dicts = json.loads(str(g["input_json"]))
    for n in ("synth_ore_train", "synth_ore_val"):
        if n in DatasetCatalog:
            DatasetCatalog.remove(n)
        DatasetCatalog.register(n, lambda: copy.deepcopy(dicts))
    try:
        assert fsod_get_detection_dataset_dicts(["synth_ore_train"], filter_empty=True) == json.loads(str(g["train_json"]))
        assert fsod_get_detection_dataset_dicts(["synth_ore_val"], filter_empty=False) == json.loads(str(g["test_json"]))
    finally:
        DatasetCatalog.remove("synth_ore_train")
        DatasetCatalog.remove("synth_ore_val")


def test_loader_builders_accept_the_reference_call_forms(tmp_path):
    """ref:fsod_train_net.py:36-57 calls `build_detection_train_loader(cfg, DatasetMapperWithSupport(cfg))` and
    `build_detection_test_loader(cfg, dataset_name)` -- the latter with NO mapper (ref:fewx/data/build.py:188-189 defaults it to
    DatasetMapper(cfg, False)).  Both forms must work and yield the list-of-dicts batches the model consumes."""
    from PIL import Image
    from detectron2.data import DatasetCatalog
    from fewx.config import get_cfg
    from fewx.data.build import build_detection_test_loader, build_detection_train_loader
    from fewx.data.dataset_mapper import DatasetMapperWithSupport
    rng = np.random.default_rng(1)
    files = []
    for i in range(3):
        f = str(tmp_path / f"im{i}.png")
        Image.fromarray(rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)).save(f)
        files.append(f)
    df = _synth_support_df()
    ids = df["id"].tolist()
    recs = [{"file_name": files[i], "height": 120, "width": 160, "image_id": int(df.iloc[i]["image_id"]),
             "annotations": [{"id": int(ids[i]), "bbox": [10.0, 20.0, 50.0, 40.0], "bbox_mode": 1, "category_id": int(df.iloc[i]["category_id"]), "iscrowd": 0}]}
            for i in range(3)]
    import copy
    for n in ("ore_unit_train", "ore_unit_val"):
        if n in DatasetCatalog:
            DatasetCatalog.remove(n)
        DatasetCatalog.register(n, lambda: copy.deepcopy(recs))
    cfg = get_cfg()
    cfg.merge_from_list(["DATASETS.TRAIN", ("ore_unit_train",), "DATASETS.TEST", ("ore_unit_val",), "DATALOADER.NUM_WORKERS", 0,
                         "INPUT.FS.SUPPORT_WAY", 1, "INPUT.FS.SUPPORT_SHOT", 2, "SOLVER.IMS_PER_BATCH", 1,
                         "INPUT.MIN_SIZE_TEST", 96, "INPUT.MAX_SIZE_TEST", 160, "INPUT.MIN_SIZE_TRAIN", (96,), "INPUT.MAX_SIZE_TRAIN", 160])

    def reader(path, format=None):
        if path in files:
            return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
        return _synth_crop(path)
    try:
        test_loader = build_detection_test_loader(cfg, "ore_unit_val")                 # the reference's form: no mapper
        batches = list(test_loader)
        assert len(batches) == 3 and all(len(b) == 1 for b in batches)
        d = batches[0][0]
        assert tuple(d["image"].shape) == (3, 96, 128) and d["height"] == 120 and d["width"] == 160 and "annotations" not in d
        train_loader = build_detection_train_loader(cfg, DatasetMapperWithSupport(cfg, support_df=df, read_image=reader))
        b = next(iter(train_loader))
        assert len(b) == 1 and tuple(b[0]["support_images"].shape) == (2, 3, 240, 240) and len(b[0]["instances"]) == 1
    finally:
        DatasetCatalog.remove("ore_unit_train")
        DatasetCatalog.remove("ore_unit_val")


def test_nan_loss_guard_of_the_train_loop():
    """d2z:engine/train_loop.py:336-341: a loss that is not finite stops training with FloatingPointError naming the iteration.
    The product looks at step i's losses behind step i+1's forward (no host sync inside a step); metrics_lag = 0 is the
    reference's blocking form.  Both raise for the same iteration; finite losses land in the EventStorage under their own
    iteration."""
    from detectron2.engine import SimpleTrainer
    from detectron2.utils.events import EventStorage

    class M(torch.nn.Module):
        def __init__(self, bad_at):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(()))
            self.calls, self.bad_at = 0, bad_at

        def forward(self, batch):
            v = self.w * (float("nan") if self.calls == self.bad_at else 1.0 + self.calls)
            self.calls += 1
            return {"loss_a": v, "loss_b": self.w * 0.5}

    def loader():
        while True:
            yield [{}]

    for lag, raised_in_call in ((1, 3), (0, 2)):
        m = M(bad_at=2)
        tr = SimpleTrainer(m, loader(), torch.optim.SGD(m.parameters(), lr=0.0))
        tr.metrics_lag = lag
        with EventStorage(0) as st:
            done = 0
            with pytest.raises(FloatingPointError, match=r"Loss became infinite or NaN at iteration=2!"):
                for tr.iter in range(5):
                    st.iter = tr.iter
                    tr.run_step()
                    done += 1
            assert done == raised_in_call
            assert st.history("total_loss") == [(1.5, 0), (2.5, 1)] and st.history("loss_a") == [(1.0, 0), (2.0, 1)]
    # the end-of-training flush sees the last step too
    m = M(bad_at=1)
    tr = SimpleTrainer(m, loader(), torch.optim.SGD(m.parameters(), lr=0.0))
    with EventStorage(0):
        tr.iter = 0
        tr.run_step()
        tr.iter = 1
        tr.run_step()
        with pytest.raises(FloatingPointError, match="iteration=1"):
            tr.flush_metrics()


def test_aspect_ratio_grouping_and_base_trainer_loaders(tmp_path):
    """d2z:data/build.py:286-295 + data/common.py:152-186: with cfg.DATALOADER.ASPECT_RATIO_GROUPING (detectron2's default, kept by
    the fsod configs -- ref:fewx/data/build.py:158) a batch never mixes landscape and portrait records and keeps the sampler's order
    inside an orientation; DefaultTrainer.build_train_loader / build_test_loader delegate to detectron2.data like
    d2z:engine/defaults.py:523-544."""
    from detectron2.data import (AspectRatioGroupedDataset, DatasetCatalog, TrainingSampler, build_batch_data_loader)
    from detectron2.engine import DefaultTrainer
    from fewx.config import get_cfg
    recs = [{"width": 4 if i % 3 else 2, "height": 3, "i": i} for i in range(12)]

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(recs)

        def __getitem__(self, i):
            return recs[i]

    order = list(range(12))
    got = []
    for b in build_batch_data_loader(DS(), order, 2, aspect_ratio_grouping=True):
        assert len(b) == 2 and len({d["width"] > d["height"] for d in b}) == 1
        got.append([d["i"] for d in b])
    assert got == [[1, 2], [0, 3], [4, 5], [7, 8], [6, 9], [10, 11]]
    assert isinstance(build_batch_data_loader(DS(), TrainingSampler(12), 2, aspect_ratio_grouping=True), AspectRatioGroupedDataset)
    plain = build_batch_data_loader(DS(), order, 3, aspect_ratio_grouping=False)
    assert [[d["i"] for d in b] for b in plain] == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]]

    from PIL import Image
    f = str(tmp_path / "im.png")
    Image.fromarray(np.zeros((120, 160, 3), dtype=np.uint8)).save(f)
    rec = {"file_name": f, "height": 120, "width": 160, "image_id": 1,
           "annotations": [{"bbox": [10.0, 20.0, 50.0, 40.0], "bbox_mode": 1, "category_id": 0, "iscrowd": 0}]}
    crowd = dict(rec, annotations=[dict(rec["annotations"][0], iscrowd=1)])
    import copy
    for n in ("ore_base_train", "ore_base_val"):
        if n in DatasetCatalog:
            DatasetCatalog.remove(n)
        DatasetCatalog.register(n, lambda: copy.deepcopy([rec, crowd]))
    cfg = get_cfg()
    cfg.merge_from_list(["DATASETS.TRAIN", ("ore_base_train",), "DATASETS.TEST", ("ore_base_val",), "DATALOADER.NUM_WORKERS", 0,
                         "SOLVER.IMS_PER_BATCH", 1, "INPUT.MIN_SIZE_TEST", 96, "INPUT.MAX_SIZE_TEST", 160,
                         "INPUT.MIN_SIZE_TRAIN", (96,), "INPUT.MAX_SIZE_TRAIN", 160])
    try:
        assert cfg.DATALOADER.ASPECT_RATIO_GROUPING is True
        b = next(iter(DefaultTrainer.build_train_loader(cfg)))
        assert len(b) == 1 and tuple(b[0]["image"].shape) == (3, 96, 128) and len(b[0]["instances"]) == 1
        batches = list(DefaultTrainer.build_test_loader(cfg, "ore_base_val"))
        assert len(batches) == 2 and batches[0][0]["width"] == 160                  # test side: unfiltered (crowd-only record kept)
    finally:
        DatasetCatalog.remove("ore_base_train")
        DatasetCatalog.remove("ore_base_val")
