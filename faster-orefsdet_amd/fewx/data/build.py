"""build_detection_{train,test}_loader and fsod_get_detection_dataset_dicts (ref:fewx/data/build.py): the data-loader entry points
the reference's training script calls (ref:fsod_train_net.py:36-57), over the thin detectron2.data of this package.

The per-category split of the training records is pinned by tests/golden/dataset_split.npz (the reference's own function executed
on a synthetic registered dataset)."""
from detectron2.data import (DatasetCatalog, InferenceSampler, TrainingSampler, build_batch_data_loader,
                             trivial_batch_collator)


def _non_crowd_only(dicts):
    """d2z:data/build.py:38-66 filter_images_with_only_crowd_annotations: keep records with at least one non-crowd annotation."""
    return [d for d in dicts if any(a.get("iscrowd", 0) == 0 for a in d["annotations"])]


def fsod_get_detection_dataset_dicts(dataset_names, filter_empty=True, min_keypoints=0, proposal_files=None):
    """ref:fewx/data/build.py:27-106.  Names without 'train' (first name decides, as there): the registered records, concatenated.
    Training names: records with only crowd annotations are dropped, then every image record is split into ONE RECORD PER CATEGORY
    (few-shot episodes are per class) holding file_name / height / width / that category's annotations -- image_id is not carried
    over, `segmentation` / `keypoints` are removed from the annotations -- and records left with only crowd annotations are dropped
    again when filter_empty."""
    assert len(dataset_names)
    assert proposal_files is None and min_keypoints == 0, "precomputed proposals / keypoints are outside the fsod configs"
    per_name = [DatasetCatalog.get(n) for n in dataset_names]
    for n, d in zip(dataset_names, per_name):
        assert len(d), "Dataset '{}' is empty!".format(n)
    flat = [r for d in per_name for r in d]
    if "train" not in dataset_names[0]:
        out = flat
    else:
        out = []
        for rec in _non_crowd_only(flat):
            by_cat = {}
            for ann in rec["annotations"]:
                ann.pop("segmentation", None)
                ann.pop("keypoints", None)
                by_cat.setdefault(ann["category_id"], []).append(ann)
            for anns in by_cat.values():                       # insertion order = first appearance in the image, as the reference
                out.append({"file_name": rec["file_name"], "height": rec["height"], "width": rec["width"], "annotations": anns})
    if filter_empty and out and "annotations" in out[0] and "sem_seg_file_name" not in out[0]:
        out = _non_crowd_only(out)
    return out


class _Mapped:
    def __init__(self, dicts, mapper):
        self.dicts, self.mapper = dicts, mapper

    def __len__(self):
        return len(self.dicts)

    def __getitem__(self, i):
        return self.mapper(self.dicts[i])


def _default_mapper(cfg, is_train):
    """The reference falls back to detectron2's plain DatasetMapper(cfg, is_train) (ref:fewx/data/build.py:137-138,188-189): read,
    resize (+ flip when training), boxes -> Instances; no support branch.  That is DatasetMapperWithSupport with its support side off."""
    from detectron2.data import DatasetMapper
    return DatasetMapper(cfg, is_train)


def build_detection_train_loader(cfg, mapper=None):
    """ref:fewx/data/build.py:108-160: TrainingSampler over the per-category records, batches grouped by orientation when
    cfg.DATALOADER.ASPECT_RATIO_GROUPING (detectron2's default True, which the fsod configs keep -- ref:fewx/data/build.py:158).
    RepeatFactorTrainingSampler is not used by the fsod configs and raises."""
    dicts = fsod_get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    if mapper is None:
        mapper = _default_mapper(cfg, True)
    ds = _Mapped(dicts, mapper)
    name = cfg.DATALOADER.SAMPLER_TRAIN
    if name != "TrainingSampler":
        raise ValueError("Unknown training sampler: {}".format(name))
    return build_batch_data_loader(ds, TrainingSampler(len(ds)), cfg.SOLVER.IMS_PER_BATCH,
                                   aspect_ratio_grouping=cfg.DATALOADER.ASPECT_RATIO_GROUPING, num_workers=cfg.DATALOADER.NUM_WORKERS)


def build_detection_test_loader(cfg, dataset_name, mapper=None):
    """ref:fewx/data/build.py:162-204: the named dataset unfiltered, batch size 1, contiguous shard per rank."""
    import torch.utils.data as torchdata
    dicts = list(DatasetCatalog.get(dataset_name))            # d2's get_detection_dataset_dicts(filter_empty=False): never split
    assert len(dicts), "Dataset '{}' is empty!".format(dataset_name)
    if mapper is None:
        mapper = _default_mapper(cfg, False)
    ds = _Mapped(dicts, mapper)
    sampler = InferenceSampler(len(ds))
    return torchdata.DataLoader(ds, num_workers=cfg.DATALOADER.NUM_WORKERS,
                                batch_sampler=torchdata.sampler.BatchSampler(sampler, 1, drop_last=False),
                                collate_fn=trivial_batch_collator)
