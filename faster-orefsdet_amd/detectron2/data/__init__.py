"""The few d2z:data names the reference's entry script and dataset registration import (ref:fsod_train_net.py:16,
ref:fewx/data/datasets/builtin.py).  Dataset decoding / augmentation is outside the hot path: the catalogs and
the batching helper are real, image I/O is not provided."""
import itertools
import operator

import torch
import torch.utils.data as torchdata


class _Catalog(dict):
    def register(self, name, func):
        assert callable(func), "You must register a function with `DatasetCatalog.register`!"
        assert name not in self, "Dataset '{}' is already registered!".format(name)
        self[name] = func

    def get(self, name):
        try:
            f = self[name]
        except KeyError as e:
            raise KeyError("Dataset '{}' is not registered! Available datasets are: {}".format(name, ", ".join(self.keys()))) from e
        return f()

    def list(self):
        return list(self.keys())

    def remove(self, name):
        self.pop(name)


class Metadata(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError("Attribute '{}' does not exist in the metadata of dataset '{}'".format(k, self.get("name")))

    def __setattr__(self, k, v):
        self[k] = v

    def set(self, **kw):
        self.update(kw)
        return self


class _MetadataCatalog(dict):
    def get(self, name):
        if name not in self:
            self[name] = Metadata(name=name)
        return self[name]


DatasetCatalog = _Catalog()
MetadataCatalog = _MetadataCatalog()


def trivial_batch_collator(batch):
    return batch


def build_batch_data_loader(dataset, sampler, total_batch_size, *, aspect_ratio_grouping=False, num_workers=0):
    """d2z:data/build.py:250-300: per-process batch = total_batch_size / world_size, list-of-dicts batches (no collation)."""
    from detectron2.utils.comm import get_world_size
    world = get_world_size()
    assert total_batch_size > 0 and total_batch_size % world == 0, "Total batch size ({}) must be divisible by the number of gpus ({}).".format(total_batch_size, world)
    batch = total_batch_size // world
    if aspect_ratio_grouping:
        # one mapped dict at a time, then two buckets by orientation (d2z:data/build.py:286-295, data/common.py:152-186)
        one = torchdata.DataLoader(dataset, sampler=sampler, num_workers=num_workers, batch_sampler=None, collate_fn=operator.itemgetter(0))
        return AspectRatioGroupedDataset(one, batch)
    return torchdata.DataLoader(dataset, sampler=sampler, num_workers=num_workers, batch_size=batch, drop_last=True,
                                collate_fn=trivial_batch_collator)


class AspectRatioGroupedDataset(torchdata.IterableDataset):
    """d2z:data/common.py:152-186: records whose width > height and the rest are batched separately (less padding inside a
    batch); a bucket is emitted when it holds `batch_size` records, so a batch never mixes orientations and the order inside a
    bucket is the sampler's."""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, batch_size
        self._buckets = ([], [])

    def __iter__(self):
        for d in self.dataset:
            bucket = self._buckets[0 if d["width"] > d["height"] else 1]
            bucket.append(d)
            if len(bucket) == self.batch_size:
                yield bucket[:]
                del bucket[:]


class InferenceSampler(torchdata.Sampler):
    """Contiguous shard of [0, size) per rank (d2z:data/samplers/distributed_sampler.py:175-200)."""

    def __init__(self, size: int):
        from detectron2.utils.comm import shard_range
        self._size = size
        b, e = shard_range(size)
        self._local = range(b, e)

    def __iter__(self):
        yield from self._local

    def __len__(self):
        return len(self._local)


class TrainingSampler(torchdata.Sampler):
    """Infinite stream of shuffled indices, strided over ranks (d2z:data/samplers/distributed_sampler.py:12-72)."""

    def __init__(self, size: int, shuffle: bool = True, seed: int = 0):
        from detectron2.utils import comm
        self._size, self._shuffle, self._seed = size, shuffle, int(seed)
        self._rank, self._world = comm.get_rank(), comm.get_world_size()

    def __iter__(self):
        yield from itertools.islice(self._infinite(), self._rank, None, self._world)

    def _infinite(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            yield from (torch.randperm(self._size, generator=g) if self._shuffle else torch.arange(self._size)).tolist()


def get_detection_dataset_dicts(names, filter_empty=True, min_keypoints=0, proposal_files=None):
    """d2z:data/build.py:207-257 for box-only datasets: the registered records of every name, concatenated; records whose
    annotations are all crowd are dropped when filter_empty."""
    if isinstance(names, str):
        names = [names]
    assert len(names), names
    assert proposal_files is None and min_keypoints == 0, "precomputed proposals / keypoints are outside the fsod configs"
    per_name = [DatasetCatalog.get(n) for n in names]
    for n, d in zip(names, per_name):
        assert len(d), "Dataset '{}' is empty!".format(n)
    dicts = list(itertools.chain.from_iterable(per_name))
    if filter_empty and "annotations" in dicts[0]:
        dicts = [d for d in dicts if any(a.get("iscrowd", 0) == 0 for a in d["annotations"])]
    assert len(dicts), "No valid data found in {}.".format(",".join(names))
    return dicts


class MapDataset(torchdata.Dataset):
    def __init__(self, dicts, mapper):
        self.dicts, self.mapper = dicts, mapper

    def __len__(self):
        return len(self.dicts)

    def __getitem__(self, i):
        return self.mapper(self.dicts[i])


def build_detection_train_loader(cfg, mapper=None):
    """d2z:data/build.py:302-386 (what the base DefaultTrainer.build_train_loader calls): plain records, TrainingSampler,
    orientation-grouped batches per cfg.DATALOADER.ASPECT_RATIO_GROUPING."""
    dicts = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    if cfg.DATALOADER.SAMPLER_TRAIN != "TrainingSampler":
        raise ValueError("Unknown training sampler: {}".format(cfg.DATALOADER.SAMPLER_TRAIN))
    ds = MapDataset(dicts, mapper if mapper is not None else DatasetMapper(cfg, True))
    return build_batch_data_loader(ds, TrainingSampler(len(ds)), cfg.SOLVER.IMS_PER_BATCH,
                                   aspect_ratio_grouping=cfg.DATALOADER.ASPECT_RATIO_GROUPING, num_workers=cfg.DATALOADER.NUM_WORKERS)


def build_detection_test_loader(cfg, dataset_name, mapper=None):
    """d2z:data/build.py:389-440: the named dataset unfiltered, one image per batch, contiguous shard per rank."""
    dicts = get_detection_dataset_dicts([dataset_name], filter_empty=False)
    ds = MapDataset(dicts, mapper if mapper is not None else DatasetMapper(cfg, False))
    return torchdata.DataLoader(ds, num_workers=cfg.DATALOADER.NUM_WORKERS,
                                batch_sampler=torchdata.sampler.BatchSampler(InferenceSampler(len(ds)), 1, drop_last=False),
                                collate_fn=trivial_batch_collator)


def DatasetMapper(cfg, is_train=True, **kw):
    """d2z:data/dataset_mapper.py DatasetMapper(cfg, is_train) for box-only configs: read the image, ResizeShortestEdge (+ RandomFlip
    when training), annotations -> Instances.  Implemented by the few-shot mapper with its support branch switched off."""
    from fewx.data.dataset_mapper import DatasetMapperWithSupport
    import pandas as pd
    m = DatasetMapperWithSupport(cfg, is_train, support_df=pd.DataFrame() if is_train else None, **kw)
    m.support_on = False
    return m
