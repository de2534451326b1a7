"""The few d2z:data names the reference's entry script and dataset registration import (ref:fsod_train_net.py:16,
ref:fewx/data/datasets/builtin.py).  Dataset decoding / augmentation is outside the hot path: the catalogs and
the batching helper are real, image I/O is not provided."""
import itertools

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
    if aspect_ratio_grouping:
        raise NotImplementedError("aspect-ratio grouping belongs to the data pipeline (SURVEY 8f row 3)")
    return torchdata.DataLoader(dataset, sampler=sampler, num_workers=num_workers, batch_size=total_batch_size // world, drop_last=True,
                                collate_fn=trivial_batch_collator)


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


def DatasetMapper(cfg, is_train=True, **kw):
    """d2z:data/dataset_mapper.py DatasetMapper(cfg, is_train) for box-only configs: read the image, ResizeShortestEdge (+ RandomFlip
    when training), annotations -> Instances.  Implemented by the few-shot mapper with its support branch switched off."""
    from fewx.data.dataset_mapper import DatasetMapperWithSupport
    import pandas as pd
    m = DatasetMapperWithSupport(cfg, is_train, support_df=pd.DataFrame() if is_train else None, **kw)
    m.support_on = False
    return m
