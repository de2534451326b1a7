"""build_detection_{train,test}_loader (ref:fewx/data/build.py): thin wrappers over detectron2.data for registered datasets."""
from detectron2.data import DatasetCatalog, InferenceSampler, TrainingSampler, build_batch_data_loader, trivial_batch_collator


class _Mapped:
    def __init__(self, dicts, mapper):
        self.dicts, self.mapper = dicts, mapper

    def __len__(self):
        return len(self.dicts)

    def __getitem__(self, i):
        return self.mapper(self.dicts[i])


def build_detection_train_loader(cfg, mapper=None):
    names = cfg.DATASETS.TRAIN
    dicts = [d for n in names for d in DatasetCatalog.get(n)]
    assert dicts and mapper is not None, "register the dataset (DatasetCatalog) and pass a mapper"
    ds = _Mapped(dicts, mapper)
    return build_batch_data_loader(ds, TrainingSampler(len(ds)), cfg.SOLVER.IMS_PER_BATCH, num_workers=cfg.DATALOADER.NUM_WORKERS)


def build_detection_test_loader(cfg, dataset_name, mapper=None):
    import torch.utils.data as torchdata
    dicts = DatasetCatalog.get(dataset_name)
    assert mapper is not None, "pass a mapper"
    ds = _Mapped(dicts, mapper)
    return torchdata.DataLoader(ds, batch_size=1, sampler=InferenceSampler(len(ds)), num_workers=cfg.DATALOADER.NUM_WORKERS,
                                collate_fn=trivial_batch_collator)
