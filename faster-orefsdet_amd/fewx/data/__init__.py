"""fewx.data names (ref:fewx/data/__init__.py).  Decoding images, the support dataframe and augmentation are the data side of the
reference, outside the built hot path: the classes exist so `fsod_train_net.py` imports resolve and say so when
used.  tools/bench_train.py shows the batch layout the training step consumes."""
from .dataset_mapper import DatasetMapperWithSupport  # noqa: F401
from .build import build_detection_train_loader, build_detection_test_loader  # noqa: F401
