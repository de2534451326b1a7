"""fewx.data names (ref:fewx/data/__init__.py): the mapper that turns a dataset dict into the model's input dict, the loaders, and
the registration of the reference's predefined splits (host logic; the ore dataset itself is not shipped with the reference)."""
from .dataset_mapper import DatasetMapperWithSupport  # noqa: F401
from .build import build_detection_train_loader, build_detection_test_loader  # noqa: F401
from . import datasets  # noqa: F401  (ensure the builtin datasets are registered, as the reference does)
