from . import builtin  # noqa: F401  (registers the predefined splits, like ref:fewx/data/datasets/__init__.py)
from .register_coco import load_coco_json, register_coco_instances  # noqa: F401
