"""The reference's predefined ore / few-shot splits (ref:fewx/data/datasets/builtin.py:8-30), registered under $DETECTRON2_DATASETS
(default ./datasets).  Registration is lazy: nothing is read until DatasetCatalog.get(name)."""
import os

from .register_coco import register_coco_instances

_PREDEFINED_SPLITS_COCO = {
    "coco": {
        "coco_2017_train_nonvoc": ("coco/train2017", "coco/new_annotations/final_split_non_voc_instances_train2017.json"),
        "coco_2017_train_voc_10_shot": ("coco/train2017", "coco/new_annotations/final_split_voc_10_shot_instances_train2017.json"),
        "coco_2017_val_stone": ("coco/val2017", "coco/annotations/instances_val2017.json"),
        "coco_2017_train_stone": ("coco/train2017", "coco/annotations/instances_train2017.json"),
    }
}


def register_all_coco(root):
    from detectron2.data import DatasetCatalog
    for _, splits in _PREDEFINED_SPLITS_COCO.items():
        for key, (image_root, json_file) in splits.items():
            if key in DatasetCatalog:
                continue
            register_coco_instances(key, {}, os.path.join(root, json_file) if "://" not in json_file else json_file,
                                    os.path.join(root, image_root))


register_all_coco(os.getenv("DETECTRON2_DATASETS", "datasets"))
