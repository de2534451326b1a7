"""register_coco_instances (ref:fewx/data/datasets/register_coco.py:16-41): a COCO-format json registered lazily in the DatasetCatalog.
The loader restates what the reference gets from d2z:data/datasets/coco.py `load_coco_json(json, image_root, name,
extra_annotation_keys=['id'])` with the plain json module (pycocotools is not needed to read the file): one dict per image with
file_name / height / width / image_id and its annotations {bbox (XYWH_ABS), bbox_mode, category_id (contiguous), iscrowd, id}."""
import json
import os

from detectron2.data import DatasetCatalog, MetadataCatalog

XYWH_ABS = 1          # BoxMode.XYWH_ABS


def load_coco_json(json_file, image_root, dataset_name=None, extra_annotation_keys=None):
    with open(json_file) as f:
        data = json.load(f)
    cats = sorted(data.get("categories", []), key=lambda c: c["id"])
    id_map = {c["id"]: i for i, c in enumerate(cats)}
    if dataset_name is not None:
        MetadataCatalog.get(dataset_name).set(thing_classes=[c["name"] for c in cats], thing_dataset_id_to_contiguous_id=id_map)
    per_image = {}
    for a in data.get("annotations", []):
        per_image.setdefault(a["image_id"], []).append(a)
    keys = ["iscrowd", "bbox", "category_id"] + list(extra_annotation_keys or [])
    out = []
    for img in sorted(data.get("images", []), key=lambda i: i["id"]):
        rec = {"file_name": os.path.join(image_root, img["file_name"]), "height": img["height"], "width": img["width"], "image_id": img["id"]}
        objs = []
        for a in per_image.get(img["id"], []):
            assert a.get("ignore", 0) == 0, '"ignore" in COCO json file is not supported.'
            o = {k: a[k] for k in keys if k in a}
            o["bbox_mode"] = XYWH_ABS
            o["category_id"] = id_map[o["category_id"]]
            objs.append(o)
        rec["annotations"] = objs
        out.append(rec)
    return out


def register_coco_instances(name, metadata, json_file, image_root):
    assert isinstance(name, str), name
    assert isinstance(json_file, (str, os.PathLike)), json_file
    assert isinstance(image_root, (str, os.PathLike)), image_root
    DatasetCatalog.register(name, lambda: load_coco_json(json_file, image_root, name, extra_annotation_keys=["id"]))
    MetadataCatalog.get(name).set(json_file=json_file, image_root=image_root, evaluator_type="coco", **metadata)
