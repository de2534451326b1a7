"""fewx.evaluation.COCOEvaluator (ref:fewx/evaluation/coco_evaluation.py) needs pycocotools + the annotation json: outside the
built path.  The name resolves; constructing it says what is missing."""
from detectron2.evaluation import DatasetEvaluator


class COCOEvaluator(DatasetEvaluator):
    def __init__(self, dataset_name, cfg, distributed, output_dir=None):
        raise NotImplementedError("COCO-style AP evaluation needs pycocotools and the ore annotations; it is outside the built hot path")
