"""detector_postprocess (d2z:modeling/postprocessing.py:10-75): rescale boxes to the requested output size, clip, drop empty."""
import torch

from detectron2.structures import Boxes, Instances


def detector_postprocess(results: Instances, output_height: int, output_width: int):
    if isinstance(output_width, torch.Tensor):
        output_width, output_height = output_width.float(), output_height.float()
    sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((int(output_height), int(output_width)), **results.get_fields())
    boxes = results.pred_boxes if results.has("pred_boxes") else (results.proposal_boxes if results.has("proposal_boxes") else None)
    assert boxes is not None, "Predictions must contain boxes!"
    boxes = boxes.clone()
    boxes.scale(sx, sy)
    boxes.clip(results.image_size)
    if results.has("pred_boxes"):
        results.pred_boxes = boxes
    else:
        results.proposal_boxes = boxes
    return results[boxes.nonempty()]
