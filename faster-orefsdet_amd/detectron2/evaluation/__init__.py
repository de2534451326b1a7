"""inference_on_dataset: the reference's evaluation / FPS protocol (d2z:evaluation/evaluator.py:101-221): model.eval(), one
`model(inputs)` per batch, device sync after every batch, the first min(5, n-1) iterations excluded from the timing."""
import datetime
import logging
import time
from contextlib import contextmanager

import torch


class DatasetEvaluator:
    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        pass


@contextmanager
def inference_context(model):
    was = model.training
    model.eval()
    yield
    model.train(was)


def inference_on_dataset(model, data_loader, evaluator):
    logger = logging.getLogger(__name__)
    total = len(data_loader)
    if evaluator is None:
        evaluator = DatasetEvaluator()
    evaluator.reset()
    num_warmup = min(5, total - 1)
    start_time = time.perf_counter()
    total_compute_time = 0.0
    with inference_context(model), torch.no_grad():
        for idx, inputs in enumerate(data_loader):
            if idx == num_warmup:
                start_time = time.perf_counter()
                total_compute_time = 0.0
            t0 = time.perf_counter()
            outputs = model(inputs)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            total_compute_time += time.perf_counter() - t0
            evaluator.process(inputs, outputs)
    n = max(total - num_warmup, 1)
    total_time = time.perf_counter() - start_time
    logger.info("Total inference time: {} ({:.6f} s / img per device)".format(str(datetime.timedelta(seconds=int(total_time))), total_time / n))
    logger.info("Total inference pure compute time: {} ({:.6f} s / img per device)".format(
        str(datetime.timedelta(seconds=int(total_compute_time))), total_compute_time / n))
    results = evaluator.evaluate()
    inference_on_dataset.last_timing = {"images": n, "seconds_per_image": total_time / n, "compute_seconds_per_image": total_compute_time / n}
    return {} if results is None else results
