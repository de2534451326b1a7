"""The slice of d2z:engine the reference's training script touches (ref:fsod_train_net.py:12-17,36-73,96-118), on the
flat-bucket data-parallel step of fewx.solver:

    SimpleTrainer.run_step   d2z:engine/train_loop.py:258-294   data -> model(data) -> sum(losses) -> zero_grad -> backward
                             [gradient exchange overlapped] -> optimizer.step() (one fused clip+SGD launch)
    create_ddp_model         d2z:engine/defaults.py:60-79       FlatDataParallel instead of DistributedDataParallel
    DefaultTrainer           d2z:engine/defaults.py:300-560     build_model / build_optimizer / build_lr_scheduler / train()
    launch                   d2z:engine/launch.py:24-83          one process per GPU, RCCL ("nccl") process group, 127.0.0.1
Dataset registration, evaluators, hooks for periodic checkpoint/eval belong to the data/evaluation side.
"""
import argparse
import logging
import os
import time

import torch
import torch.distributed as dist

from detectron2.utils import comm
from detectron2.utils.events import EventStorage

__all__ = ["SimpleTrainer", "DefaultTrainer", "create_ddp_model", "default_argument_parser", "default_setup", "launch"]


def create_ddp_model(model, cfg=None, **kwargs):
    if comm.get_world_size() == 1:
        return model
    from fewx.solver import FlatDataParallel
    return FlatDataParallel(model, cfg, **kwargs)


class SimpleTrainer:
    """d2z:engine/train_loop.py:213-341.  The reference's `_write_metrics` reads every loss on the host in every step
    (`v.detach().cpu().item()`), i.e. one device synchronisation per iteration, and raises FloatingPointError when their sum is not
    finite.  Here the step stays free of host syncs: the loss vector of step i travels to pinned host memory behind step i's kernels
    and is looked at once step i+1's forward has been enqueued (`metrics_lag` = 1; 0 = the reference's blocking form).  The guard is
    the reference's: the same exception type and message, naming the iteration whose loss was not finite, raised on every rank for
    its own losses (the reference checks the mean over ranks on the main process)."""
    metrics_lag = 1

    def __init__(self, model, data_loader, optimizer):
        model.train()
        self.model, self.data_loader, self.optimizer = model, data_loader, optimizer
        self._data_loader_iter = iter(data_loader)
        self.iter = 0
        self.storage = None
        self.last_losses = None
        self._pending = []                                       # [(iteration, names, pinned host vector, event, data_time)]

    def run_step(self):
        assert self.model.training, "[SimpleTrainer] model was changed to eval mode!"
        start = time.perf_counter()
        data = next(self._data_loader_iter)
        data_time = time.perf_counter() - start
        if getattr(self, "graph_step", False):
            # opt-in (trainer.graph_step = True; single process, the HIP detector): the whole iteration -- forward, losses, backward, clip
            # + SGD -- captured once and replayed as one hipGraph (fewx.solver.GraphedTrainStep; eager while it warms up or cannot capture)
            if self.__dict__.get("_graphed") is None:
                from fewx.solver import GraphedTrainStep
                self._graphed = GraphedTrainStep(self.model, self.optimizer)
            loss_dict = self._graphed(data)
            self.flush_metrics(keep=self.metrics_lag - 1)
            self.last_losses = {k: v.detach().clone() for k, v in loss_dict.items()}   # (the graph's loss buffers are rewritten by the next replay)
            self.last_data_time = data_time
            self._write_metrics(self.last_losses, data_time)
            return
        loss_dict = self.model(data)
        self.flush_metrics(keep=self.metrics_lag - 1)            # step i-1's losses, behind step i's forward launches
        if isinstance(loss_dict, torch.Tensor):
            losses, loss_dict = loss_dict, {"total_loss": loss_dict}
        else:
            losses = sum(loss_dict.values())
        self.optimizer.zero_grad()
        losses.backward()
        self.last_losses = {k: v.detach() for k, v in loss_dict.items()}
        self.last_data_time = data_time
        self._write_metrics(self.last_losses, data_time)
        self.optimizer.step()

    def _write_metrics(self, loss_dict, data_time, prefix=""):
        names = list(loss_dict)
        vec = torch.stack([loss_dict[k].reshape(()).float() for k in names])
        if vec.is_cuda:
            host = torch.empty(len(names), dtype=torch.float32, pin_memory=True)
            host.copy_(vec, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = vec.clone(), None
        self._pending.append((self.iter, names, host, ev, data_time, prefix))
        if self.metrics_lag <= 0:
            self.flush_metrics()

    def flush_metrics(self, keep=0):
        """Look at the stashed loss vectors, oldest first, until `keep` are left: scalars into the current EventStorage, and the
        reference's NaN / Inf guard (d2z:engine/train_loop.py:336-341)."""
        while len(self._pending) > max(keep, 0):
            it, names, host, ev, data_time, prefix = self._pending.pop(0)
            if ev is not None:
                ev.synchronize()
            metrics = {k: float(v) for k, v in zip(names, host.tolist())}
            total = sum(metrics.values())
            if not (total == total and abs(total) != float("inf")):
                self._pending.clear()
                raise FloatingPointError("Loss became infinite or NaN at iteration={}!\nloss_dict = {}".format(it, metrics))
            from detectron2.utils.events import get_event_storage, has_event_storage
            if has_event_storage() and comm.is_main_process():
                st = get_event_storage()
                st.put_scalar("data_time", data_time, cur_iter=it)
                st.put_scalar("{}total_loss".format(prefix), total, cur_iter=it)
                if len(metrics) > 1:
                    st.put_scalars(cur_iter=it, **metrics)


class DefaultTrainer(SimpleTrainer):
    def __init__(self, cfg):
        logging.getLogger("detectron2")
        model = self.build_model(cfg)
        model = create_ddp_model(model, cfg)
        optimizer = self.build_optimizer(cfg, model)
        data_loader = self.build_train_loader(cfg)
        super().__init__(model, data_loader, optimizer)
        self.scheduler = self.build_lr_scheduler(cfg, optimizer)
        self.cfg = cfg
        self.start_iter, self.max_iter = 0, cfg.SOLVER.MAX_ITER
        from detectron2.checkpoint import DetectionCheckpointer
        self.checkpointer = DetectionCheckpointer(model.module if hasattr(model, "module") else model, cfg.OUTPUT_DIR,
                                                  optimizer=optimizer, scheduler=self.scheduler)

    @classmethod
    def build_model(cls, cfg):
        from detectron2.modeling import build_model
        return build_model(cfg)

    @classmethod
    def build_optimizer(cls, cfg, model):
        from fewx.solver import build_optimizer
        return build_optimizer(cfg, model)

    @classmethod
    def build_lr_scheduler(cls, cfg, optimizer):
        from fewx.solver import build_lr_scheduler
        return build_lr_scheduler(cfg, optimizer)

    @classmethod
    def build_train_loader(cls, cfg):
        """d2z:engine/defaults.py:523-533: the plain detection loader (records as registered, plain DatasetMapper); the reference's
        script overrides it with fewx.data.build + DatasetMapperWithSupport (ref:fsod_train_net.py:38-47)."""
        from detectron2.data import build_detection_train_loader
        return build_detection_train_loader(cfg)

    @classmethod
    def build_test_loader(cls, cfg, dataset_name):
        """d2z:engine/defaults.py:535-544."""
        from detectron2.data import build_detection_test_loader
        return build_detection_test_loader(cfg, dataset_name)

    @classmethod
    def build_evaluator(cls, cfg, dataset_name, output_folder=None):
        raise NotImplementedError("COCO evaluation is outside the hot path (SURVEY 8f)")

    @classmethod
    def test(cls, cfg, model, evaluators=None):
        """d2z:engine/defaults.py:570-621 (`Trainer.test(cfg, model)` of ref:fsod_train_net.py:100): every dataset of cfg.DATASETS.TEST
        through `build_test_loader` + `inference_on_dataset`; a `build_evaluator` that raises NotImplementedError is logged and leaves
        an empty result for that dataset (no inference pass), exactly as there; a single dataset's result is returned unwrapped."""
        from collections import OrderedDict
        from detectron2.evaluation import DatasetEvaluator, inference_on_dataset
        logger = logging.getLogger(__name__)
        if isinstance(evaluators, DatasetEvaluator):
            evaluators = [evaluators]
        if evaluators is not None:
            assert len(cfg.DATASETS.TEST) == len(evaluators), "{} != {}".format(len(cfg.DATASETS.TEST), len(evaluators))
        results = OrderedDict()
        for idx, dataset_name in enumerate(cfg.DATASETS.TEST):
            data_loader = cls.build_test_loader(cfg, dataset_name)
            if evaluators is not None:
                evaluator = evaluators[idx]
            else:
                try:
                    evaluator = cls.build_evaluator(cfg, dataset_name)
                except NotImplementedError:
                    logger.warning("No evaluator found. Use `DefaultTrainer.test(evaluators=)`, or implement its `build_evaluator` method.")
                    results[dataset_name] = {}
                    continue
            results_i = inference_on_dataset(model, data_loader, evaluator)
            results[dataset_name] = results_i
            if comm.is_main_process():
                assert isinstance(results_i, dict), "Evaluator must return a dict on the main process. Got {} instead.".format(results_i)
        if len(results) == 1:
            results = list(results.values())[0]
        return results

    def resume_or_load(self, resume=True):
        self.checkpointer.resume_or_load(self.cfg.MODEL.WEIGHTS, resume=resume)
        if resume and self.checkpointer.has_checkpoint():
            self.start_iter = self.scheduler.last_epoch

    def train(self):
        with EventStorage(self.start_iter) as self.storage:
            for self.iter in range(self.start_iter, self.max_iter):
                self.storage.iter = self.iter
                self.run_step()
                self.scheduler.step()
                period = self.cfg.SOLVER.CHECKPOINT_PERIOD
                if period > 0 and (self.iter + 1) % period == 0:
                    self.flush_metrics()                         # never checkpoint behind an unexamined loss
                    if comm.is_main_process():
                        self.checkpointer.save("model_{:07d}".format(self.iter), iteration=self.iter)
            self.flush_metrics()
            if comm.is_main_process():
                self.checkpointer.save("model_final", iteration=self.max_iter - 1)


def default_argument_parser(epilog=None):
    """d2z:engine/defaults.py:82-127 (same flags)."""
    p = argparse.ArgumentParser(epilog=epilog, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--config-file", default="", metavar="FILE")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--eval-only", action="store_true")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    port = 2 ** 15 + 2 ** 14 + hash(os.getuid() if hasattr(os, "getuid") else 1) % 2 ** 14
    p.add_argument("--dist-url", default="tcp://127.0.0.1:{}".format(port))
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


def default_setup(cfg, args):
    """d2z:engine/defaults.py:130-178: output dir, logger, seed, config dump."""
    from detectron2.utils.env import seed_all_rng
    from detectron2.utils.logger import setup_logger
    out = cfg.OUTPUT_DIR
    if comm.is_main_process() and out:
        os.makedirs(out, exist_ok=True)
    setup_logger(out, distributed_rank=comm.get_rank())
    seed_all_rng(None if cfg.SEED < 0 else cfg.SEED + comm.get_rank())
    if comm.is_main_process() and out:
        with open(os.path.join(out, "config.yaml"), "w") as f:
            f.write(cfg.dump())


def _worker(local_rank, main_func, world, gpus_per_machine, machine_rank, dist_url, args):
    assert torch.cuda.is_available(), "training runs on the MI355X only"
    rank = machine_rank * gpus_per_machine + local_rank
    os.environ.setdefault("LOCAL_RANK", str(local_rank))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", init_method=dist_url, world_size=world, rank=rank)      # "nccl" is RCCL on ROCm
    try:
        comm.synchronize()
        main_func(*args)
    finally:
        dist.destroy_process_group()


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=()):
    """One process per GPU (d2z:engine/launch.py:24-83)."""
    world = num_machines * num_gpus_per_machine
    if world <= 1:
        return main_func(*args)
    import torch.multiprocessing as mp
    if dist_url in (None, "auto"):
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        dist_url = "tcp://127.0.0.1:{}".format(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_worker, nprocs=num_gpus_per_machine, args=(main_func, world, num_gpus_per_machine, machine_rank, dist_url, args), daemon=False)
