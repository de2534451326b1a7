"""One whole training iteration -- forward, the five losses, backward, value clip + SGD -- as ONE replayed hipGraph.

The reference's loop (d2z:engine/train_loop.py:258-294: `loss_dict = model(data); losses.backward(); optimizer.step()`) issues about a
thousand kernel launches per iteration; at the reference's IMS_PER_BATCH = 1 the GPU finishes them faster than the host can issue them
(7.0 ms of kernels in a 9.4 ms step).  Every stage of the training forward is sync-free and fixed-shape already (train_forward.py), the
optimizer is one kernel that can read its schedule factor from the device -- so after a few eager iterations the iteration is captured
once and replayed: per step the host stages the inputs into the graph's fixed buffers, writes the LR factor, replays.

Opt-in (`GraphedTrainStep(model, optimizer)`; the trainer's plain step stays the default).  Falls back to the eager step -- and says why
in `.error` -- when the capture fails or the inputs do not fit the captured shapes (mixed image sizes, more ground-truth boxes than the
capacity: a new capture is made for a larger capacity).  Single-process only: a data-parallel wrapper's gradient exchange is issued from
backward hooks on its own stream and stays eager."""
from __future__ import annotations

from typing import Dict, Optional

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, warmup: int = 3, gt_capacity: int = 64):
        self.model, self.opt = model, optimizer
        self.warmup = int(warmup)
        self.gt_capacity = int(gt_capacity)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.key = None
        self.static: Optional[Dict[str, torch.Tensor]] = None
        self.losses: Optional[Dict[str, torch.Tensor]] = None
        self.error: Optional[str] = None
        self.eager_steps = 0
        self.replays = 0

    # ---- the iteration, eager
    def _eager(self, batched_inputs):
        losses = self.model(batched_inputs)
        self.opt.zero_grad()
        sum(losses.values()).backward()
        self.opt.step()
        self.eager_steps += 1
        # detached: a loss that keeps its autograd graph alive keeps the parameters' AccumulateGrad nodes alive, and those remember the
        # stream they were made on -- a later capture would have to wait on that (uncaptured) stream
        return {k: v.detach() for k, v in losses.items()}

    def _stage(self, batched_inputs):
        """Uploads of this call and the fixed-shape view of them; None when the batch does not fit one graph (mixed sizes)."""
        from fewx.modeling.fsod.train_forward import stage_inputs
        n_gt = max(int(len(item["instances"])) for item in batched_inputs)
        while n_gt > self.gt_capacity:
            self.gt_capacity *= 2
        st = stage_inputs(self.model, batched_inputs, gt_capacity=self.gt_capacity)
        imgs, sups = st["imgs"], st["sups"]
        if not (all(i.shape == imgs[0].shape and i.dtype == imgs[0].dtype for i in imgs) and all(s.shape == sups[0].shape for s in sups)):
            return None, None
        xq = torch.stack(imgs) if len(imgs) > 1 else imgs[0][None]
        xs = torch.cat(sups, 0) if len(sups) > 1 else sups[0]
        cur = dict(xq=xq.contiguous(), xs=xs.contiguous(), gtp=st["gtp"], gt_n=st["gt_n"], sbx=st["sbx"].contiguous())
        key = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in cur.items())
        return cur, key

    def _capture(self, cur, key):
        from fewx.modeling.fsod.train_forward import train_core
        from orehip import autograd as A
        opt = self.opt
        self.static = {k: v.clone() for k, v in cur.items()}
        if opt.lr_dev is None:
            opt.lr_dev = torch.full((1,), float(opt.lr_factor), device=self.static["xq"].device)
        _w = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if _w is not None:                                   # (see train_forward.graphed_dense_part: the engine orders the streams itself)
            _w(False)
        saved_graph_flag = getattr(self.model, "train_graph", False)
        self.model.train_graph = False                       # no capture inside the capture
        g = torch.cuda.CUDAGraph()
        try:
            import gc
            gc.collect()                                     # (no autograd graph of an earlier iteration may survive into the capture)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                losses = train_core(self.model, self.static, static=True)
                opt.zero_grad()
                sum(losses.values()).backward()
                opt.step()
            self.graph, self.key, self.losses = g, key, {k: v.detach() for k, v in losses.items()}
        except Exception as ex:                              # noqa: BLE001 -- the capture is an optimisation, never a requirement
            self.graph, self.key, self.error = None, None, repr(ex)[:500]
            torch.cuda.synchronize()
            A.weights_changed()
        finally:
            self.model.train_graph = saved_graph_flag
        # the capture itself ran nothing: whatever the eager steps left in the caches is still right; the first replay repacks

    def __call__(self, batched_inputs):
        if self.error is not None or self.eager_steps < self.warmup:
            return self._eager(batched_inputs)
        cur, key = self._stage(batched_inputs)
        if cur is None:
            return self._eager(batched_inputs)
        if self.graph is None or key != self.key:
            self._capture(cur, key)
            if self.graph is None:
                return self._eager(batched_inputs)
        for k, v in cur.items():
            self.static[k].copy_(v, non_blocking=True)
        self.opt.lr_dev.fill_(float(self.opt.lr_factor))
        self.graph.replay()
        self.opt.after_step()                                # version counters / packed-weight caches: host-side, not in the graph
        self.replays += 1
        return self.losses
