"""Optimizer / scheduler / data-parallel gradient exchange for the train step.

Reference behaviour (ref:fewx/solver/build.py:18-60, :91-139, :142-170; d2z:engine/train_loop.py:258-294;
d2z:engine/defaults.py:60-79 create_ddp_model; d2z:solver/lr_scheduler.py:132-164, :205-238):
    losses.backward()  [DDP: bucketed gradient all-reduce, averaged, overlapped with backward]
    clip_grad_value_(p, 1.0) per parameter;  torch.optim.SGD(momentum .9, wd 1e-4, box_predictor lr x2).step()
    WarmupMultiStepLR(steps 10000/11000, gamma .1, linear warm-up 500 iters from factor 2.5e-4)

MI355X design: every parameter that receives a gradient lives in ONE flat fp32 bucket (parameters, gradients and momentum are
three parallel buffers cut into 256-float chunks; a parameter is padded to whole chunks).  `p.data` / `p.grad` are views, so
  * backward accumulates straight into the bucket (no flatten/unflatten copies),
  * the exchange is a handful of large RCCL all-reduces over contiguous slices (xGMI is per-link bound: few, big messages),
    each launched as soon as every gradient of its slice has been produced (post-accumulate hooks), on torch's RCCL stream,
  * clip + weight decay + momentum + update for the whole model is one HIP launch (ore_sgd_step_fwd), graph-capturable.
Parameters that never receive a gradient (the reference's dead branches) stay outside the bucket: they
are neither exchanged nor decayed -- torch.optim.SGD skips `grad is None` parameters the same way.
"""
from __future__ import annotations

import bisect
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

CHUNK = 256

# gradient-ready order of the detector (backward runs heads -> correlation -> support branch -> FPN -> stage5 -> stage4):
# slices of the bucket are exchanged in this order while backward is still producing the later ones.
_READY_ORDER = ("roi_heads.box_predictor", "roi_heads.box_head", "roi_heads.", "proposal_generator.", "conv3.", "vip_p", "conv",
                "backbone.fpn_output", "backbone.fpn_lateral", "backbone.bottom_up.stage5", "backbone.bottom_up.stage4", "")


def param_groups_like_reference(cfg, model) -> List[Tuple[str, torch.nn.Parameter, float, float]]:
    """(name, parameter, lr, weight_decay) exactly as ref:fewx/solver/build.py:110-134 resolves them.

    The reference walks `model.modules()` and, for each, `module.named_parameters()` RECURSIVELY (the `recurse=False` is commented
    out), keeping the first visit of every parameter.  The root module comes first, so every parameter is resolved there with
    its full dotted name: the norm-module test never fires (GroupNorm weights get the ordinary weight decay), `"bias" in key`
    is a substring test on the full name, and `'box_predictor' in key` doubles the rate of the box predictor."""
    out, memo = [], set()
    for module in model.modules():
        for key, value in module.named_parameters():
            if not value.requires_grad or id(value) in memo:
                continue
            memo.add(id(value))
            lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
            if isinstance(module, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm, torch.nn.LayerNorm,
                                   torch.nn.modules.instancenorm._InstanceNorm, torch.nn.LocalResponseNorm)):
                wd = cfg.SOLVER.WEIGHT_DECAY_NORM
            elif "bias" in key:
                lr, wd = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR, cfg.SOLVER.WEIGHT_DECAY_BIAS
            if "box_predictor" in key:
                lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.HEAD_LR_FACTOR
            out.append((key, value, float(lr), float(wd)))
    return out


class FlatBucket:
    """Re-homes parameters into one flat buffer with parallel gradient / momentum buffers (see module docstring)."""

    def __init__(self, entries: Sequence[Tuple[str, torch.nn.Parameter, float, float]], n_slices: int = 4,
                 min_slice_bytes: int = 1 << 20):
        def order(e):
            for i, pre in enumerate(_READY_ORDER):
                if e[0].startswith(pre):
                    return i
            return len(_READY_ORDER)
        self.definition_order = [e[0] for e in entries]      # the order torch.optim would number them in
        entries = sorted(entries, key=order)          # stable: keeps definition order inside a group
        assert entries, "no parameter to optimise"
        dev = entries[0][1].device
        self.names: List[str] = []
        self.offsets: List[int] = []
        self.numels: List[int] = []
        self.base_lrs: List[float] = [e[2] for e in entries]
        self.weight_decays: List[float] = [e[3] for e in entries]
        off = 0
        lr_c, wd_c = [], []
        for name, p, lr, wd in entries:
            assert p.dtype == torch.float32 and p.device == dev, name
            n = p.numel()
            chunks = (n + CHUNK - 1) // CHUNK
            self.names.append(name); self.offsets.append(off); self.numels.append(n)
            lr_c += [lr] * chunks; wd_c += [wd] * chunks
            off += chunks * CHUNK
        self.size = off
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.momentum = torch.zeros(off, dtype=torch.float32, device=dev)
        self.chunk_lr = torch.tensor(lr_c, dtype=torch.float32, device=dev)      # the group's base rate (before the schedule factor)
        self.chunk_wd = torch.tensor(wd_c, dtype=torch.float32, device=dev)
        self.tensors: List[torch.nn.Parameter] = []
        for (name, p, _, _), o, n in zip(entries, self.offsets, self.numels):
            with torch.no_grad():
                self.params[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.params[o:o + n].view(p.shape)
            p.grad = self.grads[o:o + n].view(p.shape)
            p._ore_direct_grad = True                   # zero_grad() zeroes this buffer every step: the weight-gradient kernels may add into it
            self.tensors.append(p)
        # exchange slices: contiguous, chunk aligned, about equal bytes, never splitting a parameter
        target = max(self.size // max(n_slices, 1), min_slice_bytes // 4)
        self.slices: List[Tuple[int, int, int, int]] = []       # (begin, end, first_param, last_param+1)
        b, first = 0, 0
        for i in range(len(entries)):
            end = self.offsets[i] + ((self.numels[i] + CHUNK - 1) // CHUNK) * CHUNK
            if end - b >= target or i == len(entries) - 1:
                self.slices.append((b, end, first, i + 1))
                b, first = end, i + 1
        self.param_slice = [0] * len(entries)
        for s, (_, _, f, l) in enumerate(self.slices):
            for i in range(f, l):
                self.param_slice[i] = s

    def zero_grad(self):
        self.grads.zero_()
        for p, o, n in zip(self.tensors, self.offsets, self.numels):      # a user may have set .grad = None
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * o:
                p.grad = self.grads[o:o + n].view(p.shape)

    def nbytes_exchanged(self) -> int:
        return 4 * self.size


def get_bucket(model, cfg=None, entries=None) -> FlatBucket:
    """One bucket per model, shared by the DP wrapper and the optimizer."""
    b = getattr(model, "_ore_flat_bucket", None)
    if b is None:
        if entries is None:
            entries = param_groups_like_reference(cfg, model)
        dead = tuple(getattr(model, "gradless_parameter_prefixes", lambda: ())())
        every = [e[0] for e in entries]               # torch.optim numbers ALL of them, the gradient-less ones included
        entries = [e for e in entries if not any(e[0].startswith(d) for d in dead)]
        b = FlatBucket(entries)
        b.definition_order = every
        object.__setattr__(model, "_ore_flat_bucket", b)
    return b


class FlatDataParallel(torch.nn.Module):
    """Drop-in for DistributedDataParallel(model, broadcast_buffers=False) as d2z:engine/defaults.py:60-79 builds it.

    forward = the wrapped model.  During backward, the moment the last gradient of an exchange slice has been accumulated its
    all-reduce(SUM) is issued asynchronously; `finish()` (called by the optimizer) waits for all of them.  The 1/world_size
    average is folded into the SGD kernel (grad_scale).  Ranks start from rank 0's parameters (one broadcast of the bucket).

    transport:
      "torch"  dist.all_reduce(async_op=True) on the process group (RCCL under backend "nccl", gloo in the CPU tests);
      "rccl"   the C-ABI exchange of include/ore_hip.h (ore_rccl_* / ore_allreduce_grads): a communicator of this wrapper's own,
               created from a unique id that travels through the process group's store, reduced on a HIP stream of its own that
               is ordered against the backward kernels' stream with events -- no torch collective on the gradient path.
    force_exchange: run the exchange machinery (hooks, all-reduce, wait) for world_size 1 too; a one-rank SUM is the identity, so
    the step must equal the unwrapped step bit for bit (tests/test_hip_train.py: the RCCL rehearsal on one GPU)."""

    def __init__(self, module: torch.nn.Module, cfg=None, process_group=None, entries=None, overlap: bool = True,
                 transport: str = "torch", force_exchange: bool = False):
        super().__init__()
        assert transport in ("torch", "rccl"), transport
        self.module = module
        self.group = process_group
        self.bucket = get_bucket(module, cfg, entries)
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.exchange = self.world > 1 or bool(force_exchange)
        self.transport = transport
        self.overlap = overlap
        self._pending = [0] * len(self.bucket.slices)
        self._works: List = []
        self._issued = [False] * len(self.bucket.slices)
        self.issue_log: List[Tuple[int, bool]] = []            # (slice, issued from a backward hook?) of the step in progress
        self.last_issue_log: List[Tuple[int, bool]] = []       # ... of the last finished step
        self._in_backward_hook = False
        self.bucket.grad_scale = 1.0 / self.world
        self.bucket.finish = self.finish
        self._comm = None
        self._xstream = None
        if self.world > 1:
            dist.broadcast(self.bucket.params, src=0, group=process_group)
            inb = {id(p) for p in self.bucket.tensors}
            for t in list(module.parameters()) + list(module.buffers()):     # DDP syncs the whole module state once at construction
                if id(t) not in inb:
                    dist.broadcast(t.data, src=0, group=process_group)
        if self.exchange:
            if transport == "rccl":
                self._init_rccl()
            for i, p in enumerate(self.bucket.tensors):
                p._ore_direct_grad = False              # the exchange is issued from these hooks: the engine's AccumulateGrad has to run
                p.register_post_accumulate_grad_hook(self._make_hook(i))
        self._reset()

    # ---- the C-ABI transport -------------------------------------------------------------------------------------------------
    def _init_rccl(self):
        import ctypes as C
        import os
        import orehip
        assert self.bucket.grads.is_cuda, "the RCCL transport exchanges device memory"
        L = orehip.lib()
        # inside a torch process reuse torch's RCCL (one copy per process)
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        orehip._chk(L.ore_rccl_load(cand.encode() if os.path.exists(cand) else None), "ore_rccl_load")
        rank = dist.get_rank(self.group) if self.world > 1 else 0
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            orehip._chk(L.ore_rccl_unique_id(buf), "ore_rccl_unique_id")
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world > 1:
            obj = [ident.tolist()]
            dist.broadcast_object_list(obj, src=0, group=self.group)   # through the group's store / side channel, any backend
            ident = torch.tensor(obj[0], dtype=torch.uint8)
        comm = C.c_void_p()
        raw = (C.c_char * 128).from_buffer_copy(bytes(ident.tolist()))
        orehip._chk(L.ore_rccl_comm_create(raw, self.world, rank, C.byref(comm)), "ore_rccl_comm_create")
        self._comm = comm
        self._xstream = torch.cuda.Stream(device=self.bucket.grads.device, priority=-1)

    def close(self):
        if self._comm is not None:
            import orehip
            torch.cuda.synchronize()
            orehip._chk(orehip.lib().ore_rccl_comm_destroy(self._comm), "ore_rccl_comm_destroy")
            self._comm = None

    def _reset(self):
        for s, (_, _, f, l) in enumerate(self.bucket.slices):
            self._pending[s] = l - f
            self._issued[s] = False
        self._works = []

    def _make_hook(self, i: int) -> Callable:
        s = self.bucket.param_slice[i]

        def hook(_p):
            if self._issued[s] and self.exchange:
                # this slice was already all-reduced for the current step: a second backward() before optimizer.step() would add
                # un-exchanged local gradients to it and the ranks would diverge silently (DDP's no_sync() case; not built)
                raise RuntimeError("FlatDataParallel: backward() ran twice before optimizer.step(); gradient accumulation is not "
                                   "supported (one backward per step, as d2z:engine/train_loop.py:258-294 runs it)")
            self._pending[s] -= 1
            if self._pending[s] == 0 and self.overlap:
                self._in_backward_hook = True
                try:
                    self._issue(s)
                finally:
                    self._in_backward_hook = False
        return hook

    def _issue(self, s: int):
        if self._issued[s]:
            return
        b, e, _, _ = self.bucket.slices[s]
        self._issued[s] = True
        self.issue_log.append((s, self._in_backward_hook))
        if self.transport == "rccl":
            import ctypes as C
            import orehip
            # everything enqueued so far on the stream the gradients were accumulated on (the hook runs under it) happens before
            # the reduction; the reduction runs on the exchange stream beside the rest of backward
            self._xstream.wait_stream(torch.cuda.current_stream())
            g = self.bucket.grads
            orehip._chk(orehip.lib().ore_allreduce_grads(self._comm, C.c_void_p(g.data_ptr() + 4 * b), C.c_size_t(e - b),
                                                         C.c_void_p(self._xstream.cuda_stream)), "ore_allreduce_grads")
            return
        self._works.append(dist.all_reduce(self.bucket.grads[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Complete the exchange: issue whatever a skipped hook left behind (a parameter unused this iteration), wait for all."""
        if self.exchange:
            for s in range(len(self.bucket.slices)):
                self._issue(s)
            if self.transport == "rccl":
                torch.cuda.current_stream().wait_stream(self._xstream)     # the optimizer kernel is ordered behind every slice
            for w in self._works:
                w.wait()
        self.last_issue_log, self.issue_log = self.issue_log, []
        self._reset()

    def forward(self, *a, **k):
        return self.module(*a, **k)


class FlatSGD:
    """torch.optim.SGD + clip_grad_value_ as one HIP launch over the flat bucket.  Keeps the parts of the torch.optim interface
    the reference's trainer and checkpointer touch: param_groups (lr is read by the LR logger and written by the scheduler),
    zero_grad(), step(), state_dict()/load_state_dict()."""

    def __init__(self, bucket: FlatBucket, base_lr: float, momentum: float, clip_value: float, nesterov: bool = False,
                 apply_fn: Optional[Callable] = None):
        if nesterov:
            raise NotImplementedError("SOLVER.NESTEROV is false in every reference config")
        self.bucket = bucket
        self.base_lr = float(base_lr)
        self.momentum = float(momentum)
        self.clip_value = float(clip_value)
        self.lr_factor = 1.0                       # schedule factor; param_groups[i]["lr"] = group base x factor
        self.lr_dev: Optional[torch.Tensor] = None # the same factor on the device (a captured step reads it there: GraphedTrainStep)
        self._apply = apply_fn
        self.param_groups = []
        for p, g, wd in zip(bucket.tensors, bucket.base_lrs, bucket.weight_decays):
            self.param_groups.append({"params": [p], "lr": g, "initial_lr": g, "weight_decay": wd, "momentum": momentum})
        self.defaults = {"lr": base_lr, "momentum": momentum}

    def zero_grad(self, set_to_none: bool = False):
        self.bucket.zero_grad()

    def set_lr_factor(self, f: float):
        self.lr_factor = float(f)
        if self.lr_dev is not None:
            self.lr_dev.fill_(self.lr_factor)
        for g in self.param_groups:
            g["lr"] = g["initial_lr"] * self.lr_factor

    def step(self, closure=None):
        b = self.bucket
        fin = getattr(b, "finish", None)
        if fin is not None:
            fin()
        scale = getattr(b, "grad_scale", 1.0)
        if self._apply is not None:                # tests on CPU inject the oracle's update here; the product path is the HIP kernel
            self._apply(b, self.lr_factor, self.momentum, self.clip_value, scale)
            return
        import orehip
        from orehip import autograd as A
        orehip.sgd_step(b.params, b.grads, b.momentum, b.chunk_lr, b.chunk_wd, lr_scale=self.lr_factor, momentum=self.momentum,
                        clip_value=self.clip_value, grad_scale=scale, lr_scale_dev=self.lr_dev)
        # The kernel rewrote the parameters through raw pointers: tell torch (every cache in the package -- engines, hipGraphs,
        # packed / composed weights -- keys on the parameters' version counters) and drop the packed copies.
        self.after_step()

    def after_step(self):
        """Host-side bookkeeping of a parameter update (also called after every replay of a captured step)."""
        from orehip import autograd as A
        torch.autograd.graph.increment_version(self.bucket.tensors)
        A.weights_changed()

    def state_dict(self) -> Dict:
        return {"momentum": self.bucket.momentum.detach().cpu(), "names": list(self.bucket.names), "lr_factor": self.lr_factor}

    def load_state_dict(self, sd: Dict):
        """Accepts this class's own layout and torch.optim.SGD's ({"state": {i: {"momentum_buffer"}}, "param_groups": [...]}, what a
        reference run's checkpoint holds): there the i-th state entry belongs to the i-th parameter in param-group order, which for the
        reference is model.named_parameters() order minus the parameters without a gradient (ref:fewx/solver/build.py:110-139)."""
        if "names" in sd and "momentum" in sd:
            if list(sd["names"]) != list(self.bucket.names):
                raise ValueError("optimizer state belongs to a different parameter set")
            self.bucket.momentum.copy_(sd["momentum"])
            self.set_lr_factor(sd.get("lr_factor", 1.0))
            return
        if "state" in sd and "param_groups" in sd:
            ids = [i for g in sd["param_groups"] for i in g["params"]]
            order = getattr(self.bucket, "definition_order", None)
            if order is None or len(ids) != len(order):
                raise ValueError(f"torch.optim state with {len(ids)} parameters does not match this model's {len(self.bucket.names)} "
                                 "optimised parameters; resume from a checkpoint written by this trainer or drop the optimizer state")
            by_name = dict(zip(self.bucket.names, zip(self.bucket.offsets, self.bucket.numels)))
            for pid, name in zip(ids, order):
                st = sd["state"].get(pid, {})
                if "momentum_buffer" in st and st["momentum_buffer"] is not None and name in by_name:
                    o, n = by_name[name]
                    self.bucket.momentum[o:o + n].copy_(st["momentum_buffer"].reshape(-1))
            return
        raise ValueError("unrecognised optimizer state: expected {'momentum','names'} (this trainer) or torch.optim.SGD's "
                         "{'state','param_groups'}")


def build_optimizer(cfg, model) -> FlatSGD:
    """ref:fewx/solver/build.py:91-139.  `model` may be the FlatDataParallel wrapper or the bare detector."""
    inner = model.module if isinstance(model, FlatDataParallel) else model
    bucket = get_bucket(inner, cfg)
    clip = cfg.SOLVER.CLIP_GRADIENTS
    if clip.ENABLED and clip.CLIP_TYPE != "value":
        raise NotImplementedError("only CLIP_TYPE=value is built (the reference configs use value clipping)")
    return FlatSGD(bucket, cfg.SOLVER.BASE_LR, cfg.SOLVER.MOMENTUM, clip.CLIP_VALUE if clip.ENABLED else 0.0, cfg.SOLVER.NESTEROV)


def warmup_factor_at_iter(method: str, it: int, warmup_iters: int, warmup_factor: float) -> float:
    """d2z:solver/lr_scheduler.py:205-238."""
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        alpha = it / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    raise ValueError("Unknown warmup method: {}".format(method))


class WarmupMultiStepLR:
    """d2z:solver/lr_scheduler.py:132-164 for FlatSGD: lr_i(t) = base_i * warmup(t) * gamma^(#milestones <= t)."""

    def __init__(self, optimizer: FlatSGD, milestones: Sequence[int], gamma: float = 0.1, warmup_factor: float = 0.001,
                 warmup_iters: int = 1000, warmup_method: str = "linear", last_epoch: int = -1):
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        self.optimizer, self.milestones, self.gamma = optimizer, list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.last_epoch = last_epoch
        self.step()

    def factor(self, it: int) -> float:
        return warmup_factor_at_iter(self.warmup_method, it, self.warmup_iters, self.warmup_factor) * \
            self.gamma ** bisect.bisect_right(self.milestones, it)

    def step(self):
        self.last_epoch += 1
        self.optimizer.set_lr_factor(self.factor(self.last_epoch))

    def get_last_lr(self) -> List[float]:
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self.optimizer.set_lr_factor(self.factor(self.last_epoch))


def build_lr_scheduler(cfg, optimizer: FlatSGD):
    """ref:fewx/solver/build.py:142-170."""
    name = cfg.SOLVER.LR_SCHEDULER_NAME
    if name != "WarmupMultiStepLR":
        raise ValueError("Unknown LR scheduler: {}".format(name))
    return WarmupMultiStepLR(optimizer, cfg.SOLVER.STEPS, cfg.SOLVER.GAMMA, warmup_factor=cfg.SOLVER.WARMUP_FACTOR,
                             warmup_iters=cfg.SOLVER.WARMUP_ITERS, warmup_method=cfg.SOLVER.WARMUP_METHOD)
