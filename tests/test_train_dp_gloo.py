"""Row a13 / 8e on CPU: two gloo ranks run the flat-bucket gradient exchange + clip + SGD and must land on the parameters a
single process gets from torch.optim.SGD + clip_grad_value_ on the rank-averaged gradients (what DDP + the reference's
optimizer wrapper compute, ref:fewx/solver/build.py:18-60,110-139, d2z:engine/train_loop.py:258-294)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(nn.Module):
    """Shapes chosen so parameters straddle chunk boundaries; `dead` never reaches the loss."""

    def __init__(self):
        super().__init__()
        self.backbone = nn.Sequential(nn.Linear(37, 300), nn.ReLU(), nn.Linear(300, 129))
        self.norm = nn.GroupNorm(3, 129)
        self.box_predictor = nn.Linear(129, 5)
        self.dead = nn.Linear(129, 7)
        self.frozen = nn.Linear(3, 3)
        for p in self.frozen.parameters():
            p.requires_grad_(False)

    def gradless_parameter_prefixes(self):
        return ("dead.",)

    def forward(self, x):
        return self.box_predictor(self.norm(self.backbone(x))).square().sum() * 40.0     # large enough for the clip to bite


def _cfg():
    sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
    from fewx.config import get_cfg
    cfg = get_cfg()
    cfg.SOLVER.BASE_LR, cfg.SOLVER.WARMUP_ITERS = 0.01, 2
    cfg.SOLVER.STEPS = (3, 4)
    cfg.SOLVER.WARMUP_FACTOR, cfg.SOLVER.HEAD_LR_FACTOR = 0.00025, 2.0
    cfg.SOLVER.CLIP_GRADIENTS.ENABLED, cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE, cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE = True, "value", 1.0
    return cfg


def _oracle_apply(b, lr_factor, momentum, clip, scale):
    sys.path.insert(0, ROOT)
    from oracle import ref_model as R
    R.sgd_step_flat(b.params, b.grads, b.momentum, b.chunk_lr, b.chunk_wd, lr_factor, momentum, clip, scale)


def _data(rank, step):
    g = torch.Generator().manual_seed(1000 + 10 * step + rank)
    return torch.randn(6, 37, generator=g)


def _worker(rank, world, port, q, steps, overlap):
    sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
    import torch.distributed as dist
    from fewx.solver import FlatDataParallel, build_lr_scheduler, build_optimizer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(7 + rank)                  # ranks start different: the wrapper must broadcast rank 0's parameters
        cfg = _cfg()
        model = FlatDataParallel(_Toy(), cfg, overlap=overlap)
        opt = build_optimizer(cfg, model)
        opt._apply = _oracle_apply                   # CPU test: the update comes from the oracle; the product path is the HIP kernel
        sched = build_lr_scheduler(cfg, opt)
        lrs = []
        for s in range(steps):
            loss = model(_data(rank, s))
            opt.zero_grad()
            loss.backward()
            opt.step()
            lrs.append(opt.param_groups[0]["lr"])
            sched.step()
        sd = {k: v.detach().numpy().copy() for k, v in model.module.state_dict().items()}   # numpy: no fd passing
        q.put((rank, sd, lrs, len(model.bucket.slices), model.bucket.names))
    finally:
        dist.destroy_process_group()


def _single_process_reference(world, steps):
    sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
    from fewx.solver import param_groups_like_reference, warmup_factor_at_iter
    torch.manual_seed(7)
    cfg = _cfg()
    m = _Toy()
    groups = [{"params": [p], "lr": lr, "weight_decay": wd} for _, p, lr, wd in param_groups_like_reference(cfg, m)]
    opt = torch.optim.SGD(groups, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)
    base = [g["lr"] for g in opt.param_groups]
    for s in range(steps):
        f = warmup_factor_at_iter("linear", s, cfg.SOLVER.WARMUP_ITERS, cfg.SOLVER.WARMUP_FACTOR) * \
            cfg.SOLVER.GAMMA ** sum(1 for ms in cfg.SOLVER.STEPS if ms <= s)
        for g, b in zip(opt.param_groups, base):
            g["lr"] = b * f
        opt.zero_grad(set_to_none=True)
        loss = sum(m(_data(r, s)) for r in range(world)) / world          # DDP: mean of the per-rank gradients
        loss.backward()
        for g in opt.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    torch.nn.utils.clip_grad_value_(p, cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE)
        opt.step()
    return m.state_dict()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_flat_bucket_sgd_matches_ddp_semantics(overlap):
    world, steps, port = 2, 5, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, steps, overlap)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (_, sd0, lrs0, nsl, names), (_, sd1, lrs1, _, _) = res
    ref = _single_process_reference(world, steps)
    sd0 = {k: torch.from_numpy(v) for k, v in sd0.items()}
    sd1 = {k: torch.from_numpy(v) for k, v in sd1.items()}
    assert not any(n.startswith("dead.") or n.startswith("frozen.") for n in names)       # never exchanged, never decayed
    assert any(n.startswith("box_predictor.") for n in names)
    for k in ref:
        assert torch.equal(sd0[k], sd1[k]), k                                             # ranks stay bit-identical
        assert torch.allclose(sd0[k], ref[k], rtol=2e-5, atol=2e-6), (k, (sd0[k] - ref[k]).abs().max())
    torch.manual_seed(7)
    init = _Toy().state_dict()
    assert torch.equal(sd0["dead.weight"], init["dead.weight"]) and torch.equal(sd0["frozen.weight"], init["frozen.weight"])
    assert not torch.equal(sd0["box_predictor.weight"], init["box_predictor.weight"])
    assert lrs0 == lrs1 and abs(lrs0[0] - 0.01 * 0.00025) < 1e-12 and abs(lrs0[2] - 0.01) < 1e-12 and abs(lrs0[4] - 0.01 * 0.01) < 1e-12


def test_reference_param_groups_and_schedule():
    """The reference's quirks: box_predictor lr x2, norm weight decay never applied, warm-up + steps."""
    sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
    from fewx.solver import param_groups_like_reference, FlatBucket, FlatSGD, WarmupMultiStepLR
    cfg = _cfg()
    cfg.SOLVER.BASE_LR = 0.001
    m = _Toy()
    g = {n: (lr, wd) for n, _, lr, wd in param_groups_like_reference(cfg, m)}
    assert g["box_predictor.weight"] == (0.002, 0.0001) and g["box_predictor.bias"] == (0.002, 0.0001)
    assert g["norm.weight"] == (0.001, 0.0001)                    # WEIGHT_DECAY_NORM=0 is never reached
    assert "frozen.weight" not in g and "dead.weight" in g
    b = FlatBucket([(n, p, lr, wd) for n, p, lr, wd in param_groups_like_reference(cfg, m)], n_slices=3, min_slice_bytes=1024)
    assert b.size % 256 == 0 and all(o % 256 == 0 for o in b.offsets)
    assert b.slices[0][0] == 0 and b.slices[-1][1] == b.size and all(b.slices[i][1] == b.slices[i + 1][0] for i in range(len(b.slices) - 1))
    for p, o, n in zip(b.tensors, b.offsets, b.numels):           # parameters and gradients are views into the bucket
        assert p.data_ptr() == b.params.data_ptr() + 4 * o and p.grad.data_ptr() == b.grads.data_ptr() + 4 * o
    opt = FlatSGD(b, 0.001, 0.9, 1.0)
    s = WarmupMultiStepLR(opt, (10000, 11000), 0.1, warmup_factor=0.00025, warmup_iters=500)
    assert abs(s.factor(0) - 0.00025) < 1e-15 and abs(s.factor(250) - (0.00025 * 0.5 + 0.5)) < 1e-15
    assert s.factor(500) == 1.0 and abs(s.factor(10000) - 0.1) < 1e-15 and abs(s.factor(11500) - 0.01) < 1e-12
