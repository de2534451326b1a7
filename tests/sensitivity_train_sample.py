#!/usr/bin/env python3
"""Conditioning of the oracle-based training tests' synthetic samples (GPU box only; a script, not a collected test).

The training loss has discrete decisions (assignment, top-k, sampling, ReLU); a sample on which one of them sits within fp32 rounding
of its threshold cannot carry a tight product-vs-oracle bound, because any change of summation order in a frozen convolution flips it.
For each candidate sample this prints how far the gradients move, relative to a baseline run on the round-3 kernels
(k_conv_kd / k_conv_gd / k_conv_rf off), under
  K  the shipped kernel plan (all families on)
  C  image * (1 + 1e-6 N(0,1)), 4 seeds          E  frozen backbone weights * (1 + 3e-7 N(0,1)), 6 seeds
as max over parameters of max|g - g0| / max|g0|.  A sample is usable for a tight bound when every line stays far below that bound.
Round 4 (profiles/r04_train_sample_sensitivity.txt): the samples the two tests used until then flip under C and E exactly as under K."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401  (puts the package on sys.path)
import torch
import orehip as ore
from oracle import ref_train as T
from test_hip_train import _train_model
from detectron2.structures import Boxes, Instances
from fewx.modeling.fsod.train_forward import train_forward
from fewx.solver import build_optimizer
L = ore.lib()
shots = 4


def modes(gd, kd, rf):
    L.ore_conv_set_plan_override(-14, gd, 0, 0, 0); L.ore_conv_set_plan_override(-12, kd, 0, 0, 0); L.ore_conv_set_plan_override(-10, rf, 0, 0, 0)


def sample(seed, hw, n_gt, support_hw, perm_seed):
    _, sd0, _ = _train_model(shots)
    img, gt, sup, sbox = T.synth_train_inputs(seed, hw, n_gt=n_gt, shots=shots, support_hw=support_hw)
    g = torch.Generator().manual_seed(perm_seed)
    ref = T.train_iteration(T.leaf_state(sd0), img, gt, sup, sbox, lambda n: torch.randperm(n, generator=g))
    over = {"boxes": ref["roi_boxes"], "labels": ref["roi_labels"], "gt": ref["roi_gt"]}

    def grads(m3, eps=0.0, weps=0.0, seed_=0):
        modes(*m3)
        m, _, cfg = _train_model(shots)
        if weps:
            gw = torch.Generator().manual_seed(seed_)
            with torch.no_grad():
                for k, p in m.named_parameters():
                    if not p.requires_grad and "bottom_up" in k and p.dim() == 4:
                        p.mul_((1 + weps * torch.randn(p.shape, generator=gw)).to(p.device))
        opt = build_optimizer(cfg, m)
        x = img if eps == 0 else img * (1 + eps * torch.randn(img.shape, generator=torch.Generator().manual_seed(seed_)))
        inst = Instances(hw)
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
        losses = train_forward(m, [{"image": x, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}], roi_override=over)
        opt.zero_grad()
        sum(losses.values()).backward()
        return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0}
    A = grads((0, 0, 0))

    def worst(G):
        rel = [float((G[k] - A[k]).abs().max()) / float(A[k].abs().max()) for k in A]
        return max(rel), sum(1 for v in rel if v > 1e-3)
    out = [("K", worst(grads((1, 1, 1))))]
    out += [("C%d" % i, worst(grads((0, 0, 0), eps=1e-6, seed_=20 + i))) for i in range(4)]
    out += [("E%d" % i, worst(grads((0, 0, 0), weps=3e-7, seed_=40 + i))) for i in range(6)]
    modes(1, 1, 1)
    return out


for cfgname, hw, n_gt, shw, ps in (("grad-test", (320, 384), 9, 112, 11), ("sgd-test", (256, 320), 7, 96, 3)):
    for seed in range(int(os.environ.get("SENS_SEEDS", "6"))):
        if os.environ.get("SENS_ONLY") and "%s:%d" % (cfgname, seed) not in os.environ["SENS_ONLY"].split(","):
            continue
        r = sample(seed, hw, n_gt, shw, ps)
        print("%-9s input seed %d  max %.1e, most parameters beyond 1e-3: %d | %s" % (cfgname, seed, max(v for _, (v, _) in r), max(c for _, (_, c) in r),
              "  ".join("%s %.1e/%d" % (n, v, c) for n, (v, c) in r)), flush=True)
