#!/usr/bin/env python3
"""Which kernel family moves the two oracle-based training tests (GPU box only): runs them with gd / kd / rf switched off in turn,
then measures, at the shapes those tests' frozen convolutions have, each family's error against an fp64 convolution."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import pytest, torch, torch.nn.functional as F
import orehip as ore
L = ore.lib()
TESTS = ["tests/test_hip_train.py::test_train_iteration_losses_and_gradients_vs_oracle", "tests/test_hip_train.py::test_train_step_updates_match_oracle_sgd"]
def modes(gd, kd, rf):
    L.ore_conv_set_plan_override(-14, gd, 0, 0, 0); L.ore_conv_set_plan_override(-12, kd, 0, 0, 0); L.ore_conv_set_plan_override(-10, rf, 0, 0, 0)
for name, m in (("all on", (1, 1, 1)), ("gd off", (0, 1, 1)), ("kd off", (1, 0, 1)), ("gd+kd off", (0, 0, 1)), ("gd+kd+rf off", (0, 0, 0))):
    modes(*m)
    rc = pytest.main(["-q", "-x", "--no-header", "-p", "no:cacheprovider"] + TESTS)
    print("=== %-14s pytest exit %d" % (name, int(rc)), flush=True)
g = torch.Generator().manual_seed(0)
# 320x384 input: stem_3 160x192 -> 80x96 (64->128 k3 s2), stage-2 concat 80x96 (320->112), stage-3 concat 40x48 (352->256), stage-4 concat 20x24 (
for (H, W, Cin, Cout, k, s) in ((160, 192, 64, 128, 3, 2), (80, 96, 320, 112, 1, 1), (40, 48, 352, 256, 1, 1), (20, 24, 544, 384, 1, 1), (10, 12, 768, 512, 1, 1), (20, 24, 96, 96, 3, 1)):
    x = torch.relu(torch.randn(1, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, s, k // 2)
    xn, wp = x.permute(0, 2, 3, 1).contiguous().cuda(), ore.pack_conv_weight(w).cuda()
    ys = {}
    for name, m in (("all on", (1, 1, 1)), ("gd+kd+rf off", (0, 0, 0))):
        modes(*m)
        y = ore.conv2d(xn, wp, Cout, k, s).permute(0, 3, 1, 2).cpu().double()
        e = (y - ref).abs(); ys[name] = y
        print("%dx%d %d->%d k%d s%d  %-13s max|err|/max|ref| %.3e  rms err/rms ref %.3e" % (H, W, Cin, Cout, k, s, name, e.max() / ref.abs().max(), (e ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()))
    print("      new vs old max diff / max|ref| %.3e" % float((ys["all on"] - ys["gd+kd+rf off"]).abs().max() / ref.abs().max()), flush=True)
modes(1, 1, 1)
