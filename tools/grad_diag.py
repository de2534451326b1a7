#!/usr/bin/env python3
"""Diagnostic: per-parameter gradient error of the product's training iteration against the reference-run fixture and the CPU oracle
(which matches the fixture to 5e-6), plus where in the head the two diverge.  GPU box only; not a test."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(tag="small"):
    from test_hip_train import _train_model
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from oracle import ref_train as T
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"train_iter_ref_{tag}.npz")))
    shots, hw = int(g["shots"]), tuple(int(v) for v in g["hw"])
    m, sd, cfg = _train_model(shots)
    img, gt, sup, sbox = T.synth_train_inputs(int(g["input_seed"]), hw, n_gt=int(g["n_gt"]), shots=shots, support_hw=int(g["support_hw"]))
    inst = Instances(hw)
    inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
    item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
    over = {"boxes": torch.from_numpy(g["roi_boxes"]), "labels": torch.from_numpy(g["roi_labels"]), "gt": torch.from_numpy(g["roi_gt"])}
    losses, aux = train_forward(m, [item], return_aux=True, roi_override=over)
    for h in aux["heads"]:
        h.retain_grad()
    sum(losses.values()).backward()
    leaf = T.leaf_state(sd)
    gen = torch.Generator().manual_seed(int(g["randperm_seed"]))
    ref = T.train_iteration(leaf, img, gt, sup, sbox, lambda n: torch.randperm(n, generator=gen), roi_override=over)
    for r in ref["reg"] + ref["hm"]:
        r.retain_grad()
    sum(ref["losses"].values()).backward()
    named = dict(m.named_parameters())
    rows = []
    for key in g:
        if key.startswith("gs/"):
            k = key[3:]
            f = named[k].grad.reshape(-1)
            smp = f[:: max(1, f.numel() // 1024)][:1024].cpu().numpy()
            e_fix = float(np.abs(smp - g[key]).max()) / max(float(g["gn/" + k][1]), 1e-30)
            e_orc = float((named[k].grad.cpu() - leaf[k].grad).abs().max()) / max(float(leaf[k].grad.abs().max()), 1e-30)
            rows.append((e_fix, e_orc, float(g["gc/" + k]), k))
    rows.sort()
    print("err_vs_fixture  err_vs_oracle  ref_fp32_vs_fp64  name")
    for r in rows[-25:]:
        print("%.3e  %.3e  %.3e  %s" % r)
    for l in range(3):
        hd = aux["heads"][l].detach().cpu()[0]
        reg, hm = ref["reg"][l][0].permute(1, 2, 0).detach(), ref["hm"][l][0, 0].detach()
        print(f"level {l}: reg relerr {float((hd[..., :4] - reg).abs().max() / reg.abs().max()):.2e}  hm abs err {float((hd[..., 4] - hm).abs().max()):.2e}")
        gh = aux["heads"][l].grad.cpu()[0]
        gr, gm = ref["reg"][l].grad[0].permute(1, 2, 0), ref["hm"][l].grad[0, 0]
        d = (gh[..., 4] - gm).abs()
        print(f"   d(loss)/d(hm logit): max ref {float(gm.abs().max()):.3e}  max diff {float(d.max()):.3e} at {int(d.argmax())}  "
              f"ref there {float(gm.reshape(-1)[d.argmax()]):.3e} got {float(gh[..., 4].reshape(-1)[d.argmax()]):.3e} logit {float(hm.reshape(-1)[d.argmax()]):.5f}")
        d = (gh[..., :4] - gr).abs()
        print(f"   d(loss)/d(reg): max ref {float(gr.abs().max()):.3e}  max diff {float(d.max()):.3e}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "small")
