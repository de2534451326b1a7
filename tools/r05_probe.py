"""Round-5 measurement aid (GPU): where the bf16 modes sit against the REFERENCE-RUN fixtures, and what the unmatched rows of the
"as a set" comparisons are.  Prints numbers only; the budgets asserted in tests/test_hip_bf16_anchor.py come from here.
    python tools/r05_probe.py [eval] [train] [traj]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "faster-orefsdet_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from conftest import GOLDEN, PKG, chan_err, rel_err  # noqa: E402


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def build(shots, mode="fp32"):
    from fewx.config import get_cfg
    from detectron2.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(PKG, "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", shots])
    cfg.freeze()
    torch.manual_seed(0)
    return build_model(cfg), cfg


def probe_eval():
    from oracle import ref_train as RT
    from test_oracle_golden import eval_end_to_end_state
    import tempfile
    g = golden("eval_end_to_end")
    shots = int(g["shots"])
    imgs = golden("demo_images_320")["images"]
    for mode in ("fp32", "bf16s"):
        m, _ = build(shots)
        m.eval()
        m.load_state_dict(eval_end_to_end_state(), strict=False)
        m.conv_operands = mode
        with tempfile.TemporaryDirectory() as tmp:
            m.init_model(support_file=os.path.join(tmp, "support_dir", "support_feature.pkl"), support_df=RT.eval_support_df(shots),
                         read_image=RT.eval_support_crop)
        for k in ("p3", "p4", "p5", "rcnn_8", "rcnn_4"):
            a, b = m.support_dict[k][1].float().cpu().numpy(), g["support_" + k]
            print(f"[{mode}] support {k}: rel_err {rel_err(a, b):.3e} rms-rel {np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()):.3e}")
        for i in range(2):
            for _ in range(2):
                out = m([{"image": torch.from_numpy(imgs[i]), "height": 300, "width": 300}])[0]["instances"]
            ob, os_ = out.pred_boxes.tensor.cpu().numpy(), out.scores.cpu().numpy()
            rb, rs = g[f"img{i}_boxes"], g[f"img{i}_scores"]
            d = np.abs(ob[None] - rb[:, None]).max(2)
            j = d.argmin(1)
            dd = d[np.arange(len(rb)), j]
            ds = np.abs(os_[j] - rs) / rs
            line = f"[{mode}] img{i}: ours {len(ob)} ref {len(rb)}"
            for bt, st in ((0.05, 1e-3), (0.25, 5e-3), (0.5, 1e-2), (1.0, 2e-2), (2.0, 5e-2)):
                line += f" | <= {bt}px/{st}: {((dd <= bt) & (ds <= st)).mean():.3f}"
            print(line)
            print(f"    box delta px: median {np.median(dd):.4f} p90 {np.quantile(dd, 0.9):.4f} max {dd.max():.3f}; score rel: median {np.median(ds):.2e} p90 {np.quantile(ds, 0.9):.2e}")
            bad = np.where(~((dd <= 0.05) & (ds <= 1e-3)))[0]
            for r in bad[:12]:
                print(f"    ref row {r} score {rs[r]:.5f} nearest ours {j[r]} d={dd[r]:.3f}px score {os_[j[r]]:.5f}")
        del m


def probe_train():
    import orehip
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from oracle import ref_model as R
    from oracle import ref_train as T
    for tag in ("small",):
        g = golden(f"train_iter_ref_{tag}")
        shots, hw = int(g["shots"]), tuple(int(v) for v in g["hw"])
        for mode in ("fp32", "bf16"):
            prev = orehip.set_conv_precision(mode)
            m, cfg = build(shots)
            sd = R.synth_roi_state(R.synth_state_dict(0), 0)
            sd["roi_heads.box_head.0.fc1.weight"] = sd["roi_heads.box_head.0.fc1.weight"] * 0.02
            sd["proposal_generator.centernet_head.agn_hm.bias"] = torch.full((1,), -2.0)
            m.load_state_dict(sd, strict=False)
            m.train()
            for lvl in (3, 4, 5):
                getattr(m, f"vip_p{lvl}").reweighting.drop.p = 0.0
            img, gt, sup, sbox = T.synth_train_inputs(int(g["input_seed"]), hw, n_gt=int(g["n_gt"]), shots=shots, support_hw=int(g["support_hw"]))
            inst = Instances(hw)
            inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
            item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
            over = {"boxes": torch.from_numpy(g["roi_boxes"]), "labels": torch.from_numpy(g["roi_labels"]), "gt": torch.from_numpy(g["roi_gt"])}
            losses, aux = train_forward(m, [item], return_aux=True, roi_override=over)
            n = len(g["pos_inds"])
            print(f"[{mode}/{tag}] pos_count {int(aux['pos_count'].item())} ref {n} equal {np.array_equal(aux['pos_inds'][:n].cpu().numpy(), g['pos_inds'])}")
            pb, rb = aux["proposals"].cpu(), torch.from_numpy(g["proposals"])
            d = (pb[:, None, :] - rb[None, :, :]).abs().amax(2).min(1)[0]
            print(f"[{mode}/{tag}] proposals ours {len(pb)} ref {len(rb)}: within 1e-2 px {float((d < 1e-2).float().mean()):.4f}, 0.5 px {float((d < 0.5).float().mean()):.4f}, 2 px {float((d < 2).float().mean()):.4f}")
            for k in ("loss_cls_stage0", "loss_box_reg_stage0", "loss_centernet_loc", "loss_centernet_agn_pos", "loss_centernet_agn_neg"):
                want = float(g["loss/" + k])
                print(f"[{mode}/{tag}] {k}: {float(losses[k].detach()):.6f} ref {want:.6f} rel {abs(float(losses[k].detach()) - want) / max(abs(want), 1e-3):.2e}")
            sum(losses.values()).backward()
            named = dict(m.named_parameters())
            rows = []
            for key in g:
                if not key.startswith("gs/"):
                    continue
                k = key[3:]
                f = named[k].grad.reshape(-1)
                smp = f[:: max(1, f.numel() // 1024)][:1024].cpu().numpy().astype(np.float64)
                ref = g[key].astype(np.float64)
                cos = float((smp * ref).sum() / max(np.sqrt((smp ** 2).sum() * (ref ** 2).sum()), 1e-300))
                err = float(np.abs(smp - ref).max()) / max(float(g["gn/" + k][1]), 1e-30)
                nr = float(np.sqrt((smp ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-300))
                rows.append((cos, err, nr, k))
            rows.sort()
            cs = np.array([r[0] for r in rows])
            es = np.array([r[1] for r in rows])
            print(f"[{mode}/{tag}] {len(rows)} params: cosine min {cs.min():.5f} p10 {np.quantile(cs, 0.1):.5f} median {np.median(cs):.6f}; max-err/max median {np.median(es):.2e} p90 {np.quantile(es, 0.9):.2e} max {es.max():.2e}")
            for r in rows[:6]:
                print(f"    cos {r[0]:.5f} err {r[1]:.2e} norm ratio {r[2]:.4f} {r[3]}")
            orehip.set_conv_precision(prev)
            del m


def probe_traj(steps=200):
    import orehip
    from detectron2.structures import Boxes, Instances
    from fewx.solver import build_lr_scheduler, build_optimizer
    from oracle import ref_model as R
    from oracle import ref_train as T
    shots = 4
    curves, finals = {}, {}
    for mode in ("fp32", "bf16", "fp32b"):
        prev = orehip.set_conv_precision("bf16" if mode == "bf16" else "fp32")
        from fewx.config import get_cfg
        from detectron2.modeling import build_model
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(PKG, "configs", "fsod", "finetune_vovnet.yaml"))
        cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", shots, "SOLVER.BASE_LR", float(os.environ.get("TRAJ_LR", "0.004")),
                             "SOLVER.WARMUP_ITERS", int(os.environ.get("TRAJ_WARMUP", "20"))])
        cfg.freeze()
        torch.manual_seed(0)
        m = build_model(cfg)
        if os.environ.get("TRAJ_INIT", "synth") == "ref":      # backbone: seeded He-init + non-trivial FrozenBN; heads: the reference's own initialisers
            sd = {k: v for k, v in R.synth_state_dict(0).items() if k.startswith("backbone.bottom_up.")}
        else:
            sd = R.synth_roi_state(R.synth_state_dict(0), 0)
            sd["roi_heads.box_head.0.fc1.weight"] = sd["roi_heads.box_head.0.fc1.weight"] * 0.02
            sd["proposal_generator.centernet_head.agn_hm.bias"] = torch.full((1,), -2.0)
        m.load_state_dict(sd, strict=False)
        m.train()
        for lvl in (3, 4, 5):
            getattr(m, f"vip_p{lvl}").reweighting.drop.p = 0.0
        opt = build_optimizer(cfg, m)
        sched = build_lr_scheduler(cfg, opt)
        batches = []
        for i in range(4):
            img, gt, sup, sbox = T.synth_train_inputs(30 + i, (256, 320), n_gt=5, shots=shots, support_hw=96)
            inst = Instances((256, 320))
            inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
            batches.append([{"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}])
        torch.manual_seed(1 if mode != "fp32b" else 2)          # the ROI sampler's draws
        p0 = torch.cat([p.detach().reshape(-1).clone() for p in m.parameters() if p.requires_grad])
        rec = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(steps):
            losses = m(batches[it % len(batches)])
            opt.zero_grad()
            sum(losses.values()).backward()
            opt.step()
            sched.step()
            rec.append(torch.stack([v.detach().float() for v in losses.values()]))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        curve = torch.stack(rec).cpu().numpy()
        curves[mode] = curve
        finals[mode] = (torch.cat([p.detach().reshape(-1) for p in m.parameters() if p.requires_grad]) - p0).cpu()
        tot = curve.sum(1)
        print(f"[traj {mode}] {steps} steps in {el:.1f}s; total loss every 20th: {np.round(tot[::20], 4).tolist()} last {tot[-1]:.4f}; keys {list(losses)}")
        for j, k in enumerate(losses):
            print(f"    {k}: {np.round(curve[::20, j], 4).tolist()}")
        orehip.set_conv_precision(prev)
        del m, opt
    a, b, c = curves["fp32"].sum(1), curves["bf16"].sum(1), curves["fp32b"].sum(1)
    # window means (4 batches cycle): compare the mean over each window of 20 steps
    wa, wb, wc = a.reshape(-1, 20).mean(1), b.reshape(-1, 20).mean(1), c.reshape(-1, 20).mean(1)
    print("window means fp32 :", np.round(wa, 4).tolist())
    print("window means bf16 :", np.round(wb, 4).tolist())
    print("window means fp32b:", np.round(wc, 4).tolist())
    print("rel diff bf16 vs fp32 per window:", np.round(np.abs(wb - wa) / wa, 4).tolist())
    print("rel diff fp32b(other sampler seed) vs fp32:", np.round(np.abs(wc - wa) / wa, 4).tolist())
    da, db, dc = finals["fp32"], finals["bf16"], finals["fp32b"]
    cos = lambda x, y: float((x * y).sum() / (x.norm() * y.norm()))
    print(f"parameter displacement: |fp32| {float(da.norm()):.4f} |bf16| {float(db.norm()):.4f}; |bf16-fp32|/|fp32| {float((db - da).norm() / da.norm()):.4f} cos {cos(da, db):.5f}; "
          f"|fp32b-fp32|/|fp32| {float((dc - da).norm() / da.norm()):.4f} cos {cos(da, dc):.5f}")


if __name__ == "__main__":
    what = sys.argv[1:] or ["eval", "train", "traj"]
    if "eval" in what:
        probe_eval()
    if "train" in what:
        probe_train()
    if "traj" in what:
        probe_traj()
