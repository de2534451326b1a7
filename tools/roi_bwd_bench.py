#!/usr/bin/env python3
"""ROIAlign backward at the training step's sizes, the three forms side by side: "tiled" (gather, ore_roi_align_bwd_tiled), "fixed"
(64-bit fixed-point atomics, ore_roi_align_bwd_det) and "atomic" (fp32 atomics, ore_roi_align_bwd).  Times include what each form needs
around its kernel (zeroing the maps / the accumulator planes, the finalize pass).

    python tools/roi_bwd_bench.py > gpurun_out/roi_bwd_tile.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))

import torch  # noqa: E402


def boxes_like_training(g, B, per, size):
    """per image: a quarter of the ROIs jittered around 17 ground-truth boxes (the positives), the rest spread like proposals."""
    out = []
    for _ in range(B):
        wh = torch.rand(17, 2, generator=g) * 120 + 30
        ctr = torch.rand(17, 2, generator=g) * (size - wh) + wh / 2
        gt = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
        pos = gt[torch.randint(0, 17, (per // 4,), generator=g)] + torch.randn(per // 4, 4, generator=g) * 6
        wh2 = torch.exp(torch.rand(per - per // 4, 2, generator=g) * 2.5 + 3.0)
        c2 = torch.rand(per - per // 4, 2, generator=g) * size
        out.append(torch.cat([pos, torch.cat([c2 - wh2 / 2, c2 + wh2 / 2], 1)], 0))
    return torch.cat(out, 0)


def main():
    import orehip as oh
    g = torch.Generator().manual_seed(0)
    print("# ore_version %d" % oh.lib().ore_version())
    for name, B, per, size in (("query ROIs of a bs-16 step: 16 images x 128 ROIs, 640 x 640", 16, 128, 640),
                               ("support boxes of a bs-16 step: 384 crops x 1 box, 256 x 256", 384, 1, 256),
                               ("bs-1 step: 1 image x 128 ROIs", 1, 128, 640)):
        feats = [torch.empty(B, size // s, size // s, 128, device="cuda") for s in (8, 16, 32)]
        if per == 1:
            side = torch.rand(B, 2, generator=g) * 120 + 80
            c = torch.rand(B, 2, generator=g) * (240 - side) + side / 2
            boxes = torch.cat([c - side / 2, c + side / 2], 1)
        else:
            boxes = boxes_like_training(g, B, per, size)
        boxes = boxes.cuda().contiguous()
        img = torch.arange(B, dtype=torch.int32).repeat_interleave(per).cuda()
        dout = torch.randn(B * per, 64 * 128, generator=g).cuda()
        print("## " + name)
        ref = None
        for mode in ("tiled", "fixed", "atomic"):
            oh.ROI_BWD_DETERMINISTIC, oh.ROI_BWD_MODE = mode != "atomic", "tiled" if mode == "tiled" else "fixed"
            fn = lambda: oh.roi_align_bwd(dout, feats, boxes, box_image=img)
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = out
            err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(out, ref))
            print("%-7s %9.1f us per call   max |diff| / max vs tiled %.1e" % (mode, e0.elapsed_time(e1) * 100, err))


if __name__ == "__main__":
    main()
