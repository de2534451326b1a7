#!/usr/bin/env python3
"""The headline protocol (model([{image,height,width}]) + synchronize per image) for N images: the command to put behind
`rocprofv3 --kernel-trace --stats` (tools/trace_summary.py then prints one image's kernel timeline)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if os.environ.get("ORE_XMAP"):                       # A/B aid: force the block -> tile mapping of the conv kernels (0 = plain blockIdx)
    sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
    import orehip
    orehip.lib().ore_conv_set_plan_override(-5, int(os.environ["ORE_XMAP"]), 0, 0, 0)
model, cfg = bench.build_model(torch.device("cuda", 0))
model.conv_operands = os.environ.get("ORE_OPERANDS", "fp32")      # fp32 | bf16 | bf16s (bench.py --conv-operands)
imgs = [bench.synth_image(i).cuda() for i in range(4)]
for i in range(n):
    model([{"image": imgs[i % 4], "height": 640, "width": 640}])
    torch.cuda.synchronize()
print("done")
