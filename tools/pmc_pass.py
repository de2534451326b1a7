#!/usr/bin/env python3
"""Three eager (no hipGraph) eval forwards of the bench workload, for `rocprofv3 --pmc ...` passes (one counter set per run):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 tools/pmc_pass.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -- python3 tools/pmc_pass.py
tools/pmc_summary.py turns the two counter_collection.csv files into profiles/r01_pmc_traffic.json."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

if os.environ.get("ORE_XMAP"):                       # A/B aid: force the block -> tile mapping of k_conv_kw (0 = plain blockIdx)
    import orehip
    orehip.lib().ore_conv_set_plan_override(-5, int(os.environ["ORE_XMAP"]), 0, 0, 0)
model, cfg = bench.build_model(torch.device("cuda", 0))
model.conv_operands = os.environ.get("ORE_OPERANDS", "fp32")      # fp32 | bf16s: which engine the passes profile
img = bench.synth_image(0).cuda()
eng = model.engine()
for _ in range(3):
    eng.eval_forward(img, use_graph=False)
torch.cuda.synchronize()
print("pmc pass done")
