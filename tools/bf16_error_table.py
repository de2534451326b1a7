"""Error table of the ORE_CONV_BF16 engine against the oracle in bf16-operand mode and in fp32 (max-norm and RMS, relative to the
reference tensor's max / RMS).  Run on the MI355X: python tools/bf16_error_table.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import orehip  # noqa: E402
from oracle import ref_model as R  # noqa: E402  (checker only)


def errs(got, ref):
    d = got - ref
    return float(np.abs(d).max() / np.abs(ref).max()), float(np.sqrt((d * d).mean()) / np.sqrt((ref * ref).mean()))


def main():
    sd = R.synth_state_dict(0)
    img = R.synth_image(0)
    with R.operand_precision("bf16"):
        rb = R.eval_dense(img, sd, R.synth_support(0))
    rf = R.eval_dense(img, sd, R.synth_support(0))
    rows = []
    for mode in ("fp32", "bf16"):
        prev = orehip.set_conv_precision(mode)
        e = orehip.Engine(max_batch=1, max_h=640, max_w=640)
        orehip.set_conv_precision(prev)
        e.load_state_dict(sd)
        e.set_support(R.synth_support(0))
        e.finalize()
        e.eval_forward(img.cuda(), use_graph=False)
        torch.cuda.synchronize()
        for l, k in enumerate(("p3", "p4", "p5")):
            s = 640 >> (l + 3)
            got = {k: e.buffer(k, (1, s, s)).cpu().numpy(), f"pos{l + 3}": e.buffer(f"pos{l + 3}", (1, s, s)).cpu().numpy()}
            hd = e.buffer(f"head{l + 3}", (1, s, s)).cpu().numpy()
            got[f"reg{l + 3}"], got[f"hm{l + 3}"] = hd[:, :4], hd[:, 4:5]
            refs = lambda r: {k: r["features"][k].numpy(), f"pos{l + 3}": r["pos_features"][l].numpy(), f"reg{l + 3}": r["reg"][l].numpy(),  # noqa: E731
                              f"hm{l + 3}": r["hm"][l].numpy()}
            for name, g in got.items():
                rows.append((mode, name) + errs(g, refs(rb)[name]) + errs(g, refs(rf)[name]))
        e.close()
    print("%-5s %-6s | vs bf16-mode oracle: max      rms | vs fp32 oracle: max      rms" % ("mode", "tensor"))
    for r in rows:
        print("%-5s %-6s | %24.2e %8.2e | %19.2e %8.2e" % r)


if __name__ == "__main__":
    main()
