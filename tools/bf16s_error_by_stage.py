"""Where the bf16-STORAGE engine's error against the fp32 engine comes from: rms(a - b) / rms(b) per named engine buffer, same image,
same weights (GPU).  python tools/bf16s_error_by_stage.py [H W]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "faster-orefsdet_amd")):
    sys.path.insert(0, p)
import orehip  # noqa: E402
from oracle import ref_model as R  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 640)
sd = R.synth_state_dict(0)


def engine(mode):
    prev = orehip.set_conv_precision(mode)
    try:
        e = orehip.Engine(max_batch=1, max_h=H, max_w=W)
    finally:
        orehip.set_conv_precision(prev)
    e.load_state_dict(sd)
    e.set_support(R.synth_support(0))
    e.finalize()
    return e


img = R.synth_image(0, H, W) if H != 640 or W != 640 else R.synth_image(0)
e32, ebf = engine("fp32"), engine("bf16s")
for e in (e32, ebf):
    e.eval_forward(img.cuda(), use_graph=False)
torch.cuda.synchronize()
names = ["stem1", "stem2", "stem3"]
for k in (2, 3, 4, 5):
    names += [f"cat{k}", f"stage{k}", f"gate{k}"]
for k in (3, 4, 5):
    names += [f"lat{k}", f"p{k}", f"attn{k}", f"pos{k}", f"tower{k}", f"head{k}"]
for n in names:
    try:
        a, b = ebf.buffer(n).float().cpu().numpy().astype(np.float64), e32.buffer(n).float().cpu().numpy().astype(np.float64)
    except Exception as ex:  # noqa: BLE001
        print(f"{n:8s} unavailable: {ex}")
        continue
    rms = np.sqrt(((a - b) ** 2).mean()) / max(np.sqrt((b ** 2).mean()), 1e-30)
    ch = np.sqrt(((a - b) ** 2).mean(0)) / np.maximum(np.sqrt((b ** 2).mean(0)), 1e-2 * np.sqrt((b ** 2).mean()))
    print(f"{n:8s} shape {tuple(a.shape)!s:14s} rms rel {rms:.3e}   worst channel {ch.max():.3e}  median channel {np.median(ch):.3e}")
