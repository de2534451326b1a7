import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch, bench
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev)
e = model.engine()
img = bench.synth_image(1).to(dev)
e.eval_forward(img, use_graph=False); torch.cuda.synchronize()
c = e.buffer("counts").cpu().numpy().ravel()
print("counts", c)
s = e.buffer("pre_scores").cpu().numpy().ravel()[:c[0]]
import numpy as np
ss = np.sort(s)[::-1]
print("n_pre", c[0], "top scores", ss[:5], "score[255]", ss[255] if len(ss) > 255 else None, "score[-1]", ss[-1], "unique", len(np.unique(ss)))
o = e.buffer("out_scores").cpu().numpy().ravel()[:c[1]]
print("kept", c[1], "kept scores first/last", o[0], o[-1])
k = e.buffer("keep_idx").cpu().numpy().ravel()[:c[1]]
print("max keep rank position (in sorted order) unknown; keep_idx max", k.max())
