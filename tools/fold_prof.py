import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch, bench
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev)
e = model.make_engine(max_batch=8)
imgs = torch.stack([bench.synth_image(i) for i in range(8)]).to(dev).contiguous()
for _ in range(6):
    e.eval_forward_batch(imgs, use_graph=True)
torch.cuda.synchronize()
