import sys, os
sys.path.insert(0, "faster-orefsdet_amd")
import torch, torch.nn.functional as F
import orehip as ore
g = torch.Generator().manual_seed(5)
L = ore.lib()
L.ore_conv_set_plan_override(-12, 2, 0, 0, 0)
for bm, bn in ((16, 16), (32, 32), (32, 64)):
    L.ore_conv_set_plan_override(-13, bm, bn, 4, 4)
    w3 = torch.randn(80, 96, 3, 3, generator=g) / (96 * 9) ** 0.5
    x3 = torch.randn(1, 96, 9, 8, generator=g)
    sc, sh = torch.rand(80, generator=g) + 0.5, torch.randn(80, generator=g) * 0.1
    ref3 = F.relu(F.conv2d(x3, w3, None, 1, 1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    y3, cs = ore.conv2d(x3.permute(0, 2, 3, 1).contiguous().cuda(), ore.pack_conv_weight(w3).cuda(), 80, 3, 1, scale=sc.cuda(), shift=sh.cuda(), relu_cout=80, want_colsum=True)
    rows = ref3[0].permute(1, 2, 0).reshape(72, 80)
    print(bm, bn, "cs shape", tuple(cs.shape), "y err", float((y3.cpu().reshape(72, 80) - rows).abs().max()))
    for t in range(cs.shape[0]):
        exp = rows[t * bm:(t + 1) * bm].sum(0)
        got = cs[t, :80].cpu()
        print("  tile", t, "max err", float((got - exp).abs().max()), "got[:4]", got[:4].tolist(), "exp[:4]", exp[:4].tolist(), " bad cols:", (got - exp).abs().gt(1e-3).nonzero().flatten().tolist()[:20])
