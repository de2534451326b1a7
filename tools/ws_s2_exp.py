import os, sys
sys.path.insert(0, "faster-orefsdet_amd")
import torch, orehip
L = orehip.lib()
dev = torch.device("cuda")
def t(x, w, Cout, reps=50):
    out = torch.empty(1, x.shape[1], x.shape[2], Cout, device=dev)
    for _ in range(5): orehip.conv2d(x, w, Cout, 3, 1, out=out)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): orehip.conv2d(x, w, Cout, 3, 1, out=out)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
for (H, W, Cin, Cout) in ((160, 160, 64, 64), (160, 160, 128, 64), (80, 80, 128, 128), (320, 320, 64, 64)):
    x = torch.randn(1, H, W, Cin, device=dev)
    w = orehip.pack_conv_weight(torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5).to(dev)
    L.ore_conv_set_plan_override(-1, -1, 0, 0, 0)
    auto = t(x, w, Cout)
    res = {}
    for mode in (4, 8, 16, 102):
        L.ore_conv_set_plan_override(-1, mode, 0, 0, 0)
        try: res[mode] = round(t(x, w, Cout), 2)
        except orehip.OreError as ex: res[mode] = str(ex)[:40]
    L.ore_conv_set_plan_override(-1, -1, 0, 0, 0)
    print(H, W, Cin, Cout, "auto %.2f" % auto, res, flush=True)
