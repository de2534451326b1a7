import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch, orehip
dev = torch.device("cuda")
L = orehip.lib()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("B  H   W   Cin Cout  patch4_us   ws_us   igemm_us  TFLOPs(best)")
for (B, H, W, Cin, Cout) in ((8, 160, 160, 128, 64), (8, 80, 80, 128, 128), (16, 80, 80, 128, 128), (1, 160, 160, 128, 64), (8, 160, 160, 64, 64), (24, 60, 60, 128, 64)):
    x = torch.randn(B, H, W, Cin, device=dev)
    w = orehip.pack_conv_weight(torch.randn(Cout, Cin, 3, 3)).to(dev)
    out = torch.empty(B, H, W, Cout, device=dev)
    r = []
    for mode in (4, 102, 0):
        L.ore_conv_set_plan_override(-1, mode, 0, 0, 0)
        r.append(timeit(lambda: orehip.conv2d(x, w, Cout, 3, 1, out=out)))
    L.ore_conv_set_plan_override(-1, -1, 0, 0, 0)
    fl = 2.0 * B * H * W * Cin * Cout * 9
    print("%2d %3d %3d %4d %4d %9.1f %9.1f %9.1f   %6.1f" % (B, H, W, Cin, Cout, r[0], r[1], r[2], fl / min(r) / 1e6))
