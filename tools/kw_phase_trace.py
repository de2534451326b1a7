#!/usr/bin/env python3
"""Phase timeline of k_conv_kw (make -C faster-orefsdet_amd/csrc trace -> lib/libore_hip_trace.so): thread 0 of every block stamps
s_memtime at the phase boundaries and s_memrealtime (100 MHz) at entry / exit.
usage: kw_phase_trace.py H W Cin Cout k   -> per-phase shader clocks (median / max over blocks) + the launch's wall-clock picture."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orehip  # noqa: E402

orehip.LIB_PATH = os.path.join(ROOT, "faster-orefsdet_amd", "lib", "libore_hip_trace.so")
dev = torch.device("cuda")
L = orehip.lib()


RF = False           # True: the layer runs on k_conv_rf (stamps of csrc/ore_conv_rf.hip)
KD = False           # True: ... on k_conv_kd (csrc/ore_conv_kd.hip)


def trace(H, W, Cin, Cout, k, reps=5, stride=1):
    x = torch.randn(1, H, W, Cin, device=dev)
    w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5).to(dev)
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(1, Ho, Wo, Cout, device=dev)
    other = torch.randn(64 << 20, device=dev)                       # 256 MB: evicts L2 / MALL between the timed launches

    def run():
        orehip.conv2d(x, w, Cout, k, stride, scale=sc, shift=sh, relu_cout=Cout, out=out)
    for _ in range(3):
        run()
    # plain timing, back to back, as the engine's graph replays them
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        run()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 50 * 1e3
    nb = 8192
    res = []
    for cold in (False, True):
        rows = []
        for _ in range(reps):
            buf = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
            if cold:
                other.mul_(1.0001)
            run()                                                   # the launch in front (warm: the same layer; cold: the sweep)
            L.ore_debug_set_trace_kw(C.c_void_p(buf.data_ptr()))
            L.ore_debug_set_trace_rf(C.c_void_p(buf.data_ptr()))
            L.ore_debug_set_trace_kd(C.c_void_p(buf.data_ptr()))
            torch.cuda.synchronize()
            run()
            torch.cuda.synchronize()
            L.ore_debug_set_trace_kw(C.c_void_p(0))
            L.ore_debug_set_trace_rf(C.c_void_p(0))
            L.ore_debug_set_trace_kd(C.c_void_p(0))
            t = buf.cpu().numpy().reshape(nb, 16)
            t = t[t[:, 0] != 0]
            rows.append(t)
        res.append(rows[-1])
    print("== %dx%d Cin %d Cout %d k %d: %.2f us per launch back to back (eager, same stream)" % (H, W, Cin, Cout, k, us))
    names = ["prologue (row decode, tap masks, pointers)", "issue of the first NS-1 stages", "first stage landed", "K loop (rest)",
             "drain + barrier (slowest wave)", "partials -> LDS + barrier", "reduce + epilogue (stores issued)", "stores retired"]
    idx = [(0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9)]
    if RF:
        names = ["prologue (decode, descriptors, epilogue operand requests)", "all loads of batch 0 requested", "first fragment landed + 4 MFMAs",
                 "rest of the K slice (MFMAs behind counted waits)", "partials -> LDS + barrier", "reduce + epilogue (stores issued)", "stores retired"]
        idx = [(0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 8), (8, 9)]
    if KD:
        names = ["prologue (arguments, decode, descriptors)", "DMAs of batch 0 issued", "(next batch issued,) batch 0 landed", "K loop (rest)",
                 "drain + barrier (slowest wave)", "partials -> LDS + barrier", "reduce + epilogue (stores issued)", "stores retired"]
        idx = [(0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9)]
    for tag, t in zip(("warm", "cold"), res):
        n = len(t)
        if n == 0:
            print("  [%s] (this kernel carries no stamps)" % tag)
            continue
        rt0, rt1 = t[:, 1].min(), t[:, 10].max()
        print("  [%s] %d blocks; wall (100 MHz clock): first block start -> last block end %.2f us; block start spread %.2f us; "
              "block lifetime median %.2f max %.2f us" % (tag, n, (rt1 - rt0) / 100.0, (t[:, 1].max() - rt0) / 100.0,
                                                          np.median(t[:, 10] - t[:, 1]) / 100.0, (t[:, 10] - t[:, 1]).max() / 100.0))
        clk = np.median((t[:, 9] - t[:, 0]) / np.maximum(t[:, 10] - t[:, 1], 1)) * 100.0 / 1e3
        print("        shader clock while the blocks ran: %.2f GHz" % clk)
        for (a_, b_), nm in zip(idx, names):
            d = t[:, b_] - t[:, a_]
            print("        %-46s median %6d  max %6d clocks" % (nm, np.median(d), d.max()))
    return us


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rf":                     # the register-fed kernel on the layers it serves, forced builds
        RF = True
        L.ore_conv_set_plan_override(-10, 2, 0, 0, 0)
        for shape, builds in (((20, 20, 112, 112, 3), ((1, 4, 16), (1, 8, 12), (2, 4, 16))), ((20, 20, 384, 112, 3), ((1, 16, 12), (1, 8, 16), (1, 8, 20))),
                              ((20, 20, 512, 128, 1), ((1, 4, 8), (1, 8, 12))), ((40, 40, 96, 96, 3), ((1, 4, 16), (1, 8, 12), (2, 4, 16))),
                              ((40, 40, 256, 96, 3), ((1, 8, 20), (1, 16, 12), (2, 8, 12))), ((40, 40, 384, 128, 1), ((1, 4, 8), (1, 4, 12))),
                              ((1, 320, 8192, 128, 1), ((1, 16, 12), (1, 8, 20), (2, 8, 12))), ((20, 20, 720, 512, 1), ((1, 4, 12), (2, 4, 12))),
                              ((40, 40, 544, 384, 1), ((1, 4, 12), (2, 4, 12)))):
            for gb, nw, maxs in builds:
                L.ore_conv_set_plan_override(-11, gb, nw, maxs, 0)
                print("#### k_conv_rf<GB %d, NW %d, MAXS %d>" % (gb, nw, maxs))
                trace(*shape, reps=2)
        L.ore_conv_set_plan_override(-11, 0, 0, 0, 0)
        L.ore_conv_set_plan_override(-10, 1, 0, 0, 0)
    elif len(sys.argv) > 1 and sys.argv[1] == "kd":                   # the lean LDS-DMA kernel, forced builds, on every small / medium-M layer
        KD = True
        L.ore_conv_set_plan_override(-12, 2, 0, 0, 0)
        L.ore_conv_set_plan_override(-10, 0, 0, 0, 0)
        for shape, builds in (((20, 20, 112, 112, 3), ((16, 16, 4, 8), (16, 16, 8, 4), (16, 16, 16, 2), (16, 32, 4, 4))),
                              ((20, 20, 384, 112, 3), ((16, 16, 4, 8), (16, 16, 8, 4), (16, 16, 16, 2), (16, 32, 8, 2))),
                              ((20, 20, 512, 128, 1), ((16, 16, 4, 8), (16, 16, 8, 4), (16, 32, 4, 4))),
                              ((20, 20, 720, 512, 1), ((32, 32, 4, 4), (32, 64, 4, 2), (16, 64, 4, 2), (32, 32, 8, 2), (64, 64, 4, 2))),
                              ((40, 40, 96, 96, 3), ((16, 48, 4, 4), (32, 32, 4, 4), (32, 48, 4, 2), (16, 48, 8, 2))),
                              ((40, 40, 256, 96, 3), ((16, 48, 4, 4), (16, 48, 8, 2), (32, 48, 4, 2), (32, 32, 8, 2), (16, 32, 8, 2))),
                              ((40, 40, 384, 128, 1), ((32, 32, 4, 4), (32, 64, 4, 2), (16, 64, 4, 2), (32, 32, 8, 2))),
                              ((40, 40, 544, 384, 1), ((32, 64, 4, 2), (32, 80, 4, 2), (64, 64, 4, 2), (32, 32, 4, 4), (16, 64, 4, 2))),
                              ((80, 80, 256, 128, 1), ((32, 64, 4, 2), (64, 64, 4, 2), (32, 32, 4, 4), (16, 64, 4, 2))),
                              ((80, 105, 256, 128, 1), ((32, 64, 4, 2), (64, 64, 4, 2), (32, 32, 4, 4))),
                              ((1, 320, 8192, 128, 1), ((16, 16, 8, 4), (16, 16, 16, 2), (16, 32, 8, 2), (32, 32, 8, 2)))):
            for bm, bn, nw, sb in builds:
                L.ore_conv_set_plan_override(-13, bm, bn, nw, sb)
                print("#### k_conv_kd<%dx%d, NW %d, SB %d>" % (bm, bn, nw, sb))
                try:
                    trace(*shape, reps=2)
                except orehip.OreError as ex:
                    print("   not built:", ex)
        L.ore_conv_set_plan_override(-13, 0, 0, 0, 0)
        L.ore_conv_set_plan_override(-12, 1, 0, 0, 0)
        L.ore_conv_set_plan_override(-10, 1, 0, 0, 0)
    elif len(sys.argv) > 1 and sys.argv[1] == "kdbig":                # the lean-DMA kernel on the LARGE-M layers (stem_3, the stage-2 / 3 concats)
        KD = True
        L.ore_conv_set_plan_override(-12, 2, 0, 0, 0)
        for shape, stride in (((80, 80, 352, 256, 1), 1), ((160, 160, 320, 112, 1), 1), ((320, 320, 64, 128, 3), 2), ((80, 80, 256, 128, 1), 1)):
            L.ore_conv_set_plan_override(-13, 0, 0, 0, 0)
            L.ore_conv_set_plan_override(-12, 0, 0, 0, 0)
            print("#### the plan's kernel (k_conv_gs / k_conv_igemm / k_conv_kw)")
            KD = False
            us = trace(*shape, reps=1, stride=stride)
            KD = True
            L.ore_conv_set_plan_override(-12, 2, 0, 0, 0)
            for bm, bn, nw, sb in ((64, 64, 4, 2), (32, 64, 4, 2), (32, 80, 4, 2), (64, 128, 4, 1), (64, 112, 4, 1), (128, 64, 4, 1)):
                L.ore_conv_set_plan_override(-13, bm, bn, nw, sb)
                print("#### k_conv_kd<%dx%d, NW %d, SB %d>" % (bm, bn, nw, sb))
                try:
                    trace(*shape, reps=2, stride=stride)
                except orehip.OreError as ex:
                    print("   not built:", str(ex)[:100])
        L.ore_conv_set_plan_override(-13, 0, 0, 0, 0)
        L.ore_conv_set_plan_override(-12, 1, 0, 0, 0)
    elif len(sys.argv) > 1 and sys.argv[1] == "gd":                   # the shared-stage descriptor kernel on the large-M layers (timings only)
        for shape, stride in (((80, 80, 352, 256, 1), 1), ((160, 160, 320, 112, 1), 1), ((320, 320, 64, 128, 3), 2)):
            L.ore_conv_set_plan_override(-15, 0, 0, 0, 0)
            print("#### the plan's kernel")
            trace(*shape, reps=1, stride=stride)
            for bm, bn, ns in ((64, 128, 4), (64, 64, 4), (64, 112, 4), (128, 64, 4), (128, 128, 4), (32, 128, 4), (128, 112, 4), (208, 64, 4), (224, 64, 4),
                               (112, 64, 4), (96, 128, 4), (64, 128, 2), (128, 128, 2), (208, 64, 2), (128, 112, 2),
                               (128, 128, 14), (112, 128, 14), (128, 112, 14), (64, 128, 14), (128, 64, 14), (64, 64, 14), (128, 128, 12), (112, 128, 12)):
                L.ore_conv_set_plan_override(-15, bm, bn, ns, 0)
                print("#### k_conv_gd<%dx%d, NS %d>" % (bm, bn, ns))
                try:
                    trace(*shape, reps=1, stride=stride)
                except orehip.OreError as ex:
                    print("   not built:", str(ex)[:100])
        L.ore_conv_set_plan_override(-15, 0, 0, 0, 0)
    elif len(sys.argv) > 1 and sys.argv[1] == "gdmid":                # k_conv_gd forced on the mid-size 1x1 layers k_conv_kd serves (kd off while forced)
        for shape in ((80, 80, 256, 128, 1), (80, 105, 256, 128, 1), (40, 40, 544, 384, 1), (40, 40, 512, 128, 1), (20, 20, 768, 512, 1), (40, 40, 96, 96, 3), (40, 40, 256, 96, 3)):
            L.ore_conv_set_plan_override(-15, 0, 0, 0, 0)
            L.ore_conv_set_plan_override(-12, 1, 0, 0, 0)
            print("#### the plan's kernel")
            trace(*shape, reps=1)
            L.ore_conv_set_plan_override(-12, 0, 0, 0, 0)
            for bm, bn, ns in ((64, 64, 4), (64, 128, 4), (80, 64, 4), (80, 128, 4), (48, 64, 4), (32, 64, 4), (32, 128, 4), (64, 64, 14), (128, 64, 14), (112, 64, 4)):
                L.ore_conv_set_plan_override(-15, bm, bn, ns, 0)
                print("#### k_conv_gd<%dx%d, NS %d>" % (bm, bn, ns))
                try:
                    trace(*shape, reps=1)
                except orehip.OreError as ex:
                    print("   not built:", str(ex)[:100])
        L.ore_conv_set_plan_override(-15, 0, 0, 0, 0)
        L.ore_conv_set_plan_override(-12, 1, 0, 0, 0)
    elif len(sys.argv) > 1 and sys.argv[1] == "gdabl":                # k_conv_gd with parts of its chunk loop switched off (trace build; results are wrong by design)
        for shape, stride in (((80, 80, 352, 256, 1), 1), ((160, 160, 320, 112, 1), 1), ((320, 320, 64, 128, 3), 2)):
            for bm, bn, ns in ((64, 64, 4), (64, 64, 14), (112, 128, 14)):
                L.ore_conv_set_plan_override(-15, bm, bn, ns, 0)
                for flags, what in ((0, "all"), (8, "all but the barriers"), (1, "no MFMA"), (2, "no DMA in loop"), (6, "no DMA, no reads"), (6 + 8, "MFMAs only (no DMA, reads, barriers)"),
                                    (7, "barriers only"), (7 + 16, "barriers only, no stores"),
                                    (7 + 16 + 32, "barriers only, no stores, no prologue DMA"), (15 + 16 + 32, "nothing"), (16, "all but the stores")):
                    L.ore_conv_set_plan_override(-16, flags, 0, 0, 0)
                    print("#### k_conv_gd<%dx%d, NS %d> %s" % (bm, bn, ns, what))
                    trace(*shape, reps=1, stride=stride)
        L.ore_conv_set_plan_override(-16, 0, 0, 0, 0)
        L.ore_conv_set_plan_override(-15, 0, 0, 0, 0)
    elif len(sys.argv) > 1 and sys.argv[1] == "sweep":                 # ring depth / waves per block on the latency-bound shapes
        for shape in ((20, 20, 112, 112, 3), (40, 40, 96, 96, 3), (20, 20, 512, 128, 1)):
            for nw, ns in ((4, 2), (4, 3), (4, 4), (8, 2), (16, 2)):
                L.ore_conv_set_plan_override(-6, nw, 0, 0, 0)
                L.ore_conv_set_plan_override(-3, 16, 16, ns, 1)
                print("#### forced tile 16x16, NW %d, NS %d" % (nw, ns))
                try:
                    trace(*shape, reps=2)
                except orehip.OreError as ex:
                    print("   not built:", ex)
        L.ore_conv_set_plan_override(-6, 0, 0, 0, 0)
        L.ore_conv_set_plan_override(-3, 0, 0, 0, 0)
    elif len(sys.argv) > 5:
        trace(*(int(v) for v in sys.argv[1:6]))
    else:
        for shape in ((20, 20, 112, 112, 3), (20, 20, 384, 112, 3), (20, 20, 720, 512, 1), (20, 20, 512, 128, 1), (40, 40, 96, 96, 3),
                      (40, 40, 256, 96, 3), (40, 40, 544, 384, 1), (80, 80, 256, 128, 1)):
            trace(*shape)
