#!/usr/bin/env python3
"""profiles/r01_bench_image_timeline.txt (tools/trace_summary.py on a rocprofv3 --kernel-trace of `bench.py --inflight 1`) ->
per-conv-launch efficiency table profiles/r01_conv_layers.txt.  Layer order = launch order of the engine at 640x640."""
import re
import sys

LAYERS = [("stem2", 320, 320, 64, 64, 3, 1), ("stem3", 320, 320, 64, 128, 3, 2), ("s2l0", 160, 160, 128, 64, 3, 1), ("s2l1", 160, 160, 64, 64, 3, 1),
          ("s2l2", 160, 160, 64, 64, 3, 1), ("s2cat", 160, 160, 320, 112, 1, 1), ("s3l0", 80, 80, 112, 80, 3, 1), ("s3l1", 80, 80, 80, 80, 3, 1),
          ("s3l2", 80, 80, 80, 80, 3, 1), ("s3cat", 80, 80, 352, 256, 1, 1), ("s4l0", 40, 40, 256, 96, 3, 1), ("s4l1", 40, 40, 96, 96, 3, 1),
          ("s4l2", 40, 40, 96, 96, 3, 1), ("s4cat", 40, 40, 544, 384, 1, 1), ("s5l0", 20, 20, 384, 112, 3, 1), ("s5l1", 20, 20, 112, 112, 3, 1),
          ("s5l2", 20, 20, 112, 112, 3, 1), ("s5cat", 20, 20, 720, 512, 1, 1), ("lat5", 20, 20, 512, 128, 1, 1), ("out5", 20, 20, 128, 128, 3, 1),
          ("lat4", 40, 40, 384, 128, 1, 1), ("out4", 40, 40, 128, 128, 3, 1), ("lat3", 80, 80, 256, 128, 1, 1), ("out3", 80, 80, 128, 128, 3, 1)]
EXTRA = [("conv3 (3 levels)", 8400, 256, 128, 1), ("head tower (3 levels)", 8400, 128, 128, 3), ("head reg|hm (3 levels)", 8400, 128, 5, 3),
         ("roi DSA+fc1 (320 rois)", 320, 8192, 128, 1)]
PEAK = 157.3


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02_bench_image_timeline.txt"
    dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_conv_layers.txt"
    # the head's (reg|hm) step is a VALU kernel since round 3 (k_head_pred): it is listed with the conv launches it replaced
    tl = [l for l in open(src) if l.startswith("k_conv") or l.startswith("k_head_pred")]
    # 28 launches: one per FPN output conv (round 2, and the bf16-storage engine); 26: the fp32 engine of round 3 runs the three output
    # convs as ONE per-level Winograd launch behind the lateral chain
    assert len(tl) in (26, 28), len(tl)
    layers, extra = LAYERS, EXTRA
    if len(tl) == 26:
        layers = LAYERS[:18] + [LAYERS[18], LAYERS[20], LAYERS[22]]
        extra = [("FPN output3/4/5 (one launch)", 8400, 128, 128, 3)] + EXTRA
    out = ["# per-conv-launch efficiency of ONE image, strictly sequential mode (rocprofv3 --kernel-trace of the headline protocol, tools/protocol_loop.py,",
           "# %s); algorithmic FLOPs = 2*M*Cout*Cin*k*k; peak = %.1f TFLOP/s (fp32 MFMA, gfx950)" % (src, PEAK),
           "%-28s %-44s %9s %8s %8s %7s" % ("layer", "kernel / grid", "us", "GFLOP", "TFLOP/s", "% peak")]
    tu = tg = te = 0.0
    for i, l in enumerate(tl):
        m = re.match(r"(\S+.*?)\s+grid=(\([^)]*\))\s+dur=\s*([\d.]+)", l)
        kn, grid, us = m.group(1).strip(), m.group(2), float(m.group(3))
        if i < len(layers):
            n, H, W, ci, co, k, s = layers[i]
            gf = 2.0 * (H // s) * (W // s) * co * ci * k * k / 1e9
        else:
            n, M, ci, co, k = extra[i - len(layers)]
            gf = 2.0 * M * co * ci * k * k / 1e9
        tu += us
        tg += gf
        te += gf / 2.25 if "wino" in kn else gf                    # a Winograd F(2x2,3x3) launch executes 16 of 36 multiplies per tile
        out.append("%-28s %-44s %9.2f %8.3f %8.1f %7.1f" % (n, (kn + " " + grid)[:44], us, gf, gf / us * 1e3, gf / us * 1e3 / PEAK * 100))
    out.append("%-28s %-44s %9.2f %8.3f %8.1f %7.1f" % ("all %d conv launches" % len(tl), "", tu, tg, tg / tu * 1e3, tg / tu * 1e3 / PEAK * 100))
    out.append("# multiplies actually executed by the matrix cores (Winograd launches count 1/2.25): %.3f GFLOP = %.1f TFLOP/s = %.1f %% of peak"
               % (te, te / tu * 1e3, te / tu * 1e3 / PEAK * 100))
    open(dst, "w").write("\n".join(out) + "\n")
    print(out[-2])
    if len(sys.argv) > 3:                                          # ore_version: also the JSON bench.py reads (roofline.frac_rocprof)
        import json
        with open(dst.rsplit(".", 1)[0] + ".json", "w") as f:
            json.dump({"ore_version": int(sys.argv[3]), "conv_us_per_image": round(tu, 2), "gflop_per_image": round(tg, 3),
                       "frac": round(tg / tu * 1e3 / PEAK, 4), "gflop_executed_per_image": round(te, 3),
                       "mfma_executed_frac": round(te / tu * 1e3 / PEAK, 4), "launches": len(tl), "from": src.split("/")[-1]}, f)


if __name__ == "__main__":
    main()
