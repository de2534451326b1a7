"""bench.py's FLOP pricing restated from the layer table (taken from executing the reference's VoVNet + FPN at
1x3x640x640): the function of the input size that prices the training step must give the table's rows at 640x640, and scale with
the padded size elsewhere.  CPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

# SURVEY Appendix A, MMAC per row at 640x640 (ese.fc rows are the oc*oc terms)
FROZEN_ROWS = [176.95, 3774.87, 1887.44,                                   # stem_1..3
               1887.44, 943.72, 943.72, 917.50, 0.01,                      # stage 2
               516.10, 368.64, 368.64, 576.72, 0.07]                       # stage 3 (ese.fc 256*256)
TRAIN_ROWS = [353.89, 132.71, 132.71, 334.23, 0.15,                        # stage 4 (ese.fc 384*384)
              154.83, 45.16, 45.16, 147.46, 0.26,                          # stage 5 (ese.fc 512*512)
              26.21, 58.98, 78.64, 235.93, 209.72, 943.72]                 # FPN laterals / outputs


def test_layer_table_at_the_headline_size():
    frozen, trainable, rows, no_dgrad = bench.layer_table_macs(640, 640)
    assert rows == (80 * 80, 40 * 40, 20 * 20)
    assert abs(frozen / 1e6 - sum(FROZEN_ROWS)) < 0.1
    assert abs(trainable / 1e6 - sum(TRAIN_ROWS)) < 0.1
    assert abs((frozen + trainable) / 1e6 - 15261.57) < 0.35               # the table's total (rows are rounded to 0.01)
    # convs reading a frozen map: OSA4 layers.0, the 256-channel slice of OSA4's concat, fpn_lateral3
    assert abs(no_dgrad / 1e6 - (353.89 + 1600 * 256 * 384 / 1e6 + 209.72)) < 0.05


def test_layer_table_pads_to_32_and_pools_with_ceil():
    a = bench.layer_table_macs(640, 640)
    b = bench.layer_table_macs(609, 633)                                   # pads to 640x640 (size_divisibility 32)
    assert a == b
    f, t, rows, _ = bench.layer_table_macs(240, 240)                       # a support crop: padded to 256x256
    assert rows == (32 * 32, 16 * 16, 8 * 8)
    f2, t2, rows2, _ = bench.layer_table_macs(256, 256)
    assert (f, t, rows) == (f2, t2, rows2)
    # conv work scales with the area up to the ese.fc terms, which do not depend on the size
    fc = 112 * 112 + 256 * 256 + 384 * 384 + 512 * 512
    big = sum(bench.layer_table_macs(640, 640)[:2]) - fc
    small = f + t - fc
    assert abs(big / small - (640 * 640) / (256 * 256)) < 1e-9


def test_train_step_price_per_query_image():
    g1 = bench.train_step_gflop_expected(1, 640, 24)
    g16 = bench.train_step_gflop_expected(16, 640, 24)
    assert abs(g1 - 219.05) < 0.05 and abs(g16 - 16 * g1) < 1e-6           # the figure DESIGN.md section 8 / profiles quote
    # the pieces: query conv stack (frozen fwd + 3x trainable - no-dgrad reads) dominates; the support branch adds 24 crops of 256x256
    fq, tq, _, ndq = bench.layer_table_macs(640, 640)
    fs, ts, _, nds = bench.layer_table_macs(240, 240)
    conv = 2e-9 * ((fq + 3 * tq - ndq) + 24 * (fs + 3 * ts - nds))
    assert 0.6 * g1 < conv < g1
