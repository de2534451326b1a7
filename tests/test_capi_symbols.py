"""CPU: the C-ABI library builds/loads and exports every symbol include/ore_hip.h declares (no compute calls)."""
import os
import re

import pytest

from conftest import ROOT


def test_header_symbols_exported():
    import orehip
    if not os.path.exists(orehip.LIB_PATH):
        orehip.build()
    L = orehip.lib()
    hdr = open(os.path.join(ROOT, "include", "ore_hip.h")).read()
    declared = set(re.findall(r"\b(ore_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ore_engine"}
    assert declared, "no declarations parsed"
    assert declared == set(orehip.SYMBOLS), declared ^ set(orehip.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), f"{s} declared in ore_hip.h but not exported"
    assert L.ore_version() >= 100


def test_pack_weight_host_only():
    import numpy as np
    import torch
    import orehip
    w = torch.arange(2 * 16 * 3 * 3, dtype=torch.float32).reshape(2, 16, 3, 3)
    p = orehip.pack_conv_weight(w).numpy().reshape(16, 9, 16)
    assert np.array_equal(p[1, 4], w[1, :, 1, 1].numpy())   # [n][tap][cin]
    assert (p[2:] == 0).all()


def test_struct_sizes_match_header():
    """ctypes mirrors of the descriptor structs must have the C layout (checked against a tiny C program)."""
    import ctypes, subprocess, tempfile
    import orehip
    src = '#include <stdio.h>\n#include "ore_hip.h"\nint main(){printf("%zu %zu %zu", sizeof(ore_conv_desc), sizeof(ore_detect_desc), sizeof(ore_model_cfg));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        a, b, c = (int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split())
    assert (a, b, c) == (ctypes.sizeof(orehip.ConvDesc), ctypes.sizeof(orehip.DetectDesc), ctypes.sizeof(orehip.ModelCfg))
