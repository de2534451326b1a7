#!/usr/bin/env python3
"""k_nms_scan_t with parts of its consumer loop switched off (trace build of the library): the stand-alone NMS call (k_nms_prep +
k_nms_mask_t + k_nms_scan_t, no early exit: all blocks are walked) on the bench image's 2400 candidates, HIP-event time per call for
each setting; the differences are what the parts cost.  Results of the ablated passes are garbage by design.  GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402

orehip.LIB_PATH = os.path.join(ROOT, "faster-orefsdet_amd", "lib", "libore_hip_trace.so")
import bench  # noqa: E402

dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev)
e = model.engine()
img = bench.synth_image(1).to(dev)
L = orehip.lib()
e.eval_forward(img, use_graph=False)
torch.cuda.synchronize()
n = int(e.buffer("counts")[0, 0].item())
boxes, scores = e.buffer("pre_boxes")[:n].clone(), e.buffer("pre_scores")[:n, 0].clone()
print("candidates", n, "kept by a full NMS 0.6:", len(orehip.nms(boxes, scores, 0.6)))


def timed(reps=200):
    import ctypes as C
    keep = torch.zeros(n, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = L.ore_nms_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        rc = L.ore_nms_fwd(C.c_void_p(boxes.data_ptr()), C.c_void_p(scores.data_ptr()), n, C.c_float(0.6), C.c_void_p(keep.data_ptr()),
                           C.c_void_p(cnt.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(wsb), C.c_void_p(st))
        assert rc == 0
    for _ in range(10):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3, int(cnt.item())


names = {0: "everything", 1: "no AND loop", 2: "no fixpoint", 3: "no AND, no fixpoint", 7: "no AND / fixpoint / wait for columns", 16: "no emission by the producer waves",
         32: "pollers sleep 64x longer", 48: "no emission, long sleeps", 49: "no emission, long sleeps, no AND"}
base = None
import os as _os
flags = [int(_os.environ['ORE_ABL'])] * 2 if 'ORE_ABL' in _os.environ else (0, 1, 2, 3, 7, 16, 32, 48, 49, 0)
names = dict({f: str(f) for f in range(64)}, **names)
for f in flags:
    L.ore_debug_set_det_ablate(f)
    torch.cuda.synchronize()
    us, k = timed()
    base = base or us
    print("flags %2d %-45s %8.2f us / call (3 kernels)  (%+6.2f)  kept %d" % (f, names[f], us, us - base, k))
L.ore_debug_set_det_ablate(0)
