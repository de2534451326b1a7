"""ctypes binding of libore_hip.so (include/ore_hip.h) -- the only way the Python host side reaches
the HIP kernels.  PyTorch is used for device memory (tensor.data_ptr()) and streams only.

There is NO CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libore_hip.so")
CSRC = os.path.join(_PKG, "csrc")
ESE_PARTS = 512


class OreError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libore_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if out.returncode != 0:
        raise OreError("building libore_hip.so failed:\n" + out.stdout[-4000:] + out.stderr[-4000:])
    if verbose:
        print(out.stdout[-2000:])
    return LIB_PATH


class ConvDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_ld", C.c_int32), ("in_coff", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("w", C.c_void_p),
                ("Cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("relu_cout", C.c_int32),
                ("in_mul", C.c_void_p), ("in_add", C.c_void_p), ("in_relu", C.c_int32),
                ("add", C.c_void_p), ("add_ld", C.c_int32), ("add_coff", C.c_int32),
                ("out", C.c_void_p), ("out_ld", C.c_int32), ("out_coff", C.c_int32),
                ("splitk", C.c_int32), ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t), ("colsum", C.c_void_p),
                ("w_wino", C.c_void_p), ("storage", C.c_int32), ("w_wino_level_stride", C.c_int64), ("w_level_stride", C.c_int64)]


class DetectDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("head", C.c_void_p * 8), ("head_ld", C.c_int32),
                ("H", C.c_int32 * 8), ("W", C.c_int32 * 8), ("stride", C.c_int32 * 8),
                ("score_thresh", C.c_float), ("pre_topk", C.c_int32), ("nms_thresh", C.c_float), ("post_topk", C.c_int32),
                ("pre_boxes", C.c_void_p), ("pre_scores", C.c_void_p), ("pre_loc", C.c_void_p), ("pre_level", C.c_void_p),
                ("keep_idx", C.c_void_p), ("counts", C.c_void_p), ("out_boxes", C.c_void_p), ("out_scores", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class ModelCfg(C.Structure):
    _fields_ = [("stem_ch", C.c_int32 * 3), ("stage_conv_ch", C.c_int32 * 4), ("stage_out_ch", C.c_int32 * 4),
                ("layers_per_block", C.c_int32), ("fpn_ch", C.c_int32), ("strides", C.c_int32 * 3),
                ("pixel_mean", C.c_float * 3), ("pixel_std", C.c_float * 3),
                ("score_thresh", C.c_float), ("pre_topk", C.c_int32), ("nms_thresh", C.c_float), ("post_topk", C.c_int32),
                ("max_batch", C.c_int32), ("max_h", C.c_int32), ("max_w", C.c_int32)]


# every symbol include/ore_hip.h declares (checked by tests/test_capi_symbols.py without a GPU)
SYMBOLS = ["ore_last_error", "ore_version", "ore_flop_counter_read", "ore_engine_profile_executed_flops", "ore_pack_conv_weights_multi_fwd", "ore_roi_align_bwd_det", "ore_roi_align_bwd_tiled", "ore_roi_losses_fwd", "ore_sample_rois_fwd", "ore_det_record_rows", "ore_winograd_covers", "ore_winograd_weight_floats", "ore_winograd_weight_fwd", "ore_packed_weight_bf16_elems", "ore_pack_conv_weight_bf16_host", "ore_engine_buffer_is_bf16", "ore_stem1_bf16_fwd", "ore_ese_gate_bf16_fwd", "ore_maxpool3x3s2_bf16_fwd", "ore_correlation_levels_bf16_fwd", "ore_groupnorm_affine_levels_bf16_fwd", "ore_groupnorm_apply_bf16_fwd", "ore_head_pred_fwd", "ore_head_pred_bf16_fwd", "ore_head_pred_gn_fwd", "ore_head_pred_gn_bf16_fwd", "ore_groupnorm_apply_levels_bf16_fwd", "ore_ese_gate_scaled_weight_bf16_fwd", "ore_ese_gate_pool_fwd", "ore_ese_gate_pool_bf16_fwd", "ore_roi_align_bf16_fwd", "ore_conv2d_fwd", "ore_conv2d_levels_fwd", "ore_conv_workspace_floats", "ore_conv_colsum_rows", "ore_conv_set_plan_override", "ore_packed_weight_floats", "ore_pack_conv_weight_host",
           "ore_stem1_fwd", "ore_maxpool3x3s2_fwd", "ore_ese_gate_fwd", "ore_ese_gate_from_colsum_fwd", "ore_ese_gate_scaled_weight_fwd", "ore_correlation_levels_fwd",
           "ore_groupnorm_affine_levels_fwd", "ore_scale_channels_fwd", "ore_correlation_fwd",
           "ore_support_kernels_fwd", "ore_groupnorm_affine_fwd", "ore_detect_workspace_bytes", "ore_detect_fwd", "ore_detect_batch_fwd",
           "ore_nms_workspace_bytes", "ore_nms_fwd", "ore_nms_device_n_fwd", "ore_roi_align_fwd", "ore_roi_predict_workspace_bytes",
           "ore_roi_predict_fwd", "ore_roi_align_batched_fwd", "ore_roi_align_bwd", "ore_centernet_targets_fwd", "ore_centernet_losses_fwd", "ore_centernet_losses_bwd", "ore_sgd_step_fwd", "ore_pack_conv_weight_fwd", "ore_conv_wgrad_workspace_floats",
           "ore_conv_set_precision", "ore_conv_get_precision", "ore_conv2d_wgrad_fwd", "ore_conv2d_wgrad_bias_fwd", "ore_granule_transpose_fwd", "ore_combine2_fwd", "ore_combine2_bwd", "ore_adaptive_avgpool_nhwc_fwd", "ore_adaptive_avgpool_nhwc_bwd", "ore_group_mean_fwd", "ore_group_mean_bwd", "ore_relu_affine_bwd", "ore_colsum_fwd", "ore_colsum_segments_fwd", "ore_correlation_train_fwd", "ore_correlation_train_bwd",
           "ore_groupnorm_apply_fwd", "ore_groupnorm_bwd", "ore_prod_colsum_fwd", "ore_scale_add_channels_fwd", "ore_maxpool3x3s2_bwd", "ore_sumpool2x2_fwd", "ore_engine_create", "ore_engine_destroy", "ore_engine_set_tensor",
           "ore_engine_set_support", "ore_engine_finalize", "ore_engine_set_roi_head", "ore_engine_backbone_fwd", "ore_engine_eval_fwd", "ore_engine_eval_batch_fwd", "ore_engine_detect_fwd", "ore_engine_detect_begin", "ore_engine_detect_end", "ore_roi_predict_post_fwd",
           "ore_engine_buffer", "ore_engine_last_flops", "ore_engine_set_profiling", "ore_engine_read_profile", "ore_event_pair_overhead_us",
           "ore_rccl_load", "ore_rccl_unique_id", "ore_rccl_comm_create", "ore_rccl_comm_destroy", "ore_allreduce_grads"]

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OreError(f"{LIB_PATH} not found: build it with `make -C {CSRC}` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.ore_last_error.restype = C.c_char_p
        L.ore_packed_weight_floats.restype = C.c_size_t
        L.ore_winograd_weight_floats.restype = C.c_size_t
        L.ore_packed_weight_bf16_elems.restype = C.c_size_t
        L.ore_conv_workspace_floats.restype = C.c_size_t
        L.ore_conv_colsum_rows.restype = C.c_int32
        L.ore_detect_workspace_bytes.restype = C.c_size_t
        L.ore_nms_workspace_bytes.restype = C.c_size_t
        L.ore_roi_predict_workspace_bytes.restype = C.c_size_t
        L.ore_conv_wgrad_workspace_floats.restype = C.c_size_t
        L.ore_engine_last_flops.restype = C.c_double
        L.ore_engine_last_flops.argtypes = [C.c_void_p]
        L.ore_engine_profile_executed_flops.restype = C.c_double
        L.ore_engine_profile_executed_flops.argtypes = [C.c_void_p]
        L.ore_engine_destroy.argtypes = [C.c_void_p]
        L.ore_engine_destroy.restype = None
        _lib = L
    return _lib


def _chk(rc: int, what: str):
    if rc != 0:
        raise OreError(f"{what} failed ({rc}): {lib().ore_last_error().decode()}")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> C.c_void_p:
    """The current torch stream of the current device as a hipStream_t (the raw-handle query is ~10x cheaper than building a
    torch.cuda.Stream object per launch; thousands of launches per training step go through here)."""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "libore_hip.so works on device memory only"
    return t.data_ptr()


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


# ---------------------------------------------------------------------------------------------------
# layout helpers: the library is NHWC; torch.channels_last IS that layout for a logical NCHW tensor
# ---------------------------------------------------------------------------------------------------
def to_nhwc(x_nchw: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] any layout -> contiguous [B,H,W,C] fp32."""
    return x_nchw.permute(0, 2, 3, 1).contiguous().float()


def as_nchw(x_nhwc: torch.Tensor) -> torch.Tensor:
    """[B,H,W,C] -> logical [B,C,H,W] view (channels_last strides, zero copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


def pack_conv_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """OIHW (any device) -> packed [Cout16][kh*kw*Cin] on the weight's device."""
    w = w_oihw.detach().float().cpu().contiguous()
    co, ci, kh, kw = w.shape
    n = lib().ore_packed_weight_floats(co, ci, kh, kw)
    dst = np.empty(n, dtype=np.float32)
    _chk(lib().ore_pack_conv_weight_host(C.c_void_p(w.data_ptr()), co, ci, kh, kw, dst.ctypes.data_as(C.c_void_p)),
         "ore_pack_conv_weight_host")
    return torch.from_numpy(dst).to(w_oihw.device)


def pack_conv_weight_bf16(w_oihw: torch.Tensor) -> torch.Tensor:
    """OIHW fp32 (any device) -> bf16 packed [Cout16][kh*kw][round_up(Cin, 32)] on the weight's device (ORE_ST_BF16 convs)."""
    w = w_oihw.detach().float().cpu().contiguous()
    co, ci, kh, kw = w.shape
    n = lib().ore_packed_weight_bf16_elems(co, ci, kh, kw)
    dst = np.empty(n, dtype=np.uint16)
    _chk(lib().ore_pack_conv_weight_bf16_host(C.c_void_p(w.data_ptr()), co, ci, kh, kw, dst.ctypes.data_as(C.c_void_p)),
         "ore_pack_conv_weight_bf16_host")
    return torch.from_numpy(dst.view(np.int16)).view(torch.bfloat16).to(w_oihw.device)


def winograd_covers(Cout: int, Cin: int) -> bool:
    """True if a Winograd build exists for a 3x3 stride-1 layer of these widths (ore_winograd_covers)."""
    return bool(lib().ore_winograd_covers(int(Cout), int(Cin)))


def winograd_weight(w_packed: torch.Tensor, Cout: int, Cin: int) -> torch.Tensor:
    """Packed 3x3 weights ([Cout16][9][Cin], pack_weight) -> their Winograd F(2x2,3x3) form U [16][Cout16][Cin] on the device."""
    _f32(w_packed)
    U = torch.empty(int(lib().ore_winograd_weight_floats(Cout, Cin)), device=w_packed.device, dtype=torch.float32)
    _chk(lib().ore_winograd_weight_fwd(C.c_void_p(_ptr(w_packed)), Cout, Cin, C.c_void_p(_ptr(U)), _stream()), "ore_winograd_weight_fwd")
    return U


def _bf16_in_with_slack(x: torch.Tensor, Cin: int, in_coff: int, ld: int) -> torch.Tensor:
    """ore_hip.h (ORE_ST_BF16): when Cin % 32 == 16 and the slice ends at the row end, the last K chunk of the LAST row reads 16
    channels (32 bytes) past the tensor against zero weights -- the bytes must exist and be finite (0 * NaN = NaN).  A tensor whose
    storage already extends that far (a view into a wider buffer: the caller owns what lies there, as the header says) is taken as
    it is; a plain tensor, whose storage ends with its last element, is copied once into a buffer with 16 zeroed elements of slack."""
    if Cin % 32 == 0 or in_coff + Cin != ld:
        return x
    end = x.storage_offset() + x.numel()
    if x.untyped_storage().nbytes() // 2 - end >= 16:
        return x
    buf = torch.empty(x.numel() + 16, device=x.device, dtype=torch.bfloat16)
    buf[x.numel():].zero_()
    buf[:x.numel()].copy_(x.reshape(-1))
    return buf[:x.numel()].view(x.shape)


def conv2d(x: torch.Tensor, w_packed: torch.Tensor, Cout: int, k: int, stride: int = 1, pad: Optional[int] = None, *,
           in_coff: int = 0, Cin: Optional[int] = None, scale=None, shift=None, relu_cout: int = 0, in_mul=None,
           in_add=None, in_relu: bool = False, add: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           out_coff: int = 0, splitk: int = 0, workspace: Optional[torch.Tensor] = None, want_colsum: bool = False,
           w_wino: Optional[torch.Tensor] = None, out_f32: bool = False):
    """x: [B,H,W,ld] NHWC fp32.  Returns `out` ([B,Ho,Wo,out_ld]); only channels [out_coff, out_coff+Cout) are written.
    w_wino: the same weights in Winograd F(2x2,3x3) form (winograd_weight); lets the large-M 3x3 layers run on the Winograd kernel.
    bf16 storage (ORE_ST_BF16): x / w_packed (pack_conv_weight_bf16) / add are torch.bfloat16; `out` is bf16, or fp32 when out_f32."""
    st_bf16 = x.dtype == torch.bfloat16
    if st_bf16:
        assert x.is_contiguous() and w_packed.dtype == torch.bfloat16 and (add is None or add.dtype == torch.bfloat16)
    else:
        _f32(x)
    B, H, W, ld = x.shape
    Cin = Cin if Cin is not None else ld - in_coff
    if st_bf16:
        x = _bf16_in_with_slack(x, Cin, in_coff, ld)
    pad = k // 2 if pad is None else pad
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out_dt = torch.bfloat16 if (st_bf16 and not out_f32) else torch.float32
    if out is None:
        out = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=out_dt)
    assert out.dtype == out_dt and out.is_contiguous()
    assert out.shape[:3] == (B, Ho, Wo)
    M = B * Ho * Wo
    if workspace is None and splitk != 1:
        workspace = _default_ws(x.device)
    d = ConvDesc()
    d.in_, d.in_ld, d.in_coff = _ptr(x), ld, in_coff
    d.B, d.H, d.W, d.Cin = B, H, W, Cin
    d.storage = (2 if out_f32 else 1) if st_bf16 else 0
    d.w = _ptr(w_packed if st_bf16 else _f32(w_packed))
    d.Cout, d.kh, d.kw, d.stride, d.pad = Cout, k, k, stride, pad
    d.scale, d.shift, d.relu_cout = _ptr(scale), _ptr(shift), relu_cout
    d.in_mul, d.in_add, d.in_relu = _ptr(in_mul), _ptr(in_add), int(in_relu)
    if add is not None:
        assert add.is_contiguous()
        d.add, d.add_ld, d.add_coff = _ptr(add), add.shape[-1], 0
    d.out, d.out_ld, d.out_coff = _ptr(out), out.shape[-1], out_coff
    d.splitk = splitk
    d.w_wino = _ptr(w_wino)
    if workspace is not None:
        d.workspace, d.workspace_floats = _ptr(workspace), workspace.numel()
    colsum = None
    if want_colsum:
        rows = lib().ore_conv_colsum_rows(C.byref(d))
        colsum = torch.zeros(rows, (Cout + 15) // 16 * 16, device=x.device, dtype=torch.float32)
        d.colsum = _ptr(colsum)
    _chk(lib().ore_conv2d_fwd(C.byref(d), _stream()), "ore_conv2d_fwd")
    return (out, colsum) if want_colsum else out


def conv2d_levels(x_rows: torch.Tensor, HW: Sequence[Tuple[int, int]], B: int, w_packed: torch.Tensor, Cout: int, k: int, *,
                  in_coff: int = 0, Cin: Optional[int] = None, scale=None, shift=None, ep_stride: int = 0, relu_cout: int = 0,
                  in_mul=None, in_add=None, in_relu: bool = False, out: Optional[torch.Tensor] = None, out_coff: int = 0,
                  splitk: int = 0, w_wino: Optional[torch.Tensor] = None, out_f32: bool = False, w_wino_level_stride: int = 0,
                  w_level_stride: int = 0) -> torch.Tensor:
    """One launch over several pyramid levels: x_rows [sum_l B*H_l*W_l, ld] level-major (fp32, or bf16 = ORE_ST_BF16 storage)."""
    st_bf16 = x_rows.dtype == torch.bfloat16
    if not st_bf16:
        _f32(x_rows)
    rows, ld = x_rows.shape
    assert rows == sum(B * h * w for h, w in HW)
    Cin = Cin if Cin is not None else ld - in_coff
    if st_bf16:
        x_rows = _bf16_in_with_slack(x_rows, Cin, in_coff, ld)
    out_dt = torch.bfloat16 if (st_bf16 and not out_f32) else torch.float32
    if out is None:
        out = torch.empty(rows, Cout, device=x_rows.device, dtype=out_dt)
    assert out.dtype == out_dt and out.is_contiguous()
    ws = _default_ws(x_rows.device)
    d = ConvDesc()
    d.in_, d.in_ld, d.in_coff = _ptr(x_rows), ld, in_coff
    d.B, d.H, d.W, d.Cin = B, 0, 0, Cin
    d.storage = (2 if out_f32 else 1) if st_bf16 else 0
    d.w = _ptr(w_packed)
    d.Cout, d.kh, d.kw, d.stride, d.pad = Cout, k, k, 1, k // 2
    d.scale, d.shift, d.relu_cout = _ptr(scale), _ptr(shift), relu_cout
    d.in_mul, d.in_add, d.in_relu = _ptr(in_mul), _ptr(in_add), int(in_relu)
    d.out, d.out_ld, d.out_coff = _ptr(out), out.shape[-1], out_coff
    d.splitk, d.workspace, d.workspace_floats = splitk, _ptr(ws), ws.numel()
    d.w_wino = _ptr(w_wino)
    d.w_wino_level_stride = int(w_wino_level_stride)   # != 0: the levels are different layers of one shape, w_wino = [L][stride] (Winograd only)
    d.w_level_stride = int(w_level_stride)             # the same for bf16 storage: w_packed = [L][stride] bf16 (weight-stationary 3x3 kernel)
    L = len(HW)
    Hs = (C.c_int32 * L)(*[h for h, _ in HW])
    Ws = (C.c_int32 * L)(*[w for _, w in HW])
    _chk(lib().ore_conv2d_levels_fwd(C.byref(d), L, Hs, Ws, ep_stride, _stream()), "ore_conv2d_levels_fwd")
    return out


def correlation_levels(q_rows: torch.Tensor, HW: Sequence[Tuple[int, int]], B: int, k11, k13, k31, Cc: int, q_coff: int = 0,
                       out: Optional[torch.Tensor] = None, out_coff: int = 0) -> torch.Tensor:
    _f32(q_rows)
    if out is None:
        out = torch.empty(q_rows.shape[0], Cc, device=q_rows.device, dtype=torch.float32)
    L = len(HW)
    Hs = (C.c_int32 * L)(*[h for h, _ in HW])
    Ws = (C.c_int32 * L)(*[w for _, w in HW])
    _chk(lib().ore_correlation_levels_fwd(C.c_void_p(_ptr(q_rows)), q_rows.shape[-1], q_coff, B, L, Hs, Ws, Cc,
                                          C.c_void_p(_ptr(_f32(k11))), C.c_void_p(_ptr(_f32(k13))), C.c_void_p(_ptr(_f32(k31))),
                                          C.c_void_p(_ptr(out)), out.shape[-1], out_coff, _stream()), "ore_correlation_levels_fwd")
    return out


def groupnorm_affine_levels(x_rows: torch.Tensor, HW: Sequence[int], B: int, groups: int, gamma, beta, eps: float = 1e-5):
    _f32(x_rows)
    Cc = x_rows.shape[-1]
    L = len(HW)
    mul = torch.empty(L * B, Cc, device=x_rows.device)
    add = torch.empty(L * B, Cc, device=x_rows.device)
    ws = torch.empty(sum(B * ((hw + 63) // 64) for hw in HW) * groups * 2 + 64, device=x_rows.device)
    hws = (C.c_int32 * L)(*HW)
    _chk(lib().ore_groupnorm_affine_levels_fwd(C.c_void_p(_ptr(x_rows)), Cc, 0, B, L, hws, Cc, groups, C.c_float(eps),
                                               C.c_void_p(_ptr(_f32(gamma))), C.c_void_p(_ptr(_f32(beta))), C.c_void_p(_ptr(mul)),
                                               C.c_void_p(_ptr(add)), C.c_void_p(_ptr(ws)), _stream()), "ore_groupnorm_affine_levels_fwd")
    return mul, add


def ese_gate_from_colsum(part: torch.Tensor, HW: int, fc_w: torch.Tensor, fc_b: torch.Tensor) -> torch.Tensor:
    """part [P, C] (B = 1): the fused column sums of the concat conv."""
    _f32(part)
    P, Cc = part.shape
    gate = torch.empty(1, Cc, device=part.device, dtype=torch.float32)
    mean_ws = torch.empty(Cc, device=part.device, dtype=torch.float32)
    _chk(lib().ore_ese_gate_from_colsum_fwd(C.c_void_p(_ptr(part)), P, 1, HW, Cc, C.c_void_p(_ptr(_f32(fc_w.reshape(Cc, Cc)))),
                                            C.c_void_p(_ptr(_f32(fc_b))), C.c_void_p(_ptr(gate)), C.c_void_p(_ptr(mean_ws)), _stream()),
         "ore_ese_gate_from_colsum_fwd")
    return gate


def ese_gate_scaled_weight(part: torch.Tensor, HW: int, fc_w: torch.Tensor, fc_b: torch.Tensor, w_packed: torch.Tensor):
    """ese_gate_from_colsum for one image plus, in the same launch, the gate-scaled copy of a consumer's packed 1x1 weight [rows, C]
    (ore_ese_gate_scaled_weight_fwd).  Returns (gate [1, C], w_packed * gate[None, :])."""
    _f32(part); _f32(w_packed)
    P, Cc = part.shape
    assert w_packed.dim() == 2 and w_packed.shape[1] == Cc
    gate = torch.empty(1, Cc, device=part.device, dtype=torch.float32)
    mean_ws = torch.empty(Cc, device=part.device, dtype=torch.float32)
    ws = torch.empty_like(w_packed)
    _chk(lib().ore_ese_gate_scaled_weight_fwd(C.c_void_p(_ptr(part)), P, HW, Cc, C.c_void_p(_ptr(_f32(fc_w.reshape(Cc, Cc)))),
                                              C.c_void_p(_ptr(_f32(fc_b))), C.c_void_p(_ptr(gate)), C.c_void_p(_ptr(mean_ws)),
                                              C.c_void_p(_ptr(w_packed)), w_packed.shape[0], C.c_void_p(_ptr(ws)), _stream()),
         "ore_ese_gate_scaled_weight_fwd")
    return gate, ws


_WS: Dict[str, torch.Tensor] = {}


def _default_ws(device) -> torch.Tensor:
    k = str(device)
    if k not in _WS:
        _WS[k] = torch.zeros(lib().ore_conv_workspace_floats(), device=device, dtype=torch.float32)  # counters (zero) + slabs
    return _WS[k]


def stem1(img: torch.Tensor, Hp: int, Wp: int, mean: Sequence[float], std: Sequence[float], w_oihw: torch.Tensor,
          scale: torch.Tensor, shift: torch.Tensor, out_bf16: bool = False) -> torch.Tensor:
    """img [B,3,H,W] uint8 or fp32 (planar) -> [B,Hp/2,Wp/2,Cout] fp32, or bf16 (rounded once; fp32 arithmetic) when out_bf16."""
    assert img.is_contiguous() and img.dtype in (torch.uint8, torch.float32)
    B, _, H, W = img.shape
    Cout = w_oihw.shape[0]
    out = torch.empty(B, Hp // 2, Wp // 2, Cout, device=img.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    if out_bf16:
        _chk(lib().ore_stem1_bf16_fwd(C.c_void_p(_ptr(img)), int(img.dtype == torch.uint8), B, H, W, Hp, Wp, m, s,
                                      C.c_void_p(_ptr(_f32(w_oihw))), C.c_void_p(_ptr(scale)), C.c_void_p(_ptr(shift)), Cout,
                                      C.c_void_p(_ptr(out)), Cout, 0, _stream()), "ore_stem1_bf16_fwd")
        return out
    _chk(lib().ore_stem1_fwd(C.c_void_p(_ptr(img)), int(img.dtype == torch.uint8), B, H, W, Hp, Wp, m, s,
                             C.c_void_p(_ptr(_f32(w_oihw))), C.c_void_p(_ptr(scale)), C.c_void_p(_ptr(shift)), Cout,
                             C.c_void_p(_ptr(out)), Cout, 0, _stream()), "ore_stem1_fwd")
    return out


def maxpool3x3s2(x: torch.Tensor, in_mul: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_coff: int = 0) -> torch.Tensor:
    """MaxPool2d(3, 2, ceil_mode=True) on NHWC (optionally of x * in_mul[b][c], in_mul > 0).  `out` [B,Ho,Wo,ld] + out_coff: write the
    result into a channel slice of a wider buffer (the next block's concat buffer) instead of a fresh tensor.  bf16 in -> bf16 out
    (the product with the fp32 multiplier is rounded once)."""
    bf = x.dtype == torch.bfloat16
    assert x.is_contiguous() and (bf or x.dtype == torch.float32)
    B, H, W, Cc = x.shape

    def osz(n):
        o = (n - 3 + 1) // 2 + 1 if n >= 3 else 1
        if (o - 1) * 2 >= n:
            o -= 1
        return max(o, 1)
    if out is None:
        out = torch.empty(B, osz(H), osz(W), Cc, device=x.device, dtype=x.dtype)
    else:
        assert out.dtype == x.dtype and out.is_contiguous()
        assert tuple(out.shape[:3]) == (B, osz(H), osz(W)) and out_coff + Cc <= out.shape[-1] and out_coff % 4 == 0
    if bf:
        _chk(lib().ore_maxpool3x3s2_bf16_fwd(C.c_void_p(_ptr(x)), Cc, 0, B, H, W, Cc, C.c_void_p(_ptr(in_mul)),
                                             C.c_void_p(_ptr(out)), out.shape[-1], out_coff, _stream()), "ore_maxpool3x3s2_bf16_fwd")
        return out
    _chk(lib().ore_maxpool3x3s2_fwd(C.c_void_p(_ptr(x)), Cc, 0, B, H, W, Cc, C.c_void_p(_ptr(in_mul)),
                                    C.c_void_p(_ptr(out)), out.shape[-1], out_coff, _stream()), "ore_maxpool3x3s2_fwd")
    return out


def maxpool3x3s2_out_hw(H: int, W: int) -> Tuple[int, int]:
    def osz(n):
        o = (n - 3 + 1) // 2 + 1 if n >= 3 else 1
        if (o - 1) * 2 >= n:
            o -= 1
        return max(o, 1)
    return osz(H), osz(W)


def ese_gate(x: torch.Tensor, fc_w: torch.Tensor, fc_b: torch.Tensor) -> torch.Tensor:
    """relu6(fc(mean_hw(x)) + 3) / 6 per image: [B,C] fp32, from an fp32 or a bf16 map."""
    bf = x.dtype == torch.bfloat16
    assert x.is_contiguous() and (bf or x.dtype == torch.float32)
    B, H, W, Cc = x.shape
    gate = torch.empty(B, Cc, device=x.device, dtype=torch.float32)
    ws = torch.empty(B * (ESE_PARTS + 1) * Cc, device=x.device, dtype=torch.float32)
    if bf:
        _chk(lib().ore_ese_gate_bf16_fwd(C.c_void_p(_ptr(x)), Cc, 0, B, H * W, Cc, C.c_void_p(_ptr(_f32(fc_w.reshape(Cc, Cc)))),
                                         C.c_void_p(_ptr(_f32(fc_b))), C.c_void_p(_ptr(gate)), C.c_void_p(_ptr(ws)), _stream()),
             "ore_ese_gate_bf16_fwd")
        return gate
    _chk(lib().ore_ese_gate_fwd(C.c_void_p(_ptr(x)), Cc, 0, B, H * W, Cc, C.c_void_p(_ptr(_f32(fc_w.reshape(Cc, Cc)))),
                                C.c_void_p(_ptr(_f32(fc_b))), C.c_void_p(_ptr(gate)), C.c_void_p(_ptr(ws)), _stream()),
         "ore_ese_gate_fwd")
    return gate


def ese_gate_pool(part: torch.Tensor, fc_w: torch.Tensor, fc_b: torch.Tensor, x: torch.Tensor, w_packed: Optional[torch.Tensor] = None):
    """One image: gate from the concat conv's fused column sums `part` [P, C], the gate-scaled copy of `w_packed` [rows, C] (or None)
    and maxpool3x3s2(x * gate) in one launch (ore_ese_gate_pool_fwd).  x [1,H,W,C] fp32 or bf16.  Returns (gate [1,C], pooled, scaled)."""
    _f32(part)
    P, Cc = part.shape
    _, H, W, Cx = x.shape
    assert x.shape[0] == 1 and Cx == Cc and x.is_contiguous()
    Ho, Wo = maxpool3x3s2_out_hw(H, W)
    gate = torch.empty(1, Cc, device=x.device, dtype=torch.float32)
    out = torch.empty(1, Ho, Wo, Cc, device=x.device, dtype=x.dtype)
    bf = x.dtype == torch.bfloat16
    ws = None
    if w_packed is not None:
        _f32(w_packed)
        ws = torch.empty(w_packed.shape, device=x.device, dtype=x.dtype)
    fn = lib().ore_ese_gate_pool_bf16_fwd if bf else lib().ore_ese_gate_pool_fwd
    _chk(fn(C.c_void_p(_ptr(part)), P, H * W, Cc, C.c_void_p(_ptr(_f32(fc_w.reshape(Cc, Cc)))), C.c_void_p(_ptr(_f32(fc_b))),
            C.c_void_p(_ptr(gate)), C.c_void_p(_ptr(w_packed)), 0 if w_packed is None else w_packed.shape[0], C.c_void_p(_ptr(ws)),
            C.c_void_p(_ptr(x)), Cc, 0, H, W, C.c_void_p(_ptr(out)), Cc, 0, _stream()), "ore_ese_gate_pool_fwd")
    return gate, out, ws


def scale_channels(x: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    _f32(x)
    B, H, W, Cc = x.shape
    y = torch.empty_like(x)
    _chk(lib().ore_scale_channels_fwd(C.c_void_p(_ptr(x)), Cc, 0, B, H * W, Cc, C.c_void_p(_ptr(_f32(gate))),
                                      C.c_void_p(_ptr(y)), Cc, 0, _stream()), "ore_scale_channels_fwd")
    return y


def support_kernels(proto_chw: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    _f32(proto_chw)
    Cc, s, s2 = proto_chw.shape
    assert s == s2
    k11 = torch.empty(Cc, device=proto_chw.device)
    k13 = torch.empty(Cc, 3, device=proto_chw.device)
    k31 = torch.empty(Cc, 3, device=proto_chw.device)
    _chk(lib().ore_support_kernels_fwd(C.c_void_p(_ptr(proto_chw)), Cc, s, C.c_void_p(_ptr(k11)), C.c_void_p(_ptr(k13)),
                                       C.c_void_p(_ptr(k31)), _stream()), "ore_support_kernels_fwd")
    return k11, k13, k31


def correlation(q: torch.Tensor, k11, k13, k31, out: Optional[torch.Tensor] = None, q_coff: int = 0, out_coff: int = 0,
                Cc: Optional[int] = None) -> torch.Tensor:
    _f32(q)
    B, H, W, ld = q.shape
    Cc = Cc if Cc is not None else ld
    if out is None:
        out = torch.empty(B, H, W, Cc, device=q.device, dtype=torch.float32)
    _chk(lib().ore_correlation_fwd(C.c_void_p(_ptr(q)), ld, q_coff, B, H, W, Cc, C.c_void_p(_ptr(_f32(k11))),
                                   C.c_void_p(_ptr(_f32(k13))), C.c_void_p(_ptr(_f32(k31))), C.c_void_p(_ptr(out)),
                                   out.shape[-1], out_coff, _stream()), "ore_correlation_fwd")
    return out


def groupnorm_affine(x: torch.Tensor, groups: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    _f32(x)
    B, H, W, Cc = x.shape
    mul = torch.empty(B, Cc, device=x.device)
    add = torch.empty(B, Cc, device=x.device)
    ws = torch.empty(B * ((H * W + 63) // 64) * groups * 2, device=x.device)
    _chk(lib().ore_groupnorm_affine_fwd(C.c_void_p(_ptr(x)), Cc, 0, B, H * W, Cc, groups, C.c_float(eps),
                                        C.c_void_p(_ptr(_f32(gamma))), C.c_void_p(_ptr(_f32(beta))), C.c_void_p(_ptr(mul)),
                                        C.c_void_p(_ptr(add)), C.c_void_p(_ptr(ws)), _stream()), "ore_groupnorm_affine_fwd")
    return mul, add


def set_conv_precision(mode: str) -> str:
    """"fp32" (default), "bf16" (operands of the MFMA conv kernels rounded to bf16, fp32 tensors) or "bf16s" (bf16 STORAGE: engines
    created under it keep bf16 activations / weights in HBM and LDS) -- include/ore_hip.h, ore_conv_set_precision; engines keep the mode
    in force when they were created.  Returns the previous mode."""
    prev = get_conv_precision()
    _chk(lib().ore_conv_set_precision({"fp32": 0, "bf16": 1, "bf16s": 2}[mode]), "ore_conv_set_precision")
    return prev


def get_conv_precision() -> str:
    return ("fp32", "bf16", "bf16s")[int(lib().ore_conv_get_precision())]


def _detect_spec(L: int, pre_topk: int):
    cap = L * pre_topk
    capa = (cap + 3) // 4 * 4                                   # one zero-filled allocation carved into the 8 outputs (16-byte aligned)
    spec = (("pre_boxes", torch.float32, 4 * capa, (cap, 4)), ("pre_scores", torch.float32, capa, (cap,)),
            ("pre_loc", torch.int64, capa, (cap,)), ("pre_level", torch.int32, capa, (cap,)), ("keep_idx", torch.int64, capa, (cap,)),
            ("counts", torch.int32, 4, (4,)), ("out_boxes", torch.float32, 4 * capa, (cap, 4)), ("out_scores", torch.float32, capa, (cap,)))
    return spec, sum(n * (8 if dt == torch.int64 else 4) for _, dt, n, _ in spec)


def _detect_desc(heads: Sequence[torch.Tensor], strides, score_thresh, pre_topk, nms_thresh, post_topk, d: "DetectDesc",
                 raw: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None):
    """Allocate one image's outputs + workspace (or carve them out of the caller's `raw` (zeroed) / `ws`) and fill its descriptor."""
    L = len(heads)
    dev = heads[0].device
    cap = L * pre_topk
    spec, raw_bytes = _detect_spec(L, pre_topk)
    if raw is None:
        raw = torch.zeros(raw_bytes, dtype=torch.uint8, device=dev)
    o, off = {}, 0
    for name, dt, n, shape in spec:
        nb = n * (8 if dt == torch.int64 else 4)
        o[name] = raw[off:off + nb].view(dt)[: shape[0] * (4 if len(shape) == 2 else 1)].view(shape)
        off += nb
    wsb = lib().ore_detect_workspace_bytes(L, pre_topk)
    if ws is None:
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    d.n_levels, d.head_ld = L, heads[0].shape[-1]
    for l, h in enumerate(heads):
        _f32(h)
        assert h.dim() == 3 and h.shape[-1] == d.head_ld
        d.head[l], d.H[l], d.W[l], d.stride[l] = _ptr(h), h.shape[0], h.shape[1], strides[l]
    d.score_thresh, d.pre_topk, d.nms_thresh, d.post_topk = score_thresh, pre_topk, nms_thresh, post_topk
    for k in ("pre_boxes", "pre_scores", "pre_loc", "pre_level", "keep_idx", "counts", "out_boxes", "out_scores"):
        setattr(d, k, _ptr(o[k]))
    d.workspace, d.workspace_bytes = _ptr(ws), wsb
    o["_ws"] = ws
    return o


def detect(heads: Sequence[torch.Tensor], strides: Sequence[int], score_thresh: float, pre_topk: int, nms_thresh: float,
           post_topk: int) -> Dict[str, torch.Tensor]:
    """heads[l]: [H,W,ld>=8] fp32: channels 0..3 reg (after Scale+ReLU), 4 = heatmap logit.  No host sync inside."""
    d = DetectDesc()
    o = _detect_desc(heads, strides, score_thresh, pre_topk, nms_thresh, post_topk, d)
    _chk(lib().ore_detect_fwd(C.byref(d), _stream()), "ore_detect_fwd")
    return o


def detect_batch(heads_per_image: Sequence[Sequence[torch.Tensor]], strides: Sequence[int], score_thresh: float, pre_topk: int,
                 nms_thresh: float, post_topk: int) -> List[Dict[str, torch.Tensor]]:
    """`detect` for B independent images (ore_detect_batch_fwd: the greedy scans of up to 16 images share one launch).  Per image the
    outputs of `detect`, bit for bit.  No host sync inside."""
    B = len(heads_per_image)
    ds = (DetectDesc * B)()
    # the outputs and workspaces of all images from two allocations (ONE zero fill instead of B)
    L = len(heads_per_image[0])
    _, raw_bytes = _detect_spec(L, pre_topk)
    raw_bytes = (raw_bytes + 255) // 256 * 256
    wsb = (int(lib().ore_detect_workspace_bytes(L, pre_topk)) + 255) // 256 * 256
    dev = heads_per_image[0][0].device
    raw = torch.zeros(B, raw_bytes, dtype=torch.uint8, device=dev)
    ws = torch.empty(B, wsb, dtype=torch.uint8, device=dev)
    outs = [_detect_desc(heads_per_image[b], strides, score_thresh, pre_topk, nms_thresh, post_topk, ds[b], raw[b], ws[b]) for b in range(B)]
    _chk(lib().ore_detect_batch_fwd(ds, B, _stream()), "ore_detect_batch_fwd")
    return outs


def nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """torchvision.ops.nms semantics (stable order); returns int64 keep indices (this call syncs to read the count)."""
    n = scores.numel()
    dev = boxes.device
    keep = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib().ore_nms_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    b = boxes.float().contiguous() if n else torch.zeros(1, 4, device=dev)
    s = scores.float().contiguous() if n else torch.zeros(1, device=dev)
    _chk(lib().ore_nms_fwd(C.c_void_p(_ptr(b)), C.c_void_p(_ptr(s)), n, C.c_float(thr), C.c_void_p(_ptr(keep)),
                           C.c_void_p(_ptr(cnt)), C.c_void_p(_ptr(ws)), C.c_size_t(wsb), _stream()), "ore_nms_fwd")
    return keep[: int(cnt.item())]


def roi_align(feats: Sequence[torch.Tensor], boxes: torch.Tensor, strides: Sequence[int], pooled: int = 8,
              n_dev: Optional[torch.Tensor] = None, cap: Optional[int] = None, min_level: int = 3) -> torch.Tensor:
    """feats[l]: [1,H,W,C] NHWC (or a channel-slice view described by (tensor, coff, C)); boxes [n,4] -> [cap, pooled*pooled*C]."""
    L = len(feats)
    n = boxes.shape[0]
    cap = cap or max(n, 1)
    Cc = feats[0].shape[-1]
    out = torch.empty(cap, pooled * pooled * Cc, device=boxes.device, dtype=torch.float32)
    ptrs = (C.c_void_p * L)(*[_ptr(_f32(f)) for f in feats])
    ld = (C.c_int32 * L)(*[f.shape[-1] for f in feats])
    coff = (C.c_int32 * L)(*[0] * L)
    Hs = (C.c_int32 * L)(*[f.shape[-3] for f in feats])
    Ws = (C.c_int32 * L)(*[f.shape[-2] for f in feats])
    sc = (C.c_float * L)(*[1.0 / s for s in strides])
    b = boxes.float().contiguous() if n else torch.zeros(1, 4, device=boxes.device)
    _chk(lib().ore_roi_align_fwd(ptrs, ld, coff, Hs, Ws, sc, L, min_level, Cc, pooled, C.c_void_p(_ptr(b)),
                                 C.c_void_p(_ptr(n_dev)), n, cap, C.c_void_p(_ptr(out)), _stream()), "ore_roi_align_fwd")
    return out


def roi_align_batched(feats: Sequence[torch.Tensor], boxes: torch.Tensor, box_image: torch.Tensor, strides=(8, 16, 32), pooled: int = 8,
                      min_level: int = 3) -> torch.Tensor:
    """feats[l] [B,H,W,C]; box i pools image box_image[i] (int32) -> [n, pooled*pooled*C]."""
    L = len(feats)
    n = boxes.shape[0]
    Cc = feats[0].shape[-1]
    out = torch.empty(max(n, 1), pooled * pooled * Cc, device=boxes.device, dtype=torch.float32)
    ptrs = (C.c_void_p * L)(*[_ptr(_f32(f)) for f in feats])
    ld = (C.c_int32 * L)(*[f.shape[-1] for f in feats])
    coff = (C.c_int32 * L)(*[0] * L)
    Hs = (C.c_int32 * L)(*[f.shape[-3] for f in feats])
    Ws = (C.c_int32 * L)(*[f.shape[-2] for f in feats])
    sc = (C.c_float * L)(*[1.0 / s for s in strides])
    assert box_image.dtype == torch.int32 and box_image.numel() == n
    _chk(lib().ore_roi_align_batched_fwd(ptrs, ld, coff, Hs, Ws, sc, L, min_level, Cc, pooled, C.c_void_p(_ptr(_f32(boxes))),
                                         C.c_void_p(_ptr(box_image)), n, C.c_void_p(_ptr(out)), _stream()), "ore_roi_align_batched_fwd")
    return out[:n]


ROI_BWD_DETERMINISTIC = True          # False: the fp32-atomics form of rounds 1-4 (order-dependent in the last bits; A/B aid)
ROI_BWD_MODE = "tiled"                # the deterministic form: "tiled" (gather, one block per map tile, no atomics) | "fixed" (64-bit fixed-point atomics)


def roi_align_bwd(dout: torch.Tensor, feats_like: Sequence[torch.Tensor], boxes: torch.Tensor, strides=(8, 16, 32), min_level: int = 3,
                  pooled: int = 8, dfeats: Optional[Sequence[torch.Tensor]] = None, box_image: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """dout [n, pooled*pooled*C]; returns/accumulates into per-level gradient buffers shaped like feats_like ([H,W,ld] NHWC)."""
    L = len(feats_like)
    n = boxes.shape[0]
    tiled = ROI_BWD_DETERMINISTIC and ROI_BWD_MODE == "tiled" and pooled <= 16
    given = dfeats is not None
    if dfeats is None:
        # the tiled form writes every cell of every map itself: nothing to zero
        dfeats = [torch.empty_like(f, dtype=torch.float32) if tiled else torch.zeros_like(f) for f in feats_like]
    Cc = dfeats[0].shape[-1]
    assert dout.numel() == n * pooled * pooled * Cc
    if n == 0:
        if tiled and not given:
            for f in dfeats:
                f.zero_()
        return list(dfeats)
    ptrs = (C.c_void_p * L)(*[_ptr(_f32(f)) for f in dfeats])
    ld = (C.c_int32 * L)(*[f.shape[-1] for f in dfeats])
    coff = (C.c_int32 * L)(*[0] * L)
    Hs = (C.c_int32 * L)(*[f.shape[-3] for f in dfeats])
    Ws = (C.c_int32 * L)(*[f.shape[-2] for f in dfeats])
    sc = (C.c_float * L)(*[1.0 / s for s in strides])
    n_img = 1 if dfeats[0].dim() == 3 else dfeats[0].shape[0]
    assert box_image is not None or n_img == 1
    if tiled:
        # one block per (image, level, 16 x 16-cell tile, 32 channels) gathers the ROIs that reach its cells in index order: no atomics, no
        # scratch, bit-reproducible (ore_roi_align_bwd_tiled)
        _chk(lib().ore_roi_align_bwd_tiled(ptrs, ld, coff, Hs, Ws, sc, L, min_level, Cc, pooled, C.c_void_p(_ptr(_f32(boxes.float().contiguous()))),
                                           C.c_void_p(_ptr(box_image)), n, C.c_void_p(_ptr(_f32(dout))), n_img, int(given), _stream()),
             "ore_roi_align_bwd_tiled")
        return list(dfeats)
    # Two ROIs of one image may touch the same cell: their sum must not depend on the order the blocks run in -> fixed-point accumulation
    # (ore_roi_align_bwd_det, one zeroed int64 scratch for all levels).  With at most one ROI per image (the support crops: one box each)
    # every cell receives ONE add from the column kernel, fp32 atomics are order-free there and the scratch (8 bytes per map element) is
    # saved.
    if ROI_BWD_DETERMINISTIC and (n > n_img or box_image is None and n > 1):
        sizes = [n_img * f.shape[-3] * f.shape[-2] * Cc for f in dfeats]
        scratch = torch.zeros(sum(sizes), dtype=torch.int64, device=dout.device)
        offs = [sum(sizes[:i]) for i in range(L)]
        accs = (C.c_void_p * L)(*[_ptr(scratch) + 8 * o for o in offs])
        _chk(lib().ore_roi_align_bwd_det(ptrs, ld, coff, Hs, Ws, sc, L, min_level, Cc, pooled, C.c_void_p(_ptr(_f32(boxes.float().contiguous()))),
                                         C.c_void_p(_ptr(box_image)), n, C.c_void_p(_ptr(_f32(dout))), accs, n_img, _stream()), "ore_roi_align_bwd_det")
        return list(dfeats)
    _chk(lib().ore_roi_align_bwd(ptrs, ld, coff, Hs, Ws, sc, L, min_level, Cc, pooled, C.c_void_p(_ptr(_f32(boxes.float().contiguous()))),
                                 C.c_void_p(_ptr(box_image)), n, C.c_void_p(_ptr(_f32(dout))), _stream()), "ore_roi_align_bwd")
    return list(dfeats)


def roi_predict(h: torch.Tensor, cls_w, cls_b, box_w, box_b, boxes: torch.Tensor, reg_weights, image_hw, score_thresh: float,
                nms_thresh: float, topk: int, n_dev: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    cap, Cc = h.shape
    n = boxes.shape[0]
    dev = h.device
    o = {"boxes": torch.zeros(cap, 4, device=dev), "scores": torch.zeros(cap, device=dev),
         "src": torch.zeros(cap, dtype=torch.int64, device=dev), "count": torch.zeros(1, dtype=torch.int32, device=dev)}
    wsb = lib().ore_roi_predict_workspace_bytes(cap)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rw = (C.c_float * 4)(*reg_weights)
    b = boxes.float().contiguous() if n else torch.zeros(1, 4, device=dev)
    _chk(lib().ore_roi_predict_fwd(C.c_void_p(_ptr(_f32(h))), Cc, C.c_void_p(_ptr(_f32(cls_w))), C.c_void_p(_ptr(_f32(cls_b))),
                                   C.c_void_p(_ptr(_f32(box_w))), C.c_void_p(_ptr(_f32(box_b))), C.c_void_p(_ptr(b)),
                                   C.c_void_p(_ptr(n_dev)), n, cap, rw, C.c_float(image_hw[0]), C.c_float(image_hw[1]),
                                   C.c_float(score_thresh), C.c_float(nms_thresh), topk, C.c_void_p(_ptr(o["boxes"])),
                                   C.c_void_p(_ptr(o["scores"])), C.c_void_p(_ptr(o["src"])), C.c_void_p(_ptr(o["count"])),
                                   C.c_void_p(_ptr(ws)), C.c_size_t(wsb), _stream()), "ore_roi_predict_fwd")
    o["_ws"] = ws
    return o


def to_device(t: torch.Tensor, dev) -> torch.Tensor:
    """Host tensor -> device without a blocking pageable copy (a pageable hipMemcpy drains the stream first: one hidden host sync per
    small upload): staged through torch's cached pinned allocator and issued asynchronously.  Device tensors pass through."""
    dev = torch.device(dev)
    if t.is_cuda:
        return t if t.device == dev or dev.index is None else t.to(dev)
    if dev.type != "cuda":
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


def centernet_targets(gt_boxes: Sequence[torch.Tensor], shapes: Sequence[Tuple[int, int]], strides=(8, 16, 32),
                      soi=((0, 64), (48, 192), (128, 1000000)), hm_min_overlap: float = 0.8, min_radius: float = 4.0,
                      device=None, padded: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """CenterNet._get_ground_truth (ref:fewx/modeling/fsod/fsod_rpn.py:803-901) on device.  gt_boxes: per image [N_i,4]; or
    `padded` = (boxes [B,G,4] fp32, counts [B] int32) already on the device (fixed shapes: a captured training step)."""
    dev = torch.device(device or "cuda")
    L = len(strides)
    if padded is not None:
        gt, cnt = padded
        assert gt.is_cuda and gt.dtype == torch.float32 and gt.is_contiguous() and cnt.dtype == torch.int32 and cnt.numel() == gt.shape[0]
        B, max_n = int(gt.shape[0]), int(gt.shape[1])
    else:
        B = len(gt_boxes)
        ns = [int(b.shape[0]) for b in gt_boxes]
        max_n = max(1, max(ns))
        # no blocking copy in either direction: device boxes are padded on the device, host boxes go up through pinned memory
        if any(b.is_cuda for b in gt_boxes):
            gt = torch.zeros(B, max_n, 4, dtype=torch.float32, device=dev)
            for i, b in enumerate(gt_boxes):
                if ns[i]:
                    gt[i, :ns[i]] = to_device(b.detach().float(), dev)
        else:
            gt = torch.zeros(B, max_n, 4, dtype=torch.float32)
            for i, b in enumerate(gt_boxes):
                gt[i, :ns[i]] = b.detach().float()
            gt = to_device(gt, dev)
        cnt = to_device(torch.tensor(ns, dtype=torch.int32), dev)
    rows = sum(B * h * w for h, w in shapes)
    o = {"reg_targets": torch.empty(rows, 4, device=dev), "hm_targets": torch.empty(rows, device=dev),
         "pos_inds": torch.zeros(B * max_n * L, dtype=torch.int64, device=dev), "pos_count": torch.zeros(1, dtype=torch.int32, device=dev)}
    Hs = (C.c_int32 * L)(*[h for h, _ in shapes]); Ws = (C.c_int32 * L)(*[w for _, w in shapes]); St = (C.c_int32 * L)(*strides)
    so = (C.c_float * (2 * L))(*[float(v) for r in soi for v in r])
    _chk(lib().ore_centernet_targets_fwd(L, Hs, Ws, St, B, C.c_void_p(_ptr(gt)), C.c_void_p(_ptr(cnt)), max_n, so,
                                         C.c_float(hm_min_overlap), C.c_float(min_radius), C.c_void_p(_ptr(o["reg_targets"])),
                                         C.c_void_p(_ptr(o["hm_targets"])), C.c_void_p(_ptr(o["pos_inds"])),
                                         C.c_void_p(_ptr(o["pos_count"])), _stream()), "ore_centernet_targets_fwd")
    o["_keep"] = (gt, cnt)
    return o


def centernet_loss_sums(head: torch.Tensor, reg_targets: torch.Tensor, hm_targets: torch.Tensor, pos_inds: torch.Tensor,
                        pos_count: torch.Tensor, gamma: float = 2.0, beta: float = 4.0, sigmoid_clamp: float = 1e-4,
                        ignore_high_fp: float = 0.85) -> torch.Tensor:
    """[giou sum, #reg rows, pos focal sum, neg focal sum] of CenterNet.losses (ref:fewx/modeling/fsod/fsod_rpn.py:702-779).
    head [rows, ld>=5]: cols 0..3 ltrb prediction, col 4 heatmap logit."""
    rows, ld = head.shape
    out = torch.zeros(4, device=head.device)
    ws = torch.empty(4 * 256, device=head.device)
    _chk(lib().ore_centernet_losses_fwd(C.c_void_p(_ptr(_f32(head))), ld, C.c_void_p(_ptr(_f32(reg_targets))),
                                        C.c_void_p(_ptr(_f32(hm_targets))), rows, C.c_void_p(_ptr(pos_inds)),
                                        C.c_void_p(_ptr(pos_count)), C.c_float(gamma), C.c_float(beta), C.c_float(sigmoid_clamp),
                                        C.c_float(ignore_high_fp), C.c_void_p(_ptr(out)), C.c_void_p(_ptr(ws)), _stream()),
         "ore_centernet_losses_fwd")
    return out


def centernet_loss_grad(head: torch.Tensor, reg_targets: torch.Tensor, hm_targets: torch.Tensor, pos_inds: torch.Tensor,
                        pos_count: torch.Tensor, coef3: torch.Tensor, gamma: float = 2.0, beta: float = 4.0, sigmoid_clamp: float = 1e-4,
                        ignore_high_fp: float = 0.85) -> torch.Tensor:
    """d(losses)/d(head[:, :5]) for coef3 = [reg_w/reg_norm, pos_w*alpha/num_pos_avg, neg_w*(1-alpha)/num_pos_avg] (device)."""
    rows, ld = head.shape
    dhead = torch.zeros(rows, ld, device=head.device, dtype=torch.float32)
    _chk(lib().ore_centernet_losses_bwd(C.c_void_p(_ptr(_f32(head))), ld, C.c_void_p(_ptr(_f32(reg_targets))),
                                        C.c_void_p(_ptr(_f32(hm_targets))), rows, C.c_void_p(_ptr(pos_inds)), C.c_void_p(_ptr(pos_count)),
                                        int(pos_inds.numel()), C.c_float(gamma), C.c_float(beta), C.c_float(sigmoid_clamp),
                                        C.c_float(ignore_high_fp), C.c_void_p(_ptr(_f32(coef3))), C.c_void_p(_ptr(dhead)), ld, _stream()),
         "ore_centernet_losses_bwd")
    return dhead


def sample_rois(prop: torch.Tensor, prop_n: torch.Tensor, gtp: torch.Tensor, gt_n: torch.Tensor, keys: torch.Tensor, R: int, P: int,
                iou_thr: float, append_gt: bool = True):
    """label_and_sample_proposals of B images in one launch (ore_sample_rois_fwd).  prop [B,cap,4], prop_n [B] int64, gtp [B,G,4],
    gt_n [B] int64, keys [B, cap (+G)] fp32 -> boxes [B,R,4], labels [B,R] int64, matched gt [B,R,4], valid [B,R] bool."""
    B, cap, _ = prop.shape
    G = gtp.shape[1]
    N = cap + (G if append_gt else 0)
    dev = prop.device
    assert keys.shape == (B, N) and keys.dtype == torch.float32 and keys.is_contiguous()
    assert prop_n.dtype == torch.int64 and gt_n.dtype == torch.int64 and prop.is_contiguous() and gtp.is_contiguous()
    boxes = torch.empty(B, R, 4, device=dev, dtype=torch.float32)
    gt = torch.empty(B, R, 4, device=dev, dtype=torch.float32)
    labels = torch.empty(B, R, device=dev, dtype=torch.int64)
    valid = torch.empty(B, R, device=dev, dtype=torch.bool)
    _chk(lib().ore_sample_rois_fwd(C.c_void_p(_ptr(_f32(prop))), C.c_void_p(_ptr(prop_n)), C.c_void_p(_ptr(_f32(gtp))), C.c_void_p(_ptr(gt_n)),
                                   C.c_void_p(_ptr(keys)), B, cap, G, int(bool(append_gt)), R, P, C.c_float(iou_thr), C.c_void_p(_ptr(boxes)),
                                   C.c_void_p(_ptr(labels)), C.c_void_p(_ptr(gt)), C.c_void_p(_ptr(valid)), _stream()), "ore_sample_rois_fwd")
    return boxes, labels, gt, valid


def roi_losses(scores: torch.Tensor, deltas: torch.Tensor, boxes: torch.Tensor, gt: torch.Tensor, labels: torch.Tensor, valid: torch.Tensor,
               B: int, R: int, reg_weights) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The second stage's two losses and their gradients in one launch (ore_roi_losses_fwd): scores [B*R,2], deltas / boxes / gt [B*R,4],
    labels [B*R] int64 (0 fg / 1 bg), valid [B*R] bool -> (losses [2] = (cls, box), dscores [B*R,2], ddeltas [B*R,4])."""
    RT = B * R
    dev = scores.device
    assert scores.shape == (RT, 2) and deltas.shape == (RT, 4) and boxes.shape == (RT, 4) and gt.shape == (RT, 4)
    assert labels.dtype == torch.int64 and labels.numel() == RT and valid.dtype == torch.bool and valid.numel() == RT
    out = torch.empty(2, device=dev, dtype=torch.float32)
    ds = torch.empty(RT, 2, device=dev, dtype=torch.float32)
    dd = torch.empty(RT, 4, device=dev, dtype=torch.float32)
    rw = (C.c_float * 4)(*[float(v) for v in reg_weights])
    _chk(lib().ore_roi_losses_fwd(C.c_void_p(_ptr(_f32(scores))), C.c_void_p(_ptr(_f32(deltas))), C.c_void_p(_ptr(_f32(boxes))),
                                  C.c_void_p(_ptr(_f32(gt))), C.c_void_p(_ptr(labels.contiguous())), C.c_void_p(_ptr(valid.contiguous())), B, R, rw,
                                  C.c_void_p(_ptr(out)), C.c_void_p(_ptr(ds)), C.c_void_p(_ptr(dd)), _stream()), "ore_roi_losses_fwd")
    return out, ds, dd


def sgd_step(params: torch.Tensor, grads: torch.Tensor, momentum_buf: torch.Tensor, chunk_lr: torch.Tensor, chunk_wd: torch.Tensor,
             lr_scale: float = 1.0, momentum: float = 0.9, clip_value: float = 1.0, grad_scale: float = 1.0,
             lr_scale_dev: Optional[torch.Tensor] = None) -> None:
    """In-place clip + SGD over a flat bucket of 256-float chunks (see ore_sgd_step_fwd)."""
    n = params.numel()
    assert n % 256 == 0 and grads.numel() == n and momentum_buf.numel() == n and chunk_lr.numel() == n // 256 == chunk_wd.numel()
    for t in (params, grads, momentum_buf, chunk_lr, chunk_wd):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    _chk(lib().ore_sgd_step_fwd(C.c_void_p(_ptr(params)), C.c_void_p(_ptr(grads)), C.c_void_p(_ptr(momentum_buf)),
                                C.c_int64(n // 256), C.c_void_p(_ptr(chunk_lr)), C.c_void_p(_ptr(chunk_wd)),
                                C.c_void_p(_ptr(lr_scale_dev)), C.c_float(lr_scale), C.c_float(momentum), C.c_float(clip_value),
                                C.c_float(grad_scale), _stream()), "ore_sgd_step_fwd")


def pack_conv_weight_dev(w_oihw: torch.Tensor, dgrad: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Device-side repack of OIHW master weights (runs every optimizer step): forward layout [Cout16][tap][Cin], or the
    data-gradient layout [Cin16][flipped tap][Cout16]."""
    w = _f32(w_oihw.detach())
    co, ci, kh, kw = w.shape
    co16, ci16 = (co + 15) // 16 * 16, (ci + 15) // 16 * 16
    n = ci16 * kh * kw * co16 if dgrad else co16 * kh * kw * ci
    if out is None:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    assert out.numel() == n
    _chk(lib().ore_pack_conv_weight_fwd(C.c_void_p(_ptr(w)), co, ci, kh, kw, int(dgrad), C.c_void_p(_ptr(out)), _stream()),
         "ore_pack_conv_weight_fwd")
    return out


class PackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("dgrad", C.c_int32), ("reserved", C.c_int32), ("first", C.c_int64)]


_PACK_TABLES: Dict[tuple, torch.Tensor] = {}


def pack_conv_weights_multi(jobs) -> None:
    """jobs: [(w_oihw contiguous fp32 device tensor, dgrad, out)] -> every `out` holds pack_conv_weight_dev(w, dgrad): ONE launch
    (ore_pack_conv_weights_multi_fwd).  The device-side job table is cached by its contents (pointers and shapes): a training loop hands
    the same list every step, and a captured step replays the launch on the same table."""
    if not jobs:
        return
    assert len(jobs) <= 256
    dev = jobs[0][0].device
    rows, first = [], 0
    for w, dgrad, out in jobs:
        co, ci, kh, kw = w.shape
        co16, ci16 = (co + 15) // 16 * 16, (ci + 15) // 16 * 16
        n = ci16 * kh * kw * co16 if dgrad else co16 * kh * kw * ci
        assert w.is_contiguous() and w.dtype == torch.float32 and out.numel() == n and (dgrad or ci % 16 == 0)
        rows.append((_ptr(w), _ptr(out), co, ci, kh, kw, int(bool(dgrad)), first))
        first += n
    key = (str(dev), tuple(rows))
    tab = _PACK_TABLES.get(key)
    if tab is None:
        arr = (PackJob * len(rows))()
        for i, (src, dst, co, ci, kh, kw, dg, f) in enumerate(rows):
            arr[i] = PackJob(src, dst, co, ci, kh, kw, dg, 0, f)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        if len(_PACK_TABLES) > 64:
            _PACK_TABLES.clear()
        tab = _PACK_TABLES[key] = host.to(dev)
    _chk(lib().ore_pack_conv_weights_multi_fwd(C.c_void_p(_ptr(tab)), len(rows), C.c_int64(first), _stream()), "ore_pack_conv_weights_multi_fwd")


_WGWS: Dict[str, torch.Tensor] = {}


def _wgrad_ws(device, n: int) -> torch.Tensor:
    k = str(device)
    if k not in _WGWS or _WGWS[k].numel() < n:
        _WGWS[k] = torch.empty(max(n, 16 << 20), device=device, dtype=torch.float32)
    return _WGWS[k]


def conv2d_wgrad(x: torch.Tensor, dz: torch.Tensor, k: int, *, x_coff: int = 0, Cin: Optional[int] = None, dz_coff: int = 0,
                 Cout: Optional[int] = None, out: Optional[torch.Tensor] = None, beta: float = 0.0, want_bias: bool = False,
                 db_out: Optional[torch.Tensor] = None, beta_b: float = 0.0):
    """x [B,H,W,x_ld], dz [B,H,W,dz_ld] NHWC -> dW [Cout,Cin,k,k] (OIHW); stride 1, pad k//2.  want_bias: also the bias gradient
    (column sums of dz) from the same launch -> (dW, db).  out / beta, db_out / beta_b: dW = beta * out + ..., db = beta_b * db_out + ...
    written in place (a parameter's gradient accumulated by the launch itself)."""
    _f32(x); _f32(dz)
    B, H, W, xld = x.shape
    dld = dz.shape[-1]
    assert dz.shape[:3] == (B, H, W)
    Cin = Cin if Cin is not None else xld - x_coff
    Cout = Cout if Cout is not None else dld - dz_coff
    if out is None:
        out = torch.empty(Cout, Cin, k, k, device=x.device, dtype=torch.float32)
        beta = 0.0
    assert out.shape == (Cout, Cin, k, k) and out.is_contiguous()
    n = lib().ore_conv_wgrad_workspace_floats(B * H * W, Cin, Cout, k, k)
    ws = _wgrad_ws(x.device, n)
    if want_bias and db_out is not None:
        assert db_out.shape == (Cout,) and db_out.is_contiguous() and db_out.dtype == torch.float32
        db = db_out
    else:
        db, beta_b = (torch.empty(Cout, device=x.device, dtype=torch.float32) if want_bias else None), 0.0
    _chk(lib().ore_conv2d_wgrad_bias_fwd(C.c_void_p(_ptr(x)), xld, x_coff, C.c_void_p(_ptr(dz)), dld, dz_coff, B, H, W, Cin, Cout, k, k,
                                         k // 2, C.c_void_p(_ptr(out)), C.c_float(beta), C.c_void_p(_ptr(db)), C.c_float(beta_b),
                                         C.c_void_p(_ptr(ws)), C.c_size_t(ws.numel()), _stream()), "ore_conv2d_wgrad_bias_fwd")
    return (out, db) if want_bias else out


def relu_affine_bwd(dy: torch.Tensor, y: torch.Tensor, scale: Optional[torch.Tensor] = None, *, dy_coff: int = 0, y_coff: int = 0,
                    Cc: Optional[int] = None, out: Optional[torch.Tensor] = None, out_coff: int = 0) -> torch.Tensor:
    """dZ = dY * (Y > 0) * scale over channel slices of NHWC/row-major buffers (last dim = ld)."""
    _f32(dy); _f32(y)
    Cc = Cc if Cc is not None else y.shape[-1] - y_coff
    rows = y.numel() // y.shape[-1]
    if out is None:
        out = torch.empty(*y.shape[:-1], Cc, device=y.device, dtype=torch.float32)
    _chk(lib().ore_relu_affine_bwd(C.c_void_p(_ptr(dy)), dy.shape[-1], dy_coff, C.c_void_p(_ptr(y)), y.shape[-1], y_coff,
                                   C.c_void_p(_ptr(scale)), C.c_int64(rows), Cc, C.c_void_p(_ptr(_f32(out))), out.shape[-1], out_coff,
                                   _stream()), "ore_relu_affine_bwd")
    return out


def colsum(x: torch.Tensor, *, coff: int = 0, Cc: Optional[int] = None, out: Optional[torch.Tensor] = None, beta: float = 0.0) -> torch.Tensor:
    """Bias gradient: sums over all leading dims of x[..., coff:coff+Cc]."""
    _f32(x)
    ld = x.shape[-1]
    Cc = Cc if Cc is not None else ld - coff
    rows = x.numel() // ld
    if out is None:
        out = torch.empty(Cc, device=x.device, dtype=torch.float32)
        beta = 0.0
    ws = _wgrad_ws(x.device, ((rows + 63) // 64) * Cc)
    _chk(lib().ore_colsum_fwd(C.c_void_p(_ptr(x)), ld, coff, C.c_int64(rows), Cc, C.c_float(beta), C.c_void_p(_ptr(out)),
                              C.c_void_p(_ptr(ws)), C.c_size_t(ws.numel()), _stream()), "ore_colsum_fwd")
    return out


def colsum_segments(x: torch.Tensor, segments: int) -> torch.Tensor:
    """x [rows, C] made of `segments` equal runs of rows (the images of a batch) -> [segments, C] per-run column sums; each run's sums
    are bitwise those of `colsum` on that run alone (ore_colsum_segments_fwd)."""
    _f32(x)
    rows, Cc = x.shape
    assert rows % segments == 0
    rps = rows // segments
    out = torch.empty(segments, Cc, device=x.device, dtype=torch.float32)
    ws = _wgrad_ws(x.device, segments * ((rps + 255) // 256) * Cc)
    _chk(lib().ore_colsum_segments_fwd(C.c_void_p(_ptr(x)), Cc, 0, segments, C.c_int64(rps), Cc, C.c_float(0.0), C.c_void_p(_ptr(out)),
                                       C.c_void_p(_ptr(ws)), C.c_size_t(ws.numel()), _stream()), "ore_colsum_segments_fwd")
    return out


def correlation_train_fwd(q: torch.Tensor, k11: torch.Tensor, k13: torch.Tensor, k31: torch.Tensor):
    """q [B,H,W,C]; k11 [C], k13 / k31 [C,3] shared by the B images, or [B,C] / [B,C,3] = one support kernel set per image
    -> (cat [B,H,W,2C] = [attn | q], t, u)."""
    _f32(q)
    B, H, W, Cc = q.shape
    per_image = int(k11.dim() == 2)
    assert k11.numel() == (B if per_image else 1) * Cc and k13.numel() == k31.numel() == 3 * k11.numel()
    cat = torch.empty(B, H, W, 2 * Cc, device=q.device, dtype=torch.float32)
    t = torch.empty(B, H, W, Cc, device=q.device, dtype=torch.float32)
    u = torch.empty_like(t)
    _chk(lib().ore_correlation_train_fwd(C.c_void_p(_ptr(q)), Cc, 0, B, H, W, Cc, C.c_void_p(_ptr(_f32(k11))), C.c_void_p(_ptr(_f32(k13))),
                                         C.c_void_p(_ptr(_f32(k31))), per_image, C.c_void_p(_ptr(cat)), C.c_void_p(_ptr(t)),
                                         C.c_void_p(_ptr(u)), _stream()), "ore_correlation_train_fwd")
    return cat, t, u


def correlation_train_bwd(q, k11, k13, k31, dcat, t, u):
    """-> dq [B,H,W,C], dk11, dk13, dk31 shaped like k11 / k13 / k31."""
    B, H, W, Cc = q.shape
    rows = B * H * W
    per_image = int(k11.dim() == 2)
    S = B if per_image else 1
    dq = torch.empty_like(q)
    dk = torch.empty(S, 7, Cc, device=q.device, dtype=torch.float32)
    ws = _wgrad_ws(q.device, rows * Cc * 8 + S * ((rows // S + 255) // 256) * 7 * Cc)
    _chk(lib().ore_correlation_train_bwd(C.c_void_p(_ptr(_f32(q))), Cc, 0, B, H, W, Cc, C.c_void_p(_ptr(_f32(k11))),
                                         C.c_void_p(_ptr(_f32(k13))), C.c_void_p(_ptr(_f32(k31))), per_image, C.c_void_p(_ptr(_f32(dcat))),
                                         C.c_void_p(_ptr(t)), C.c_void_p(_ptr(u)), C.c_void_p(_ptr(dq)), C.c_void_p(_ptr(dk)),
                                         C.c_void_p(_ptr(ws)), C.c_size_t(ws.numel()), _stream()), "ore_correlation_train_bwd")
    d11, d13, d31 = dk[:, 0], dk[:, 1:4].transpose(1, 2).contiguous(), dk[:, 4:7].transpose(1, 2).contiguous()
    if not per_image:
        d11, d13, d31 = d11[0], d13[0], d31[0]
    return dq, d11.contiguous(), d13, d31


def groupnorm_apply(x: torch.Tensor, rstd_c: torch.Tensor, shift_c: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, relu: bool):
    """x [B,H,W,C], rstd_c / shift_c [B,C] (per image) -> relu?((x*rstd_c + shift_c)*gamma + beta)."""
    _f32(x)
    B, Cc = x.shape[0], x.shape[-1]
    rpi = x.numel() // (Cc * B)
    assert rstd_c.numel() == B * Cc and shift_c.numel() == B * Cc
    y = torch.empty_like(x)
    _chk(lib().ore_groupnorm_apply_fwd(C.c_void_p(_ptr(x)), Cc, 0, B, C.c_int64(rpi), Cc, C.c_void_p(_ptr(_f32(rstd_c))),
                                       C.c_void_p(_ptr(_f32(shift_c))), C.c_void_p(_ptr(_f32(gamma))), C.c_void_p(_ptr(_f32(beta))), int(relu),
                                       C.c_void_p(_ptr(y)), _stream()), "ore_groupnorm_apply_fwd")
    return y


def groupnorm_bwd(dy, y, x, groups: int, rstd_c, shift_c, gamma, relu: bool):
    """x [B,H,W,C], rstd_c / shift_c [B,C] -> dx like x, dbeta [B,C], dgamma [B,C] (per image; sum over B for the parameters)."""
    _f32(dy); _f32(y); _f32(x)
    B, Cc = x.shape[0], x.shape[-1]
    rpi = x.numel() // (Cc * B)
    dx = torch.empty_like(x)
    sums = torch.empty(B, 2 * Cc, device=x.device, dtype=torch.float32)
    ws = _wgrad_ws(x.device, B * rpi * 2 * Cc + B * ((rpi + 255) // 256) * 2 * Cc)
    _chk(lib().ore_groupnorm_bwd(C.c_void_p(_ptr(dy)), C.c_void_p(_ptr(y)), C.c_void_p(_ptr(x)), Cc, 0, B, C.c_int64(rpi), Cc, groups,
                                 C.c_void_p(_ptr(_f32(rstd_c))), C.c_void_p(_ptr(_f32(shift_c))), C.c_void_p(_ptr(_f32(gamma))), int(relu),
                                 C.c_void_p(_ptr(dx)), C.c_void_p(_ptr(sums)), C.c_void_p(_ptr(ws)), C.c_size_t(ws.numel()), _stream()),
         "ore_groupnorm_bwd")
    return dx, sums[:, :Cc], sums[:, Cc:]


def prod_colsum(p: torch.Tensor, q: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
    """p (, q) [B,H,W,C] -> [B,C]: per-image sum over pixels of p (* q), times scale."""
    _f32(p)
    B, Cc = p.shape[0], p.shape[-1]
    rows = p.numel() // (B * Cc)
    out = torch.empty(B, Cc, device=p.device, dtype=torch.float32)
    ws = _wgrad_ws(p.device, p.numel())
    _chk(lib().ore_prod_colsum_fwd(C.c_void_p(_ptr(p)), C.c_void_p(_ptr(_f32(q)) if q is not None else None), B, rows, Cc, C.c_float(scale),
                                   C.c_void_p(_ptr(out)), C.c_void_p(_ptr(ws)), C.c_size_t(ws.numel()), _stream()), "ore_prod_colsum_fwd")
    return out


def scale_add_channels(x: torch.Tensor, scale_bc: torch.Tensor, add_bc: Optional[torch.Tensor] = None) -> torch.Tensor:
    _f32(x)
    B, Cc = x.shape[0], x.shape[-1]
    rows = x.numel() // (B * Cc)
    out = torch.empty_like(x)
    _chk(lib().ore_scale_add_channels_fwd(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(_f32(scale_bc))),
                                          C.c_void_p(_ptr(_f32(add_bc)) if add_bc is not None else None), B, rows, Cc, C.c_void_p(_ptr(out)),
                                          _stream()), "ore_scale_add_channels_fwd")
    return out


def adaptive_avgpool_nhwc(x: torch.Tensor, OH: int, OW: int, grad_of: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """F.adaptive_avg_pool2d on an NHWC map: x [B,H,W,C] -> [B,OH,OW,C].  grad_of=(H, W): x is the gradient w.r.t. the pooled map
    [B,OH,OW,C] and the result is the gradient w.r.t. the [B,H,W,C] input."""
    _f32(x)
    B, Cc = x.shape[0], x.shape[-1]
    if grad_of is None:
        H, W = x.shape[1:3]
        y = torch.empty(B, OH, OW, Cc, device=x.device, dtype=torch.float32)
        _chk(lib().ore_adaptive_avgpool_nhwc_fwd(C.c_void_p(_ptr(x)), B, H, W, Cc, OH, OW, C.c_void_p(_ptr(y)), _stream()), "ore_adaptive_avgpool_nhwc_fwd")
        return y
    H, W = grad_of
    assert tuple(x.shape[1:3]) == (OH, OW)
    dx = torch.empty(B, H, W, Cc, device=x.device, dtype=torch.float32)
    _chk(lib().ore_adaptive_avgpool_nhwc_bwd(C.c_void_p(_ptr(x)), B, H, W, Cc, OH, OW, C.c_void_p(_ptr(dx)), _stream()), "ore_adaptive_avgpool_nhwc_bwd")
    return dx


def group_mean(x: torch.Tensor, groups: int, backward: bool = False, members: int = 0) -> torch.Tensor:
    """x [groups*N, ...] -> [groups, ...]: mean over the N consecutive members of each group (the prototype over an image's shots);
    backward=True: x [groups, ...] is the gradient of that mean -> [groups*members, ...]."""
    _f32(x)
    if not backward:
        N = x.shape[0] // groups
        assert N * groups == x.shape[0]
        M = x.numel() // x.shape[0]
        y = torch.empty(groups, *x.shape[1:], device=x.device, dtype=torch.float32)
        _chk(lib().ore_group_mean_fwd(C.c_void_p(_ptr(x)), groups, N, C.c_int64(M), C.c_void_p(_ptr(y)), _stream()), "ore_group_mean_fwd")
        return y
    M = x.numel() // groups
    dx = torch.empty(groups * members, *x.shape[1:], device=x.device, dtype=torch.float32)
    _chk(lib().ore_group_mean_bwd(C.c_void_p(_ptr(x)), groups, members, C.c_int64(M), C.c_void_p(_ptr(dx)), _stream()), "ore_group_mean_bwd")
    return dx


def sm_permute(x: torch.Tensor, B: int, H: int, W: int, G: int, S: int, axis: str, inverse: bool, out: Optional[torch.Tensor] = None,
               accumulate: bool = False) -> torch.Tensor:
    """The SM_Block's mixing layouts as one coalesced pass (ore_granule_transpose_fwd).  axis 'h': [B,H,W,G,S] <-> [B,G,W,H,S]
    (= .permute(0,3,2,1,4)); axis 'w': [B,H,W,G,S] <-> [B,G,H,W,S] (= .permute(0,3,1,2,4) and back).  x contiguous with B*H*W*G*S elements;
    returns a contiguous tensor in the other layout (forward: the mixing layout, inverse: NHWC).  out + accumulate: out += the result."""
    _f32(x)
    assert x.numel() == B * H * W * G * S and axis in ("h", "w")
    Cc, img = G * S, H * W * G * S
    if axis == "h":
        nb2, mix = W, (B, G, W, H, S)
        fwd = (H, G, img, Cc, W * Cc, img, H * S, W * H * S)          # A, Bc, in_b1, in_b2, in_rs, out_b1, out_b2, out_rs
        inv = (G, H, img, H * S, W * H * S, img, Cc, W * Cc)
    else:
        nb2, mix = H, (B, G, H, W, S)
        fwd = (W, G, img, W * Cc, Cc, img, W * S, H * W * S)
        inv = (G, W, img, W * S, H * W * S, img, W * Cc, Cc)
    A_, Bc, ib1, ib2, irs, ob1, ob2, ors = inv if inverse else fwd
    if out is None:
        assert not accumulate
        out = torch.empty((B, H, W, G * S) if inverse else mix, device=x.device, dtype=torch.float32)
    else:
        _f32(out)
        assert out.numel() == x.numel()
    _chk(lib().ore_granule_transpose_fwd(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(out)), B, nb2, A_, Bc, S, C.c_int64(ib1), C.c_int64(ib2),
                                         C.c_int64(irs), C.c_int64(ob1), C.c_int64(ob2), C.c_int64(ors), int(accumulate), _stream()),
         "ore_granule_transpose_fwd")
    return out


def combine2(w: torch.Tensor, h: torch.Tensor, a0: torch.Tensor, a1: torch.Tensor) -> torch.Tensor:
    """y = w * a0[b, c] + h * a1[b, c];  w, h [B, ..., C], a0, a1 [B, C]."""
    _f32(w); _f32(h)
    B, Cc = w.shape[0], w.shape[-1]
    y = torch.empty_like(w)
    _chk(lib().ore_combine2_fwd(C.c_void_p(_ptr(w)), C.c_void_p(_ptr(h)), C.c_void_p(_ptr(_f32(a0))), C.c_void_p(_ptr(_f32(a1))), B,
                                w.numel() // (B * Cc), Cc, C.c_void_p(_ptr(y)), _stream()), "ore_combine2_fwd")
    return y


def combine2_bwd(dy: torch.Tensor, a0: torch.Tensor, a1: torch.Tensor, add_bc: Optional[torch.Tensor] = None):
    """(dw, dh) = (dy * a0 + v, dy * a1 + v) with the optional per-(image, channel) constant v."""
    _f32(dy)
    B, Cc = dy.shape[0], dy.shape[-1]
    dw, dh = torch.empty_like(dy), torch.empty_like(dy)
    _chk(lib().ore_combine2_bwd(C.c_void_p(_ptr(dy)), C.c_void_p(_ptr(_f32(a0))), C.c_void_p(_ptr(_f32(a1))),
                                C.c_void_p(_ptr(_f32(add_bc)) if add_bc is not None else None), B, dy.numel() // (B * Cc), Cc,
                                C.c_void_p(_ptr(dw)), C.c_void_p(_ptr(dh)), _stream()), "ore_combine2_bwd")
    return dw, dh


def maxpool3x3s2_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _f32(x); _f32(dy)
    B, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    _chk(lib().ore_maxpool3x3s2_bwd(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(dy)), B, H, W, Cc, C.c_void_p(_ptr(dx)), _stream()), "ore_maxpool3x3s2_bwd")
    return dx


def sumpool2x2(x: torch.Tensor) -> torch.Tensor:
    _f32(x)
    B, H, W, Cc = x.shape
    out = torch.empty(B, (H + 1) // 2, (W + 1) // 2, Cc, device=x.device, dtype=torch.float32)
    _chk(lib().ore_sumpool2x2_fwd(C.c_void_p(_ptr(x)), Cc, B, H, W, Cc, C.c_void_p(_ptr(out)), _stream()), "ore_sumpool2x2_fwd")
    return out


def flop_counter(reset: bool = False) -> Tuple[float, int]:
    """(algorithmic FLOPs, calls) of the per-op conv entry points since the last reset (ore_flop_counter_read)."""
    v, n = C.c_double(0.0), C.c_int64(0)
    _chk(lib().ore_flop_counter_read(C.byref(v), C.byref(n), int(reset)), "ore_flop_counter_read")
    return float(v.value), int(n.value)


def event_pair_overhead_us(reps: int = 200) -> float:
    """Median (event, empty launch, event) time on the current stream (see ore_event_pair_overhead_us)."""
    v = C.c_double(0.0)
    _chk(lib().ore_event_pair_overhead_us(_stream(), reps, C.byref(v)), "ore_event_pair_overhead_us")
    return float(v.value)


def compose_roi_head(sd, support_8: torch.Tensor, prefix: str = "roi_heads."):
    """The eval second stage between ROIAlign and fc1's ReLU is linear in the pooled features:
         a = conv3(cat(x, s)) + cat(conv1(x), conv2(s));  h = relu(fc1(flatten(a)))        (fsod_roi_heads.py:500-520)
    so it folds, once per (weights, support set), into h = relu(W' x_flat + b') with x_flat ordered [pos][channel]
    (the ROIAlign kernel's output order).  Returns (W' [fc, 64*C] fp32, b' [fc] fp32) on the CPU; done in fp64."""
    f64 = torch.float64
    W3 = sd[prefix + "conv3.weight"].detach().cpu().to(f64).flatten(1)
    b3 = sd[prefix + "conv3.bias"].detach().cpu().to(f64)
    W1 = sd[prefix + "conv1.weight"].detach().cpu().to(f64).flatten(1)
    b1 = sd[prefix + "conv1.bias"].detach().cpu().to(f64)
    W2 = sd[prefix + "conv2.weight"].detach().cpu().to(f64).flatten(1)
    b2 = sd[prefix + "conv2.bias"].detach().cpu().to(f64)
    fw = sd[prefix + "box_head.0.fc1.weight"].detach().cpu().to(f64)
    fb = sd[prefix + "box_head.0.fc1.bias"].detach().cpu().to(f64)
    Cc = W1.shape[1]
    P = fw.shape[1] // Cc
    s = support_8.detach().cpu().to(torch.float32).mean(0).to(f64).reshape(Cc, P)        # mean over shots in fp32 like the reference
    Wc = W3[:, :Cc].clone()
    Wc[: W1.shape[0]] += W1
    const = W3[:, Cc:] @ s + b3[:, None]
    const[: W1.shape[0]] += b1[:, None]
    const[W1.shape[0]:] += W2 @ s + b2[:, None]
    f3 = fw.reshape(fw.shape[0], Cc, P)                                                    # [o][c][pos]  (NCHW flatten)
    Wp = torch.einsum("ocp,cd->opd", f3, Wc).reshape(fw.shape[0], P * Cc)                  # [o][pos*C + c']
    bp = fb + torch.einsum("ocp,cp->o", f3, const)
    return Wp.to(torch.float32).contiguous(), bp.to(torch.float32).contiguous()


# ---------------------------------------------------------------------------------------------------
class Engine:
    """ore_engine: whole eval hot path (backbone+FPN -> correlation -> head -> top-k/NMS) behind one call."""

    def __init__(self, *, stem=(64, 64, 128), conv=(64, 80, 96, 112), out=(112, 256, 384, 512), layers=3, fpn_ch=128,
                 strides=(8, 16, 32), pixel_mean=(103.530, 116.280, 123.675), pixel_std=(1.0, 1.0, 1.0),
                 score_thresh=1e-5, pre_topk=1000, nms_thresh=0.6, post_topk=256, max_batch=1, max_h=640, max_w=640,
                 device: int = 0):
        cfg = ModelCfg()
        cfg.stem_ch[:] = stem
        cfg.stage_conv_ch[:] = conv
        cfg.stage_out_ch[:] = out
        cfg.layers_per_block, cfg.fpn_ch = layers, fpn_ch
        cfg.strides[:] = strides
        cfg.pixel_mean[:] = pixel_mean
        cfg.pixel_std[:] = pixel_std
        cfg.score_thresh, cfg.pre_topk, cfg.nms_thresh, cfg.post_topk = score_thresh, pre_topk, nms_thresh, post_topk
        cfg.max_batch, cfg.max_h, cfg.max_w = max_batch, max_h, max_w
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.pre_topk = pre_topk
        self.fpn_ch = fpn_ch
        self._h = C.c_void_p()
        _chk(lib().ore_engine_create(C.byref(cfg), device, C.byref(self._h)), "ore_engine_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().ore_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd) -> None:
        """All tensors by their reference state_dict names; unknown keys are ignored by the engine."""
        for k, v in sd.items():
            t = v.detach().float().cpu().contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            _chk(lib().ore_engine_set_tensor(self._h, k.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()),
                 f"ore_engine_set_tensor({k})")

    def set_support(self, protos: Dict[str, torch.Tensor]) -> None:
        """{'p3': [1,C,32,32], 'p4': [1,C,16,16], 'p5': [1,C,8,8]} -- support_feature.pkl prototypes."""
        for lvl in (3, 4, 5):
            t = protos[f"p{lvl}"].detach().float().cpu().contiguous()
            t = t.reshape(t.shape[-3], t.shape[-2], t.shape[-1])
            _chk(lib().ore_engine_set_support(self._h, lvl, C.c_void_p(t.data_ptr()), t.shape[0], t.shape[1]),
                 "ore_engine_set_support")

    def finalize(self) -> None:
        _chk(lib().ore_engine_finalize(self._h), "ore_engine_finalize")

    def set_roi_head(self, sd, support_8: torch.Tensor, reg_weights=(10.0, 10.0, 5.0, 5.0), score_thresh: float = 0.0,
                     nms_thresh: float = 0.9, topk: int = 100, prefix: str = "roi_heads.") -> None:
        """Second stage inside the engine graph: composes DSA-mix + flatten + fc1 once (compose_roi_head) and uploads it."""
        Wp, bp = compose_roi_head(sd, support_8, prefix)
        f = lambda k: sd[prefix + k].detach().float().cpu().contiguous()
        cw, cb, bw, bb = f("box_predictor.0.cls_score.weight"), f("box_predictor.0.cls_score.bias"), f("box_predictor.0.bbox_pred.weight"), f("box_predictor.0.bbox_pred.bias")
        assert cw.shape[0] == 2 and bw.shape[0] == 4, "one foreground class, class-agnostic box regression"
        pooled = int(round((Wp.shape[1] // self.fpn_ch) ** 0.5))
        rw = (C.c_float * 4)(*reg_weights)
        _chk(lib().ore_engine_set_roi_head(self._h, C.c_void_p(Wp.data_ptr()), C.c_void_p(bp.data_ptr()), Wp.shape[0], pooled,
                                           C.c_void_p(cw.data_ptr()), C.c_void_p(cb.data_ptr()), C.c_void_p(bw.data_ptr()),
                                           C.c_void_p(bb.data_ptr()), rw, C.c_float(score_thresh), C.c_float(nms_thresh), topk),
             "ore_engine_set_roi_head")
        self.has_roi = True
        self._det_cap = min(max(int(topk), 1), 320)        # rows of the result tensors ore_engine_detect_fwd fills

    def detections(self, b: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(boxes [k,4], scores [k], source proposal index [k]) of the second stage for image b -- one sync to read the count."""
        k = int(self.buffer(f"det_count#{b}")[0, 0].item())
        return self.buffer(f"det_boxes#{b}")[:k], self.buffer(f"det_scores#{b}")[:k, 0], self.buffer(f"det_src#{b}")[:k, 0]

    def backbone(self, img: torch.Tensor) -> Dict[str, torch.Tensor]:
        """img [B,3,H,W] u8/f32 on device -> {'p3','p4','p5'} as logical NCHW views of engine-owned NHWC buffers."""
        assert img.is_cuda and img.is_contiguous()
        B, _, H, W = img.shape
        _chk(lib().ore_engine_backbone_fwd(self._h, C.c_void_p(img.data_ptr()), int(img.dtype == torch.uint8), B, H, W, _stream()),
             "ore_engine_backbone_fwd")
        Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
        return {f"p{l}": self.buffer(f"p{l}", (B, Hp >> l, Wp >> l)) for l in (3, 4, 5)}

    def eval_forward(self, img: torch.Tensor, use_graph: bool = True) -> None:
        """img [3,H,W] u8/f32 on device.  Enqueues the whole path; results via .buffer()/.proposals()."""
        assert img.is_cuda and img.is_contiguous() and img.dim() == 3
        _, H, W = img.shape
        _chk(lib().ore_engine_eval_fwd(self._h, C.c_void_p(img.data_ptr()), int(img.dtype == torch.uint8), H, W,
                                       int(use_graph), _stream()), "ore_engine_eval_fwd")

    def detect_begin(self, img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
        """First half of `detect`: enqueue the pass (ore_engine_detect_begin) and return the result record it will fill.  Host work
        done before `detect_end` runs in the shadow of the device pass."""
        assert img.is_contiguous() and img.dim() == 3
        _, H, W = img.shape
        R = self.__dict__.get("_rec_rows") or self.__dict__.setdefault("_rec_rows", int(lib().ore_det_record_rows()))   # ORE_DET_RECORD_ROWS of the loaded library (= the engine's roi_cap)
        rec = torch.empty(R * 7, dtype=torch.float32, device=self.device)  # [R][4] f32 | [R] f32 | [R] i64, written by the last kernel
        _chk(lib().ore_engine_detect_begin(self._h, C.c_void_p(img.data_ptr()), int(img.dtype == torch.uint8), H, W, int(out_h), int(out_w),
                                           C.c_void_p(rec.data_ptr()), _stream()), "ore_engine_detect_begin")
        return rec

    def detect_end(self, rec: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Second half: wait for the count (ore_engine_detect_end) and cut the record into (boxes [n,4], scores [n], classes [n] int64)."""
        n = C.c_int32(0)
        _chk(lib().ore_engine_detect_end(self._h, _stream(), C.byref(n)), "ore_engine_detect_end")
        k, R = n.value, self._rec_rows                                     # one strided view each: the protocol pays per torch call
        return rec.as_strided((k, 4), (4, 1)), rec.as_strided((k,), (1,), R * 4), rec.view(torch.int64).as_strided((k,), (1,), R * 5 // 2)

    def detect(self, img: torch.Tensor, out_h: int, out_w: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The reference's eval call for one image, end to end (ore_engine_detect_begin + _end): image [3,H,W] u8/f32 on
        the device or on the host -> both stages + detector_postprocess as one hipGraph replay whose last kernel writes the results into
        a freshly allocated tensor, the count through a device-mapped pinned word the call polls.
        Returns (boxes [n,4], scores [n], classes [n] int64) -- the caller's own tensors, nothing of the engine aliases them."""
        return self.detect_end(self.detect_begin(img, out_h, out_w))

    def eval_forward_batch(self, imgs: torch.Tensor, use_graph: bool = True) -> None:
        """imgs [B,3,H,W] u8/f32 on device, B <= max_batch: dense stages batched, detection tail + second stage per image
        (ore_engine_eval_batch_fwd).  Image b's results: .proposals(b) / .detections(b) / .buffer("name#b")."""
        assert imgs.is_cuda and imgs.is_contiguous() and imgs.dim() == 4
        B, _, H, W = imgs.shape
        _chk(lib().ore_engine_eval_batch_fwd(self._h, C.c_void_p(imgs.data_ptr()), int(imgs.dtype == torch.uint8), B, H, W,
                                             int(use_graph), _stream()), "ore_engine_eval_batch_fwd")

    def last_flops(self) -> float:
        return float(lib().ore_engine_last_flops(self._h))

    def set_profiling(self, enable: bool) -> None:
        _chk(lib().ore_engine_set_profiling(self._h, int(enable)), "ore_engine_set_profiling")

    def read_profile(self) -> Tuple[float, float, int]:
        """(conv kernel ms, conv algorithmic FLOPs, conv launches) accumulated over eager forwards since the last read."""
        ms, fl, n = C.c_double(), C.c_double(), C.c_int32()
        _chk(lib().ore_engine_read_profile(self._h, C.byref(ms), C.byref(fl), C.byref(n)), "ore_engine_read_profile")
        return ms.value, fl.value, n.value

    def profile_executed_flops(self) -> float:
        """Of the FLOPs the last read_profile() reported, the multiplies actually executed (Winograd layers: algorithmic / 2.25)."""
        return float(lib().ore_engine_profile_executed_flops(self._h))

    def buffer(self, name: str, bhw: Optional[Tuple[int, int, int]] = None) -> torch.Tensor:
        """Zero-copy torch view of a named engine buffer.  With bhw=(B,H,W): logical NCHW view of the channel slice."""
        p = C.c_void_p()
        dims = (C.c_int64 * 4)()
        _chk(lib().ore_engine_buffer(self._h, name.encode(), C.byref(p), dims), f"ore_engine_buffer({name})")
        rows, ch, ld, coff = (int(x) for x in dims)
        dt = {"pre_loc": torch.int64, "keep_idx": torch.int64, "pre_level": torch.int32, "counts": torch.int32, "det_src": torch.int64,
              "det_count": torch.int32, "roi_ok": torch.int32}.get(name.split("#")[0], torch.float32)
        if lib().ore_engine_buffer_is_bf16(self._h, name.encode()):
            dt = torch.bfloat16                               # activation buffers of a bf16-storage engine
        flat = _from_ptr(p.value, rows * ld, dt, self.device)
        t = flat.view(rows, ld)[:, coff:coff + ch]
        if bhw is not None:
            B, H, W = bhw
            assert B * H * W == rows, (bhw, rows)
            t = t.reshape(B, H, W, ch) if coff == 0 and ch == ld else flat.view(B, H, W, ld)[..., coff:coff + ch]
            return t.permute(0, 3, 1, 2)
        return t

    def proposals(self, b: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(boxes [n,4], scores [n], keep_idx [n]) of image b -- reads the device-side count (one sync)."""
        n = int(self.buffer(f"counts#{b}")[1, 0].item())
        return self.buffer(f"out_boxes#{b}")[:n], self.buffer(f"out_scores#{b}")[:n, 0], self.buffer(f"keep_idx#{b}")[:n, 0]


class _Arr:
    def __init__(self, ptr, n, dt):
        tstr = {torch.float32: "<f4", torch.int64: "<i8", torch.int32: "<i4", torch.int16: "<i2"}[dt]
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": tstr, "data": (ptr, False), "version": 2}


def _from_ptr(ptr: int, n: int, dt, device) -> torch.Tensor:
    if dt == torch.bfloat16:                                  # not expressible in __cuda_array_interface__: view 16-bit integers
        return torch.as_tensor(_Arr(ptr, n, torch.int16), device=device).view(torch.bfloat16)
    return torch.as_tensor(_Arr(ptr, n, dt), device=device)
