"""Shim loader: run the reference's own files, unmodified, by path.

TEST INFRASTRUCTURE ONLY, this container only.  Nothing from /root/reference is copied: files are
executed where they lie (the vendored detectron2.7z is unpacked into a temp dir outside the repo).

Whole-package import of ``fewx`` / ``detectron2`` is impossible here (fvcore, yacs, iopath,
torchvision, pycocotools, cv2, black ... are absent), so:
  * absent third-party modules get stub modules in ``sys.modules``;
  * ``detectron2`` / ``CenterNet2`` become skeleton packages (empty ``__init__``) whose ``__path__``
    points at the real directories, so relative imports inside the reference files resolve;
  * the handful of names the reference files import from package ``__init__``s are set by hand.
The only non-reference arithmetic in the chain is ``batched_nms`` (torchvision is un-vendored):
it is oracle.decode.nms, i.e. the restated published algorithm.
"""
from __future__ import annotations

import importlib.util
import math
import os
import subprocess
import sys
import tempfile
import types

import torch

REF = "/root/reference"
_D2 = None


def d2_root() -> str:
    """Unpack detectron2.7z once (libarchive via `cmake -E tar`; no 7z tool in the image)."""
    global _D2
    if _D2 is None:
        d = os.path.join(tempfile.gettempdir(), "orefsdet_d2x")
        if not os.path.exists(os.path.join(d, "modeling", "backbone", "vovnet.py")):
            os.makedirs(d, exist_ok=True)
            subprocess.check_call(["cmake", "-E", "tar", "xf", os.path.join(REF, "detectron2.7z")], cwd=d)
        _D2 = d
    return _D2


def mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class _Registry(dict):
    def __init__(self, name="r"):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            return lambda o: self.register(o)
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


def _is_cfg(x):
    return hasattr(x, "MODEL") and not isinstance(x, torch.nn.Module)


def _configurable(init_func=None, *, from_config=None):
    """Our own stand-in for detectron2.config.configurable (d2z:config/config.py needs fvcore/yacs): an ``__init__`` called
    with a cfg as first argument (or ``cfg=``) gets its explicit arguments from the class's ``from_config``; explicit keyword
    arguments override; a call with explicit arguments passes straight through."""
    import functools

    if init_func is None:
        return lambda f: f

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        if (args and _is_cfg(args[0])) or _is_cfg(kwargs.get("cfg")):
            import inspect
            fc = type(self).from_config
            params = inspect.signature(fc).parameters
            if any(p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in params.values()):
                explicit = fc(*args, **kwargs)
            else:                                   # keyword arguments from_config does not know override its result
                extra = {k: kwargs.pop(k) for k in list(kwargs) if k not in params}
                explicit = fc(*args, **kwargs)
                explicit.update(extra)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)
    return wrapped


_LOADED = {}


def setup():
    """Install stubs + load the reference files.  Returns a namespace of the loaded modules."""
    if _LOADED:
        return types.SimpleNamespace(**_LOADED)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import decode as odec

    D2 = d2_root()
    # ---- third-party stubs
    def c2_xavier_fill(m):
        torch.nn.init.kaiming_uniform_(m.weight, a=1)
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)

    def c2_msra_fill(m):
        torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)

    wi = mod("fvcore.nn.weight_init", c2_xavier_fill=c2_xavier_fill, c2_msra_fill=c2_msra_fill)
    fd = mod("fvcore.nn.distributed", differentiable_all_reduce=None)
    fnn = mod("fvcore.nn", weight_init=wi, distributed=fd, smooth_l1_loss=None, giou_loss=None)
    mod("fvcore", nn=fnn)
    mod("cv2")
    mod("black", T=None)
    mod("demo_visualizer", Have_a_Look=None)

    # ---- detectron2 skeleton
    pkg("detectron2", D2)
    pkg("detectron2.layers", D2 + "/layers")
    pkg("detectron2.utils", D2 + "/utils")
    pkg("detectron2.modeling", D2 + "/modeling")
    pkg("detectron2.modeling.backbone", D2 + "/modeling/backbone")
    pkg("detectron2.modeling.proposal_generator", D2 + "/modeling/proposal_generator")
    pkg("detectron2.structures", D2 + "/structures")
    mod("detectron2.config", configurable=_configurable)
    sys.modules["detectron2"].config = sys.modules["detectron2.config"]
    mod("detectron2.utils.comm", get_world_size=lambda: 1, get_rank=lambda: 0)
    mod("detectron2.utils.env", TORCH_VERSION=tuple(int(x) for x in torch.__version__.split(".")[:2]))
    mod("detectron2.utils.events", get_event_storage=lambda: None)
    mod("detectron2.utils.registry", Registry=_Registry)
    mod("detectron2.utils.logger", log_first_n=lambda *a, **k: None)
    U = sys.modules["detectron2.utils"]
    for n in ("comm", "env", "events", "registry", "logger"):
        setattr(U, n, sys.modules["detectron2.utils." + n])
    L = sys.modules["detectron2.layers"]
    sh = load("detectron2.layers.shape_spec", D2 + "/layers/shape_spec.py")
    L.ShapeSpec = sh.ShapeSpec
    wr = load("detectron2.layers.wrappers", D2 + "/layers/wrappers.py")
    for n in ("Conv2d", "cat", "BatchNorm2d", "ConvTranspose2d", "interpolate", "Linear", "nonzero_tuple",
              "cross_entropy"):
        setattr(L, n, getattr(wr, n))
    bn = load("detectron2.layers.batch_norm", D2 + "/layers/batch_norm.py")
    L.FrozenBatchNorm2d, L.get_norm, L.NaiveSyncBatchNorm = bn.FrozenBatchNorm2d, bn.get_norm, bn.NaiveSyncBatchNorm
    L.DeformConv = L.ModulatedDeformConv = None

    def batched_nms(boxes, scores, idxs, thr):  # torchvision un-vendored -> restated algorithm
        assert int(idxs.max()) == 0 if len(idxs) else True
        keep = odec.nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), float(thr))
        return torch.from_numpy(keep)

    L.batched_nms = batched_nms
    L.nms = lambda b, s, t: batched_nms(b, s, torch.zeros(len(s)), t)

    B = sys.modules["detectron2.modeling.backbone"]
    bb = load("detectron2.modeling.backbone.backbone", D2 + "/modeling/backbone/backbone.py")
    B.Backbone = bb.Backbone
    bd = load("detectron2.modeling.backbone.build", D2 + "/modeling/backbone/build.py")
    B.BACKBONE_REGISTRY, B.build_backbone = bd.BACKBONE_REGISTRY, bd.build_backbone
    mod("detectron2.modeling.backbone.resnet", build_resnet_backbone=None)
    fpn = load("detectron2.modeling.backbone.fpn", D2 + "/modeling/backbone/fpn.py")
    B.FPN = fpn.FPN
    vov = load("detectron2.modeling.backbone.vovnet", D2 + "/modeling/backbone/vovnet.py")

    S = sys.modules["detectron2.structures"]
    bx = load("detectron2.structures.boxes", D2 + "/structures/boxes.py")
    ins = load("detectron2.structures.instances", D2 + "/structures/instances.py")
    iml = load("detectron2.structures.image_list", D2 + "/structures/image_list.py")
    S.Boxes, S.BoxMode, S.pairwise_iou = bx.Boxes, bx.BoxMode, bx.pairwise_iou
    S.Instances, S.ImageList = ins.Instances, iml.ImageList
    load("detectron2.utils.memory", D2 + "/utils/memory.py")
    mod("detectron2.modeling.anchor_generator", build_anchor_generator=None)
    mod("detectron2.modeling.box_regression", Box2BoxTransform=None)
    mod("detectron2.modeling.matcher", Matcher=None)
    mod("detectron2.modeling.sampling", subsample_labels=None)
    mod("detectron2.modeling.proposal_generator.build", PROPOSAL_GENERATOR_REGISTRY=_Registry("PG"))
    mod("detectron2.modeling.proposal_generator.proposal_utils", find_top_rpn_proposals=None)

    # ---- CenterNet2 skeleton
    C2 = REF + "/CenterNet2"
    pkg("CenterNet2", C2)
    pkg("CenterNet2.centernet", C2 + "/centernet")
    pkg("CenterNet2.centernet.modeling", C2 + "/centernet/modeling")
    pkg("CenterNet2.centernet.modeling.layers", C2 + "/centernet/modeling/layers")
    pkg("CenterNet2.centernet.modeling.dense_heads", C2 + "/centernet/modeling/dense_heads")
    mod("CenterNet2.centernet.modeling.debug", debug_train=None, debug_test=None)
    lp = C2 + "/centernet/modeling/layers/"
    hfl = load("CenterNet2.centernet.modeling.layers.heatmap_focal_loss", lp + "heatmap_focal_loss.py")
    iou = load("CenterNet2.centernet.modeling.layers.iou_loss", lp + "iou_loss.py")
    load("CenterNet2.centernet.modeling.layers.ml_nms", lp + "ml_nms.py")
    load("CenterNet2.centernet.modeling.layers.deform_conv", lp + "deform_conv.py")
    dp = C2 + "/centernet/modeling/dense_heads/"
    load("CenterNet2.centernet.modeling.dense_heads.utils", dp + "utils.py")
    head = load("CenterNet2.centernet.modeling.dense_heads.centernet_head", dp + "centernet_head.py")

    # ---- fewx files that sit on the path
    pkg("fewx", REF + "/fewx")
    pkg("fewx.modeling", REF + "/fewx/modeling")
    pkg("fewx.modeling.fsod", REF + "/fewx/modeling/fsod")
    rpn = load("fewx.modeling.fsod.fsod_rpn", REF + "/fewx/modeling/fsod/fsod_rpn.py")

    _LOADED.update(vovnet=vov, fpn=fpn, batch_norm=bn, boxes=bx, instances=ins, image_list=iml,
                   centernet_head=head, fsod_rpn=rpn, heatmap_focal_loss=hfl, iou_loss=iou, layers=L)
    return types.SimpleNamespace(**_LOADED)


def load_fsod_cen():
    """Load ref:fewx/modeling/fsod/fsod_cen.py (SM_Block, MLP, CenterNet2Detector) unmodified."""
    ns = setup()
    if "fsod_cen" in _LOADED:
        return _LOADED["fsod_cen"]
    mod("detectron2.data", detection_utils=None, catalog=None)
    mod("detectron2.data.detection_utils", convert_image_to_rgb=None, read_image=None)
    mod("detectron2.data.catalog", MetadataCatalog=None)
    sys.modules["detectron2.data"].detection_utils = sys.modules["detectron2.data.detection_utils"]
    M = sys.modules["detectron2.modeling"]
    mod("detectron2.modeling.postprocessing", detector_postprocess=None)
    mod("detectron2.modeling.poolers", ROIPooler=None)
    pkg("detectron2.modeling.meta_arch", d2_root() + "/modeling/meta_arch")
    mod("detectron2.modeling.meta_arch.build", META_ARCH_REGISTRY=_Registry("META"))
    sys.modules["detectron2.modeling.backbone"].build_backbone = None
    sys.modules["detectron2.modeling.proposal_generator"].build_proposal_generator = None
    mod("fewx.modeling.fsod.fsod_roi_heads", build_roi_heads=None)
    mod("fewx.modeling.fsod.fsod_fast_rcnn", FsodFastRCNNOutputs=None)
    m = load("fewx.modeling.fsod.fsod_cen", REF + "/fewx/modeling/fsod/fsod_cen.py")
    _LOADED["fsod_cen"] = m
    return m


def vovnet_cfg(body="V-19-slim-eSE", fpn_ch=128):
    NS = types.SimpleNamespace
    return NS(MODEL=NS(
        VOVNET=NS(NORM="FrozenBN", CONV_BODY=body, OUT_FEATURES=["stage3", "stage4", "stage5"],
                  STAGE_WITH_DCN=(False, False, False, False), WITH_MODULATED_DCN=False, DEFORMABLE_GROUPS=1),
        BACKBONE=NS(FREEZE_AT=3),
        FPN=NS(IN_FEATURES=["stage3", "stage4", "stage5"], OUT_CHANNELS=fpn_ch, NORM="", FUSE_TYPE="sum"),
        FCOS=NS(TOP_LEVELS=0)))


class AttrDict(dict):
    """yaml mapping with attribute access (stands in for a CfgNode holding the reference's logged, resolved config)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def logged_cfg():
    """The reference's own resolved config (`ref:log/fsod_finetune_stone_vovnet_25_test_log.txt:117-544`, kept as data under
    tests/golden/) as an attribute tree, so reference classes are built by their own ``from_config``."""
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    def wrap(x):
        if isinstance(x, dict):
            return AttrDict({k: wrap(v) for k, v in x.items()})
        if isinstance(x, list):
            return [wrap(v) for v in x]
        return x
    with open(os.path.join(root, "tests", "golden", "vovnet_25shot_full_config.yaml")) as f:
        return wrap(yaml.safe_load(f))


class _Storage:
    """EventStorage stand-in: the second-stage code logs scalars while computing the losses."""
    def put_scalar(self, *a, **k):
        pass

    def name_scope(self, name):
        import contextlib
        return contextlib.nullcontext()


def load_roi_heads():
    """Load the second stage unmodified: d2z:modeling/{poolers,matcher,sampling,box_regression}.py,
    d2z:modeling/proposal_generator/proposal_utils.py, d2z:modeling/roi_heads/{box_head,fast_rcnn,roi_heads,cascade_rcnn}.py,
    d2z:layers/roi_align.py, ref:CenterNet2/centernet/modeling/roi_heads/{fed_loss,custom_fast_rcnn}.py and
    ref:fewx/modeling/fsod/{fsod_fast_rcnn,fsod_roi_heads}.py.  Non-reference arithmetic in the chain (un-vendored third
    parties): torchvision.ops.roi_align and batched_nms (restated: oracle.ref_model.roi_align, oracle.decode.nms) and
    fvcore.nn.smooth_l1_loss (restated below from its published definition)."""
    ns = setup()
    if "roi_heads" in _LOADED:
        return types.SimpleNamespace(**_LOADED)
    from oracle import ref_model as R
    D2 = d2_root()

    def smooth_l1_loss(input, target, beta, reduction="none"):
        # fvcore.nn.smooth_l1_loss: beta < 1e-5 -> plain L1; else 0.5 n^2 / beta below beta, n - 0.5 beta above
        n = torch.abs(input - target)
        loss = n if beta < 1e-5 else torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
        return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss

    sys.modules["fvcore.nn"].smooth_l1_loss = smooth_l1_loss

    def roi_align(input, rois, output_size, spatial_scale, sampling_ratio, aligned):
        assert aligned and sampling_ratio == 0
        ps = output_size[0] if isinstance(output_size, (tuple, list)) else output_size
        out = input.new_zeros(len(rois), input.shape[1], ps, ps)
        for n in range(input.shape[0]):
            idx = torch.nonzero(rois[:, 0] == n).squeeze(1)
            if len(idx):
                out[idx] = R.roi_align(input[n], rois[idx, 1:], spatial_scale, ps)
        return out

    mod("torchvision", __version__="0.8.2")
    mod("torchvision.ops", RoIPool=None, roi_align=roi_align)
    sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]
    mod("turtle", shape=None)
    L = sys.modules["detectron2.layers"]
    ra = load("detectron2.layers.roi_align", D2 + "/layers/roi_align.py")
    L.ROIAlign, L.ROIAlignRotated = ra.ROIAlign, None
    sys.modules["detectron2.utils.events"].get_event_storage = lambda: _Storage()
    M = D2 + "/modeling/"
    br = load("detectron2.modeling.box_regression", M + "box_regression.py")
    mt = load("detectron2.modeling.matcher", M + "matcher.py")
    sp = load("detectron2.modeling.sampling", M + "sampling.py")
    pl = load("detectron2.modeling.poolers", M + "poolers.py")
    pu = load("detectron2.modeling.proposal_generator.proposal_utils", M + "proposal_generator/proposal_utils.py")
    res = sys.modules["detectron2.modeling.backbone.resnet"]
    res.BottleneckBlock = res.ResNet = res.make_stage = None
    pkg("detectron2.modeling.roi_heads", M + "roi_heads")
    mod("detectron2.modeling.roi_heads.keypoint_head", build_keypoint_head=None)
    mod("detectron2.modeling.roi_heads.mask_head", build_mask_head=None)
    bh = load("detectron2.modeling.roi_heads.box_head", M + "roi_heads/box_head.py")
    fr = load("detectron2.modeling.roi_heads.fast_rcnn", M + "roi_heads/fast_rcnn.py")
    rh = load("detectron2.modeling.roi_heads.roi_heads", M + "roi_heads/roi_heads.py")
    cr = load("detectron2.modeling.roi_heads.cascade_rcnn", M + "roi_heads/cascade_rcnn.py")
    C2 = REF + "/CenterNet2/centernet/modeling/roi_heads"
    pkg("CenterNet2.centernet.modeling.roi_heads", C2)
    load("CenterNet2.centernet.modeling.roi_heads.fed_loss", C2 + "/fed_loss.py")
    cf = load("CenterNet2.centernet.modeling.roi_heads.custom_fast_rcnn", C2 + "/custom_fast_rcnn.py")
    load("fewx.modeling.fsod.fsod_fast_rcnn", REF + "/fewx/modeling/fsod/fsod_fast_rcnn.py")
    fro = load("fewx.modeling.fsod.fsod_roi_heads", REF + "/fewx/modeling/fsod/fsod_roi_heads.py")
    _LOADED.update(roi_heads=fro, d2_roi_heads=rh, cascade_rcnn=cr, fast_rcnn=fr, custom_fast_rcnn=cf, box_head=bh, poolers=pl,
                   matcher=mt, sampling=sp, box_regression=br, proposal_utils=pu)
    return types.SimpleNamespace(**_LOADED)


def load_dataset_mapper(read_image):
    """ref:fewx/data/dataset_mapper.py loaded unmodified; `utils.read_image` (detectron2.data.detection_utils, needs cv2/PIL paths of
    the ore dataset) is the caller's synthetic reader."""
    setup()
    if "dataset_mapper" in _LOADED:
        sys.modules["detectron2.data.detection_utils"].read_image = read_image
        return _LOADED["dataset_mapper"]
    mod("fvcore.common", file_io=None)
    mod("fvcore.common.file_io", PathManager=None)
    sys.modules["fvcore"].common = sys.modules["fvcore.common"]
    sys.modules["fvcore.common"].file_io = sys.modules["fvcore.common.file_io"]
    du = mod("detectron2.data.detection_utils", read_image=read_image, convert_image_to_rgb=None)
    tr = mod("detectron2.data.transforms")
    cat = mod("detectron2.data.catalog", MetadataCatalog=types.SimpleNamespace(get=lambda name: None))
    d = mod("detectron2.data", detection_utils=du, transforms=tr, catalog=cat)
    sys.modules["detectron2"].data = d
    pkg("fewx.data", REF + "/fewx/data")
    m = load("fewx.data.dataset_mapper", REF + "/fewx/data/dataset_mapper.py")
    _LOADED["dataset_mapper"] = m
    return m
