"""DatasetMapperWithSupport (ref:fewx/data/dataset_mapper.py:27-269): maps a dataset dict to the model's input dict
{image, instances, support_images [way*shot,3,240,240], support_bboxes [way*shot,4], support_cls}: the layout the hot path consumes
(fewx/modeling/fsod/train_forward.py).

Host logic only (pandas + numpy + PIL).  The ore dataset -- images and `./datasets/coco/*_shot_support_df.pkl` -- is not shipped with
the reference, so the two sources are injectable: `support_df` (a DataFrame with the columns id / image_id / category_id / file_path /
support_box) and `read_image(path, format) -> HxWx3 uint8`.  Left at their defaults they are the reference's own paths and a PIL
reader, and a missing file raises FileNotFoundError -- there is no synthetic fallback.
`generate_support` is pinned against the reference's own method executed on a synthetic dataframe (tests/golden/generate_support.npz)."""
import copy
import os

import numpy as np
import torch


def _pil_read_image(path, format="BGR"):
    """detectron2.data.detection_utils.read_image for RGB / BGR (d2z:data/detection_utils.py): HxWx3 uint8."""
    from PIL import Image
    with open(path, "rb") as f:
        img = np.asarray(Image.open(f).convert("RGB"))
    return np.ascontiguousarray(img[:, :, ::-1]) if format == "BGR" else img


class DatasetMapperWithSupport:
    SUPPORT_DIR = "./datasets/coco/"                      # ref:fewx/data/dataset_mapper.py:80-82,227

    def __init__(self, cfg, is_train=True, support_df=None, read_image=None):
        self.is_train = is_train
        self.img_format = cfg.INPUT.FORMAT
        self.support_way = cfg.INPUT.FS.SUPPORT_WAY
        self.support_shot = cfg.INPUT.FS.SUPPORT_SHOT
        self.few_shot = cfg.INPUT.FS.FEW_SHOT
        self.min_size = cfg.INPUT.MIN_SIZE_TRAIN if is_train else (cfg.INPUT.MIN_SIZE_TEST,)
        self.max_size = cfg.INPUT.MAX_SIZE_TRAIN if is_train else cfg.INPUT.MAX_SIZE_TEST
        self.sample_style = cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING if is_train else "choice"
        self.read_image = read_image or _pil_read_image
        self.support_on = is_train
        self.support_df = support_df
        if is_train and support_df is None:
            import pandas as pd
            path = os.path.join(self.SUPPORT_DIR, "10_shot_support_df.pkl" if self.few_shot else "train_support_df.pkl")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: the support dataframe of the ore dataset is needed for training "
                                        "(or pass support_df=...)")
            self.support_df = pd.read_pickle(path)

    # ---- ref:fewx/data/dataset_mapper.py:198-269 -----------------------------------------------------------------------------
    def generate_support(self, dataset_dict):
        """support_way x support_shot support crops for the query: `shot` annotations of the query's class drawn one by one with
        pandas `.sample(random_state=<query annotation id>)` from the rows that are neither in the query image nor used already, then
        (way > 1) the same for `way - 1` other classes absent from the query image.  Returns (float32 [way*shot,3,240,240] in the
        reader's channel order, float32 [way*shot,4] boxes inside the crops, class flags: 0 = the query's class, 1 = another)."""
        df = self.support_df
        way, shot = self.support_way, self.support_shot
        qid = dataset_dict["annotations"][0]["id"]
        query_cls = df.loc[df["id"] == qid, "category_id"].tolist()[0]
        query_img = df.loc[df["id"] == qid, "image_id"].tolist()[0]
        all_cls = df.loc[df["image_id"] == query_img, "category_id"].tolist()
        data = np.zeros((way * shot, 3, 240, 240), dtype=np.float32)
        boxes = np.zeros((way * shot, 4), dtype=np.float32)
        used_images = [query_img]
        used_ids = [a["id"] for a in dataset_dict["annotations"]]
        used_cats = list(set(all_cls))
        flags, k = [], 0

        def draw(cls, flag):
            nonlocal k
            pool = df.loc[(df["category_id"] == cls) & (~df["image_id"].isin(used_images)) & (~df["id"].isin(used_ids)), "id"]
            sid = pool.sample(random_state=qid).tolist()[0]
            row = df.loc[df["id"] == sid, :]
            used_ids.append(sid)
            used_images.append(row["image_id"].tolist()[0])
            img = self.read_image(self.SUPPORT_DIR + row["file_path"].tolist()[0], format=self.img_format)
            data[k] = np.ascontiguousarray(img.transpose(2, 0, 1))
            boxes[k] = row["support_box"].tolist()[0]
            flags.append(flag)
            k += 1
        for _ in range(shot):
            draw(query_cls, 0)
        for _ in range(way - 1):
            other = df.loc[~df["category_id"].isin(used_cats), "category_id"].drop_duplicates().sample(random_state=qid).tolist()[0]
            used_cats.append(other)
            for _ in range(shot):
                draw(other, 1)
        return data, boxes, flags

    # ---- ref:fewx/data/dataset_mapper.py:99-196 (boxes only: MASK_ON / KEYPOINT_ON are false in every fsod config) -----------
    def _resize_flip(self, image, rng):
        """ResizeShortestEdge(MIN_SIZE, MAX_SIZE, sampling) + RandomFlip(0.5) of d2z:data/detection_utils.py build_transform_gen;
        returns the image and the box transform."""
        from PIL import Image
        h, w = image.shape[:2]
        sizes = list(self.min_size) if not isinstance(self.min_size, int) else [self.min_size]
        size = int(rng.integers(sizes[0], sizes[-1] + 1)) if self.sample_style == "range" and len(sizes) > 1 else int(rng.choice(sizes))
        scale = size * 1.0 / min(h, w)
        nh, nw = (size, scale * w) if h < w else (scale * h, size)
        if max(nh, nw) > self.max_size:
            s2 = self.max_size * 1.0 / max(nh, nw)
            nh, nw = nh * s2, nw * s2
        nh, nw = int(nh + 0.5), int(nw + 0.5)
        out = np.asarray(Image.fromarray(image).resize((nw, nh), Image.BILINEAR))
        flip = bool(self.is_train and rng.random() < 0.5)
        if flip:
            out = np.ascontiguousarray(out[:, ::-1])

        def tf_boxes(b):
            b = np.asarray(b, dtype=np.float32).reshape(-1, 4).copy()
            b[:, 0::2] *= nw * 1.0 / w
            b[:, 1::2] *= nh * 1.0 / h
            if flip:
                x1 = nw - b[:, 2]
                b[:, 2] = nw - b[:, 0]
                b[:, 0] = x1
            return b
        return out, tf_boxes

    def __call__(self, dataset_dict):
        from detectron2.structures import Boxes, Instances
        d = copy.deepcopy(dataset_dict)
        image = self.read_image(d["file_name"], format=self.img_format)
        if self.is_train and self.support_on:
            sup, sbox, scls = self.generate_support(d)
            d["support_images"] = torch.as_tensor(np.ascontiguousarray(sup))
            d["support_bboxes"] = sbox
            d["support_cls"] = scls
        rng = np.random.default_rng()
        image, tf_boxes = self._resize_flip(image, rng)
        d["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if not self.is_train:
            d.pop("annotations", None)
            return d
        if "annotations" in d:
            annos = [a for a in d.pop("annotations") if a.get("iscrowd", 0) == 0]
            raw = np.array([a["bbox"] for a in annos], dtype=np.float32).reshape(-1, 4)
            if len(annos) and annos[0].get("bbox_mode", 1) == 1:         # BoxMode.XYWH_ABS -> XYXY_ABS
                raw[:, 2:] += raw[:, :2]
            b = tf_boxes(raw)
            H, W = image.shape[:2]
            b[:, 0::2] = b[:, 0::2].clip(0, W)
            b[:, 1::2] = b[:, 1::2].clip(0, H)
            keep = (b[:, 2] > b[:, 0]) & (b[:, 3] > b[:, 1])                # filter_empty_instances
            inst = Instances((H, W))
            inst.gt_boxes = Boxes(torch.from_numpy(b[keep]))
            inst.gt_classes = torch.tensor([a["category_id"] for a, k_ in zip(annos, keep) if k_], dtype=torch.int64)
            d["instances"] = inst
        return d
