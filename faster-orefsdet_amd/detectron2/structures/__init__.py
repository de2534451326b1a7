"""Boxes / Instances / ImageList: the container types the reference modules exchange (d2z:structures/*.py).
Host-side glue only -- minimal re-provision with the same method names and semantics."""
from __future__ import annotations

import itertools
from typing import Any, Dict, List, Tuple, Union

import torch
import torch.nn.functional as F


class Boxes:
    """Nx4 (x1, y1, x2, y2) float tensor wrapper (d2z:structures/boxes.py:130-307)."""

    def __init__(self, tensor):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *a, **k):
        return Boxes(self.tensor.to(*a, **k))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold: float = 0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, f"Indexing on Boxes with {item} failed to return a matrix!"
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    @classmethod
    def cat(cls, boxes_list: List["Boxes"]) -> "Boxes":
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


def pairwise_iou(boxes1: Boxes, boxes2: Boxes) -> torch.Tensor:
    a1, a2 = boxes1.area(), boxes2.area()
    b1, b2 = boxes1.tensor, boxes2.tensor
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype, device=inter.device))


class Instances:
    """Per-image bag of equally long fields (d2z:structures/instances.py)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        n = len(value)
        if len(self._fields):
            assert len(self) == n, f"Adding a field of length {n} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *a, **k):
        ret = Instances(self._image_size)
        for key, v in self._fields.items():
            ret.set(key, v.to(*a, **k) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item):
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        size = instance_lists[0].image_size
        ret = Instances(size)
        for k in instance_lists[0]._fields.keys():
            vals = [i.get(k) for i in instance_lists]
            v0 = vals[0]
            if isinstance(v0, torch.Tensor):
                vals = torch.cat(vals, dim=0)
            elif isinstance(v0, list):
                vals = list(itertools.chain(*vals))
            elif hasattr(type(v0), "cat"):
                vals = type(v0).cat(vals)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret.set(k, vals)
        return ret

    def __repr__(self):
        return f"Instances(num_instances={len(self) if self._fields else 0}, image_size={self._image_size}, fields={list(self._fields)})"


class ImageList:
    """Batch of images padded to one size (d2z:structures/image_list.py)."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        h, w = self.image_sizes[idx]
        return self.tensor[idx, ..., :h, :w]

    def to(self, *a, **k):
        return ImageList(self.tensor.to(*a, **k), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors: List[torch.Tensor], size_divisibility: int = 0, pad_value: float = 0.0) -> "ImageList":
        assert len(tensors) > 0
        sizes = [(t.shape[-2], t.shape[-1]) for t in tensors]
        mh, mw = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            mh, mw = (mh + d - 1) // d * d, (mw + d - 1) // d * d
        if len(tensors) == 1:
            h, w = sizes[0]
            batched = F.pad(tensors[0], [0, mw - w, 0, mh - h], value=pad_value).unsqueeze_(0)
        else:
            batched = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (mh, mw), pad_value)
            for img, dst in zip(tensors, batched):
                dst[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), sizes)


__all__ = ["Boxes", "Instances", "ImageList", "pairwise_iou"]
