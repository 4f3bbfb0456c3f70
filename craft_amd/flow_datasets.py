"""Evaluation-side dataset walkers: the file-list logic of ``core/datasets.py`` without
augmentation, DataLoader workers or sklearn -- just (image1, image2, flow, valid) for the validation harness.

* ``MpiSintel``     datasets.py:155-201   <root>/<split>/<clean|final>/<scene>/frame_XXXX.png, flow/<scene>/*.flo
* ``KITTI``         datasets.py:282-307   <root>/<split>/image_2/*_10.png, *_11.png, flow_occ/*_10.png (sparse, 16-bit PNG)
* ``FlyingChairs``  datasets.py:203-221   <root>/*.ppm, *.flo + a split file (1 = training, 2 = validation)

``__getitem__`` mirrors ``FlowDataset.__getitem__`` (datasets.py:59-140): images float32 [3, H, W] in 0..255, flow
[2, H, W], valid [H, W] float (dense: |u| < 1000 and |v| < 1000; sparse: the file's valid channel); test splits
return (img1, img2, extra_info).
"""
from __future__ import annotations

import os
import os.path as osp
from glob import glob
from typing import List, Optional

import numpy as np
import torch

from . import flow_io


class FlowDataset:
    def __init__(self, sparse: bool = False):
        self.sparse = sparse
        self.is_test = False
        self.flow_list: List[str] = []
        self.image_list: List[List[str]] = []
        self.extra_info: Optional[list] = None

    def __len__(self) -> int:
        return len(self.image_list)

    def __getitem__(self, index: int):
        extra = self.extra_info[index] if self.extra_info is not None else 0
        img1 = torch.from_numpy(flow_io.read_image(self.image_list[index][0])).permute(2, 0, 1).float()
        img2 = torch.from_numpy(flow_io.read_image(self.image_list[index][1])).permute(2, 0, 1).float()
        if self.is_test:
            return img1, img2, extra
        index = index % len(self.image_list)
        if self.sparse:
            flow, valid = flow_io.read_flow_kitti(self.flow_list[index])
            valid_t = torch.from_numpy(np.ascontiguousarray(valid))
        else:
            flow = flow_io.read_gen(self.flow_list[index])
            valid_t = None
        flow_t = torch.from_numpy(np.ascontiguousarray(np.asarray(flow, dtype=np.float32))).permute(2, 0, 1).float()
        if valid_t is None:
            valid_t = (flow_t[0].abs() < 1000) & (flow_t[1].abs() < 1000)
        return img1, img2, flow_t, valid_t.float(), extra


class MpiSintel(FlowDataset):
    def __init__(self, split: str = "training", root: str = "datasets/Sintel", dstype: str = "clean"):
        super().__init__()
        self.ds_name = f"sintel-{split}-{dstype}"
        flow_root = osp.join(root, split, "flow")
        image_root = osp.join(root, split, dstype)
        if split == "test":
            self.is_test = True
        self.extra_info = []
        for scene in sorted(os.listdir(image_root)):
            images = sorted(glob(osp.join(image_root, scene, "*.png")))
            for i in range(len(images) - 1):
                self.image_list.append([images[i], images[i + 1]])
                self.extra_info.append((scene, i))
            if split != "test":
                self.flow_list += sorted(glob(osp.join(flow_root, scene, "*.flo")))
        if not self.is_test and len(self.flow_list) != len(self.image_list):
            raise ValueError(f"{self.ds_name}: {len(self.image_list)} image pairs but {len(self.flow_list)} flow files")


class KITTI(FlowDataset):
    def __init__(self, split: str = "training", root: str = "datasets/KITTI"):
        super().__init__(sparse=True)
        self.ds_name = f"kitti-{split}"
        if split == "testing":
            self.is_test = True
        root = osp.join(root, split)
        images1 = sorted(glob(osp.join(root, "image_2/*_10.png")))
        images2 = sorted(glob(osp.join(root, "image_2/*_11.png")))
        self.extra_info = []
        for a, b in zip(images1, images2):
            self.image_list.append([a, b])
            self.extra_info.append([a.split("/")[-1]])
        if split == "training":
            self.flow_list = sorted(glob(osp.join(root, "flow_occ/*_10.png")))


class FlyingChairs(FlowDataset):
    def __init__(self, split: str = "validation", root: str = "datasets/FlyingChairs_release/data",
                 split_file: Optional[str] = None):
        super().__init__()
        self.ds_name = f"chairs-{split}"
        images = sorted(glob(osp.join(root, "*.ppm")))
        flows = sorted(glob(osp.join(root, "*.flo")))
        if len(images) // 2 != len(flows):
            raise ValueError(f"{self.ds_name}: {len(images)} images for {len(flows)} flows")
        split_file = split_file or osp.join(osp.dirname(root.rstrip("/")), "FlyingChairs_train_val.txt")
        split_list = np.loadtxt(split_file, dtype=np.int32).reshape(-1)
        for i in range(len(flows)):
            xid = int(split_list[i])
            if (split == "training" and xid == 1) or (split == "validation" and xid == 2):
                self.flow_list.append(flows[i])
                self.image_list.append([images[2 * i], images[2 * i + 1]])
