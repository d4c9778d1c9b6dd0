"""Input pipeline of the reference, either side of the GPU hot path: the MPI-Sintel dataset layouts and the sample rules
of models/data/datasets.py:113-330 -- file pairing, centre crop to multiples of 64 (:146-150), optional bilinear resize with flow
rescaling (flow_utils.py:120-138), ToTensor + Normalize(0.5, 0.5) (lightning_datamodule.py:20-23), occlusion binarisation at 0.5
(:260-261) -- and `DatasetModule` (80/10/10 `random_split(seed 42)`, lightning_datamodule.py:50-69).

Two ways to produce a batch:
  * `MpiSintel*.__getitem__`  -- the reference's own per-sample host path (numpy + torch on the CPU), same return values;
  * `raw=True` + `prepare_batch_on_device(...)` -- the dataset only DECODES (uint8 frames, raw flow / occlusion) and crop, resize,
    normalisation, flow rescaling and binarisation run as a handful of batched device operations on the training GPU: at several
    hundred pairs per second per GPU the per-sample host path (two float conversions + a resize per frame in a DataLoader worker) is
    what the input side would be bound by.  Same arithmetic (`tests/test_data_formats.py`).

Decoding uses PIL (the reference uses imageio, absent here; both decode PNG to the same uint8 array).  Resizing: the reference calls
`cv2.resize(..., INTER_LINEAR)` = bilinear with half-pixel centres and no antialiasing; here `F.interpolate(mode='bilinear',
align_corners=False)` on float data.  PARITY UNPINNED for this file: the reference's dataset classes need imageio, cv2 and torchvision,
none of which this image has, so they cannot be run to produce fixtures; the rules are restated from the reference's source and
checked against hand-computed expectations on synthetic trees (`tests/test_data_formats.py`).

FlyingChairs (`*.ppm` pairs + `*.flo`, datasets.py:465-530) and FlyingChairs2 (`*-img_0/1.png`, `*-flow_01.flo`, `*-occ_01.png`,
:587-669 -- the dataset of the reference's shipped `config/unsupervised_config.yml`) share the sample rules; only the file pairing differs."""
from __future__ import annotations

from glob import glob
from math import ceil
from os.path import isfile, join, splitext
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset

from .flow_io import normalize_image, read_flow, render_size


def read_gen(filename: str) -> np.ndarray:
    """image ([h, w, 3] uint8; [h, w, 1] for single-channel maps), `.flo` flow ([h, w, 2] float32) or `.bin` / `.raw` numpy dump
    (frame_utils.py:6-29)"""
    ext = splitext(filename)[-1]
    if ext in (".png", ".jpeg", ".ppm", ".jpg"):
        from PIL import Image
        im = np.asarray(Image.open(filename))
        if im.ndim == 2:
            im = np.expand_dims(im, axis=-1)  # occlusion map
        return im[:, :, :3] if im.shape[2] > 3 else im
    if ext in (".bin", ".raw"):
        return np.load(filename)
    if ext == ".flo":
        return read_flow(filename).astype(np.float32)
    raise ValueError(f".{ext} is not supported")


class StaticCenterCrop:
    """datasets.py:48-53"""

    def __init__(self, image_size, crop_size):
        self.th, self.tw = crop_size
        self.h, self.w = image_size

    def __call__(self, img):
        return img[(self.h - self.th) // 2:(self.h + self.th) // 2, (self.w - self.tw) // 2:(self.w + self.tw) // 2, :]


def resize_linear(img: np.ndarray, height: int, width: int) -> np.ndarray:
    """cv2.resize(img, (width, height), interpolation=cv2.INTER_LINEAR) for an [h, w, c] array: bilinear, half-pixel centres, no
    antialiasing; uint8 inputs are rounded back to uint8 as cv2 does"""
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).unsqueeze(0).float()
    out = F.interpolate(t, size=(height, width), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    if img.dtype == np.uint8:
        return out.round().clamp(0, 255).to(torch.uint8).numpy()
    return out.numpy().astype(img.dtype)


def resize_flow(flow: np.ndarray, des_height: int, des_width: int) -> np.ndarray:
    """flow_utils.py:120-138 (bilinear): resize, then u *= W'/W, v *= H'/H"""
    src_h, src_w = flow.shape[:2]
    if (src_h, src_w) == (des_height, des_width):
        return flow
    out = resize_linear(flow.astype(np.float32), des_height, des_width)
    out[:, :, 0] *= float(des_width) / float(src_w)
    out[:, :, 1] *= float(des_height) / float(src_h)
    return out


def _default_transform(img_uint8: np.ndarray) -> torch.Tensor:
    """ToTensor + Normalize([0.5]*3, [0.5]*3) (lightning_datamodule.py:20-23)"""
    return torch.from_numpy(normalize_image(img_uint8))


class _PairDataset(Dataset):
    """shared sample logic of the pair datasets; subclasses provide `_list(root)` -> fills image_list / flow_list / occ_list"""

    def __init__(self, transform=_default_transform, root="", replicates=1, image_size=None, stack_imgs=True, raw=False, want_flow=True,
                 want_occ=False):
        self.transform = transform
        self.replicates, self.image_size, self.stack_imgs, self.raw = replicates, image_size, stack_imgs, raw
        self.want_flow, self.want_occ = want_flow, want_occ
        self.flow_list, self.occ_list, self.image_list = [], [], []
        self._list(root)
        self.size = len(self.image_list)
        if self.size == 0:
            raise FileNotFoundError(f"no samples under {root!r} ({self._layout})")
        h, w = read_gen(self.image_list[0][0]).shape[:2]
        self.render_size = list(render_size(h, w)) if (h % 64 or w % 64) else [h, w]

    def __len__(self):
        return self.size * self.replicates

    def _decode(self, index):
        index = index % self.size
        img1, img2 = read_gen(self.image_list[index][0]), read_gen(self.image_list[index][1])
        flow = read_gen(self.flow_list[index]).astype(np.float32) if self.want_flow else None
        occ = read_gen(self.occ_list[index]) if self.want_occ else None
        return img1, img2, flow, occ

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [self[ii] for ii in range(*index.indices(len(self)))]
        img1, img2, flow, occ = self._decode(index)
        if self.raw:  # decode only: prepare_batch_on_device does the rest on the GPU
            out = [torch.from_numpy(np.concatenate((img1, img2), axis=2))]
            if flow is not None:
                out.append(torch.from_numpy(flow))
            if occ is not None:
                out.append(torch.from_numpy(np.ascontiguousarray(occ)))
            return tuple(out)
        cropper = StaticCenterCrop(img1.shape[:2], self.render_size)
        img1, img2 = cropper(img1), cropper(img2)
        if self.image_size:
            img1 = resize_linear(img1, self.image_size[0], self.image_size[1])
            img2 = resize_linear(img2, self.image_size[0], self.image_size[1])
        if self.transform:
            img1, img2 = self.transform(img1), self.transform(img2)
        else:
            img1, img2 = torch.from_numpy(img1), torch.from_numpy(img2)
        images = torch.stack((img1, img2)) if self.stack_imgs else torch.cat((img1, img2))
        out = [images]
        if flow is not None:
            flow = cropper(flow)
            if self.image_size:
                flow = resize_flow(flow, self.image_size[0], self.image_size[1])
            out.append(torch.from_numpy(np.ascontiguousarray(flow.transpose(2, 0, 1))))
        if occ is not None:
            o = torch.from_numpy(np.ascontiguousarray(cropper(occ.astype(np.float32)).transpose(2, 0, 1)))  # ToTensor of a float array: no /255
            if self.image_size:
                o = F.interpolate(o.unsqueeze(0), size=tuple(self.image_size), mode="bilinear", align_corners=False, antialias=True)[0]
            o = torch.where(o > 0.5, torch.ones_like(o), torch.zeros_like(o))  # datasets.py:260-261
            out.append(o)
        return tuple(out)


class _SintelBase(_PairDataset):
    """listing of MpiSintel (flow), MpiSintelOcc (occlusions) and MpiSintelFlowOcc (both): datasets.py:113-330"""
    _layout = "expected <dstype>/<scene>/frame_XXXX.png with flow/ or occlusions/"

    def __init__(self, transform=_default_transform, root="", dstype="clean", replicates=1, image_size=None, stack_imgs=True, raw=False,
                 want_flow=True, want_occ=False):
        self.dstype = dstype
        super().__init__(transform, root, replicates, image_size, stack_imgs, raw, want_flow, want_occ)

    def _list(self, root):
        want_flow, want_occ, dstype = self.want_flow, self.want_occ, self.dstype
        flow_root, occ_root, image_root = join(root, "flow"), join(root, "occlusions"), join(root, dstype)
        flow_files = sorted(glob(join(flow_root, "*/*.flo"))) if want_flow else []
        occ_files = sorted(glob(join(occ_root, "*/*.png"))) if want_occ else []
        if want_flow and want_occ:
            assert len(occ_files) == len(flow_files), "number of occlusion maps should be equal to number of flows"
        keys = flow_files if want_flow else occ_files
        key_root = flow_root if want_flow else occ_root
        for k, file in enumerate(keys):
            fbase = file[len(key_root) + 1:]
            fprefix, fnum = fbase[:-8], int(fbase[-8:-4])
            img1 = join(image_root, fprefix + "{:04d}".format(fnum + 0) + ".png")
            img2 = join(image_root, fprefix + "{:04d}".format(fnum + 1) + ".png")
            for f in (img1, img2, file):
                assert isfile(f), "Cannot find file: {}".format(f)
            self.image_list.append([img1, img2])
            if want_flow:
                self.flow_list.append(file)
            if want_occ:
                occ_file = occ_files[k]
                assert isfile(occ_file), "Cannot find file: {}".format(occ_file)
                self.occ_list.append(occ_file)


class _ChairsBase(_PairDataset):
    """FlyingChairs / FlyingChairs2: a flat directory, consecutive sorted images form the pairs (datasets.py:465-530, 587-669)"""

    def __init__(self, transform=_default_transform, root="", replicates=1, image_size=None, stack_imgs=True, raw=False):
        super().__init__(transform, root, replicates, image_size, stack_imgs, raw, want_flow=True, want_occ=self._occ_glob is not None)

    def _list(self, root):
        images = sorted(glob(join(root, self._img_glob)))
        self.flow_list = sorted(glob(join(root, self._flow_glob)))
        self.occ_list = sorted(glob(join(root, self._occ_glob))) if self._occ_glob else []
        assert len(images) // 2 == len(self.flow_list)
        self.image_list = [[images[2 * i], images[2 * i + 1]] for i in range(len(self.flow_list))]
        assert len(self.image_list) == len(self.flow_list), "The number of image pairs should be equal to the number of optical flow"
        if self._occ_glob:
            assert len(self.occ_list) == len(self.flow_list), "number of occlusion maps should be equal to number of flows"


class FlyingChairs(_ChairsBase):
    """`XXXXX_img1.ppm`, `XXXXX_img2.ppm`, `XXXXX_flow.flo` -> (images, flow)  (datasets.py:465-530)"""
    _img_glob, _flow_glob, _occ_glob, _layout = "*.ppm", "*.flo", None, "expected *.ppm image pairs and *.flo flows in one directory"


class FlyingChairs2(_ChairsBase):
    """`XXXXXXX-img_0.png`, `-img_1.png`, `-flow_01.flo`, `-occ_01.png` -> (images, flow, occ)  (datasets.py:587-669)"""
    _img_glob, _flow_glob, _occ_glob = "*-img_*.png", "*-flow_01.flo", "*-occ_01.png"
    _layout = "expected *-img_0.png / *-img_1.png / *-flow_01.flo / *-occ_01.png in one directory"


def _variant(name, dstype, **kw):
    def __init__(self, transform=_default_transform, root="", replicates=1, image_size=None, stack_imgs=True, raw=False):
        _SintelBase.__init__(self, transform=transform, root=root, dstype=dstype, replicates=replicates, image_size=image_size,
                             stack_imgs=stack_imgs, raw=raw, **kw)
    return type(name, (_SintelBase,), {"__init__": __init__, "__doc__": f"{name} (models/data/datasets.py)"})


MpiSintelClean = _variant("MpiSintelClean", "clean", want_flow=True, want_occ=False)
MpiSintelFinal = _variant("MpiSintelFinal", "final", want_flow=True, want_occ=False)
MpiSintelCleanOcc = _variant("MpiSintelCleanOcc", "clean", want_flow=False, want_occ=True)
MpiSintelFinalOcc = _variant("MpiSintelFinalOcc", "final", want_flow=False, want_occ=True)
MpiSintelCleanFlowOcc = _variant("MpiSintelCleanFlowOcc", "clean", want_flow=True, want_occ=True)
MpiSintelFinalFlowOcc = _variant("MpiSintelFinalFlowOcc", "final", want_flow=True, want_occ=True)
DATASETS = {c.__name__: c for c in (MpiSintelClean, MpiSintelFinal, MpiSintelCleanOcc, MpiSintelFinalOcc, MpiSintelCleanFlowOcc,
                                    MpiSintelFinalFlowOcc, FlyingChairs, FlyingChairs2)}


@torch.no_grad()
def prepare_batch_on_device(frames: torch.Tensor, flow: Optional[torch.Tensor] = None, occ: Optional[torch.Tensor] = None,
                            image_size: Optional[Sequence[int]] = None, device=None, channels_last: bool = True):
    """The sample rules of the datasets as batched device operations.  frames [B, h, w, 6] uint8 (both frames channel-concatenated, as
    decoded), flow [B, h, w, 2] float32, occ [B, h, w, 1]: -> (imgs [B, 6, H, W] float32 in [-1, 1], flow [B, 2, H, W], occ [B, 1, H, W])
    after centre crop to multiples of 64, optional bilinear resize (flow rescaled by the size ratio), Normalize(0.5, 0.5) and occlusion
    binarisation at 0.5 -- datasets.py:146-192, 243-262."""
    dev = device if device is not None else frames.device
    B, h, w, _ = frames.shape
    th, tw = render_size(h, w) if (h % 64 or w % 64) else (h, w)
    ys, xs = slice((h - th) // 2, (h + th) // 2), slice((w - tw) // 2, (w + tw) // 2)
    x = frames.to(dev, non_blocking=True)[:, ys, xs].permute(0, 3, 1, 2)
    fl = flow.to(dev, non_blocking=True)[:, ys, xs].permute(0, 3, 1, 2).float() if flow is not None else None
    oc = occ.to(dev, non_blocking=True)[:, ys, xs].permute(0, 3, 1, 2).float() if occ is not None else None
    if image_size:
        H, W = int(image_size[0]), int(image_size[1])
        # the host path rounds the resized frame back to uint8 before ToTensor (cv2 on a uint8 array): same here
        x = F.interpolate(x.float(), size=(H, W), mode="bilinear", align_corners=False).round().clamp(0, 255)
        if fl is not None:
            fl = F.interpolate(fl, size=(H, W), mode="bilinear", align_corners=False)
            fl = fl * torch.tensor([W / tw, H / th], device=dev, dtype=torch.float32).view(1, 2, 1, 1)
        if oc is not None:
            oc = F.interpolate(oc, size=(H, W), mode="bilinear", align_corners=False, antialias=True)
    x = (x.float() / 255.0 - 0.5) / 0.5
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    if oc is not None:
        oc = (oc > 0.5).float()
    out = [x]
    if fl is not None:
        out.append(fl.contiguous())
    if oc is not None:
        out.append(oc.contiguous())
    return tuple(out)


class DatasetModule:
    """lightning_datamodule.py:8-69 without Lightning: same constructor arguments, `setup()`, the three dataloaders, the 80/10/10
    `random_split` with generator seed 42.  `raw=True`: samples are decoded only and `collate_to_device` / `prepare_batch_on_device`
    finish them on the GPU."""

    def __init__(self, root="", image_size=None, batch_size=32, dataset_name="MpiSintelClean", num_workers=6, overfit=False,
                 occlusion_ratio=0.3, static_occ=False, raw=False):
        self.root, self.image_size, self.batch_size, self.dataset_name = root, image_size, batch_size, dataset_name
        self.num_workers, self.overfit, self.occlusion_ratio, self.static_occ, self.raw = num_workers, overfit, occlusion_ratio, static_occ, raw
        self.datasets = {}

    def prepare_data(self):
        pass

    def setup(self, stage=None):
        if self.dataset_name not in DATASETS:
            raise ValueError("Unsupported dataset type: {}".format(self.dataset_name))
        dataset = DATASETS[self.dataset_name](root=self.root, image_size=None if self.raw else self.image_size, stack_imgs=False, raw=self.raw)
        if not self.overfit:
            n_train, n_val = ceil(0.8 * len(dataset)), ceil(0.1 * len(dataset))
            n_train = min(n_train, len(dataset))
            n_val = min(n_val, len(dataset) - n_train)
            splits = torch.utils.data.random_split(dataset, [n_train, n_val, len(dataset) - n_train - n_val],
                                                   generator=torch.Generator().manual_seed(42))
        else:
            splits = (dataset, dataset, dataset)
        self.datasets = dict(zip(("train", "val", "test"), splits))

    def _loader(self, split, shuffle):
        return DataLoader(self.datasets[split], shuffle=shuffle, batch_size=self.batch_size, num_workers=self.num_workers, pin_memory=True)

    def train_dataloader(self):
        return self._loader("train", True)

    def val_dataloader(self):
        return self._loader("val", False)

    def test_dataloader(self):
        return self._loader("test", False)

    def to_device(self, batch, device):
        """a collated batch -> the tuple the training step takes, on `device` (raw mode: crop / resize / normalise there)"""
        if self.raw:
            frames = batch[0]
            flow = batch[1] if len(batch) > 1 and batch[1].shape[-1] == 2 else None
            occ = batch[-1] if len(batch) > 1 and batch[-1].shape[-1] == 1 else None
            return prepare_batch_on_device(frames, flow, occ, self.image_size, device)
        return tuple(t.to(device, non_blocking=True) for t in batch)
