"""Middlebury `.flo` reader / writer, end-point error and the dataset-side shape rules of the reference
(models/data/utils/flow_utils.py:10-68, 179-232; models/data/datasets.py:146-150, 178-179).  Host-side numpy: these sit either
side of the GPU hot path, not on it."""
from __future__ import annotations

import numpy as np

FLO_MAGIC = np.float32(202021.25)  # TAG_CHAR, flow_utils.py:7
UNKNOWN_FLOW_THRESH = 1e7


def read_flow(path: str) -> np.ndarray:
    """-> [h, w, 2] float32 (u, v interleaved on disk); raises on a bad magic number (the reference prints and returns None)."""
    with open(path, "rb") as f:
        raw = f.read()
    return decode_flo(raw)


def decode_flo(raw: bytes) -> np.ndarray:
    if len(raw) < 12 or np.frombuffer(raw, np.float32, 1)[0] != FLO_MAGIC:
        raise ValueError("Magic number incorrect. Invalid .flo file")
    w, h = (int(v) for v in np.frombuffer(raw, np.int32, 2, offset=4))
    data = np.frombuffer(raw, np.float32, 2 * w * h, offset=12)
    return data.reshape(h, w, 2).copy()


def encode_flo(uv: np.ndarray, v: np.ndarray | None = None) -> bytes:
    """bytes of a .flo file: magic, int32 width, int32 height, then rows of interleaved float32 (u, v)."""
    if v is None:
        if uv.ndim != 3 or uv.shape[2] != 2:
            raise ValueError("flow must be [h, w, 2]")
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u = uv
    if u.shape != v.shape:
        raise ValueError("u and v shapes differ")
    h, w = u.shape
    inter = np.empty((h, w, 2), dtype=np.float32)
    inter[..., 0], inter[..., 1] = u, v
    return FLO_MAGIC.tobytes() + np.int32(w).tobytes() + np.int32(h).tobytes() + inter.tobytes()


def save_flow(path: str, uv: np.ndarray, v: np.ndarray | None = None) -> None:
    with open(path, "wb") as f:
        f.write(encode_flo(uv, v))


def flow_error(tu, tv, u, v, occ=None) -> float:
    """Average end-point error; pixels whose ground truth exceeds 1e7 in magnitude are zeroed in all four maps first;
    with `occ` (1 = occluded) only visible pixels count (flow_utils.py:179-232)."""
    tu, tv, u, v = (np.array(a, dtype=np.float64) for a in (tu, tv, u, v))
    bad = (np.abs(tu) > UNKNOWN_FLOW_THRESH) | (np.abs(tv) > UNKNOWN_FLOW_THRESH)
    for a in (tu, tv, u, v):
        a[bad] = 0
    epe = np.sqrt((tu - u) ** 2 + (tv - v) ** 2)
    if occ is not None:
        epe = epe.reshape(-1)[(1 - np.asarray(occ)).astype(bool).reshape(-1)]
    return float(epe.mean())


def render_size(height: int, width: int) -> tuple:
    """centre-crop target of the reference's datasets: both sides floored to a multiple of 64 (datasets.py:146-150)"""
    return (height // 64) * 64, (width // 64) * 64


def normalize_image(img_uint8: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize(0.5, 0.5): [h, w, 3] uint8 -> [3, h, w] float32 in [-1, 1] (lightning_datamodule.py:20-23)"""
    return (img_uint8.astype(np.float32) / 255.0 - 0.5).transpose(2, 0, 1) / 0.5
