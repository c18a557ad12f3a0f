"""Host-side image handling on both sides of the boundary.

* ``encode_image_optimized`` mirrors the reference's request-side step a3
  (src/agents/vlm_inspector.py:46-88, auditor copy src/agents/vlm_auditor.py:85-108):
  thumbnail (LANCZOS) -> RGB -> JPEG q85 (q60 above 5 MB, error above 10 MB) -> base64 data URI.
* ``decode_data_uri`` + ``resize_for_model`` are the service-side counterpart the local backend
  now owns: decode the JPEG, RGB, ``smart_resize`` to multiples of 28 inside the pixel budget,
  bicubic resample (PIL) - the geometry of TF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:57-84.
  Rescale / normalise / patchify happen on the GPU (csrc/misc.hip, vis_patchify_u8).
"""
from __future__ import annotations

import base64
import collections
import io
import math
import os
import threading
from concurrent.futures import Future
from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 28 * 28 * 1280) -> Tuple[int, int]:
    """Target (h, w): multiples of ``factor``, area within [min_pixels, max_pixels], aspect kept."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError(
            f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


# Both agents encode the same file (Inspector, then Auditor: nodes.py:128,:230 of the reference), and for every image
# that is no larger than the Auditor's 1024 px limit and is not in mode LA their bytes are identical (same thumbnail
# decision, same conversion, same JPEG settings).  The second encode is the larger part of the host work per image at
# batch rates (PNG inflate + JPEG q85 with optimize=True: ~0.15 s of one core for a 1024 x 1024 frame), so results are
# shared: key = file identity + every input of the pipeline below; a concurrent second request waits for the first.
_ENC_LOCK = threading.Lock()
_ENC_CACHE: "collections.OrderedDict[tuple, Future]" = collections.OrderedDict()
_ENC_BYTES = [0]


def _encode_cache_limit() -> int:
    return int(float(os.environ.get("VIS_ENCODE_CACHE_MB", "128")) * (1 << 20))


def clear_encode_cache() -> None:
    with _ENC_LOCK:
        _ENC_CACHE.clear()
        _ENC_BYTES[0] = 0


def encode_image_optimized(image_path: Union[str, Path], max_size: int = 2048, convert_la: bool = True,
                           enforce_limit: bool = True, logger=None) -> str:
    """Request-side encode, byte-compatible with the reference's Inspector (``convert_la=True,
    enforce_limit=True``) and Auditor (``max_size=1024, convert_la=False, enforce_limit=False``)."""
    img = Image.open(image_path)
    shrink = max(img.size) > max_size
    modes = ("RGBA", "P", "LA") if convert_la else ("RGBA", "P")
    key = fut = None
    limit = _encode_cache_limit()
    if limit > 0:
        try:
            st = os.stat(image_path)
            key = (os.path.abspath(str(image_path)), st.st_mtime_ns, st.st_size, max_size if shrink else 0, img.mode in modes)
        except OSError:
            key = None
    if key is not None:
        with _ENC_LOCK:
            fut = _ENC_CACHE.get(key)
            owner = fut is None
            if owner:
                fut = _ENC_CACHE[key] = Future()
            else:
                _ENC_CACHE.move_to_end(key)
        if not owner:
            uri, payload_size = fut.result()          # raises what the first encode raised
            if enforce_limit and payload_size > 10_000_000:
                raise ValueError(f"Image too large even after optimization: {payload_size} bytes")
            return uri
    try:
        uri, payload_size = _encode(img, max_size, shrink, modes, logger)
    except BaseException as e:
        if fut is not None:
            with _ENC_LOCK:
                _ENC_CACHE.pop(key, None)
            fut.set_exception(e)
        raise
    if fut is not None:
        fut.set_result((uri, payload_size))
        with _ENC_LOCK:
            _ENC_BYTES[0] += len(uri)
            while _ENC_BYTES[0] > limit and _ENC_CACHE:
                k, f = next(iter(_ENC_CACHE.items()))
                if not f.done():
                    break
                _ENC_CACHE.pop(k)
                if f.exception() is None:
                    _ENC_BYTES[0] -= len(f.result()[0])
    if enforce_limit and payload_size > 10_000_000:
        raise ValueError(f"Image too large even after optimization: {payload_size} bytes")
    return uri


def _encode(img: Image.Image, max_size: int, shrink: bool, modes: tuple, logger) -> Tuple[str, int]:
    original_size = img.size
    if shrink:
        img.thumbnail((max_size, max_size), Image.Resampling.LANCZOS)
        if logger:
            logger.debug(f"Resized image from {original_size} to {img.size}")
    if img.mode in modes:
        img = img.convert("RGB")
    buffer = io.BytesIO()
    img.save(buffer, format="JPEG", quality=85, optimize=True)
    if buffer.tell() > 5_000_000:
        buffer = io.BytesIO()
        img.save(buffer, format="JPEG", quality=60, optimize=True)
    payload_size = buffer.tell()
    return "data:image/jpeg;base64," + base64.b64encode(buffer.getvalue()).decode(), payload_size


# ---- optional: no JPEG round trip.  The reference always ships the image to the service as a JPEG-q85
# data URI; when the "service" is the engine in this process, encode + base64 + decode (~0.2 s of one core per
# 1024 x 1024 frame and agent) buy nothing.  With VIS_DIRECT_FRAMES=1 the agents apply the same open / thumbnail / mode
# conversion as a3 and hand the RGB pixels over under a process-local ``vis-frame:<id>`` URL.  OFF by default: the model then
# sees the un-quantised pixels, not the JPEG-decoded ones the reference's service sees.
_FRAMES: "collections.OrderedDict[str, np.ndarray]" = collections.OrderedDict()
_FRAMES_LOCK = threading.Lock()
_FRAME_SEQ = [0]


def direct_frames_enabled() -> bool:
    return os.environ.get("VIS_DIRECT_FRAMES", "0") == "1"


def frame_url_for(image_path: Union[str, Path], max_size: int = 2048, convert_la: bool = True, logger=None) -> str:
    """a3 without the JPEG: open, thumbnail (LANCZOS) above ``max_size``, RGB; returns a ``vis-frame:`` URL that
    ``decode_data_uri`` resolves inside this process.  The agent releases the handle when its request is over
    (``release_frames``); as a backstop only the newest ``frame_cap()`` handles are kept."""
    img = Image.open(image_path)
    if max(img.size) > max_size:
        img.thumbnail((max_size, max_size), Image.Resampling.LANCZOS)
    arr = np.array(img.convert("RGB"), dtype=np.uint8)
    with _FRAMES_LOCK:
        _FRAME_SEQ[0] += 1
        url = f"vis-frame:{_FRAME_SEQ[0]}"
        _FRAMES[url] = arr
        cap = frame_cap()
        while len(_FRAMES) > cap:
            _FRAMES.popitem(last=False)
    return url


def frame_cap() -> int:
    """Handles kept at most: the batch seam holds two groups (the one on the GPU and the one encoded ahead) of
    VIS_MAX_BATCH images for two agents, and a retried image re-encodes - so 4 x group x agents, never fewer than 256."""
    return max(256, 8 * max(1, int(os.environ.get("VIS_MAX_BATCH", "64"))))


def release_frames(messages) -> None:
    """Drop the ``vis-frame:`` handles a finished request carried (a frame is ~3 MB; nothing else ever frees it)."""
    if not isinstance(messages, list):
        return
    urls = []
    for m in messages:
        content = m.get("content") if isinstance(m, dict) else None
        if isinstance(content, list):
            for part in content:
                if isinstance(part, dict) and part.get("type") == "image_url":
                    u = part["image_url"]["url"] if isinstance(part.get("image_url"), dict) else part.get("image_url")
                    if isinstance(u, str) and u.startswith("vis-frame:"):
                        urls.append(u)
    if urls:
        with _FRAMES_LOCK:
            for u in urls:
                _FRAMES.pop(u, None)


def decode_data_uri(url: str) -> Image.Image:
    """``data:image/...;base64,XXXX`` -> PIL RGB image.  Remote URLs are refused (no network)."""
    if url.startswith("vis-frame:"):
        with _FRAMES_LOCK:
            arr = _FRAMES.get(url)
        if arr is None:
            raise ValueError(f"unknown or expired frame handle {url}")
        return Image.fromarray(arr)
    if not url.startswith("data:"):
        raise ValueError("local backend accepts only data: URIs for images (no remote fetch)")
    try:
        header, b64 = url.split(",", 1)
    except ValueError:
        raise ValueError("malformed data URI")
    if ";base64" not in header:
        raise ValueError("data URI must be base64 encoded")
    img = Image.open(io.BytesIO(base64.b64decode(b64)))
    img.load()
    return img.convert("RGB")


def resize_for_model(img: Image.Image, patch: int = 14, merge: int = 2, min_pixels: int = 56 * 56,
                     max_pixels: int = 28 * 28 * 1280) -> np.ndarray:
    """RGB PIL image -> uint8 [H, W, 3] with H, W multiples of patch*merge (bicubic, like the HF PIL backend)."""
    img = img.convert("RGB")
    w, h = img.size
    th, tw = smart_resize(h, w, patch * merge, min_pixels, max_pixels)
    if (th, tw) != (h, w):
        img = img.resize((tw, th), resample=Image.Resampling.BICUBIC)
    return np.array(img, dtype=np.uint8)  # writable copy


# ----------------------------------------------------------------------------- resampling tables (row f3)
_PRECISION_BITS = 32 - 8 - 2  # Pillow's fixed-point format for 8-bit resampling


def _bicubic(x: float) -> float:
    """Keys cubic, a = -0.5 (Pillow's BICUBIC filter, support 2.0), same operation order as the C code."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x: float) -> float:
    """Triangle filter (Pillow's BILINEAR, support 1.0) - what the HF mllama image processor resamples with."""
    if x < 0.0:
        x = -x
    if x < 1.0:
        return 1.0 - x
    return 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def resample_coeffs(in_size: int, out_size: int, kind: str = "bicubic") -> Tuple[np.ndarray, np.ndarray]:
    """Per-output-coordinate windows and fixed-point weights of Pillow's 8-bit bicubic / bilinear resampling
    (Pillow 12.2 src/libImaging/Resample.c, precompute_coeffs + normalize_coeffs_8bpc; restated in float64 with
    the same operation order, so the integers are identical).  Returns (bounds int32 [out,2] = {first, taps},
    kk int32 [out, ksize])."""
    filt, support_f = _FILTERS[kind]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support_f * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    one = float(1 << _PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx, 0] = xmin
        bounds[xx, 1] = xmax
    return bounds, kk


def resample_u8_reference(img: np.ndarray, out_h: int, out_w: int, kind: str = "bicubic") -> np.ndarray:
    """numpy statement of the two integer passes the GPU kernels perform (host logic check, tests only)."""
    in_h, in_w, _ = img.shape

    def one_pass(a: np.ndarray, in_size: int, out_size: int) -> np.ndarray:   # resamples axis 1
        if in_size == out_size:
            return a
        b, k = resample_coeffs(in_size, out_size, kind)
        out = np.empty((a.shape[0], out_size, a.shape[2]), dtype=np.uint8)
        for o in range(out_size):
            x0, n = int(b[o, 0]), int(b[o, 1])
            acc = (a[:, x0:x0 + n, :].astype(np.int64) * k[o, :n].astype(np.int64)[None, :, None]).sum(axis=1)
            out[:, o, :] = np.clip((acc + (1 << (_PRECISION_BITS - 1))) >> _PRECISION_BITS, 0, 255)
        return out

    tmp = one_pass(img, in_w, out_w)
    return one_pass(tmp.transpose(1, 0, 2), in_h, out_h).transpose(1, 0, 2).copy()


def target_size(img_size: Tuple[int, int], patch: int = 14, merge: int = 2, min_pixels: int = 56 * 56,
                max_pixels: int = 28 * 28 * 1280) -> Tuple[int, int]:
    """(w, h) of a PIL image -> (target_h, target_w) of the model frame."""
    w, h = img_size
    return smart_resize(h, w, patch * merge, min_pixels, max_pixels)


def grid_of(frame_u8: np.ndarray, patch: int = 14) -> Tuple[int, int, int]:
    h, w, _ = frame_u8.shape
    return (1, h // patch, w // patch)
