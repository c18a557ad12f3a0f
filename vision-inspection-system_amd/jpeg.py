"""Service-side JPEG decode split between a host core and the GPU.

``parse`` (any thread of the ingest pool): marker parsing + Huffman decoding in C (csrc/jpeg_host.c ->
libvis_jpeg_host.so) - quantised DCT coefficients, no pixels.  ``to_rgb_device`` (the thread that owns the stream):
upload of the coefficients (the size of the RGB frame it replaces) and two HIP kernels - dequantise + integer inverse
DCT per 8x8 block, then triangle chroma upsampling + fixed-point YCbCr -> RGB - bit-exact with libjpeg-turbo's default
decoder, i.e. with what ``PIL.Image.open(...).convert("RGB")`` returns (tests/test_jpeg.py, tests/test_jpeg_gpu.py).
JPEG flavours the C parser declines (progressive, CMYK, ...) return None from ``parse``: the caller uses PIL for them.
``VIS_GPU_JPEG=0`` switches the whole path off (A/B)."""
from __future__ import annotations

import base64
import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Info(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("ncomp", ctypes.c_int),
                ("hs", ctypes.c_int * 3), ("vs", ctypes.c_int * 3), ("bw", ctypes.c_int * 3), ("bh", ctypes.c_int * 3),
                ("dw", ctypes.c_int * 3), ("dh", ctypes.c_int * 3), ("mcus_x", ctypes.c_int), ("mcus_y", ctypes.c_int),
                ("restart_interval", ctypes.c_int), ("total_blocks", ctypes.c_int), ("sos_offset", ctypes.c_int),
                ("qt", (ctypes.c_uint16 * 64) * 3), ("dc_tab", ctypes.c_uint8 * 3), ("ac_tab", ctypes.c_uint8 * 3),
                ("huff_counts", ((ctypes.c_uint8 * 16) * 4) * 2), ("huff_syms", ((ctypes.c_uint8 * 256) * 4) * 2),
                ("huff_present", (ctypes.c_uint8 * 4) * 2)]


def enabled() -> bool:
    return os.environ.get("VIS_GPU_JPEG", "1") != "0"


def host_lib():
    """libvis_jpeg_host.so (built by `make` / __graft_entry__.build()); raises if it is missing or does not match."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "csrc", "libvis_jpeg_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` (or "
                               "__graft_entry__.build()); VIS_GPU_JPEG=0 decodes every image with PIL instead")
        lib = ctypes.CDLL(path)
        lib.vis_jpeg_info_size.restype = ctypes.c_int
        if lib.vis_jpeg_info_size() != ctypes.sizeof(_Info):
            raise RuntimeError("libvis_jpeg_host.so does not match include/vis_jpeg_host.h (rebuild: make -C csrc)")
        lib.vis_jpeg_probe.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(_Info)]
        lib.vis_jpeg_probe.restype = ctypes.c_int
        lib.vis_jpeg_decode_coeffs.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(_Info), ctypes.c_void_p]
        lib.vis_jpeg_decode_coeffs.restype = ctypes.c_int
        _LIB = lib
    return _LIB


@dataclass
class JpegCoeffs:
    """Entropy-decoded image: geometry + quantisation tables + [total_blocks, 64] int16 coefficients (host memory)."""
    width: int
    height: int
    ncomp: int
    hs: tuple
    vs: tuple
    bw: tuple
    bh: tuple
    dw: tuple
    dh: tuple
    qt: np.ndarray          # [3, 64] uint16, natural order
    coeffs: np.ndarray      # [total_blocks, 64] int16, natural order, quantised
    pinned: object = None   # the page-locked torch tensor `coeffs` is a view of (when the Huffman decode wrote straight into one)

    @property
    def size(self):         # PIL convention: (width, height)
        return (self.width, self.height)

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k in ("width", "height", "ncomp", "hs", "vs", "bw", "bh", "dw", "dh", "qt")}


def parse(data: bytes) -> Optional[JpegCoeffs]:
    """Huffman-decode a JPEG byte string; None when the flavour is not handled here or the data is damaged (the caller
    then lets PIL decide what to do with it, like before)."""
    lib = host_lib()
    info = _Info()
    if lib.vis_jpeg_probe(data, len(data), ctypes.byref(info)) != 0:
        return None
    # On a GPU process the Huffman decoder writes straight into page-locked memory (this pool thread allocates it; PyTorch's
    # caching host allocator recycles the blocks): the engine thread then only queues an async copy instead of first copying
    # 3 MB into a staging buffer - 64-image seam: 6.9 ms of launch-thread time per image for upload + IDCT / resize launches
    # (profiles/r04_seam64_host_timeline.json).  CPU-only processes (tests, dry runs) never initialise CUDA: plain numpy.
    pinned = None
    if _pin_coeffs():
        import torch
        try:
            pinned = torch.empty((info.total_blocks, 64), dtype=torch.int16, pin_memory=True)
        except RuntimeError:
            pinned = None
    coeffs = pinned.numpy() if pinned is not None else np.empty((info.total_blocks, 64), dtype=np.int16)
    if lib.vis_jpeg_decode_coeffs(data, len(data), ctypes.byref(info), coeffs.ctypes.data) != 0:   # releases the GIL
        return None
    n = info.ncomp
    t = lambda a: tuple(int(a[i]) for i in range(n))
    qt = np.array([[info.qt[c][k] for k in range(64)] for c in range(3)], dtype=np.uint16)
    return JpegCoeffs(info.width, info.height, n, t(info.hs), t(info.vs), t(info.bw), t(info.bh), t(info.dw), t(info.dh),
                      qt, coeffs, pinned)


def _pin_coeffs() -> bool:
    if os.environ.get("VIS_PIN_COEFFS", "1") == "0":
        return False
    import torch
    return torch.cuda.is_initialized()


_QT_DEVICE: dict = {}       # quantisation tables already on the device (the agents' q85 tables never change)


def parse_data_uri(url: str) -> Optional[JpegCoeffs]:
    """``data:image/jpeg;base64,...`` -> JpegCoeffs, or None (not a base64 JPEG data URI / unsupported flavour)."""
    if not url.startswith("data:image/jp"):
        return None
    try:
        header, b64 = url.split(",", 1)
    except ValueError:
        return None
    if ";base64" not in header:
        return None
    return parse(base64.b64decode(b64))


def to_rgb_device(jc: JpegCoeffs, device):
    """H2D of the coefficients + the two HIP kernels -> uint8 [H, W, 3] on ``device`` (current stream)."""
    import torch
    from . import hip
    coeffs = hip.upload(jc.pinned if jc.pinned is not None else jc.coeffs, device)   # upload stream: the host does not wait
    key = (str(device), jc.qt.tobytes())
    qt = _QT_DEVICE.get(key)
    if qt is None:
        if len(_QT_DEVICE) > 64:
            _QT_DEVICE.clear()
        qt = _QT_DEVICE[key] = torch.from_numpy(jc.qt.astype(np.int32)).to(device)
    return hip.jpeg_to_rgb(coeffs, qt, jc)
