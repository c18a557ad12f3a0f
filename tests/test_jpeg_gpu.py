"""Service-side JPEG decode, GPU half (row f3): vis_jpeg_to_rgb (dequantise + integer IDCT, triangle chroma upsampling,
fixed-point YCbCr -> RGB) on the coefficients the host parser produced must equal PIL's decode of the same bytes bit for bit
- and so must the frames the client hands to the engine, whichever of the two decode paths produced them."""
import base64
import io

import numpy as np
import pytest
import torch
from PIL import Image

from test_jpeg import _smooth, cases
from vision_inspection_system_amd import jpeg as J

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,data", cases(), ids=[c[0] for c in cases()])
def test_gpu_decode_equals_pil(device, name, data):
    ref = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
    jc = J.parse(data)
    got = J.to_rgb_device(jc, device)
    assert got.dtype == torch.uint8 and tuple(got.shape) == ref.shape
    assert np.array_equal(got.cpu().numpy(), ref)


def test_gpu_decode_1024_frame_and_the_request_side_encode(device, tmp_path):
    """The benchmark's frame: 1024 x 1024 noise through the agents' a3 encode (JPEG q85, optimised tables, 4:2:0)."""
    from vision_inspection_system_amd.image_processing import decode_data_uri, encode_image_optimized
    rng = np.random.default_rng(1234)
    p = tmp_path / "frame.png"
    Image.fromarray(rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)).save(p)
    url = encode_image_optimized(p)
    ref = np.array(decode_data_uri(url))
    got = J.to_rgb_device(J.parse_data_uri(url), device).cpu().numpy()
    assert np.array_equal(got, ref)


def test_client_frames_are_the_same_on_both_decode_paths(device, monkeypatch):
    """LocalVLMClient._prepare + the upload: VIS_GPU_JPEG=1 (Huffman on the pool thread, the rest on the GPU) and
    VIS_GPU_JPEG=0 (PIL) give the engine identical frames and identical token ids; a progressive JPEG (not handled by the
    parser) silently takes the PIL path."""
    from vision_inspection_system_amd import client as CL, config as C, hip
    C.set_config(C.Config(vlm_inspector_provider="mi355x", vlm_inspector_model="synthetic:tiny"))
    try:
        cl = CL.LocalVLMClient()
        lm = CL.get_model("synthetic:tiny", None)
        img = Image.fromarray(_smooth(np.random.default_rng(4), 220, 340))
        for kw in (dict(quality=85), dict(quality=85, progressive=True)):
            b = io.BytesIO()
            img.save(b, format="JPEG", **kw)
            url = "data:image/jpeg;base64," + base64.b64encode(b.getvalue()).decode()
            msgs = [{"role": "user", "content": [{"type": "text", "text": "Inspect."}, {"type": "image_url", "image_url": {"url": url}}]}]
            out = {}
            for flag in ("1", "0"):
                monkeypatch.setenv("VIS_GPU_JPEG", flag)
                ids, frames = cl._prepare(lm, msgs)
                f, (th, tw) = frames[0]
                if flag == "1" and not kw.get("progressive"):
                    assert isinstance(f, J.JpegCoeffs)
                else:
                    assert isinstance(f, np.ndarray)
                out[flag] = (ids, hip.resize_rgb(CL._frame_to_device(f, lm.engine.device), th, tw))
            assert out["1"][0] == out["0"][0] and torch.equal(out["1"][1], out["0"][1])
    finally:
        C.set_config(None)
