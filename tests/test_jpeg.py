"""Service-side JPEG decode, host half + oracle (row f3): csrc/jpeg_host.c (marker parsing + Huffman decoding) followed by
the numpy restatement of libjpeg-turbo's islow IDCT / fancy upsampling / YCbCr->RGB (oracle/jpeg_ref.py) must reproduce
PIL's decode of the same bytes BIT FOR BIT - PIL links libjpeg-turbo and stands for the reference's service side (the
reference itself only writes these JPEGs: src/agents/vlm_inspector.py:46-88).  This pins the oracle the HIP kernels are
checked against (tests/test_jpeg_gpu.py)."""
import ctypes
import io
import os
import re

import numpy as np
import pytest
from PIL import Image

from vision_inspection_system_amd import jpeg as J

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _smooth(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 9.0 + yy / 13.0), 128 + 90 * np.cos(xx / 7.0), 128 + 80 * np.sin(yy / 5.0 + 1)], -1)
    return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


def jpeg_bytes(img, **kw):
    b = io.BytesIO()
    try:
        img.save(b, format="JPEG", optimize=True, **kw)
    except OSError:            # PIL's optimise pass can overflow its buffer on tiny images ("Suspension not allowed here")
        b = io.BytesIO()
        img.save(b, format="JPEG", **kw)
    return b.getvalue()


def cases():
    rng = np.random.default_rng(0)
    out = []
    for (h, w) in [(37, 53), (64, 64), (1, 1), (8, 8), (17, 16), (100, 3), (3, 100), (256, 300), (2, 5), (5, 2), (16, 33), (33, 17)]:
        for sub in (0, 1, 2):                                  # 4:4:4, 4:2:2, 4:2:0
            for noise in (False, True):
                a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if noise else _smooth(rng, h, w)
                out.append((f"{h}x{w} sub{sub} {'noise' if noise else 'smooth'}", jpeg_bytes(Image.fromarray(a), quality=85, subsampling=sub)))
        out.append((f"{h}x{w} grey", jpeg_bytes(Image.fromarray(_smooth(rng, h, w)[..., 0]), quality=85)))
    big = _smooth(rng, 480, 640)
    for q in (5, 30, 60, 95, 100):
        out.append((f"480x640 q{q}", jpeg_bytes(Image.fromarray(big), quality=q)))
    out.append(("480x640 restart every 3 MCU rows", jpeg_bytes(Image.fromarray(big), quality=85, restart_marker_rows=3)))
    out.append(("37x53 restart every 2 blocks", jpeg_bytes(Image.fromarray(big[:37, :53]), quality=85, restart_marker_blocks=2)))
    return out


def test_host_library_matches_its_header():
    lib = J.host_lib()
    header = open(os.path.join(ROOT, "include", "vis_jpeg_host.h")).read()
    for name in sorted(set(re.findall(r"\b(vis_jpeg_[a-z_]+)\s*\(", header))):
        assert hasattr(lib, name), name
    assert lib.vis_jpeg_info_size() == ctypes.sizeof(J._Info)


@pytest.mark.parametrize("name,data", cases(), ids=[c[0] for c in cases()])
def test_host_huffman_plus_oracle_equals_pil(name, data):
    from oracle import jpeg_ref as R
    ref = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
    jc = J.parse(data)
    assert jc is not None, "a baseline JPEG written by PIL must be handled"
    assert jc.size == (ref.shape[1], ref.shape[0])
    got = R.decode(jc.as_dict(), jc.coeffs)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_request_side_encode_is_the_supported_flavour(tmp_path):
    """What the agents send (a3: JPEG q85, optimised tables, 4:2:0) is what the split decoder takes."""
    from vision_inspection_system_amd.image_processing import encode_image_optimized
    p = tmp_path / "part.png"
    Image.fromarray(_smooth(np.random.default_rng(1), 300, 420)).save(p)
    url = encode_image_optimized(p)
    jc = J.parse_data_uri(url)
    assert jc is not None and jc.size == (420, 300) and jc.ncomp == 3 and (jc.hs[0], jc.vs[0]) == (2, 2)
    assert J.parse_data_uri("vis-frame:1") is None and J.parse_data_uri("data:image/png;base64,AAAA") is None


def test_flavours_the_parser_declines_and_damaged_data():
    rng = np.random.default_rng(2)
    img = Image.fromarray(_smooth(rng, 64, 80))
    b = io.BytesIO(); img.save(b, format="JPEG", quality=85, progressive=True)
    assert J.parse(b.getvalue()) is None                         # progressive -> PIL
    b = io.BytesIO(); img.convert("CMYK").save(b, format="JPEG", quality=85)
    assert J.parse(b.getvalue()) is None                         # 4 components -> PIL
    good = jpeg_bytes(img, quality=85)
    assert J.parse(good) is not None
    assert J.parse(good[:len(good) // 2]) is not None or True    # a truncated scan decodes zeros or is refused: never crashes
    assert J.parse(b"not a jpeg at all") is None
    assert J.parse(good[:20]) is None                            # header cut short
    for cut in (2, 10, 100, 200, 300, len(good) - 3):
        J.parse(good[:cut])                                      # no crash, no out-of-bounds read (run under ASan in CI by hand)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x5A
    J.parse(bytes(bad))                                          # bit flip inside the scan: any result, no crash
