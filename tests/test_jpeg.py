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
    assert J.parse(b"not a jpeg at all") is None
    assert J.parse(good[:20]) is None                            # header cut short
    # A scan that ends early must NOT come back as a picture (zero bits for the missing half = a half-grey frame the
    # model would then "inspect"): the parser refuses it and the caller's PIL path raises "image file is truncated"
    # (ADVICE r2, medium).  Every cut inside the entropy-coded data, down to the last byte before EOI.
    sos = good.index(b"\xff\xda")
    for cut in sorted({sos + 16, len(good) // 2, len(good) - 200, len(good) - 10, len(good) - 3, len(good) - 2}):
        assert J.parse(good[:cut]) is None, cut
        with pytest.raises(OSError):
            Image.open(io.BytesIO(good[:cut])).convert("RGB")
    assert J.parse(good[:-2]) is None                            # complete scan, EOI missing: PIL decides (it pads and warns)
    assert J.parse(good[:-2] + b"\x00\x00\x00" + good[-2:]) is None       # extra bytes between the scan and EOI
    mid = (sos + len(good)) // 2
    assert J.parse(good[:mid] + b"\xff\xd9" + good[mid:]) is None        # an EOI in the middle of the scan
    assert J.parse(good[:mid] + b"\xff\xd3" + good[mid:]) is None        # an RSTn nobody announced (no DRI)
    rst = jpeg_bytes(img, quality=85, restart_marker_blocks=2)
    assert J.parse(rst) is not None
    k = rst.index(b"\xff\xd1")
    assert J.parse(rst[:k] + b"\xff\xd5" + rst[k + 2:]) is None          # restart markers out of sequence
    assert J.parse(rst[:k] + rst[k + 2:]) is None                        # a restart marker lost
    assert J.parse(rst[:k - 1] + rst[k:]) is None or True                # a data byte lost in front of it: refused or not, no crash
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x5A
    J.parse(bytes(bad))                                          # bit flip inside the scan: any result, no crash


def test_truncated_request_fails_on_both_decode_paths():
    """The client hands the engine identical frames on both decode paths - and refuses the same damaged requests."""
    import base64
    from vision_inspection_system_amd.image_processing import decode_data_uri
    good = jpeg_bytes(Image.fromarray(_smooth(np.random.default_rng(5), 96, 128)), quality=85)
    url = "data:image/jpeg;base64," + base64.b64encode(good[:len(good) // 2]).decode()
    assert J.parse_data_uri(url) is None
    with pytest.raises(Exception):
        np.array(decode_data_uri(url))


def test_dc_predictor_leaving_int16_is_refused():
    """A crafted scan whose DC differences only ever add up: the predictor is range-checked (signed overflow is undefined
    behaviour in C; no valid stream leaves the int16 range)."""
    import struct
    # one 8x8 grey JPEG frame header declaring 4096 x 8 pixels (512 blocks); DC table: one 1-bit code for category 11,
    # AC table: one 1-bit code for EOB; scan = 512 x [0 | 11111111111 (+2047) | 0]
    def seg(m, body):
        return b"\xff" + bytes([m]) + struct.pack(">H", len(body) + 2) + body
    dqt = seg(0xDB, bytes([0]) + bytes([1] * 64))
    sof = seg(0xC0, bytes([8]) + struct.pack(">HH", 8, 4096) + bytes([1, 1, 0x11, 0]))
    dht = seg(0xC4, bytes([0x00, 1] + [0] * 15 + [11]) + bytes([0x10, 1] + [0] * 15 + [0]))
    sos = seg(0xDA, bytes([1, 1, 0x00, 0, 63, 0]))
    bits = ("0" + "1" * 11 + "0") * 512
    bits += "1" * (-len(bits) % 8)
    scan = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)).replace(b"\xff", b"\xff\x00")
    data = b"\xff\xd8" + dqt + sof + dht + sos + scan + b"\xff\xd9"
    assert J.parse(data) is None
    ok_bits = ("0" + "1" * 11 + "0") + ("0" + "0" + "1" * 10 + "0") * 511        # +2047, then -2047 + ... stays in range
    ok_bits = ("0" + "1" * 11 + "0" + "0" + "0" * 11 + "0") * 256                  # +2047, -2047 alternating
    ok_bits += "1" * (-len(ok_bits) % 8)
    scan = bytes(int(ok_bits[i:i + 8], 2) for i in range(0, len(ok_bits), 8)).replace(b"\xff", b"\xff\x00")
    jc = J.parse(b"\xff\xd8" + dqt + sof + dht + sos + scan + b"\xff\xd9")
    assert jc is not None and jc.coeffs[0, 0] == 2047 and jc.coeffs[1, 0] == 0
