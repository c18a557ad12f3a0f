"""Row f4: image-quality pre-check.  PARITY UNPINNED against the reference's cv2 calls (OpenCV absent, no fixtures):
the GPU kernel is checked against a numpy statement of OpenCV's published 8-bit algorithm, and the scoring arithmetic
against hand-computed values of the reference's formulas (src/safety/image_quality.py:105-168)."""
import numpy as np
import pytest
from PIL import Image

from oracle.image_quality_ref import gray_u8, laplacian_reflect101
from vision_inspection_system_amd.image_quality import ImageQualityAssessment, check_image_quality


def test_numpy_statement_basics():
    rgb = np.zeros((4, 5, 3), dtype=np.uint8)
    rgb[..., 0], rgb[..., 1], rgb[..., 2] = 255, 255, 255
    assert (gray_u8(rgb) == 255).all()                      # coefficients sum to 2^14
    assert (gray_u8(np.zeros((2, 2, 3), np.uint8)) == 0).all()
    g = np.arange(12, dtype=np.int64).reshape(3, 4) ** 2
    lap = laplacian_reflect101(g)
    assert lap[1, 1] == g[0, 1] + g[2, 1] + g[1, 0] + g[1, 2] - 4 * g[1, 1]
    assert lap[0, 0] == 2 * g[1, 0] + 2 * g[0, 1] - 4 * g[0, 0]   # reflect-101: index -1 -> 1


def test_scores_follow_the_reference_formulas():
    q = ImageQualityAssessment()
    assert q._sharpness_score(50.0) == pytest.approx(0.25) and q._sharpness_score(300.0) == pytest.approx(1.0)
    assert q._sharpness_score(200.0) == pytest.approx(0.75)
    assert q._brightness_score(125.0) == pytest.approx(1.0) and q._brightness_score(30.0) == pytest.approx(0.7)
    assert q._brightness_score(15.0) == pytest.approx(0.3) and q._brightness_score(237.5) == pytest.approx(0.6)
    assert q._resolution_score(50, 400, 20000) == 0.3 and q._resolution_score(100, 99, 9900) == 0.3
    assert q._resolution_score(640, 480, 307200) == pytest.approx(0.1536) and q._resolution_score(2000, 2000, 4000000) == 1.0
    bad = q.assess_quality("/nonexistent/file.png")
    assert bad["quality_passed"] is False and "Failed to load image" in bad["error"]


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (37, 53), (480, 640), (1024, 1024)])
def test_gpu_stats_equal_numpy_statement(h, w):
    import torch
    from vision_inspection_system_amd import hip
    rng = np.random.default_rng(h * 1000 + w)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    g = gray_u8(rgb)
    lap = laplacian_reflect101(g) if min(h, w) > 1 else None
    sg, sl, sq = hip.image_stats(torch.from_numpy(rgb).to("cuda:0"))
    assert sg == int(g.sum())
    if lap is not None:
        assert sl == int(lap.sum()) and sq == int((lap * lap).sum())       # exact integers


@pytest.mark.gpu
def test_assess_and_node_on_gpu(tmp_path):
    rng = np.random.default_rng(9)
    sharp = tmp_path / "sharp.png"
    Image.fromarray(rng.integers(0, 256, (1200, 1700, 3), dtype=np.uint8)).save(sharp)
    flat = tmp_path / "flat.png"
    Image.fromarray(np.full((300, 300, 3), 10, dtype=np.uint8)).save(flat)
    a = ImageQualityAssessment().assess_quality(sharp)
    assert a["quality_passed"] and a["sharpness"]["score"] == 1.0 and a["resolution"]["score"] == 1.0
    rgb = np.array(Image.open(sharp).convert("RGB"))
    lap = laplacian_reflect101(gray_u8(rgb)).astype(np.float64)
    assert a["sharpness"]["laplacian_variance"] == pytest.approx(lap.var(), rel=1e-9)
    assert a["brightness"]["mean_value"] == round(float(gray_u8(rgb).mean()), 1)
    b = ImageQualityAssessment().assess_quality(flat)
    assert not b["quality_passed"] and b["sharpness"]["laplacian_variance"] == 0.0 and b["brightness"]["mean_value"] == 10.0
    st = check_image_quality({"image_path": [str(flat), str(sharp)]})
    assert st["current_step"] == "quality_check" and st["low_quality_image"] is True and st["image_quality"] == b
    st2 = check_image_quality({"image_path": "/nonexistent.png"})
    assert st2["image_quality"]["quality_passed"] is False
