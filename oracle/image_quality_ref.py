"""ORACLE (test infrastructure only - never imported by the product path): numpy statement of the two OpenCV 8-bit
operations behind the reference's image-quality pre-check (/root/reference/src/safety/image_quality.py:30-103:
cv2.cvtColor(..., COLOR_BGR2GRAY) and cv2.Laplacian(gray, CV_64F)), against which the GPU statistics kernel
(csrc/misc.hip: image_stats_kernel) is tested bit-exactly.

PARITY UNPINNED: OpenCV is not installed in this image and the reference ships no fixture for this module, so these
functions restate OpenCV's published algorithm (fixed-point RGB2GRAY with the 4899 / 9617 / 1868 >> 14 coefficients,
ksize-1 4-neighbour Laplacian, BORDER_REFLECT_101) and have themselves not been checked against cv2."""
import numpy as np


def gray_u8(rgb: np.ndarray) -> np.ndarray:
    """OpenCV's 8-bit RGB->gray rule."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.int64)


def laplacian_reflect101(gray: np.ndarray) -> np.ndarray:
    """ksize-1 Laplacian (4-neighbour) with BORDER_REFLECT_101."""
    p = np.pad(gray, 1, mode="reflect")
    return p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4 * gray
