#!/bin/bash
# Fuzz csrc/jpeg_host.c under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (no GPU involved).
#   tools/jpeg_fuzz.sh [iterations=200000]
set -e
cd "$(dirname "$0")/.."
D=$(mktemp -d)
python3 - "$D" <<'PY'
import io, sys
import numpy as np
from PIL import Image
d = sys.argv[1]
rng = np.random.default_rng(0)
k = 0
for (h, w) in [(37, 53), (64, 64), (8, 8), (100, 3), (160, 200)]:
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.clip(np.stack([128 + 100 * np.sin(xx / 9 + yy / 13), 128 + 90 * np.cos(xx / 7), 128 + 80 * np.sin(yy / 5)], -1)
                + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
    for kw in (dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, restart_marker_blocks=2),
               dict(subsampling=2, restart_marker_rows=1)):
        Image.fromarray(a).save(f"{d}/s{k}.jpg", quality=85, **kw); k += 1
    Image.fromarray(a[..., 0]).save(f"{d}/s{k}.jpg", quality=85); k += 1
PY
gcc -g -O1 -std=c99 -fsanitize=address,undefined -fno-sanitize-recover=all -o "$D/jpeg_fuzz" tools/jpeg_fuzz.c
"$D/jpeg_fuzz" "${1:-200000}" "$D"/s*.jpg
rm -rf "$D"
