"""Image-tile partition of the framebuffer across GPUs.

Pixels are independent (a pixel depends only on x, y, sample, W, H and the
scene: ComputeShader.wgsl:85-86,98,107), so the frame splits into per-rank
horizontal strips with the scene replicated and no data-path collective; the
only exchange is the gather of finished strips.
"""
from __future__ import annotations


def strip_rows(height: int, world: int, rank: int) -> tuple[int, int]:
    """Rows [y0, y1) of rank `rank`: ceil(H/world) rows each, the last ranks may be short/empty."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    per = -(-height // world)
    y0 = min(rank * per, height)
    return y0, min(y0 + per, height)


def band_rows(height: int, world: int, rank: int, band: int = 8):
    """Global row indices of rank `rank` under the row-interleaved partition (bands of `band`
    rows dealt round-robin): rows y with (y // band) % world == rank, in local order."""
    import numpy as np
    if world < 1 or not (0 <= rank < world) or band < 1:
        raise ValueError("bad world/rank/band")
    y = np.arange(height)
    return y[(y // band) % world == rank]
