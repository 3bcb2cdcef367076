"""Multi-GPU plumbing: one process per GPU, frame split into horizontal strips,
strips gathered with one all_gather per buffer (RCCL over xGMI on the GPU box,
gloo in the CPU tests).  torch is used only for device memory and the
collective; rendering goes through the C ABI.

No data-path collective exists in the reference (single GPUDevice); pixels are
independent, so the gather of finished strips is the only exchange.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .partition import strip_rows


class StripFrame:
    """Owns this rank's strip buffers (padded to the common strip height so the
    collective is a plain all_gather_into_tensor) and assembles the full frame."""

    def __init__(self, width: int, height: int, world: int, rank: int, device):
        self.W, self.H, self.world, self.rank = width, height, world, rank
        self.y0, self.y1 = strip_rows(height, world, rank)
        self.rows_max = strip_rows(height, world, 0)[1]
        self.accum = torch.zeros((self.rows_max, width, 4), dtype=torch.float32, device=device)
        self.rgba = torch.zeros((self.rows_max, width, 4), dtype=torch.uint8, device=device)
        if world > 1:
            # concatenation form [world * rows_max, W, 4]: accepted by both RCCL and gloo
            self.full_accum = torch.empty((world * self.rows_max, width, 4), dtype=torch.float32, device=device)
            self.full_rgba = torch.empty((world * self.rows_max, width, 4), dtype=torch.uint8, device=device)

    @property
    def tile(self):
        """(x0, y0, x1, y1) for crt_set_tile."""
        return 0, self.y0, self.W, self.y1

    def gather(self):
        if self.world > 1:
            dist.all_gather_into_tensor(self.full_accum, self.accum)
            dist.all_gather_into_tensor(self.full_rgba, self.rgba)

    def image(self):
        """(accum[H,W,4], rgba[H,W,4]) assembled from the last gather."""
        if self.world == 1:
            return self.accum[: self.H], self.rgba[: self.H]
        pa, pr = [], []
        for r in range(self.world):
            y0, y1 = strip_rows(self.H, self.world, r)
            base = r * self.rows_max
            pa.append(self.full_accum[base: base + (y1 - y0)])
            pr.append(self.full_rgba[base: base + (y1 - y0)])
        return torch.cat(pa, 0), torch.cat(pr, 0)
