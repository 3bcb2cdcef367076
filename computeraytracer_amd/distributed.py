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

from .partition import band_rows, strip_rows


class StripFrame:
    """Owns this rank's strip buffers (padded to the common strip height so the
    collective is a plain all_gather_into_tensor) and assembles the full frame."""

    def __init__(self, width: int, height: int, world: int, rank: int, device, band: int = 0):
        """band = 0: contiguous strips; band > 0: bands of `band` rows dealt round-robin to the
        ranks (balances regions of different path length; use crt_set_row_bands on the renderer)."""
        self.W, self.H, self.world, self.rank, self.band = width, height, world, rank, band
        if band:
            self.rows = [band_rows(height, world, r, band) for r in range(world)]
            self.y0, self.y1 = 0, int(len(self.rows[rank]))      # local row count in y1
            self.rows_max = max(len(r) for r in self.rows)
        else:
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
        """(x0, y0, x1, y1) for crt_set_tile (contiguous strips only)."""
        return 0, self.y0, self.W, self.y1

    @property
    def local_rows(self) -> int:
        return self.y1 - self.y0

    def apply(self, renderer):
        """Point a Renderer at this rank's part of the frame."""
        if self.band:
            renderer.set_row_bands(self.band, self.world, self.rank)
        else:
            renderer.set_tile(*self.tile)

    def gather(self, accum: bool = True):
        """all_gather of the strips: the rgba8 framebuffer always (what the reference's frame()
        produces for display), the f32 XYZ accumulator on request (checkpoint / final readout)."""
        if self.world > 1:
            if self.accum.is_cuda and dist.get_backend() != "nccl":
                raise RuntimeError("StripFrame.gather: device strips need the nccl (RCCL) backend; backend is "
                                   f"{dist.get_backend()!r} (rehearsals on a one-GPU box stage through the host themselves)")
            if accum:
                dist.all_gather_into_tensor(self.full_accum, self.accum)
            dist.all_gather_into_tensor(self.full_rgba, self.rgba)

    def image(self):
        """(accum[H,W,4], rgba[H,W,4]) assembled from the last gather."""
        if self.world == 1:
            return self.accum[: self.H], self.rgba[: self.H]
        if self.band:
            acc = torch.empty((self.H, self.W, 4), dtype=torch.float32, device=self.accum.device)
            rgb = torch.empty((self.H, self.W, 4), dtype=torch.uint8, device=self.accum.device)
            for r in range(self.world):
                idx = torch.as_tensor(self.rows[r], device=self.accum.device)
                base = r * self.rows_max
                acc[idx] = self.full_accum[base: base + len(idx)]
                rgb[idx] = self.full_rgba[base: base + len(idx)]
            return acc, rgb
        pa, pr = [], []
        for r in range(self.world):
            y0, y1 = strip_rows(self.H, self.world, r)
            base = r * self.rows_max
            pa.append(self.full_accum[base: base + (y1 - y0)])
            pr.append(self.full_rgba[base: base + (y1 - y0)])
        return torch.cat(pa, 0), torch.cat(pr, 0)
