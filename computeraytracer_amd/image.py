"""Readout step after the path (replaces the reference's blit pass,
TextureRenderShader.wgsl + src/main.js:612-617): rgba8 -> PPM / PNG files, and the
f32 XYZ accumulator + sample index as a resumable checkpoint (.npz)."""
from __future__ import annotations

import struct
import zlib

import numpy as np


def write_ppm(path: str, rgba: np.ndarray) -> None:
    """Binary P6, row 0 = top (same orientation as the reference's canvas)."""
    h, w = rgba.shape[:2]
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgba[..., :3]).tobytes())


def write_png(path: str, rgba: np.ndarray) -> None:
    """8-bit RGBA PNG with the standard library only."""
    h, w = rgba.shape[:2]
    rows = np.empty((h, 1 + w * 4), np.uint8)
    rows[:, 0] = 0
    rows[:, 1:] = np.ascontiguousarray(rgba, np.uint8).reshape(h, w * 4)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)) + chunk(b"IEND", b""))


def save_checkpoint(path: str, renderer) -> None:
    """alt_color_buffer + sample: everything needed to resume."""
    np.savez_compressed(path, accum=renderer.read_accum(), sample=np.uint32(renderer.sample), tile=np.asarray(renderer.tile))


def load_checkpoint(path: str, renderer) -> int:
    d = np.load(path)
    if tuple(int(v) for v in d["tile"]) != tuple(renderer.tile):
        raise ValueError("checkpoint tile does not match the renderer's tile")
    renderer.write_accum(d["accum"], int(d["sample"]))
    return int(d["sample"])
