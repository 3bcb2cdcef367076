"""Host driver over the C ABI: the Python counterpart of the reference's
``Main()`` / ``frame()`` (src/main.js:7-624) minus the browser.

    Main():  requestDevice                -> Renderer(device)
             createBuffer(b4..b8) + unmap -> Renderer.upload(PackedScene)
             accumulator + sample = 0     -> (done by upload / reset)
    frame(): dispatch(1); dispatch(W/8,H/8) -> Renderer.frame(n)   (n frames fused)
    (the reference never reads back; read_accum/read_rgba8 replace its blit pass)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ACCEL_BVH2, ACCEL_LBVH, ACCEL_NONE, CNT, CrtError, NCOUNTERS
from .scene import PackedScene

_ACCEL = {"none": ACCEL_NONE, "brute": ACCEL_NONE, "bvh2": ACCEL_BVH2, "bvh": ACCEL_BVH2, "lbvh": ACCEL_LBVH, ACCEL_LBVH: ACCEL_LBVH,
          ACCEL_NONE: ACCEL_NONE, ACCEL_BVH2: ACCEL_BVH2}


class Renderer:
    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.crt_create(C.byref(h), int(device))
        if rc != 0:
            raise CrtError(rc, (self._lib.crt_last_error(None) or b"").decode())
        self._h = h
        self.device = int(device)
        self.scene: PackedScene | None = None

    # -- plumbing
    def _chk(self, rc: int):
        if rc != 0:
            raise CrtError(rc, (self._lib.crt_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.crt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- Main()
    def upload(self, ps: PackedScene):
        prim = np.ascontiguousarray(ps.primitives)
        lights = np.ascontiguousarray(ps.lights)
        spectra = np.ascontiguousarray(ps.spectra, np.float32)
        cie = np.ascontiguousarray(ps.cie, np.float32)
        cam = np.ascontiguousarray(ps.camera, np.float32)
        if prim.nbytes % 80 or lights.nbytes % 80:
            raise ValueError("primitive / light buffers must be multiples of 80 bytes")
        if spectra.size % 301 or cie.size != 3 * 471 or cam.size != 16:
            raise ValueError("spectra must be n x 301, cie 3 x 471, camera 16 floats")
        self._chk(self._lib.crt_upload_scene(self._h, prim.ctypes.data, prim.nbytes // 80,
                                             lights.ctypes.data, lights.nbytes // 80,
                                             spectra.ctypes.data, spectra.size // 301,
                                             cie.ctypes.data, cam.ctypes.data))
        self.scene = ps
        return self

    def set_tile(self, x0: int, y0: int, x1: int, y1: int):
        self._chk(self._lib.crt_set_tile(self._h, x0, y0, x1, y1))
        return self

    def set_row_bands(self, band_rows: int, parts: int, part: int):
        self._chk(self._lib.crt_set_row_bands(self._h, band_rows, parts, part))
        return self

    def build_accel(self, mode="bvh2"):
        self._chk(self._lib.crt_build_accel(self._h, _ACCEL[mode]))
        return self

    def reset(self):
        self._chk(self._lib.crt_reset(self._h))
        return self

    def set_option(self, name: str, value: int):
        self._chk(self._lib.crt_set_option(self._h, name.encode(), int(value)))
        return self

    # -- frame()
    def frame(self, n_samples: int = 1):
        """n x { sample++ ; trace }  (asynchronous)."""
        self._chk(self._lib.crt_trace(self._h, int(n_samples)))
        return self

    def sync(self):
        self._chk(self._lib.crt_sync(self._h))
        return self

    @property
    def sample(self) -> int:
        v = C.c_uint32()
        self._chk(self._lib.crt_sample_count(self._h, C.byref(v)))
        return v.value

    @property
    def tile(self):
        out = (C.c_uint32 * 4)()
        self._chk(self._lib.crt_tile(self._h, out))
        return tuple(out)

    # -- readback
    def read_accum(self) -> np.ndarray:
        _, _, tw, th = self.tile
        out = np.empty((th, tw, 4), np.float32)
        self._chk(self._lib.crt_read_accum(self._h, out.ctypes.data))
        return out

    def read_rgba8(self) -> np.ndarray:
        _, _, tw, th = self.tile
        out = np.empty((th, tw, 4), np.uint8)
        self._chk(self._lib.crt_read_rgba8(self._h, out.ctypes.data))
        return out

    def read_latest_rgba8(self):
        """(frame, sample index): the newest complete frame in stream order, without finishing what is in flight."""
        _, _, tw, th = self.tile
        out = np.empty((th, tw, 4), np.uint8)
        s = C.c_uint32()
        self._chk(self._lib.crt_read_latest_rgba8(self._h, out.ctypes.data, C.byref(s)))
        return out, s.value

    @property
    def latest_sample(self) -> int:
        v = C.c_uint32()
        self._chk(self._lib.crt_latest_sample(self._h, C.byref(v)))
        return v.value

    def read_sample_rgba8(self, sample: int) -> np.ndarray:
        """The frame after exactly `sample` samples (option frame_ring = F keeps the last F)."""
        _, _, tw, th = self.tile
        out = np.empty((th, tw, 4), np.uint8)
        self._chk(self._lib.crt_read_sample_rgba8(self._h, int(sample), out.ctypes.data))
        return out

    def write_accum(self, accum: np.ndarray, sample: int):
        a = np.ascontiguousarray(accum, np.float32)
        _, _, tw, th = self.tile
        if a.size != tw * th * 4:
            raise ValueError("accum must be th x tw x 4 float32")
        self._chk(self._lib.crt_write_accum(self._h, a.ctypes.data, int(sample)))
        return self

    def device_buffers(self):
        a, r = C.c_void_p(), C.c_void_p()
        self._chk(self._lib.crt_device_buffers(self._h, C.byref(a), C.byref(r)))
        return a.value, r.value

    def bind_output(self, accum_dev_ptr: int | None, rgba_dev_ptr: int | None):
        self._chk(self._lib.crt_bind_output(self._h, C.c_void_p(accum_dev_ptr or 0), C.c_void_p(rgba_dev_ptr or 0)))
        return self

    def set_stream(self, hip_stream: int | None):
        self._chk(self._lib.crt_set_stream(self._h, C.c_void_p(hip_stream or 0)))
        return self

    # -- multi-GPU: the frame across the ranks of a communicator (include/crt.h, "Multi-GPU")
    @staticmethod
    def comm_unique_id(local: bool = False) -> bytes:
        """128-byte communicator id: RCCL (one rank makes it, every rank gets it) or, local=True, the in-process transport."""
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        rc = lib.crt_comm_unique_id(buf, 1 if local else 0)
        if rc != 0:
            raise CrtError(rc, (lib.crt_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, comm_id: bytes, rank: int, world: int):
        if len(comm_id) != 128:
            raise ValueError("communicator id must be 128 bytes")
        self._chk(self._lib.crt_comm_init(self._h, C.create_string_buffer(comm_id, 128), int(rank), int(world)))
        return self

    def comm_partition(self, band_rows: int = 8):
        self._chk(self._lib.crt_comm_partition(self._h, int(band_rows)))
        return self

    def gather(self, rgba8: bool = True, accum: bool = False):
        self._chk(self._lib.crt_gather(self._h, (1 if rgba8 else 0) | (2 if accum else 0)))
        return self

    @property
    def image_size(self):
        out = (C.c_uint32 * 2)()
        self._chk(self._lib.crt_image_size(self._h, out))
        return int(out[0]), int(out[1])

    def read_frame_rgba8(self) -> np.ndarray:
        W, H = self.image_size
        out = np.empty((H, W, 4), np.uint8)
        self._chk(self._lib.crt_read_frame_rgba8(self._h, out.ctypes.data))
        return out

    def read_frame_accum(self) -> np.ndarray:
        W, H = self.image_size
        out = np.empty((H, W, 4), np.float32)
        self._chk(self._lib.crt_read_frame_accum(self._h, out.ctypes.data))
        return out

    def frame_device_buffers(self):
        a, r = C.c_void_p(), C.c_void_p()
        self._chk(self._lib.crt_frame_device_buffers(self._h, C.byref(a), C.byref(r)))
        return a.value, r.value

    def comm_info(self) -> dict:
        out = (C.c_int * 4)()
        self._chk(self._lib.crt_comm_info(self._h, out))
        return dict(rank=out[0], world=out[1], transport={0: "rccl", 1: "local"}.get(out[2]), rows=out[3])

    def comm_destroy(self):
        self._chk(self._lib.crt_comm_destroy(self._h))
        return self

    # -- measurement
    def enable_counters(self, on: bool = True):
        self._chk(self._lib.crt_enable_counters(self._h, 1 if on else 0))
        return self

    def reset_counters(self):
        self._chk(self._lib.crt_reset_counters(self._h))
        return self

    def counters(self) -> dict:
        out = np.zeros(NCOUNTERS, np.uint64)
        self._chk(self._lib.crt_counters(self._h, out.ctypes.data))
        return {k: int(out[i]) for k, i in CNT.items()}

    def last_trace_ms(self):
        ms, n = C.c_float(), C.c_uint32()
        self._chk(self._lib.crt_last_trace_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_kernel_ms(self):
        ms, n = C.c_float(), C.c_uint32()
        self._chk(self._lib.crt_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def accel_stats(self) -> dict:
        out = np.zeros(8, np.uint64)
        self._chk(self._lib.crt_accel_stats(self._h, out.ctypes.data))
        return dict(nodes=int(out[0]), leaves=int(out[1]), max_depth=int(out[2]), bytes=int(out[3]),
                    bytes_per_box=int(out[4]), width=int(out[5]), wide_nodes=int(out[6]),
                    builder=("sah-host", "lbvh-gpu")[int(out[7])])

    # -- test hooks
    def debug_intersect(self, origins, directions, exclude=None) -> np.ndarray:
        o = np.asarray(origins, np.float32).reshape(-1, 3)
        d = np.asarray(directions, np.float32).reshape(-1, 3)
        n = o.shape[0]
        rays = np.zeros((n, 8), np.float32)
        rays[:, 0:3] = o
        rays[:, 3:6] = d
        ex = np.full(n, 0xFFFFFFFF, np.uint32) if exclude is None else np.asarray(exclude, np.uint32)
        rays[:, 6] = ex.view(np.float32)
        out = np.zeros((n, 8), np.float32)
        self._chk(self._lib.crt_debug_intersect(self._h, rays.ctypes.data, n, out.ctypes.data))
        return out

    def debug_probes(self) -> list:
        out = np.zeros(8, np.uint64)
        self._chk(self._lib.crt_debug_probes(self._h, out.ctypes.data))
        return [int(v) for v in out]

    def debug_math(self, fn: int, a, b=None) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
        out = np.empty_like(a)
        self._chk(self._lib.crt_debug_math(self._h, int(fn), a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size))
        return out
