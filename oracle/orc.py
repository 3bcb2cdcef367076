"""ctypes wrapper around the CPU oracle (oracle/liborc*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_U32 = 0xFFFFFFFF
FN = dict(sin=0, cos=1, exp=2, log2=3, exp2=4, pow=5, sqrt=6, div=7, tan=8)


class _Scene(C.Structure):
    _fields_ = [("primitives", C.c_void_p), ("nprim", C.c_uint32),
                ("lights", C.c_void_p), ("nlight", C.c_uint32),
                ("spectra", C.c_void_p), ("nspectra", C.c_uint32),
                ("cie", C.c_void_p), ("camera", C.c_void_p)]


class Transcript(C.Structure):
    _fields_ = [("n_hits", C.c_uint32), ("hits", C.c_uint32 * 128), ("n_rand", C.c_uint32),
                ("wavelengths", C.c_uint32 * 4), ("radiance", C.c_float * 4), ("xyz", C.c_float * 3)]


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    """Compile the oracle (gcc; seconds)."""
    if force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("liborc.so", "liborc_nofma.so")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


_lib = None


def lib():
    global _lib
    if _lib is None:
        name = "liborc.so" if _cpu_has_fma() else "liborc_nofma.so"
        path = os.environ.get("ORC_LIB") or os.path.join(_HERE, name)     # ORC_LIB: oracle/sensitivity.py's variants only
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        L.orc_trace_pixel.restype = C.c_int
        L.orc_trace_pixel.argtypes = [C.POINTER(_Scene), C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.POINTER(Transcript)]
        L.orc_intersect.restype = C.c_int
        L.orc_intersect.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_uint32,
                                    C.c_void_p, C.c_void_p]
        L.orc_ray_log.restype = C.c_int
        L.orc_ray_log.argtypes = [C.POINTER(_Scene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_tea.restype = C.c_uint32
        L.orc_tea.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_rand_kat.restype = None
        L.orc_rand_kat.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_math_eval.restype = None
        L.orc_math_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_hit_pad.restype = C.c_float
        L.orc_hit_pad.argtypes = [C.POINTER(_Scene)]
        L.orc_camera_frame.restype = None
        L.orc_camera_frame.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


class Scene:
    """Holds references to the packed host buffers (any object with the
    attributes primitives, lights, spectra, cie, camera as numpy arrays)."""

    def __init__(self, primitives, lights, spectra, cie, camera):
        self.primitives = np.ascontiguousarray(primitives).view(np.uint8).reshape(-1)
        self.lights = np.ascontiguousarray(lights).view(np.uint8).reshape(-1)
        self.spectra = np.ascontiguousarray(spectra, np.float32)
        self.cie = np.ascontiguousarray(cie, np.float32)
        self.camera = np.ascontiguousarray(camera, np.float32)
        assert self.primitives.size % 80 == 0 and self.lights.size % 80 == 0
        assert self.spectra.size % 301 == 0 and self.cie.size == 3 * 471 and self.camera.size == 16
        self.c = _Scene(self.primitives.ctypes.data, self.primitives.size // 80,
                        self.lights.ctypes.data, self.lights.size // 80,
                        self.spectra.ctypes.data, self.spectra.size // 301,
                        self.cie.ctypes.data, self.camera.ctypes.data)

    @classmethod
    def from_packed(cls, ps):
        return cls(ps.primitives, ps.lights, ps.spectra, ps.cie, ps.camera)

    @property
    def width(self):
        return int(self.camera[11])

    @property
    def height(self):
        return int(self.camera[12])

    def render(self, n_samples=1, first_sample=1, rect=None, accum=None, nthreads=0):
        """Returns (accum[H,W,4] f32, rgba8[H,W,4] u8, counters[8] u64)."""
        W, H = self.width, self.height
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
        if accum is None:
            accum = np.zeros((H, W, 4), np.float32)
        rgba = np.zeros((H, W, 4), np.uint8)
        counters = np.zeros(8, np.uint64)
        rc = lib().orc_render(C.byref(self.c), accum.ctypes.data, rgba.ctypes.data, first_sample,
                              n_samples, x0, y0, x1, y1, counters.ctypes.data, nthreads)
        if rc != 0:
            raise RuntimeError("orc_render failed")
        return accum, rgba, counters

    def trace_pixel(self, x, y, sample=1) -> Transcript:
        t = Transcript()
        if lib().orc_trace_pixel(C.byref(self.c), x, y, sample, C.byref(t)) != 0:
            raise RuntimeError("orc_trace_pixel failed")
        return t

    def ray_log(self, x, y, sample=1, cap=512) -> np.ndarray:
        """[n,12] float32: o, d, exclude bits, hit index bits, t, normal."""
        log = np.zeros((cap, 12), np.float32)
        n = lib().orc_ray_log(C.byref(self.c), x, y, sample, log.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("orc_ray_log failed")
        return log[:n]

    def intersect(self, o, d, exclude=MAX_U32):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        of = np.zeros(7, np.float32)
        ou = np.zeros(5, np.uint32)
        lib().orc_intersect(C.byref(self.c), o.ctypes.data, d.ctypes.data, exclude, of.ctypes.data,
                            ou.ctypes.data)
        return of, ou

    def hit_pad(self) -> float:
        return float(lib().orc_hit_pad(C.byref(self.c)))

    def camera_frame(self) -> np.ndarray:
        out = np.zeros(12, np.float32)
        lib().orc_camera_frame(self.camera.ctypes.data, out.ctypes.data)
        return out


def max_threads() -> int:
    return int(lib().orc_max_threads())


def tea(v0, v1):
    return int(lib().orc_tea(v0, v1))


def rand_kat(x, y, sample, n):
    out = np.zeros(n, np.uint32)
    seed = np.zeros(4, np.uint32)
    lib().orc_rand_kat(x, y, sample, n, out.ctypes.data, seed.ctypes.data)
    return out, seed


def math_eval(fn: str, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
    out = np.empty_like(a)
    lib().orc_math_eval(FN[fn], a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
    return out
