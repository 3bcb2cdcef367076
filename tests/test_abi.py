"""CPU-only: the C-ABI library loads, exports every symbol include/crt.h
declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "crt.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("crt_create", "crt_upload_scene", "crt_build_accel", "crt_reset", "crt_trace",
                 "crt_read_accum", "crt_read_rgba8", "crt_counters", "crt_last_error", "crt_destroy"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from computeraytracer_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libcrt.so first (__graft_entry__.build())"
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/crt.h but not exported"
    # and the binding covers the whole header
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.load().crt_abi_version() == 2


def test_library_contains_gfx950_code_object():
    from computeraytracer_amd import _lib
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", _lib.LIB_PATH], capture_output=True, text=True)
    assert ".hip_fatbin" in out.stdout
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from computeraytracer_amd import Renderer
    from computeraytracer_amd._lib import CrtError
    with pytest.raises(CrtError) as e:
        Renderer(0)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under the product tree may
    include, import or link it."""
    bad = []
    for base in ("computeraytracer_amd", "include", "host", "addon"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c", ".js", ".mk")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"(#include\s*[\"<].*oracle|from\s+oracle|import\s+oracle|liborc|crt_oracle)", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_row_layout_matches_the_python_partition():
    """crt_layout_rows (no GPU involved) is the one definition of the multi-GPU row partition: it agrees with
    partition.py (what the torch-side StripFrame and the gloo test use) for strips and for interleaved bands, and
    the parts cover the frame exactly once."""
    import numpy as np
    from computeraytracer_amd import _lib
    from computeraytracer_amd.partition import band_rows, strip_rows
    lib = _lib.load()
    for H, world, band in [(1080, 8, 8), (2160, 8, 8), (363, 3, 5), (7, 4, 8), (64, 2, 8), (61, 2, 0), (1080, 8, 0), (5, 8, 0), (1, 1, 0)]:
        seen = []
        for part in range(world):
            n = C.c_uint32()
            assert lib.crt_layout_rows(H, band, world, part, C.byref(n), None) == 0
            rows = np.zeros(max(n.value, 1), np.uint32)
            assert lib.crt_layout_rows(H, band, world, part, C.byref(n), rows.ctypes.data) == 0
            rows = rows[: n.value]
            if band:
                assert np.array_equal(rows, band_rows(H, world, part, band))
            else:
                y0, y1 = strip_rows(H, world, part)
                assert np.array_equal(rows, np.arange(y0, y1))
            seen.append(rows)
        assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(H))
    n = C.c_uint32()
    assert lib.crt_layout_rows(10, 8, 2, 2, C.byref(n), None) != 0          # part out of range


def test_local_communicator_ids_are_distinct_and_need_no_gpu():
    from computeraytracer_amd import _lib
    lib = _lib.load()
    a, b = C.create_string_buffer(128), C.create_string_buffer(128)
    assert lib.crt_comm_unique_id(a, 1) == 0 and lib.crt_comm_unique_id(b, 1) == 0
    assert a.raw[:8] == b"CRTLOCAL" and a.raw != b.raw
