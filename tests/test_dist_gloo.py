"""CPU, world_size 2, gloo: the N>1 host path (strip partition -> render the
strip -> all_gather -> assemble) reproduces the single-process frame exactly.
The oracle stands in for the GPU renderer here (tests may use it)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_strip_rows_cover_the_frame_exactly():
    from computeraytracer_amd.partition import strip_rows
    for H in (1, 7, 8, 135, 1080, 2160, 1001):
        for world in (1, 2, 3, 4, 8):
            rows = [strip_rows(H, world, r) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            assert max(y1 - y0 for y0, y1 in rows) == rows[0][1] - rows[0][0]
    with pytest.raises(ValueError):
        strip_rows(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, out_dir, band=0):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from computeraytracer_amd import cornell
    from computeraytracer_amd.distributed import StripFrame
    from oracle import orc
    ps = cornell(W, H)
    sc = orc.Scene.from_packed(ps)
    sf = StripFrame(W, H, world, rank, "cpu", band=band)
    if band:
        # this rank's rows from the C ABI's own definition of the partition (crt_layout_rows: what crt_comm_partition,
        # crt_set_row_bands and the assembly kernel use; no GPU involved) -- must be the rows StripFrame gathers into
        import ctypes as C
        from computeraytracer_amd import _lib
        n = C.c_uint32()
        buf = np.zeros(H, np.uint32)
        assert _lib.load().crt_layout_rows(H, band, world, rank, C.byref(n), buf.ctypes.data) == 0
        rows = buf[: n.value].astype(np.int64)
        assert np.array_equal(rows, np.asarray(sf.rows[rank]))
        acc = np.zeros((H, W, 4), np.float32)
        rgba = np.zeros((H, W, 4), np.uint8)
        for y in range(0, H, band):
            if (y // band) % world == rank:
                a, r, _ = sc.render(spp, rect=(0, y, W, min(y + band, H)), nthreads=2)
                acc[y:y + band], rgba[y:y + band] = a[y:y + band], r[y:y + band]
        sf.accum[: len(rows)] = torch.from_numpy(acc[rows])
        sf.rgba[: len(rows)] = torch.from_numpy(rgba[rows])
    else:
        x0, y0, x1, y1 = sf.tile
        acc, rgba, _ = sc.render(spp, rect=(x0, y0, x1, y1), nthreads=2)
        sf.accum[: y1 - y0] = torch.from_numpy(acc[y0:y1])
        sf.rgba[: y1 - y0] = torch.from_numpy(rgba[y0:y1])
    sf.gather()
    a, r = sf.image()
    np.save(os.path.join(out_dir, f"acc{rank}.npy"), a.numpy())
    np.save(os.path.join(out_dir, f"rgba{rank}.npy"), r.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_band_rows_partition_covers_the_frame():
    from computeraytracer_amd.partition import band_rows
    for H, world, band in [(1080, 8, 8), (363, 3, 5), (7, 4, 8), (64, 2, 8)]:
        parts = [band_rows(H, world, r, band) for r in range(world)]
        allrows = np.sort(np.concatenate(parts))
        assert np.array_equal(allrows, np.arange(H))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= band


@pytest.mark.parametrize("H,band", [(64, 0), (61, 0), (61, 8)])   # even split, short last strip, interleaved bands
def test_two_rank_gather_equals_single_process(tmp_path, orc, H, band):
    W, spp, world = 80, 2, 2
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, str(tmp_path), band), nprocs=world, join=True)
    from computeraytracer_amd import cornell
    acc_o, rgba_o, _ = orc.Scene.from_packed(cornell(W, H)).render(spp)
    for rank in range(world):                     # all_gather: every rank holds the frame
        a = np.load(tmp_path / f"acc{rank}.npy")
        r = np.load(tmp_path / f"rgba{rank}.npy")
        assert a.shape == (H, W, 4) and np.array_equal(a.view(np.uint32), acc_o.view(np.uint32))
        assert np.array_equal(r, rgba_o)
