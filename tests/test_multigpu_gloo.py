"""CPU coverage of the N>1 path (world_size 2 and 4, gloo): the band partition + the single gather that
bench.py runs over RCCL.  Each rank holds the image a partitioned render leaves behind (only its own
interleaved tile rows written); after the gather rank 0 must hold the full single-GPU image."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, full_np, out_path):
    import torch
    import torch.distributed as dist
    from f_renderer_amd.multigpu import BandGather, band_layout, owned_tile_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, HP = band_layout(H, world)
    full = torch.from_numpy(full_np)
    local = torch.zeros((HP, W), dtype=torch.float32)          # what frr_clear leaves: background
    for ty in owned_tile_rows(H, rank, world):                 # what a partitioned frr_draw writes
        y0, y1 = ty * 32, min(ty * 32 + 32, H)
        local[y0:y1] = full[y0:y1]
    g = BandGather(H, W, torch.float32, "cpu", rank, world)
    for _ in range(2):                                         # twice: the staging buffers are reused per frame
        final = g(local)
    # bench.py's pipelined form: two target sets, the gather of frame i in flight while frame i+1 is produced
    g2 = [BandGather(H, W, torch.float32, "cpu", rank, world) for _ in range(2)]
    locals2 = [local.clone(), local.clone() * 2.0]             # set 1 holds a different "frame"
    inflight, finals = [None, None], [None, None]
    for i in range(4):
        s = i % 2
        if inflight[s] is not None:
            finals[s] = g2[s].finish(inflight[s])
        inflight[s] = g2[s].start(locals2[s])
    for s in range(2):
        finals[s] = g2[s].finish(inflight[s])
    # blocked layout (frr_set_partition_layout(1)): contiguous slabs, gathered straight from the image
    from f_renderer_amd.multigpu import BlockGather, block_rows
    y0, y1 = block_rows(H, rank, world)
    local_b = torch.zeros((HP, W), dtype=torch.float32)
    y1c = min(y1, H)
    if y0 < H:
        local_b[y0:y1c] = full[y0:y1c]
    bg = BlockGather(H, W, torch.float32, "cpu", rank, world)
    final_b = bg.finish(bg.start(local_b))
    final_b2 = bg.finish_host(bg.start(local_b))              # bench.py's host-polled completion
    assert (final_b2 is None) == (rank != 0)
    # bench.py's frame exchange: several planes per frame (RGBA8 as a trailing (4,) u8 image, depth, ids), one gather each
    from f_renderer_amd.multigpu import FrameGather
    rgba_full = (full.unsqueeze(-1) * torch.tensor([17.0, 31.0, 59.0, 97.0])).to(torch.uint8)      # [H, W, 4]
    ids_full = (full * 1000.0).to(torch.int32)
    local_c = torch.zeros((HP, W, 4), dtype=torch.uint8)
    local_i = torch.full((HP, W), -1, dtype=torch.int32)
    if y0 < H:
        local_c[y0:y1c] = rgba_full[y0:y1c]
        local_i[y0:y1c] = ids_full[y0:y1c]
    fg = FrameGather(H, W, [(torch.uint8, (4,)), (torch.float32, ()), (torch.int32, ())], "cpu", rank, world)
    fin = fg.finish(fg.start([local_c, local_b, local_i]))
    fin = fg.finish_host(fg.start([local_c, local_b, local_i]))
    if rank == 0:
        assert torch.equal(finals[0][:H], final[:H]) and torch.equal(finals[1][:H], final[:H] * 2.0)
        assert torch.equal(final_b[:H], final[:H])
        assert torch.equal(fin[0][:H], rgba_full) and torch.equal(fin[1][:H], full) and torch.equal(fin[2][:H], ids_full)
        np.save(out_path, final[:H].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 1080, 64), (2, 100, 48), (2, 32, 16), (4, 1080, 32), (4, 70, 16)])
def test_band_gather_gloo(tmp_path, world, H, W):
    """world 2 and 4; (4, 70, 16): three tile rows over four ranks, so one rank owns nothing."""
    import torch.multiprocessing as mp
    rng = np.random.default_rng(H)
    full = rng.random((H, W), dtype=np.float32) + 0.25
    out = str(tmp_path / "final.npy")
    mp.spawn(_worker, args=(world, _free_port(), H, W, full, out), nprocs=world, join=True)
    np.testing.assert_array_equal(np.load(out), full)


def test_blocked_rows_agree_between_library_and_python():
    """frr_partition_rows (what a native host posts its receives with) == multigpu.block_rows (what bench.py uses), every
    rank owns something while there are rows to go round, and the blocks tile the window exactly once."""
    import ctypes as C
    import f_renderer_amd as fr
    from f_renderer_amd.multigpu import block_rows, block_tile_rows
    L = fr.lib()
    for H in (1080, 2160, 4096, 333, 33, 1):
        tiles_y = (H + 31) // 32
        for world in (1, 2, 3, 4, 7, 8):
            seen = 0
            for rank in range(world):
                a, b = C.c_int32(), C.c_int32()
                n = L.frr_partition_rows(0, H, rank, world, 1, 0, C.byref(a), C.byref(b))
                y0, y1 = block_rows(H, rank, world)
                t0, t1 = block_tile_rows(tiles_y, rank, world)
                assert n == (1 if t1 > t0 else 0)
                assert (t1 > t0) or rank >= tiles_y            # nobody is left without rows while there are enough
                if n:
                    assert (a.value, b.value) == (y0, min(y1, H)) and a.value == seen
                    seen = b.value
            assert seen == H
    a, b = C.c_int32(), C.c_int32()
    assert [L.frr_partition_rows(0, 1080, r, 8, 1, 0, C.byref(a), C.byref(b)) and (b.value - a.value + 31) // 32 for r in range(8)] == [5, 5, 4, 4, 4, 4, 4, 4]
    assert L.frr_partition_rows(0, 100, 1, 2, 0, 0, C.byref(a), C.byref(b)) == 2 and (a.value, b.value) == (32, 64)   # interleaved: bands
    assert L.frr_partition_rows(5, 1, 0, 1, 0, 0, None, None) < 0


def test_band_layout_matches_partition_rule():
    from f_renderer_amd.multigpu import band_layout, owned_tile_rows
    for H in (1080, 2160, 4096, 33, 1):
        for world in (1, 2, 3, 4, 8):
            tiles_y, rpr, HP = band_layout(H, world)
            assert tiles_y == (H + 31) // 32 and HP >= H and HP == rpr * world * 32
            rows = sorted(sum((owned_tile_rows(H, r, world) for r in range(world)), []))
            assert rows == list(range(tiles_y))                # every tile row has exactly one owner
            for r in range(world):
                assert all(ty % world == r for ty in owned_tile_rows(H, r, world))   # frr_set_partition's rule


def test_exchange_plan_of_every_rank_without_gpus():
    """frr_exchange_plan (the operations examples/gather_rccl.cpp posts per frame) for every rank of worlds 2, 4 and 8 --
    also 3 and 40, more ranks than tile rows -- and both layouts: every sender's operations are the receives the root posts
    for it, the root's receives + own copies tile the plane exactly once, and the blocked plan is multigpu.block_rows'."""
    import ctypes as C
    import f_renderer_amd as fr
    from f_renderer_amd import _native as N
    from f_renderer_amd.multigpu import block_rows
    L = fr.lib()
    SEND, RECV, COPY = 0, 1, 2

    def plan(H, W, rank, world, blocked, root=0):
        ops = (N.Xfer * 256)()
        n = L.frr_exchange_plan(0, H, W, rank, world, blocked, root, ops, 256)
        assert 0 <= n <= 256
        return [(ops[i].kind, ops[i].peer, ops[i].offset, ops[i].count) for i in range(n)]

    for H, W in ((1080, 1920), (4096, 4096), (2160, 3840), (500, 640), (33, 7)):
        for world in (1, 2, 3, 4, 8, 40):
            for blocked in (1, 0):
                for root in ({0, world - 1} if world > 1 else {0}):
                    rootp = plan(H, W, root, world, blocked, root)
                    covered = np.zeros(H * W, np.uint8)
                    for kind, peer, off, cnt in rootp:
                        assert kind in (RECV, COPY) and (peer == root) == (kind == COPY)
                        covered[off:off + cnt] += 1
                    assert covered.min() == 1 and covered.max() == 1, (H, W, world, blocked)
                    for r in range(world):
                        if r == root:
                            continue
                        mine = plan(H, W, r, world, blocked, root)
                        assert all(k == SEND and p == root for k, p, _, _ in mine)
                        assert [(o, c) for _, _, o, c in mine] == [(o, c) for k, p, o, c in rootp if k == RECV and p == r]
                        if blocked:
                            a, b = block_rows(H, r, world)
                            b = min(b, H)
                            assert [(o, c) for _, _, o, c in mine] == ([(a * W, (b - a) * W)] if b > a else [])
    assert L.frr_exchange_plan(0, 100, 10, 2, 2, 1, 0, None, 0) < 0      # rank out of range
    assert L.frr_exchange_plan(0, 1080, 1920, 3, 8, 1, 0, None, 0) == 1  # counting only
