"""Screen-tile partition across ranks and the final-image gather (one process per GPU).

Two ownership layouts (`frr_set_partition_layout`): interleaved tile rows (BandGather below: the owned
bands are staged into one contiguous buffer and interleaved back on the destination) and blocked tile
rows (BlockGather: a rank's part of the row-major image is one contiguous slab, so the gather reads the
render target and writes the final image directly -- no staging copies; bench.py uses this one).

The path shards by screen tiles: rank r owns the 32-pixel tile rows ty with ty % world == r
(`frr_set_partition`, include/frr.h); geometry is replicated; there is no data-path collective
during a frame.  The only exchange is ONE gather of the owned bands to rank 0 for the final image
(RCCL over xGMI through torch.distributed's "nccl" backend on GPUs; "gloo" in the CPU tests).

Frame targets are allocated with the height padded to rows_per_rank * world tile rows, so that a
rank's interleaved bands are a single strided view [rows_per_rank, 32, W] of its own image.
"""
import torch

TILE = 32


def band_layout(height, world):
    """(tiles_y, rows_per_rank, padded_height) for a frame of `height` pixel rows."""
    tiles_y = (height + TILE - 1) // TILE
    rows_per_rank = (tiles_y + world - 1) // world
    return tiles_y, rows_per_rank, rows_per_rank * world * TILE


def owned_tile_rows(height, rank, world):
    tiles_y, _, _ = band_layout(height, world)
    return list(range(rank, tiles_y, world))


def owned_view(image, rank, world):
    """Strided view [rows_per_rank, 32, W, ...] of the bands rank owns in a padded image [HP, W, ...]."""
    hp = image.shape[0]
    rows_per_rank = hp // (world * TILE)
    assert rows_per_rank * world * TILE == hp, "image height must be padded with band_layout()"
    return image.view(rows_per_rank, world, TILE, *image.shape[1:])[:, rank]


class BandGather:
    """One gather per frame of every rank's owned bands into the full (padded) image on `dst`."""

    def __init__(self, height, width, dtype, device, rank, world, dst=0, trailing=()):
        self.rank, self.world, self.dst = rank, world, dst
        _, self.rows_per_rank, self.padded_height = band_layout(height, world)
        shape = (self.rows_per_rank, TILE, width, *trailing)
        self.band = torch.zeros(shape, dtype=dtype, device=device)
        if rank == dst:
            self.gbuf = torch.empty((world, *shape), dtype=dtype, device=device)
            self.gathered = list(self.gbuf.unbind(0))
            self.final = torch.zeros((self.padded_height, width, *trailing), dtype=dtype, device=device)
        else:
            self.gbuf, self.gathered, self.final = None, None, None

    def start(self, local_image, group=None):
        """Asynchronous form: stages the owned bands and launches the gather without waiting for it, so that the
        next frame (rendered into ANOTHER target set) overlaps the exchange.  Returns a handle for finish()."""
        import torch.distributed as dist
        self.band.copy_(owned_view(local_image, self.rank, self.world))
        return dist.gather(self.band, self.gathered, dst=self.dst, group=group, async_op=True)

    def finish(self, handle):
        """Waits (on the current stream) for a gather started with start() and interleaves the bands on dst."""
        handle.wait()
        if self.rank == self.dst:
            v = self.final.view(self.rows_per_rank, self.world, TILE, *self.final.shape[1:])
            v.copy_(self.gbuf.transpose(0, 1))
            return self.final
        return None

    def __call__(self, local_image, group=None):
        """local_image: this rank's padded image [HP, W, ...] (only its owned bands are meaningful)."""
        import torch.distributed as dist
        self.band.copy_(owned_view(local_image, self.rank, self.world))   # owned bands -> contiguous staging
        dist.gather(self.band, self.gathered, dst=self.dst, group=group)   # the frame's only collective
        if self.rank == self.dst:
            v = self.final.view(self.rows_per_rank, self.world, TILE, *self.final.shape[1:])
            v.copy_(self.gbuf.transpose(0, 1))                             # interleave the bands back
            return self.final
        return None


def block_tile_rows(tiles_y, rank, world):
    """Tile rows [t0, t1) of `rank` in the blocked layout (frr_set_partition_layout(1), frr_owned_rows): tiles_y // world
    rows each, the first tiles_y % world ranks one more -- no rank of world <= tiles_y is left without rows."""
    q, r = divmod(tiles_y, world)
    t0 = rank * q + min(rank, r)
    return t0, t0 + q + (1 if rank < r else 0)


def block_rows(height, rank, world):
    """Pixel rows [y0, y1) of the padded image that `rank` owns in the blocked layout."""
    tiles_y = (height + TILE - 1) // TILE
    t0, t1 = block_tile_rows(tiles_y, rank, world)
    return t0 * TILE, t1 * TILE


def tile_row_owner(tiles_y, world, blocked):
    """owner[ty] = the rank that rasterizes tile row ty (both layouts): what tests and tools stitch partial images with."""
    if not blocked:
        return [ty % world for ty in range(tiles_y)]
    owner = [0] * tiles_y
    for rank in range(world):
        t0, t1 = block_tile_rows(tiles_y, rank, world)
        for ty in range(t0, t1):
            owner[ty] = rank
    return owner


class BlockGather:
    """Blocked layout: rank r owns the contiguous pixel rows block_rows(height, r, world) of the padded image, so its
    slab goes straight from the render target into the final image on `dst`: one send per rank, the receives on `dst`
    posted into views of the final image, all in ONE group (ncclGroupStart/End under torch.distributed's "nccl"
    backend) -- the slabs differ by a tile row when the rows do not divide evenly, so this is send/recv, not a gather
    of equal parts.  The native form of the same exchange: examples/gather_rccl.cpp."""

    def __init__(self, height, width, dtype, device, rank, world, dst=0, trailing=()):
        self.rank, self.world, self.dst, self.height = rank, world, dst, height
        _, _, self.padded_height = band_layout(height, world)
        self.y0, self.y1 = block_rows(height, rank, world)
        self.final = torch.zeros((self.padded_height, width, *trailing), dtype=dtype, device=device) if rank == dst else None

    def start(self, local_image, group=None):
        """Launches the exchange of this rank's slab of `local_image` ([HP, W, ...]) without waiting for it."""
        import torch.distributed as dist
        ops = []
        if self.rank == self.dst:
            self.final[self.y0:self.y1].copy_(local_image[self.y0:self.y1])
            for r in range(self.world):
                a, b = block_rows(self.height, r, self.world)
                if r != self.dst and b > a:
                    ops.append(dist.P2POp(dist.irecv, self.final[a:b], r, group))
        elif self.y1 > self.y0:
            ops.append(dist.P2POp(dist.isend, local_image[self.y0:self.y1], self.dst, group))
        return dist.batch_isend_irecv(ops) if ops else []

    def finish(self, handles):
        for h in handles:
            h.wait()
        return self.final

    def finish_host(self, handles):
        """Like finish(), but the HOST polls for completion instead of making the current stream wait: with enough
        target sets in flight the exchange has long completed, and the render stream carries no dependency packet
        (a stream-side wait costs the 150-us frame several microseconds even when it is already satisfied)."""
        import time
        t0 = time.perf_counter()
        for h in handles:
            while not h.is_completed():
                if time.perf_counter() - t0 > 2.0:   # never expected; fall back to a full wait rather than spin forever
                    h.wait()
                    import torch
                    if torch.cuda.is_available():
                        torch.cuda.current_stream().synchronize()   # (callers rely on HOST-visible completion: bench.py re-binds the set)
                    break
        return self.final

    def __call__(self, local_image, group=None):
        return self.finish(self.start(local_image, group))


class FrameGather:
    """The final-image exchange of one frame in the blocked layout: ONE gather per plane (colour, depth, ids ...)
    of the rank's contiguous slab, straight from the render targets into the final images on `dst`.
    planes = [(dtype, trailing), ...]; start() takes the rank's padded plane images in the same order."""

    def __init__(self, height, width, planes, device, rank, world, dst=0):
        self.parts = [BlockGather(height, width, dt, device, rank, world, dst=dst, trailing=tr) for dt, tr in planes]
        self.rank, self.world, self.dst = rank, world, dst
        self.padded_height = self.parts[0].padded_height

    def start(self, images, group=None):
        assert len(images) == len(self.parts)
        return [p.start(img, group) for p, img in zip(self.parts, images)]

    def finish(self, handles):
        return [p.finish(h) for p, h in zip(self.parts, handles)]   # (None on ranks other than dst)

    def finish_host(self, handles):
        return [p.finish_host(h) for p, h in zip(self.parts, handles)]

    def __call__(self, images, group=None):
        return self.finish(self.start(images, group))
