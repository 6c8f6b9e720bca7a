"""Image-tile sharding of one frame across the GPUs of a node: one process per GPU.

The reference renders on a single GPU (one dispatch covers the frame, src/renderer.rs:865-867), so
this module has no counterpart there. Pixels are independent and the scene is read-only, so the
frame shards without any data-path exchange until the end:

  scene     rank 0 holds the wire-format arrays; they are broadcast once per scene (RCCL when the
            tensors live on GPUs) and every rank bins its own grid on its GPU - binning is
            deterministic, so all ranks hold identical cell lists without shipping them
  tiles     the frame is cut into tile_w x tile_h tiles dealt round-robin: rank r traces tiles
            r, r+N, r+2N, ... (sky tiles are cheap, dense tiles expensive: interleaving balances)
  gather    every rank's packed hit records go to rank 0 in ONE gather (primary and shadow blocks
            travel together); on xGMI that is seven point-to-point transfers into rank 0 over
            seven separate links, no ring
  assemble  rank 0 scatters the gathered tiles into the row-major frame (rt_assemble_tiles)

No step synchronises the host: calls are queued on the process's current torch stream.
The tracer is passed in (`core`): the product passes a ray_tracer_amd.TraceCore; the CPU tests pass
a stand-in so the sharding logic runs under gloo without a GPU.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._lib import RtShard


def n_tiles(width, height, tile_w, tile_h):
    return (-(-width // tile_w)) * (-(-height // tile_h))


def per_rank_pixels(width, height, tile_w, tile_h, world):
    """Records in one rank's block of the gather: ceil(n_tiles / world) whole tiles."""
    return -(-n_tiles(width, height, tile_w, tile_h) // world) * tile_w * tile_h


def broadcast_scene(boxes, tris, device, src=0):
    """Replicates rank `src`'s scene arrays (numpy structured arrays in the wire format; ignored on
    the other ranks) and returns (boxes, tris) as numpy arrays on every rank."""
    from ._lib import BOX_DTYPE, TRIANGLE_DTYPE
    rank = dist.get_rank()
    counts = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        counts[0], counts[1] = len(boxes), len(tris)
    dist.broadcast(counts, src)
    nb, nt = int(counts[0]), int(counts[1])
    blob = torch.empty(nb * BOX_DTYPE.itemsize + nt * TRIANGLE_DTYPE.itemsize, dtype=torch.uint8, device=device)
    if rank == src:
        raw = np.concatenate([np.ascontiguousarray(boxes, BOX_DTYPE).view(np.uint8).reshape(-1),
                              np.ascontiguousarray(tris, TRIANGLE_DTYPE).view(np.uint8).reshape(-1)])
        blob.copy_(torch.from_numpy(raw))
    if blob.numel():
        dist.broadcast(blob, src)
    raw = blob.cpu().numpy()
    b = raw[:nb * BOX_DTYPE.itemsize].view(BOX_DTYPE).copy()
    t = raw[nb * BOX_DTYPE.itemsize:].view(TRIANGLE_DTYPE).copy()
    return b, t


class FrameSharder:
    """Traces this rank's tiles of a frame and gathers the frame on rank 0."""

    def __init__(self, core, width, height, tile_w=64, tile_h=64, rank=None, world=None, device="cuda", shadow=True):
        self.core, self.width, self.height, self.tile_w, self.tile_h = core, width, height, tile_w, tile_h
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.shadow = shadow
        self.device = device
        # Trace kernels, the gather and the assemble kernel must share one in-order stream: the collective
        # only waits for work queued on torch's current stream.
        self.stream = None
        if torch.device(device).type == "cuda":
            self.stream = torch.cuda.Stream(device)
            core.set_stream(self.stream.cuda_stream)
        self.shard = RtShard(tile_w, tile_h, self.rank, self.world)
        self.per_rank = per_rank_pixels(width, height, tile_w, tile_h, self.world)
        self.blocks = 2 if shadow else 1
        # [block (primary, shadow)][record][id, t-bits]; tiles this rank does not own stay "miss"
        self.local = torch.full((self.blocks, self.per_rank, 2), -1, dtype=torch.int32, device=device)
        if self.rank == 0:
            # the gather writes rank r's [primary | shadow] block straight into packed[r]
            self.packed = torch.empty((self.world, self.blocks, self.per_rank, 2), dtype=torch.int32, device=device)
            self.gathered = [self.packed[r] for r in range(self.world)]
            self.image = torch.empty((self.blocks, height * width, 2), dtype=torch.int32, device=device)
        else:
            self.gathered = self.packed = self.image = None

    def trace_local(self, cam):
        """This rank's tiles only (no communication): the per-GPU share of the frame."""
        self.core.trace_primary(cam, self.width, self.height, shard=self.shard, shadow=self.shadow,
                                out_primary=self.local[0], out_shadow=self.local[1] if self.shadow else None)
        return self.local

    def trace(self, cam):
        """One frame. Returns [primary, shadow] images (height*width, 2) int32 on rank 0, None elsewhere."""
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                return self._trace(cam)
        return self._trace(cam)

    def synchronize(self):
        if self.stream is not None:
            self.stream.synchronize()

    def _trace(self, cam):
        self.trace_local(cam)
        if self.world > 1:
            dist.gather(self.local, self.gathered, dst=0)
        if self.rank != 0:
            return None
        src = self.packed if self.world > 1 else self.local.unsqueeze(0)
        for b in range(self.blocks):
            self.core.assemble_tiles(src[0, b], self.width, self.height, self.tile_w, self.tile_h, self.world,
                                     self.image[b], rank_stride=self.blocks * self.per_rank)
        return [self.image[b] for b in range(self.blocks)]
