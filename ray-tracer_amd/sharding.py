"""Image-tile sharding of one frame across the GPUs of a node: one process per GPU.

The reference renders on a single GPU (one dispatch covers the frame, src/renderer.rs:865-867), so
this module has no counterpart there. Pixels are independent and the scene is read-only, so the
frame shards without any data-path exchange until the end:

  scene     rank 0 holds the wire-format arrays; they are broadcast once per scene (RCCL when the
            tensors live on GPUs) and every rank bins its own grid on its GPU - binning is
            deterministic, so all ranks hold identical cell lists without shipping them
  tiles     the frame is cut into tile_w x tile_h tiles dealt round-robin: rank r traces tiles
            r, r+N, r+2N, ... (sky tiles are cheap, dense tiles expensive: interleaving balances)
  gather    every rank's packed tiles go to rank 0 in ONE gather; on xGMI that is seven point-to-point
            transfers into rank 0 over seven separate links, no ring. What travels is either the
            shaded RGBA8 picture (4 bytes per pixel) or the hit records (16 bytes per pixel with
            shadow records); the links into rank 0 carry (N-1)/N of a frame, so the byte count per
            pixel decides how many frames per second N GPUs can deliver
  assemble  rank 0 scatters the gathered tiles into the row-major frame (rt_assemble_tiles[_rgba8])

No step synchronises the host. Frames can be kept in flight (separate streams and buffers per frame), which
overlaps a frame's gather, and the long tail of its trace kernel, with the next frame's tracing.
For throughput (a camera path rendered ahead of display) frames travel in GROUPS: `trace_group` renders this rank's
tiles of up to 64 consecutive frames in one to four launches (rt_render_frames), gathers them in one collective and assembles
them in one launch, so launch, collective and host costs are paid once per group instead of once per frame, and
the expensive pixel blocks of one frame overlap the other frames of the group. Groups are kept in flight like
frames.
Tiles that show nothing but sky need not travel at all (`enable_sparse`): which tiles those are follows from the camera
alone (TraceCore.tile_flags: the frame's block mask reduced to tiles, identical on every rank), so for cameras announced
ahead (`prepare_cameras`) the ranks deal only the other tiles among themselves, send only those, and rank 0's assemble
kernel writes the sky tiles from the camera. On the bundled static model's frame a third of the 32x32 tiles are sky:
as many bytes less on the links into rank 0, which is what bounds the frame rate at 8 GPUs.
The tracer is passed in (`core`): the product passes a ray_tracer_amd.TraceCore; the CPU tests pass
a stand-in so the sharding logic runs under gloo without a GPU.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._lib import RT_OWNER_SKY, RtShard
from .host import shard_pixels, shared_stream


def n_tiles(width, height, tile_w, tile_h):
    return (-(-width // tile_w)) * (-(-height // tile_h))


def per_rank_pixels(width, height, tile_w, tile_h, world):
    """Records in one rank's block of the gather: ceil(n_tiles / world) whole tiles."""
    return -(-n_tiles(width, height, tile_w, tile_h) // world) * tile_w * tile_h


def owner_table(weights):
    """Which rank traces each tile of a round, for ranks that own weights[r] tiles of every sum(weights): the tiles are
    dealt round-robin among the ranks that still have some to take, so every rank's tiles are spread over the round
    (and with it over the frame) as evenly as its weight allows. [3, 4, 4] -> 0 1 2 0 1 2 0 1 2 1 2."""
    left = [int(k) for k in weights]
    table = []
    while any(left):
        for r, k in enumerate(left):
            if k:
                table.append(r)
                left[r] -= 1
    return table


def gatherer_weights(world, resolution=64, lag=3):
    """Tiles per round that leave the gathering rank (0) `lag` parts in `resolution` short of an equal share:
    [K - lag, K, K, ...] with K = resolution / world. On a 1080p frame of the bundled model the gathering rank spends
    about 5 % of a single GPU's frame time on assembling (while it renders), which the others do not."""
    if world < 2:
        return None
    k = max(lag + 1, round(resolution / world))
    return [k - lag] + [k] * (world - 1)


def init_comm(core, rank=None, world=None, src=0):
    """Gives `core` (a TraceCore) an RCCL communicator over all ranks of the torch.distributed job, through the C ABI
    (rt_comm_unique_id on rank `src`, the 128 bytes handed round with the job's own process group, rt_comm_init on
    every rank). From then on FrameSharder gathers with rt_gather_frames and `replicate_scene` broadcasts with
    rt_broadcast_scene: RCCL on the trace core's own stream, device to device, no torch tensors in the data path."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    # ncclCommInitRank is itself a collective, and so is handing the id round: a rank that cannot bind RCCL must not
    # leave the others waiting in either. Every rank first tries to make an id (which binds the library and touches
    # nothing else), the outcomes are exchanged, and all ranks raise together or go on together.
    try:
        mine = ("ok", core.comm_unique_id())
    except Exception as e:  # noqa: BLE001
        mine = ("error", f"rank {rank}: {e}")
    everyone = [mine]
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    bad = [m for status, m in everyone if status != "ok"]
    if bad:
        raise RuntimeError("no RCCL communicator: " + "; ".join(bad))
    core.comm_init(everyone[src][1], rank, world)


def replicate_scene(core, boxes, tris, src=0):
    """Rank `src`'s scene arrays become every rank's scene: rt_broadcast_scene when the core has a communicator
    (init_comm), else the torch.distributed fall-back below followed by set_scene."""
    if getattr(core, "comm_size", 0):
        core.broadcast_scene(src, boxes, tris)
        return
    b, t = broadcast_scene(boxes, tris, torch.device("cuda", core.device_id))
    core.set_scene(b, t)


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


def _order_after_fill(stream, device):
    """The buffers of a slot are filled (zeros / "miss") on torch's current stream; the slot's own stream, which is
    non-blocking and carries the first frame's kernels, must not start before those fills have finished."""
    if stream is not None:
        stream.wait_stream(torch.cuda.current_stream(device))


class _Slot:
    """Buffers and stream of one frame in flight."""

    def __init__(self, sharder, device, k=0):
        s = sharder
        self.stream = shared_stream(device, k) if torch.device(device).type == "cuda" else None
        if s.mode == "rgba":
            # [record][r, g, b, a]; tiles this rank does not own stay zero
            self.local = torch.zeros((s.per_rank, 4), dtype=torch.uint8, device=device)
        else:
            # [block (primary, shadow)][record][id, t-bits]; tiles this rank does not own stay "miss"
            self.local = torch.full((s.blocks, s.per_rank, 2), -1, dtype=torch.int32, device=device)
        self.packed = self.gathered = self.image = None
        if s.rank == 0:
            # the gather writes rank r's block straight into packed[r]
            self.packed = torch.empty((s.world,) + tuple(self.local.shape), dtype=self.local.dtype, device=device)
            self.gathered = [self.packed[r] for r in range(s.world)]
            if s.mode == "rgba":
                self.image = torch.empty((s.height * s.width, 4), dtype=torch.uint8, device=device)
            else:
                self.image = torch.empty((s.blocks, s.height * s.width, 2), dtype=torch.int32, device=device)
        _order_after_fill(self.stream, device)


class _GroupSlot:
    """Buffers and stream of one group of frames in flight: [frame][record][r, g, b, a] per rank."""

    def __init__(self, sharder, device, k=0):
        s = sharder
        self.stream = shared_stream(device, k) if torch.device(device).type == "cuda" else None
        self.local = torch.zeros((s.group, s.per_rank, s.wire_bytes), dtype=torch.uint8, device=device)
        self.packed = self.images = None
        self.views = {}
        if s.rank == 0:
            self.packed = torch.empty((s.world, s.group, s.per_rank, s.wire_bytes), dtype=torch.uint8, device=device)
            self.images = torch.empty((s.group, s.height * s.width, 4), dtype=torch.uint8, device=device)
        _order_after_fill(self.stream, device)


class FrameSharder:
    """Traces this rank's tiles of a frame and gathers the frame on rank 0.

    mode "hits": rt_hit records, primary (and shadow) blocks, 8 (16) bytes per pixel on the wire.
    mode "rgba": the shaded picture of TraceCore.render, 4 bytes per pixel on the wire - the framebuffer proper;
                 with shadow=True the lighting includes the shadow-ray term, so the rays traced are the same
                 primary + shadow set as in "hits" mode (max_bounces bounds the reflection rays on top).
    frames_in_flight: consecutive frames use separate streams and buffers, so a rank traces frame k+1 while frame
                 k is on the links and while the last waves of frame k drain. The images returned for a frame stay
                 valid until that many further frames have been started."""

    def __init__(self, core, width, height, tile_w=64, tile_h=64, rank=None, world=None, device="cuda", shadow=True,
                 mode="hits", frames_in_flight=1, max_bounces=8, collective=None, weights=None, host_gather=False):
        if mode not in ("hits", "rgba"):
            raise ValueError("mode must be 'hits' or 'rgba'")
        self.core, self.width, self.height, self.tile_w, self.tile_h = core, width, height, tile_w, tile_h
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.shadow, self.mode, self.max_bounces = shadow, mode, max_bounces
        # the gather is skipped for a lone rank unless asked for (a one-rank process group exercises the backend)
        self.collective = self.world > 1 if collective is None else bool(collective)
        self.device = device
        # a core with a communicator of its own (init_comm) gathers through the C ABI; the CPU tests' stand-in cores
        # and cores without one go through torch.distributed
        # host_gather: the torch.distributed gather of frame groups runs on host copies (device -> host, gloo on CPU
        # tensors, host -> device), synchronously. Only for rehearsing several ranks on ONE GPU, where RCCL is not
        # available and gloo's own handling of device tensors (private streams, pinned staging, host-blocking waits
        # in every process that shares the GPU) stalled for seconds to minutes.
        self.host_gather = bool(host_gather)
        self.native_comm = bool(getattr(core, "comm_size", 0))
        if self.native_comm and (core.comm_size != self.world or core.comm_rank != self.rank):
            raise ValueError("the core's communicator and the sharder disagree on rank / world size")
        # weights[r]: tiles rank r owns out of every sum(weights) ("rgba" mode), dealt as owner_table says. Rank 0 also
        # gathers and assembles, so it is given less to render than the others; None = one tile each.
        self.weights = None if weights is None else [int(k) for k in weights]
        self.owner = None
        if self.weights is not None:
            if mode != "rgba":
                raise ValueError("weights need the picture path: mode must be 'rgba'")
            if len(self.weights) != self.world or min(self.weights) < 1:
                raise ValueError("weights: one positive integer per rank")
            self.owner = owner_table(self.weights)
            shards = [RtShard(tile_w, tile_h, r, owner=self.owner) for r in range(self.world)]
            self.shard = shards[self.rank]
            self.per_rank = max(shard_pixels(width, height, s) for s in shards)
        else:
            self.shard = RtShard(tile_w, tile_h, self.rank, self.world)
            self.per_rank = per_rank_pixels(width, height, tile_w, tile_h, self.world)
        self.blocks = 2 if shadow else 1
        # Trace kernels, the gather and the assemble kernel of a frame share one in-order stream: the collective
        # only waits for work queued on torch's current stream, and makes that stream wait for its completion.
        self.slots = [_Slot(self, device, k) for k in range(max(1, frames_in_flight))]
        self.frame = 0
        self.sparse, self._tile_flags, self._plans = False, {}, {}
        self.last_wire_bytes = 0  # bytes one rank's block of the most recent group holds (what a rank sends)
        self.group, self.gslots, self.group_no, self.wire_bytes = 0, [], 0, 4
        self.local, self.packed, self.image = self.slots[0].local, self.slots[0].packed, self.slots[0].image

    def _local(self, cam, slot):
        if self.mode == "rgba":
            self.core.render(cam, self.width, self.height, shard=self.shard, max_bounces=self.max_bounces,
                             shadows=self.shadow, out=slot.local)
        else:
            self.core.trace_primary(cam, self.width, self.height, shard=self.shard, shadow=self.shadow,
                                    out_primary=slot.local[0], out_shadow=slot.local[1] if self.shadow else None)
        return slot.local

    def trace_local(self, cam):
        """This rank's tiles only (no communication): the per-GPU share of the frame."""
        return self._local(cam, self.slots[0])

    def trace(self, cam):
        """One frame. On rank 0 returns [primary, shadow] images (height*width, 2) int32 in "hits" mode, the
        (height*width, 4) uint8 picture in "rgba" mode; None on the other ranks."""
        slot = self.slots[self.frame % len(self.slots)]
        self.frame += 1
        if slot.stream is not None:
            self.core.set_stream(slot.stream.cuda_stream)
            with torch.cuda.stream(slot.stream):
                return self._trace(cam, slot)
        return self._trace(cam, slot)

    def synchronize(self):
        for slot in self.slots + self.gslots:
            if slot.stream is not None:
                slot.stream.synchronize()

    # ---- groups of frames ("rgba" mode)
    def enable_groups(self, group=8, groups_in_flight=2, rgb8=False):
        """Allocates buffers for `groups_in_flight` groups of up to `group` frames each. rgb8: the pictures travel
        without their constant alpha byte (3 bytes per pixel on the links instead of 4); rank 0 puts it back."""
        if self.mode != "rgba":
            raise ValueError("frame groups carry the shaded picture: mode must be 'rgba'")
        if not 1 <= group <= 64:
            raise ValueError("a group holds 1 to 64 frames (up to four rt_render_frames launches of 16)")
        self.group = group
        self.wire_bytes = 3 if rgb8 else 4
        self.gslots = [_GroupSlot(self, self.device, k) for k in range(max(1, groups_in_flight))]

    # ---- sky tiles stay at home
    def enable_sparse(self):
        """Groups whose cameras have all been announced (prepare_cameras) leave their sky tiles out: nobody renders or
        sends them, rank 0 writes them from the camera. Other groups travel whole, as without this call."""
        if self.mode != "rgba" or not self.gslots:
            raise ValueError("sky tiles are left out of frame groups: mode 'rgba', enable_groups first")
        if self.tile_w % 8 or self.tile_h % 8:
            raise ValueError("tiles must be multiples of 8 pixels (the block mask's raster)")
        if n_tiles(self.width, self.height, self.tile_w, self.tile_h) > 65536:
            raise ValueError("more than 65 536 tiles: an owner table over the whole frame does not fit")
        self.sparse = True

    def prepare_cameras(self, cams):
        """Classifies the tiles of every camera not seen before (one small synchronous GPU pass each, identical on every
        rank). Call it ahead of the frames - at scene load for a fixed camera or a recorded path - on EVERY rank with the
        same cameras: with more than one rank the ranks compare what they found (one small all-reduce per new camera),
        and a camera whose tiles they do not agree on is left unannounced (its groups travel whole), because ranks that
        dealt the tiles differently would wait for each other for ever."""
        for cam in cams:
            key = bytes(cam)
            if key in self._tile_flags:
                continue
            flags = np.asarray(self.core.tile_flags(cam, self.width, self.height, self.tile_w, self.tile_h), np.uint8) != 0
            if self.world > 1 and self.collective and dist.is_available() and dist.is_initialized():
                digest = int(np.flatnonzero(flags).astype(np.int64).sum() * 31 + int(flags.sum())) % (1 << 40)
                where = self.device if dist.get_backend() == "nccl" else "cpu"
                both = torch.tensor([digest, -digest], dtype=torch.int64, device=where)
                dist.all_reduce(both, op=dist.ReduceOp.MAX)
                if int(both[0]) != -int(both[1]):  # max(digest) != min(digest)
                    continue
            self._tile_flags[key] = flags

    def _sparse_plan(self, cams):
        """Who renders what for this group, or None when one of its cameras has not been announced. A tile is left out
        only if it is sky in EVERY frame of the group (one launch renders the group with one shard); the kept tiles are
        dealt among the ranks in the pattern of owner_table(weights), so a rank's tiles stay spread over the frame."""
        keys = tuple(bytes(cam) for cam in cams)
        plan = self._plans.get(keys)
        if plan is not None:
            return plan
        if any(k not in self._tile_flags for k in keys):
            return None
        kept = np.zeros(n_tiles(self.width, self.height, self.tile_w, self.tile_h), bool)
        for k in set(keys):
            kept |= self._tile_flags[k]
        owner = np.full(len(kept), RT_OWNER_SKY, np.uint32)
        idx = np.flatnonzero(kept)
        owner[idx] = np.resize(np.asarray(owner_table(self.weights or [1] * self.world), np.uint32), len(idx))
        most = int(np.bincount(owner[idx], minlength=self.world).max()) if len(idx) else 0
        plan = {"owner": [int(k) for k in owner], "per_rank": max(1, most) * self.tile_w * self.tile_h, "kept_tiles": int(kept.sum())}
        plan["shard"] = RtShard(self.tile_w, self.tile_h, self.rank, owner=plan["owner"])
        if hasattr(self.core, "_L"):  # the C ABI's caller: the table and the cameras as ctypes arrays, made once per plan
            import ctypes as C
            from ._lib import CameraUniform
            plan["owner_c"] = (C.c_uint32 * len(plan["owner"]))(*plan["owner"])
            arr = (CameraUniform * len(cams))()
            for j, cam in enumerate(cams):
                C.memmove(C.byref(arr, j * C.sizeof(CameraUniform)), C.byref(cam), C.sizeof(CameraUniform))
            plan["cams_c"] = arr
        if len(self._plans) > 64:
            self._plans.clear()
        self._plans[keys] = plan
        return plan

    def trace_group(self, cams):
        """len(cams) <= group consecutive frames. On rank 0 returns the (len(cams), height*width, 4) uint8 pictures
        (valid until `groups_in_flight` further groups have been started); None on the other ranks."""
        if not self.gslots:
            self.enable_groups()
        g = len(cams)
        if not 1 <= g <= self.group:
            raise ValueError(f"a group holds 1 to {self.group} frames")
        slot = self.gslots[self.group_no % len(self.gslots)]
        self.group_no += 1
        if slot.stream is not None:
            self.core.set_stream(slot.stream.cuda_stream)
            with torch.cuda.stream(slot.stream):
                return self._trace_group(cams, g, slot)
        return self._trace_group(cams, g, slot)

    def _trace_group(self, cams, g, slot):
        rgb8 = self.wire_bytes == 3
        plan = self._sparse_plan(cams) if self.sparse else None
        # one rank's block of the gather: `group` frames of `stride` pixels each. With sky tiles left out a block holds
        # fewer tiles; it sits at the front of the same buffers.
        shard, stride = (plan["shard"], plan["per_rank"]) if plan else (self.shard, self.per_rank)
        block = self.group * stride * self.wire_bytes  # bytes (unused frames of a short group travel too: a group is short only at the end of a path)
        self.last_wire_bytes = block
        local = slot.local.view(-1)[:block].view(self.group, stride, self.wire_bytes)
        packed = slot.packed.view(-1)[:self.world * block].view(self.world, self.group, stride, self.wire_bytes) if self.rank == 0 else None
        for j in range(0, g, 16):  # one launch renders up to 16 frames
            self.core.render_frames(cams[j:j + 16], self.width, self.height, shard=shard, max_bounces=self.max_bounces,
                                    shadows=self.shadow, out=local[j:min(g, j + 16)], rgb8=rgb8, frame_stride=stride)
        if self.collective and self.native_comm:
            # rt_gather_frames on the trace core's stream: whole per-rank blocks
            self.core.gather_frames(0, local, block, packed)
        elif self.collective and self.host_gather:
            if slot.stream is not None:
                slot.stream.synchronize()
            mine = local[:g].cpu()
            parts = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(mine, parts, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    packed[r, :g].copy_(parts[r])
        elif self.collective:
            key = (g, stride)
            if key not in slot.views:  # tensor views are not free to make: once per group length and layout
                if len(slot.views) > 16:
                    slot.views.clear()
                slot.views[key] = (local[:g], [packed[r, :g] for r in range(self.world)] if self.rank == 0 else None)
            mine, parts = slot.views[key]
            dist.gather(mine, parts, dst=0)
        if self.rank != 0:
            return None
        src = packed if self.collective else local.unsqueeze(0)
        if plan:
            self.core.assemble_frames_sky(src, rgb8, self.width, self.height, self.tile_w, self.tile_h, self.world,
                                          plan.get("owner_c", plan["owner"]), plan.get("cams_c", cams), self.group * stride, stride,
                                          slot.images[:g])
        else:
            self.core.assemble_frames(src, rgb8, self.width, self.height, self.tile_w, self.tile_h, self.world, self.owner, g,
                                      self.group * stride, stride, slot.images[:g])
        return slot.images[:g]

    def _trace(self, cam, slot):
        self._local(cam, slot)
        if self.collective and self.native_comm:
            self.core.gather_frames(0, slot.local, slot.local.numel() * slot.local.element_size(), slot.packed)
        elif self.collective:
            dist.gather(slot.local, slot.gathered, dst=0)
        if self.rank != 0:
            return None
        src = slot.packed if self.collective else slot.local.unsqueeze(0)
        if self.mode == "rgba":
            self.core.assemble_frames(src[0], False, self.width, self.height, self.tile_w, self.tile_h, self.world,
                                      self.owner, 1, self.per_rank, self.per_rank, slot.image)
            return slot.image
        for b in range(self.blocks):
            self.core.assemble_tiles(src[0, b], self.width, self.height, self.tile_w, self.tile_h, self.world,
                                     slot.image[b], rank_stride=self.blocks * self.per_rank)
        return [slot.image[b] for b in range(self.blocks)]
