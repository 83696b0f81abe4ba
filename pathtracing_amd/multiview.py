"""Multi-view (-rm) rendering across processes: one process per GPU.

The reference farms image rows of one view to MPI workers with blocking
point-to-point messages (mainOMP.cpp:154-220) and every rank reads the scene
and builds its own kd-tree (mainOMP.cpp:338-339).  Here the unit of
distribution is the view: views are independent, so view v goes to rank
v % world and the data path needs no collective.  The two exchanges that
remain are

* once per job: rank 0 reads the scene, builds the trees, uploads the scene and
  broadcasts its device data -- one position-independent block, GPU to GPU over
  xGMI with RCCL when the tensors live on GPUs -- which the other ranks turn
  into their scene without preparing anything again;
* per batch of rounds: the finished BGR8 frames are gathered on rank 0.  A gather makes the ranks wait for each
  other, and views differ in cost by tens of percent, so the ranks render several views each (8 by default) between
  two gathers instead of meeting after every view.

A second mode splits every single view over all ranks (run_split_views): the
reference's own distribution, rows of one view dealt round-robin to the workers
(mainOMP.cpp:169-174), for when the time to the next frame matters more than
frames per second.  The unit is a band of 8 rows, the height of the device's
pixel tiles; per view, every rank sends only its own bands to rank 0.

torch.distributed is plumbing only; with the "gloo" backend and CPU tensors the
same code runs without a GPU, which is how the tests cover it.
"""
import numpy as np


def views_of_rank(n_views, rank, world):
    """Views owned by `rank`: round-robin, the reference's row dealing (mainOMP.cpp:169-174) applied to views."""
    return list(range(rank, n_views, world))


def rounds(n_views, world):
    """Number of gather rounds: every rank renders at most one view per round."""
    return (n_views + world - 1) // world


def broadcast_bytes(payload, device, dist, src=0):
    """Broadcast a byte blob (numpy uint8, only needed on `src`) to every rank; returns numpy uint8."""
    import torch
    rank = dist.get_rank()
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        n[0] = int(payload.size)
    dist.broadcast(n, src)
    buf = torch.empty(int(n[0]), dtype=torch.uint8, device=device)
    if rank == src:
        buf.copy_(torch.from_numpy(np.ascontiguousarray(payload, np.uint8)))
    dist.broadcast(buf, src)
    return buf.cpu().numpy()


def broadcast_scene(scene, device, dist, src=0):
    """Rank `src` holds a DeviceScene; every other rank passes None and gets one made from the broadcast copy of its
    device data (the slab: all immutable arrays, the fp32 tree included).  With `device` a GPU the payload is exported
    device to device, broadcast by RCCL and imported device to device; with a CPU device (gloo rehearsal) it is staged
    through host memory.  Returns the rank's DeviceScene."""
    import torch
    from . import device as ptdev
    rank = dist.get_rank()
    on_gpu = torch.device(device).type == "cuda"
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        n[0] = scene.slab_size()
    dist.broadcast(n, src)
    buf = torch.empty(int(n[0]), dtype=torch.uint8, device=device)
    if rank == src:
        scene.slab_export(buf.data_ptr(), buf.numel(), on_gpu)
    dist.broadcast(buf, src)
    if rank == src:
        return scene
    if on_gpu:
        torch.cuda.synchronize()                     # the broadcast ran on torch's stream; the library copies on its own
    return ptdev.DeviceScene(slab=(buf.data_ptr(), buf.numel(), on_gpu))


def broadcast_cameras(cams, device, dist, src=0):
    """Broadcast an (n, 5) float64 array of (eye xyz, angle, zoom)."""
    import torch
    rank = dist.get_rank()
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        n[0] = len(cams)
    dist.broadcast(n, src)
    buf = torch.empty((int(n[0]), 5), dtype=torch.float64, device=device)
    if rank == src:
        buf.copy_(torch.from_numpy(np.ascontiguousarray(cams, np.float64)))
    dist.broadcast(buf, src)
    return buf.cpu().numpy()


class FrameGather:
    """Gathers the frames of up to `batch` rounds per rank on rank 0 in one collective and files them under their view
    indices.  Each rank puts the frame of round k into slot(k); flush() sends the slots filled since the last flush."""

    def __init__(self, frame_numel, device, dist, batch=1):
        import torch
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        self.batch, self.numel = max(1, int(batch)), frame_numel
        self.send = torch.zeros((self.batch, frame_numel), dtype=torch.uint8, device=device)
        self.recv = ([torch.empty_like(self.send) for _ in range(self.world)] if self.rank == 0 else None)
        self.first_round, self.filled = 0, 0

    def slot(self, round_index):
        """The tensor (frame_numel bytes) that takes this rank's frame of round `round_index`; rounds come in order."""
        if self.filled == 0:
            self.first_round = round_index
        assert round_index == self.first_round + self.filled and self.filled < self.batch
        self.filled += 1
        return self.send[round_index - self.first_round]

    def full(self):
        return self.filled == self.batch

    def flush(self, n_views, out=None):
        """One gather of the filled slots (every rank has filled the same number).  out: on rank 0, optional dict
        view index -> numpy frame; views at or beyond n_views (the ragged end of the last round) are dropped."""
        if self.filled == 0:
            return
        nb = self.filled
        if nb == self.batch:
            self.dist.gather(self.send, self.recv, dst=0)
        else:
            self.dist.gather(self.send[:nb].contiguous(), [t[:nb] for t in self.recv] if self.rank == 0 else None, dst=0)
        if self.rank == 0 and out is not None:
            for j in range(nb):
                for r in range(self.world):
                    v = (self.first_round + j) * self.world + r
                    if v < n_views:
                        out[v] = self.recv[r][j].cpu().numpy().copy()
        self.filled = 0

    def gather(self, frame, round_index, n_views, out=None):
        """One frame per rank, gathered at once (a batch of one round)."""
        self.slot(round_index).copy_(frame)
        self.flush(n_views, out)


def run_views(render, n_views, frame_numel, device, dist, on_frame=None, batch=8, render_many=None):
    """The -rm job (renderMultiple, mainOMP.cpp:225-247) over all ranks: view v is rendered by rank v % world.

    render(v) -> uint8 tensor of frame_numel bytes on `device` (called only for views this rank owns; the tensor may
    be reused by the next call); on_frame(v, numpy frame) is called on rank 0 for every view, in view order.  The
    ranks meet once per `batch` rounds.  Works without a process group (world 1).  Returns the number of rounds.

    render_many(first_view, stride, count) -> uint8 numpy array (count, frame_numel), the frames of views first_view,
    first_view + stride, ...: when given, a rank renders the views of a whole batch of rounds in one call (the library
    then keeps several of them in flight on the GPU) instead of one `render` call per view."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    gather = FrameGather(frame_numel, device, dist, batch) if world > 1 else None
    n_rounds = rounds(n_views, world)

    def deliver():
        got = {} if rank == 0 else None
        gather.flush(n_views, got)
        if rank == 0 and on_frame is not None:
            for vv in sorted(got):
                on_frame(vv, got[vv])

    if render_many is not None:
        import torch
        for k0 in range(0, n_rounds, max(1, int(batch))):
            nb = min(max(1, int(batch)), n_rounds - k0)
            first = k0 * world + rank
            mine = len(range(first, min(n_views, (k0 + nb) * world), world)) if first < n_views else 0
            frames = render_many(first, world, mine) if mine else None
            if world == 1:
                if on_frame is not None:
                    for j in range(mine):
                        on_frame(first + j, frames[j])
                continue
            for j in range(nb):
                slot = gather.slot(k0 + j)
                if j < mine:
                    slot.copy_(torch.from_numpy(frames[j]))
                else:
                    slot.zero_()
            deliver()
        return n_rounds

    for k in range(n_rounds):
        v = k * world + rank
        if world == 1:
            frame = render(v)
            if on_frame is not None:
                on_frame(v, frame.cpu().numpy().copy())
            continue
        slot = gather.slot(k)
        if v < n_views:
            slot.copy_(render(v))
        else:
            slot.zero_()
        if gather.full():
            deliver()
    if world > 1:
        deliver()
    return n_rounds


BAND_ROWS = 8       # rows per band of the row split: PtOptions.part / n_parts count bands of this height


def bands_of_rank(size, rank, world):
    """Row bands of a size x size frame owned by `rank`: band b (rows 8b .. 8b+7) goes to rank b % world."""
    return list(range(rank, (size + BAND_ROWS - 1) // BAND_ROWS, world))


class BandGather:
    """Collects on rank 0 the row bands that every rank rendered of ONE view and interleaves them into the frame."""

    def __init__(self, size, device, dist):
        import torch
        self.torch, self.dist, self.size = torch, dist, size
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_bands = (size + BAND_ROWS - 1) // BAND_ROWS
        self.band_bytes = BAND_ROWS * size * 3
        self.share = (self.n_bands + self.world - 1) // self.world          # bands per rank, the last ranks may own one fewer
        self.padded = torch.zeros(self.n_bands * self.band_bytes, dtype=torch.uint8, device=device)
        self.send = torch.zeros(self.share * self.band_bytes, dtype=torch.uint8, device=device)
        self.recv = ([torch.empty_like(self.send) for _ in range(self.world)] if self.rank == 0 else None)

    def gather(self, frame):
        """frame: uint8 tensor of size*size*3 bytes whose own bands are rendered.  Returns the whole frame (numpy) on rank 0."""
        n = self.size * self.size * 3
        self.padded[:n] = frame
        mine = self.padded.view(self.n_bands, self.band_bytes)[self.rank::self.world]
        self.send.view(self.share, self.band_bytes)[: mine.shape[0]] = mine
        self.dist.gather(self.send, self.recv, dst=0)
        if self.rank != 0:
            return None
        whole = self.padded.view(self.n_bands, self.band_bytes)
        for r in range(self.world):
            k = len(range(r, self.n_bands, self.world))
            whole[r::self.world] = self.recv[r].view(self.share, self.band_bytes)[:k]
        return self.padded[:n].cpu().numpy().copy()


def run_split_views(render_part, n_views, size, device, dist, on_frame=None):
    """Every view rendered by all ranks together, rank r the row bands b with b % world == r (renderParallel's
    row dealing, mainOMP.cpp:169-174, in bands of 8 rows).

    render_part(v, part, n_parts) -> uint8 tensor of size*size*3 bytes on `device`, rows outside the part's bands
    unspecified; on_frame(v, numpy frame) is called on rank 0 for every view in order.  Works without a process
    group (world 1).  Returns the number of views."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    gather = BandGather(size, device, dist) if world > 1 else None
    for v in range(n_views):
        frame = render_part(v, rank, world)
        whole = gather.gather(frame) if world > 1 else frame.cpu().numpy().copy()
        if rank == 0 and on_frame is not None:
            on_frame(v, whole)
    return n_views
