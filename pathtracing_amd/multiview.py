"""Multi-view (-rm) rendering across processes: one process per GPU.

The reference farms image rows of one view to MPI workers with blocking
point-to-point messages (mainOMP.cpp:154-220) and every rank reads the scene
and builds its own kd-tree (mainOMP.cpp:338-339).  Here the unit of
distribution is the view: views are independent, so view v goes to rank
v % world and the data path needs no collective.  The two exchanges that
remain are

* once per job: rank 0 reads the scene, builds the tree and broadcasts the
  packed scene blob (RCCL over xGMI when the tensors live on GPUs);
* per round of views: the finished BGR8 frames are gathered on rank 0.

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
    """Gathers one frame per rank per round on rank 0 and files it under its view index."""

    def __init__(self, frame_numel, device, dist):
        import torch
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        self.recv = ([torch.empty(frame_numel, dtype=torch.uint8, device=device) for _ in range(self.world)]
                     if self.rank == 0 else None)

    def gather(self, frame, round_index, n_views, out=None):
        """frame: this rank's uint8 tensor for view round_index * world + rank (ignored beyond n_views).
        out: on rank 0, optional dict view index -> numpy frame."""
        self.dist.gather(frame, self.recv, dst=0)
        if self.rank == 0 and out is not None:
            for r in range(self.world):
                v = round_index * self.world + r
                if v < n_views:
                    out[v] = self.recv[r].cpu().numpy().copy()


def run_views(render, n_views, frame_numel, device, dist, on_frame=None):
    """The -rm job (renderMultiple, mainOMP.cpp:225-247) over all ranks: view v is rendered by rank v % world.

    render(v) -> uint8 tensor of frame_numel bytes on `device` (called only for views this rank owns);
    on_frame(v, numpy frame) is called on rank 0 for every view, in view order within a round.
    Works without a process group (world 1).  Returns the number of rounds."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    gather = FrameGather(frame_numel, device, dist) if world > 1 else None
    blank = torch.zeros(frame_numel, dtype=torch.uint8, device=device)
    n_rounds = rounds(n_views, world)
    for k in range(n_rounds):
        v = k * world + rank
        frame = render(v) if v < n_views else blank
        if world == 1:
            if on_frame is not None:
                on_frame(v, frame.cpu().numpy().copy())
            continue
        got = {} if rank == 0 else None
        gather.gather(frame, k, n_views, got)
        if rank == 0 and on_frame is not None:
            for vv in sorted(got):
                on_frame(vv, got[vv])
    return n_rounds
