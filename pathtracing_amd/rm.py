"""The multi-view job of the reference (`mpirun ./mainOMP <scene> -gi [-rb] -rm`, mainOMP.cpp:225-247,328-354) on GPUs.

    python -m pathtracing_amd.rm <scene> [-gi] [-rb] [--model omp|main] [--spp N] [--size N] [--views N] [--out DIR] [--split-rows] [--resume]
    python -m torch.distributed.run --nproc-per-node N -m pathtracing_amd.rm <scene> -gi ...

One process per GPU; rank 0 reads the scene, builds the trees and uploads it, its device data travels to the other ranks
in one broadcast (RCCL, GPU to GPU, when the backend is nccl), view v of scenes/paths/<scene>path.txt x 16 angles is rendered by rank
v % N, and rank 0 collects the frames and writes <out>/<scene><v>.ppm.  Frames do not depend on N.
With --split-rows every view is rendered by all N GPUs together instead, each the 8-row bands b with b % N == rank
(the reference's own dealing of rows to its workers, mainOMP.cpp:169-174): same frames, shorter time to each frame.
With --resume the job starts at the first view whose file is missing or incomplete (the reference restarts from view 0
and only shows its frames: "change to imwrite", mainOMP.cpp:242); frames do not depend on where a job started.
"""
import argparse
import os
import queue
import sys
import threading
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m pathtracing_amd.rm", description=__doc__.split("\n")[0])
    ap.add_argument("scene")
    ap.add_argument("-gi", action="store_true")
    ap.add_argument("-rb", action="store_true")
    ap.add_argument("--model", default="omp", choices=["main", "omp"])
    ap.add_argument("--spp", type=int, default=0, help="0: the reference's default (16 for mainOMP, 4 for main)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--views", type=int, default=-1, help="render only the first N views of the path")
    ap.add_argument("--out", default=".")
    ap.add_argument("--root", default=os.getcwd(), help="directory that holds scenes/")
    ap.add_argument("--split-rows", action="store_true", help="all GPUs share every view (bands of 8 rows) instead of one view each")
    ap.add_argument("--exact-gi", action="store_true", help="fp64 GI rays (bit-identical to the CPU oracle)")
    ap.add_argument("--writers", type=int, default=4, help="threads of rank 0 that write the frame files")
    ap.add_argument("--resume", action="store_true", help="skip the views at the start of the path whose files are already complete")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: every rank on GPU 0, exchanges staged through host memory (rehearsal on a one-GPU box)")
    a = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from . import device, host, multiview

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("pathtracing_amd.rm needs a GPU")
    gpu = local_rank if a.backend == "nccl" else 0
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    xdev = dev if a.backend == "nccl" else torch.device("cpu")
    if world > 1:
        dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
    device.init(gpu)

    ds = cams = None
    first_view = 0
    if rank == 0:
        sdir = os.path.join(a.root, "scenes", a.scene)
        hs = (host.HostScene.from_dir(sdir) if os.path.exists(os.path.join(sdir, "master.txt"))
              else host.HostScene.from_packed(sdir + ".ptscene"))
        st = hs.build_tree()
        print(f"Real Triangles: {st['triangles']}; Naive: {st['naive']}; Optimized: {st['optimized']}")     # mainOMP.cpp:348
        print(f"World Size: {world}")
        views = host.views_from_path(os.path.join(a.root, "scenes", "paths", a.scene + "path.txt"), hs.camera()["zoom"])
        cams = np.array([[*v.eye, v.angle, v.zoom] for v in views])
        if a.views >= 0:
            cams = cams[: a.views]
        ds = device.DeviceScene(hs.desc())                    # the only rank that prepares the scene (fp32 tree included)
        if a.resume:
            # a PPM frame of this size has one length; the first view whose file has another (or none) is where the job goes on
            want = len(b"P6\n%d %d\n255\n" % (a.size, a.size)) + a.size * a.size * 3
            while first_view < len(cams) and os.path.exists(os.path.join(a.out, f"{a.scene}{first_view}.ppm")) and \
                    os.path.getsize(os.path.join(a.out, f"{a.scene}{first_view}.ppm")) == want:
                first_view += 1
            print(f"Resuming at view {first_view} of {len(cams)}")
            cams = np.concatenate([cams[first_view:], np.full((1, 5), float(first_view))])     # the offset travels with the cameras
    if world > 1:
        ds = multiview.broadcast_scene(ds, xdev, dist)        # its device data, GPU to GPU
        cams = multiview.broadcast_cameras(cams, xdev, dist)
    if a.resume:
        first_view, cams = int(cams[-1][0]), cams[:-1]
    n = a.size
    frame = torch.empty(n * n * 3, dtype=torch.uint8, device=dev)
    opt = dict(gi=a.gi, rb=a.rb, model=device.MODEL_OMP if a.model == "omp" else device.MODEL_MAIN, spp=a.spp, size=n,
               gi_precision=device.GI_F64_EXACT if a.exact_gi else device.GI_F32_FAST)

    def render(v, part=0, n_parts=0):
        c = cams[v]
        ds.render_to_device(device.make_camera(c[:3].tolist(), float(c[3]), float(c[4])), frame.data_ptr(), view=first_view + v,
                            part=part, n_parts=n_parts, **opt)
        return frame if a.backend == "nccl" else frame.cpu()

    def render_many(first, stride, count):
        """Views first, first + stride, ... in one library call, which keeps several of them in flight on the GPU."""
        sel = cams[first: first + stride * count: stride]
        frames, _ = ds.render_views([device.make_camera(c[:3].tolist(), float(c[3]), float(c[4])) for c in sel],
                                    view0=first_view + first, view_stride=stride, **opt)
        return frames.reshape(count, -1)

    t_last = [time.perf_counter()]
    # Files are written by a few more threads, so that the GPU renders the next views meanwhile (the render call and the
    # file write both release the interpreter lock).  Eight GPUs at 300 views/s each hand rank 0 about 2 GB/s of frames:
    # more than one thread copies into the page cache.
    pending = queue.Queue(maxsize=64)
    failed = []

    def writer():
        while True:
            item = pending.get()
            if item is None:
                return
            try:
                host.write_image(os.path.join(a.out, f"{a.scene}{first_view + item[0]}.ppm"), item[1].reshape(n, n, 3))
            except Exception as e:                                                                      # reported after the loop
                failed.append(e)

    def on_frame(v, pixels):
        now = time.perf_counter()
        print(f"Elapsed time: {now - t_last[0]:g} s")                                                   # mainOMP.cpp:241
        t_last[0] = now
        pending.put((v, pixels))

    wthreads = []
    if rank == 0:
        os.makedirs(a.out, exist_ok=True)
        wthreads = [threading.Thread(target=writer, daemon=True) for _ in range(max(1, a.writers))]
        for t in wthreads:
            t.start()
    t0 = time.perf_counter()
    if a.split_rows:
        multiview.run_split_views(render, len(cams), n, xdev, dist if world > 1 else None, on_frame if rank == 0 else None)
    else:
        multiview.run_views(render, len(cams), frame.numel(), xdev, dist if world > 1 else None, on_frame if rank == 0 else None,
                            batch=32, render_many=render_many)
    if wthreads:
        for _ in wthreads:
            pending.put(None)
        for t in wthreads:
            t.join()
        if failed:
            sys.exit(f"writing frames failed: {failed[0]}")
    if world > 1:
        dist.barrier()
    if rank == 0:
        dt = time.perf_counter() - t0
        print(f"{len(cams)} views on {world} GPU(s) in {dt:.3f} s: {len(cams) / dt:.2f} views/s")
    ds.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
