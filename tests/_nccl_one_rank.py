"""Run by tests/test_gpu_cli.py in a process of its own: the exchanges of the one-process-per-GPU job with the nccl (RCCL)
backend, as far as one GPU allows -- a process group of one rank on device tensors.  What a second rank would do with a
broadcast scene (make its scene from the device buffer the collective filled) is done here on the first rank's buffer."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracing_amd import device, host, multiview  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    device.init(0)
    hs = host.HostScene.from_packed(os.path.join(ROOT, "scenes", "car.ptscene"))
    hs.build_tree()
    ds = device.DeviceScene(hs.desc())
    same = multiview.broadcast_scene(ds, dev, dist)              # rank 0 of 1: export to a device tensor + RCCL broadcast
    assert same is ds
    n = ds.slab_size()
    buf = torch.empty(n, dtype=torch.uint8, device=dev)
    ds.slab_export(buf.data_ptr(), n, True)
    dist.broadcast(buf, 0)
    torch.cuda.synchronize()
    peer = device.DeviceScene(slab=(buf.data_ptr(), n, True))    # what every other rank does with its received buffer
    del buf
    cams = multiview.broadcast_cameras(np.array([[-400.0, 100.0, 300.0, -0.5, 1.0]]), dev, dist)
    c = device.make_camera(cams[0, :3].tolist(), float(cams[0, 3]), float(cams[0, 4]))
    kw = dict(gi=True, spp=8, size=96, model=device.MODEL_OMP, gi_precision=device.GI_F32_FAST)
    frames = []
    for sc in (ds, peer):
        f = torch.empty(96 * 96 * 3, dtype=torch.uint8, device=dev)
        sc.render_to_device(c, f.data_ptr(), view=5, **kw)
        frames.append(f)
    assert torch.equal(frames[0], frames[1]) and int(frames[0].max()) > 0
    # the frame gather of the job on one rank: slot -> gather -> the frame comes back under its view index
    g = multiview.FrameGather(frames[0].numel(), dev, dist, batch=2)
    g.slot(0).copy_(frames[0]); g.slot(1).copy_(frames[1])
    got = {}
    g.flush(2, got)
    assert sorted(got) == [0, 1] and np.array_equal(got[0], frames[0].cpu().numpy())
    agg = torch.tensor([1.5, 2.0], dtype=torch.float64, device=dev)
    dist.all_reduce(agg, op=dist.ReduceOp.MAX)
    dist.barrier()
    peer.close(); ds.close()
    dist.destroy_process_group()
    print("nccl one-rank exchanges ok")


if __name__ == "__main__":
    main()
