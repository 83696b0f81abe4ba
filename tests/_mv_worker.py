"""Worker of tests/test_multiview.py: run under torch.distributed.run with the gloo backend (CPU)."""
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracing_amd import device, host, multiview  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cpu")
    blob = cams = None
    if rank == 0:
        hs = host.HostScene.from_packed(os.path.join(ROOT, "scenes", "cbox.ptscene"))
        hs.build_tree()
        blob = device.scene_blob(hs.desc())            # host-only call: no GPU needed to serialise
        views = host.views_from_path(os.path.join(ROOT, "scenes", "paths", "cboxpath.txt"), hs.camera()["zoom"])
        cams = np.array([[*v.eye, v.angle, v.zoom] for v in views])[:37]     # a count that does not divide evenly
    blob = multiview.broadcast_bytes(blob, dev, dist)
    cams = multiview.broadcast_cameras(cams, dev, dist)
    n_views = len(cams)
    digest = hashlib.sha256(blob.tobytes()).hexdigest()
    digests = [None] * world
    dist.all_gather_object(digests, (digest, cams.tobytes()))
    mine = multiview.views_of_rank(n_views, rank, world)
    owners = [None] * world
    dist.all_gather_object(owners, mine)
    got = {}

    def render(v):
        assert v % world == rank                       # only owned views are asked for
        return torch.full((48,), (v * 7) % 251, dtype=torch.uint8)

    n_rounds = multiview.run_views(render, n_views, 48, dev, dist, (lambda v, px: got.__setitem__(v, px)) if rank == 0 else None)
    assert n_rounds == multiview.rounds(n_views, world)
    dist.barrier()
    if rank == 0:
        ok = (len(set(digests)) == 1 and sorted(sum(owners, [])) == list(range(n_views)) and sorted(got) == list(range(n_views))
              and all((got[v] == (v * 7) % 251).all() for v in got))
        print(json.dumps({"ok": bool(ok), "world": world, "views": n_views, "blob_bytes": int(blob.size),
                          "max_share": max(len(o) for o in owners), "min_share": min(len(o) for o in owners)}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
