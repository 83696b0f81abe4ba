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
    # the same job with a whole batch of rounds rendered per call; a batch size that divides neither the rounds nor the views
    got_many, calls = {}, []

    def render_many(first, stride, count):
        assert first % world == rank and stride == world and count >= 1
        calls.append(count)
        return np.stack([np.full(48, ((first + j * stride) * 7) % 251, np.uint8) for j in range(count)])

    multiview.run_views(None, n_views, 48, dev, dist, (lambda v, px: got_many.__setitem__(v, px.copy())) if rank == 0 else None,
                        batch=5, render_many=render_many)
    assert sum(calls) == len(mine) and max(calls) <= 5
    # one view split over the ranks by row bands; sizes that are not multiples of 8 rows or of the world size
    split_ok = True
    for size in (20, 64, 9):
        def pattern(v):
            return ((np.arange(size * size * 3, dtype=np.int64) * 31 + v * 17) % 253).astype(np.uint8).reshape(size, size * 3)

        def render_part(v, part, n_parts, size=size, pattern=pattern):
            assert (part, n_parts) == (rank, world)
            full = pattern(v)
            f = np.full_like(full, 255)                # rows of other ranks hold junk that must not reach the frame
            for b in multiview.bands_of_rank(size, part, n_parts):
                f[8 * b: 8 * b + 8] = full[8 * b: 8 * b + 8]
            return torch.from_numpy(f.reshape(-1))

        whole = {}
        multiview.run_split_views(render_part, 3, size, dev, dist, (lambda v, px: whole.__setitem__(v, px)) if rank == 0 else None)
        if rank == 0:
            split_ok = split_ok and sorted(whole) == [0, 1, 2] and all((whole[v].reshape(size, size * 3) == pattern(v)).all() for v in whole)
    dist.barrier()
    if rank == 0:
        ok = (split_ok and len(set(digests)) == 1 and sorted(sum(owners, [])) == list(range(n_views)) and sorted(got) == list(range(n_views))
              and all((got[v] == (v * 7) % 251).all() for v in got)
              and sorted(got_many) == list(range(n_views)) and all((got_many[v] == (v * 7) % 251).all() for v in got_many))
        print(json.dumps({"ok": bool(ok), "world": world, "views": n_views, "blob_bytes": int(blob.size),
                          "max_share": max(len(o) for o in owners), "min_share": min(len(o) for o in owners)}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
