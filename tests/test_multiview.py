"""The N > 1 path on CPU: world_size 2 and 3 over gloo (scene blob broadcast, view sharding, frame gather)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT
from pathtracing_amd import multiview


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_broadcast_shard_gather(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_mv_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(lines[-1])
    assert r["ok"] and r["world"] == world and r["views"] == 37
    assert r["max_share"] - r["min_share"] <= 1


def test_round_robin_partition():
    for n in (0, 1, 7, 320):
        for world in (1, 2, 4, 8):
            owned = [multiview.views_of_rank(n, r, world) for r in range(world)]
            assert sorted(sum(owned, [])) == list(range(n))
            assert all(v % world == r for r, vs in enumerate(owned) for v in vs)
            assert multiview.rounds(n, world) == max((len(o) for o in owned), default=0)


def test_row_band_partition():
    for size in (1, 8, 9, 100, 512):
        for world in (1, 2, 3, 8):
            owned = [multiview.bands_of_rank(size, r, world) for r in range(world)]
            assert sorted(sum(owned, [])) == list(range((size + 7) // 8))
            assert all(b % world == r for r, bs in enumerate(owned) for b in bs)
