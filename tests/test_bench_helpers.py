"""Host-side pieces of bench.py that need no GPU."""
import json
import os

from conftest import ROOT


def test_pmc_traffic_reads_the_committed_counter_summaries():
    import bench
    for workload in ("atrium", "car"):
        got = bench.pmc_traffic(workload, 16777216)
        assert got is not None, workload
        gb, source, gb_face = got
        assert source.startswith("profiles/") and os.path.exists(os.path.join(ROOT, source))
        assert 0.1 < gb_face <= gb < 100.0          # FETCH_SIZE with the guide's correction, and at face value
    assert bench.pmc_traffic("no-such-workload", 1) is None
    half = bench.pmc_traffic("atrium", 8388608)[0]
    assert abs(half - bench.pmc_traffic("atrium", 16777216)[0] / 2) < 0.02


def test_counter_collections_carry_the_kernel_source_digest():
    """The newest collection per workload records the digest of the kernel sources it was taken with; bench.py flags
    traffic and hit rates as stale when the sources have changed since."""
    import bench
    m = bench.pmc_means("atrium")
    assert m["source"].startswith("profiles/r0") and m["kernel_source_hash"] and m["rays_per_launch"] > 6e7
    assert len(bench.kernel_source_hash()) == 16
    p = bench.physical_roofline("atrium", 63.6, 15.25, 4450.0, 15.0, 67108800)
    assert p["counters_stale"] == (m["kernel_source_hash"] != bench.kernel_source_hash())
    # the busier unit decides; for the vector ALUs the fraction is lane-weighted (issue x live lanes per instruction)
    assert 0.2 < p["frac"] <= 1.0 and 0.2 < p["valu_live_lanes"] <= 1.0
    if p["valu_issue"] >= p["ta_busy"]:
        assert abs(p["frac"] - min(p["valu_issue"], 1.0) * p["valu_live_lanes"]) < 2e-3
    else:
        assert p["frac"] == p["ta_busy"]
    assert 0.05 < p["hbm_side_frac_of_peak"] < 0.5          # the kernel is far from HBM-bound


def test_committed_bench_lines_have_the_contract_fields():
    for name in ("bench_default_atrium.json", "bench_default_car.json"):
        j = json.load(open(os.path.join(ROOT, "profiles", "r02", name)))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in j, (name, k)
        assert "workload" in j["config"] and "model" in j["config"]
        r = j["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (name, k)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["traffic"] is not None and r["bound"].startswith("hbm") and r["physical"]["limiter"]
        c = j["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0


def test_supplied_sponza_directory_becomes_the_default_workload(tmp_path, monkeypatch):
    """bench.py runs the stand-in only while scenes/sponza/ is incomplete (the reference ships no sponza mesh): a
    directory with master.txt and every mesh it names is picked up and read by the text reader."""
    import shutil

    import pytest

    import bench
    assert bench.default_workload() == "atrium"                      # nothing supplied in this repository
    ref = "/root/reference/scenes/cbox"
    if not os.path.isdir(ref):
        pytest.skip("reference scene directory not present")
    root = tmp_path / "root"
    shutil.copytree(ref, root / "scenes" / "sponza")                 # any complete scene directory will do
    monkeypatch.setattr(bench, "ROOT", str(root))
    assert bench.default_workload() == "sponza"
    hs, st, views = bench.load_host_scene("sponza")
    assert st["triangles"] == 2190 and len(views) == 16              # no camera path supplied: 16 angles at the scene camera
    os.remove(root / "scenes" / "sponza" / "cbox2.txt")
    assert bench.default_workload() == "atrium"                      # mesh missing again


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` as one plain process (no WORLD_SIZE in the environment, the way the round driver starts
    it) launches the two ranks itself as a child process and passes rank 0's line through; no GPU is needed for the
    rendezvous-only form."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert lines == [{"rendezvous_only": True, "n_gpus": 2, "rank_sum": 3.0}]
