"""The command-line front end (the reference's `[scene] -gi -rb -rm` surface) on a GPU: frames it writes are the
frames the C ABI returns, for the plain run, the camera steps, the stdin protocol and the data-set triplets."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from pathtracing_amd import device, host

pytestmark = pytest.mark.gpu
PTMAIN = os.path.join(ROOT, "pathtracing_amd", "ptmain")


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def read_ppm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = (int(x) for x in parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)[:, :, ::-1]      # -> BGR


def run(args, tmp, stdin=None):
    r = subprocess.run([PTMAIN] + args + ["--out", str(tmp)], cwd=ROOT, input=stdin, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@pytest.fixture(scope="module")
def cbox(gpu_scenes, host_scenes):
    return host_scenes("cbox"), gpu_scenes("cbox")


def test_plain_run_and_camera_steps(tmp_path, cbox):
    hs, ds = cbox
    out = run(["cbox", "--size", "64", "--keys", "wl"], tmp_path)
    assert "Real Triangles: 2190; Naive: 12440; Optimized: 9073" in out      # project.pdf Table 1
    assert out.count("Elapsed time:") == 3
    c = hs.camera()
    want = ds.render(c["eye"], c["angle"], c["zoom"], size=64, aov=False)["bgr"]
    assert np.array_equal(read_ppm(tmp_path / "cbox.ppm"), want)
    eye, angle = host.camera_step(c["eye"], c["angle"], c["step"], c["astep"], "w")
    eye, angle = host.camera_step(eye, angle, c["step"], c["astep"], "l")
    want = ds.render(eye, angle, c["zoom"], size=64, aov=False)["bgr"]
    assert np.array_equal(read_ppm(tmp_path / "cbox_step2.ppm"), want)


def test_stdin_protocol(tmp_path, cbox):
    hs, ds = cbox
    c = hs.camera()
    deg = c["angle"] / 0.01745329251
    out = run(["cbox", "--size", "48", "--stdin", "-gi", "--spp", "2"], tmp_path,
              stdin=f"d\n\nC {c['eye'][0]} {c['eye'][1]} {c['eye'][2]} {deg!r}\nq\nw\n")
    assert out.count("Elapsed time:") == 3                                   # first frame + two commands; nothing after q
    first, placed = read_ppm(tmp_path / "cbox.ppm"), read_ppm(tmp_path / "cbox_frame2.ppm")
    assert np.array_equal(first, placed)                                     # "C" put the camera back where it started
    assert not np.array_equal(first, read_ppm(tmp_path / "cbox_frame1.ppm"))
    r = subprocess.run([PTMAIN, "cbox", "--size", "16", "--stdin", "--out", str(tmp_path)], cwd=ROOT, input="x\n",
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "unknown command" in r.stderr


def test_dataset_triplets(tmp_path, cbox):
    hs, ds = cbox
    run(["cbox", "-rm", "-gi", "--views", "2", "--size", "32", "--dataset", str(tmp_path / "data"), "--noisy-spp", "1",
         "--converged-spp", "3"], tmp_path)
    cams = host.views_from_path(os.path.join(ROOT, "scenes", "paths", "cboxpath.txt"), hs.camera()["zoom"])
    for v in range(2):
        files = {k: tmp_path / "data" / "CBOX" / f"CBOX{k}" / f"cbox{v}.ppm" for k in ("4ssp", "4096ssp", "dif")}
        cam = cams[v]
        for k, kw in (("4ssp", dict(gi=True, spp=1)), ("4096ssp", dict(gi=True, spp=3)), ("dif", dict(gi=False))):
            want = ds.render(list(cam.eye), cam.angle, cam.zoom, size=32, aov=False, model=device.MODEL_OMP, view=v,
                             gi_precision=device.GI_F32_FAST, **kw)["bgr"]
            assert np.array_equal(read_ppm(files[k]), want), (v, k)


def test_dataset_as_jpeg_files(tmp_path, cbox):
    """--format jpg: the triplets under the reference's own file names (<scene><view>.jpg, what cv::imwrite wrote for its
    data sets); each file decodes to the frame the C ABI renders, within JPEG error."""
    from PIL import Image
    hs, ds = cbox
    run(["cbox", "-rm", "-gi", "--views", "1", "--size", "64", "--dataset", str(tmp_path / "data"), "--noisy-spp", "2",
         "--converged-spp", "8", "--format", "jpg"], tmp_path)
    cam = host.views_from_path(os.path.join(ROOT, "scenes", "paths", "cboxpath.txt"), hs.camera()["zoom"])[0]
    for k, kw in (("4ssp", dict(gi=True, spp=2)), ("4096ssp", dict(gi=True, spp=8)), ("dif", dict(gi=False))):
        f = tmp_path / "data" / "CBOX" / f"CBOX{k}" / "cbox0.jpg"
        got = np.asarray(Image.open(f).convert("RGB"))[:, :, ::-1].astype(float)
        want = ds.render(list(cam.eye), cam.angle, cam.zoom, size=64, aov=False, model=device.MODEL_OMP, view=0,
                         gi_precision=device.GI_F32_FAST, **kw)["bgr"].astype(float)
        assert got.shape == want.shape and np.sqrt(np.mean((got - want) ** 2)) < (12.0 if kw["gi"] else 4.0), k


def test_rm_job_frames_do_not_depend_on_the_number_of_ranks(tmp_path, cbox):
    """`python -m pathtracing_amd.rm` (the -rm job, views dealt round-robin to ranks): one process and two processes
    (gloo rehearsal, both ranks on this GPU) write byte-identical frames, equal to what the C ABI renders."""
    import sys
    hs, ds = cbox
    args = ["cbox", "-gi", "--spp", "4", "--size", "48", "--views", "5", "--root", ROOT]
    one, two = tmp_path / "one", tmp_path / "two"
    r = subprocess.run([sys.executable, "-m", "pathtracing_amd.rm"] + args + ["--out", str(one)], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Real Triangles: 2190; Naive: 12440; Optimized: 9073" in r.stdout and "World Size: 1" in r.stdout
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(free_port()), "-m", "pathtracing_amd.rm"] + args +
                       ["--out", str(two), "--backend", "gloo"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "World Size: 2" in r.stdout and r.stdout.count("Elapsed time:") == 5
    # latency mode: both ranks share every view, bands of 8 rows dealt round-robin (48 rows = 6 bands)
    split = tmp_path / "split"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(free_port()), "-m", "pathtracing_amd.rm"] + args +
                       ["--out", str(split), "--backend", "gloo", "--split-rows"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("Elapsed time:") == 5
    cams = host.views_from_path(os.path.join(ROOT, "scenes", "paths", "cboxpath.txt"), hs.camera()["zoom"])
    for v in range(5):
        a, b = read_ppm(one / f"cbox{v}.ppm"), read_ppm(two / f"cbox{v}.ppm")
        assert np.array_equal(a, b), v
        assert np.array_equal(a, read_ppm(split / f"cbox{v}.ppm")), v
        want = ds.render(list(cams[v].eye), cams[v].angle, cams[v].zoom, size=48, aov=False, gi=True, spp=4,
                         model=device.MODEL_OMP, view=v, gi_precision=device.GI_F32_FAST)["bgr"]
        assert np.array_equal(a, want), v


@pytest.mark.parametrize("scene,model,spp,size", [("car", "omp", 70, 96), ("cbox", "main", 70, 96), ("car", "omp", 16, 100), ("cbox", "main", 4, 100)])
def test_kernel_variants_write_the_same_frame(tmp_path, scene, model, spp, size):
    """The fp32 GI path has two forms of its tree walk (quad records, two levels per record, and the binary records kept
    for the no-allocation entry point) and, for the mainOMP model, shading and summing as one kernel or as two; the ray
    generator sorts the rays of one tile or, in passes of at most 16 samples, of four tiles as one run, in blocks of 1 024 or
    512 threads (size 100: 169 tiles, so the last group of four is not full).  They are the same computation: a -gi frame
    is the same bytes whichever is chosen (the switches are read once per process, hence one process per variant)."""
    frames = {}
    for name, env in (("default", {}), ("binary records", {"PT_TUNE_QUAD": "0"}), ("two kernels", {"PT_TUNE_FUSED_SUM": "0"}),
                      ("both", {"PT_TUNE_QUAD": "0", "PT_TUNE_FUSED_SUM": "0"}), ("one tile per run", {"PT_TUNE_SORT_TILES": "1"}),
                      ("small generator blocks", {"PT_TUNE_GEN_THREADS": "512"}),
                      ("one tile, small blocks, two kernels", {"PT_TUNE_SORT_TILES": "1", "PT_TUNE_GEN_THREADS": "512", "PT_TUNE_FUSED_SUM": "0"})):
        out = tmp_path / name.replace(" ", "_").replace(",", "")
        os.makedirs(out)
        r = subprocess.run([PTMAIN, scene, "-gi", "--model", model, "--spp", str(spp), "--size", str(size), "--out", str(out)], cwd=ROOT,
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        frames[name] = read_ppm(out / f"{scene}.ppm")
    assert int(frames["default"].max()) > 0
    for name, f in frames.items():
        assert np.array_equal(f, frames["default"]), name


def test_bench_with_two_ranks_started_by_itself(cbox):
    """`python bench.py --gpus 2` without a launcher around it (how the round driver starts it) runs two ranks -- here the
    gloo rehearsal, both on this one GPU -- and rank 0 prints one line for the whole job; the one-rank line has the
    same fields."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    lines = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--workload", "cbox",
                            "--spp", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        out = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
        assert len(out) == 1, r.stdout
        lines[n] = out[0]
    one, two = lines[1], lines[2]
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["steps"] == 3 and two["scaling"] == "weak"
    assert set(one) == set(two)
    assert two["config"]["views_per_step"] == 2 and two["value"] > 0 and two["roofline"]["frac"] > 0
    # weak scaling: every rank renders the same cameras, so the job traced twice the rays
    rays = lambda j: j["value"] * 1e6 * j["ms_per_step"] * 1e-3 * j["steps"]
    assert abs(rays(two) / rays(one) - 2.0) < 0.02
    # who took part and what the start of the job cost: one entry per rank, the backend, the scene broadcast timed by itself
    assert one["ranks"] is None and one["backend"] is None and one["scene_broadcast"] is None
    assert [r["rank"] for r in two["ranks"]] == [0, 1] and all(r["device_name"] and r["pci_bus_id"] for r in two["ranks"])
    assert two["backend"] == "gloo" and two["scene_broadcast"]["bytes"] > 1 << 20 and two["scene_broadcast"]["ms"] > 0
    # ray classes, and the second leg: the -gi -rb -rm job at the reference's settings over every view of the path
    for j in (one, two):
        c = j["mrays_per_s_by_class"]
        assert set(c) == {"primary", "mirror", "shadow", "gi"} and abs(sum(c.values()) - j["value"]) < 0.01 * j["value"] + 0.1
        rm = j["rm_rb16"]
        assert rm["n_views"] == rm["frames_on_rank0"] == 320 and rm["views_per_s"] > 0 and rm["primary_ms"] > 0 and rm["closest_ms"] > 0
        assert abs(rm["views_per_s"] * rm["ms_per_view"] * 1e-3 - 1.0) < 0.01
    # several views per rank and step, handed to the library in one call (which keeps them in flight together)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "cbox", "--spp", "16",
                        "--steps", "2", "--warmup", "1", "--views-per-step", "3", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    many = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(many) == 1 and many[0]["config"]["views_per_step"] == 6 and many[0]["n_gpus"] == 2
    assert abs(many[0]["views_per_s"] * many[0]["ms_per_step"] * 1e-3 - 6.0) < 0.01


def test_resume_keeps_whole_frames_and_renders_the_rest(tmp_path):
    """--resume: the -rm job goes on where an earlier one stopped.  Frame files that are there and whole are kept (not even
    opened for writing), missing and truncated ones are rendered; the files end up those of an uninterrupted job."""
    import sys
    full, part = tmp_path / "full", tmp_path / "part"
    args = ["cbox", "-rm", "-gi", "--views", "6", "--size", "48", "--spp", "2"]
    for mode in ("loop", "all"):                                 # the plain loop of ptmain, and pt_render_views in batches
        extra = ["--gpus", "all"] if mode == "all" else []
        for d in (full, part):
            if d.exists():
                for f in d.iterdir():
                    f.unlink()
            else:
                os.makedirs(d)
        run(args + extra, full)
        run(args + extra, part)
        (part / "cbox3.ppm").unlink()
        whole = (part / "cbox4.ppm").read_bytes()
        (part / "cbox4.ppm").write_bytes(whole[: len(whole) // 2])
        stamp = {k: (part / f"cbox{k}.ppm").stat().st_mtime_ns for k in (0, 1, 2)}
        out = run(args + extra + ["--resume"], part)
        for k in range(6):
            assert (part / f"cbox{k}.ppm").read_bytes() == (full / f"cbox{k}.ppm").read_bytes(), (mode, k)
        for k in (0, 1, 2):
            assert (part / f"cbox{k}.ppm").stat().st_mtime_ns == stamp[k], (mode, k)
        assert ("Resuming at view 3 of 6" in out) if mode == "all" else (out.count("kept ") == 4)
    # the one-process-per-GPU job
    rm = [sys.executable, "-m", "pathtracing_amd.rm", "cbox", "-gi", "--views", "6", "--size", "48", "--spp", "2"]
    a, b = tmp_path / "rm_full", tmp_path / "rm_part"
    for d in (a, b):
        r = subprocess.run(rm + ["--out", str(d)], cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
    for k in (2, 3, 4, 5):
        (b / f"cbox{k}.ppm").unlink()
    r = subprocess.run(rm + ["--out", str(b), "--resume"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Resuming at view 2 of 6" in r.stdout, r.stdout + r.stderr
    for k in range(6):
        assert (b / f"cbox{k}.ppm").read_bytes() == (a / f"cbox{k}.ppm").read_bytes(), k


def test_a_view_that_fails_gets_a_second_attempt_on_another_worker(cbox, monkeypatch):
    """A view whose render fails on one worker of pt_render_views is rendered again by a worker that is in order; the frames
    and ray counts are those of the undisturbed job.  With a single worker there is nobody to take it: the call fails and
    names the view's error."""
    hs, ds = cbox
    cams = host.views_from_path(os.path.join(ROOT, "scenes", "paths", "cboxpath.txt"), hs.camera()["zoom"])[:9]
    kw = dict(gi=True, rb=True, spp=4, size=64, model=device.MODEL_OMP, gi_precision=device.GI_F32_FAST)
    monkeypatch.delenv("PT_TUNE_REPLICAS", raising=False)
    monkeypatch.delenv("PT_TUNE_FAIL_VIEW", raising=False)
    monkeypatch.setenv("PT_TUNE_WORKERS", "3")
    good, st0 = ds.render_views(cams, view0=0, view_stride=1, **kw)
    for bad in ("0", "4", "8"):
        monkeypatch.setenv("PT_TUNE_FAIL_VIEW", bad)
        frames, st = ds.render_views(cams, view0=0, view_stride=1, **kw)
        assert np.array_equal(good, frames), bad
        for k in ("primary", "gi", "mirror", "shadow", "omp_retrace"):
            assert st0[k] == st[k], (bad, k)
    monkeypatch.setenv("PT_TUNE_WORKERS", "1")
    with pytest.raises(device.DeviceError, match="injected failure"):
        ds.render_views(cams, view0=0, view_stride=1, **kw)
    monkeypatch.delenv("PT_TUNE_FAIL_VIEW")
    again, _ = ds.render_views(cams, view0=0, view_stride=1, **kw)          # the scene is as usable as before
    assert np.array_equal(good, again)


def test_view_loop_over_several_workers_gives_the_same_frames(tmp_path, cbox, monkeypatch):
    """pt_render_views hands the views to worker threads: one copy of the scene per device, and on every device up to
    four workers that share that copy (several small views in flight fill the GPU better than one).
    On a one-GPU machine the workers all use that GPU (PT_TUNE_REPLICAS): three workers -- scene copied device to
    device, and once more through an RCCL broadcast in a communicator of one rank, so that the RCCL calls run too --
    write exactly the frames of one worker, with the same ray counts; `ptmain -rm --gpus all` on top of it writes the
    same files as the plain loop."""
    hs, ds = cbox
    cams = host.views_from_path(os.path.join(ROOT, "scenes", "paths", "cboxpath.txt"), hs.camera()["zoom"])[:7]
    kw = dict(gi=True, rb=True, spp=4, size=64, model=device.MODEL_OMP, gi_precision=device.GI_F32_FAST)
    monkeypatch.delenv("PT_TUNE_REPLICAS", raising=False)
    monkeypatch.setenv("PT_TUNE_WORKERS", "1")
    one, st1 = ds.render_views(cams, view0=0, view_stride=1, **kw)
    for workers in ("4", None):                                 # four workers on the one copy; the library's own choice
        if workers is None:
            monkeypatch.delenv("PT_TUNE_WORKERS")
        else:
            monkeypatch.setenv("PT_TUNE_WORKERS", workers)
        shared, sts = ds.render_views(cams, view0=0, view_stride=1, **kw)
        assert np.array_equal(one, shared)
        for k in ("primary", "gi", "mirror", "shadow", "omp_retrace"):
            assert st1[k] == sts[k], k
    for self_rccl in ("0", "1"):
        monkeypatch.setenv("PT_TUNE_REPLICAS", "3")
        monkeypatch.setenv("PT_TUNE_RCCL_SELF", self_rccl)
        fresh = device.DeviceScene(hs.desc())                   # its copies are made on first use
        three, st3 = fresh.render_views(cams, view0=0, view_stride=1, **kw)
        again, _ = fresh.render_views(cams[:2], view0=0, view_stride=1, **kw)      # copies reused; fewer views than workers
        fresh.close()
        assert np.array_equal(one, three) and np.array_equal(one[:2], again)
        for k in ("primary", "gi", "mirror", "shadow", "omp_retrace"):
            assert st1[k] == st3[k], k
    monkeypatch.delenv("PT_TUNE_RCCL_SELF")
    a, b = tmp_path / "a", tmp_path / "b"
    os.makedirs(a); os.makedirs(b)
    monkeypatch.delenv("PT_TUNE_REPLICAS")
    run(["cbox", "-rm", "-gi", "--views", "5", "--size", "48", "--spp", "2"], a)
    env = dict(os.environ, PT_TUNE_REPLICAS="2")
    r = subprocess.run([PTMAIN, "cbox", "-rm", "-gi", "--views", "5", "--size", "48", "--spp", "2", "--gpus", "all", "--out", str(b)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "World Size: 1" in r.stdout and r.stdout.count("Elapsed time:") == 5 and "views/s" in r.stdout
    for v in range(5):
        assert np.array_equal(read_ppm(a / f"cbox{v}.ppm"), read_ppm(b / f"cbox{v}.ppm")), v


def test_scene_slab_round_trip(cbox):
    """A scene's device data as one position-independent block: exported to host memory and to device memory, a scene
    made from either renders the same frame as the original (nothing is prepared again: the fp32 tree travels inside);
    a block with a damaged header, or with a bit flipped anywhere in its arrays, is refused."""
    import ctypes as C
    hs, ds = cbox
    c = hs.camera()
    kw = dict(gi=True, spp=4, size=64, gi_precision=device.GI_F32_FAST)
    want = ds.render(c["eye"], c["angle"], c["zoom"], **kw)
    n = ds.slab_size()
    assert n > 1 << 20
    buf = np.zeros(n, np.uint8)
    ds.slab_export(buf.ctypes.data, n, False)
    from_host = device.DeviceScene(slab=(buf.ctypes.data, n, False))
    got = from_host.render(c["eye"], c["angle"], c["zoom"], **kw)
    assert np.array_equal(got["rgb"], want["rgb"]) and np.array_equal(got["tri"], want["tri"])
    assert from_host.info() == ds.info()
    # device to device, through the HIP runtime the library is linked to
    hip = None
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            hip = C.CDLL(line.split()[-1])
            break
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), n) == 0
    try:
        ds.slab_export(d.value, n, True)
        from_dev = device.DeviceScene(slab=(d.value, n, True))
    finally:
        hip.hipFree(d)
    got = from_dev.render(c["eye"], c["angle"], c["zoom"], **kw)
    assert np.array_equal(got["rgb"], want["rgb"])
    from_dev.close(); from_host.close()
    bad = buf.copy()
    bad[100] ^= 0xFF
    with pytest.raises(device.DeviceError, match="slab"):
        device.DeviceScene(slab=(bad.ctypes.data, n, False))
    with pytest.raises(device.DeviceError):
        device.DeviceScene(slab=(buf.ctypes.data, n // 2, False))
    # damage anywhere in the arrays -- a node link, a triangle offset, a texel -- is refused too (checksum over the body),
    # not followed by the kernels
    for at in (8192, n // 3, n // 2 + 5, n - 300):
        bad = buf.copy()
        bad[at] ^= 0x01
        with pytest.raises(device.DeviceError, match="checksum"):
            device.DeviceScene(slab=(bad.ctypes.data, n, False))


def test_nccl_backend_exchanges_on_one_rank():
    """The RCCL side of the one-process-per-GPU job (scene broadcast from device memory, camera broadcast, frame gather,
    reductions) on a process group of one rank, plus the receiving side of the scene broadcast done by hand on the
    same GPU.  Two ranks cannot share a GPU under RCCL, so this is as far as a one-GPU machine goes."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for attempt in range(3):                # the port found free can be taken by the time the store binds it
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_one_rank.py"), str(free_port())], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        if "EADDRINUSE" not in r.stderr:
            break
    assert r.returncode == 0 and "nccl one-rank exchanges ok" in r.stdout, r.stdout + r.stderr


def test_two_host_threads_on_one_scene(cbox):
    """Calls on one scene from two host threads are serialised by the library (one workspace, one stream pair per scene):
    both threads get exactly the frames a single thread gets."""
    import threading
    hs, ds = cbox
    c = hs.camera()
    kw = dict(gi=True, spp=8, size=96, model=device.MODEL_OMP, gi_precision=device.GI_F32_FAST, aov=False)
    want = [ds.render(c["eye"], c["angle"] + 0.3 * k, c["zoom"], view=k, **kw)["bgr"] for k in range(6)]
    got, errors = {}, []

    def worker(ks):
        try:
            for k in ks:
                got[k] = ds.render(c["eye"], c["angle"] + 0.3 * k, c["zoom"], view=k, **kw)["bgr"]
        except Exception as e:                                  # noqa: BLE001 -- reported by the assertion below
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(range(w, 6, 2),)) for w in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(6):
        assert np.array_equal(got[k], want[k]), k
