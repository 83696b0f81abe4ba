#!/usr/bin/env python3
"""Benchmark of the render hot path: views of a -gi job through the HIP path.

    python bench.py --gpus N --steps K --warmup W

A step is one 512x512 view rendered per GPU (primary + mirror + shadow rays in
fp64, GI sample rays through the fp32 wavefront path), the frame left in device
memory and gathered to rank 0.  With N > 1 (one process per GPU, launched by
torch.distributed.run) rank 0 reads the scene, builds the trees, uploads the
scene and hands its device data to the other ranks with one RCCL broadcast; step i renders
camera i of the path on every rank under the RNG key of view i * N + rank of
the multi-view job (equal work per GPU for every N), the frames go to rank 0
in one gather per 8 steps, and no other collective touches the data path.
Rank 0 prints one JSON line.

value    = rays of all classes traced by all ranks / wall time of the K steps
roofline = the GI closest-hit kernel (the dominant kernel): algorithmic bytes
           per launch / mean launch duration (HIP events on the kernel's stream)
cpu_baseline = the CPU oracle (a port of the reference's render path) timed on
           a bounded sample of the same workload, rank 0, N = 1 only.
rm_rb16  = after the timed steps, a second leg timed by itself: the multi-view job at the reference's own settings
           (`-gi -rb -rm`: mainOMP shading model, 16 samples, ray bundles; mainOMP.cpp:111-136,154-247) over every
           view of the camera path, views dealt round-robin to the ranks, each rank's views handed to
           pt_render_views in batches (several views in flight per GPU), frames gathered on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (packed scene or generator, path file, label)
    "car": ("scenes/car.ptscene", "scenes/paths/carpath.txt", "car -gi (10 264 triangles), views of scenes/paths/carpath.txt"),
    "cbox": ("scenes/cbox.ptscene", "scenes/paths/cboxpath.txt", "cbox -gi (2 190 triangles), views of scenes/paths/cboxpath.txt"),
    # the reference ships the sponza textures, master file and camera path but not the mesh (sponza2.txt); when a
    # complete scenes/sponza/ is supplied next to this file it is the default workload, read by the text reader as it is
    "sponza": ("dir:scenes/sponza", "scenes/paths/sponzapath.txt", "sponza -gi, scene directory supplied by the user"),
    "atrium": ("gen:262267", None, "atrium-262k -gi: procedural stand-in (262 267 triangles, seed 12345) for the sponza mesh, "
                                   "which is not available; not Crytek Sponza"),
    # the room mesh (bedroom.txt, 340 265 triangles) is not shipped either; its configuration is -gi -rb
    "room": ("gen:340265", None, "atrium-340k -gi -rb: procedural stand-in (340 265 triangles, seed 12345) at the triangle count of the "
                                 "room mesh, which is not available; ray bundles on"),
}
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E, spec


KERNEL_SOURCES = ("pt_fast.hip", "pt_trace_f32.h", "pt_trace_quad.h", "pt_persistent.h", "pt_refine.hip", "pt_scene_dev.h")


def kernel_source_hash():
    """Digest of the sources that define the closest-hit kernel and the tree it walks: counter summaries under
    profiles/ carry the digest they were collected with, and go stale when it changes."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "pathtracing_amd", "csrc", "device", name), "rb").read())
    return h.hexdigest()[:16]


def pmc_means(workload):
    """Mean per launch of the PMC counters of k_trace_closest from the newest committed collection (profiles/<round>/
    <workload>/pmc_k_trace_closest.txt), with the path and the kernel-source digest recorded in it; or None."""
    best = None
    for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        f = os.path.join(ROOT, "profiles", rnd, workload, "pmc_k_trace_closest.txt")
        if os.path.exists(f):
            best = f
    if best is None:
        return None
    vals, digest, rays = {}, None, 67108864.0
    for line in open(best):
        t = line.split()
        if len(t) >= 3 and t[0] == "#" and t[1] == "kernel_source_hash":
            digest = t[2]
        elif len(t) >= 3 and t[0] == "#" and t[1] == "rays_per_launch":
            rays = float(t[2])
        elif len(t) >= 3 and t[-1].startswith("mean="):
            vals[t[0]] = float(t[-1][5:])
    return {"values": vals, "source": os.path.relpath(best, ROOT), "kernel_source_hash": digest, "rays_per_launch": rays}


def pmc_traffic(workload, rays_per_launch):
    """HBM-side bytes per launch of k_trace_closest from the committed PMC passes (profiles/), or None.
    FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies wide reads at half size, hence the factor 2."""
    m = pmc_means(workload)
    if m is None or "FETCH_SIZE" not in m["values"] or "WRITE_SIZE" not in m["values"]:
        return None
    scale = rays_per_launch / m["rays_per_launch"]          # the launches the counters were collected on may differ in size
    gb = lambda f: round((f * m["values"]["FETCH_SIZE"] + m["values"]["WRITE_SIZE"]) * 1024.0 * scale / 1e9, 2)
    # [0] with the guide's correction, [2] with FETCH_SIZE taken at face value: TCC_EA0_RDREQ x 64 B equals the uncorrected
    # figure for this kernel's divergent 16-byte loads, so the truth lies between the two
    return gb(2.0), m["source"], gb(1.0)


def measure_load_rates():
    """tools/ta_rate as a child process: per-lane 16-byte loads per clock per CU that hit L1, L2, or go further."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "ta_rate")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120)
        return json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else None
    except Exception:
        return None


def physical_roofline(workload, n_node, n_tri, kernel_mrays, ms_launch, rays_launch):
    """What really bounds the traversal kernel: per-lane load requests through the L1 address path.
    Requests per ray are counted (one 16-byte record per node visited, three per triangle tested, ray origin,
    direction and slot); what a request costs depends on where it hits, which the load-rate micro-benchmark measures
    in this very run; the hit rates and the memory-side traffic come from the committed counter collection."""
    rates = measure_load_rates()
    m = pmc_means(workload)
    phys = {"limiter": "per-lane load requests through the L1 address path (TA), then VALU issue",
            "lane_requests_per_ray": round(n_node + 3.0 * n_tri + 3.0, 1), "load_rates_per_clk_per_cu": rates,
            "load_rates_source": "tools/ta_rate, run by this process beside the timed steps" if rates else None}
    if m:
        v = m["values"]
        phys["counters_source"] = m["source"]
        phys["counters_stale"] = m["kernel_source_hash"] != kernel_source_hash()
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in v and "TCP_TCC_READ_REQ_sum" in v and "TCC_HIT_sum" in v and "TCC_REQ_sum" in v:
            phys["l1_hit"] = round(1.0 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)
            phys["l2_hit"] = round(v["TCC_HIT_sum"] / v["TCC_REQ_sum"], 3)
            phys["hw_requests_per_ray"] = round(v["TCP_TOTAL_CACHE_ACCESSES_sum"] / m["rays_per_launch"], 1)
        if "TA_BUSY_avr" in v and "GRBM_GUI_ACTIVE" in v:
            phys["ta_busy"] = round(v["TA_BUSY_avr"] / (v["GRBM_GUI_ACTIVE"] / 8.0), 3)
        if "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v:
            # instructions x 4 cycles (a wave64 instruction on a 16-lane SIMD) over the SIMD cycles of the launch; the raw figure
            # can pass 1 by a percent (an instruction whose upper or lower 32 lanes are all idle takes fewer cycles)
            # (a model, not a counter: 4 cycles per instruction is the common case, not every case)
            phys["valu_issue"] = round(v["SQ_INSTS_VALU"] * 4.0 / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 3)
        if "SQ_THREAD_CYCLES_VALU" in v and "SQ_ACTIVE_INST_VALU" in v:
            # lanes that do work per issued vector instruction: each lane of the traversal uses one or two of the three
            # blocks (triangle test, quad step, leaf step) that the whole wave walks through on every turn
            phys["valu_live_lanes"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 3)
        tr = pmc_traffic(workload, rays_launch)
        if tr and ms_launch > 0:
            phys["hbm_side_gb_per_s"] = round(tr[0] / (ms_launch * 1e-3), 1)
            phys["hbm_side_frac_of_peak"] = round(tr[0] / (ms_launch * 1e-3) / HBM_PEAK_GBS, 4)
    if rates and "l1_hit" in phys:
        h1, h2 = phys["l1_hit"], phys["l2_hit"]
        clocks = h1 / rates["l1"] + (1 - h1) * h2 / rates["l2"] + (1 - h1) * (1 - h2) / rates["beyond"]
        req = phys.get("hw_requests_per_ray", phys["lane_requests_per_ray"])
        ceiling = 256 * 2.4e9 / (req * clocks) / 1e6           # the micro-benchmark normalises by the same 256 CUs x 2.4 GHz
        phys.update(clocks_per_request=round(clocks, 3), request_ceiling_mrays_per_s=round(ceiling, 1),
                    achieved_mrays_per_s=round(kernel_mrays, 1), request_model_frac=round(kernel_mrays / ceiling, 4))
    # which of the two units that bound the kernel is busier, and the fraction that counts: when it is the vector ALUs, the
    # share of their LANE-cycles that do work (issue x live lanes) -- a wave that issues on every cycle with 40 % of its
    # lanes alive uses 40 % of the ALUs
    units = {k: phys[k] for k in ("ta_busy", "valu_issue") if k in phys}
    if units:
        top = max(units, key=units.get)
        phys["limiter"] = ("per-lane load requests through the L1 address path (TA), then VALU issue" if top == "ta_busy" else
                           "VALU issue, then per-lane load requests through the L1 address path (TA)")
        if top == "valu_issue" and "valu_live_lanes" in phys:
            phys["frac"] = round(min(phys["valu_issue"], 1.0) * phys["valu_live_lanes"], 3)
            phys["frac_of"] = "vector ALU lane-cycles that do work: VALU issue (capped at 1) x live lanes per instruction"
        else:
            phys["frac"] = units[top]
            phys["frac_of"] = {"ta_busy": "L1 address path (TA) busy cycles / cycles", "valu_issue": "VALU issue cycles / SIMD cycles (model)"}[top]
    return phys


def default_workload():
    """sponza when a complete scene directory has been supplied (master file and every mesh file it names), else the stand-in."""
    d = os.path.join(ROOT, "scenes", "sponza")
    master = os.path.join(d, "master.txt")
    if not os.path.exists(master):
        return "atrium"
    meshes = [t[1] for t in (line.split() for line in open(master, errors="replace")) if len(t) >= 2 and t[0] == "O"]
    return "sponza" if meshes and all(os.path.exists(os.path.join(d, m)) for m in meshes) else "atrium"


def load_host_scene(name):
    from pathtracing_amd import host
    packed, path_file, _ = WORKLOADS[name]
    if packed.startswith("gen:"):
        hs = host.HostScene.atrium(int(packed[4:]), 12345)
    elif packed.startswith("dir:"):
        hs = host.HostScene.from_dir(os.path.join(ROOT, packed[4:]))
    else:
        hs = host.HostScene.from_packed(os.path.join(ROOT, packed))
    st = hs.build_tree()
    cam = hs.camera()
    if path_file is not None and os.path.exists(os.path.join(ROOT, path_file)):
        views = list(host.views_from_path(os.path.join(ROOT, path_file), cam["zoom"]))
    elif path_file is not None:                      # no camera path: 16 yaw angles at the scene's own camera
        views = [host.PtCamera((C.c_double * 3)(*cam["eye"]), cam["angle"] + np.pi / 8 * j, cam["zoom"]) for j in range(16)]
    else:
        views = atrium_views(cam)
    return hs, st, views


def atrium_views(cam):
    """20 positions x 16 yaw angles along the nave, the shape of the reference's path files."""
    from pathtracing_amd import host
    views = []
    for k in range(20):
        x = -1500.0 + 150.0 * k
        for j in range(16):
            v = host.PtCamera((C.c_double * 3)(x, 300.0, -50.0 + 20.0 * ((k % 5) - 2)), cam["angle"] + np.pi / 8 * j, cam["zoom"])
            views.append(v)
    return views


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` started as one plain process: start the N ranks (one per GPU) as a
    torch.distributed.run child process -- a child, not an exec, and before this process has touched a GPU or
    imported torch -- pass rank 0's JSON line through on stdout (whatever else the ranks or their libraries write to
    stdout goes to stderr, so that stdout holds the one line) and return its exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, errors="replace")
    for line in child.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        (sys.stdout if line.startswith("{") else sys.stderr).flush()
    return child.wait()


def oracle_sample(orc, po, view, view_index, model, rb, spp, size, seconds):
    """The CPU oracle on a bounded sample of one view: every k-th image row, sized for about `seconds` of CPU work."""
    cores = min(16, os.cpu_count() or 1)
    okw = dict(eye=list(view.eye), angle=view.angle, zoom=view.zoom, model=po.MODEL_OMP if model == "omp" else po.MODEL_MAIN,
               gi=True, rb=rb, spp=spp, rng=po.RNG_PHILOX, view=view_index, size=size, threads=cores, aov=False)
    # one row per thread first, to size the sample
    probe = orc.render(row0=size // (2 * cores), row_step=size // cores, nrows_limit=cores, **okw)
    rows = int(min(size, max(cores, cores * round(seconds / max(probe["seconds"], 1e-3)))))
    rows = max(cores, (rows // cores) * cores)
    step_rows = max(1, size // rows)
    r = orc.render(row0=step_rows // 2, row_step=step_rows, nrows_limit=rows, **okw)
    crays = sum(v["searches"] for k, v in r["counts"].items() if k != "omp_retrace")
    return {"value": round(crays / r["seconds"] / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "host_cpus_visible": os.cpu_count(),       # the whole machine; a one-GPU job's share of it is 16 threads
            "views_per_s_extrapolated": round(rows / size / r["seconds"], 5),
            "sample": f"{rows} of {size} image rows (every {step_rows}th) of one view at {spp} spp{' with ray bundles' if rb else ''}: "
                      f"{crays} rays in {r['seconds']:.1f} s on {cores} threads"}


def rm_rb16_leg(ds, cams, size, rank, world, dev, xdev, dist, backend):
    """The multi-view job at the reference's settings, timed by itself: -gi -rb, mainOMP model, 16 samples
    (mainOMP.cpp:117-118,132), every view of the path, view v on rank v % world, frames gathered on rank 0."""
    import torch
    from pathtracing_amd import device, multiview
    opt = dict(gi=True, rb=True, model=device.MODEL_OMP, spp=16, gi_precision=device.GI_F32_FAST, size=size)
    keys = ("primary", "shadow", "mirror", "gi", "ms_gi_trace", "ms_primary", "gi_trace_launches")
    acc = dict.fromkeys(keys, 0.0)

    def render_many(first, stride, count):
        sel = cams[first: first + stride * count: stride]
        frames, sd = ds.render_views([device.make_camera(c[:3].tolist(), float(c[3]), float(c[4])) for c in sel],
                                     view0=first, view_stride=stride, **opt)
        for k in keys:
            acc[k] += sd[k]
        return frames.reshape(count, -1)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n = len(cams)
    render_many(rank, world, min(8, len(range(rank, n, world))))           # warm-up: the view workers and their workspaces
    for k in keys:
        acc[k] = 0.0
    got = [0]
    sync()
    t0 = time.perf_counter()
    multiview.run_views(None, n, size * size * 3, xdev, dist if world > 1 else None,
                        (lambda v, px: got.__setitem__(0, got[0] + 1)) if rank == 0 else None, batch=32, render_many=render_many)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt] + [acc[k] for k in keys], dtype=torch.float64, device=xdev)
    if world > 1:
        tmax = t[0:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        t[0] = tmax[0]
    v = [float(x) for x in t.cpu()]
    dt, tot = v[0], dict(zip(keys, v[1:]))
    if rank != 0:
        return None
    assert got[0] == n, f"rank 0 holds {got[0]} of {n} frames"
    rays = tot["primary"] + tot["shadow"] + tot["mirror"] + tot["gi"]
    return {"config": "-gi -rb -rm: mainOMP shading model, 16 spp, 3x3 ray bundles at +-0.1 px, 512x512, every view of the path",
            "n_views": n, "views_per_s": round(n / dt, 2), "ms_per_view": round(dt / n * 1e3, 4), "seconds": round(dt, 4),
            "scaling": "strong (the views of the path are shared by all ranks, view v on rank v % N)",
            "mrays_per_s": round(rays / dt / 1e6, 1),
            # device time per view of the two passes that make up a view (several views are in flight per GPU, so these overlap in wall time)
            "primary_ms": round(tot["ms_primary"] / n, 4), "closest_ms": round(tot["ms_gi_trace"] / n, 4),
            "stage_ms_note": "event-to-event time on the view's stream with several views in flight per GPU: each includes waiting for and running "
                             "beside other views' kernels, so the two add up to more than ms_per_view; one view at a time: bench.py --rb --spp 16",
            "frames_on_rank0": got[0], "backend": backend if world > 1 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("PT_BENCH_WORKLOAD", default_workload()), choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=1024, help="GI samples per pixel (the reference's converged sponza/room renders use 1024)")
    ap.add_argument("--model", default="omp", choices=["main", "omp"])
    ap.add_argument("--rb", action="store_true")
    ap.add_argument("--views-per-step", type=int, default=1,
                    help="views each GPU renders per step, handed to the library in one call (it keeps several small views in flight)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rm-leg", action="store_true", help="skip the second leg (the -gi -rb -rm job at the reference's settings)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path with every rank on GPU 0 and the exchanges staged through host memory")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, meet over gloo, print a JSON line with the world size and stop: checks the "
                         "launcher on a machine without a GPU")
    a = ap.parse_args()
    if a.workload == "room":
        a.rb = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus))
    a.gpus = world
    if a.rendezvous_only:
        import torch
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        seen = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(seen)
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "rank_sum": float(seen[0])}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    import torch
    import torch.distributed as dist
    from pathtracing_amd import device, multiview
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU")
    gpu = local_rank if a.backend == "nccl" else 0
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    xdev = dev if a.backend == "nccl" else torch.device("cpu")     # where the exchanged tensors live
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    device.init(gpu)
    model = device.MODEL_OMP if a.model == "omp" else device.MODEL_MAIN

    # ---- scene: read + kd build on rank 0, RCCL broadcast of the packed blob ----
    t_scene = time.time()
    hs = st = ds = None
    if rank == 0:
        hs, st, views = load_host_scene(a.workload)
        ds = device.DeviceScene(hs.desc())                    # the one place the scene is prepared (fp32 tree included)
        cams = np.array([[*v.eye, v.angle, v.zoom] for v in views])
    scene_broadcast = ranks = None
    if world > 1:
        # who is here: one line per rank, and -- with RCCL -- a check that the ranks really sit on different devices
        props = torch.cuda.get_device_properties(gpu)
        me = {"rank": rank, "local_rank": local_rank, "device": gpu, "device_name": props.name,
              "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xFF,
                                                getattr(props, "pci_device_id", 0)),
              "uuid": str(getattr(props, "uuid", ""))}
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        if a.backend == "nccl":
            ids = [(r["pci_bus_id"], r["uuid"]) for r in ranks]
            assert len(set(ids)) == world, f"ranks share a device: {ids}"
        dist.barrier()
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        ds = multiview.broadcast_scene(ds, xdev, dist)        # the scene's device data travels over xGMI once, GPU to GPU
        torch.cuda.synchronize()
        dist.barrier()
        scene_broadcast = {"bytes": int(ds.slab_size()), "ms": round((time.perf_counter() - t_b) * 1e3, 2),
                           "what": "slab exported device to device, one broadcast, every other rank's scene made from its copy (body checksum included)"}
        cams = multiview.broadcast_cameras(cams if rank == 0 else None, xdev, dist)
    scene_seconds = time.time() - t_scene
    views = [device.make_camera(c[:3].tolist(), float(c[3]), float(c[4])) for c in cams]
    nviews = len(views)

    size = 512
    opt = dict(gi=True, rb=a.rb, model=model, spp=a.spp, gi_precision=device.GI_F32_FAST, size=size)
    frame = torch.empty(size * size * 3, dtype=torch.uint8, device=dev)
    # Finished frames go to rank 0 in one gather per 8 steps (and at the end of the warm-up and of the timed steps):
    # a gather makes the ranks wait for each other, and views differ in cost.
    B = max(1, a.views_per_step)
    gather = multiview.FrameGather(frame.numel(), xdev, dist, batch=8 * B) if world > 1 else None

    # Weak scaling: step i renders camera i of the path on every rank, each rank with its own RNG key (the view index
    # i * world + rank of the multi-view job), so that the work per GPU is the same for every N and the N = 1, 2, 4, 8
    # lines differ by what the extra GPUs cost, not by which cameras happened to fall to them (views differ in cost
    # by tens of percent).
    def step(i, stats):
        if B > 1:
            # B views per rank in one call: cameras i*B .. i*B + B - 1 of the path, view indices (i*B + j) * world + rank
            frames, sd = ds.render_views([views[(i * B + j) % nviews] for j in range(B)], view0=i * B * world + rank,
                                         view_stride=world, **opt)
            for k in ("primary", "shadow", "mirror", "gi", "omp_retrace", "ms_gi_trace", "gi_trace_launches"):
                setattr(stats, k, getattr(stats, k) + sd[k])
            if world > 1:
                for j in range(B):
                    gather.slot(i * B + j).copy_(torch.from_numpy(frames[j].reshape(-1)))
                if gather.full():
                    gather.flush(1 << 30)
            return
        ds.render_to_device(views[i % nviews], frame.data_ptr(), stats=stats, view=i * world + rank, **opt)
        if world > 1:
            gather.slot(i).copy_(frame if a.backend == "nccl" else frame.cpu())
            if gather.full():
                gather.flush(1 << 30)

    def flush():
        if world > 1:
            gather.flush(1 << 30)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warm = device.PtStats()
    for i in range(a.warmup):
        step(i, warm)
    flush()
    stats = device.PtStats()
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i, stats)
    flush()                                  # inside the timed region: the job is done when rank 0 holds every frame
    sync()
    dt = time.perf_counter() - t0

    rays_local = stats.primary + stats.shadow + stats.mirror + stats.gi
    agg = torch.tensor([dt, float(rays_local), float(stats.gi), float(stats.primary), float(stats.mirror), float(stats.shadow)],
                       dtype=torch.float64, device=xdev)
    if world > 1:
        tmax = agg[0:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg[1:], op=dist.ReduceOp.SUM)
        agg[0] = tmax[0]
    dt_max, rays_total, gi_total, n_primary, n_mirror, n_shadow = (float(x) for x in agg.cpu())
    by_class = {"primary": n_primary, "mirror": n_mirror, "shadow": n_shadow, "gi": gi_total}

    # ---- second leg: the multi-view job at the reference's own settings ----
    rm = None if a.no_rm_leg else rm_rb16_leg(ds, cams, size, rank, world, dev, xdev, dist, a.backend)

    if rank == 0:
        # ---- algorithmic bytes per ray of the dominant kernel: a counting run (untimed) ----
        # over exactly the views the timed steps of this rank rendered
        c_gi = c_node = c_tri = 0
        for i in range(a.warmup * B, (a.warmup + a.steps) * B):
            cst = device.PtStats()
            ds.render_to_device(views[i % nviews], frame.data_ptr(), stats=cst, view=i * world + rank,
                                flags=device.FLAG_COUNT_WORK, **opt)
            c_gi += cst.gi; c_node += cst.gi_nodes_visited; c_tri += cst.gi_tris_tested
        n_node = c_node / max(c_gi, 1)
        n_tri = c_tri / max(c_gi, 1)
        # per closest-hit ray: 32 B ray in + 16 B hit out + 16 B per node visited + (4 + 36) B per triangle tested
        bytes_per_ray = 32 + 16 + 16 * n_node + 40 * n_tri
        launches = max(stats.gi_trace_launches, 1)
        ms_launch = stats.ms_gi_trace / launches
        rays_launch = stats.gi / launches
        achieved = bytes_per_ray * rays_launch / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        kernel_mrays = rays_launch / (ms_launch * 1e-3) / 1e6 if ms_launch > 0 else 0.0
        # The contract's yardstick: algorithmic bytes of the dominant kernel against the HBM peak.  The kernel is not
        # HBM-bound -- its nodes and triangles are served by L1 and L2, memory-side traffic is a small fraction of the
        # algorithmic bytes -- so the fraction says how much traffic the caches absorb, and can exceed 1; what bounds
        # the kernel physically is in "physical".
        roofline = {"bound": "hbm (algorithmic bytes vs the HBM peak: the contract's yardstick, not the physical limiter -- see physical)",
                    "kernel": "k_trace_closest", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "traffic_unit": "GB per launch",
                    "bytes_per_ray": round(bytes_per_ray, 1), "nodes_per_ray": round(n_node, 2),
                    "tris_per_ray": round(n_tri, 2), "rays_per_launch": int(rays_launch),
                    "ms_per_launch": round(ms_launch, 4),
                    "kernel_mrays_per_s": round(kernel_mrays, 1),
                    # warm-up launches included: the figure a kernel trace of this command averages over
                    "ms_per_launch_incl_warmup": round((stats.ms_gi_trace + warm.ms_gi_trace) /
                                                       max(stats.gi_trace_launches + warm.gi_trace_launches, 1), 4)}
        tr = pmc_traffic(a.workload, rays_launch)
        if tr:
            roofline["traffic"], roofline["traffic_source"], roofline["traffic_fetch_at_face_value"] = tr
            roofline["traffic_stale"] = pmc_means(a.workload)["kernel_source_hash"] != kernel_source_hash()
            roofline["algorithmic_gb_per_launch"] = round(bytes_per_ray * rays_launch / 1e9, 2)
            # what the memory side really moves (counters of the committed collection, scaled to this run's launches):
            # the kernel's HBM roofline position as a physicist would state it -- `achieved` / `frac` above are the
            # algorithmic bytes, most of which the caches serve
            if ms_launch > 0:
                roofline["hbm_measured_gb_per_s"] = round(tr[0] / (ms_launch * 1e-3), 1)
                roofline["hbm_measured_frac"] = round(tr[0] / (ms_launch * 1e-3) / HBM_PEAK_GBS, 4)
        roofline["physical"] = physical_roofline(a.workload, n_node, n_tri, kernel_mrays, ms_launch, rays_launch) if world == 1 else None
        cpu = cpu_rm = None
        if world == 1 and not a.no_cpu_baseline:
            from oracle import ptoracle as po
            packed = WORKLOADS[a.workload][0]
            tmp = None
            if packed.startswith("gen:") or packed.startswith("dir:"):
                tmp = os.path.join("/tmp", f"bench_{a.workload}_{os.getpid()}.ptscene")
                hs.save_packed(tmp)
                packed_path = tmp
            else:
                packed_path = os.path.join(ROOT, packed)
            orc = po.OracleScene(packed_path)
            if tmp:
                os.remove(tmp)
            cpu = oracle_sample(orc, po, views[a.warmup % nviews], a.warmup % nviews, a.model, a.rb, a.spp, size, a.cpu_seconds)
            if not a.no_rm_leg:
                cpu_rm = oracle_sample(orc, po, views[0], 0, "omp", True, 16, size, min(a.cpu_seconds, 8.0))

        out = {
            "metric": "Mrays/s (all ray classes of -gi views; GI sample rays dominate)",
            "value": round(rays_total / dt_max / 1e6, 1),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt_max / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (GI sample rays) + f64 (primary, mirror, their shadow rays)",
            "data": "synthetic" if a.workload == "atrium" else ("scene directory supplied by the user" if a.workload == "sponza" else "reference scene file, packed"),
            "config": {"workload": WORKLOADS[a.workload][2], "spp": a.spp, "model": a.model, "rb": bool(a.rb),
                       "image": "512x512", "views_per_step": world * B,
                       "parallelism": (f"one view per GPU per step on {world} GPU(s): camera = step, RNG key = step * GPUs + rank; frames gathered on rank 0" if B == 1 else
                                       f"{B} views per GPU per step on {world} GPU(s), one library call per step; frames gathered on rank 0"),
                       "triangles": st["triangles"], "kd_refs": st["optimized"], "kd_nodes": st["nodes"],
                       "device_tree": ds.info()},
            "views_per_s": round(a.steps * world * B / dt_max, 3),
            "gi_mrays_per_s": round(gi_total / dt_max / 1e6, 1),
            # rays of each class of all ranks over the same wall time (the classes are traced one after the other, not side by side)
            "mrays_per_s_by_class": {k: round(v / dt_max / 1e6, 2) for k, v in by_class.items()},
            "rays_by_class": {k: int(v) for k, v in by_class.items()},
            "backend": a.backend if world > 1 else None, "ranks": ranks, "scene_broadcast": scene_broadcast,
            "scene_setup_s": round(scene_seconds, 2),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "rm_rb16": rm,
        }
        if rm is not None:
            rm["cpu_baseline"] = cpu_rm
        print(json.dumps(out), flush=True)
    ds.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
