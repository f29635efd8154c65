#!/usr/bin/env python3
"""bench.py - Mrays/s of the HIP ray-trace hot path on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W

N > 1 needs no launcher: the process starts its own N ranks as child processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, rendezvous on 127.0.0.1) before it has imported torch or touched HIP, relays rank 0's JSON line and exits
non-zero if any rank failed.  Started under `python -m torch.distributed.run` it simply is one of the ranks.

Workload (config C4): synthetic 1,002,528-triangle height-field OBJ in 1,024 groups, written
to disk and loaded through the host OBJ loader, 1920x1080, 8 spp, bounce depth 2, seed 1234, the reference's
single directional light.  One "step" = one frame: ray generation -> traversal -> shading -> resolved float4
framebuffer, all on the device with the scene already resident in HBM; for N > 1 the frame is sharded in
interleaved 8-row blocks, one rank per GPU, and a step ends after the RCCL gather of the shards to rank 0
and their de-interleave on GPU 0.  Excluded, as in the reference's TIME_BLOCK("Render, sync")
(main.cpp:327): OBJ parse, hierarchy/BVH build, upload, tone map, PNG.

A ray is one TraceRay call (raytracer.cpp:161): primary, shadow, bounce.  value = rays of all ranks / time.
With several GPUs up to --frames-in-flight (default 2; 1 on a single GPU) consecutive frames overlap on each GPU, each on
its own context / stream: the persistent kernels of a frame leave the GPU partly idle while their last rays drain, and the
next frame fills that.  Every frame is rendered, gathered and assembled inside the timed region; `render_ms_device` stays
the per-frame device latency.  With N > 1 the render streams leave 8 compute units to the RCCL gather (PRT_RESERVE_CUS).
Total work is fixed as N grows -> "scaling": "strong".

Extra objects on the JSON line:
  roofline      the dominant kernel (k_pool: one launch = one frame): SURVEY.md §8d algorithmic bytes of a launch (rays x 52 B +
                BVH nodes fetched x the build's node size (64 B, 4-wide) + triangle tests x 36 B + shaded hits x 80 B + pixels
                x 16 B) / its duration, measured with HIP events on the kernel's own stream inside the timed region.  That
                figure counts every per-lane node fetch, most of which the caches serve, so its label says what it is
                ("cache-inclusive algorithmic bytes") and `hbm_measured` / `limiter` say what the memory system and the
                kernel really do.  `traffic` = HBM bytes per launch, FETCH_SIZE x 2 + WRITE_SIZE as the gfx950 guide prescribes,
                MEASURED IN THIS RUN (since round 4): before it touches the GPU the default command runs two short children of
                itself under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (a pass each, counters only; ~6 s each) and takes
                the timed frames' kernel from their counter files (`traffic_source` says so).  Where that is not possible (no
                rocprofv3, a refused or failing pass, N > 1, the quick forms with --no-cpu-baseline / --no-other-workloads) the
                committed profiles/traffic_C4.json is used - only when it was taken with the library that is running (its
                sha256 is in the file), else null.  The issue counters (valu_busy, ta_busy, wait_frac, L1 latency) always come
                from that file, under the same condition.
  cpu_baseline  the CPU oracle ("port", oracle/prt_oracle.cpp, bit-identical to the compiled reference on every fixture) timed
                on the physical cores of this host's socket 0, one pinned thread each, on a sparse pixel lattice of the SAME
                frame; rank 0, N = 1 only.  The same lattice is the parity check of the GPU frame (max |dRGB|, ray counts).
  extra.other_workloads   C2, C3, C5 and C4 with the reference's adaptive 10..50 spp sampling: 3 timed frames each and a lattice
                of the frame compared with the unmodified reference's own pixels (tests/golden/).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene generator, width, height, spp, bounce_depth, description)
    "C4": ("terrain_1m", 1920, 1080, 8, 2,
           "C4: synthetic 1,002,528-triangle terrain OBJ (1,024 groups), 1920x1080, 8 spp, bounce_depth 2"),
    "C3": ("icosphere_l6", 1920, 1080, 8, 2, "C3: displaced icosphere L6 (81,922 triangles, 65 groups), 1920x1080, 8 spp"),
    "C2": ("cornell_box", 512, 512, 4, 2, "C2: Cornell-box-style 12-triangle OBJ, 512x512, 4 spp"),
    "C5": ("terrain_1m", 3840, 2160, 64, 8, "C5: 1M-triangle terrain, 3840x2160, 64 spp, bounce_depth 8"),
    "tiny": ("terrain_64", 320, 180, 2, 2, "tiny: 8,192-triangle terrain, 320x180, 2 spp (plumbing check)"),
}
SHARD_BLOCK_ROWS = 8
SEED = 1234
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E spec peak
# SURVEY.md §8d algorithmic bytes per unit of work: ray 52 (32 read + 20 written), BVH node = the build's node (64 B, 4-wide;
# read from prt_scene_info.bvh_node_bytes),
# triangle test 36 (three float3), shaded hit 80 (3 normals + 3 indices + material), pixel 16.  `roofline.achieved` / `frac`
# use THESE.  The build's own records are fatter (48 B pre-differenced triangle, 64 B shading record + 64 B material + 16 B
# hit record = 144 B per shaded hit); the same figure with those sizes is reported next to it, not instead of it.
B_RAY, B_TRI, B_SHADE, B_PIXEL = 52, 36, 80, 16
B_TRI_BUILD, B_SHADE_BUILD = 48, 144

# The other BASELINE configs, timed in the same run (extra.other_workloads) and checked against the reference's golden pixels
# (tests/golden/*.npz: the unmodified reference's own output on a pixel lattice of exactly this frame).
#   key: (workload, golden fixture, adaptive max_spp or 0)
OTHER_WORKLOADS = [("C2", "c2_cornell_512_l4", 0), ("C3", "c3_icosphere_1080p_l24", 0), ("C5", "c5_terrain1m_4k_l120", 0),
                   ("C4-adaptive", "c4_terrain1m_adaptive_l60", 50)]


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench] " + msg, file=sys.stderr, flush=True)


def launch_ranks(n):
    """Start one child process per rank (what `python -m torch.distributed.run --nproc-per-node n` would do), wait for
    all of them, return 0 only if every rank exited 0.  Rank 0's stdout (the JSON line) is this process's stdout.  The
    reference's equivalent is `mpirun -n N` around main.cpp:311-347 (partition + MPI_Gather)."""
    import signal
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = set(range(n))
    kill_at = None
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("[bench] rank %d exited with code %d: stopping the other ranks" % (r, code), file=sys.stderr, flush=True)
                    for o in live:
                        procs[o].send_signal(signal.SIGTERM)           # exact PIDs we started
                    kill_at = time.time() + 20.0
            if kill_at is not None and time.time() > kill_at:
                for o in live:
                    procs[o].kill()
                kill_at = None
            time.sleep(0.05)
    except KeyboardInterrupt:
        for o in live:
            procs[o].kill()
        rc = 130
    return rc


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def physical_cpus_socket0():
    """One logical CPU per physical core of socket 0 (no torch, no oracle import: runs before anything touches the GPU)."""
    seen, cpus = set(), []
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        allowed = list(range(os.cpu_count() or 1))
    for c in allowed:
        base = "/sys/devices/system/cpu/cpu%d/topology/" % c
        try:
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
        except Exception:
            pkg, core = 0, c
        if pkg != 0 or (pkg, core) in seen:
            continue
        seen.add((pkg, core))
        cpus.append(c)
    return cpus or allowed[:1]


def time_reference_on_host(obj_dir, cam_pos, cam_dir, fov, width, height, spp, depth, core_seconds, lattice_arg):
    """cpu_baseline kind "reference": the UNMODIFIED reference (oracle/_ref/ref_harness = /root/reference/main.cpp compiled
    in the build container with build.sh:6's flags; the binary travels to the GPU box, the sources do not) on a sparse pixel
    lattice of the SAME frame, one process per physical core of socket 0, each pinned, each rendering the lattice rows
    ly % T == k - the reference's own scaling model (one MPI rank per core, main.cpp:311-347).  Runs BEFORE this process
    imports torch or touches HIP: the children are started by a process that has not initialised the GPU.
    Returns None when the binary is absent or does not run here (the port is the baseline then)."""
    import subprocess
    if not os.path.exists(REF_BIN):
        return None
    cpus = physical_cpus_socket0()[:64]
    T = len(cpus)
    base = [REF_BIN, "-w", str(width), "-h", str(height), "--fov", repr(float(fov)),
            "--camera_position"] + [repr(float(v)) for v in cam_pos] + ["--camera_facing"] + [repr(float(v)) for v in cam_dir] + \
           ["--bounce_depth", str(depth), "--reflection_samples", "1", "--specular_samples", "1",
            "-d", obj_dir.rstrip("/") + "/", "--obj", "scene.obj", "--spp", str(spp), "--seed", str(SEED), "--light-mode", "0"]
    tmp = tempfile.mkdtemp(prefix="prt_bench_ref_")

    def run(lattice, procs):
        ps = []
        for k in range(procs):
            cmd = base + ["--lattice", str(lattice), "--rows", str(k), str(procs), "--stats", os.path.join(tmp, "s%d.json" % k),
                          "--out", os.path.join(tmp, "o%d.f32" % k)]
            cpu = cpus[k % T]
            ps.append(subprocess.Popen(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                       preexec_fn=(lambda c=cpu: os.sched_setaffinity(0, {c}))))
        stats, img = [], None
        for k, pr in enumerate(ps):
            _, err = pr.communicate(timeout=3600)
            if pr.returncode != 0:
                raise RuntimeError("ref_harness exited %d: %s" % (pr.returncode, err.decode()[-500:]))
            with open(os.path.join(tmp, "s%d.json" % k)) as f:
                st = json.load(f)
            stats.append(st)
            part = np.fromfile(os.path.join(tmp, "o%d.f32" % k), dtype=np.float32).reshape(st["lattice_height"], st["lattice_width"], 4)
            if img is None:
                img = np.zeros_like(part)
            img[k::procs] = part[k::procs]
        return stats, img

    try:
        lattice = lattice_arg
        if lattice <= 0:
            probe = max(8, int(round((width * height / 400.0) ** 0.5)))
            st, _ = run(probe, 1)                                  # ~400 pixels on one core: seconds
            px = st[0]["lattice_width"] * st[0]["lattice_height"]
            per_px = st[0]["render_seconds"] / max(1, px)
            lattice = int(max(1, min(probe, round((width * height / (core_seconds / max(per_px, 1e-9))) ** 0.5))))
        t0 = time.perf_counter()
        stats, img = run(lattice, T)
        wall = time.perf_counter() - t0
    except Exception as e:                                         # not runnable here (no MPI library, ...): fall back to the port
        log("reference binary present but not usable here (%r): cpu_baseline falls back to the port" % (e,))
        return None
    rays = sum(s["ray_count"] for s in stats)
    render_s = max(s["render_seconds"] for s in stats)            # the ranks run side by side: the slowest one ends the frame
    return {"lattice": lattice, "image": img, "ray_count": int(rays), "render_seconds": float(render_s), "wall_seconds_with_scene_load": float(wall),
            "cores": T, "cpus": cpus, "core_seconds": float(sum(s["render_seconds"] for s in stats)),
            "sphere_check_count": int(sum(s["sphere_check_count"] for s in stats)), "mesh_check_count": int(sum(s["mesh_check_count"] for s in stats))}


def measure_traffic_live(workload, pipeline):
    """roofline.traffic measured in THIS run, on THIS box: FETCH_SIZE and WRITE_SIZE of the workload's kernels, each in a pass of
    its own (counters only, no tracing - the gfx950 guide's recipe) over a short child run of this very file's timed loop.  Called
    before this process touches the GPU; the children are ordinary child processes.  None when rocprofv3 is not there, fails,
    or takes too long - the committed file is the fallback."""
    import csv, glob, shutil, subprocess
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="prt_bench_pmc_")
    per_kernel = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(out, counter), "--",
                   sys.executable, os.path.abspath(__file__), "--workload", workload, "--pipeline", str(pipeline), "--steps", "3", "--warmup", "1",
                   "--no-cpu-baseline", "--no-other-workloads", "--no-live-traffic"]
            env = dict(os.environ, TMPDIR="/tmp")
            # its own session, so that a pass that hangs is ended WITH its children (a pass takes ~6 s; a profiler that hangs costs
            # the bench 90 s, not its line)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = pr.communicate(timeout=90)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                pr.communicate()
                return None, "rocprofv3 --pmc %s did not end within 90 s" % counter
            r = type("R", (), {"returncode": pr.returncode, "stderr": err})()
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-200:].replace("\n", " "))
            for f in glob.glob(os.path.join(out, counter, "**", "*_counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] != counter:
                            continue
                        k = row["Kernel_Name"].replace("void prt::", "").replace("prt::", "").split("(")[0]
                        d = per_kernel.setdefault(k, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                        d[counter][0] += float(row["Counter_Value"]); d[counter][1] += 1
    except Exception as e:                                       # a profiler that hangs or is refused must not cost the bench line
        return None, "live PMC passes not usable here: %r" % (e,)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    res = {}
    for k, d in per_kernel.items():
        nf, nw = d["FETCH_SIZE"][1], d["WRITE_SIZE"][1]
        if nf and nw:
            f_kib, w_kib = d["FETCH_SIZE"][0] / nf, d["WRITE_SIZE"][0] / nw
            # KiB units; FETCH_SIZE counts 64-byte requests as 32 on gfx950: corrected = 2 x FETCH + WRITE (upper bound), raw = FETCH + WRITE
            res[k] = {"launches": nf, "bytes_corrected": int((2.0 * f_kib + w_kib) * 1024), "bytes_raw": int((f_kib + w_kib) * 1024)}
    return (res or None), ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (a pass each, counters only) of 3 + 1 frames of this command, run by this "
                           "process as children before it touched the GPU")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--cpu-lattice", type=int, default=0, help="lattice stride of the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "reference", "port"],
                    help="what is timed on the host cores: the unmodified reference (oracle/_ref/ref_harness, one pinned process per "
                         "core) or the CPU restatement (oracle/prt_oracle.cpp, one pinned thread per core); auto = the reference "
                         "where its binary is present and runs, else the port")
    ap.add_argument("--cpu-seconds", type=float, default=120.0, help="CPU work of the baseline sample in core-seconds (the lattice is chosen to match)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip extra.other_workloads (C2, C3, C5, C4-adaptive)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes of a short child run before the GPU is touched); "
                         "the committed profiles/traffic_<workload>.json is used instead when it was taken with this library")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (host-staged) only exists to rehearse the N > 1 path on a 1-GPU box")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--force-collective", action="store_true",
                    help="--gpus 1 only: run the N > 1 machinery with ONE rank - the process group is initialised (RCCL with --backend "
                         "nccl), the render streams are CU-masked (PRT_RESERVE_CUS=8), two frames are in flight, every frame's shard "
                         "is gathered to rank 0 (= itself) on the side stream and assembled.  Everything of the multi-GPU path "
                         "except xGMI meets the real library (reference: the MPI_Gather of main.cpp:345-347)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="consecutive frames rendered concurrently (own context, stream and workspace each): the drain tail "
                         "of frame k overlaps the start of frame k + 1.  1 = strictly one frame at a time; 0 = auto: 1 on one "
                         "GPU (a full frame gains 2 %% and the kernel timings of the roofline would be taken under "
                         "contention), 2 on several (a 1/8-frame shard gains 13 %%)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: this process becomes the launcher.  It has not imported torch or touched HIP
        # and never will; the ranks are its children (no exec of a process that has initialised the GPU).
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if args.force_collective and world != 1:
        raise SystemExit("--force-collective is the one-rank rehearsal of the collective path: use it with --gpus 1")
    collective = world > 1 or args.force_collective        # the gather / assemble / frames-in-flight machinery runs
    if args.force_collective:
        # a one-rank process group of our own (no launcher): rendezvous on 127.0.0.1, a free port
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

    scene_name, width, height, spp, depth, descr = WORKLOADS[args.workload]
    # ---- the reference itself, timed on this host's cores (N = 1 only) - BEFORE torch / HIP are touched: its processes are
    # children of a process that has not initialised the GPU
    pre = None
    ref_run = None
    if world == 1 and rank == 0 and not collective:
        from par_raytracer_amd import scenes as _scenes
        t0 = time.time()
        _sc = _scenes.make_scene(scene_name)
        _dir = tempfile.mkdtemp(prefix="prt_bench_%s_" % scene_name)
        _scenes.write_obj(_sc, _dir, "scene.obj")
        pre = (_dir, list(_sc.camera_position), list(_sc.camera_facing), float(_sc.fov), time.time() - t0)
        if not args.no_cpu_baseline and args.cpu_baseline in ("auto", "reference"):
            ref_run = time_reference_on_host(_dir, pre[1], pre[2], pre[3], width, height, spp, depth, args.cpu_seconds, args.cpu_lattice)
            if ref_run is None and args.cpu_baseline == "reference":
                raise SystemExit("--cpu-baseline reference: oracle/_ref/ref_harness is missing or does not run on this host")
            if ref_run:
                log("reference on %d pinned cores: lattice %d, %d rays in %.1f s (%.4f Mrays/s)" % (
                    ref_run["cores"], ref_run["lattice"], ref_run["ray_count"], ref_run["render_seconds"], ref_run["ray_count"] / ref_run["render_seconds"] / 1e6))
    # ---- HBM traffic of the workload's kernels, measured on this box (N = 1 only; never from inside a profiler's own run)
    live_traffic, live_traffic_note = None, None
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    # (the quick forms of this command that the A/B scripts use - no CPU baseline, no other workloads - skip it too)
    if world == 1 and rank == 0 and not collective and not args.no_live_traffic and not under_profiler and not args.no_cpu_baseline and not args.no_other_workloads:
        t0 = time.time()
        live_traffic, live_traffic_note = measure_traffic_live(args.workload, args.pipeline)
        log("live HBM traffic passes: %s (%.0f s)" % ("ok, %d kernels" % len(live_traffic) if live_traffic else live_traffic_note, time.time() - t0))

    import torch
    import torch.distributed as dist

    from par_raytracer_amd import api, capi, scenes, sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if collective and args.backend == "nccl":
        # The RCCL gather of frame k runs while frame k + 1 renders (frames in flight).  The render kernels are persistent and
        # fill every wave slot they are offered: their streams are created with a CU mask that leaves 8 compute units free, so
        # the gather's kernels never wait for a render block to retire (csrc/prt_api.hip prt_create).
        os.environ.setdefault("PRT_RESERVE_CUS", "8")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            # the gather shares the GPU with persistent render kernels that fill every wave slot: let its kernels go first
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
            dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective payloads live
    if collective:
        # The context's CU-masked render streams (PRT_RESERVE_CUS) are created by hipExtStreamCreateWithCUMask, which has no
        # flags: they are BLOCKING streams, implicitly ordered against the legacy null stream - torch's default.  Everything
        # this rank does in torch from here on (gather, cat, assemble, event records) therefore runs on a stream of its own
        # (torch streams are non-blocking), so the frames in flight never serialise against it.
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))

    # ---- scene: rank 0 writes the OBJ once, every rank loads it (as every MPI rank of the reference does)
    t0 = time.time()
    if pre is not None:
        meta = list(pre[:4])
        t0 -= pre[4]
    elif rank == 0:
        scene = scenes.make_scene(scene_name)
        obj_dir = tempfile.mkdtemp(prefix="prt_bench_%s_" % scene_name)
        scenes.write_obj(scene, obj_dir, "scene.obj")
        meta = [obj_dir, list(scene.camera_position), list(scene.camera_facing), float(scene.fov)]
    else:
        meta = [None, None, None, None]
    if collective:
        dist.broadcast_object_list(meta, src=0)
    obj_dir, cam_pos, cam_dir, fov = meta
    t1 = time.time()
    hs = api.HostScene(obj_dir, "scene.obj", 0, cam_pos)
    t2 = time.time()
    r = api.Renderer(local_rank)
    info = r.upload(hs)
    t3 = time.time()
    log("scene %s: %d triangles; OBJ write %.1fs, parse %.2fs + hierarchy %.2fs, upload+BVH %.2fs (%d nodes, depth %d, %.1f MB resident)" % (
        scene_name, info.triangle_count, t1 - t0, hs.parse_seconds, hs.hierarchy_seconds, t3 - t2, info.bvh_node_count,
        info.bvh_max_depth, info.device_bytes / 1e6))

    cam = api.make_camera(fov, width, height, cam_pos, cam_dir)
    params = api.default_params(spp, SEED, bounce_depth=depth, pipeline=args.pipeline)

    # ---- frames in flight: F independent contexts on this GPU (scene replicated: 133 MB for C4), each with its own stream,
    # workspace and output buffer; frame k runs on context k % F from its own host thread (the render call blocks its
    # caller: the wavefront pipeline needs host round trips).  Every frame is still complete - rendered, gathered,
    # assembled - inside the timed region; only the GPU idle time at the end of one frame is filled by the next.
    F = args.frames_in_flight if args.frames_in_flight > 0 else (2 if collective else 1)
    renderers = [r]
    for _ in range(F - 1):
        extra = api.Renderer(local_rank)
        extra.upload(hs)
        renderers.append(extra)

    # ---- output buffers (device).  Shards are padded to the largest shard so the gather has equal sizes.
    my_rows = r.shard_rows(height, SHARD_BLOCK_ROWS, rank, world)
    assert my_rows == sharding.shard_rows(height, SHARD_BLOCK_ROWS, rank, world)
    max_rows = sharding.max_shard_rows(height, SHARD_BLOCK_ROWS, world)
    # more output buffers than contexts: a buffer is only reused NB frames later, so a gather that the GPU schedules late
    # (the persistent render kernels leave it few free wave slots) does not stall the frames behind it
    NB = F + 2 if collective else F
    shards = [torch.zeros((max_rows, width, 4), dtype=torch.float32, device=dev) for _ in range(NB)]
    shard = shards[0]
    torch.cuda.synchronize()               # the zero fills ran on torch's stream; the renderer writes from its own streams
    gathered = [None] * NB                 # per buffer: event recorded after the RCCL gather that read it
    gather_list = None
    row_index = None
    frame = None
    gather_events = []                     # (start, end) event pairs around every gather + assemble on the side stream
    if collective and rank == 0:
        gather_list = [torch.empty_like(shard) for _ in range(world)]
        row_index = torch.from_numpy(sharding.row_index(height, SHARD_BLOCK_ROWS, world)).to(dev)

    def render_frame(slot, bslot=None):
        """Render one frame into buffer `bslot` on context `slot` (called from a worker thread when F > 1)."""
        rr, buf = renderers[slot], shards[slot if bslot is None else bslot]
        if not collective:
            return rr.render_device(cam, params, width, height, 0, width * height, buf.data_ptr(), True)
        return rr.render_shard_device(cam, params, width, height, SHARD_BLOCK_ROWS, rank, world, buf.data_ptr(), True)

    def deliver_frame(slot):
        """Main thread, after the frame in buffer `slot` is rendered: gather the shards to rank 0 and assemble."""
        nonlocal frame
        if not collective:
            return
        buf = shards[slot]
        if args.backend == "nccl":
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
            dist.gather(buf, gather_list, dst=0)                    # RCCL over xGMI: 7 shards -> GPU 0 (asynchronous)
            if rank == 0:
                frame = sharding.assemble(torch.cat(gather_list, dim=0), row_index, height)
            gathered[slot] = torch.cuda.Event(enable_timing=True)
            gathered[slot].record()
            gather_events.append((ev0, gathered[slot]))
        else:
            host = buf.cpu()
            gl = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, gl, dst=0)
            if rank == 0:
                frame = sharding.assemble(torch.cat(gl, dim=0).to(dev), row_index, height)

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=F) if F > 1 else None

    def run_frames(n):
        """n frames, at most F in flight.  Returns the per-frame counters in frame order."""
        out, pending = [], []
        for k in range(n):
            slot, bslot = k % F, k % NB
            if len(pending) == F:                                   # the frame that used this context F frames ago
                b0, fut = pending.pop(0)
                out.append(fut.result() if pool else fut)
                deliver_frame(b0)
            if gathered[bslot] is not None:
                gathered[bslot].synchronize()                       # its gather must have read the buffer before we overwrite it
                gathered[bslot] = None
            pending.append((bslot, pool.submit(render_frame, slot, bslot) if pool else render_frame(slot, bslot)))
        for b0, fut in pending:
            out.append(fut.result() if pool else fut)
            deliver_frame(b0)
        return out

    def sync_all():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- set-up of every context (workspace allocation on the first call) - like the scene upload, not a step
    for slot in range(F):
        render_frame(slot)
    # ---- W warmup steps, then EXACTLY K timed steps
    run_frames(args.warmup)
    sync_all()
    t_start = time.perf_counter()
    counters = run_frames(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t_start
    rays_local = sum(c.ray_count for c in counters)
    trace_ms = [c.trace_kernel_ms for c in counters]
    render_ms = [c.render_ms for c in counters]

    t = torch.tensor([elapsed, float(rays_local)], dtype=torch.float64, device=cdev)
    if collective:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        rays_total = float(tsum[1].item())
    else:
        rays_total = float(rays_local)
    ms_per_step = elapsed * 1e3 / args.steps
    value = rays_total / elapsed / 1e6
    ranks_seen = 1
    if collective:
        one = torch.ones(1, dtype=torch.int32, device=cdev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)                 # every rank of the group took part in the collective path
        ranks_seen = int(one.item())

    # ---- roofline of the dominant kernel: one extra, untimed render with visit counting (identical pixels)
    pcount = api.default_params(spp, SEED, bounce_depth=depth, pipeline=args.pipeline | capi.FLAG_COUNT_VISITS)
    if not collective:
        cc = r.render_device(cam, pcount, width, height, 0, width * height, shard.data_ptr(), True)
    else:
        cc = r.render_shard_device(cam, pcount, width, height, SHARD_BLOCK_ROWS, rank, world, shard.data_ptr(), True)
    st = r.render_stats()
    n_px_local = my_rows * width if world > 1 else width * height
    used = int(cc.pipeline)                       # what PRT_PIPELINE_DEFAULT resolved to for this shard size
    pipeline_name = {1: "megakernel", 2: "wavefront", 3: "persistent", 4: "pool"}[used]
    kernel_name = {1: "k_render_mega", 2: "k_trace", 3: "k_render_persistent", 4: "k_pool"}[used]
    fused = used != 2
    # k_trace moves rays, nodes and triangle records; shading records and the framebuffer belong to k_shade / k_resolve.
    # The single-launch pipelines do all of it in the one kernel that is timed.
    # Shadow rays whose radiance-if-unoccluded is exactly zero are counted (ray_count is the reference's) but not traced: they
    # have no ray record, so they are not priced either
    rays_elided = int(st.elided_shadow_rays)
    rays_traced = int(cc.ray_count) - rays_elided
    B_NODE = int(info.bvh_node_bytes)           # the build's node: 64 B (4-wide), 80 B for a -DPRT_BVH8 library
    def algorithmic_bytes(b_tri, b_shade):
        n = rays_traced * B_RAY + cc.node_visits * B_NODE + cc.tri_tests * b_tri
        if fused:
            n += cc.shaded_hits * b_shade + n_px_local * B_PIXEL
        return n
    alg_bytes = algorithmic_bytes(B_TRI, B_SHADE)
    alg_bytes_build = algorithmic_bytes(B_TRI_BUILD, B_SHADE_BUILD)
    kernel_ms = float(np.mean(trace_ms))
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # The PMC figures come from a committed file (counters cannot be collected inside this run).  They are only printed when
    # the file says it was taken with THIS library: tools/profile_summary.py stamps it with the sha256 of libprt_hip.so, its ABI
    # version and build flags; anything else and traffic / the counter-derived fields are null, with the reason beside them.
    traffic = valu_busy = ta_busy = wait_frac = l1_latency = None
    traffic_note = "no profiles/traffic_%s.json" % args.workload
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tpath) and world == 1 and used in (2, 4):
        try:
            import hashlib
            with open(tpath) as f:
                tj = json.load(f)
            with open(os.path.join(ROOT, "par_raytracer_amd", os.path.basename(os.environ.get("PRT_HIP_LIB", "libprt_hip.so"))), "rb") as f:
                lib_sha = hashlib.sha256(f.read()).hexdigest()
            stamp = tj.get("library", {})
            if stamp.get("sha256") != lib_sha or stamp.get("abi_version") != int(capi.hip_lib().prt_abi_version()):
                traffic_note = "profiles/traffic_%s.json was taken with another build of libprt_hip.so (%s..., this one is %s...): not printed" % (
                    args.workload, str(stamp.get("sha256"))[:12], lib_sha[:12])
            else:
                traffic_note = "profiles/traffic_%s.json, taken with this library (sha256 %s..., commit %s)" % (args.workload, lib_sha[:12], stamp.get("commit"))
                traffic = tj.get("hbm_bytes_per_frame_" + kernel_name)
                valu_busy = tj.get("valu_busy_" + kernel_name)      # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x resident waves per SIMD (PMC)
                ta_busy = tj.get("ta_busy_" + kernel_name)          # TA_TA_BUSY / (GRBM_GUI_ACTIVE per XCD x compute units)
                wait_frac = tj.get("wait_frac_" + kernel_name)      # SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of a wave's cycles parked on s_waitcnt
                l1_latency = tj.get("l1_latency_clk_" + kernel_name)  # TCP_TCP_LATENCY_sum / TCP_TA_TCP_STATE_READ_sum
        except Exception as e:
            traffic = None
            traffic_note = "profiles/traffic_%s.json unreadable: %r" % (args.workload, e)
    if live_traffic:
        # this run's own measurement wins over the file: the timed frames' kernel is the instantiation of `kernel_name` that was
        # launched most often and, among those, moved the most bytes (the counting render's instantiation runs once; the EXACT
        # follow-up launch of the same template runs as often as the fast kernel and moves next to nothing)
        cand = [((v["launches"], v["bytes_corrected"]), k, v) for k, v in live_traffic.items() if k.startswith(kernel_name + "<") or k == kernel_name]
        if cand:
            _, lk, lv = max(cand)
            traffic = lv["bytes_corrected"] * max(1, int(cc.trace_kernel_launches))      # per frame: mean launch x launches of a frame
            traffic_note = "measured in this run: %s; kernel %s, %d launches, %.2f GB per launch corrected (2 x FETCH + WRITE), %.2f GB raw" % (
                live_traffic_note, lk, lv["launches"], lv["bytes_corrected"] / 1e9, lv["bytes_raw"] / 1e9)
    elif live_traffic_note:
        traffic_note += "; live measurement: " + live_traffic_note
    launches = max(1, int(cc.trace_kernel_launches))
    lane_util = None
    if st.wave_node_steps and st.wave_tri_steps:
        # what actually bounds the kernel: vector issue with divergent lanes.  Lane-level work / (64 x
        # wave-level steps) of the two traversal loops, from the counting render above.
        lane_util = {"node_loop": round(st.node_visits / (64.0 * st.wave_node_steps), 4),
                     "triangle_loop": round(st.tri_tests / (64.0 * st.wave_tri_steps), 4),
                     "wave_node_steps": int(st.wave_node_steps), "wave_triangle_steps": int(st.wave_tri_steps),
                     "rays_parked_for_the_exact_launch": int(st.parked_rays)}
    # `achieved` = algorithmic bytes per launch / mean launch duration; `traffic` = measured HBM bytes per launch.  Both are
    # also given per frame (launches_per_frame launches of the kernel make one frame).
    roofline = {"bound": "cache-inclusive algorithmic bytes", "kernel": kernel_name, "launches_per_frame": launches,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": int(traffic / launches) if traffic else None,
                "traffic_per_frame": traffic, "algorithmic_bytes_per_launch": int(alg_bytes / launches),
                "launch_ms_mean": round(kernel_ms / launches, 4),
                "algorithmic_bytes_per_frame": int(alg_bytes), "kernel_ms_per_frame": round(kernel_ms, 4),
                "per_frame": {"rays": int(cc.ray_count), "rays_traced": rays_traced, "rays_counted_not_traced": rays_elided,
                               "node_visits": int(cc.node_visits), "tri_tests": int(cc.tri_tests),
                               "shaded_hits": int(cc.shaded_hits), "pixels": int(n_px_local)},
                "bytes_per_unit": {"ray": B_RAY, "node": B_NODE, "tri_test": B_TRI, "shaded_hit": B_SHADE, "pixel": B_PIXEL},
                "with_this_builds_record_sizes": {"bytes_per_unit": {"tri_test": B_TRI_BUILD, "shaded_hit": B_SHADE_BUILD},
                                                  "achieved": round(alg_bytes_build / (kernel_ms * 1e-3) / 1e9, 2) if kernel_ms > 0 else 0.0,
                                                  "frac": round(alg_bytes_build / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kernel_ms > 0 else 0.0},
                # The algorithmic figure counts every per-lane node fetch; most are served by L1 / L2 / Infinity Cache.  What
                # the kernel is bounded by is vector issue with divergent lanes, with the texture addresser not far behind:
                "lane_utilisation": lane_util, "valu_busy": valu_busy, "ta_busy": ta_busy,
                # what bounds the kernel: not HBM (hbm_measured below) and not one unit - its waves hold 5 per SIMD (96 VGPRs) and
                # issue vector instructions for lanes of which little more than half carry a walking ray, while each waits on
                # its node-to-node chain for about half its cycles; profiles/r03_ab_bvh8.txt shows vector instructions, vector
                # memory instructions and chain length each returning a quarter of what they cost
                "limiter": {"kind": "vector issue at partial lane occupancy, latency of the dependent node chain at 5 waves per SIMD",
                            "valu_busy": valu_busy, "lane_utilisation": ({"node_loop": lane_util["node_loop"], "triangle_loop": lane_util["triangle_loop"]} if lane_util else None),
                            "wait_frac": wait_frac, "l1_latency_clk": l1_latency, "ta_busy": ta_busy},
                "traffic_source": traffic_note,
                # SURVEY.md §8d asks for these two beside the algorithmic figure: what the kernel really moved through HBM
                # (PMC, per second of kernel time, as a fraction of the 8 TB/s peak) and the compulsory minimum of a frame
                # (every ray record once, the resident scene once, the framebuffer once)
                "hbm_measured": ({"GBps": round(traffic / (kernel_ms * 1e-3) / 1e9, 1),
                                  "frac_of_peak": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} if traffic and kernel_ms > 0 else None),
                "compulsory_bytes_per_frame": int(rays_traced * B_RAY + info.device_bytes + n_px_local * B_PIXEL)}

    # ---- CPU baseline + parity on a sparse lattice of the same frame (rank 0, N = 1 only)
    cpu_baseline = None
    parity = None
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as orc
        # one thread per physical core of socket 0, pinned; the CPU model is part of the record
        cpus, cpu_model = orc.socket0_physical_cpus()
        cpus = cpus[:64]
        cores = len(cpus)
        orc.set_worker_cpus(cpus)
        lattice = ref_run["lattice"] if ref_run else args.cpu_lattice      # the reference's lattice, when it ran: same pixels for all three
        if lattice <= 0:
            # a bounded sample: --cpu-seconds core-seconds of CPU work (default 40).  Probe a very sparse lattice first, then scale.
            probe = max(8, int(round((width * height / 400.0) ** 0.5)))
            _, pc = orc.render(hs.desc, cam, params, width, height, probe, cores)
            per_px = pc.render_seconds * cores / max(1, ((width + probe - 1) // probe) * ((height + probe - 1) // probe))
            want_px = args.cpu_seconds / max(per_px, 1e-9)
            lattice = int(max(1, min(probe, round((width * height / want_px) ** 0.5))))
        cpu_img, octr = orc.render(hs.desc, cam, params, width, height, lattice, cores)
        orc.set_worker_cpus([])
        gpu_img, gctr = r.render_lattice(cam, params, width, height, lattice)
        diff = np.abs(gpu_img[:, :, :3] - cpu_img[:, :, :3])
        cpu_mrays = octr.ray_count / octr.render_seconds / 1e6
        cpu_baseline = {"value": round(cpu_mrays, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                        "cpu_model": cpu_model, "pinned_to_cpus": cpus,
                        "pinned_to": "tests/test_oracle_golden.py: the port reproduces the compiled, unmodified reference bit for bit "
                                     "(float framebuffer and all three DebugCounters) on 26 fixtures; the reference binary (oracle/_ref) was absent or did not run on this host, so the port stands in for it",
                        "sample": "every %dth pixel in x and y of the same %dx%d x %d spp frame (%d pixels, %d rays, %.1f s wall on %d pinned threads = %.0f core-seconds)" % (
                            lattice, width, height, spp, cpu_img.shape[0] * cpu_img.shape[1], octr.ray_count, octr.render_seconds, cores,
                            octr.render_seconds * cores),
                        "speedup_gpu_over_cpu": round(value / cpu_mrays, 1) if cpu_mrays > 0 else None}
        parity = {"pixels": int(cpu_img.shape[0] * cpu_img.shape[1]), "max_abs_diff_rgb": float(diff.max()),
                  "pixels_over_1e-4": int((diff.max(axis=2) > 1e-4).sum()),
                  "ray_count_gpu": int(gctr.ray_count), "ray_count_cpu": int(octr.ray_count),
                  "ray_count_equal": bool(gctr.ray_count == octr.ray_count), "against": "port"}
        if ref_run:
            # the unmodified reference ran on this host before the GPU was touched (time_reference_on_host): IT is the baseline
            # and the parity reference; the port, on the same lattice, is checked against it once more - at the headline's size
            ref_img = ref_run["image"]
            ref_mrays = ref_run["ray_count"] / ref_run["render_seconds"] / 1e6
            port_record = {"value": cpu_baseline["value"], "cores": cores, "speedup_gpu_over_cpu": cpu_baseline["speedup_gpu_over_cpu"],
                           "equals_reference_bitwise": bool(np.array_equal(ref_img.view(np.uint32), cpu_img.view(np.uint32))),
                           "ray_count_equals_reference": bool(int(octr.ray_count) == ref_run["ray_count"])}
            cpu_baseline = {"value": round(ref_mrays, 4), "unit": "Mrays/s", "cores": ref_run["cores"], "kind": "reference",
                            "cpu_model": cpu_model, "pinned_to_cpus": ref_run["cpus"],
                            "what": "oracle/_ref/ref_harness: /root/reference/main.cpp compiled unmodified in the build container (oracle/Makefile, "
                                    "build.sh:6's flags, real MPI header); one process per physical core of socket 0, each pinned and rendering "
                                    "the lattice rows ly % cores == k - the reference's own model of one rank per core (main.cpp:311-347)",
                            "sample": "every %dth pixel in x and y of the same %dx%d x %d spp frame (%d pixels, %d rays; slowest process %.1f s, all "
                                      "processes %.0f core-seconds; %.1f s wall with every process's OBJ parse and BuildHierarchy)" % (
                                          lattice, width, height, spp, ref_img.shape[0] * ref_img.shape[1], ref_run["ray_count"], ref_run["render_seconds"],
                                          ref_run["core_seconds"], ref_run["wall_seconds_with_scene_load"]),
                            "speedup_gpu_over_cpu": round(value / ref_mrays, 1) if ref_mrays > 0 else None,
                            "port_on_the_same_lattice": port_record}
            dref = np.abs(gpu_img[:, :, :3] - ref_img[:, :, :3])
            parity = {"pixels": int(ref_img.shape[0] * ref_img.shape[1]), "max_abs_diff_rgb": float(dref.max()),
                      "pixels_over_1e-4": int((dref.max(axis=2) > 1e-4).sum()),
                      "ray_count_gpu": int(gctr.ray_count), "ray_count_cpu": int(ref_run["ray_count"]),
                      "ray_count_equal": bool(int(gctr.ray_count) == ref_run["ray_count"]), "against": "reference"}

    # ---- the other BASELINE configs, in the same run: time a few frames, check a lattice against the reference's own pixels
    other = None
    if world == 1 and rank == 0 and args.workload == "C4" and not args.no_other_workloads:
        other = []
        golden_dir = os.path.join(ROOT, "tests", "golden")
        for wl, fixture, max_spp in OTHER_WORKLOADS:
            gpath = os.path.join(golden_dir, fixture + ".npz")
            base_wl = "C4" if wl == "C4-adaptive" else wl
            sname, w2, h2, spp2, depth2, descr2 = WORKLOADS[base_wl]
            if max_spp:
                spp2 = 10                                      # the reference's own default: RenderPixel(min 10, max 50), main.cpp:308-309
            try:
                if sname == scene_name:
                    r2, own = r, False
                    cam_pos2, cam_dir2, fov2 = cam_pos, cam_dir, fov
                else:
                    sc2 = scenes.make_scene(sname)
                    d2 = tempfile.mkdtemp(prefix="prt_bench_%s_" % sname)
                    scenes.write_obj(sc2, d2, "scene.obj")
                    cam_pos2, cam_dir2, fov2 = list(sc2.camera_position), list(sc2.camera_facing), float(sc2.fov)
                    hs2 = api.HostScene(d2, "scene.obj", 0, cam_pos2)
                    r2, own = api.Renderer(local_rank), True
                    r2.upload(hs2)
                cam2 = api.make_camera(fov2, w2, h2, cam_pos2, cam_dir2)
                p2 = api.default_params(spp2, SEED, bounce_depth=depth2, pipeline=args.pipeline, max_spp=max_spp)
                n_out = w2 * h2
                buf2 = torch.zeros((n_out, 4), dtype=torch.float32, device=dev)
                torch.cuda.synchronize()
                r2.render_device(cam2, p2, w2, h2, 0, n_out, buf2.data_ptr(), True)           # workspace allocation + warm-up
                t0 = time.perf_counter()
                cs = [r2.render_device(cam2, p2, w2, h2, 0, n_out, buf2.data_ptr(), True) for _ in range(3)]
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3.0
                rec = {"workload": wl, "description": descr2 + (", adaptive %d..%d spp (the reference's default sampling)" % (spp2, max_spp) if max_spp else ""),
                       "steps": 3, "ms_per_frame": round(dt * 1e3, 4), "kernel_ms_per_frame": round(float(np.mean([c.trace_kernel_ms for c in cs])), 4),
                       "rays_per_frame": int(cs[0].ray_count), "mrays_per_s": round(cs[0].ray_count / dt / 1e6, 1),
                       "pipeline": {1: "megakernel", 2: "wavefront", 3: "persistent", 4: "pool"}[int(cs[0].pipeline)]}
                if os.path.exists(gpath):
                    g = np.load(gpath, allow_pickle=False)
                    lat = int(g["lattice"])
                    img, c = r2.render_lattice(cam2, p2, w2, h2, lat)
                    dd = np.abs(img[:, :, :3] - g["rgb"])
                    # the timed frames went through the full-size path (C5: 9 passes of whole pixels): its pixels at the lattice
                    # positions must be the lattice render's, bit for bit, and with them the reference's
                    full = buf2.reshape(h2, w2, 4)[::lat, ::lat].cpu().numpy()
                    rec["full_frame_path"] = {"equals_lattice_render_bitwise": bool(np.array_equal(full.view(np.uint32), img.view(np.uint32))),
                                              "max_abs_diff_rgb_vs_reference": float(np.abs(full[:, :, :3] - g["rgb"]).max()),
                                              "launches_per_frame": int(cs[0].trace_kernel_launches)}
                    rec["parity_vs_reference_golden"] = {"fixture": fixture, "lattice": lat, "pixels": int(img.shape[0] * img.shape[1]),
                                                         "max_abs_diff_rgb": float(dd.max()), "pixels_over_1e-4": int((dd.max(axis=2) > 1e-4).sum()),
                                                         "ray_count_gpu": int(c.ray_count), "ray_count_reference": int(g["ray_count"]),
                                                         "ray_count_equal": bool(int(c.ray_count) == int(g["ray_count"]))}
                other.append(rec)
                del buf2
                if own:
                    r2.close()
            except Exception as e:                                 # an extra must not take the headline line down with it
                other.append({"workload": wl, "error": repr(e)[:300]})

    # ---- one GPU, two frames in flight: what a caller that renders a sequence of frames gets (the drain tail of frame k and
    # the exact-phase launches behind it overlap the start of frame k + 1).  Not `value`: the roofline's kernel durations are
    # taken one frame at a time.
    pipelined = None
    if world == 1 and rank == 0 and F == 1 and args.workload == "C4" and not args.no_other_workloads:
        try:
            r_b = api.Renderer(local_rank)
            r_b.upload(hs)
            buf_b = torch.zeros((max_rows, width, 4), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            pair = [(r, shard), (r_b, buf_b)]
            def lane(t, n):                      # context t renders frames t, t + 2, ... one after the other
                rr, bb = pair[t]
                return [rr.render_device(cam, params, width, height, 0, width * height, bb.data_ptr(), True) for _ in range(t, n, 2)]
            lane(1, 2)
            n_fr = max(4, args.steps)
            with ThreadPoolExecutor(max_workers=2) as ex:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                futs = [ex.submit(lane, t, n_fr) for t in range(2)]
                cs = [c for f in futs for c in f.result()]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            pipelined = {"frames_in_flight": 2, "frames": n_fr, "ms_per_frame": round(dt * 1e3 / n_fr, 4),
                         "mrays_per_s": round(sum(c.ray_count for c in cs) / dt / 1e6, 1)}
            # the same in the reference's adaptive 10..50 spp mode, whose frame ends with a long thin tail (the pixels started last
            # run their up to 50 samples one after the other): a second frame in flight fills it
            p_ad = api.default_params(10, SEED, bounce_depth=depth, pipeline=args.pipeline, max_spp=50)
            def lane_ad(t, n):
                rr, bb = pair[t]
                return [rr.render_device(cam, p_ad, width, height, 0, width * height, bb.data_ptr(), True) for _ in range(t, n, 2)]
            lane_ad(0, 1); lane_ad(1, 2)
            with ThreadPoolExecutor(max_workers=2) as ex:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                futs = [ex.submit(lane_ad, t, 4) for t in range(2)]
                cs = [c for f in futs for c in f.result()]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            pipelined["adaptive_10_50"] = {"frames": 4, "ms_per_frame": round(dt * 1e3 / 4, 4), "mrays_per_s": round(sum(c.ray_count for c in cs) / dt / 1e6, 1)}
            r_b.close()
            del buf_b
        except Exception as e:
            pipelined = {"error": repr(e)[:300]}

    multi_check = None
    collective_info = None
    if collective and rank == 0:
        gms = []
        for a, b in gather_events[-args.steps:]:
            try:
                gms.append(a.elapsed_time(b))
            except Exception:
                pass
        collective_info = {"backend": args.backend, "world_size": world, "forced_at_world_size_1": bool(args.force_collective),
                           "reserved_cus": int(os.environ.get("PRT_RESERVE_CUS", "0")), "frames_in_flight": F,
                           "gathers_timed": len(gms),
                           "gather_and_assemble_ms_mean": round(float(np.mean(gms)), 4) if gms else None,
                           "gather_and_assemble_ms_max": round(float(np.max(gms)), 4) if gms else None,
                           "note": "device time from the gather's enqueue to the end of the assembly on the side stream (they run beside "
                                   "the next frame's render kernels, so this is latency, not time added to a frame)"}
    if collective and rank == 0 and frame is not None:
        # the assembled multi-GPU frame must be bit-identical to what one GPU computes for the same pixels
        lat = 16
        one, _ = r.render_lattice(cam, params, width, height, lat)
        got = frame[::lat, ::lat].cpu().numpy()
        multi_check = {"pixels": int(one.shape[0] * one.shape[1]),
                       "bit_identical_to_single_gpu": bool(np.array_equal(one.view(np.uint32), got.view(np.uint32))),
                       "max_abs_diff": float(np.abs(one - got).max())}

    if rank == 0:
        out = {
            "metric": "Mrays/s at 1920x1080x8spp, 1M-tri OBJ" if args.workload == "C4" else "Mrays/s (%s)" % args.workload,
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # `value` divides the reference's ray_count (which the device reproduces exactly) by the time; the device does not
            # trace the shadow rays that cannot change the image: the rate over the rays it really traces, for the same frames
            "value_traced_rays_only": round(value * rays_traced / max(1, int(cc.ray_count)), 3) if world == 1 else None,
            "ranks_seen": ranks_seen,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": descr, "seed": SEED, "triangles": int(info.triangle_count), "width": width, "height": height,
                       "spp": spp, "bounce_depth": depth, "rays_per_frame": int(rays_total / args.steps),
                       # A ray is one TraceRay call of the reference (raytracer.cpp:161) and ray_count equals the reference's.  Of
                       # those, the shadow rays whose radiance-if-unoccluded is exactly zero (surface facing away from the light, no
                       # highlight) cannot change the image and are counted without being traced; the option TRACE_DEAD_SHADOW_RAYS=1
                       # traces them too.  `value_traced_rays_only` is the rate without them.
                       "ray_definition": "one TraceRay call of the reference (raytracer.cpp:161): `value` divides the reference's ray_count, which the "
                                         "device reproduces exactly, by the time; value_traced_rays_only and the roofline count only the rays the device traces",
                       "rays_counted_not_traced_per_frame": rays_elided if world == 1 else None,
                       "value_traced_rays_only": round(value * rays_traced / max(1, int(cc.ray_count)), 3) if world == 1 else None,
                       "parallelism": "pixel rows sharded in %d-row blocks over %d GPU(s)%s" % (
                           SHARD_BLOCK_ROWS, world, (", RCCL gather to rank 0" if args.backend == "nccl" else ", gloo gather (rehearsal)") if collective else ""),
                       "pipeline": pipeline_name, "frames_in_flight": F},
            "render_ms_device": round(float(np.mean(render_ms)), 4),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity, "multi_gpu_check": multi_check,
            "collective": collective_info,
            "extra": {"other_workloads": other, "two_frames_in_flight_one_gpu": pipelined},
        }
        print(json.dumps(out), flush=True)
    if pool:
        pool.shutdown()
    for rr in renderers:
        rr.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
