#!/usr/bin/env python3
"""Benchmark of the rendering hot path on MI355X.

A "step" is one renderImage frame.  Default workload: the configuration
BASELINE.json's metric is quoted on -- the dragon (a deterministic 87 130
triangle stand-in, data/dragon.obj is not in the reference snapshot) at
1920x1080, 16 pixel samples, one parallelogram area light with 4 shadow
samples, reflections with a depth cap of 4, default camera.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU under torch.distributed.run.  Started by such a
launcher (RANK / WORLD_SIZE in the environment) the process is one rank of the
job; started plainly as `python bench.py --gpus N`, it starts the launcher
itself as a child process -- before this process has made any GPU call -- and
passes on rank 0's JSON line and the exit status.  Every rank renders its
interleaved 8x8 tiles of the frame (scene and BVH replicated, no data-path
collective), the only collective is the RCCL gather of the packed tile buffers
to rank 0, which scatters them into Screen order.  The frame is the same
whatever N is, so scaling is "strong".

`--launch-check` runs the N > 1 control flow without a GPU: the ranks
rendezvous over gloo, "render" their tiles of a synthetic frame on the CPU,
gather, de-interleave, reduce the timings like the real run, and rank 0 prints
one JSON line with the ranks the collective library saw.

Frames in flight: consecutive steps go round-robin to `--in-flight` (default 3; 4 at N = 2, 6 from N = 4) contexts, each
with its own stream, workspace and output buffers, the way a double- or triple-buffered
renderer runs.  Every frame is still rendered, gathered and post-processed completely; the
persistent traversal kernels of one frame merely start on the compute units the previous
frame's kernels are draining from, which a single stream leaves idle.  `--in-flight 1` gives
the one-frame-at-a-time latency.

Rank 0 prints one JSON line.  value = rays of all ranks / wall time of the
timed region (barrier + synchronize on both sides, max over ranks), rays being
BVHInterface::intersect calls counted by the kernel.  The scene, BVH, lights
and features are resident in HBM before the timed region; the frame stays on
the device.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# With N > 1 the collective library's own stream joins the frames' streams; HIP maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams on one queue run one after the
# other (measured at N = 1: a fourth frame in flight on four queues costs 12 %, on eight it does not).
# Has to be set before the HIP runtime starts; RT_BENCH_HW_QUEUES overrides the 8.
def _hw_queues_for_many_ranks(env):
    env["GPU_MAX_HW_QUEUES"] = env.get("RT_BENCH_HW_QUEUES", "8")


if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    _hw_queues_for_many_ranks(os.environ)

sio = importlib.import_module("ray-tracer_amd.scene_io")
capi = importlib.import_module("ray-tracer_amd.capi")
sharding = importlib.import_module("ray-tracer_amd.sharding")

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E, spec
FP32_PEAK_TFLOPS = 157.3  # MI355X vector FP32, spec (packed fused multiply-adds)
SHADER_CLOCK_GHZ = 2.4    # MI355X peak engine clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time reads 2.4 - 2.5 under this load)
L1_LOOKUPS_PER_CU_CLOCK = 1.10  # sixteen-byte loads of single lanes one compute unit's L1 serves per clock (profiles/round2_a_ta_probe.txt)


def build_workload(name: str):
    """Returns (scene, features, width, height, description)."""
    data = os.path.join(ROOT, "tests", "golden", "data")
    if name == "dragon":
        s = sio.load_scene_prebuilt("Dragon", data)
        s.lights = [sio.cornell_parallelogram_light()]
        f = sio.default_features(enableAccelStructure=1, enableShading=1, enableShadows=1, enableReflections=1,
                                 shadingModel=sio.SHADING_BLINN_PHONG, numPixelSamples=16, numShadowSamples=4,
                                 maxRayDepth=4)
        desc = ("dragon stand-in (87130 tris) 1920x1080, 16 spp, parallelogram light x4 shadow samples, "
                "Blinn-Phong, reflections depth cap 4, default camera")
        return s, f, 1920, 1080, desc
    if name == "teapot":
        s = sio.load_scene_prebuilt("Teapot", data)
        f = sio.default_features(enableAccelStructure=1, enableShading=1, enableShadows=1,
                                 shadingModel=sio.SHADING_BLINN_PHONG)
        return s, f, 1920, 1080, "teapot 1920x1080, 1 spp, point light shadows, Blinn-Phong, default camera"
    if name == "cornell":  # src/scene.cpp:34-39
        s = sio.load_scene_prebuilt("CornellBox", data)
        f = sio.default_features(enableAccelStructure=1, enableShading=1, enableShadows=1, enableReflections=1)
        return s, f, 512, 512, "CornellBox preset 512x512, 1 spp, point light, shading + reflections + shadows"
    if name in ("monkey", "cube"):  # src/scene.cpp:75-81 / :29-33
        s = sio.load_scene_prebuilt("Monkey" if name == "monkey" else "CubeTextured", data)
        # six faces of the size of the reference's sky textures (src/scene.cpp:112-115: 1024 x 1024, held as
        # float3: 12.6 MB each, 75 MB together -- the one read-only input of the path that does not fit L2);
        # procedural, the PNGs themselves are not carried
        sio.attach_synthetic_env_map(s, size=1024)
        f = sio.default_features(enableAccelStructure=1, enableShading=1, enableShadows=1, enableNormalInterp=1,
                                 enableTextureMapping=1, enableBilinearTextureFiltering=1, enableEnvironmentMap=1)
        return s, f, 1920, 1080, (f"{'Monkey' if name == 'monkey' else 'CubeTextured'} preset 1920x1080, 1 spp, interpolated normals, "
                                  "bilinear textures, environment map (six procedural 1024x1024 float3 faces, 75 MB)")
    if name == "dragon4k":
        s = sio.load_scene_prebuilt("Dragon", data)
        s.lights = [sio.cornell_parallelogram_light()]
        s.meshes[0].material["transparency"] = 0.6
        f = sio.default_features(enableAccelStructure=1, enableShading=1, enableShadows=1, enableReflections=1,
                                 shadingModel=sio.SHADING_BLINN_PHONG, numPixelSamples=64, numShadowSamples=4,
                                 enableTransparency=1, enableDepthOfField=1, depthOfFieldDistance=2.5,
                                 maxRayDepth=4)
        return s, f, 3840, 2160, "dragon stand-in 3840x2160, 64 spp, area light, DoF x5, transparency 0.6, depth cap 4"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(scene, features, width, height, scale: int, distance: float = 3.0):
    """The CPU oracle (a port of renderImage, OpenMP over rows like src/render.cpp:27-29) on the
    same workload: at the workload's own size one frame (about ten seconds of all host cores for
    the headline frame), at 1/scale of the resolution the median of three."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    osc = orc.OracleScene(scene)
    w, h = width // scale, height // scale
    cam = sio.default_camera(w, h, distance=distance)
    times = []
    for _ in range(1 if scale == 1 else 3):
        t0 = time.perf_counter()
        _, st = osc.render(features, cam, w, h, policy=0, threads=0)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    size = f"the whole workload ({w}x{h}), one frame" if scale == 1 else f"same scene/features at {w}x{h} (1/{scale * scale} of the pixels), median of 3 frames"
    return {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{size}, {dt:.1f} s, {st['rays']} rays", "rays": int(st["rays"])}


def stored_counters(workload, kernel, world, width, height, distance=3.0):
    """Hardware counters of `kernel`, mean per launch, from the committed passes over this workload
    (profiles/pmc_counters_<workload>.json, written by tools/pmc.sh at N = 1 and the workload's own
    resolution), and where they come from; (None, why) when no pass covers this run.  rocprofv3 cannot
    collect counters inside this process, so these are stored numbers and are labelled so; the hash of the
    kernel sources stored with them says whether they belong to the library that is running."""
    path = os.path.join(ROOT, "profiles", f"pmc_counters_{workload}.json")
    if world != 1 or distance != 3.0 or not os.path.exists(path):
        return None, "no stored counter pass for this configuration", False
    _, _, w0, h0, _ = build_workload(workload)
    if (w0, h0) != (width, height):
        return None, "no stored counter pass at this resolution", False
    with open(path) as fh:
        doc = json.load(fh)
    k = doc["kernels"].get(kernel)
    if k is None:
        return None, "kernel not in the stored counter pass", False
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summarise
    same = pmc_summarise.sources_sha1() == doc.get("sources_sha1")
    return k, (f"stored: profiles/{doc['summary']} ({doc['source']}); "
               + ("collected on these kernel sources" if same else "the kernel sources have changed since")), same


def valu_probe_peak(waves_per_simd=6):
    """The vector ALU's measured issue rate for the instruction mix of one traversal step (tools/valu_probe.hip,
    stored as profiles/valu_probe_gfx950.json), at the shadow kernel's occupancy."""
    path = os.path.join(ROOT, "profiles", "valu_probe_gfx950.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rows = json.load(fh)["rows"]
    for r in rows:
        if r["stream"].startswith("mix of one sibling-pair step") and r["waves_per_simd"] == waves_per_simd:
            return r
    return None


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a launcher: start torch.distributed.run with N ranks
    of this same command as a child process and hand back its exit status.  Runs before this
    process has imported torch.cuda or made any HIP call; the ranks are fresh processes (a
    process that has touched the GPU must not be replaced by another program)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    _hw_queues_for_many_ranks(env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC handles on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def reduce_over_ranks(dt: float, rays: float, world: int, dist, device):
    """(max over ranks of the timed region, sum over ranks of the rays of one frame)."""
    import torch
    t_all = torch.tensor([dt, rays], dtype=torch.float64, device=device)
    if world == 1:
        return dt, rays
    tmax, tsum = t_all.clone(), t_all.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return float(tmax[0]), float(tsum[1])


def launch_check(args, rank: int, world: int):
    """The control flow of an N-rank run with the GPU taken out: rendezvous (gloo), every rank
    fills its own tiles of a synthetic frame, the one gather, rank 0's de-interleave, the
    reduction of the timings.  Rank 0 prints what the collective library saw."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 100, 52  # partial tiles on both edges
    ys, xs = np.mgrid[0:h, 0:w]
    frame = np.stack([xs, ys, xs * 1000 + ys], axis=-1).astype(np.float32)
    stride = sharding.shard_floats(w, h, 0, world) if world > 1 else w * h * 3
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        if world > 1:
            packed = torch.from_numpy(sharding.pack_shard(frame, rank, world))
            gathered = torch.empty((world, stride), dtype=torch.float32) if rank == 0 else None
            sharding.gather_packed(packed, rank, world, dist, gathered)
            whole = sharding.deinterleave(gathered.numpy(), w, h, world) if rank == 0 else None
        else:
            whole = frame
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    mine = float(sharding.shard_tiles(w, h, rank, world))
    dt_max, units = reduce_over_ranks(dt, mine, world, dist, torch.device("cpu"))
    seen = dist.get_world_size() if world > 1 else 1
    if rank == 0:
        ok = bool(np.array_equal(whole, frame)) and int(units) == ((w + 7) // 8) * ((h + 7) // 8)
        print(json.dumps({"launch_check": ok, "n_gpus": args.gpus, "world_size": seen, "tiles": int(units),
                          "max_rank_seconds": round(dt_max, 4)}), flush=True)
        if not ok:
            raise SystemExit("launch check: the gathered frame differs from the synthetic one")
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-scale", type=int, default=0,
                    help="CPU baseline at 1/scale of the resolution; 0 = the workload's own size when that is at most 120 M rays "
                         "(the headline frame: 69 M), else the smallest power of two that gets it there")
    ap.add_argument("--camera-distance", type=float, default=3.0,
                    help="the reference's default camera is at 3.0 (src/config.h:18-23); 1.6 fills the frame with the model")
    ap.add_argument("--in-flight", type=int, default=0, help="frames in flight (contexts/streams used round-robin)")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="NAME=VALUE",
                    help="rt_ctx_set_option on every context (A/B measurements), e.g. reference_boxes=0")
    ap.add_argument("--launch-check", action="store_true",
                    help="run the N-rank control flow (launcher, rendezvous, gather, reductions) on the CPU over gloo and leave")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))  # this process never touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    num_cu = 256
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): use --nproc-per-node {args.gpus}")
    if args.launch_check:
        return launch_check(args, rank, world)

    import torch
    import torch.distributed as dist

    if torch.cuda.device_count() < world:  # (counting devices does not start the GPU runtime)
        raise SystemExit(f"--gpus {world} but this node shows {torch.cuda.device_count()} GPU(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rendering path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    num_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    scene, features, width, height, desc = build_workload(args.workload)
    if args.width and args.height:
        width, height = args.width, args.height
        desc += f" [resolution overridden to {width}x{height}]"
    camera = sio.default_camera(width, height, distance=args.camera_distance)
    if args.camera_distance != 3.0:
        desc += f" [camera distance {args.camera_distance} instead of the default 3.0]"

    # three at N = 1 (measured: four to six frames in flight are no faster at the headline size); more on a
    # shard, where the fixed part of a frame weighs more (an eighth of the frame, eight hardware queues:
    # 0.92 ms a frame with three, 0.86 with four, 0.85 with six; half the frame: no difference)
    lanes = args.in_flight if args.in_flight > 0 else (3 if world == 1 else (4 if world < 4 else 6))
    stride = capi.shard_floats(width, height, 0, world)
    ctxs, streams, frames, packeds, gathereds = [], [], [], [], []
    for _ in range(lanes):
        c = capi.Context(local_rank, **{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.ctx_option})
        c.upload_scene(scene)
        c.set_features(features)
        ctxs.append(c)
        streams.append(torch.cuda.Stream(dev))
        frames.append(torch.empty(width * height * 3, dtype=torch.float32, device=dev))
        packeds.append(torch.zeros(stride, dtype=torch.float32, device=dev) if world > 1 else None)
        gathereds.append(torch.empty((world, stride), dtype=torch.float32, device=dev) if world > 1 and rank == 0 else None)
    ctx = ctxs[0]

    def step(i):
        k = i % lanes
        c, frame = ctxs[k], frames[k]
        with torch.cuda.stream(streams[k]):
            sptr = streams[k].cuda_stream
            if world == 1:
                c.render_device(camera, width, height, frame.data_ptr(), frame.numel(), 0, 1, sptr)
            else:
                c.render_device(camera, width, height, packeds[k].data_ptr(), packeds[k].numel(), rank, world, sptr)
                sharding.gather_packed(packeds[k], rank, world, dist, gathereds[k])
                if rank == 0:
                    c.deinterleave(gathereds[k].data_ptr(), width, height, world, frame.data_ptr(), sptr)
                    c.postprocess(frame.data_ptr(), width, height, sptr)  # bloom needs the whole frame (no-op when off)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def drain_stats():
        tot = {"kernel_ms": 0.0, "path_ms": 0.0, "shadow_ms": 0.0, "launches": 0, "batches": 0, "rays": 0}
        for k in range(lanes):
            st = ctxs[k].sync(streams[k].cuda_stream)
            for key in ("kernel_ms", "path_ms", "shadow_ms", "launches", "batches"):
                tot[key] += st[key]
            tot["rays"] = max(tot["rays"], st["rays"])  # rays of one frame (the same on every lane)
        return tot

    # set-up, not a step: every context sizes and allocates its workspace on its first frame
    for k in range(lanes):
        step(k)
    fence()
    for i in range(args.warmup):
        step(i)
    fence()
    drain_stats()  # drop warm-up launches from the kernel-time sums
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    t_enqueued = time.perf_counter() - t0  # host side only: how long it took to hand the frames to the device
    fence()
    dt = time.perf_counter() - t0
    st = drain_stats()  # rays of one frame, summed kernel times of the timed launches
    nl = max(1, st["launches"])
    frame_ms, path_ms, shadow_ms = st["kernel_ms"] / nl, st["path_ms"] / nl, st["shadow_ms"] / nl
    # a frame whose working set exceeds the workspace is rendered in several pixel batches: that many
    # launches of every kernel per frame
    per_frame = max(1, st["batches"] // nl)

    dt_max, rays_total = reduce_over_ranks(dt, float(st["rays"]), world, dist, dev)
    ranks_seen = dist.get_world_size() if world > 1 else 1  # what the collective library itself counts

    # one frame on its own (outside the timed region): the kernels' durations when they do not share
    # the compute units with a neighbouring frame's kernels
    ctx.render_device(camera, width, height, (packeds[0] if world > 1 else frames[0]).data_ptr(),
                      (packeds[0] if world > 1 else frames[0]).numel(), rank if world > 1 else 0, world, streams[0].cuda_stream)
    alone = ctx.sync(streams[0].cuda_stream)
    # one more launch with the node / triangle counters on (outside the timed region)
    tmp = packeds[0] if world > 1 else frames[0]
    cst = ctx.render_device(camera, width, height, tmp.data_ptr(), tmp.numel(), rank if world > 1 else 0, world,
                            streams[0].cuda_stream, collect_stats=True)

    # N > 1: the frame rank 0 put together from the gathered shards against the same frame rendered whole by one
    # rank (pixels are independent, so the two must agree bit for bit); outside the timed region
    frame_equal = None
    if world > 1 and rank == 0:
        last = (args.warmup + args.steps - 1) % lanes
        whole = torch.empty(width * height * 3, dtype=torch.float32, device=dev)
        ctx.render_device(camera, width, height, whole.data_ptr(), whole.numel(), 0, 1, streams[0].cuda_stream)
        ctx.sync(streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        frame_equal = bool(torch.equal(whole.view(torch.int32), frames[last].view(torch.int32)))

    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        value = rays_total * args.steps / dt_max / 1e6
        # Dominant kernel: the shadow-ray kernel of the wavefront path (the whole per-pixel
        # kernel when a frame has to take that path).  Algorithmic traffic of one launch on
        # this rank: 64 B per sibling-pair record fetched, 36 B of vertex positions per
        # triangle tested, 28 B ray + 16 B hit record per ray.
        if shadow_ms > 0:
            kname, k_ms = "shadowTraceKernel", shadow_ms / per_frame
            k_rays, k_inner, k_tris = (cst[k] / per_frame for k in ("shadow_rays", "shadow_inner_nodes", "shadow_tri_tests"))
        else:
            kname, k_ms = "renderKernel", frame_ms
            k_rays, k_inner, k_tris = cst["rays"], cst["inner_nodes"], cst["tri_tests"]
        k_alone = (alone["shadow_ms"] / per_frame if shadow_ms > 0 else alone["kernel_ms"]) / max(1, alone["launches"])
        t_alone = max(k_alone, 1e-9) * 1e-3  # seconds one launch takes with the chip to itself
        # What bounds the kernel, as measured (profiles/round3_c_sensitivity.txt: work of one kind added to every
        # traversal step): the vector memory path.  Every lane's 16-byte load is one lookup in the compute unit's
        # L1, which serves about one a clock (tools/ta_probe.hip: 1.10 per CU-clock for two loads per 32-byte
        # record out of an L1-resident table, 0.78 out of an L2-resident one); added loads cost nearly in
        # proportion, added vector or scalar instructions a tenth of theirs.  So `achieved` / `peak` are lookups
        # per second against the probe's L1-resident rate.  The vector ALU's own figure rides along
        # (`valu_issue`): wave-instructions issued against the rate the probe measures for the instruction mix
        # of one traversal step at the kernel's six waves per SIMD (tools/valu_probe.hip: multiply-adds, adds,
        # moves and bit operations issue every 2.2-2.4 clocks of a SIMD, permutes, minima, compares, selects
        # and conversions every 4.1; one wave alone issues every 5-8 clocks whatever the instruction).
        # Counters (TCP_TOTAL_CACHE_ACCESSES, SQ_INSTS_VALU) cannot be collected inside this process: they come
        # from the committed counter pass over this same launch and count only if it ran on these kernel
        # sources; the time is this run's HIP events around the launch on a stream it has to itself.
        ctrs, ctr_src, ctr_fresh = stored_counters(args.workload, kname, world, width, height, args.camera_distance)
        l1_peak = L1_LOOKUPS_PER_CU_CLOCK * num_cu * SHADER_CLOCK_GHZ  # G lookups / s
        roof = {"bound": "l1_lookups", "unit": "G lookups/s", "peak": round(l1_peak, 1),
                "peak_is": f"{num_cu} compute units x {L1_LOOKUPS_PER_CU_CLOCK} sixteen-byte lane-loads per clock (measured, "
                           f"profiles/round2_a_ta_probe.txt, L1-resident table) x {SHADER_CLOCK_GHZ} GHz",
                "achieved": None, "frac": None, "counters": ctr_src, "traffic": None}
        if ctrs and not ctr_fresh:
            roof["stale_counters"] = "the stored counter pass ran on other kernel sources: no achieved / frac from it"
        if ctrs and ctr_fresh:
            if ctrs.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                acc = ctrs["TCP_TOTAL_CACHE_ACCESSES_sum"]
                roof["achieved"] = round(acc / t_alone / 1e9, 2)
                roof["frac"] = round(acc / t_alone / 1e9 / l1_peak, 4)
                roof["l1_lookups_per_ray"] = round(acc / max(1.0, k_rays), 1)
                roof["l1_lookups_per_cu_clock"] = round(acc / t_alone / (num_cu * SHADER_CLOCK_GHZ * 1e9), 3)
                if ctrs.get("TCP_TCC_READ_REQ_sum"):
                    roof["l1_miss_rate"] = round(ctrs["TCP_TCC_READ_REQ_sum"] / acc, 3)
            if ctrs.get("SQ_INSTS_VALU"):
                valu = ctrs["SQ_INSTS_VALU"]
                vp = valu_probe_peak()
                vi = {"unit": "G wave-instructions/s", "achieved": round(valu / t_alone / 1e9, 2),
                      "valu_wave_instructions_per_launch": valu, "valu_wave_instructions_per_ray": round(valu / max(1.0, k_rays), 2)}
                if vp:
                    vi.update({"peak": vp["chip_G_wave_instructions_per_s"], "frac": round(valu / t_alone / 1e9 / vp["chip_G_wave_instructions_per_s"], 4),
                               "peak_is": f"measured: '{vp['stream']}' at {vp['waves_per_simd']} waves per SIMD, {vp['per_simd_clock']} "
                                          f"wave-instructions per SIMD-clock at {vp['shader_clock_GHz']} GHz (profiles/valu_probe_gfx950.json)"})
                if ctrs.get("SQ_THREAD_CYCLES_VALU"):
                    vi["lanes_per_valu"] = round(ctrs["SQ_THREAD_CYCLES_VALU"] / valu, 1)
                roof["valu_issue"] = vi
            if "fetch_bytes" in ctrs and "write_bytes" in ctrs:
                roof["traffic"] = ctrs["fetch_bytes"] + ctrs["write_bytes"]
        # The two figures the scope table asks for, on the same launch alone.  Algorithmic traffic: 64 B per
        # sibling-pair record visited, 36 B of vertex positions per triangle tested, 28 B ray + 16 B result per
        # ray -- bytes the algorithm asks for, served by L1 / L2 (the scene is a few MB), which is why they
        # exceed what HBM could deliver; the HBM-side bytes are the counters' `traffic`.
        alg_bytes = 64.0 * k_inner + 36.0 * k_tris + 44.0 * k_rays
        alg_flops = 26.0 * (k_rays + 2.0 * k_inner) + 50.0 * k_tris
        roof.update({
            "kernel": kname, "kernel_ms": round(k_alone, 4), "launches_per_frame": per_frame,
            "kernel_ms_in_timed_region": round(k_ms, 4),
            "timing_note": "kernel_ms: HIP events around the launch, one frame at a time; in the timed region frames overlap "
                           "(frames_in_flight), so a launch there shares the compute units and its event time is no denominator",
            "frame_device_ms_alone": round(alone["kernel_ms"] / max(1, alone["launches"]), 4),
            "kernel_rays_per_launch": int(k_rays), "kernel_mrays_per_s": round(k_rays / t_alone / 1e6, 1),
            "inner_nodes_per_ray": round(k_inner / max(1.0, k_rays), 3), "tri_tests_per_ray": round(k_tris / max(1.0, k_rays), 3),
            "hbm": {"peak_GBs": HBM_PEAK_GBS,
                    "measured_GBs": round(roof["traffic"] / t_alone / 1e9, 1) if roof["traffic"] else None,
                    "measured_frac": round(roof["traffic"] / t_alone / 1e9 / HBM_PEAK_GBS, 4) if roof["traffic"] else None,
                    "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_ray": round(alg_bytes / max(1.0, k_rays), 1),
                    "algorithmic_GBs": round(alg_bytes / t_alone / 1e9, 1),
                    "note": "the algorithmic bytes are cache-served; HBM is not this kernel's roof"},
            "fp32": {"achieved_tflops": round(alg_flops / t_alone / 1e12, 3), "peak_tflops": FP32_PEAK_TFLOPS,
                     "frac": round(alg_flops / t_alone / 1e12 / FP32_PEAK_TFLOPS, 4),
                     "note": "peak counts packed fused multiply-adds on all lanes (4 flops per lane and instruction); a scalar-per-lane "
                             "compare or permute is worth a quarter of that at best"},
            "frame_device_ms_in_timed_region": {"all_kernels": round(frame_ms, 4), "cameraKernel+pathKernel": round(path_ms, 4),
                                                "shadowTraceKernel": round(shadow_ms, 4)}})
        out = {
            "metric": "Mrays/s (whole node) on dragon.obj 1080p 16spp; ms/frame" if args.workload == "dragon"
                      else f"Mrays/s ({args.workload})",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "world_size": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "rays_per_frame": int(rays_total),
                       "tile_sharding": f"8x8 tiles interleaved over {world} GPU(s)", "frames_in_flight": lanes,
                       **({"gathered_frame_equals_one_rank_frame": frame_equal} if frame_equal is not None else {}), **({"ctx_options": args.ctx_option} if args.ctx_option else {}),
                       "scene_note": "dragon.obj is absent from the reference snapshot; procedural stand-in with the same triangle count"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu:
            scale = args.cpu_scale
            if scale <= 0:
                scale = 1
                while rays_total / (scale * scale) > 1.2e8:
                    scale *= 2
            cb = cpu_baseline(scene, features, width, height, scale, args.camera_distance)
            if scale == 1:  # same frame on both sides: the two ray counts have to be the same number
                cb["rays_equal_device"] = cb["rays"] == int(rays_total)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
