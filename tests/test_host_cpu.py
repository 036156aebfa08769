"""Host-side logic that needs no GPU: the C ABI loads and exports every
declared symbol, the host BVH build equals the oracle's bit for bit, error
behaviour, scene ingest, and the world-size-2 tile sharding over gloo."""
import ctypes
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import common
import orc
from common import sio

capi = importlib.import_module("ray-tracer_amd.capi")
sharding = importlib.import_module("ray-tracer_amd.sharding")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(common.ROOT, "include", "rt_amd.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|size_t|const char\*)\s+(rt_[a-z_]+)\s*\(", header, re.M))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None


@pytest.mark.parametrize("name", ["SingleTriangle", "Cube", "CornellBox", "Monkey", "Teapot", "Dragon"])
def test_host_bvh_build_equals_oracle(name):
    prims = common.scene(name).primitives()
    n1, p1, l1, f1 = capi.bvh_build(prims)
    n2, p2, l2, f2 = orc.bvh_build(prims)
    assert (l1, f1) == (l2, f2)
    assert n1.tobytes() == n2.tobytes()
    assert p1.tobytes() == p2.tobytes()


def test_host_bvh_build_edge_cases():
    nodes, prims, levels, leaves = capi.bvh_build(np.zeros(0, dtype=sio.PRIMITIVE))
    assert len(nodes) == 2 and len(prims) == 0 and (levels, leaves) == (1, 1)
    assert nodes[0]["data"].tolist() == [sio.LEAF_BIT, 0]
    # degenerate: 9 identical triangles (all centroid keys tie)
    p = np.zeros(9, dtype=sio.PRIMITIVE)
    p["v1"]["position"] = (1, 0, 0)
    p["v2"]["position"] = (0, 1, 0)
    p["meshID"] = np.arange(9)
    n1, p1, _, _ = capi.bvh_build(p)
    n2, p2, _, _ = orc.bvh_build(p)
    assert n1.tobytes() == n2.tobytes() and p1.tobytes() == p2.tobytes()
    # the same degenerate case through the SAH + binning split: no extent along the axis
    n1, p1, _, _ = capi.bvh_build(p, sah_binning=True)
    n2, p2, _, _ = orc.bvh_build(p, sah_binning=True)
    assert n1.tobytes() == n2.tobytes() and p1.tobytes() == p2.tobytes()


@pytest.mark.parametrize("name", ["Cube", "CornellBox", "Monkey", "Teapot", "Dragon"])
def test_host_sah_binned_bvh_equals_oracle(name):
    """Features::extra.enableBvhSahBinning: splitPrimitivesBySAHBin (src/extra.cpp:493-543) inside BVH::BVH.
    The reference holds no test or stored tree for it (parity with the reference itself: unpinned);
    the library's index-based builder against the oracle's statement-by-statement form."""
    prims = common.scene(name).primitives()
    n1, p1, l1, f1 = capi.bvh_build(prims, sah_binning=True)
    n2, p2, l2, f2 = orc.bvh_build(prims, sah_binning=True)
    assert (l1, f1) == (l2, f2)
    assert n1.tobytes() == n2.tobytes() and p1.tobytes() == p2.tobytes()
    # a permutation of the input, every leaf within LeafSize, every primitive inside its leaf's box
    rows = lambda a: np.sort(np.frombuffer(a.tobytes(), dtype="S100"))
    assert np.array_equal(rows(p1), rows(prims))
    data = n1["data"]
    is_leaf = (data[:, 0] & sio.LEAF_BIT) != 0
    is_leaf[1] = False  # the unused node that keeps sibling pairs adjacent
    counts = data[is_leaf, 1]
    assert int(counts.sum()) == len(prims) and counts.min() >= 1 and counts.max() <= 4
    nm, _, lm, _ = capi.bvh_build(prims)
    assert n1.tobytes() != nm.tobytes() or len(prims) <= 4  # not the median tree


def test_fast_division_is_exact(tmp_path):
    """The multiply-high division the kernels use for per-launch divisors (rays per pixel, light samples
    per hit, lens samples, tiles per row) against the machine's own: every divisor up to 70 000 with edge
    and random dividends, and random divisors of every magnitude."""
    exe = str(tmp_path / "fastdiv_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(common.ROOT, "tests", "cpp", "fastdiv_check.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("0 wrong"), out.stdout


def test_device_sine_and_cosine_equal_the_c_librarys(tmp_path):
    """The lens and glossy samples take the sine and cosine of an angle in [0, 2 pi]
    (src/extra.cpp:27-44, :353-356).  The header the kernels use evaluates them the way the host's C
    library does; compiled for the host it must give the library's bits for every float in that range."""
    exe = str(tmp_path / "trig_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(common.ROOT, "tests", "cpp", "trig_check.cpp")],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith(" 0 wrong"), out.stdout


def test_device_power_function_equals_the_c_librarys(tmp_path):
    """The specular terms raise a cosine to the shininess with the C library's powf
    (src/shading.cpp:92-153).  The header the kernels use evaluates it the way glibc does; compiled for
    the host it must give the library's bits: 2e8 pairs of cosines and shininess values plus the
    special arguments.  (A C library built without fused multiply-adds rounds a few intermediate doubles
    differently, some three results in 1e8: that many are let through.)"""
    exe = str(tmp_path / "pow_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(common.ROOT, "tests", "cpp", "pow_check.cpp")],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    total, wrong = (int(v) for v in out.stdout.strip().splitlines()[-1].replace(" pairs,", "").replace(" wrong", "").split())
    assert total > 200000000 and wrong <= 20, out.stdout


def test_shard_floats_matches_python_layout():
    for (w, h) in [(1920, 1080), (17, 9), (8, 8), (3840, 2160), (100, 1)]:
        for world in (1, 2, 3, 4, 8):
            for r in range(world):
                assert capi.shard_floats(w, h, r, world) == sharding.shard_floats(w, h, r, world)
    assert capi.shard_floats(0, 5, 0, 1) == 0 and capi.shard_floats(8, 8, 2, 2) == 0


def test_pack_and_deinterleave_round_trip():
    rng = np.random.default_rng(0)
    for (w, h, world) in [(17, 9, 2), (64, 40, 3), (8, 8, 8), (33, 31, 4)]:
        frame = rng.random((h, w, 3), dtype=np.float32)
        packed = np.stack([sharding.pack_shard(frame, r, world) for r in range(world)])
        assert np.array_equal(sharding.deinterleave(packed, w, h, world), frame)


def test_obj_loader_submeshes_and_materials():
    s = common.scene("CornellBox")
    assert len(s.meshes) == 8 and sum(len(m.triangles) for m in s.meshes) == 32
    c = common.scene("Cube")
    assert len(c.meshes) == 6 and abs(float(c.meshes[0].material["transparency"]) - 0.452632) < 1e-6
    t = common.scene("Teapot")
    assert t.meshes[0].material["kdTexture"] == 0 and t.images[0].shape == (128, 128, 3)
    assert np.allclose(t.meshes[0].material["ks"], 0.2) and float(t.meshes[0].material["shininess"]) == 10.0
    pos = np.concatenate([m.vertices["position"] for m in t.meshes])
    assert abs(np.linalg.norm(pos, axis=1).max() - 1.0) < 1e-5  # centre + unit scale
    with pytest.raises(FileNotFoundError):
        sio.load_mesh(os.path.join(common.DATA, "dragon.obj"))


def test_dragon_standin_is_deterministic():
    a, b = sio.dragon_standin(), sio.dragon_standin()
    assert len(a.triangles) == 87130
    assert a.vertices.tobytes() == b.vertices.tobytes() and a.triangles.tobytes() == b.triangles.tobytes()


GLOO_WORKER = r"""
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [os.environ["RT_ROOT"], os.path.join(os.environ["RT_ROOT"], "oracle"), os.path.join(os.environ["RT_ROOT"], "tests")]
import common, orc
from common import sio
sharding = importlib.import_module("ray-tracer_amd.sharding")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w, h = 52, 30
scn = common.scene("CornellBox")
f = common.features(enableShading=1, enableReflections=1, enableShadows=1, shadingModel=2)
full, _ = orc.OracleScene(scn).render(f, sio.default_camera(w, h), w, h, threads=1)
# each rank contributes only its own tiles of the frame
packed = torch.from_numpy(sharding.pack_shard(full, rank, world))
stride = sharding.shard_floats(w, h, 0, world)
assert packed.numel() == stride
gathered = torch.empty((world, stride), dtype=torch.float32) if rank == 0 else None
sharding.gather_packed(packed, rank, world, dist, gathered)
if rank == 0:
    frame = sharding.deinterleave(gathered.numpy(), w, h, world)
    assert np.array_equal(frame.view(np.uint32), full.view(np.uint32))
    print("GLOO_OK")
dist.barrier()
dist.destroy_process_group()
"""


def test_world_size_2_tile_gather_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, RT_ROOT=common.ROOT, OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29653", str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO_OK" in out.stdout


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: the parent starts torch.distributed.run
    as a child before touching the GPU, the ranks rendezvous (gloo here), gather their tiles, reduce
    their timings and rank 0 prints one JSON line naming the ranks the collective library saw."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", str(n), "--launch-check", "--steps", "2"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    assert lines[0]["launch_check"] is True and lines[0]["world_size"] == n and lines[0]["n_gpus"] == n
    assert lines[0]["tiles"] == 13 * 7


def test_bench_refuses_a_launcher_of_another_size():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "4", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "4" in out.stderr


def test_stored_counters_and_probe_peaks_belong_to_the_kernels_as_they_are():
    """bench.py's roofline takes hardware counters from a committed rocprofv3 pass (it cannot collect them inside its
    own process) and the vector ALU's peak from a committed probe run.  The counters only count if the pass ran on the
    kernel sources as they are now: a pass on other sources must yield no fraction at all, and the committed one must
    be current."""
    import bench
    ctrs, src, fresh = bench.stored_counters("dragon", "shadowTraceKernel", 1, 1920, 1080)
    assert ctrs and fresh, src
    assert ctrs["SQ_INSTS_VALU"] > 1e9 and ctrs["TCP_TOTAL_CACHE_ACCESSES_sum"] > 1e9
    peak = bench.valu_probe_peak()
    assert peak and peak["waves_per_simd"] == 6 and 300.0 < peak["chip_G_wave_instructions_per_s"] < 1300.0
    # no pass for another resolution, camera or world size
    assert bench.stored_counters("dragon", "shadowTraceKernel", 1, 960, 540)[0] is None
    assert bench.stored_counters("dragon", "shadowTraceKernel", 2, 1920, 1080)[0] is None
    assert bench.stored_counters("dragon", "shadowTraceKernel", 1, 1920, 1080, 1.6)[0] is None
