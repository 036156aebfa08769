"""One drawn configuration of tests/test_gpu_random_configs.py under every tree / renderer option: ray counts and the
pixels that differ from the oracle.   python tools/random_config_debug.py <seed>"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import orc  # noqa: E402
import test_gpu_random_configs as t  # noqa: E402

capi = importlib.import_module("ray-tracer_amd.capi")

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
name, scn, f, cam = t._draw(rng)
w, h = int(rng.integers(*t.W_RANGE)), int(rng.integers(*t.H_RANGE))
cam["aspect"] = np.float32(w) / np.float32(h)
sah = bool(f["enableBvhSahBinning"])
nodes, prims, _, _ = orc.bvh_build(scn.primitives(), sah)
osc = orc.OracleScene(scn, nodes, prims)
motion = bool(f["enableMotionBlur"]) and not bool(f["enableDepthOfField"])
want, so = osc.render_motion_blur(scn.primitives(), f, cam, w, h) if motion else osc.render(f, cam, w, h)
print(name, w, h, {k: f[k].item() for k in f.dtype.names if f[k]}, "lights", [int(l["type"]) for l in scn.lights])
print("materials", [(round(float(m.material["transparency"]), 2), tuple(np.round(m.material["ks"], 2))) for m in scn.meshes])
print("camera", cam)
print("oracle rays", so["rays"])
for tree in (0, 1, 2):
    for pp in (0, 1):
        ctx = capi.Context(0, tree=tree, per_pixel=pp)
        ctx.upload_scene(scn, sah_binning=sah)
        ctx.set_features(f)
        got, sg = ctx.render(cam, w, h)
        ctx.close()
        bad = np.argwhere((got.view(np.uint32) != want.view(np.uint32)).any(axis=2))
        print(f"tree {tree} per_pixel {pp}: rays {sg['rays']} differing pixels (row, col) {bad.tolist()[:8]}",
              [(got[r, c].tolist(), want[r, c].tolist()) for r, c in bad[:2]])

# every intersect call the oracle makes for one pixel, replayed on the device ray by ray (plain and strict)
if len(sys.argv) >= 4:
    x, y = int(sys.argv[2]), int(sys.argv[3])
    sio = common.sio
    path = "/tmp/orc_dump.txt"
    os.environ["ORC_DUMP_PIXEL"] = f"{x} {y}"
    os.environ["ORC_DUMP_FILE"] = path
    osc.render(f, cam, w, h, threads=1)
    rows = [l.split() for l in open(path)]
    bits = np.array([[int(v, 16) for v in r[:8]] for r in rows], dtype=np.uint32).view(np.float32)
    hit_o = np.array([int(r[8]) for r in rows])
    prim_o = np.array([int(r[9]) for r in rows])
    rays = np.zeros(len(rows), dtype=sio.RAY)
    rays["origin"] = bits[:, 0:3]
    rays["direction"] = bits[:, 3:6]
    rays["t"] = bits[:, 6]
    print(len(rows), "intersect calls in pixel", x, y)
    for strict in (0, 1):
        c = capi.Context(0, tree=2, strict_intersect=strict)
        c.upload_scene(scn, sah_binning=sah)
        c.set_features(f)
        r, hits = c.intersect(rays)
        bad = np.nonzero((hits["hit"] != hit_o) | (hits["primitive"] != prim_o) | (r["t"].view(np.uint32) != bits[:, 7].view(np.uint32)))[0]
        print("strict", strict, "differing calls:", len(bad))
        for i in bad[:6]:
            print(" call", i, "o", rays["origin"][i], "d", rays["direction"][i], "t in", rays["t"][i], "| oracle hit", hit_o[i], prim_o[i], bits[i, 7],
                  "| device hit", hits["hit"][i], hits["primitive"][i], r["t"][i])
        c.close()
    print("nodes", [(n["lower"].tolist(), n["upper"].tolist(), n["data"].tolist()) for n in nodes])
