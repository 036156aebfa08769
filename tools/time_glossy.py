"""Glossy reflections (src/extra.cpp:329-368) at 1920x1080 on the two renderers that take them: the queue-based one
(tree kernel + shared light-sample kernels) and the per-pixel kernel.  Device frame time (kernel_ms of rt_render_device),
ray counts, and a bit comparison of the two frames.
  python tools/time_glossy.py [scene] [glossy samples] [depth cap] [pixel samples] [area light: 0/1] [transparency of every other mesh, 1 = opaque]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import common  # noqa: E402
from common import sio  # noqa: E402

capi = importlib.import_module("ray-tracer_amd.capi")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "CornellBox"
    g = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    area = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    clear = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
    w, h = 1920, 1080
    import copy
    scn = copy.deepcopy(common.scene(name))
    if clear != 1.0:
        for i, m in enumerate(scn.meshes):
            if i % 2:
                m.material["transparency"] = clear
    if area:
        scn.lights = [sio.cornell_parallelogram_light()]
    f = common.features(enableShading=1, enableShadows=1, enableReflections=1, shadingModel=2, enableGlossyReflection=1,
                        numGlossySamples=g, glossyExponent=0.5, maxRayDepth=cap, numPixelSamples=spp, numShadowSamples=4,
                        enableTransparency=int(clear != 1.0))
    cam = sio.default_camera(w, h)
    out = {"scene": name, "frame": [w, h], "glossy_samples": g, "depth_cap": cap, "pixel_samples": spp, "area_light": bool(area), "transparency": clear}
    frames = {}
    for label, opts in (("queues", dict(per_pixel=0)), ("per_pixel", dict(per_pixel=1))):
        c = capi.Context(0, **opts)
        c.upload_scene(scn)
        c.set_features(f)
        buf = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        for _ in range(2):
            c.render_device(cam, w, h, buf.data_ptr(), buf.numel(), 0, 1, 0)
        c.sync(0)
        ms = []
        for _ in range(5):
            c.render_device(cam, w, h, buf.data_ptr(), buf.numel(), 0, 1, 0)
            st = c.sync(0)
            ms.append(st["kernel_ms"] / max(1, st["launches"]))
        frames[label] = buf.cpu().numpy()
        out[label] = {"ms": round(float(np.median(ms)), 3), "rays": st["rays"], "Mrays_s": round(st["rays"] / np.median(ms) / 1e3, 1),
                      "batches": st["batches"] // max(1, st["launches"])}
        c.close()
    out["frames_bit_equal"] = bool(np.array_equal(frames["queues"].view(np.uint32), frames["per_pixel"].view(np.uint32)))
    out["rays_equal"] = out["queues"]["rays"] == out["per_pixel"]["rays"]
    out["speedup"] = round(out["per_pixel"]["ms"] / out["queues"]["ms"], 2)
    print(json.dumps(out))


main()
