"""Motion blur (renderImageWithMotionBlur): the device's frame time beside the oracle's, which does what the
reference does (a tree built over a moved copy of the scene per time sample per pixel), and a bit comparison.
  python tools/time_motion_blur.py [scene] [width] [height] [samples]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import orc  # noqa: E402
from common import sio  # noqa: E402

capi = importlib.import_module("ray-tracer_amd.capi")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Monkey"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    samples = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    big = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1920, 1080)
    scn = common.scene(name)
    f = common.features(enableMotionBlur=1, motionBlurSamples=samples, movement=0.5, enableShading=1, enableShadows=1,
                        enableReflections=1, enableNormalInterp=1, shadingModel=2, numPixelSamples=1)
    cam = sio.default_camera(w, h, distance=5.0)
    ctx = capi.Context(0)
    ctx.upload_scene(scn)
    ctx.set_features(f)
    got, sg = ctx.render(cam, w, h)  # warm-up
    t0 = time.perf_counter()
    got, sg = ctx.render(cam, w, h)
    dev_s = time.perf_counter() - t0
    threads = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    want, so = orc.OracleScene(scn).render_motion_blur(scn.primitives(), f, cam, w, h, threads=threads)
    cpu_s = time.perf_counter() - t0
    out = {"scene": name, "triangles": int(len(scn.primitives())), "frame": [w, h], "time_samples": samples,
           "rays": sg["rays"], "rays_equal": sg["rays"] == so["rays"],
           "frames_bit_equal": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))),
           "device_ms": round(dev_s * 1e3, 3), "oracle_s": round(cpu_s, 3), "oracle_threads": threads,
           "device_Mrays_s": round(sg["rays"] / dev_s / 1e6, 1), "oracle_Mrays_s": round(so["rays"] / cpu_s / 1e6, 3)}
    # a full-size frame on the device alone (the oracle would need hours)
    bw, bh = big
    cam = sio.default_camera(bw, bh, distance=5.0)
    ctx.render(cam, bw, bh)
    t0 = time.perf_counter()
    _, sb = ctx.render(cam, bw, bh)
    big_s = time.perf_counter() - t0
    out["device_full_frame"] = {"frame": [bw, bh], "ms": round(big_s * 1e3, 2), "rays": sb["rays"],
                                "Mrays_s": round(sb["rays"] / big_s / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
