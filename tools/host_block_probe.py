"""How long rt_render_device holds the calling thread on a frame of many pixel batches (configuration 5's
features), with the per-launch timing events on and off.
  python tools/host_block_probe.py [width height]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

capi = importlib.import_module("ray-tracer_amd.capi")
sio = importlib.import_module("ray-tracer_amd.scene_io")


def main():
    scene, features, width, height, desc = bench.build_workload("dragon4k")
    if len(sys.argv) > 2:
        width, height = int(sys.argv[1]), int(sys.argv[2])
    cam = sio.default_camera(width, height)
    dev = torch.device("cuda", 0)
    frame = torch.empty(width * height * 3, dtype=torch.float32, device=dev)
    for timing in (1, 0):
        c = capi.Context(0, timing=timing)
        c.upload_scene(scene)
        c.set_features(features)
        s = torch.cuda.Stream(dev)
        c.render_device(cam, width, height, frame.data_ptr(), frame.numel(), 0, 1, s.cuda_stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        c.render_device(cam, width, height, frame.data_ptr(), frame.numel(), 0, 1, s.cuda_stream)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        t_all = time.perf_counter() - t0
        st = c.sync(s.cuda_stream)
        print(json.dumps({"frame": [width, height], "timing_events": timing, "host_ms_in_call": round(t_host * 1e3, 1),
                          "frame_ms": round(t_all * 1e3, 1), "batches": st.get("batches"), "launches": st.get("launches")}))
        c.close()


if __name__ == "__main__":
    main()
