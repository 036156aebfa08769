"""rt_render (the frame handed back in a host buffer) against rt_render_device (the frame stays in HBM) on the
headline workload: what the copy to the host adds to a frame.
  python tools/host_buffer_rate.py [workload] [frames]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

capi = importlib.import_module("ray-tracer_amd.capi")
sio = importlib.import_module("ray-tracer_amd.scene_io")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scene, features, w, h, desc = bench.build_workload(name)
cam = sio.default_camera(w, h)
c = capi.Context(0)
c.upload_scene(scene)
c.set_features(features)
dev = torch.empty(w * h * 3, dtype=torch.float32, device="cuda")
pinned = torch.empty(w * h * 3, dtype=torch.float32).pin_memory()

for _ in range(3):
    _, st = c.render(cam, w, h)
rays = st["rays"]
t0 = time.perf_counter()
for _ in range(frames):
    c.render(cam, w, h)  # a fresh pageable numpy array per call, as a plain caller would pass
host_ms = (time.perf_counter() - t0) / frames * 1e3

for _ in range(3):
    c.render_device(cam, w, h, dev.data_ptr(), dev.numel())
c.sync()
t0 = time.perf_counter()
for _ in range(frames):
    c.render_device(cam, w, h, dev.data_ptr(), dev.numel())
    c.sync()
device_ms = (time.perf_counter() - t0) / frames * 1e3

# the frame kept on the device, then copied into pinned host memory by the caller
t0 = time.perf_counter()
for _ in range(frames):
    c.render_device(cam, w, h, dev.data_ptr(), dev.numel())
    pinned.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
pinned_ms = (time.perf_counter() - t0) / frames * 1e3

print(json.dumps({"workload": desc, "frames": frames, "rays_per_frame": rays, "frame_bytes": w * h * 12,
                  "rt_render_host_buffer_ms": round(host_ms, 3), "rt_render_host_buffer_mrays_s": round(rays / host_ms / 1e3, 1),
                  "rt_render_device_one_at_a_time_ms": round(device_ms, 3),
                  "rt_render_device_then_copy_to_pinned_ms": round(pinned_ms, 3),
                  "note": "one frame at a time on the null stream, each followed by a wait; rt_render copies into the caller's pageable memory"}))
